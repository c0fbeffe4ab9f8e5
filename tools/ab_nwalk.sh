# one Euler step under rocprofv3: the unpatchify GEMM's launch time at DEX B = 32 and GeDEX B = 32
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in dex_b32 gedex_b32; do
  rm -rf /tmp/pn_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn_$w -o t -- python $R/bench.py --workload $w --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > /dev/null 2>&1
  echo "== $w"; grep -E "nwalk" $(find /tmp/pn_$w -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
  python $R/tools/trace_step.py $(find /tmp/pn_$w -name "*kernel_trace.csv" | head -1) | grep "step:"
done
