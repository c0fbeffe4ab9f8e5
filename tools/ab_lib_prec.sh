# Same-box A/B of two builds of the library, explicit precision, kernel rows + end to end:
#   bash tools/ab_lib_prec.sh <old.so> "kernel-substring ..." precision workload [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OLD=$R/$1; PAT=$2; P=$3; shift 3; O=$R/gpurun_out/ab_lib_$P.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  for which in old new; do
    rm -rf /tmp/p_ab
    L=; [ $which = old ] && L="DEX_AMD_LIB=$OLD"
    env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o t -- python $R/bench.py --workload $w --precision $P --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_ab -name '*kernel_stats.csv' | head -1)" $w $which "$PAT" >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
sel = sum(float(r["TotalDurationNs"]) for r in rows if any(t in r["Name"] for t in sys.argv[4].split()))
print(f"== {sys.argv[2]} {sys.argv[3]}: kernel time {tot / 1e3:.0f} us in total, selected kernels {sel / 1e3:.0f} us")
PY
  done
  for rep in 1 2; do for which in old new; do
    L=; [ $which = old ] && L="DEX_AMD_LIB=$OLD"
    env $L python $R/bench.py --workload $w --precision $P --steps 8 --warmup 3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $P $which: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O
  done; done
done
cat $O
