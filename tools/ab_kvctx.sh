# linattn_kvctx_kernel under rocprofv3 at the three regimes (one Euler step traced)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in gedex_b1 gedex_b32 gedex_long dex_b32; do
  rm -rf /tmp/pk_$w
  P=""; [ $w = gedex_long ] && P="--precision fp16"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$w -o t -- python $R/bench.py --workload $w $P --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > /dev/null 2>&1
  echo "== $w"; python - "$(find /tmp/pk_$w -name "*kernel_stats.csv" | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "kvctx" in r["Name"]:
        print(f'  {r["Name"][:70]:70s} x{r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
  python $R/tools/trace_step.py $(find /tmp/pk_$w -name "*kernel_trace.csv" | head -1) | grep "step:"
done
