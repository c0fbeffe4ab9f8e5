# Same-box A/B/C of three library builds: rocprofv3 rows of one kernel family + end to end.  bash tools/ab_libs3.sh "pattern" workload...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; PAT=$1; shift; O=$R/gpurun_out/ab_libs3.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  p=bf16; [ $w = gedex_long ] && p=fp16
  for which in A B C; do
    rm -rf /tmp/p_ab
    L=; [ $which != C ] && L="DEX_AMD_LIB=$R/tools/lib_$which.so"
    env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o t -- python $R/bench.py --workload $w --precision $p --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_ab -name '*kernel_stats.csv' | head -1)" $w $which "$PAT" >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} {sys.argv[3]}: kernel time {tot / 1e3:.0f} us in total")
for r in rows:
    if any(t in r["Name"] for t in sys.argv[4].split()):
        print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us")
PY
  done
  for rep in 1 2; do for which in A B C; do
    L=; [ $which != C ] && L="DEX_AMD_LIB=$R/tools/lib_$which.so"
    env $L python $R/bench.py --workload $w --precision $p --steps 6 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $p $which: %.1f frames/s' % d['value'])" >> $O
  done; done
done
cat $O
