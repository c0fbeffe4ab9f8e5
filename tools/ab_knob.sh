# Same-box A/B of one launcher knob under rocprofv3 + end to end:  bash tools/ab_knob.sh KNOB "v0 v1" "kernel-substring ..." workload [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; K=$1; VALS=$2; PAT=$3; shift 3; O=$R/gpurun_out/ab_$K.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  p=bf16; [ $w = gedex_long ] && p=fp16
  for v in $VALS; do
    rm -rf /tmp/p_k
    env $K=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k -o t -- python $R/bench.py --workload $w --precision $p --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_k -name '*kernel_stats.csv' | head -1)" $w "$K=$v" "$PAT" >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} {sys.argv[3]}: kernel time {tot / 1e3:.0f} us in total")
for r in rows:
    if any(t in r["Name"] for t in sys.argv[4].split()):
        print(f"   {r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us")
PY
  done
  for rep in 1 2; do for v in $VALS; do
    env $K=$v python $R/bench.py --workload $w --precision $p --steps 6 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $p $K=$v: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O
  done; done
done
cat $O
