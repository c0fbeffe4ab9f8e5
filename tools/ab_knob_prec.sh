# like tools/ab_knob.sh with an explicit precision:  bash tools/ab_knob_prec.sh KNOB "v0 v1" "kernel-substring ..." precision workload [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; K=$1; VALS=$2; PAT=$3; P=$4; shift 4; O=$R/gpurun_out/ab_${K}_$P.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  for v in $VALS; do
    rm -rf /tmp/p_k
    env $K=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k -o t -- python $R/bench.py --workload $w --precision $P --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
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
done
cat $O
