# The ping-pong strip convolution's segment split at the long form (T = 4000): kernel rows + end-to-end value.  Usage: bash tools/ab_pp_long.sh
cd /tmp; export TMPDIR=/tmp
for w in gedex_long dex_b32 gedex_b32; do rm -rf /tmp/p_pp
prec=bf16; [ $w = gedex_long ] && prec=fp16
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_pp -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --precision $prec --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > /dev/null 2>&1
python - "$(find /tmp/p_pp -name '*kernel_stats.csv' | head -1)" $w <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]}: step {tot/150e3:.1f} us")
for r in rows:
    if "pp64" in r["Name"] or "stream64" in r["Name"]: print(f"   {r['Name'].replace('(anonymous namespace)::','')[:86]:86s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:7.1f} us")
PY
python $GRAFT_REPO_ROOT/bench.py --workload $w --precision $prec --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end', d['value'], d['ms_per_step'])"
done
