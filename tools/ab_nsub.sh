# A/B of the linear attention context pass's sub-tile count at batch size: DEX_LINATTN_NSUB=4 (the round-4 rule's answer at B = 32) against
# the round-aware choice.  Kernel rows under rocprofv3 + end-to-end values.  Usage (GPU box, repo root): bash tools/ab_nsub.sh [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/nsub; mkdir -p $O
B="--no-cpu-baseline --no-profile"
WL=${*:-gedex_b32 dex_b32}
: > $O/summary.txt
for w in $WL; do
  for f in 4 0; do
    rm -rf /tmp/p_ns_$f
    DEX_LINATTN_NSUB=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ns_$f -o t -- python $R/bench.py --workload $w --precision bf16 --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_ns_$f -name '*kernel_stats.csv' | head -1)" $w $f >> $O/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} DEX_LINATTN_NSUB={sys.argv[3]} (0 = round-aware rule): kernel time per Euler step {tot / 150e3:.1f} us")
for r in rows:
    n = r["Name"]
    if "linattn" in n:
        print(f"   {n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
    DEX_LINATTN_NSUB=$f python $R/bench.py --workload $w --precision bf16 --steps 8 --warmup 3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w bf16 nsub=$f: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O/summary.txt
  done
done
cat $O/summary.txt
