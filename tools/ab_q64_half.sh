# Same-box A/B of the half-unit plan of the 64-query attention (attention_q64.hip: DEX_ATTN_Q64_HALF=0 whole units only / 1 launcher's plan):
# kernel rows under rocprofv3 (attention as its own launch is the default at these shapes) + end-to-end bench values.
# Usage (GPU box, repo root): bash tools/ab_q64_half.sh [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/q64_half; mkdir -p $O
B="--no-cpu-baseline --no-profile --no-configs"
WL=${*:-dex_b32}
: > $O/summary.txt
for w in $WL; do
  for f in 0 1; do
    rm -rf /tmp/p_h_$f
    DEX_ATTN_Q64_HALF=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_h_$f -o t -- python $R/bench.py --workload $w --precision bf16 --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_h_$f -name '*kernel_stats.csv' | head -1)" $w $f >> $O/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} DEX_ATTN_Q64_HALF={sys.argv[3]}: kernel time {tot / 1e3:.0f} us in total")
for r in rows:
    n = r["Name"]
    if any(t in n for t in ("attn_q64", "attn_direct", "dit_rowchain")):
        print(f"   {n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  done
  for rep in 1 2; do for f in 0 1; do
    DEX_ATTN_Q64_HALF=$f python $R/bench.py --workload $w --precision bf16 --steps 6 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w bf16 half=$f: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O/summary.txt
  done; done
done
cat $O/summary.txt
