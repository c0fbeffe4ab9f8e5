# Same-box A/B of the DEX adaptor work of round 5: the TV adaptor as one launch (DEX_TV_CHAIN) and the TIV adaptor folded into the patch
# embedding's load (DEX_TIV_FOLD) against the separate launches (both knobs 0): kernel rows under rocprofv3 + end-to-end bench values.
# Usage (GPU box, repo root): bash tools/ab_tv_chain.sh [workload ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tv_chain; mkdir -p $O
B="--no-cpu-baseline --no-profile"
WL=${*:-dex_b32}
: > $O/summary.txt
for w in $WL; do
  for f in 0 1; do
    rm -rf /tmp/p_tv_$f
    DEX_TV_CHAIN=$f DEX_TIV_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tv_$f -o t -- python $R/bench.py --workload $w --precision bf16 --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_tv_$f -name '*kernel_stats.csv' | head -1)" $w $f >> $O/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} DEX_TV_CHAIN=DEX_TIV_FOLD={sys.argv[3]}: kernel time {tot / 1e3:.0f} us for 150 Euler steps = {tot / 150e3:.1f} us per step")
for r in rows:
    n = r["Name"]
    if any(t in n for t in ("tv_", "tiv_", "attn_lp_shared", "igemm_lp_ss", "in_stats", "in_fold", "dwconv")):
        print(f"   {n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
    for prec in bf16 fp16 fp16x2; do
      DEX_TV_CHAIN=$f DEX_TIV_FOLD=$f python $R/bench.py --workload $w --precision $prec --steps 8 --warmup 3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $prec knobs=$f: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O/summary.txt
    done
  done
done
cat $O/summary.txt
