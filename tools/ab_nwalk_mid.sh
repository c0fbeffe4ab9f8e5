# The unpatchify GEMM at mid-size grids (long form: 79 row tiles): column walker with 4 / 8 / 16 column splits.  Usage: bash tools/ab_nwalk_mid.sh
cd /tmp; export TMPDIR=/tmp
for w in gedex_long gedex_b1; do for f in 0 8 16; do rm -rf /tmp/p_nw
prec=bf16; [ $w = gedex_long ] && prec=fp16
DEX_NWALK_SPLIT=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nw -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --precision $prec --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > /dev/null 2>&1
python - "$(find /tmp/p_nw -name '*kernel_stats.csv' | head -1)" $w $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} DEX_NWALK_SPLIT={sys.argv[3]} (0 = the launcher's rule): step {tot/150e3:.1f} us")
for r in rows:
    if any(t in r["Name"] for t in ("nwalk", "igemm_lp_ss")): print(f"   {r['Name'][:70]:70s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:7.1f} us")
PY
done; done
