# Same-box A/B of two builds of the library under rocprofv3 (kernel rows of the DiT block) + end to end:
#   bash tools/ab_lib_prof.sh <old.so> [workload ...]      (GPU box, repo root; results in gpurun_out/ab_lib_prof.txt)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OLD=$R/$1; shift; O=$R/gpurun_out/ab_lib_prof.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in ${*:-dex_b32}; do
  p=bf16; [ $w = gedex_long ] && p=fp16
  for which in old new; do
    rm -rf /tmp/p_ab
    L=; [ $which = old ] && L="DEX_AMD_LIB=$OLD"
    env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o t -- python $R/bench.py --workload $w --precision $p --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
    python - "$(find /tmp/p_ab -name '*kernel_stats.csv' | head -1)" $w $which >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]} {sys.argv[3]}: kernel time {tot / 1e3:.0f} us in total")
for r in rows:
    if any(t in r["Name"] for t in ("attn_q64", "attn_direct", "dit_rowchain")):
        print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us")
PY
  done
  for rep in 1 2; do for which in old new; do
    L=; [ $which = old ] && L="DEX_AMD_LIB=$OLD"
    env $L python $R/bench.py --workload $w --precision $p --steps 6 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $p $which: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O
  done; done
done
cat $O
