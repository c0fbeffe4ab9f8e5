# Same-box end-to-end sweep of launcher knobs in the split-weight mode (fp16x2) at B = 32: bash tools/sweep_fp16x2_knobs.sh workload...
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/sweep_fp16x2.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  for kv in "DEX_CONV_TH8:1 0" "DEX_CONV_W8:1 0" "DEX_LINATTN_NSUB:0 2 4 6" "DEX_POS_CT:0 1 2 3" "DEX_OUT2_MIN:256 2048" "DEX_CONV_SKIP_DEAD:1 0" "DEX_ROWCHAIN64A:1 0" "DEX_ATTN_Q64_HALF:1 0" "DEX_NWALK_SPLIT:0 1 2 4"; do
    K=${kv%%:*}; V=${kv#*:}
    for rep in 1 2; do for v in $V; do
      env $K=$v python bench.py --workload $w --precision fp16x2 --steps 4 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w fp16x2 $K=$v: %.1f' % d['value'])" >> $O
    done; done
  done
done
cat $O
