# A/B of the cluster form of the fused DiT block at the small batch sizes it covers (and one it does not)
for w in gedex_b1 gedex_b2 gedex_b3 gedex_b1_t800 dex_b1; do
  for e in "DEX_DIT_CLUSTER=0" "DEX_DIT_CLUSTER=1"; do
    env $e python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $w', d['value'], d['ms_per_euler_step'])"
  done
done
