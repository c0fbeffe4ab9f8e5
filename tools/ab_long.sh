# A/B of the long-form (configs[4]) attention placement and of small-batch cluster forms
for e in "DEX_ATTN_SEPARATE=0" "DEX_ATTN_SEPARATE=1"; do
  env $e python bench.py --workload gedex_long --precision fp16 --graph on --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e gedex_long fp16', d['value'], d['ms_per_euler_step'])"
done
