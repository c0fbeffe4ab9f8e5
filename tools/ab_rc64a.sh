mkdir -p gpurun_out
run() { python bench.py --workload $1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"; }
run gedex_b32 default
DEX_ATTN_SEPARATE=1 DEX_ATTN_Q64=1 DEX_ROWCHAIN64=2 run gedex_b32 "sep+q64+rc64a"
DEX_ATTN_SEPARATE=1 DEX_ATTN_Q64=0 DEX_ROWCHAIN64=2 run gedex_b32 "sep+ring+rc64a"
run gedex_long default
DEX_ROWCHAIN64=2 run gedex_long "rc64"
run dex_b32 default
DEX_ROWCHAIN64A=0 run dex_b32 "a=0"
