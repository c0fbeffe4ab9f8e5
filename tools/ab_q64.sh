#!/bin/bash
# A/B of the 64-queries-per-wave attention form end to end (same box, alternating): tools/ab_q64.sh > gpurun_out/ab_q64.txt
run() { # label, env..., workload, precision
  local label=$1; shift
  env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'frames/s', d['ms_per_euler_step'], 'ms/step')"
}
for rep in 1 2; do
for wl in "dex_b32 bf16" "gedex_long fp16" "gedex_b32 bf16" "dex_b32_t512 bf16"; do
  set -- $wl
  run "$1 q64=0        " DEX_ATTN_Q64=0 python bench.py --workload $1 --precision $2 --steps 4 --warmup 2 --no-profile --no-cpu-baseline --graph on
  run "$1 q64=auto     " python bench.py --workload $1 --precision $2 --steps 4 --warmup 2 --no-profile --no-cpu-baseline --graph on
  run "$1 q64=1 sep=1  " DEX_ATTN_Q64=1 DEX_ATTN_SEPARATE=1 python bench.py --workload $1 --precision $2 --steps 4 --warmup 2 --no-profile --no-cpu-baseline --graph on
done
done
