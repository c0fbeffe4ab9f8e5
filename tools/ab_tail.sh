#!/bin/bash
# A/B of the attention's tail split end to end (same box, alternating): tools/ab_tail.sh > gpurun_out/ab_tail.txt
run() { # label, env..., command
  local label=$1; shift
  env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'frames/s', d['ms_per_euler_step'], 'ms/step')"
}
for rep in 1 2 3; do
for wl in "dex_b32 bf16" "dex_esd_b32_n100 bf16"; do
  set -- $wl
  run "$1 tail=0" DEX_ATTN_Q64_TAIL=0 python bench.py --workload $1 --precision $2 --steps 4 --warmup 2 --no-profile --no-cpu-baseline --graph on
  run "$1 tail=1" DEX_ATTN_Q64_TAIL=1 python bench.py --workload $1 --precision $2 --steps 4 --warmup 2 --no-profile --no-cpu-baseline --graph on
done
done
