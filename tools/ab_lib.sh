#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <old.so> [workloads...]   (alternating runs, whole-call graph replay)
OLD=$1; shift; WL=${@:-"dex_b32 gedex_b32 gedex_long"}
one() { env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_euler_step'], 'ms/step', [(k['kernel'], k['avg_us']) for k in d.get('kernels', [])[:3]])"; }
for rep in 1 2; do for w in $WL; do p=bf16; [ $w = gedex_long ] && p=fp16
  echo -n "$w old: "; one DEX_AMD_LIB=$OLD python bench.py --workload $w --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-configs --graph on
  echo -n "$w new: "; one python bench.py --workload $w --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-configs --graph on
done; done
