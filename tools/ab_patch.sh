python -m pytest tests/test_gpu_cluster.py -m gpu -q --timeout 300 -k patch_embed 2>&1 | tail -4
for e in "DEX_PATCH_FUSED=0" "DEX_PATCH_FUSED=1" "DEX_PATCH_FUSED=0" "DEX_PATCH_FUSED=1"; do
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e gedex_b1', d['value'], d['ms_per_euler_step'])"
done
DEX_BENCH_TOPK=40 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k in d['kernels']:
    if 'patch' in k['kernel'] or 'pos' in k['kernel'] or 'qkv' in k['kernel']: print(k)"
