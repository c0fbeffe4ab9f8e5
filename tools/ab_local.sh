# A/B of the XCD-local cluster form (hand-offs through one XCD's L2) against the cross-XCD cluster form
python -c "
import torch
from tests import gpu_util as U
cfg, eng, w = U.engine_for('gedex_lj')
print('xcd_local probe:', eng.xcd_local())
"
tools/clbench 650 1 | head -4
for w in gedex_b1 gedex_b2 gedex_b3 gedex_b1_t800; do
  for e in "DEX_DIT_CLUSTER_LOCAL=0" "DEX_DIT_CLUSTER_LOCAL=1"; do
    env $e python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $w', d['value'], d['ms_per_euler_step'])"
  done
done
