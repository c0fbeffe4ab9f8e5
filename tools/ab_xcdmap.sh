# A/B of the XCD-aware block order (K / V^T of one utterance fetched into one L2): speed, then bits
for w in gedex_b32 dex_b32 gedex_long; do
  for e in "DEX_XCD_MAP=0" "DEX_XCD_MAP=1"; do
    env $e python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $w', d['dtype'], d['value'], d['ms_per_euler_step'])"
  done
done
python - <<'P'
import os, numpy as np, torch
from tests import gpu_util as U
for name, kw in [("gedex_lj", dict(B=32, T=512, lengths=[512 - 9 * i for i in range(32)])), ("dex_vctk", dict(B=32, T=512, lengths=[512 - 9 * i for i in range(32)], Tr=60, Ts=60)), ("gedex_lj", dict(B=8, T=512, lengths=[512 - 30 * i for i in range(8)]))]:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    eng.set_precision("bf16")
    ys = []
    for e in ("0", "1"):
        os.environ["DEX_XCD_MAP"] = e
        ys.append(eng.sample(z, mask, mu, 3, **U.engine_kwargs(case)).cpu().numpy())
    del os.environ["DEX_XCD_MAP"]
    eng.set_precision("fp32")
    print(name, kw["B"], "bitwise equal:", np.array_equal(ys[0], ys[1]), "finite:", bool(np.isfinite(ys[1]).all()))
P
