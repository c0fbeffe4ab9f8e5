# A/B of the two 16-bit residual-stream stores of the batch regime: speed and error against the fp32 mode
#   DEX_ATTN_X_LP: the ResnetBlock output the linear attention's context pass hands to its tail kernel
#   DEX_RES_X_LP:  the first ResnetBlock's output the second block's fused conv hands to that context pass
for w in gedex_b32 dex_b32 gedex_long; do
  for e in "DEX_ATTN_X_LP=0 DEX_RES_X_LP=0" "DEX_ATTN_X_LP=1 DEX_RES_X_LP=0" "DEX_ATTN_X_LP=1 DEX_RES_X_LP=1"; do
    env $e python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $w', d['dtype'], d['value'], d['ms_per_euler_step'])"
  done
done
python - <<'P'
import os, numpy as np, torch
from tests import gpu_util as U
for name, kw in [("gedex_lj", dict(B=32, T=512, lengths=[512 - 9 * i for i in range(32)])), ("dex_vctk", dict(B=32, T=256, lengths=[256 - 5 * i for i in range(32)], Tr=60, Ts=60))]:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    ref = eng.sample(z, mask, mu, 10, **U.engine_kwargs(case)).cpu().numpy()
    for prec in ("bf16", "fp16"):
        eng.set_precision(prec)
        for a, r in (("0", "0"), ("1", "0"), ("1", "1")):
            os.environ["DEX_ATTN_X_LP"] = a; os.environ["DEX_RES_X_LP"] = r
            y = eng.sample(z, mask, mu, 10, **U.engine_kwargs(case)).cpu().numpy()
            d = np.abs(y - ref)
            print(f"{name} B=32 n=10 {prec} DEX_ATTN_X_LP={a} DEX_RES_X_LP={r}: max|d| {d.max():.4e} mean|d| {d.mean():.4e} vs the fp32 mode")
        del os.environ["DEX_ATTN_X_LP"], os.environ["DEX_RES_X_LP"]
        eng.set_precision("fp32")
P
