"""bf16-mode error against the fp32 mode of the same library for an env knob's values (diagnostic, GPU).
usage: python tools/bf16_error.py ENVVAR v1 v2 ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
var, vals = sys.argv[1], sys.argv[2:]
cases = [("gedex_lj", dict(B=1, T=512)), ("gedex_lj", dict(B=2, T=128, lengths=[128, 77])), ("gedex_lj", dict(B=3, T=96, lengths=[96, 61, 7])),
         ("gedex_lj", dict(B=8, T=512, lengths=[512 - 40 * i for i in range(8)])), ("dex_vctk", dict(B=1, T=256, Tr=200, Ts=200, sty_lengths=[180]))]
for name, kw in cases:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, z = (torch.from_numpy(case[k]) for k in ("mu", "mask", "z"))
    ekw = U.engine_kwargs(case)
    eng.set_precision("fp32")
    ref = eng.sample(z, mask, mu, 10, **ekw).cpu().numpy()
    eng.set_precision("bf16")
    row = []
    for v in vals:
        os.environ[var] = v
        e = [np.abs(eng.sample(z, mask, mu, 10, **ekw).cpu().numpy() - ref) for _ in range(3)]
        row.append(f"{var}={v}: max {max(x.max() for x in e):.4f} mean {np.mean([x.mean() for x in e]):.5f}")
    eng.set_precision("fp32")
    print(name, kw.get("B"), kw.get("T"), " | ".join(row))
