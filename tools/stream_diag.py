"""bf16 tile kernel vs strip-streaming kernel vs fp32 mode on a few shapes (diagnostic, GPU)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U
cases = [("gedex_lj", dict(B=2, T=128, lengths=[128, 77])), ("gedex_lj", dict(B=3, T=100, lengths=[100, 61, 7])),
         ("gedex_lj", dict(B=3, T=96, lengths=[96, 61, 7])), ("gedex_lj", dict(B=1, T=100)), ("gedex_lj", dict(B=1, T=4)),
         ("gedex_lj", dict(B=2, T=160, lengths=[160, 131]))]
for name, kw in cases:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, eps = (torch.from_numpy(case[k]) for k in ("mu", "mask", "eps"))
    for sigma in (80.0, 0.5):
        x = mu + sigma * eps
        eng.set_precision("fp32")
        ref = eng.denoise_once(x, sigma, mask, mu).cpu().numpy()
        eng.set_precision("bf16")
        outs = {}
        for mode in ("0", "0", "2", "2"):
            os.environ["DEX_CONV_STREAM"] = mode
            outs.setdefault(mode, []).append(eng.denoise_once(x, sigma, mask, mu).cpu().numpy())
        t0, t1 = outs["0"]; s0, s1 = outs["2"]
        f = lambda a, b: (float(np.abs(a - b).max()), float(np.abs(a - b).mean()))
        print(kw.get("B"), kw.get("T"), sigma, "tile-ref", f(t0, ref), "stream-ref", f(s0, ref), "tile-tile", f(t0, t1), "stream-stream", f(s0, s1), "stream-tile", f(s0, t0))
        eng.set_precision("fp32")
