"""Bitwise A/B of an environment knob: runs the sampler (2 Euler steps, bf16 and fp16) on batch-regime cases and prints a SHA-1 of
the outputs; run it once per setting (the library reads its knobs once per process) and compare: tools/env_bitwise.sh VAR a b"""
import sys, os, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as U

rng = np.random.default_rng(11)
cases = [("gedex_lj", dict(B=32, T=512, lengths=[512] + [int(rng.integers(170, 513)) for _ in range(31)])),
         ("gedex_lj", dict(B=13, T=388, lengths=[388] + [int(rng.integers(130, 389)) for _ in range(12)])),
         ("gedex_lj", dict(B=1, T=4000, lengths=[4000]))]
Ts = 148
cases.append(("dex_vctk", dict(B=32, T=256, lengths=[256] + [int(rng.integers(128, 257)) for _ in range(31)], Tr=Ts, Ts=Ts,
                               sty_lengths=[Ts] + [int(rng.integers(5, Ts + 1)) for _ in range(31)])))
for name, kw in cases:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    for prec in ("bf16", "fp16"):
        eng.set_precision(prec)
        out = eng.sample(z, mask, mu, 2, **U.engine_kwargs(case)).cpu().numpy()
        print(f"{name} B={kw['B']} T={kw['T']} {prec}: {hashlib.sha1(out.tobytes()).hexdigest()} finite={bool(np.isfinite(out).all())}", flush=True)
    eng.set_precision("fp32")
