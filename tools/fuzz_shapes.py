"""Shape fuzz on the GPU: bf16 mode vs the library's own fp32 mode (which the parity tests pin to the oracle) over
odd batch sizes / lengths, both model families.  Prints one line per case; exits non-zero on a violation."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as U

rng = np.random.default_rng(7)
cases = []
for T in (4, 8, 36, 100, 132, 260, 388, 512, 804):
    for B in (1, 2, 5):
        if T * B > 1700: continue
        lens = [T] + [int(rng.integers(max(1, T // 3), T + 1)) for _ in range(B - 1)]
        cases.append(("gedex_lj", dict(B=B, T=T, lengths=lens)))
for T in (8, 68, 132, 256):
    for B in (1, 3):
        lens = [T] + [int(rng.integers(max(1, T // 2), T + 1)) for _ in range(B - 1)]
        Ts = int(rng.integers(20, 200))
        cases.append(("dex_vctk", dict(B=B, T=T, lengths=lens, Tr=Ts, Ts=Ts, sty_lengths=[Ts] + [int(rng.integers(5, Ts + 1)) for _ in range(B - 1)])))
bad = 0
for name, kw in cases:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, eps = (torch.from_numpy(case[k]) for k in ("mu", "mask", "eps"))
    worst = (0.0, 0.0)
    for sigma in (80.0, 0.3):
        x = mu + float(sigma) * eps
        eng.set_precision("fp32")
        ref = eng.denoise_once(x, sigma, mask, mu, **U.engine_kwargs(case)).cpu().numpy()
        eng.set_precision("bf16")
        got = eng.denoise_once(x, sigma, mask, mu, **U.engine_kwargs(case)).cpu().numpy()
        got2 = eng.denoise_once(x, sigma, mask, mu, **U.engine_kwargs(case)).cpu().numpy()
        e = np.abs(got - ref)
        rep = float(np.abs(got2 - got).max())          # run-to-run (atomics order) noise
        worst = (max(worst[0], float(e.max())), max(worst[1], float(e.mean())))
        ok = np.isfinite(got).all() and e.max() <= 5e-2 and e.mean() <= 8e-3
        if not ok: bad += 1
    eng.set_precision("fp32")
    print(f"{name:10s} B={kw['B']} T={kw['T']:4d} lens={kw['lengths']}  max={worst[0]:.3e} mean={worst[1]:.3e} repeat_diff={rep:.1e} {'OK' if ok else 'FAIL'}", flush=True)
# ---- batch regime (the throughput kernels: ping-pong convolutions, 16-bit single-consumer activations, column-walking GEMM,
# 8-row tiles, 64-row DiT block): two sampler steps in bf16 / fp16 against the library's fp32 mode
bcases = []
for T in (36, 132, 260, 388, 512, 516):
    for B in (7, 13, 32):
        if T * B > 17000: continue
        lens = [T] + [int(rng.integers(max(1, T // 3), T + 1)) for _ in range(B - 1)]
        bcases.append(("gedex_lj", dict(B=B, T=T, lengths=lens)))
for T in (68, 132, 256):
    for B in (9, 32):
        lens = [T] + [int(rng.integers(max(1, T // 2), T + 1)) for _ in range(B - 1)]
        Ts = int(rng.integers(20, 200))
        bcases.append(("dex_vctk", dict(B=B, T=T, lengths=lens, Tr=Ts, Ts=Ts, sty_lengths=[Ts] + [int(rng.integers(5, Ts + 1)) for _ in range(B - 1)])))
# round 6: token counts around the 64-query attention's unit plans (whole / half / passive waves; N = 20 (T / 4 + 1) for DEX-VCTK):
# N = 1000 (32 blocks: no ragged wave), 1060, 1300 (4 whole + 3 half units), 1540, 2020 (one utterance short of a round), B with 2 B % 8 != 0
for T, B in ((196, 32), (208, 32), (256, 30), (304, 32), (400, 16), (512, 12)):
    lens = [T] + [int(rng.integers(max(1, T // 2), T + 1)) for _ in range(B - 1)]
    bcases.append(("dex_vctk", dict(B=B, T=T, lengths=lens, Tr=60, Ts=60, sty_lengths=[60] + [int(rng.integers(5, 61)) for _ in range(B - 1)])))
for name, kw in bcases:
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    eng.set_precision("fp32")
    ref = eng.sample(z, mask, mu, 2, **U.engine_kwargs(case)).cpu().numpy()
    line = []
    ok_all = True
    for prec in ("bf16", "fp16", "fp16x2"):
        eng.set_precision(prec)
        got = eng.sample(z, mask, mu, 2, **U.engine_kwargs(case)).cpu().numpy()
        got2 = eng.sample(z, mask, mu, 2, **U.engine_kwargs(case)).cpu().numpy()
        e = np.abs(got - ref)
        ok = np.isfinite(got).all() and e.max() <= (6e-2 if prec == "bf16" else 1e-2) and e.mean() <= (8e-3 if prec == "bf16" else 1.5e-3 if prec == "fp16" else 1.0e-3) and np.array_equal(got, got2)
        ok_all = ok_all and ok
        line.append(f"{prec} max={e.max():.2e} mean={e.mean():.2e}")
    eng.set_precision("fp32")
    if not ok_all: bad += 1
    print(f"{name:10s} B={kw['B']:2d} T={kw['T']:4d}  {' | '.join(line)}  {'OK' if ok_all else 'FAIL'}", flush=True)
print("violations:", bad)
sys.exit(1 if bad else 0)
