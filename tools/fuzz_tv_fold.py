"""Shape fuzz of the DEX TV adaptor on the GPU: the folded one-launch form (DEX_TV_FOLD=1) against the projection form (=0) and the oracle's
`tv` / `tiv` taps over style lengths around the 64-key tile boundaries, tiny / ragged style lengths, odd batch sizes and frame counts
(bf16 and fp16).  Prints one line per case; exits non-zero on a violation of the test suite's tap bounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpu_util as U

REL = {"bf16": {"tv": 2.1e-2, "tiv": 4.5e-2}, "fp16": {"tv": 2.5e-3, "tiv": 6.0e-3}}
rng = np.random.default_rng(11)
cases = []
for Ts in (1, 2, 62, 63, 64, 65, 127, 128, 129, 191, 200, 255, 256, 300, 348):
    B = int(rng.integers(1, 6)); T = int(rng.choice([8, 36, 52, 100, 132, 256]))
    lens = [T] + [int(rng.integers(max(1, T // 3), T + 1)) for _ in range(B - 1)]
    sl = [Ts] + [int(rng.integers(1, Ts + 1)) for _ in range(B - 1)]
    cases.append(dict(B=B, T=T, lengths=lens, Tr=max(2, min(Ts, 60)), Ts=Ts, sty_lengths=sl))
bad = 0
cfg, eng, w = U.engine_for("dex_vctk")
for kw in cases:
    case = U.make_case(cfg, **kw)
    line = f"B={kw['B']} T={kw['T']:3d} Ts={kw['Ts']:3d} sty={kw['sty_lengths']}"
    for prec in ("bf16", "fp16"):
        eng.set_precision(prec)
        res = {}
        for fold in ("0", "1"):
            os.environ["DEX_TV_CHAIN"] = "2"; os.environ["DEX_TV_FOLD"] = fold
            got, ref, terr = U.run_precond("dex_vctk", case, 0.7)
            res[fold] = terr
        ok = True
        for fold, terr in res.items():
            for k in ("tv", "tiv"):
                err, mag = terr[k]
                ok &= bool(np.isfinite(err)) and err <= REL[prec][k] * max(1.0, mag)
        line += f" | {prec} tv fold0 {res['0']['tv'][0]:.2e} fold1 {res['1']['tv'][0]:.2e} (|ref| {res['1']['tv'][1]:.1f}) {'OK' if ok else 'FAIL'}"
        bad += 0 if ok else 1
    print(line, flush=True)
os.environ.pop("DEX_TV_CHAIN", None); os.environ.pop("DEX_TV_FOLD", None)
eng.set_precision("fp32")
print("violations:", bad)
sys.exit(1 if bad else 0)
