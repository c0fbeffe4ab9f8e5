"""GPU diagnostic (not a pytest file): prints per-stage errors of the HIP path against the CPU oracle
for every preset, so one gpurun call localises a wrong kernel.   python tests/diag_gpu.py"""
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import gpu_util as U  # noqa: E402


def main():
    print("device:", torch.cuda.get_device_name(0), flush=True)
    cases = [("gedex_lj", dict(B=2, T=64, lengths=[64, 44])),
             ("gedex_vctk", dict(B=2, T=32, lengths=[32, 21])),
             ("dex_vctk", dict(B=1, T=64, lengths=[57], Tr=40, Ts=40, sty_lengths=[33])),
             ("dex_vctk", dict(B=2, T=48, lengths=[48, 30], Tr=36, Ts=50, sty_lengths=[50, 17]))]
    for name, kw in cases:
        try:
            cfg, eng, w = U.engine_for(name)
            case = U.make_case(cfg, **kw)
            for sigma in (80.0, 1.0, 0.002):
                got, ref, terr = U.run_precond(name, case, sigma)
                e = np.abs(got - ref)
                print(f"[{name} {kw}] sigma={sigma}: out max_err={e.max():.3e} mean_err={e.mean():.3e} ref_max={np.abs(ref).max():.3f} nan={np.isnan(got).any()}", flush=True)
                if sigma == 1.0:
                    for k, (err, mx) in terr.items():
                        print(f"      tap {k:10s} max_err={err:.3e} ref_max={mx:.3f}", flush=True)
            for n in (4, 10):
                t0 = time.time()
                got, ref = U.run_sampler(name, case, n)
                e = np.abs(got - ref)
                print(f"[{name}] sampler n={n}: max_err={e.max():.3e} mean_err={e.mean():.3e} ({time.time()-t0:.1f}s)", flush=True)
            got, ref = U.run_sampler(name, case, 10, use_graph=True)
            e = np.abs(got - ref)
            print(f"[{name}] sampler n=10 hipGraph: max_err={e.max():.3e} mean_err={e.mean():.3e}", flush=True)
        except Exception:
            traceback.print_exc()
    # bf16-MFMA mode: error against the fp32 oracle (no reference counterpart: the reference cannot run in bf16)
    for name, kw in cases[:1] + cases[2:3]:
        try:
            cfg, eng, w = U.engine_for(name)
            eng.set_precision("bf16")
            case = U.make_case(cfg, **kw)
            for sigma in (80.0, 1.0, 0.002):
                got, ref, terr = U.run_precond(name, case, sigma)
                e = np.abs(got - ref)
                print(f"[bf16 {name}] sigma={sigma}: out max_err={e.max():.3e} mean_err={e.mean():.3e} ref_max={np.abs(ref).max():.3f}", flush=True)
                if sigma == 1.0:
                    for k, (err, mx) in terr.items():
                        print(f"      tap {k:10s} max_err={err:.3e} ref_max={mx:.3f}", flush=True)
            for n in (10, 50):
                got, ref = U.run_sampler(name, case, n)
                e = np.abs(got - ref)
                print(f"[bf16 {name}] sampler n={n}: max_err={e.max():.3e} mean_err={e.mean():.3e} rms_ref={np.sqrt((ref**2).mean()):.3f}", flush=True)
            eng.set_precision("fp32")
        except Exception:
            traceback.print_exc()
    # timing at the bench shape
    for prec in ("fp32", "bf16"):
        try:
            cfg, eng, w = U.engine_for("gedex_lj")
            eng.set_precision(prec)
            case = U.make_case(cfg, B=1, T=512)
            mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
            for graph in (False, True):
                for _ in range(2):
                    eng.sample(z, mask, mu, 50, use_graph=graph)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(3):
                    out = eng.sample(z, mask, mu, 50, use_graph=graph)
                torch.cuda.synchronize()
                dt = (time.time() - t0) / 3
                print(f"[{prec}] gedex_lj B=1 T=512 n=50 graph={graph}: {dt*1e3:.2f} ms  -> {512/dt:.0f} frames/s, {dt/50*1e3:.3f} ms/step", flush=True)
            eng.profile(True)
            eng.sample(z, mask, mu, 50)
            torch.cuda.synchronize()
            rows = sorted(eng.profile_rows(), key=lambda r: -r["ms"])
            tot = sum(r["ms"] for r in rows)
            print(f"[{prec}] per-kernel (event-timed, eager, 50 steps) total {tot:.2f} ms")
            for r in rows:
                tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
                gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
                print(f"  {r['name']:22s} calls={r['calls']:5d} ms={r['ms']:8.3f} avg_us={r['ms']/r['calls']*1e3:8.2f} TF/s={tf:7.2f} GB/s={gb:8.1f}")
            eng.profile(False)
            eng.set_precision("fp32")
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    main()
