"""Soak of the generated row-chain streams: the configs[2]-shaped job (DEX B = 32, T = 256) and a T = 512 job, many sampler calls in bf16 and
fp16, eager and graph replay - every result must be BITWISE equal to the first of its kind (a missing wait state or a miscounted
s_waitcnt in the streams shows up as a rare different bit, not as a crash).   python tools/soak_rowchain_a.py [calls]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_util as U  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cfg, eng, w = U.engine_for("dex_vctk")
    bad = 0
    for T, B in ((256, 32), (512, 32), (512, 10)):       # N = 1300 (672 tiles), 2580 (1312), 2580 at B = 10 (410 tiles: one or two per workgroup)
        case = U.make_case(cfg, B=B, T=T, lengths=[T - (3 * i) % (T // 3) for i in range(B)], Tr=60, Ts=60)
        mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
        for prec in ("bf16", "fp16"):
            eng.set_precision(prec)
            for graph in (False, True):
                t0 = time.time()
                first = None
                for c in range(calls):
                    y = eng.sample(z, mask, mu, 6, use_graph=graph, **U.engine_kwargs(case)).cpu().numpy()
                    if first is None:
                        first = y
                        assert np.isfinite(y).all()
                    elif not np.array_equal(first, y):
                        bad += 1
                        print(f"  MISMATCH T={T} B={B} {prec} graph={graph} call {c}: max|d| {np.abs(first - y).max():.3e}, {np.count_nonzero(first != y)} values")
                print(f"T={T} B={B} {prec} graph={graph}: {calls} calls of 6 steps, {time.time() - t0:.1f} s, last symbol of the row chain: generated streams" )
    eng.set_precision("fp32")
    print("soak:", "FAILED" if bad else "bitwise stable", f"({bad} mismatching calls)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
