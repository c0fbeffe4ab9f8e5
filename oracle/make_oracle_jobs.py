"""Writes tests/golden/oracle_jobs/*.npy: the CPU oracle's outputs of the long jobs the GPU suite pins (tests/test_gpu_full_jobs.py):
BASELINE.json configs[2] (DEX-VCTK B = 32, T = 256, 50 Euler steps), the per-GPU share of configs[3] (DEX-ESD B = 32, 100 steps) and
configs[4] (GeDEX long form, B = 1, T = 4000, 50 steps).
Test infrastructure (the oracle is the checker, never the product).  Inputs are the portable synthetic weights / inputs of
dex_tts_amd/synth.py; the file name carries a hash of the inputs, the packed weights, the preset's config and the oracle version
(tests/gpu_util.py oracle_job_path), so a changed generator simply misses the store and the test recomputes.

    python -m oracle.make_oracle_jobs            # ~20 minutes on 8 cores
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch
    from dex_tts_amd import config as C
    from tests import gpu_util as U
    os.makedirs(U.ORACLE_JOBS, exist_ok=True)
    for name, n in (("dex_vctk", 50), ("dex_esd", 100), ("gedex_lj", 50)):
        cfg = C.PRESETS[name]()
        if name == "gedex_lj":
            case = U.make_case(cfg, B=1, T=4000)               # configs[4]
        else:
            case = U.make_case(cfg, B=32, T=256, lengths=[256 - 3 * i for i in range(32)], Tr=348, Ts=348, sty_lengths=[348 - 5 * i for i in range(32)])
        path = U.oracle_job_path(name, case, n)
        if os.path.exists(path):
            print("kept", path)
            continue
        t0 = time.time()
        ref = U.oracle_sampler_stored(name, case, n)
        np.save(path, ref.astype(np.float32))
        print(f"wrote {path}  {ref.shape}  {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
