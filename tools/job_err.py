#!/usr/bin/env python3
"""abs_err of a pinned whole job (bench.py job_abs_err: the engine's result vs the CPU oracle's stored output) for A/B builds:
    [DEX_AMD_LIB=other.so] python tools/job_err.py dex_b32 fp16x2 [gedex_long fp16 ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
args = sys.argv[1:]
for w, prec in zip(args[0::2], args[1::2]):
    r = bench.job_abs_err(w, prec, dev, st)
    print(f"{w} {prec}: abs_err max {r['abs_err'][0]:.4e} mean {r['abs_err'][1]:.4e}" if r["abs_err"] else f"{w} {prec}: {r['abs_err_source']}", flush=True)
