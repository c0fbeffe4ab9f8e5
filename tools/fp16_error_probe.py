#!/usr/bin/env python
"""Where does the fp16-operand mode lose its accuracy at configs[1] (GeDEX-LJ B=1 T=512)?  Against the library's own exact-fp32 mode
(4e-6 from the CPU oracle on this job): per-stage taps of single calls, and the 50-step job under the knobs that keep intermediates
fp32.  python tools/fp16_error_probe.py > gpurun_out/fp16_error_probe.txt"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_tts_amd import config as C, synth
from dex_tts_amd.engine import ScoreNetEngine

dev = torch.device("cuda", 0)
cfg = C.gedex_lj()
eng = ScoreNetEngine(cfg, dev)
eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
mu, mask, z, _ = synth.make_inputs(1, 512, None, seed=1234)
eps = synth.normalish("eps", (1, 80, 512), 1239)
mu, mask, z, eps = (torch.from_numpy(a).to(dev) for a in (mu, mask, z, eps))

def taps_of(prec, sigma):
    eng.set_precision(prec)
    y = eng.denoise_once(mu + sigma * eps, sigma, mask, mu)
    return y, eng.taps()

for sigma in (80.0, 1.0, 0.05):
    y32, t32 = taps_of("fp32", sigma)
    for prec in ("fp16", "bf16"):
        y, t = taps_of(prec, sigma)
        line = [f"sigma {sigma:6} {prec}: out max {float((y - y32).abs().max()):.2e} mean {float((y - y32).abs().mean()):.2e} |"]
        for k in t32:
            if k in t and t[k].shape == t32[k].shape:
                d = (t[k] - t32[k]).abs()
                line.append(f"{k} {float(d.max()):.1e}/{float(d.mean()):.1e} (rms {float(t32[k].pow(2).mean().sqrt()):.2f})")
        print(" ".join(line), flush=True)

eng.set_precision("fp32")
y32 = eng.sample(z, mask, mu, 50)
def job(prec, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        eng.set_precision(prec)
        y = eng.sample(z, mask, mu, 50)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    d = (y - y32).abs()
    print(f"50-step {prec} {env}: max {float(d.max()):.3e} mean {float(d.mean()):.3e}", flush=True)
for prec in ("fp16", "bf16"):
    job(prec)
    job(prec, DEX_H_BF16=0)
    job(prec, DEX_DIT_CHAIN=0)
    job(prec, DEX_DIT_CLUSTER=0)
    job(prec, DEX_H_BF16=0, DEX_DIT_CHAIN=0, DEX_PATCH_FUSED=0)
eng.set_precision("fp32")
