#!/usr/bin/env python3
"""Does the padding-only-tile skip of conv3x3_lp_kernel trigger?  A batch whose utterances are much shorter than T: time and result of a
4-step sampler call with DEX_CONV_SKIP_DEAD=0 / 1 / 2 (2: experiment - dead tiles return at once, wrong results)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dex_tts_amd import config as C, synth
from dex_tts_amd.engine import ScoreNetEngine
dev = torch.device("cuda", 0)
cfg = C.PRESETS["gedex_lj"]()
eng = ScoreNetEngine(cfg, dev)
eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp16x2")
B, T = 32, 512
lens = [T] + [64] * (B - 1)
mu, mask, z, _ = synth.make_inputs(B, T, lens, seed=1)
mu, mask, z = (torch.from_numpy(a).to(dev) for a in (mu, mask, z))
ys = {}
for flag in ("0", "1", "2"):
    os.environ["DEX_CONV_SKIP_DEAD"] = flag
    ys[flag] = eng.sample(z, mask, mu, 2).cpu()
    print(flag, "finite", bool(torch.isfinite(ys[flag]).all()), "max|d vs 0|", float((ys[flag] - ys["0"]).abs().max()), flush=True)
