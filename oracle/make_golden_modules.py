"""ORACLE TOOLING — per-module golden vectors: forward hooks on EVERY module of the hot path's score network in the REAL
reference (imported from /root/reference via oracle/ref_import.py), one EDMPrecond call at sigma = 1.0 on the inputs of the existing
fixtures (same seeds: only outputs are stored).  Module outputs are stored under the reference's own module path
(`downs.0.0.block1`, `downs.0.1`, `downs.0.3`, `vit.x_embedder`, `vit.t_embedder`, `vit.blocks.2`, `vit.final_layer`, `final_block` ...),
strided like the stage checkpoints of make_golden.py so that a fixture stays small; the oracle's taps carry the same names.

Run only in the build container:   python -m oracle.make_golden_modules
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SIGMA = 1.0


def sub(t: torch.Tensor) -> np.ndarray:
    """Strided subsample: [B,C,H,W] -> [:, ::4, ::4, ::4]; [B,N,D] -> [:, ::4, ::8]; vectors whole."""
    t = t.detach()
    if t.dim() == 4:
        t = t[:, ::4, ::4, ::4]
    elif t.dim() == 3:
        t = t[:, ::4, ::8]
    return t.contiguous().numpy().astype(np.float32)


def module_names(cfg):
    ns = len(cfg.dim_mults)
    names = ["mlp"]
    for i in range(ns):
        names += [f"downs.{i}.0.block1", f"downs.{i}.0", f"downs.{i}.1", f"downs.{i}.2"]
        if i < ns - 1:
            names.append(f"downs.{i}.3")
    names += ["vit.x_embedder", "vit.t_embedder"] + [f"vit.blocks.{k}" for k in range(cfg.dit.depth)] + ["vit.final_layer", "vit"]
    for j in range(ns - 1):
        names += [f"ups.{j}.0.block1", f"ups.{j}.0", f"ups.{j}.1", f"ups.{j}.2", f"ups.{j}.3"]
    names += ["final_block", "final_conv"]
    if cfg.variant == "dex":
        names += ["tv_adaptor", "tiv_adaptor"]
    return names


@torch.no_grad()
def golden_modules(name, cfg, B, T, lengths, dex_dims=None):
    w = synth.make_weights(C.param_shapes(cfg), seed=0)
    m = ref_import.build_reference_diffusion(cfg, w)
    dn = m.denoise_fn
    mods = dict(dn.named_modules())
    mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
    tmu, tmask = torch.from_numpy(mu), torch.from_numpy(mask)
    eps = torch.from_numpy(synth.normalish("eps", (B, 80, T), 5))
    extra = []
    if cfg.variant == "dex":
        Tr, Ts, sl = dex_dims
        ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
        extra = [[torch.from_numpy(r) for r in ref], torch.from_numpy(ref_len), torch.from_numpy(sty), torch.from_numpy(sty_len)]
    store, hooks = {}, []
    for n in module_names(cfg):
        hooks.append(mods[n].register_forward_hook(lambda mod, a, o, n=n: store.__setitem__(n, sub(o))))
    d = m.precond_model(tmu + SIGMA * eps, torch.tensor(SIGMA), tmask, tmu, *extra)
    for h in hooks:
        h.remove()
    out = {"case": np.asarray([B, T] + list(lengths), dtype=np.int64), "precond": d.numpy()}
    out.update({f"mod_{k}": v for k, v in store.items()})
    np.savez_compressed(os.path.join(OUT, f"modules_{name}.npz"), **out)
    print(name, {k: v.shape for k, v in store.items()})


if __name__ == "__main__":
    golden_modules("gedex_lj", C.gedex_lj(), 2, 64, [64, 44])
    golden_modules("dex_vctk", C.dex_vctk(), 1, 64, [57], (40, 40, [33]))
