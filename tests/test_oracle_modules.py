"""per-MODULE goldens.  tests/golden/modules_*.npz hold the outputs of every module of the hot path's score network in
the REAL reference (forward hooks, oracle/make_golden_modules.py: one EDMPrecond call at sigma = 1 on the inputs of the existing fixtures)
under the reference's own module paths; the oracle's restatement of each module (a5 time MLP, a6 Block, a7 ResnetBlock, a8
LinearAttention, a9 Down / Upsample, a10 PatchEmbed2D + positional conv, a11 TimestepEmbedder, a12 DiTBlock, a13 FinalLayer +
unpatchify, a14 final block + 1x1 conv, a15 / a16 the DEX adaptors) is held to them one by one."""
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import config as C, synth
from oracle import dex_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"gedex_lj": (C.gedex_lj, None), "dex_vctk": (C.dex_vctk, (40, 40, [33]))}
# oracle tap name -> reference module path, where the two differ
ALIAS = {"down{i}": "downs.{i}.2", "up{j}": "ups.{j}.2", "tok_blk{k}": "vit.blocks.{k}", "dit_out": "vit", "tv": "tv_adaptor",
         "tiv": "tiv_adaptor"}


def sub(t):
    if t.dim() == 4:
        t = t[:, ::4, ::4, ::4]
    elif t.dim() == 3:
        t = t[:, ::4, ::8]
    return t.contiguous().numpy().astype(np.float32)


def module_path(tap, cfg):
    ns = len(cfg.dim_mults)
    for i in range(ns):
        if tap == f"down{i}":
            return f"downs.{i}.2"
        if tap == f"up{i}":
            return f"ups.{i}.2"
    for k in range(cfg.dit.depth):
        if tap == f"tok_blk{k}":
            return f"vit.blocks.{k}"
    if tap == "up_out":
        return f"ups.{ns - 2}.3"
    return {"dit_out": "vit", "tv": "tv_adaptor", "tiv": "tiv_adaptor"}.get(tap, tap)


@pytest.mark.parametrize("name", list(CASES))
def test_every_module_of_the_score_network_against_the_reference(name):
    mk, dex_dims = CASES[name]
    cfg = mk()
    g = np.load(os.path.join(ROOT, "tests", "golden", f"modules_{name}.npz"))
    B, T = int(g["case"][0]), int(g["case"][1])
    lengths = [int(v) for v in g["case"][2:]]
    W = O.as_torch(synth.make_weights(C.param_shapes(cfg)), torch.float32)
    mu, mask, z, _ = synth.make_inputs(B, T, lengths, seed=1234)
    tmu, tmask = torch.from_numpy(mu), torch.from_numpy(mask)
    eps = torch.from_numpy(synth.normalish("eps", (B, 80, T), 5))
    kw = {}
    if cfg.variant == "dex":
        Tr, Ts, sl = dex_dims
        ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
        kw = dict(ref=[torch.from_numpy(r) for r in ref], sty=torch.from_numpy(sty), sty_lengths=torch.from_numpy(sty_len))
    taps = {}
    with torch.no_grad():
        d = O.edm_precond(W, cfg, tmu + 1.0 * eps, torch.tensor(1.0), tmask, tmu, taps=taps, **kw)
    assert np.abs(d.numpy() - g["precond"]).max() == 0.0
    seen = set()
    for tap, v in taps.items():
        path = module_path(tap, cfg)
        key = f"mod_{path}"
        if key not in g:
            continue
        r = g[key]
        got = sub(v)
        if got.shape != r.shape and got.size == r.size:
            got = got.reshape(r.shape)
        assert got.shape == r.shape, (tap, path, got.shape, r.shape)
        err = float(np.abs(got - r).max())
        assert err <= 1e-6 * max(1.0, float(np.abs(r).max())), (tap, path, err)
        seen.add(key)
    missing = [k for k in g.files if k.startswith("mod_") and k not in seen]
    assert not missing, missing          # every stored module has an oracle checkpoint
