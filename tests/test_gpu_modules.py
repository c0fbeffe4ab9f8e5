"""SURVEY §8(c) G1 on the GPU: the library's module-level checkpoints (exact-fp32 mode, debug call through the C ABI) against the outputs of the
REAL reference's modules (tests/golden/modules_*.npz, forward hooks: oracle/make_golden_modules.py) - ResnetBlocks, LinearAttention,
Downsample / Upsample, the time MLP, TimestepEmbedder, every DiTBlock, the DiT as a whole, the DEX adaptors - directly, no oracle in between
(tests/test_oracle_modules.py holds the oracle's restatement of each module to the same file on the CPU)."""
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import config as C, synth
from tests import gpu_util as U
from tests.tolerances import FP32_TAP_REL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"gedex_lj": None, "dex_vctk": (40, 40, [33])}


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


def sub_like(rows, r, B, T):
    """library tap [B * n, C] (channels-last rows) -> the golden's strided view"""
    n, Cc = rows.shape[0] // max(B, 1), rows.shape[1]
    if r.ndim == 4:                                   # [B, C, H, W] image: H = 80 at full resolution, 40 at half
        H = 80 if n == 80 * T else 40
        img = rows.reshape(B, H, n // H, Cc).transpose(0, 3, 1, 2)
        return img[:, ::4, ::4, ::4]
    if r.ndim == 3:                                   # [B, N, D] tokens
        return rows.reshape(B, n, Cc)[:, ::4, ::8]
    return rows.reshape(r.shape)


@pytest.mark.parametrize("name", list(CASES))
def test_library_modules_against_the_reference_modules(name):
    cfg, eng, w = U.engine_for(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"modules_{name}.npz"))
    B, T = int(g["case"][0]), int(g["case"][1])
    lengths = [int(v) for v in g["case"][2:]]
    mu, mask, z, _ = synth.make_inputs(B, T, lengths, seed=1234)
    eps = synth.normalish("eps", (B, 80, T), 5)
    kw = {}
    if cfg.variant == "dex":
        Tr, Ts, sl = CASES[name]
        ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
        kw = dict(ref=[torch.from_numpy(r) for r in ref], sty=torch.from_numpy(sty), sty_lengths=torch.from_numpy(np.asarray(sty_len)))
    eng.set_precision("fp32")
    x = torch.from_numpy(mu + 1.0 * eps)
    got = eng.denoise_once(x, 1.0, torch.from_numpy(mask), torch.from_numpy(mu), **kw).cpu().numpy()
    taps = {k: v.cpu().numpy() for k, v in eng.taps().items()}
    e = float(np.abs(got - g["precond"]).max())
    U.record(f"modules_{name}:fp32:call_vs_reference", max=e, ref_absmax=float(np.abs(g["precond"]).max()))
    assert e <= 1e-4 * max(1.0, float(np.abs(g["precond"]).max()))
    seen = []
    for tap, rows in taps.items():
        key = f"mod_{module_path(tap, cfg)}"
        if key not in g.files:
            continue
        r = g[key]
        v = sub_like(rows, r, B if r.shape[0] == B else 1, T)
        assert v.shape == r.shape, (tap, v.shape, r.shape)
        err, mag = float(np.abs(v - r).max()), float(np.abs(r).max())
        U.record(f"modules_{name}.{tap}:fp32:module_vs_reference", max=err, ref_absmax=mag)
        assert err <= FP32_TAP_REL * max(1.0, mag), (tap, err, mag)
        seen.append(key)
    need = {"mod_mlp", "mod_vit.t_embedder", "mod_vit", "mod_downs.0.0", "mod_downs.0.1", "mod_downs.0.2", "mod_downs.0.3", "mod_downs.1.0",
            "mod_downs.1.1", "mod_downs.1.2", "mod_ups.0.0", "mod_ups.0.1", "mod_ups.0.2", "mod_ups.0.3"} | {f"mod_vit.blocks.{k}" for k in range(cfg.dit.depth)}
    if cfg.variant == "dex":
        need |= {"mod_tv_adaptor", "mod_tiv_adaptor"}
    assert need <= set(seen), sorted(need - set(seen))
