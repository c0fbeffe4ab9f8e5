"""ORACLE TOOLING — golden values of the reference's own training-loss branch, ``Diffusion.forward(..., infer=False)`` =
``EDMLoss.forward`` (GeDEX-TTS/model/edm.py:22-68 called from diffusion.py:222-224; DEX-TTS/model/edm.py:22-68 from
diffusion.py:252-254), on portable synthetic weights / inputs and FIXED draws: the script seeds torch, records the two draws
``EDMLoss`` makes (``randn([B,1,1])`` then ``randn_like(x0)``), re-seeds and lets the reference make them itself.

Run only in the build container:   python -m oracle.make_golden_loss      -> tests/golden/edm_loss.npz  (data only)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402
from oracle import ref_import, dex_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LOSS_TYPES = ["base", "base_min_5", "base_log_5", "min_snr_5", "max_snr_5", "snr", "inv_snr"]
CASES = [("gedex_lj", 3, 64, [64, 44, 20], None, 11), ("gedex_vctk", 2, 36, [36, 20], None, 12),
         ("dex_vctk", 1, 64, [57], (40, 40, [33]), 13)]


def case_inputs(name, B, T, lengths, dex_dims):
    cfg = C.PRESETS[name]()
    mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
    x0 = (synth.normalish("x0", (B, 80, T), 21) * np.float32(1.2) - np.float32(4.0)) * mask      # a mel-like clean target
    kw = {}
    if cfg.variant == "dex":
        Tr, Ts, sl = dex_dims
        ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
        kw = dict(ref=ref, ref_lengths=ref_len, sty=sty, sty_lengths=sty_len)
    if cfg.n_spks > 1:
        kw["spk"] = synth.normalish("spk", (B, cfg.spk_emb_dim), 9)
    return cfg, mu, mask, x0.astype(np.float32), kw


@torch.no_grad()
def main():
    out = {}
    for name, B, T, lengths, dex_dims, seed in CASES:
        cfg, mu, mask, x0, kw = case_inputs(name, B, T, lengths, dex_dims)
        w = synth.make_weights(C.param_shapes(cfg), seed=0)
        m = ref_import.build_reference_diffusion(cfg, w)
        t = torch.from_numpy
        torch.manual_seed(seed)
        rnd = torch.randn([B, 1, 1])
        eps = torch.randn_like(t(x0))
        out[f"{name}_rnd_normal"], out[f"{name}_eps"] = rnd.numpy(), eps.numpy()
        for lt in LOSS_TYPES:
            m.loss_fn.loss_type = lt
            torch.manual_seed(seed)
            if cfg.variant == "dex":
                loss = m(t(x0), t(mask), t(mu), [t(r) for r in kw["ref"]], t(kw["ref_lengths"]), t(kw["sty"]), t(kw["sty_lengths"]), infer=False)
            else:
                loss = m(t(x0), t(mask), t(mu), spk=(t(kw["spk"]) if "spk" in kw else None), infer=False)
            out[f"{name}_{lt}"] = np.float32(loss.item())
            # pin the oracle's restatement right here
            W = O.as_torch(w)
            okw = {k: ([t(r) for r in v] if k == "ref" else t(np.asarray(v))) for k, v in kw.items() if k != "ref_lengths"}
            ol = O.edm_loss(W, cfg, t(x0), t(mask), t(mu), rnd, eps, loss_type=lt, **okw)
            print(f"{name:12s} {lt:12s} reference {loss.item():.7f}  oracle {float(ol):.7f}  |d| {abs(loss.item() - float(ol)):.2e}")
    np.savez_compressed(os.path.join(OUT, "edm_loss.npz"), **out)
    print("wrote edm_loss.npz")


if __name__ == "__main__":
    main()
