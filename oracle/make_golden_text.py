"""Checker infrastructure (build container only): run the REAL reference TextEncoder (model/text_encoder.py + RetNet) and the
duration / alignment lines of the TTS forward (tts.py:37-50, restated inline around the reference's own ``generate_path`` /
``sequence_mask`` / ``fix_len_compatibility`` because tts.py itself cannot be imported: cp38 Cython) on portable synthetic
weights and inputs; commit tests/golden/text_<case>.npz + manifest_text_<case>.json and pin oracle/text_oracle.py on the way.

    python -m oracle.make_golden_text
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from dex_tts_amd import synth  # noqa: E402
from oracle import ref_import, text_oracle as TO  # noqa: E402

CASES = {          # name -> (reference tree, YAML, n_spks override, n_vocab)
    "gedex_lj": ("GeDEX-TTS", "config/LJSpeech/base.yaml", 1),
    "gedex_vctk": ("GeDEX-TTS", "config/VCTK/base.yaml", 108),
    "dex_vctk": ("DEX-TTS", "config/VCTK/base.yaml", 0),
}
N_VOCAB = 149      # len(symbols) + 1 with add_blank (text/symbols.py); any value works for the arithmetic


@torch.no_grad()
def run(name, length_scale=1.0):
    sub, yml, n_spks = CASES[name]
    m = yaml.safe_load(open(f"/root/reference/{sub}/{yml}"))["model"]
    kw = dict(m["encoder"], n_vocab=N_VOCAB, n_feats=m["n_feats"], n_spks=n_spks, spk_emb_dim=m["spk_emb_dim"])
    enc, utils = ref_import.build_reference_text_encoder(sub, kw)
    shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
    w = synth.make_text_weights(shapes)
    for k in ("encoder.retnet_rel_pos.angle", "encoder.retnet_rel_pos.decay"):      # registered buffers: the reference's own values (torch's
        assert np.allclose(w[k], enc.state_dict()[k].numpy(), rtol=1e-6), k            # pow differs from numpy's in the last bit) travel in the fixture
        w[k] = enc.state_dict()[k].numpy().copy()
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    B, L, lengths = 2, 23, [23, 14]
    tok, lengths = synth.make_text_inputs(B, L, lengths, N_VOCAB)
    x, xl = torch.from_numpy(tok), torch.from_numpy(lengths)
    spk = torch.from_numpy(synth.normalish("text_spk", (B, m["spk_emb_dim"]), 5)) if n_spks > 1 else None
    sty = torch.from_numpy(synth.normalish("text_sty", (B, m["encoder"]["n_channels"]), 6) * np.float32(0.5)) if sub == "DEX-TTS" else None
    if sub == "DEX-TTS":
        mu, logw, x_mask = enc(x, xl, sty, spk=None)
    else:
        mu, logw, x_mask = enc(x, xl, spk=spk)
    # ---- tts.py:37-50 around the reference's own helpers
    wd = torch.exp(logw) * x_mask
    w_ceil = torch.ceil(wd) * length_scale
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_max = int(y_lengths.max())
    y_max_ = utils.fix_len_compatibility(y_max)
    y_mask = utils.sequence_mask(y_lengths, y_max_).unsqueeze(1).to(x_mask.dtype)
    attn = utils.generate_path(w_ceil.squeeze(1), (x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1)).unsqueeze(1)
    mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu.transpose(1, 2)).transpose(1, 2)
    out = dict(mu=mu.numpy(), logw=logw.numpy(), x_mask=x_mask.numpy(), w_ceil=w_ceil.numpy(), y_lengths=y_lengths.numpy(),
               y_mask=y_mask.numpy(), attn=attn.numpy().astype(np.int8), mu_y=mu_y.numpy(),
               angle=w["encoder.retnet_rel_pos.angle"], decay=w["encoder.retnet_rel_pos.decay"])
    # ---- pin the restatement
    cfg = dict(n_channels=m["encoder"]["n_channels"], n_layers=m["encoder"]["n_layers"], n_heads=m["encoder"]["n_heads"], n_spks=n_spks,
               kernel_size=m["encoder"]["kernel_size"])
    W = {k: torch.from_numpy(v) for k, v in w.items()}
    omu, ologw, omask = TO.text_encoder_forward(W, cfg, x, xl, spk=spk, sty=sty)
    oa = TO.align(omu, ologw, omask, length_scale)
    print(f"{name}: mu {tuple(mu.shape)} |mu|max {float(mu.abs().max()):.3f}  oracle max|d| mu {float((omu - mu).abs().max()):.2e} logw {float((ologw - logw).abs().max()):.2e}"
          f"  durations {w_ceil[0, 0, :8].int().tolist()} y_lengths {y_lengths.tolist()}  attn equal {bool((oa['attn'] == attn).all())}"
          f" mu_y max|d| {float((oa['mu_y'] - mu_y).abs().max()):.2e}")
    np.savez_compressed(os.path.join(OUT, f"text_{name}.npz"), **out)
    with open(os.path.join(OUT, f"manifest_text_{name}.json"), "w") as f:
        json.dump({"config": dict(kw), "keys": shapes}, f, indent=0)


if __name__ == "__main__":
    for n in CASES:
        run(n)
