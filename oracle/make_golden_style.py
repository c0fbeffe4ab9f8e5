"""Checker infrastructure (build container only): run the REAL reference style encoders (DEX-TTS/model/ref_encoder.py
TVEncoder / LF0Encoder / TIVEncoder built from config/VCTK/base.yaml) and the pre-decoder part of DeXTTS.forward
(tts.py:55-66, restated inline because tts.py itself cannot be imported: cp38 Cython + transformers-4.35) on portable
synthetic weights and inputs; commit tests/golden/style.npz + manifest_style_vctk.json.

    python -m oracle.make_golden_style
"""
import importlib
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
from oracle import ref_import, style_oracle as SO  # noqa: E402


@torch.no_grad()
def main():
    ref_import.import_reference("DEX-TTS")
    R = importlib.import_module("model.ref_encoder")
    m = yaml.safe_load(open("/root/reference/DEX-TTS/config/VCTK/base.yaml"))["model"]
    mods = {"tv_encoder": R.TVEncoder(**m["tv_encoder"]).eval(), "lf0_encoder": R.LF0Encoder(**m["lf0_encoder"]).eval(),
            "tiv_encoder": R.TIVEncoder(**m["tiv_encoder"]).eval(),
            "conv_sty": torch.nn.Conv1d(m["tv_encoder"]["c_out_g"], m["decoder"]["dim"] * 2, 1, 1).eval()}      # tts.py:31
    shapes = {f"{n}.{k}": list(v.shape) for n, mod in mods.items() for k, v in mod.state_dict().items()}
    w = synth.make_style_weights(shapes)
    for n, mod in mods.items():
        mod.load_state_dict({k[len(n) + 1:]: torch.from_numpy(v) for k, v in w.items() if k.startswith(n + ".")}, strict=True)
    B, T, lengths = 2, 40, [40, 27]
    mel, lf0, lengths = synth.make_style_inputs(B, T, lengths)
    ref, sty, lf0_t, L = torch.from_numpy(mel), torch.from_numpy(mel), torch.from_numpy(lf0), torch.from_numpy(lengths)
    # ---- tts.py:55-66 with the real modules
    ref_mask = SO.sequence_mask(L, ref.size(2)).unsqueeze(1).to(ref.dtype)
    lf0_mask = SO.sequence_mask(L, lf0_t.size(1)).unsqueeze(1).to(lf0_t.dtype)
    sty_mask = SO.sequence_mask(L, sty.size(2)).unsqueeze(1).to(sty.dtype)
    lf0_enc, lf0_dec = mods["lf0_encoder"](lf0_t, lf0_mask)
    sty_enc, sty_dec, _ = mods["tv_encoder"](sty, sty_mask)
    sty_enc = (sty_enc.sum(dim=-1) / sty_mask.sum(dim=-1)) + (lf0_enc.sum(dim=-1) / lf0_mask.sum(dim=-1))
    sty_dec = sty_dec + (lf0_dec.sum(dim=-1) / lf0_mask.sum(dim=-1)).unsqueeze(-1)
    sty_dec = mods["conv_sty"](sty_dec)
    _, skips = mods["tiv_encoder"](ref, ref_mask)
    out = {"sty_enc": sty_enc.numpy(), "sty_dec": sty_dec.numpy(), "ref_skips": np.stack([s.numpy() for s in skips])}
    # pin the oracle here too
    o = SO.style_forward({k: torch.from_numpy(v) for k, v in w.items()}, ref, L, sty, L, lf0_t, L)
    for k in ("sty_enc", "sty_dec"):
        print(k, out[k].shape, "oracle vs reference max|d| =", float(np.abs(o[k].numpy() - out[k]).max()), "|ref|max", float(np.abs(out[k]).max()))
    print("ref_skips oracle vs reference max|d| =", float(np.abs(np.stack([s.numpy() for s in o["ref_skips"]]) - out["ref_skips"]).max()))
    out["vq_idx"] = o["vq_idx"].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "style.npz"), **out)
    with open(os.path.join(OUT, "manifest_style_vctk.json"), "w") as f:
        json.dump({"config": {k: m[k] for k in ("tv_encoder", "lf0_encoder", "tiv_encoder")} | {"dim": m["decoder"]["dim"]}, "keys": shapes}, f, indent=0)
    print("wrote style.npz, manifest_style_vctk.json")


if __name__ == "__main__":
    main()
