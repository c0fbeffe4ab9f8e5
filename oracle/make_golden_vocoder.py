"""Checker infrastructure (build container only): run the REAL reference HiFi-GAN Generator (GeDEX-TTS/hifigan/models.py,
config hifigan/config.json) on portable synthetic weights and a synthetic mel, and commit tests/golden/vocoder.npz:
the mel, the waveform, the state-dict key/shape manifest, and a weight-norm fold check (weight_g / weight_v -> weight).

    python -m oracle.make_golden_vocoder
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/GeDEX-TTS"
OUT = os.path.join(ROOT, "tests", "golden")

from dex_tts_amd import synth, vocoder as V  # noqa: E402
from oracle import vocoder_oracle as VO  # noqa: E402


@torch.no_grad()
def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import hifigan                                      # the reference package
    h = hifigan.AttrDict(json.load(open(os.path.join(REF, "hifigan", "config.json"))))
    torch.manual_seed(0)                                # the weight-norm fold vectors below come from this init: seeded, so the fixture reproduces
    g = hifigan.Generator(h).eval()
    sd_wn = {k: v.clone() for k, v in g.state_dict().items()}          # with weight norm: *.weight_g / *.weight_v
    g.remove_weight_norm()
    keys = {k: list(v.shape) for k, v in g.state_dict().items()}
    shapes = V.param_shapes(V.HIFIGAN_V1)
    assert {k: tuple(v) for k, v in keys.items()} == {k: tuple(v) for k, v in shapes.items()}, "param_shapes disagrees with the reference"
    w = VO.synth_weights(shapes)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    B, T = 2, 12
    mel = np.clip(synth.normalish("voc_mel", (B, 80, T), 55) * 1.5 - 5.0, -11.5, 2.5).astype(np.float32)
    wav = g(torch.from_numpy(mel)).numpy()
    # pin the oracle right here as well
    ow = VO.generator({k: torch.from_numpy(v) for k, v in w.items()}, V.HIFIGAN_V1, torch.from_numpy(mel)).numpy()
    print("oracle vs reference: max|d| =", float(np.abs(ow - wav).max()), " |wav|max =", float(np.abs(wav).max()),
          " saturated (|wav| > 0.99):", float((np.abs(wav) > 0.99).mean()), " std:", float(wav.std()), wav.shape)
    # weight-norm fold: one Conv1d and one ConvTranspose1d of the weight-normed module vs what remove_weight_norm leaves
    g2 = hifigan.Generator(h).eval()
    g2.load_state_dict(sd_wn)
    fold_in = {k: sd_wn[k].numpy() for k in ("conv_post.weight_g", "conv_post.weight_v", "ups.3.weight_g", "ups.3.weight_v")}
    g2.remove_weight_norm()
    fold_out = {"conv_post.weight": g2.conv_post.weight.numpy().copy(), "ups.3.weight": g2.ups[3].weight.numpy().copy()}
    np.savez_compressed(os.path.join(OUT, "vocoder.npz"), mel=mel, wav=wav,
                        **{"foldin__" + k: v for k, v in fold_in.items()}, **{"foldout__" + k: v for k, v in fold_out.items()})
    with open(os.path.join(OUT, "manifest_hifigan_v1.json"), "w") as f:
        json.dump({"config": {k: h[k] for k in ("upsample_rates", "upsample_kernel_sizes", "upsample_initial_channel", "resblock",
                                                "resblock_kernel_sizes", "resblock_dilation_sizes", "num_mels")}, "keys": keys}, f, indent=0)
    print("wrote vocoder.npz, manifest_hifigan_v1.json")


if __name__ == "__main__":
    main()
