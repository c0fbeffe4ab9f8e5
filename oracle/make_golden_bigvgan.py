"""Checker infrastructure (build container only): run the REAL reference BigVGAN generator (DEX-TTS/bigvgan/models.py with its
alias_free_torch resamplers and Snake activations; published bigvgan_base_22khz_80band configuration — the config.json that
src/utils.py:267 opens is not in the reference tree) on portable synthetic weights and a synthetic mel; commit
tests/golden/bigvgan.npz (mel, waveform, the resampling filter the reference registers) + manifest_bigvgan_base.json and pin
oracle/bigvgan_oracle.py on the way.

    python -m oracle.make_golden_bigvgan
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/DEX-TTS"
OUT = os.path.join(ROOT, "tests", "golden")

from dex_tts_amd import synth, vocoder as V  # noqa: E402
from oracle import bigvgan_oracle as BO  # noqa: E402


@torch.no_grad()
def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import bigvgan                                      # the reference package
    import contextlib, io
    h = bigvgan.AttrDict(dict(V.BIGVGAN_BASE))
    g = bigvgan.Generator(h).eval()
    with contextlib.redirect_stdout(io.StringIO()):
        g.remove_weight_norm()
    keys = {k: list(v.shape) for k, v in g.state_dict().items()}
    shapes = V.param_shapes(V.BIGVGAN_BASE)
    assert {k: tuple(v) for k, v in keys.items()} == {k: tuple(v) for k, v in shapes.items()}, "param_shapes disagrees with the reference"
    w = synth.make_vocoder_weights(shapes)
    filt = g.state_dict()["activation_post.upsample.filter"].numpy().copy()
    for k in w:
        if k.endswith(".filter"):                       # registered buffers: the reference's own constant (torch's kaiser window differs
            assert np.allclose(w[k], filt, atol=1e-7), k    # from numpy's in the last bit) travels in the fixture
            w[k] = filt.copy()
    assert np.array_equal(BO.kaiser_sinc_filter1d(0.25, 0.3, 12).numpy(), filt.flatten())
    g.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    B, T = 2, 9
    mel = np.clip(synth.normalish("bvg_mel", (B, 80, T), 56) * 1.5 - 5.0, -11.5, 2.5).astype(np.float32)
    wav = g(torch.from_numpy(mel)).numpy()
    ow = BO.generator({k: torch.from_numpy(v) for k, v in w.items()}, V.BIGVGAN_BASE, torch.from_numpy(mel)).numpy()
    print("oracle vs reference: max|d| =", float(np.abs(ow - wav).max()), " |wav|max =", float(np.abs(wav).max()),
          " saturated (|wav| > 0.99):", float((np.abs(wav) > 0.99).mean()), " std:", float(wav.std()), wav.shape)
    np.savez_compressed(os.path.join(OUT, "bigvgan.npz"), mel=mel, wav=wav, filter=filt)
    with open(os.path.join(OUT, "manifest_bigvgan_base.json"), "w") as f:
        json.dump({"config": dict(V.BIGVGAN_BASE), "keys": keys}, f, indent=0)
    print("wrote bigvgan.npz, manifest_bigvgan_base.json")


if __name__ == "__main__":
    main()
