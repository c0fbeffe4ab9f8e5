"""Checker infrastructure (build container only): END-TO-END goldens of the two TTS facades — the REAL reference
``GeDEXTTS.forward`` (GeDEX-TTS/model/tts.py:27-56) and ``DeXTTS.forward`` (DEX-TTS/model/tts.py:33-74), imported from
/root/reference and run on the portable synthetic weights / inputs, with the one random draw of the path
(``torch.randn`` in Diffusion.forward, diffusion.py:227) replaced by a stored tensor.

    python -m oracle.make_golden_tts          ->  tests/golden/tts_{gedex_lj,gedex_vctk,dex_vctk}.npz

What is stored: inputs (tokens, lengths, speaker ids / style inputs), the latent draw z0 (what ``torch.randn`` returned) and the
reference's three outputs ``enc_out / dec_out / attn`` plus ``y_lengths`` — i.e. BASELINE.json configs[0]'s plumbing job
(duration ceil -> y_lengths -> fix_len_compatibility padding -> generate_path -> mu_y -> sampler -> crop) as the reference
itself performs it.  tests/test_tts_golden.py compares (CPU) the chained oracle restatements and (GPU) ``dex_tts_amd.tts``
against it.  Stand-ins (none of them on the arithmetic path): ``model.monotonic_align`` (training-only Cython, tts.py:9 imports
it at module level), the ones ``oracle/ref_import.import_reference_text`` documents.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from dex_tts_amd import config as C, synth  # noqa: E402
from oracle import ref_import  # noqa: E402

N_VOCAB = 149
CASES = {          # name -> (reference tree, YAML, n_spks, B, L, lengths, n_timesteps, length_scale)
    "gedex_lj": ("GeDEX-TTS", "config/LJSpeech/base.yaml", 1, 2, 21, [21, 12], 10, 1.0),
    "gedex_vctk": ("GeDEX-TTS", "config/VCTK/base.yaml", 108, 2, 19, [19, 11], 4, 1.0),
    "dex_vctk": ("DEX-TTS", "config/VCTK/base.yaml", 0, 1, 17, [17], 4, 1.0),      # the reference runs DEX one utterance at a time
}


class Attr(dict):
    """cfg.model with attribute access AND assignment (DeXTTS.__init__ writes cfg.n_spks, tts.py:18; Diffusion writes dit_cfg fields)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def to_attr(d):
    return Attr({k: to_attr(v) if isinstance(v, dict) else v for k, v in d.items()})


def import_tts(sub):
    ref_import.import_reference_text(sub)                  # timm / transformers stand-ins, 'model' package without its __init__
    ma = types.ModuleType("model.monotonic_align")         # tts.py:9: training only (compute_loss), never called here

    def maximum_path(*a, **k):
        raise RuntimeError("monotonic_align stand-in: training only")

    ma.maximum_path = maximum_path
    sys.modules["model.monotonic_align"] = ma
    sys.modules["model"].monotonic_align = ma
    return importlib.import_module("model.tts")


@torch.no_grad()
def run(name):
    sub, yml, n_spks, B, L, lengths, n_steps, length_scale = CASES[name]
    m = yaml.safe_load(open(f"/root/reference/{sub}/{yml}"))["model"]
    cfg = to_attr(dict(m, n_vocab=N_VOCAB, n_spks=n_spks))
    tts = import_tts(sub)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = (tts.DeXTTS if sub == "DEX-TTS" else tts.GeDEXTTS)(cfg).eval()
    c = model.encoder.encoder.config
    for k, v in dict(use_cache=True, output_retentions=False, output_hidden_states=False).items():
        if not hasattr(c, k):
            setattr(c, k, v)
    # ---- portable weights, sub-module by sub-module (the generators key on the sub-module's own parameter names)
    sd = model.state_dict()
    new = {}
    enc_shapes = {k[len("encoder."):]: list(v.shape) for k, v in sd.items() if k.startswith("encoder.")}
    tw = synth.make_text_weights(enc_shapes)
    for k in ("encoder.retnet_rel_pos.angle", "encoder.retnet_rel_pos.decay"):      # registered buffers: the reference's own values travel
        tw[k] = sd["encoder." + k].numpy().copy()
    new.update({"encoder." + k: v for k, v in tw.items()})
    scfg = (C.dex_vctk() if sub == "DEX-TTS" else (C.gedex_vctk() if n_spks > 1 else C.gedex_lj()))
    dw = synth.make_weights(C.param_shapes(scfg))
    for k, v in dw.items():
        new["decoder.denoise_fn." + k] = v
        new["decoder.precond_model.model." + k] = v
    if n_spks > 1:
        new["spk_emb.weight"] = synth.normalish("spk_emb", tuple(sd["spk_emb.weight"].shape), 2)
    if sub == "DEX-TTS":
        st_shapes = {k: list(v.shape) for k, v in sd.items() if k.split(".")[0] in ("tv_encoder", "lf0_encoder", "tiv_encoder", "conv_sty")}
        new.update(synth.make_style_weights(st_shapes))
    missing = set(sd) - set(new)
    extra = set(new) - set(sd)
    assert not missing and not extra, (sorted(missing)[:6], sorted(extra)[:6])
    model.load_state_dict({k: torch.as_tensor(v) for k, v in new.items()}, strict=True)
    # ---- inputs
    tok, lengths = synth.make_text_inputs(B, L, lengths, N_VOCAB)
    x, xl = torch.from_numpy(tok), torch.from_numpy(lengths)
    out = {"tokens": tok, "lengths": lengths, "n_timesteps": np.int64(n_steps), "temperature": np.float32(1.5),
           "length_scale": np.float32(length_scale), "angle": tw["encoder.retnet_rel_pos.angle"], "decay": tw["encoder.retnet_rel_pos.decay"]}
    # ---- the one random draw (diffusion.py:227): a stored tensor
    drawn = {}
    real_randn = torch.randn

    def fixed_randn(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        assert "z0" not in drawn, "one draw expected"
        drawn["z0"] = synth.normalish("tts_z0", shape, 4242)
        return torch.from_numpy(drawn["z0"])

    dmod = sys.modules["model.diffusion"]
    dmod.torch.randn = fixed_randn
    try:
        if sub == "DEX-TTS":
            mel, lf0, SL = synth.make_style_inputs(B, 40, [33][:B])
            ref, lf0_t, SLt = torch.from_numpy(mel), torch.from_numpy(lf0), torch.from_numpy(SL)
            enc_out, dec_out, attn = model(x, xl, ref, SLt, ref, SLt, lf0_t, SLt, n_timesteps=n_steps, temperature=1.5, length_scale=length_scale)
            out.update(style_mel=mel, style_lf0=lf0, style_lengths=SL)
        else:
            spk = torch.tensor([3, 77][:B]) if n_spks > 1 else None
            enc_out, dec_out, attn = model(x, xl, n_timesteps=n_steps, temperature=1.5, spk=spk, length_scale=length_scale)
            if spk is not None:
                out["spk"] = spk.numpy()
    finally:
        dmod.torch.randn = real_randn
    # y_lengths as the reference computes them (tts.py:38-39) is not returned: recover it from the alignment (every frame of an
    # utterance is owned by exactly one token, frames beyond y_length by none)
    y_len = attn[:, 0].sum(dim=1).gt(0).sum(dim=1).numpy().astype(np.int64)
    out.update(z0=drawn["z0"], enc_out=enc_out.numpy(), dec_out=dec_out.numpy(), attn=attn.numpy().astype(np.int8), y_lengths=y_len)
    np.savez_compressed(os.path.join(OUT, f"tts_{name}.npz"), **out)
    print(f"tts_{name}: enc_out {tuple(enc_out.shape)} dec_out |max| {float(dec_out.abs().max()):.3f} y_lengths {y_len.tolist()} z0 {drawn['z0'].shape}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for n in CASES:
        run(n)
