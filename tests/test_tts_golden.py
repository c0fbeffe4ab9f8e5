"""End-to-end goldens of the TTS facades (VERDICT r3 Missing #2): tests/golden/tts_*.npz hold what the REAL reference
``GeDEXTTS.forward`` (GeDEX-TTS/model/tts.py:27-56) / ``DeXTTS.forward`` (DEX-TTS/model/tts.py:33-74) return on the portable
synthetic weights and inputs, with the latent draw of Diffusion.forward stored (oracle/make_golden_tts.py).  This pins the
plumbing of BASELINE configs[0] — duration ceil -> y_lengths -> fix_len_compatibility padding -> generate_path -> mu_y ->
sampler -> crop — against the reference itself, not against the build's own stages:
  * CPU: the chained oracle restatements (style_oracle -> text_oracle -> dex_oracle) reproduce the reference's outputs;
  * GPU: ``dex_tts_amd.tts.{GeDEXTTS, DeXTTS}.forward`` (every stage in libdexamd.so) reproduces them within the fp32 bounds."""
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import config as C, synth
from tests.test_tts_module import full_state_dict, model_cfg

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["gedex_lj", "gedex_vctk", "dex_vctk"]


def load(name):
    return dict(np.load(os.path.join(GOLD, f"tts_{name}.npz")))


@pytest.mark.parametrize("which", CASES)
def test_oracle_chain_reproduces_reference_forward(which):
    from oracle import dex_oracle as O, style_oracle as SO, text_oracle as TO
    g = load(which)
    mcfg = model_cfg(which)
    enc = mcfg["encoder"]
    n_spks = {"gedex_lj": 1, "gedex_vctk": 108, "dex_vctk": 0}[which]
    from dex_tts_amd import text as T
    shapes = T.param_shapes(mcfg["n_vocab"], 80, enc["n_channels"], enc["filter_channels"], enc["filter_channels_dp"], enc["n_heads"], enc["n_layers"],
                            enc["kernel_size"], 64, n_spks, "dex" if which.startswith("dex") else "gedex")
    tw = synth.make_text_weights(shapes)
    tw["encoder.retnet_rel_pos.angle"], tw["encoder.retnet_rel_pos.decay"] = g["angle"], g["decay"]
    W = {k: torch.from_numpy(v) for k, v in tw.items()}
    x, xl = torch.from_numpy(g["tokens"]), torch.from_numpy(g["lengths"])
    tcfg = dict(n_channels=enc["n_channels"], n_layers=enc["n_layers"], n_heads=enc["n_heads"], n_spks=n_spks, kernel_size=enc["kernel_size"])
    kw, spk, sty_enc = {}, None, None
    scfg = {"gedex_lj": C.gedex_lj, "gedex_vctk": C.gedex_vctk, "dex_vctk": C.dex_vctk}[which]()
    if which.startswith("dex"):
        from dex_tts_amd import style as S
        sm = dict(tv_encoder=mcfg["tv_encoder"], lf0_encoder=mcfg["lf0_encoder"], tiv_encoder=mcfg["tiv_encoder"], dim=mcfg["decoder"]["dim"])
        sw = {k: torch.from_numpy(v) for k, v in synth.make_style_weights(S.param_shapes(sm)).items()}
        mel, lf0, SL = (torch.from_numpy(g[k]) for k in ("style_mel", "style_lf0", "style_lengths"))
        so = SO.style_forward(sw, mel, SL, mel, SL, lf0, SL)
        sty_enc = so["sty_enc"]
        kw = dict(ref=list(so["ref_skips"]), sty=so["sty_dec"], sty_lengths=SL)
    elif n_spks > 1:
        spk = torch.from_numpy(synth.normalish("spk_emb", (n_spks, 64), 2))[torch.from_numpy(g["spk"])]
        kw = dict(spk=spk)
    mu_x, logw, x_mask = TO.text_encoder_forward(W, tcfg, x, xl, spk=spk, sty=sty_enc)
    a = TO.align(mu_x, logw, x_mask, float(g["length_scale"]))
    assert a["y_lengths"].tolist() == g["y_lengths"].tolist()
    y_max = int(a["y_max_length"])
    # (the reference returns attn[:, :, :y_max_length], tts.py:56 — a slice of the TOKEN axis of [B, 1, Tx, Ty_], i.e. the frame axis
    # keeps its fix_len_compatibility padding; the mirror repeats the expression)
    assert np.array_equal(a["attn"][:, :, :y_max].numpy().astype(np.int8), g["attn"])
    enc_out = a["mu_y"][:, :, :y_max].numpy()
    assert np.abs(enc_out - g["enc_out"]).max() <= 1e-5 * max(1.0, np.abs(g["enc_out"]).max())
    assert g["z0"].shape[2] == a["y_max_length_"] and a["y_max_length_"] % 4 == 0
    z = torch.from_numpy(g["z0"]) / float(g["temperature"]) + a["mu_y"]
    DW = O.as_torch(synth.make_weights(C.param_shapes(scfg)), torch.float32)
    dec = O.diffusion_infer(DW, scfg, a["y_mask"], a["mu_y"], int(g["n_timesteps"]), z, **kw)[:, :, :y_max].numpy()
    err = np.abs(dec - g["dec_out"])
    assert err.max() <= 2e-4 and err.mean() <= 2e-5, (err.max(), err.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("which", CASES)
def test_tts_forward_reproduces_reference_forward(which, monkeypatch):
    from dex_tts_amd import tts
    from tests.gpu_util import record
    from tests import tolerances as TL
    g = load(which)
    m = (tts.DeXTTS if which.startswith("dex") else tts.GeDEXTTS)(model_cfg(which))
    sd = full_state_dict(m, which)
    sd["encoder.encoder.retnet_rel_pos.angle"], sd["encoder.encoder.retnet_rel_pos.decay"] = torch.from_numpy(g["angle"]), torch.from_numpy(g["decay"])
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x, xl = torch.from_numpy(g["tokens"]).cuda(), torch.from_numpy(g["lengths"]).cuda()
    z0 = torch.from_numpy(g["z0"]).cuda()
    calls = []

    def fixed_randn(*shape, **kw):          # the module's one draw (dex_tts_amd/diffusion.py, reference diffusion.py:227)
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        assert shape == tuple(z0.shape), (shape, tuple(z0.shape))
        calls.append(shape)
        return z0.clone()

    import dex_tts_amd.diffusion as D
    monkeypatch.setattr(D.torch, "randn", fixed_randn)
    m.decoder.rng_parity = False
    n, temp, ls = int(g["n_timesteps"]), float(g["temperature"]), float(g["length_scale"])
    if which.startswith("dex"):
        mel, lf0, SL = (torch.from_numpy(g[k]).cuda() for k in ("style_mel", "style_lf0", "style_lengths"))
        enc_out, dec_out, attn = m(x, xl, mel, SL, mel, SL, lf0, SL, n_timesteps=n, temperature=temp, length_scale=ls)
    elif "spk" in g:
        enc_out, dec_out, attn = m(x, xl, n_timesteps=n, temperature=temp, spk=torch.from_numpy(g["spk"]).cuda(), length_scale=ls)
    else:
        enc_out, dec_out, attn = m(x, xl, n_timesteps=n, temperature=temp, length_scale=ls)
    assert len(calls) == 1
    assert m.encoder._last["y_len"].cpu().tolist() == g["y_lengths"].tolist()
    assert tuple(dec_out.shape) == g["dec_out"].shape and tuple(attn.shape) == g["attn"].shape
    assert np.array_equal(attn.cpu().numpy().astype(np.int8), g["attn"])
    e = np.abs(enc_out.cpu().numpy() - g["enc_out"])
    assert e.max() <= 3e-4 * max(1.0, np.abs(g["enc_out"]).max()), e.max()          # the text encoder's own bound (tests/test_text.py)
    d = np.abs(dec_out.cpu().numpy() - g["dec_out"])
    record(f"tts_{which}:fp32:facade", max=d.max(), mean=d.mean(), enc_max=e.max())
    # the sampler amplifies the encoder's 1e-4-level differences in mu_y; bounds = 10x the fp32 sampler bounds
    assert np.isfinite(dec_out.cpu().numpy()).all() and d.max() <= 10 * TL.FP32_SAMPLER_MAX and d.mean() <= 10 * TL.FP32_SAMPLER_MEAN, (d.max(), d.mean())
