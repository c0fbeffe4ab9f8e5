"""TTS-level mirrors (dex_tts_amd/tts.py: GeDEXTTS / DeXTTS, reference tts.py): construction from the reference's model
sections, checkpoint routing by sub-module prefix, and (GPU) text -> mel wiring: the module's output equals its stages called
one by one under the same seed (each stage is pinned to the reference by its own tests)."""
import json
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import config as C, style as S, synth, text as T, tts

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SECTIONS = json.load(open(os.path.join(GOLD, "ref_model_sections.json")))


def model_cfg(which):
    """cfg.model of a shipped base.yaml, reassembled from the committed fixtures (decoder / dit sections, encoder and style manifests)."""
    yml, tman = {"gedex_lj": ("GeDEX-TTS/config/LJSpeech/base.yaml", "gedex_lj"), "gedex_vctk": ("GeDEX-TTS/config/VCTK/base.yaml", "gedex_vctk"),
                 "dex_vctk": ("DEX-TTS/config/VCTK/base.yaml", "dex_vctk")}[which]
    m = dict(SECTIONS[yml])
    enc = json.load(open(os.path.join(GOLD, f"manifest_text_{tman}.json")))["config"]
    m["encoder"] = {k: enc[k] for k in ("n_channels", "filter_channels", "filter_channels_dp", "n_layers", "kernel_size", "p_dropout", "n_heads",
                                        "window_size", "use_softmax", "use_decay")}
    m["n_vocab"] = enc["n_vocab"]
    if which.startswith("dex"):
        sm = json.load(open(os.path.join(GOLD, "manifest_style_vctk.json")))["config"]
        m.update(tv_encoder=sm["tv_encoder"], lf0_encoder=sm["lf0_encoder"], tiv_encoder=sm["tiv_encoder"])
    return m


def full_state_dict(model, which):
    sd = {}
    sd.update({"encoder." + k: torch.from_numpy(v) for k, v in synth.make_text_weights(model.encoder.shapes).items()})
    sd.update({"decoder." + k: v.clone() for k, v in model.decoder.state_dict().items()})
    dw = synth.make_weights(C.param_shapes(model.decoder.cfg))
    for k, v in dw.items():
        sd["decoder.denoise_fn." + k] = torch.from_numpy(v)
        sd["decoder.precond_model.model." + k] = torch.from_numpy(v)
    if hasattr(model, "spk_emb"):
        sd["spk_emb.weight"] = torch.from_numpy(synth.normalish("spk_emb", tuple(model.spk_emb.weight.shape), 2))
    if which.startswith("dex"):
        sd.update({k: torch.from_numpy(v) for k, v in synth.make_style_weights(model.style.shapes).items()})
    return sd


@pytest.mark.parametrize("which", ["gedex_lj", "gedex_vctk", "dex_vctk"])
def test_construct_and_route_checkpoint(which):
    m = (tts.DeXTTS if which.startswith("dex") else tts.GeDEXTTS)(model_cfg(which))
    sd = full_state_dict(m, which)
    m.load_state_dict(sd)
    assert torch.equal(m.encoder.state_dict()["proj_m.bias"], sd["encoder.proj_m.bias"])
    k = next(iter(C.param_shapes(m.decoder.cfg)))
    assert torch.equal(m.decoder.state_dict()["denoise_fn." + k], sd["decoder.denoise_fn." + k])
    with pytest.raises(RuntimeError):
        m.load_state_dict(dict(sd, bogus=torch.zeros(1)))
    with pytest.raises(NotImplementedError):
        m.compute_loss()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["gedex_vctk", "dex_vctk"])
def test_text_to_mel_equals_its_stages(which):
    m = (tts.DeXTTS if which.startswith("dex") else tts.GeDEXTTS)(model_cfg(which))
    m.load_state_dict(full_state_dict(m, which))
    m = m.cuda().eval()
    tok, lengths = synth.make_text_inputs(2, 19, [19, 11], 149)
    x, xl = torch.from_numpy(tok).cuda(), torch.from_numpy(lengths).cuda()
    if which.startswith("dex"):
        mel, lf0, L = synth.make_style_inputs(2, 40, [40, 27])
        ref, lf0, L = torch.from_numpy(mel).cuda(), torch.from_numpy(lf0).cuda(), torch.from_numpy(L).cuda()
        torch.manual_seed(7)
        enc_out, dec_out, attn = m(x, xl, ref, L, ref, L, lf0, L, n_timesteps=4, temperature=1.5)
        skips, sty_dec, sty_enc = m.style(ref, L, ref, L, lf0, L)
        m.encoder(x, xl, sty_enc)
        mu_y, y_mask, attn2, y_len, y_max = m.encoder.align()
        torch.manual_seed(7)
        want = m.decoder(mu_y, y_mask, mu_y, skips, L, sty_dec, L, temperature=1.5, n_timesteps=4, spk=None, infer=True)
    else:
        spk = torch.tensor([3, 77]).cuda()
        torch.manual_seed(7)
        enc_out, dec_out, attn = m(x, xl, n_timesteps=4, temperature=1.5, spk=spk)
        e = m.spk_emb(spk)
        m.encoder(x, xl, spk=e)
        mu_y, y_mask, attn2, y_len, y_max = m.encoder.align()
        torch.manual_seed(7)
        want = m.decoder(mu_y, y_mask, mu_y, temperature=1.5, n_timesteps=4, spk=e, infer=True)
    assert dec_out.shape == (2, 80, y_max) and enc_out.shape == (2, 80, y_max) and torch.isfinite(dec_out).all()
    assert torch.equal(dec_out, want[:, :, :y_max]) and torch.equal(enc_out, mu_y[:, :, :y_max])
    assert int(y_len.max()) == y_max and mu_y.shape[2] % 4 == 0
    # frames past an utterance's length carry nothing
    for b in range(2):
        assert float(enc_out[b, :, int(y_len[b]):].abs().max() if int(y_len[b]) < y_max else 0.0) == 0.0


@pytest.mark.gpu
def test_tokens_to_waveform():
    """synthesize.py:31-38 end to end on the GPU: tokens -> text encoder -> alignment -> 10 sampler steps -> HiFi-GAN -> int16 audio."""
    from dex_tts_amd import synthesize as SY, vocoder as V
    m = tts.GeDEXTTS(model_cfg("gedex_lj"))
    m.load_state_dict(full_state_dict(m, "gedex_lj"))
    m = m.cuda().eval()
    voc = V.Generator()
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_vocoder_weights(V.param_shapes(V.HIFIGAN_V1)).items()})
    voc = voc.cuda().eval()
    tok, lengths = synth.make_text_inputs(2, 21, [21, 12], 149)
    SY.seed_init(100)
    audio, y_dec, attn = SY.synthesize_tokens(m, voc, torch.from_numpy(tok).cuda(), torch.from_numpy(lengths).cuda(), n_timesteps=10)
    y_len = m.encoder._last["y_len"].cpu().numpy()
    assert len(audio) == 2 and all(a.dtype == np.int16 for a in audio)
    assert [len(a) for a in audio] == [int(n) * 256 for n in y_len]
    assert torch.isfinite(y_dec).all() and all(np.abs(a.astype(np.int32)).max() > 0 for a in audio)
    SY.seed_init(100)                                   # same seed, same waveform (bitwise)
    audio2, _, _ = SY.synthesize_tokens(m, voc, torch.from_numpy(tok).cuda(), torch.from_numpy(lengths).cuda(), n_timesteps=10)
    assert all(np.array_equal(a, b) for a, b in zip(audio, audio2))
