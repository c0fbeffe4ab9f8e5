"""CPU: the host logic of the synthesize-flow driver (dex_tts_amd/synthesize.py) and the reference-YAML config loader
against fixtures extracted from the reference's shipped configs (tests/golden/ref_model_sections.json, written by
oracle/dump_ref_configs.py) and the state-dict manifests dumped from the real reference."""
import json
import os

import pytest
import torch

from dex_tts_amd import config as C, synthesize as S

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SECTIONS = json.load(open(os.path.join(GOLD, "ref_model_sections.json")))


@pytest.mark.parametrize("rel,manifest", [
    ("GeDEX-TTS/config/LJSpeech/base.yaml", "manifest_gedex_lj.json"),
    ("GeDEX-TTS/config/VCTK/base.yaml", "manifest_gedex_vctk.json"),
    ("DEX-TTS/config/VCTK/base.yaml", "manifest_dex_vctk.json"),
    ("DEX-TTS/config/ESD/base.yaml", "manifest_dex_vctk.json"),          # same model section as VCTK
    ("DEX-TTS/config/LibriTTS/base.yaml", "manifest_dex_libritts.json"),
])
def test_from_reference_yaml_matches_reference_state_dict(rel, manifest):
    sec = SECTIONS[rel]
    cfg = S.config_from_model_section(sec, sec["variant"])
    keys = json.load(open(os.path.join(GOLD, manifest)))["keys"]         # the real reference's decoder state dict
    want = {k[len("denoise_fn."):]: tuple(v) for k, v in keys.items() if k.startswith("denoise_fn.")}
    got = {k: tuple(v) for k, v in C.param_shapes(cfg).items()}
    assert got == want


def test_shipped_configs_equal_presets():
    for rel, preset in [("GeDEX-TTS/config/LJSpeech/base.yaml", "gedex_lj"), ("GeDEX-TTS/config/VCTK/base.yaml", "gedex_vctk"),
                        ("DEX-TTS/config/VCTK/base.yaml", "dex_vctk"), ("DEX-TTS/config/ESD/base.yaml", "dex_esd")]:
        sec = SECTIONS[rel]
        assert S.config_from_model_section(sec, sec["variant"]).to_dict() == C.PRESETS[preset]().to_dict(), rel
    lib = SECTIONS["DEX-TTS/config/LibriTTS/base.yaml"]
    cfg = S.config_from_model_section(lib, "dex")
    assert (cfg.dim, cfg.dit.hidden_size, cfg.mid_dim) == (128, 384, 256)


def test_prepare_pads_masks_and_crops():
    lengths = torch.tensor([210, 57])
    mu = torch.randn(2, 80, 210)
    mu_p, mask, y_max = S.prepare(mu, lengths)
    assert y_max == 210 and mu_p.shape == (2, 80, 212) and mask.shape == (2, 1, 212)      # fix_len_compatibility(210) = 212
    assert mask[0, 0].sum() == 210 and mask[1, 0].sum() == 57
    assert torch.equal(mu_p[0, :, :210], mu[0]) and mu_p[1, :, 57:].abs().max() == 0
    assert S.sequence_mask(torch.tensor([3, 1])).tolist() == [[True, True, True], [True, False, False]]
    assert C.fix_len_compatibility(212) == 212 and C.fix_len_compatibility(213) == 216


def test_decoder_state_dict_filter():
    ck = {"ema": {"decoder.denoise_fn.mlp.0.weight": 1, "encoder.x": 2}, "state_dict": {"decoder.a": 3}}
    assert S.decoder_state_dict(ck) == {"denoise_fn.mlp.0.weight": 1}
    assert S.decoder_state_dict(ck, ema=False) == {"a": 3}
