"""DEX style encoders.  CPU: the oracle (oracle/style_oracle.py) and the host mirror's checkpoint surface /
BatchNorm fold against fixtures from the real reference modules (tests/golden/style.npz, manifest_style_vctk.json, written
by oracle/make_golden_style.py).  GPU (-m gpu): dex_style_encode through the C ABI against the golden and the oracle.
fp32 tolerance: exact-fp32 MFMA contractions in another summation order than oneDNN's, a two-layer GRU over T steps:
max|d| <= 2e-4 * max(1, |ref|max) (measured ~1e-5); the VQ indices must agree exactly."""
import json
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import style as S, synth
from oracle import style_oracle as SO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def weights():
    return synth.make_style_weights(S.param_shapes(S.VCTK))


def test_param_shapes_match_reference_state_dict():
    man = json.load(open(os.path.join(GOLD, "manifest_style_vctk.json")))
    assert {k: tuple(v) for k, v in man["keys"].items()} == {k: tuple(v) for k, v in S.param_shapes(S.VCTK).items()}
    assert man["config"]["tv_encoder"] == S.VCTK["tv_encoder"] and man["config"]["tiv_encoder"] == S.VCTK["tiv_encoder"]


def _case(B=2, T=40, lengths=(40, 27)):
    mel, lf0, L = synth.make_style_inputs(B, T, list(lengths))
    return torch.from_numpy(mel), torch.from_numpy(lf0), torch.from_numpy(L)


def test_oracle_matches_reference_golden():
    g = dict(np.load(os.path.join(GOLD, "style.npz")))
    W = {k: torch.from_numpy(v) for k, v in weights().items()}
    mel, lf0, L = _case()
    with torch.no_grad():
        o = SO.style_forward(W, mel, L, mel, L, lf0, L)
    assert np.abs(o["sty_enc"].numpy() - g["sty_enc"]).max() <= 1e-6
    assert np.abs(o["sty_dec"].numpy() - g["sty_dec"]).max() <= 1e-6
    assert np.abs(np.stack([s.numpy() for s in o["ref_skips"]]) - g["ref_skips"]).max() <= 1e-6
    assert np.array_equal(o["vq_idx"].numpy().astype(np.int32), g["vq_idx"])


def test_batchnorm_fold_equals_eval_batchnorm():
    W = {k: torch.from_numpy(v) for k, v in weights().items()}
    f = S.fold_batchnorm(W)
    x = torch.from_numpy(synth.normalish("bnx", (2, 128, 17), 3))
    with torch.no_grad():
        ref = SO.basic_conv(W, "tiv_encoder.conv_blocks.2.conv_block.0", x, True, "bn")
        got = torch.relu(torch.nn.functional.conv1d(x, f["tiv_encoder.conv_blocks.2.conv_block.0.conv.weight"],
                                                    f["tiv_encoder.conv_blocks.2.conv_block.0.conv.bias"], padding=1))
    assert (ref - got).abs().max() <= 2e-6
    assert not any(".bn." in k for k in f)


def test_module_checkpoint_surface():
    m = S.StyleEncoders()
    w = weights()
    full = {"decoder.x": torch.zeros(1), **{k: torch.from_numpy(v) for k, v in w.items()}}
    m.load_state_dict(full, strict=False)                       # a whole DeXTTS checkpoint: foreign keys ignored
    assert set(m.state_dict()) == set(w)
    with pytest.raises(RuntimeError):
        m.load_state_dict({"conv_sty.bias": torch.zeros(128)})
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 80, 8), torch.tensor([8]), torch.zeros(1, 80, 8), torch.tensor([8]), torch.zeros(1, 8), torch.tensor([8]))


# ---------------------------------------------------------------------------------------------------- GPU
def _gpu_module():
    m = S.StyleEncoders()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights().items()})
    return m.cuda().eval()


def _close(got, ref, tag):
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all() and err <= 2e-4 * max(1.0, np.abs(ref).max()), (tag, float(err))


@pytest.mark.gpu
def test_style_matches_reference_golden():
    g = dict(np.load(os.path.join(GOLD, "style.npz")))
    m = _gpu_module()
    mel, lf0, L = _case()
    skips, sty_dec, sty_enc, idx = m(mel.cuda(), L.cuda(), mel.cuda(), L.cuda(), lf0.cuda(), L.cuda(), return_indices=True)
    assert np.array_equal(idx.cpu().numpy(), g["vq_idx"])
    _close(sty_enc.cpu().numpy(), g["sty_enc"], "sty_enc")
    _close(sty_dec.cpu().numpy(), g["sty_dec"], "sty_dec")
    _close(np.stack([s.cpu().numpy() for s in skips]), g["ref_skips"], "ref_skips")


@pytest.mark.gpu
@pytest.mark.parametrize("B,Tr,Ts,Tl,lens", [(1, 348, 348, 348, [348]), (3, 50, 37, 61, [50, 20, 33]), (2, 2, 1, 1, [2, 1])])
def test_style_matches_oracle(B, Tr, Ts, Tl, lens):
    m = _gpu_module()
    ref, _, _ = synth.make_style_inputs(B, Tr, None, seed=5)
    sty, _, _ = synth.make_style_inputs(B, Ts, None, seed=6)
    _, lf0, _ = synth.make_style_inputs(B, Tl, None, seed=7)
    rl = torch.tensor([min(l, Tr) for l in lens]); sl = torch.tensor([min(l, Ts) for l in lens]); ll = torch.tensor([min(l, Tl) for l in lens])
    t = torch.from_numpy
    skips, sty_dec, sty_enc, idx = m(t(ref).cuda(), rl.cuda(), t(sty).cuda(), sl.cuda(), t(lf0).cuda(), ll.cuda(), return_indices=True)
    W = {k: t(v) for k, v in weights().items()}
    with torch.no_grad():
        o = SO.style_forward(W, t(ref), rl, t(sty), sl, t(lf0), ll)
    flips = int((idx.cpu().numpy() != o["vq_idx"].numpy()).sum())             # fixed seeds: no exact tie between two codes
    assert flips == 0, f"{flips} VQ code flips"
    _close(sty_enc.cpu().numpy(), o["sty_enc"].numpy(), "sty_enc")
    _close(np.stack([s.cpu().numpy() for s in skips]), np.stack([s.numpy() for s in o["ref_skips"]]), "ref_skips")
    _close(sty_dec.cpu().numpy(), o["sty_dec"].numpy(), "sty_dec")


@pytest.mark.gpu
def test_style_feeds_the_decoder():
    """End of the chain: the encoders' outputs go straight into Diffusion.forward's ref / sty arguments (tts.py:84)."""
    from tests import gpu_util as U
    cfg, eng, w = U.engine_for("dex_vctk")
    m = _gpu_module()
    mel, lf0, L = _case(1, 40, (33,))
    skips, sty_dec, sty_enc = m(mel.cuda(), L.cuda(), mel.cuda(), L.cuda(), lf0.cuda(), L.cuda())
    mu, mask, z, _ = synth.make_inputs(1, 64, [57])
    out = eng.sample(torch.from_numpy(z), torch.from_numpy(mask), torch.from_numpy(mu), 4, ref=skips, sty=sty_dec, sty_lengths=L)
    assert out.shape == (1, 80, 64) and torch.isfinite(out).all()
