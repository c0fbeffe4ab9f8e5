"""The preprocess-side front-end (north_star: "the STFT/mel front-end (preprocess + audio)"): batched TacotronSTFT.mel_spectrogram
(audio/stft.py:159-178, caller preprocess/preprocessor/preprocessor.py:100) and the deterministic tail of the DEX f0 front-end
(log f0 + normalize_lf0, DEX-TTS/synthesize.py:26-38,55-58).  Goldens: tests/golden/audio_mel.npz (real reference TacotronSTFT),
tests/golden/lf0.npz (the reference's own normalize_lf0, oracle/make_golden_lf0.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import style_oracle as SO

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LF0 = dict(np.load(os.path.join(GOLD, "lf0.npz")))
TRACKS = sorted(k[:-3] for k in LF0 if k.endswith("_f0"))


@pytest.mark.parametrize("k", TRACKS)
def test_oracle_lf0_bit_exact_with_reference(k):
    assert np.array_equal(SO.lf0_from_f0(LF0[f"{k}_f0"]), LF0[f"{k}_lf0"])


def test_constant_track_is_roundoff_in_the_reference():
    """Why the kernel repeats numpy's summation order: a constant 200 Hz track does NOT normalise to 0 in the reference."""
    v = LF0["constant_lf0"]
    assert np.abs(v).max() > 0.5 and len(np.unique(v)) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("k", TRACKS)
def test_gpu_lf0_matches_reference_golden(k):
    """Against the reference's own outputs, and — operation by operation — against numpy's float32 arithmetic on the DEVICE's
    log values.  The second check is what pins the constant track: there the reference's std is pure summation round-off
    (5e-7 instead of 0), so its output (-0.979...) moves by 1 % when ONE log value moves by one ulp, as the device's logf does
    (measured: -0.9896); given the same log values the kernel and numpy agree to the last bits."""
    from dex_tts_amd.audio import lf0_from_f0
    f0 = torch.from_numpy(LF0[f"{k}_f0"]).cuda()
    got = lf0_from_f0(f0).cpu().numpy()
    want = LF0[f"{k}_lf0"]
    assert np.array_equal(got == 0, want == 0)
    if k != "constant":
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), float(np.abs(got - want).max())
    dev_log = torch.where(f0 != 0, torch.log(f0), torch.zeros_like(f0)).cpu().numpy()
    same = SO.normalize_lf0(dev_log)
    assert np.abs(got - same).max() <= 2e-6 * max(1.0, np.abs(same).max()), float(np.abs(got - same).max())


@pytest.mark.gpu
def test_gpu_lf0_batched_with_lengths():
    """[B,T] with lengths: frames past an utterance's length neither enter the statistics nor come back non-zero."""
    from dex_tts_amd.audio import lf0_from_f0
    a, e = LF0["contour_f0"], LF0["one_hz_f0"]
    T = len(a)
    f0 = np.zeros((3, T), np.float32)
    f0[0] = a; f0[1, :100] = e; f0[1, 100:] = 777.0; f0[2, :40] = LF0["single_f0"]; f0[2, 40:] = 55.0      # garbage past the lengths
    got = lf0_from_f0(torch.from_numpy(f0).cuda(), torch.tensor([T, 100, 40])).cpu().numpy()
    assert np.abs(got[0] - LF0["contour_lf0"]).max() <= 1e-5 * 3
    assert np.abs(got[1, :100] - LF0["one_hz_lf0"]).max() <= 1e-5 * 3 and not got[1, 100:].any()
    assert not got[2].any()                                           # a single voiced frame: lf0 - mean = 0


@pytest.mark.gpu
def test_gpu_batched_mel_spectrogram():
    """mel_spectrogram(y [B,L]) in ONE pass: row 0 against the real reference's output, every row bitwise equal to the
    single-utterance entry point (same kernels), the [-1,1] assert of stft.py:169-170 kept."""
    from dex_tts_amd.audio import TacotronSTFT, get_mel_from_wav
    from tests import gpu_util as U
    g = dict(np.load(os.path.join(GOLD, "audio_mel.npz")))
    w = g["sample1_1s_wav"]
    y = np.stack([w, 0.5 * w[::-1], np.roll(w, 1234)]).astype(np.float32)
    st = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000)
    mel, energy = st.mel_spectrogram(torch.from_numpy(y))
    assert mel.shape == (3, 80, len(w) // 256 + 1) and energy.shape == (3, len(w) // 256 + 1)
    assert np.abs(mel[0].cpu().numpy() - g["sample1_1s_mel"]).max() <= 2e-3
    np.testing.assert_allclose(energy[0].cpu().numpy(), g["sample1_1s_energy"], rtol=2e-4, atol=1e-4)
    cfg, eng, _ = U.engine_for("gedex_lj")
    for b in range(3):
        m1, e1 = eng.mel_from_wav(torch.from_numpy(y[b]))
        assert torch.equal(m1, mel[b]) and torch.equal(e1, energy[b])
    m2, e2 = get_mel_from_wav(g["chirp_wav"], st)                      # clips (|chirp| reaches 1.2), like tools.py:9
    assert np.abs(m2 - g["chirp_mel"]).max() <= 2e-3
    with pytest.raises(AssertionError):
        st.mel_spectrogram(torch.from_numpy(y * 3.0))
    with pytest.raises(ValueError):
        TacotronSTFT(2048, 256, 1024, 80, 22050, 0, 8000)


def test_front_end_argument_errors_need_no_gpu():
    """Configuration errors are raised before anything touches the device; the product path has no CPU fallback."""
    from dex_tts_amd.audio import TacotronSTFT, lf0_from_f0
    with pytest.raises(ValueError):
        TacotronSTFT(2048, 256, 1024, 80, 22050, 0, 8000)
    with pytest.raises(RuntimeError):
        lf0_from_f0(torch.zeros(8))                      # CPU tensor: refused, not emulated
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000, device="cpu")
