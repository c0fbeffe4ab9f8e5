"""Host-side mirror of the reference's mel front-end call surface (audio/stft.py:128-178, audio/tools.py:8-15):
``TacotronSTFT(filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax)``
with ``mel_spectrogram(y)`` and ``get_mel_from_wav(audio, stft)``.  The arithmetic (clip, reflect pad,
windowed DFT as a GEMM on the matrix cores, magnitude, Slaney mel filterbank, log clamp, energy) runs in
libdexamd.so (``dex_mel_from_wav``); only the reference's fixed configuration 1024/256/1024/80/22050/0/8000
(config/*/base.yaml:14-21, synthesize.py:79-85) is supported."""
from __future__ import annotations

import numpy as np
import torch

from .engine import ScoreNetEngine
from .config import gedex_lj

_FIXED = (1024, 256, 1024, 80, 22050, 0, 8000)


class TacotronSTFT:
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0, mel_fmax=8000, device=None):
        got = (filter_length, hop_length, win_length, n_mel_channels, sampling_rate, int(mel_fmin), int(mel_fmax))
        if got != _FIXED:
            raise ValueError(f"only the reference configuration {_FIXED} is built into the HIP front-end, got {got}")
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._eng = ScoreNetEngine(gedex_lj(), self.device)      # context only; no score-net weights needed

    def mel_spectrogram(self, y: torch.Tensor):
        """y: [B, T] in [-1, 1] -> (mel [B, 80, frames], energy [B, frames]) — audio/stft.py:159-178."""
        if torch.min(y) < -1 or torch.max(y) > 1:
            raise AssertionError("input must lie in [-1, 1]")            # stft.py:169-170
        mels, ens = zip(*(self._eng.mel_from_wav(row) for row in y))
        return torch.stack(mels), torch.stack(ens)


def get_mel_from_wav(audio, _stft: TacotronSTFT):
    """audio/tools.py:8-15: clip to [-1,1], mel + energy as float32 numpy arrays."""
    wav = torch.clip(torch.as_tensor(np.asarray(audio), dtype=torch.float32), -1, 1)
    mel, energy = _stft._eng.mel_from_wav(wav)
    return mel.cpu().numpy().astype(np.float32), energy.cpu().numpy().astype(np.float32)
