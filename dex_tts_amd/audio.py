"""Host-side mirror of the reference's mel front-end call surface (audio/stft.py:128-178, audio/tools.py:8-15):
``TacotronSTFT(filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax)``
with ``mel_spectrogram(y)`` and ``get_mel_from_wav(audio, stft)``, plus the deterministic tail of the DEX f0 front-end
(``normalize_lf0`` of DEX-TTS/synthesize.py:26-38 applied to log f0, :55-58).  The arithmetic (clip, reflect pad, windowed
DFT as a batched GEMM on the matrix cores, magnitude, Slaney mel filterbank, log clamp, energy; log-f0 statistics) runs in
libdexamd.so (``dex_mel_spectrogram`` / ``dex_lf0_normalize``) — a whole [B, L] batch in one pass, no score-network context.
Only the reference's fixed configuration 1024/256/1024/80/22050/0/8000 (config/*/base.yaml:14-21, synthesize.py:79-85) is
supported.  There is no CPU path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_FIXED = (1024, 256, 1024, 80, 22050, 0, 8000)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class TacotronSTFT:
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0, mel_fmax=8000, device=None):
        got = (filter_length, hop_length, win_length, n_mel_channels, sampling_rate, int(mel_fmin), int(mel_fmax))
        if got != _FIXED:
            raise ValueError(f"only the reference configuration {_FIXED} is built into the HIP front-end, got {got}")
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("the mel front-end runs on an MI355X only (no CPU path)")
        self._lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.dex_mel_create(C.byref(h))
        if rc != _lib.DEX_OK:
            raise RuntimeError(f"dex_mel_create failed ({rc})")
        self._h = h
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.dex_mel_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _run(self, y: torch.Tensor):
        """y [B, L] on any device -> (mel [B,80,frames], energy [B,frames]) on self.device; rows are clipped to [-1, 1] in the kernel."""
        with torch.cuda.device(self.device):
            y = y.to(device=self.device, dtype=torch.float32).contiguous()
            B, L = y.shape
            frames = self._lib.dex_mel_frames(L)
            need = int(self._lib.dex_mel_workspace_bytes(B, L))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            mel = torch.empty(B, 80, frames, dtype=torch.float32, device=self.device)
            energy = torch.empty(B, frames, dtype=torch.float32, device=self.device)
            rc = self._lib.dex_mel_spectrogram(self._h, y.data_ptr(), B, L, mel.data_ptr(), energy.data_ptr(), self._ws.data_ptr(),
                                               self._ws.numel(), _stream(self.device))
            if rc != _lib.DEX_OK:
                raise RuntimeError(f"dex_mel_spectrogram: {self._lib.dex_mel_last_error(self._h).decode()} ({rc})")
            return mel, energy

    def mel_spectrogram(self, y: torch.Tensor):
        """y: [B, T] in [-1, 1] -> (mel [B, 80, frames], energy [B, frames]) — audio/stft.py:159-178; the whole batch in one pass."""
        lo, hi = torch.aminmax(y)
        if float(lo) < -1 or float(hi) > 1:
            raise AssertionError("input must lie in [-1, 1]")            # stft.py:169-170 (the reference's asserts read the tensor too)
        return self._run(y)


def get_mel_from_wav(audio, _stft: TacotronSTFT):
    """audio/tools.py:8-15: clip to [-1,1], mel + energy as float32 numpy arrays."""
    wav = torch.as_tensor(np.asarray(audio), dtype=torch.float32).reshape(1, -1)      # the clip is the pad kernel's
    mel, energy = _stft._run(wav)
    return mel[0].cpu().numpy().astype(np.float32), energy[0].cpu().numpy().astype(np.float32)


def lf0_from_f0(f0: torch.Tensor, lengths: torch.Tensor = None) -> torch.Tensor:
    """DEX-TTS/synthesize.py:55-58 + normalize_lf0 (:26-38) on the device: f0 [B,T] (or [T]) in Hz, 0 = unvoiced — the output of
    the host's DIO/StoneMask — -> normalised log-f0 of the same shape, the ``lf0`` input of the style encoders.  ``lengths`` [B]:
    frames past an utterance's length are excluded from the statistics and come back 0."""
    lib = _lib.load()
    one = f0.dim() == 1
    if not f0.is_cuda:
        raise RuntimeError("lf0_from_f0 runs on an MI355X only (no CPU path)")
    with torch.cuda.device(f0.device):
        x = f0.to(torch.float32).reshape(1, -1).contiguous() if one else f0.to(torch.float32).contiguous()
        B, T = x.shape
        ln = None if lengths is None else lengths.to(device=x.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(x)
        rc = lib.dex_lf0_normalize(x.data_ptr(), None if ln is None else ln.data_ptr(), B, T, out.data_ptr(), _stream(x.device))
        if rc != _lib.DEX_OK:
            raise RuntimeError(f"dex_lf0_normalize failed ({rc})")
        return out[0] if one else out
