"""TEST INFRASTRUCTURE — CPU restatement of the reference's HiFi-GAN generator forward (GeDEX-TTS/hifigan/models.py:
ResBlock.forward :95-102, Generator.forward :150-167) with plain torch functional ops on a flat weight dict keyed like
``Generator.state_dict()`` after ``remove_weight_norm()``.  Pinned against the real reference module by
oracle/make_golden_vocoder.py (tests/golden/vocoder.npz).  Only tests/, smoke() and bench.py's cpu_baseline leg may use it."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1


def synth_weights(shapes: Dict[str, tuple], seed: int = 0) -> Dict[str, np.ndarray]:
    """Portable non-degenerate generator weights: dex_tts_amd.synth.make_vocoder_weights (shared data generator)."""
    from dex_tts_amd import synth
    return synth.make_vocoder_weights(shapes, seed)


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """models.py:16-17."""
    return int((kernel_size * dilation - dilation) / 2)


def resblock(W, p: str, x: torch.Tensor, k: int, dilations) -> torch.Tensor:
    """ResBlock.forward, models.py:95-102."""
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, W[f"{p}.convs1.{m}.weight"], W[f"{p}.convs1.{m}.bias"], dilation=d, padding=get_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, W[f"{p}.convs2.{m}.weight"], W[f"{p}.convs2.{m}.bias"], dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def generator(W, h, mel: torch.Tensor) -> torch.Tensor:
    """Generator.forward, models.py:150-167: mel [B,80,T] -> wav [B,1,T*prod(rates)]."""
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    rk, rd = list(h["resblock_kernel_sizes"]), list(h["resblock_dilation_sizes"])
    x = F.conv1d(mel, W["conv_pre.weight"], W["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, W[f"ups.{i}.weight"], W[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(len(rk)):
            r = resblock(W, f"resblocks.{i * len(rk) + j}", x, rk[j], rd[j])
            xs = r if xs is None else xs + r
        x = xs / len(rk)
    x = F.leaky_relu(x)                      # default slope 0.01 (models.py:165)
    x = F.conv1d(x, W["conv_post.weight"], W["conv_post.bias"], padding=3)
    return torch.tanh(x)
