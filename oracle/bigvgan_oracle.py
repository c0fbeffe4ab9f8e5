"""TEST INFRASTRUCTURE — CPU restatement of the reference's BigVGAN generator forward (DEX-TTS/bigvgan/models.py: AMPBlock1.forward
:76-85, BigVGAN.forward :186-211; anti-aliased activation alias_free_torch/act.py:23-28, resample.py:27-49, filter.py:30-94;
Snake / SnakeBeta activations.py:46-57,103-117) with plain torch functional ops on a flat weight dict keyed like
``BigVGAN.state_dict()`` after ``remove_weight_norm()``.  Pinned against the real reference module by
oracle/make_golden_bigvgan.py (tests/golden/bigvgan.npz).  Only tests/, smoke() and bench.py's cpu_baseline leg may use it."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """filter.py:30-60 -> [kernel_size]."""
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if kernel_size % 2 == 0 else torch.arange(kernel_size) - half_size
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return f / f.sum()


def upsample2(x: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    """UpSample1d(ratio 2, kernel 12).forward, resample.py:27-35; x [B,C,T], filt [K]."""
    K, ratio = filt.numel(), 2
    pad = K // ratio - 1
    pad_left = pad * ratio + (K - ratio) // 2
    pad_right = pad * ratio + (K - ratio + 1) // 2
    C = x.shape[1]
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.view(1, 1, K).expand(C, -1, -1), stride=ratio, groups=C)
    return x[..., pad_left:-pad_right]


def downsample2(x: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    """DownSample1d(ratio 2, kernel 12) = LowPassFilter1d(stride 2).forward, filter.py:84-94."""
    K = filt.numel()
    even = K % 2 == 0
    C = x.shape[1]
    x = F.pad(x, (K // 2 - int(even), K // 2), mode="replicate")
    return F.conv1d(x, filt.view(1, 1, K).expand(C, -1, -1), stride=2, groups=C)


def snake(W, p: str, x: torch.Tensor, kind: str, logscale: bool) -> torch.Tensor:
    """Snake / SnakeBeta.forward (activations.py:46-57 / :103-117): x + 1/(b + 1e-9) * sin^2(a x)."""
    a = W[p + ".alpha"].view(1, -1, 1)
    b = W[p + ".beta"].view(1, -1, 1) if kind == "snakebeta" else a
    if logscale:
        a, b = torch.exp(a), torch.exp(b)
    return x + (1.0 / (b + 0.000000001)) * torch.pow(torch.sin(x * a), 2)


def activation1d(W, p: str, x: torch.Tensor, kind: str, logscale: bool) -> torch.Tensor:
    """Activation1d.forward, act.py:23-28: upsample x2 -> activation -> low-pass + downsample x2."""
    x = upsample2(x, W[p + ".upsample.filter"].flatten())
    x = snake(W, p + ".act", x, kind, logscale)
    return downsample2(x, W[p + ".downsample.lowpass.filter"].flatten())


def get_padding(k: int, d: int = 1) -> int:
    return int((k * d - d) / 2)


def ampblock1(W, p: str, x: torch.Tensor, k: int, dilations, kind: str, logscale: bool) -> torch.Tensor:
    """AMPBlock1.forward, models.py:76-85."""
    for m, d in enumerate(dilations):
        xt = activation1d(W, f"{p}.activations.{2 * m}", x, kind, logscale)
        xt = F.conv1d(xt, W[f"{p}.convs1.{m}.weight"], W[f"{p}.convs1.{m}.bias"], dilation=d, padding=get_padding(k, d))
        xt = activation1d(W, f"{p}.activations.{2 * m + 1}", xt, kind, logscale)
        xt = F.conv1d(xt, W[f"{p}.convs2.{m}.weight"], W[f"{p}.convs2.{m}.bias"], dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def generator(W, h, mel: torch.Tensor) -> torch.Tensor:
    """BigVGAN.forward, models.py:186-211: mel [B,80,T] -> wav [B,1,T*prod(rates)]."""
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    rk, rd = list(h["resblock_kernel_sizes"]), list(h["resblock_dilation_sizes"])
    kind, logscale = h["activation"], bool(h["snake_logscale"])
    x = F.conv1d(mel, W["conv_pre.weight"], W["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.conv_transpose1d(x, W[f"ups.{i}.0.weight"], W[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)     # no activation in front
        xs = None
        for j in range(len(rk)):
            r = ampblock1(W, f"resblocks.{i * len(rk) + j}", x, rk[j], rd[j], kind, logscale)
            xs = r if xs is None else xs + r
        x = xs / len(rk)
    x = activation1d(W, "activation_post", x, kind, logscale)
    x = F.conv1d(x, W["conv_post.weight"], W["conv_post.bias"], padding=3)
    return torch.tanh(x)
