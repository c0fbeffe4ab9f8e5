"""TEST INFRASTRUCTURE — CPU restatement of the DEX style encoders and the part of ``DeXTTS.forward`` that feeds the
decoder (DEX-TTS/model/ref_encoder.py: Projection :8-34, LF0Encoder :36-55, TIVEncoderBlock/TVEncoderBlock :57-81,
TIVEncoder :83-108, TVEncoder :110-140, VQEmbeddingEMA.forward :199-237; model/base.py: BasicConv :32-64, InstanceNorm1D
:66-88, LayerNorm :139-158; model/tts.py:55-66) with plain torch ops on a flat weight dict keyed like the reference
state dict (``tv_encoder.* / lf0_encoder.* / tiv_encoder.* / conv_sty.*``).  Eval mode: dropout off, BatchNorm on running
statistics, the VQ codebook frozen.  Pinned against the real reference modules by oracle/make_golden_style.py
(tests/golden/style.npz).  Only tests/ may use it."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor


def sequence_mask(length: Tensor, max_length: int) -> Tensor:
    x = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def basic_conv(W, p: str, x: Tensor, relu: bool, norm: str) -> Tensor:
    """BasicConv.forward (base.py:54-64): conv (no bias) -> bn -> relu -> ln (over channels)."""
    x = F.conv1d(x, W[f"{p}.conv.weight"], None, padding=1)
    if norm == "bn":
        x = F.batch_norm(x, W[f"{p}.bn.running_mean"], W[f"{p}.bn.running_var"], W[f"{p}.bn.weight"], W[f"{p}.bn.bias"], False, 0.01, 1e-5)
    if relu:
        x = torch.relu(x)
    if norm == "ln":
        x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), W[f"{p}.ln.weight"], W[f"{p}.ln.bias"], 1e-5).transpose(1, 2)
    return x


def channel_layer_norm(W, p: str, x: Tensor) -> Tensor:
    """base.LayerNorm (base.py:139-158): over the channel dim of [B,C,T], eps 1e-4, gamma / beta."""
    mean = torch.mean(x, 1, keepdim=True)
    variance = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(variance + 1e-4)
    return x * W[f"{p}.gamma"].view(1, -1, 1) + W[f"{p}.beta"].view(1, -1, 1)


def projection(W, p: str, x: Tensor, mask: Tensor) -> Tensor:
    """Projection.forward (ref_encoder.py:24-34), dropout off."""
    x = F.conv1d(x * mask, W[f"{p}.conv_1.weight"], W[f"{p}.conv_1.bias"], padding=1)
    x = channel_layer_norm(W, f"{p}.norm_1", torch.relu(x))
    x = F.conv1d(x * mask, W[f"{p}.conv_2.weight"], W[f"{p}.conv_2.bias"], padding=1)
    x = channel_layer_norm(W, f"{p}.norm_2", torch.relu(x))
    x = F.conv1d(x * mask, W[f"{p}.proj.weight"], W[f"{p}.proj.bias"])
    return x * mask


def lf0_encoder(W, lf0: Tensor, mask: Tensor, c_h: int = 192, num_layer: int = 2) -> Tuple[Tensor, Tensor]:
    """LF0Encoder.forward (ref_encoder.py:45-55)."""
    p = "lf0_encoder"
    x = basic_conv(W, f"{p}.in_conv", lf0.unsqueeze(1) * mask, True, "ln") * mask
    rnn = nn.GRU(c_h, c_h // 2, num_layer, batch_first=True, bidirectional=True)
    rnn.load_state_dict({k[len(f"{p}.rnn_layer."):]: v for k, v in W.items() if k.startswith(f"{p}.rnn_layer.")})
    with torch.no_grad():
        y, _ = rnn(x.transpose(1, 2))
    x = basic_conv(W, f"{p}.out_conv", y.transpose(1, 2) * mask, True, "ln") * mask
    return x, projection(W, f"{p}.proj", x, mask)


def inorm1d(x: Tensor) -> Tensor:
    """InstanceNorm1D.forward (base.py:72-88): over the FULL padded length, unbiased variance."""
    mean = x.mean(-1).unsqueeze(-1)
    std = (x.var(-1) + 1e-5).sqrt().unsqueeze(-1)
    return (x - mean) / std


def tiv_encoder(W, x: Tensor, mask: Tensor, num_layer: int = 6) -> Tuple[Tensor, List[Tensor]]:
    """TIVEncoder.forward (ref_encoder.py:96-108)."""
    p = "tiv_encoder"
    x = basic_conv(W, f"{p}.in_conv", x * mask, True, "bn") * mask
    skips = []
    for i in range(num_layer):
        xin = x * mask
        b = f"{p}.conv_blocks.{i}.conv_block"
        x = (xin + basic_conv(W, f"{b}.1", basic_conv(W, f"{b}.0", xin, True, "bn"), False, "")) * mask
        skips.append(x)
        x = inorm1d(x)
    x = basic_conv(W, f"{p}.out_conv", x * mask, True, "bn") * mask
    return x, skips


def vq_eval(W, x: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """VQEmbeddingEMA.forward in eval mode (ref_encoder.py:199-237): x [B,T,D], mask [B,1,T] -> quantized [B,T,D] * mask
    (straight-through value == the codebook row), and the chosen indices."""
    emb = W["tv_encoder.vq.embedding"]
    m = mask.transpose(1, 2)
    x = x * m
    M, D = emb.shape
    x_flat = x.reshape(-1, D)
    distances = torch.addmm(torch.sum(emb ** 2, dim=1) + torch.sum(x_flat ** 2, dim=1, keepdim=True), x_flat, emb.t(), alpha=-2.0, beta=1.0)
    idx = torch.argmin(distances.float(), dim=-1)
    q = F.embedding(idx, emb).view_as(x)
    q = x + (q - x)
    return q * m, idx.view(x.shape[0], x.shape[1])


def tv_encoder(W, x: Tensor, mask: Tensor, num_layer: int = 6) -> Tuple[Tensor, Tensor, Tensor]:
    """TVEncoder.forward (ref_encoder.py:122-140): returns (z_beforeVQ, z_dec, vq indices)."""
    p = "tv_encoder"
    x = basic_conv(W, f"{p}.in_conv", x * mask, True, "ln") * mask
    for i in range(num_layer):
        xin = x * mask
        b = f"{p}.conv_blocks.{i}.conv_block"
        x = (xin + basic_conv(W, f"{b}.1", basic_conv(W, f"{b}.0", xin, True, "ln"), False, "")) * mask
    z_before = basic_conv(W, f"{p}.out_conv", x * mask, False, "") * mask
    z, idx = vq_eval(W, z_before.transpose(1, 2), mask)
    z_dec = projection(W, f"{p}.proj_0", z.transpose(1, 2), mask)
    z_dec = basic_conv(W, f"{p}.proj_1", z_dec * mask, True, "bn") * mask
    return z_before, z_dec, idx


def style_forward(W: Dict[str, Tensor], ref: Tensor, ref_lengths: Tensor, sty: Tensor, sty_lengths: Tensor, lf0: Tensor,
                  lf0_lengths: Tensor) -> Dict[str, Tensor]:
    """DeXTTS.forward up to the text encoder (tts.py:55-66): masks, LF0 / TV / TIV encoders, the pooled style vector for the
    text encoder (sty_enc), the decoder's style tokens after conv_sty (sty_dec) and the TIV skips."""
    ref_mask = sequence_mask(ref_lengths, ref.size(2)).unsqueeze(1).to(ref.dtype)
    lf0_mask = sequence_mask(lf0_lengths, lf0.size(1)).unsqueeze(1).to(lf0.dtype)
    sty_mask = sequence_mask(sty_lengths, sty.size(2)).unsqueeze(1).to(sty.dtype)
    lf0_enc, lf0_dec = lf0_encoder(W, lf0, lf0_mask)
    sty_enc, sty_dec, idx = tv_encoder(W, sty, sty_mask)
    sty_enc = (sty_enc.sum(dim=-1) / sty_mask.sum(dim=-1)) + (lf0_enc.sum(dim=-1) / lf0_mask.sum(dim=-1))
    sty_dec = sty_dec + (lf0_dec.sum(dim=-1) / lf0_mask.sum(dim=-1)).unsqueeze(-1)
    sty_dec = F.conv1d(sty_dec, W["conv_sty.weight"], W["conv_sty.bias"])
    _, skips = tiv_encoder(W, ref, ref_mask)
    return {"sty_enc": sty_enc, "sty_dec": sty_dec, "ref_skips": skips, "vq_idx": idx}


def normalize_lf0(lf0):
    """normalize_lf0 — DEX-TTS/synthesize.py:26-38 — on a 1-D float32 array of log-f0 (0 = unvoiced): over the entries != 0
    (lf0 - mean) / (std + 1e-8), or lf0 - mean when std == 0; the others stay 0.  numpy float32 arithmetic, as in the reference."""
    import numpy as np
    lf0 = np.asarray(lf0, dtype=np.float32).copy()
    zero = lf0 == 0
    if (~zero).any():
        mean, std = np.mean(lf0[~zero]), np.std(lf0[~zero])
        lf0 = (lf0 - mean) if std == 0 else (lf0 - mean) / (std + 1e-8)
        lf0[zero] = 0.0
    return lf0.astype(np.float32)


def lf0_from_f0(f0):
    """DEX-TTS/synthesize.py:55-58: log of the voiced frames of an f0 track in Hz (1-D float32), then normalize_lf0."""
    import numpy as np
    f0 = np.asarray(f0, dtype=np.float32)
    lf0 = f0.copy()
    nz = np.nonzero(f0)
    lf0[nz] = np.log(f0[nz])
    return normalize_lf0(lf0)
