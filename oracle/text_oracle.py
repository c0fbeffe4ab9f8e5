"""CHECKER INFRASTRUCTURE (tests / golden generation only; never imported by the product).

CPU restatement of the reference text path, plain torch fp32, no transformers / timm:

* ``text_encoder_forward``  TextEncoder.forward              GeDEX-TTS/model/text_encoder.py:129-146 (DEX :126-142)
    - ConvReluNorm prenet                                     text_encoder.py:34-67
    - RetNetModel, parallel form, use_softmax=True / use_decay=False, subln + GLU, pre-norm RMSNorm
                                                              retnet.py:56-178, retention.py:182-294 (MultiScaleRetention),
                                                              :357-390 (GLU), :446-501 (RetNetDecoderLayer), :67-166 (RetNetRelPos)
    - DEX: AdaptiveLayerNorm after each residual sum          DEX-TTS/model/retention.py:489-509, base.py:161-194
    - DurationPredictor                                       text_encoder.py:70-93
* ``align``                  durations -> lengths -> generate_path -> mu_y
                                                              GeDEX-TTS/model/tts.py:37-50, model/utils.py:26-39

Pinned against the imported reference modules by oracle/make_golden_text.py (tests/golden/text_*.npz)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def sequence_mask(length: Tensor, max_length=None) -> Tensor:                     # utils.py:6-10
    if max_length is None:
        max_length = length.max()
    x = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def fix_len_compatibility(length: int, n_down: int = 2) -> int:                    # utils.py:13-17
    while length % (2 ** n_down):
        length += 1
    return length


def channel_layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-4) -> Tensor:   # text_encoder.py:13-31, x [B,C,T]
    mean = torch.mean(x, 1, keepdim=True)
    var = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + eps)
    return x * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float = 1e-6) -> Tensor:    # retention.py:48-66
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return y if weight is None else y * weight


def rotate_every_two(x: Tensor) -> Tensor:                                         # retention.py:26-30
    x1, x2 = x[..., ::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def adaptive_layer_norm(W, p: str, x: Tensor, sty: Tensor, eps: float = 1e-5) -> Tensor:   # base.py:180-194, x [B,T,C], sty [B,C]
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mean) / (var + eps).sqrt()
    scale = F.linear(sty, W[p + ".W_scale.weight"], W[p + ".W_scale.bias"])
    bias = F.linear(sty, W[p + ".W_bias.weight"], W[p + ".W_bias.bias"])
    return y * scale.unsqueeze(1) + bias.unsqueeze(1)


def retnet_forward(W: Dict[str, Tensor], p: str, h: Tensor, mask: Tensor, n_layers: int, n_heads: int, sty: Optional[Tensor] = None) -> Tensor:
    """RetNetModel.forward(inputs_embeds=h [B,T,E], attention_mask=mask [B,1,T]) -> last_hidden_state, eval mode."""
    B, T, E = h.shape
    kd = E // n_heads
    angle = W[p + ".retnet_rel_pos.angle"]                                         # retention.py:76-77 (a registered buffer)
    index = torch.arange(T).to(angle)
    sin = torch.sin(index[:, None] * angle[None, :])                               # retention.py:137-138
    cos = torch.cos(index[:, None] * angle[None, :])
    dmask = mask.unsqueeze(2) * mask.unsqueeze(-1)                                 # :139  [B,1,T,T] (use_decay=False: no decay, not causal)
    for i in range(n_layers):
        q = f"{p}.layers.{i}"
        res = h
        a = rms_norm(h, W[q + ".retention_layer_norm.weight"])                     # :470-471 (normalize_before)
        qq = F.linear(a, W[q + ".retention.q_proj.weight"]); kk = F.linear(a, W[q + ".retention.k_proj.weight"])
        vv = F.linear(a, W[q + ".retention.v_proj.weight"]); gg = F.linear(a, W[q + ".retention.g_proj.weight"])
        qq, kk, vv = [t.view(B, T, n_heads, -1).transpose(1, 2) for t in (qq, kk, vv)]     # :21-23
        kk = kk * kd ** -0.5                                                        # :274
        qr = qq * cos + rotate_every_two(qq) * sin                                  # :277-278
        kr = kk * cos + rotate_every_two(kk) * sin
        ret = (qr @ kr.transpose(-1, -2)) * dmask                                   # :235-236
        ret = F.softmax(ret.masked_fill(dmask == 0, -1e4), dim=-1)                  # :238-240
        out = (ret @ vv).transpose(1, 2)                                            # :245-246  [B,T,h,hd]
        normed = rms_norm(out, None).reshape(B, T, E)                               # :285 (group_norm: RMSNorm per head, no affine)
        out = F.linear(F.silu(gg) * normed, W[q + ".retention.out_proj.weight"])    # :287-288
        h = res + out                                                               # :486 (alpha = 1, drop_path off)
        if sty is not None:
            h = adaptive_layer_norm(W, q + ".adaln_1", h, sty)
        res = h
        a = rms_norm(h, W[q + ".final_layer_norm.weight"])
        g = F.linear(a, W[q + ".ffn.gate.weight"])                                  # GLU :379-388
        a = F.gelu(F.linear(a, W[q + ".ffn.fc1.weight"])) * g
        h = res + F.linear(a, W[q + ".ffn.fc2.weight"])
        if sty is not None:
            h = adaptive_layer_norm(W, q + ".adaln_2", h, sty)
    return rms_norm(h, W[p + ".layer_norm.weight"])                                # retnet.py:166-167


def text_encoder_forward(W: Dict[str, Tensor], cfg: dict, x: Tensor, x_lengths: Tensor, spk: Optional[Tensor] = None,
                         sty: Optional[Tensor] = None):
    """-> mu [B,n_feats,T], logw [B,1,T], x_mask [B,1,T].  cfg: n_channels, n_layers, n_heads, n_spks (sty given = the DEX form)."""
    nc = cfg["n_channels"]
    h = F.embedding(x, W["emb.weight"]) * math.sqrt(nc)                             # text_encoder.py:130
    h = h.transpose(1, -1)
    x_mask = sequence_mask(x_lengths, h.size(2)).unsqueeze(1).to(h.dtype)
    org = h                                                                         # ConvReluNorm :59-66
    for i in range(3):
        h = F.conv1d(h * x_mask, W[f"prenet.conv_layers.{i}.weight"], W[f"prenet.conv_layers.{i}.bias"], padding=2)
        h = torch.relu(channel_layer_norm(h, W[f"prenet.norm_layers.{i}.gamma"], W[f"prenet.norm_layers.{i}.beta"]))
    h = (org + F.conv1d(h, W["prenet.proj.weight"], W["prenet.proj.bias"])) * x_mask
    if cfg.get("n_spks", 1) > 1:
        h = torch.cat([h, spk.unsqueeze(-1).repeat(1, 1, h.shape[-1])], dim=1)      # :137-138
    h = retnet_forward(W, "encoder", h.transpose(1, 2), x_mask, cfg["n_layers"], cfg["n_heads"], sty).transpose(1, 2) * x_mask
    mu = F.conv1d(h, W["proj_m.weight"], W["proj_m.bias"]) * x_mask
    d = F.conv1d(h * x_mask, W["proj_w.conv_1.weight"], W["proj_w.conv_1.bias"], padding=cfg.get("kernel_size", 3) // 2)      # DurationPredictor :81-93
    d = channel_layer_norm(torch.relu(d), W["proj_w.norm_1.gamma"], W["proj_w.norm_1.beta"])
    d = F.conv1d(d * x_mask, W["proj_w.conv_2.weight"], W["proj_w.conv_2.bias"], padding=cfg.get("kernel_size", 3) // 2)
    d = channel_layer_norm(torch.relu(d), W["proj_w.norm_2.gamma"], W["proj_w.norm_2.beta"])
    logw = F.conv1d(d * x_mask, W["proj_w.proj.weight"], W["proj_w.proj.bias"]) * x_mask
    return mu, logw, x_mask


def generate_path(duration: Tensor, mask: Tensor) -> Tensor:                        # utils.py:26-39
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1)
    path = sequence_mask(cum.view(b * t_x), t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path * mask


def align(mu_x: Tensor, logw: Tensor, x_mask: Tensor, length_scale: float = 1.0):
    """tts.py:37-50 -> dict(w_ceil [B,1,T], y_lengths [B], y_max_length, y_max_length_, y_mask [B,1,Ty_], attn [B,1,T,Ty_], mu_y [B,F,Ty_])."""
    w = torch.exp(logw) * x_mask
    w_ceil = torch.ceil(w) * length_scale
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_max = int(y_lengths.max())
    y_max_ = fix_len_compatibility(y_max)
    y_mask = sequence_mask(y_lengths, y_max_).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)
    attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
    mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
    return dict(w_ceil=w_ceil, y_lengths=y_lengths, y_max_length=y_max, y_max_length_=y_max_, y_mask=y_mask, attn=attn, mu_y=mu_y)
