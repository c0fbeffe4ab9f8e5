"""ORACLE — test infrastructure only.  Never imported by the product path (dex_tts_amd/*).

CPU restatement (torch tensor ops on the host, fp32 or fp64) of the reference's reverse-diffusion
hot path: ``Diffusion.forward(infer=True)`` -> ``ablation_sampler`` (euler/edm/linear/none) ->
``EDMPrecond`` -> ``DiffusionDenoiser`` (+ ``DiTMask``, DEX ``TVAdaptor``/``TIVAdaptor``).
Every function cites the reference file:line it follows.  Weights are a flat dict keyed like the
reference state-dict *relative to* ``denoise_fn.`` (see dex_tts_amd/config.py:param_shapes).

Pinning: this restatement is checked against golden vectors produced by importing the real
reference in the build container (oracle/make_golden.py -> tests/golden/*.npz,
tests/test_oracle_golden.py).  The timm ``Attention``/``Mlp`` arithmetic is third-party
(timm, unpinned, not in /root/reference); it is restated from its published definition:
qkv = Linear(x) -> (B,N,3,H,hd); softmax(q * hd**-0.5 @ k^T) @ v; proj; Mlp = fc2(GELU_erf(fc1)).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def as_torch(weights: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}


# ---------------------------------------------------------------------------------------------
# element-wise pieces
def mish(x: Tensor) -> Tensor:
    """x * tanh(softplus(x)), torch softplus threshold 20 — GeDEX-TTS/model/diffusion.py:8-10."""
    return x * torch.tanh(F.softplus(x))


def sinusoid_unet(t: Tensor, dim: int, scale: float) -> Tensor:
    """SinusoidalPosEmb — diffusion.py:105-117: scale*t*exp(-k ln(1e4)/(half-1)), cat(sin, cos)."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32).to(t.dtype)
    freqs = torch.exp(k * -(math.log(10000.0) / (half - 1)))
    arg = scale * t[:, None] * freqs[None, :]
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


def sinusoid_dit(t: Tensor, dim: int = 256) -> Tensor:
    """TimestepEmbedder.timestep_embedding — dit.py:239-257: t*exp(-ln(1e4) k/half), cat(cos, sin)."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32).to(t.dtype)
    freqs = torch.exp(-math.log(10000.0) * k / half)
    arg = t[:, None] * freqs[None, :]
    return torch.cat([arg.cos(), arg.sin()], dim=-1)


def linear(W, name: str, x: Tensor, bias: bool = True) -> Tensor:
    return F.linear(x, W[f"{name}.weight"], W[f"{name}.bias"] if bias else None)


# ---------------------------------------------------------------------------------------------
# U-Net stages (diffusion.py)
def block(W, p: str, x: Tensor, mask: Tensor, groups: int) -> Tensor:
    """Block.forward — diffusion.py:41-50: mask * Mish(GroupNorm_8(Conv3x3(x*mask)))."""
    y = F.conv2d(x * mask, W[f"{p}.block.0.weight"], W[f"{p}.block.0.bias"], padding=1)
    y = F.group_norm(y, groups, W[f"{p}.block.1.weight"], W[f"{p}.block.1.bias"], eps=1e-5)
    return mish(y) * mask


def resnet_block(W, p: str, x: Tensor, mask: Tensor, temb: Tensor, groups: int, taps: Optional[dict] = None) -> Tensor:
    """ResnetBlock.forward — diffusion.py:66-71 (time bias added after the mask, result unmasked)."""
    h = block(W, f"{p}.block1", x, mask, groups)
    if taps is not None:
        taps[f"{p}.block1"] = h               # module-level checkpoints (tests/golden/modules_*.npz): Block (a6)
    h = h + linear(W, f"{p}.mlp.1", mish(temb))[:, :, None, None]
    h = block(W, f"{p}.block2", h, mask, groups)
    if f"{p}.res_conv.weight" in W:
        r = F.conv2d(x * mask, W[f"{p}.res_conv.weight"], W[f"{p}.res_conv.bias"])
    else:
        r = x * mask
    return h + r


def linear_attention(W, p: str, x: Tensor, heads: int, dim_head: int) -> Tensor:
    """Residual(Rezero(LinearAttention)) — diffusion.py:74-102,31-38: softmax over positions on k only."""
    b, c, hh, ww = x.shape
    qkv = F.conv2d(x, W[f"{p}.fn.fn.to_qkv.weight"])
    qkv = qkv.reshape(b, 3, heads, dim_head, hh * ww)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, W[f"{p}.fn.fn.to_out.weight"], W[f"{p}.fn.fn.to_out.bias"])
    return out * W[f"{p}.fn.g"] + x


# ---------------------------------------------------------------------------------------------
# DiT bottleneck (dit.py)
def layer_norm_noaffine(x: Tensor) -> Tensor:
    """nn.LayerNorm(hidden, elementwise_affine=False, eps=1e-6) — dit.py:275,277,321."""
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """dit.py:78-79."""
    return x * (1 + scale[:, None, :]) + shift[:, None, :]


def mha(W, p: str, x: Tensor, heads: int) -> Tensor:
    """timm Attention(dim, num_heads, qkv_bias=True) as used at dit.py:276 (restated; see header)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = linear(W, f"{p}.qkv", x).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    y = (att @ v).transpose(1, 2).reshape(B, N, C)
    return linear(W, f"{p}.proj", y)


def dit_block(W, p: str, x: Tensor, c: Tensor, heads: int) -> Tensor:
    """DiTBlock.forward — dit.py:286-290; chunk order shift/scale/gate (msa) then (mlp)."""
    mod = linear(W, f"{p}.adaLN_modulation.1", F.silu(c))
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
    x = x + g1[:, None, :] * mha(W, f"{p}.attn", modulate(layer_norm_noaffine(x), sh1, sc1), heads)
    h = linear(W, f"{p}.mlp.fc1", modulate(layer_norm_noaffine(x), sh2, sc2))
    h = linear(W, f"{p}.mlp.fc2", F.gelu(h))
    return x + g2[:, None, :] * h


def dit_patchify(W, cfg, x: Tensor, taps: Optional[dict] = None):
    """DiTMask.patchify — dit.py:440-456 with PatchEmbed2D :57-59, make_conv_pos :81-96, SamePad :128-139."""
    t = cfg.dit
    w = x.shape[-1]
    if w % t.patch_size != 0:
        x = F.pad(x, (0, t.patch_size - w % t.patch_size))
    e = F.conv2d(x, W["vit.x_embedder.proj.0.weight"], W["vit.x_embedder.proj.0.bias"],
                 stride=t.stride_size, padding=t.patch_size // 2, groups=x.shape[1])
    e = F.conv2d(F.silu(e), W["vit.x_embedder.proj.2.weight"], W["vit.x_embedder.proj.2.bias"])
    if taps is not None:
        taps["vit.x_embedder"] = e            # PatchEmbed2D (a10)
    pos = F.conv2d(e, W["vit.pos_conv.0.weight"], W["vit.pos_conv.0.bias"],
                   padding=t.conv_pos // 2, groups=t.conv_pos_groups)
    if t.conv_pos % 2 == 0:
        pos = pos[:, :, :-1, :-1]
    pos = F.gelu(pos).mean(dim=2, keepdim=True)
    e = e + pos[:, :, :, : e.shape[-1]] + W["vit.freq_new_pos_embed"]
    hh, ww = e.shape[2], e.shape[3]
    return e.flatten(2).transpose(1, 2), w, hh, ww


def dit_forward(W, cfg, x: Tensor, mask_mid: Tensor, t: Tensor, taps: Optional[dict] = None) -> Tensor:
    """DiTMask.forward, eval / mask_ratio=0 / use_decoder=False path — dit.py:485-525."""
    td = cfg.dit
    tok, w_orig, hh, ww = dit_patchify(W, cfg, x, taps)
    c = linear(W, "vit.t_embedder.mlp.2", F.silu(linear(W, "vit.t_embedder.mlp.0", sinusoid_dit(t, 256))))
    if taps is not None:
        taps["tok_in"] = tok
        taps["vit.t_embedder"] = c            # TimestepEmbedder (a11)
    for k in range(td.depth):
        tok = dit_block(W, f"vit.blocks.{k}", tok, c, td.num_heads)
        if taps is not None:
            taps[f"tok_blk{k}"] = tok
    mod = linear(W, "vit.final_layer.adaLN_modulation.1", F.silu(c))
    shift, scale = mod.chunk(2, dim=1)
    tok = linear(W, "vit.final_layer.linear", modulate(layer_norm_noaffine(tok), shift, scale))
    if taps is not None:
        taps["vit.final_layer"] = tok         # FinalLayer before unpatchify (a13)
    # unpatchify 'B (h w) (p1 p2 C) -> B C (h p1) (w p2)' — dit.py:458-463
    B, N, _ = tok.shape
    s, C = td.stride_size, x.shape[1]
    y = tok.reshape(B, hh, ww, s, s, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, hh * s, ww * s)
    return y[..., :w_orig] * mask_mid


# ---------------------------------------------------------------------------------------------
# DEX style adaptors (DEX-TTS/model/ref_encoder.py, base.py)
def instance_norm2d(x: Tensor, eps: float = 1e-5) -> Tensor:
    """InstanceNorm2D — DEX-TTS/model/base.py:90-114: unbiased var, stats incl. padding."""
    B, C = x.shape[:2]
    flat = x.reshape(B, C, -1)
    mean = flat.mean(dim=2).reshape(B, C, 1, 1)
    std = (flat.var(dim=2) + eps).sqrt().reshape(B, C, 1, 1)
    return (x - mean) / std


def stack_stats(ref_skips: List[Tensor], eps: float = 1e-5):
    """DiffusionDenoiser._stack_stats + InstanceNorm1D.cal_stats — DEX diffusion.py:177-188,
    base.py:72-78: per-skip mean and sqrt(unbiased var + eps) over the full padded length."""
    means = torch.stack([r.mean(-1) for r in ref_skips], dim=1)                 # [B, L, C]
    stds = torch.stack([(r.var(-1) + eps).sqrt() for r in ref_skips], dim=1)
    return means, stds


def tv_adaptor(W, x: Tensor, x_mask: Tensor, sty: Tensor, sty_lengths: Tensor, t_sty: Tensor) -> Tensor:
    """TVAdaptor.forward — ref_encoder.py:154-179.  ``t_sty`` [1|B, C] is the time token (key 0,
    always valid); padded style keys are filled with -1e4 (not -inf)."""
    B, C, H, Wd = x.shape
    Ts = sty.shape[-1]
    keys = torch.cat([t_sty.expand(B, C)[:, :, None], sty], dim=-1).transpose(1, 2)        # [B, Ts+1, C]
    valid = torch.arange(Ts)[None, :] < sty_lengths[:, None]
    valid = torch.cat([torch.ones(B, 1, dtype=torch.bool), valid], dim=1)                  # [B, Ts+1]
    q = F.linear(instance_norm2d(x).permute(0, 2, 3, 1), W["tv_adaptor.w_q.weight"])       # [B,H,W,C]
    k = F.linear(keys, W["tv_adaptor.w_k.weight"])
    v = F.linear(keys, W["tv_adaptor.w_v.weight"])
    att = torch.einsum("bhwc,btc->bhwt", q / (C ** 0.5), k)
    att = att.masked_fill(~valid[:, None, None, :], -1e4).softmax(dim=-1)
    out = F.linear(torch.einsum("bhwt,btc->bhwc", att, v), W["tv_adaptor.linear.weight"])
    return (x + out.permute(0, 3, 1, 2)) * x_mask


def sap(W, p: str, stats: Tensor, t_tok: Tensor) -> Tensor:
    """SelfAttentionPooling.forward — ref_encoder.py:246-253 (time token prepended)."""
    B = stats.shape[0]
    xs = torch.cat([t_tok.expand(B, -1)[:, None, :], stats], dim=1)                         # [B, L+1, C]
    a = F.linear(xs, W[f"{p}.W.weight"], W[f"{p}.W.bias"]).squeeze(-1).softmax(dim=-1)
    return (xs * a[:, :, None]).sum(dim=1)


def tiv_adaptor(W, x: Tensor, ref_mean: Tensor, ref_std: Tensor, t_adap: Tensor) -> Tensor:
    """TIVAdaptor.forward — ref_encoder.py:264-273: AdaIN with SAP-pooled stats; output NOT masked."""
    m = sap(W, "tiv_adaptor.mean_sap", ref_mean, t_adap)[:, :, None, None]
    s = sap(W, "tiv_adaptor.std_sap", ref_std, t_adap)[:, :, None, None]
    return instance_norm2d(x) * s + m


# ---------------------------------------------------------------------------------------------
def denoiser_forward(W, cfg, x: Tensor, mask: Tensor, mu: Tensor, t: Tensor, spk: Optional[Tensor] = None,
                     ref: Optional[List[Tensor]] = None, sty: Optional[Tensor] = None,
                     sty_lengths: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """DiffusionDenoiser.forward — GeDEX diffusion.py:168-207 / DEX diffusion.py:190-236.

    x, mu: [B,80,T]; mask: [B,1,T]; t: [1] (sampler: one time-conditioning vector for the whole
    batch, edm.py:94-96) or [B].  ``taps`` optionally collects stage checkpoints for tests."""
    g, d = cfg.groups, cfg.dim
    planes = [mu, x]
    if cfg.n_spks > 1:
        s = linear(W, "spk_mlp.2", mish(linear(W, "spk_mlp.0", spk)))
        planes.append(s[:, :, None].expand(-1, -1, x.shape[-1]))
    h = torch.stack(planes, dim=1)
    t_init = sinusoid_unet(t, d, cfg.pe_scale)
    temb = linear(W, "mlp.2", mish(linear(W, "mlp.0", t_init)))
    if taps is not None:
        taps["mlp"] = temb                    # sinusoid + time MLP (a5)
    if cfg.variant == "dex":
        t_adap = linear(W, "mlp_adap.2", mish(linear(W, "mlp_adap.0", t_init)))            # [1|B, 2d]
        t_sty = linear(W, "mlp_adap_sty.2", mish(linear(W, "mlp_adap_sty.0", t_init)))
        ref_mean, ref_std = stack_stats(ref)
    m = mask[:, None]                                                                      # [B,1,1,T]
    masks, hiddens = [m], []
    n_stage = len(cfg.dim_mults)
    for i in range(n_stage):
        md = masks[-1]
        h = resnet_block(W, f"downs.{i}.0", h, md, temb, g, taps)
        if taps is not None:
            taps[f"downs.{i}.0"] = h          # ResnetBlock (a7)
        h = resnet_block(W, f"downs.{i}.1", h, md, temb, g, taps)
        if taps is not None:
            taps[f"downs.{i}.1"] = h
        h = linear_attention(W, f"downs.{i}.2", h, cfg.lin_heads, cfg.lin_dim_head)
        hiddens.append(h)
        if taps is not None:
            taps[f"down{i}"] = h
        h = h * md
        if i < n_stage - 1:
            h = F.conv2d(h, W[f"downs.{i}.3.conv.weight"], W[f"downs.{i}.3.conv.bias"], stride=2, padding=1)
            if taps is not None:
                taps[f"downs.{i}.3"] = h      # Downsample (a9)
        masks.append(md[:, :, :, ::2])
    masks = masks[:-1]
    mm = masks[-1]
    if cfg.variant == "dex":
        h = tv_adaptor(W, h, mm, sty, sty_lengths, t_sty)
        if taps is not None:
            taps["tv"] = h
        h = tiv_adaptor(W, h, ref_mean, ref_std, t_adap)
        if taps is not None:
            taps["tiv"] = h
    if taps is not None:
        taps["dit_in"] = h
    h = dit_forward(W, cfg, h, mm, t, taps)
    if taps is not None:
        taps["dit_out"] = h
    for j in range(n_stage - 1):
        mu_ = masks.pop()
        h = torch.cat([h, hiddens.pop()], dim=1)
        h = resnet_block(W, f"ups.{j}.0", h, mu_, temb, g, taps)
        if taps is not None:
            taps[f"ups.{j}.0"] = h
        h = resnet_block(W, f"ups.{j}.1", h, mu_, temb, g, taps)
        if taps is not None:
            taps[f"ups.{j}.1"] = h
        h = linear_attention(W, f"ups.{j}.2", h, cfg.lin_heads, cfg.lin_dim_head)
        if taps is not None:
            taps[f"up{j}"] = h
        h = F.conv_transpose2d(h * mu_, W[f"ups.{j}.3.conv.weight"], W[f"ups.{j}.3.conv.bias"],
                               stride=2, padding=1)
    if taps is not None:
        taps["up_out"] = h
    h = block(W, "final_block", h, m, g)
    out = F.conv2d(h * m, W["final_conv.weight"], W["final_conv.bias"])
    if taps is not None:
        taps["final_block"] = h               # final Block + 1x1 conv (a14)
        taps["final_conv"] = out
    return (out * m).squeeze(1)


# ---------------------------------------------------------------------------------------------
# EDM preconditioning + Euler sampler (edm.py)
SIGMA_DATA, SIGMA_MIN, SIGMA_MAX, RHO = 0.5, 0.002, 80.0, 7


def edm_sigmas(n: int, dtype=torch.float32) -> Tensor:
    """EDM discretisation — edm.py:141,157,184-185 (computed in fp32 like the reference), plus t_N = 0.

    NB the reference evaluates ``step_indices / (num_steps - 1)`` with an int64 arange, i.e. in
    fp32 tensor arithmetic; num_steps == 1 yields 0/0 = nan there and is rejected here."""
    if n < 2:
        raise ValueError("n_timesteps must be >= 2 (reference divides by num_steps - 1)")
    idx = torch.arange(n)
    s = (SIGMA_MAX ** (1 / RHO) + idx / (n - 1) * (SIGMA_MIN ** (1 / RHO) - SIGMA_MAX ** (1 / RHO))) ** RHO
    s = s.to(torch.float32)
    return torch.cat([s, torch.zeros(1)]).to(dtype)


def edm_precond(W, cfg, x: Tensor, sigma: Tensor, mask: Tensor, mu: Tensor, **kw) -> Tensor:
    """EDMPrecond.forward — edm.py:88-98; sigma 0-dim -> c_noise.flatten() has shape (1,)."""
    sigma = sigma.reshape(-1, 1, 1)
    c_skip = SIGMA_DATA ** 2 / (sigma ** 2 + SIGMA_DATA ** 2)
    c_out = sigma * SIGMA_DATA / (sigma ** 2 + SIGMA_DATA ** 2).sqrt()
    c_in = 1 / (SIGMA_DATA ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    f = denoiser_forward(W, cfg, c_in * x, mask, mu, c_noise.flatten(), **kw)
    return c_skip * x + c_out * f


def edm_loss_weight(sigma: Tensor, loss_type: str = "base") -> Tensor:
    """The per-utterance weight lambda(sigma) of EDMLoss.forward — edm.py:37-63, branch by branch (note the reference's
    if / if / if-elif chain: 'base_min_*' computes its weight in the second ``if`` and then falls through the third chain
    untouched, which is what happens here too)."""
    snr = 1 / sigma ** 2
    base = (sigma ** 2 + SIGMA_DATA ** 2) / (sigma * SIGMA_DATA) ** 2
    if loss_type == "base":
        return base
    if loss_type.startswith("base_min_"):
        k = float(loss_type.split("base_min_")[-1])
        return torch.minimum(base, k * torch.ones_like(sigma))
    if loss_type.startswith("base_log_"):
        k = float(loss_type.split("base_log_")[-1])
        w = base.clone()
        hit = w >= k
        w[hit] = torch.log(w[hit]) + (k - np.log(k))
        return w
    if loss_type.startswith("min_snr_"):
        return torch.minimum(snr, float(loss_type.split("min_snr_")[-1]) * torch.ones_like(sigma))
    if loss_type.startswith("max_snr_"):
        return torch.maximum(snr, float(loss_type.split("max_snr_")[-1]) * torch.ones_like(sigma))
    if loss_type == "snr":
        return snr
    if loss_type == "inv_snr":
        return 1.0 / snr
    raise ValueError(f"loss_type {loss_type!r}: the reference would leave `weight` undefined (edm.py:37-63)")


def edm_loss(W, cfg, x0: Tensor, mask: Tensor, mu: Tensor, rnd_normal: Tensor, eps: Tensor, loss_type: str = "base",
             P_mean: float = -1.2, P_std: float = 1.2, n_feats: int = 80, **kw) -> Tensor:
    """EDMLoss.forward — edm.py:31-68 — with its two draws as inputs (``rnd_normal`` = randn([B,1,1]), ``eps`` = randn_like(x0)):
    sigma = exp(rnd P_std + P_mean) per utterance; n = (eps + mu) sigma; D = EDMPrecond(x0 + n, sigma);
    loss = sum(weight (D - x0)^2) / sum(mask n_feats).  The denoiser is evaluated utterance by utterance (every normalisation
    and attention of the net is per utterance, and the DEX adaptors of the reference only broadcast a single t)."""
    sigma = (rnd_normal * P_std + P_mean).exp()
    weight = edm_loss_weight(sigma, loss_type)
    n = (eps + mu) * sigma
    B = x0.shape[0]
    d = []
    for b in range(B):
        kb = {}
        for k, v in kw.items():
            kb[k] = [r[b:b + 1] for r in v] if isinstance(v, (list, tuple)) else v[b:b + 1]
        d.append(edm_precond(W, cfg, (x0 + n)[b:b + 1], sigma[b].reshape(()), mask[b:b + 1], mu[b:b + 1], **kb))
    D = torch.cat(d, 0)
    return torch.sum(weight * ((D - x0) ** 2)) / torch.sum(mask * n_feats)


def churn_step(x: Tensor, t_cur: Tensor, n_steps: int, noise_i: Optional[Tensor], S_churn: float, S_min: float, S_max: float,
               S_noise: float):
    """"Increase noise temporarily" — edm.py:194-196 with schedule='linear', scaling='none' (sigma(t) = t, s(t) = 1):
    gamma = min(S_churn/num_steps, sqrt(2)-1) if S_min <= t_cur <= S_max else 0;  t_hat = t_cur + gamma t_cur;
    x_hat = x_cur + sqrt(clip(t_hat^2 - t_cur^2, 0)) * S_noise * randn_like(x_cur).  The noise draw is an input here
    (ablation_sampler's ``randn_like`` argument).  Returns (x_hat, t_hat)."""
    gamma = min(S_churn / n_steps, np.sqrt(2) - 1) if S_min <= float(t_cur) <= S_max else 0
    t_hat = torch.as_tensor(t_cur + gamma * t_cur)
    if noise_i is None:
        return x, t_hat
    x_hat = x + (t_hat ** 2 - t_cur ** 2).clip(min=0).sqrt() * S_noise * noise_i
    return x_hat, t_hat


def edm_euler_sampler(W, cfg, z: Tensor, mask: Tensor, mu: Tensor, n_steps: int, trace: Optional[list] = None,
                      noise: Optional[Tensor] = None, S_churn: float = 0, S_min: float = 0, S_max: float = float("inf"),
                      S_noise: float = 1, **kw) -> Tensor:
    """ablation_sampler(solver='euler', discretization='edm', schedule='linear', scaling='none')
    — edm.py:109-216: x0 = z*sigma_0; per step (x_hat, t_hat) = churn(x, t) (edm.py:194-196), d = (x_hat - D(x_hat, t_hat))/t_hat,
    x = x_hat + (t_next - t_hat) d.  With S_churn = 0 (what Diffusion wires) the per-step ``0 * randn_like`` contributes
    exactly zero and ``noise`` may be None; otherwise noise[i] is the draw of step i."""
    ts = edm_sigmas(n_steps, z.dtype)
    x = z * ts[0]
    for i in range(n_steps):
        t_cur, t_next = ts[i], ts[i + 1]
        x, t_hat = churn_step(x, t_cur, n_steps, None if noise is None else noise[i], S_churn, S_min, S_max, S_noise)
        den = edm_precond(W, cfg, x, t_hat, mask, mu, **kw)
        d = (1 / t_hat) * x - (1 / t_hat) * den
        x = x + (t_next - t_hat) * d
        if trace is not None:
            trace.append(x.clone())
    return x


def edm_heun_sampler(W, cfg, z: Tensor, mask: Tensor, mu: Tensor, n_steps: int, trace: Optional[list] = None,
                     noise: Optional[Tensor] = None, S_churn: float = 0, S_min: float = 0, S_max: float = float("inf"),
                     S_noise: float = 1, **kw) -> Tensor:
    """ablation_sampler(solver='heun', alpha=1, discretization='edm', schedule='linear', scaling='none')
    — edm.py:186-214.  Predictor as in Euler (edm.py:199-204); every step but the last then evaluates the
    network a second time at (x', t') with t' = t_hat + 1*h (fp32: not necessarily bit-equal to t_next) and
    averages the two slopes (edm.py:207-214)."""
    ts = edm_sigmas(n_steps, z.dtype)
    x = z * ts[0]
    for i in range(n_steps):
        t_cur, t_next = ts[i], ts[i + 1]
        x, t_hat = churn_step(x, t_cur, n_steps, None if noise is None else noise[i], S_churn, S_min, S_max, S_noise)
        h = t_next - t_hat
        den = edm_precond(W, cfg, x, t_hat, mask, mu, **kw)
        d_cur = (1 / t_hat) * x - (1 / t_hat) * den
        if i == n_steps - 1:
            x = x + h * d_cur
        else:
            x_prime = x + h * d_cur
            t_prime = t_hat + h
            den = edm_precond(W, cfg, x_prime, t_prime, mask, mu, **kw)
            d_prime = (1 / t_prime) * x_prime - (1 / t_prime) * den
            x = x + h * (0.5 * d_cur + 0.5 * d_prime)
        if trace is not None:
            trace.append(x.clone())
    return x


def diffusion_infer(W, cfg, mask: Tensor, mu: Tensor, n_timesteps: int, z: Tensor, solver: str = "euler", **kw) -> Tensor:
    """Diffusion.forward(infer=True) with the latent z = randn/temperature + mu supplied explicitly
    (GeDEX diffusion.py:225-229 / DEX :255-259).  The reference wires solver='euler' (diffusion.py:216);
    'heun' is the other branch of the same ablation_sampler."""
    if solver not in ("euler", "heun"):
        raise ValueError(f"solver must be 'euler' or 'heun', got {solver!r}")
    fn = edm_euler_sampler if solver == "euler" else edm_heun_sampler
    return fn(W, cfg, z, mask, mu, n_timesteps, **kw)


# ---------------------------------------------------------------------------------------------
# STFT / mel front-end (audio/)
def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True) — audio/stft.py:41."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def slaney_mel_basis(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """librosa 0.9.2 ``filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (htk=False, norm='slaney') —
    third-party, restated from its published algorithm; call site audio/stft.py:145-147."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def stft_basis(n_fft: int = 1024) -> np.ndarray:
    """Windowed DFT basis rows 0..n_fft/2 (real) then (imag) — audio/stft.py:26-47 (fp32 product)."""
    k = np.arange(n_fft // 2 + 1)[:, None]
    n = np.arange(n_fft)[None, :]
    ang = 2.0 * np.pi * k * n / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)]).astype(np.float32)       # np.fft.fft sign convention
    return basis * hann_periodic(n_fft).astype(np.float32)[None, :]


def mel_from_wav(wav: np.ndarray, n_fft=1024, hop=256, n_mels=80, sr=22050, fmin=0.0, fmax=8000.0,
                 dtype=np.float64):
    """get_mel_from_wav -> TacotronSTFT.mel_spectrogram -> STFT.transform — audio/tools.py:8-15,
    audio/stft.py:159-178,52-81: clip, reflect-pad n_fft/2, framed windowed DFT, magnitude,
    mel matmul, log(clamp(.,1e-5)); also energy = ||mag||_2 over frequency."""
    y = np.clip(np.asarray(wav, dtype=np.float32), -1.0, 1.0)
    y = np.pad(y, (n_fft // 2, n_fft // 2), mode="reflect").astype(dtype)
    n_frames = (len(y) - n_fft) // hop + 1
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = y[idx]                                                          # [frames, n_fft]
    spec = frames @ stft_basis(n_fft).astype(dtype).T                        # [frames, 2*(n_fft/2+1)]
    c = n_fft // 2 + 1
    mag = np.sqrt(spec[:, :c] ** 2 + spec[:, c:] ** 2).T                     # [513, frames]
    mel = slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax).astype(dtype) @ mag
    mel = np.log(np.maximum(mel, 1e-5))
    energy = np.sqrt((mag ** 2).sum(axis=0))
    return mel.astype(np.float32), energy.astype(np.float32)
