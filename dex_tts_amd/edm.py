"""Host-side mirror of the reference's ``EDMLoss`` (GeDEX-TTS/model/edm.py:22-68, DEX-TTS/model/edm.py:22-68) — the
``infer=False`` branch of ``Diffusion.forward`` (diffusion.py:222-224 / :252-254), SURVEY §8 row f4.

Same constructor, same ``forward`` argument order, same two generator draws in the same order (``randn([B,1,1])`` for the
noise level, then ``randn_like(x0)``), same weighting branches (``loss_type`` 'base', 'base_min_k', 'base_log_k', 'min_snr_k',
'max_snr_k', 'snr', 'inv_snr').  The denoiser evaluation ``D(x0 + n; sigma)`` — all of the arithmetic that matters — runs in
libdexamd.so (``dex_denoise_once``), one utterance per call because every utterance has its own sigma and the library's
conditioning tables are per call; the per-utterance scalars (sigma, weight) and the final masked mean are a handful of stock
torch ops on [B,1,1] / [B,80,T] tensors, as VERDICT round 2 item 7 allows.

This is the loss VALUE (validation / monitoring, parity with the reference on fixed draws — tests/golden/edm_loss.npz).  The HIP
path has no backward pass: training stays with the reference module, and a tensor that requires grad is refused loudly rather
than silently detached.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def loss_weight(sigma: torch.Tensor, loss_type: str, sigma_data: float = 0.5) -> torch.Tensor:
    """lambda(sigma) of edm.py:37-63 (per utterance, [B,1,1])."""
    snr = 1 / sigma ** 2
    base = (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2
    if loss_type == "base":
        return base
    for prefix, fn in (("base_min_", lambda k: torch.clamp(base, max=k)),
                       ("base_log_", lambda k: torch.where(base >= k, torch.log(base) + (k - math.log(k)), base)),
                       ("min_snr_", lambda k: torch.clamp(snr, max=k)),
                       ("max_snr_", lambda k: torch.clamp(snr, min=k))):
        if loss_type.startswith(prefix):
            return fn(float(loss_type.split(prefix)[-1]))
    if loss_type == "snr":
        return snr
    if loss_type == "inv_snr":
        return 1.0 / snr
    raise ValueError(f"unknown loss_type {loss_type!r} (edm.py:37-63 knows base, base_min_k, base_log_k, min_snr_k, max_snr_k, snr, inv_snr)")


class EDMLoss(nn.Module):
    def __init__(self, P_mean=-1.2, P_std=1.2, sigma_data=0.5, n_feats=80, loss_type="base"):
        super().__init__()
        self.P_mean, self.P_std, self.sigma_data, self.n_feats, self.loss_type = P_mean, P_std, sigma_data, n_feats, loss_type

    @torch.no_grad()
    def forward(self, precond_model, x0, mask, mu, *dex_args, spk=None, mask_ratio=0, rnd_normal=None, eps=None):
        """GeDEX: ``forward(precond_model, x0, mask, mu, spk=None, mask_ratio=0)``; DEX: ``forward(precond_model, x0, mask, mu,
        ref, ref_lengths, sty, sty_lengths, spk=None, mask_ratio=0)``.  ``precond_model`` is the owning ``Diffusion``'s
        ``precond_model`` (it carries the engine hook); ``rnd_normal`` / ``eps`` inject the two draws (tests), otherwise they
        come from the device generator exactly as in the reference."""
        if mask_ratio:
            raise NotImplementedError("mask_ratio > 0 (DiT token masking, dit.py:145-163) is unreachable with the shipped configs and not built")
        if any(t is not None and torch.is_tensor(t) and t.requires_grad for t in (x0, mu)):
            raise RuntimeError("EDMLoss here is forward-only (the HIP score network has no backward); train with the reference module")
        denoise = getattr(precond_model, "_denoise_once", None)
        if denoise is None:
            raise TypeError("precond_model must be dex_tts_amd.diffusion.Diffusion.precond_model")
        ref = ref_lengths = sty = sty_lengths = None
        if dex_args:
            if len(dex_args) != 4:
                raise TypeError("DEX: forward(precond_model, x0, mask, mu, ref, ref_lengths, sty, sty_lengths, ...)")
            ref, ref_lengths, sty, sty_lengths = dex_args
        B = x0.shape[0]
        if rnd_normal is None:
            rnd_normal = torch.randn([B, 1, 1], device=x0.device)                 # edm.py:33
        sigma = (rnd_normal.to(x0.device) * self.P_std + self.P_mean).exp()
        weight = loss_weight(sigma, self.loss_type, self.sigma_data)
        if eps is None:
            eps = torch.randn_like(x0)                                            # edm.py:65
        n = (eps.to(x0.device) + mu) * sigma
        xn = x0 + n
        sig_host = sigma.reshape(B).tolist()                                       # one host read: each utterance has its own sigma
        rows = []
        for b in range(B):
            kw = {}
            if ref is not None:
                kw = dict(ref=[r[b:b + 1] for r in ref], sty=sty[b:b + 1], sty_lengths=sty_lengths[b:b + 1])
            if spk is not None:
                kw["spk"] = spk[b:b + 1]
            rows.append(denoise(xn[b:b + 1], sig_host[b], mask[b:b + 1], mu[b:b + 1], **kw))
        D_yn = torch.cat(rows, 0)
        return torch.sum(weight * ((D_yn - x0) ** 2)) / torch.sum(mask * self.n_feats)          # edm.py:66
