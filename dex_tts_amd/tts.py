"""Host-side mirrors of the two reference TTS modules, text in -> mel out on the GPU:

    GeDEXTTS(cfg).forward(x, x_lengths, n_timesteps, temperature=1.0, spk=None, length_scale=1.0)          GeDEX-TTS/model/tts.py:15-55
    DeXTTS(cfg).forward(x, x_lengths, ref, ref_lengths, sty, sty_lengths, lf0, lf0_lengths, n_timesteps,
                        temperature=1.0, spk=None, length_scale=1.0)                                       DEX-TTS/model/tts.py:14-73

``cfg`` is the reference's ``cfg.model`` (attribute or dict access: n_vocab, n_feats, n_spks, spk_emb_dim, encoder, decoder, dit
[, tv_encoder, lf0_encoder, tiv_encoder]).  Sub-modules carry the reference's names, so ``load_state_dict(ckpt['ema'])`` of a
reference checkpoint routes every key.  Every stage runs in libdexamd.so (dex_text_encode / dex_text_align / dex_style_encode /
dex_sample); torch is the container of the tensors and the device RNG.  Inference only: ``compute_loss`` is not built."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from .diffusion import DEXDiffusion, GeDEXDiffusion
from .style import StyleEncoders
from .text import TextEncoder


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def _dict(c):
    if c is None:
        raise ValueError("the model section lacks a sub-section this module needs (encoder / decoder / *_encoder)")
    return dict(c) if isinstance(c, dict) else {k: getattr(c, k) for k in vars(c)}


class _TTSBase(nn.Module):
    def _route(self, sd: Dict[str, torch.Tensor], strict: bool):
        """Split a TTS-level state dict by sub-module prefix; returns the keys nobody claimed."""
        left = dict(sd)
        take = lambda pfx: {k[len(pfx):]: left.pop(k) for k in [k for k in left if k.startswith(pfx)]}
        self.encoder.load_state_dict(take("encoder."), strict=strict)
        dec = take("decoder.")
        self.decoder.load_state_dict(dec, strict=strict)
        if hasattr(self, "spk_emb"):
            w = left.pop("spk_emb.weight", None)
            if w is not None:
                self.spk_emb.weight.data.copy_(w)
            elif strict:
                raise RuntimeError("missing spk_emb.weight")
        return left

    def compute_loss(self, *a, **k):
        raise NotImplementedError("training (MAS alignment + EDMLoss, tts.py:57-121) is out of scope: train with the reference module, the "
                                  "checkpoint keys are the same")


class GeDEXTTS(_TTSBase):
    def __init__(self, cfg):
        super().__init__()
        self.n_spks, self.n_feats = int(_get(cfg, "n_spks")), int(_get(cfg, "n_feats"))
        sed = int(_get(cfg, "spk_emb_dim", 64))
        if self.n_spks > 1:
            self.spk_emb = nn.Embedding(self.n_spks, sed)
        self.encoder = TextEncoder(**_dict(_get(cfg, "encoder")), n_vocab=int(_get(cfg, "n_vocab")), n_feats=self.n_feats, n_spks=self.n_spks, spk_emb_dim=sed)
        self.decoder = GeDEXDiffusion(**_dict(_get(cfg, "decoder")), dit_cfg=_get(cfg, "dit"), n_feats=self.n_feats, n_spks=self.n_spks, spk_emb_dim=sed)

    def load_state_dict(self, sd, strict: bool = True):
        left = self._route(sd, strict)
        if strict and left:
            raise RuntimeError(f"unexpected keys {sorted(left)[:4]}")
        return self

    @torch.no_grad()
    def forward(self, x, x_lengths, n_timesteps, temperature=1.0, spk=None, length_scale=1.0):
        if self.n_spks > 1:
            spk = self.spk_emb(spk)                                                          # tts.py:30-31 (a row gather)
        mu_x, logw, x_mask = self.encoder(x, x_lengths, spk=spk, length_scale=length_scale)   # :34, :37-38
        mu_y, y_mask, attn, y_lengths, y_max_length = self.encoder.align()                    # :39-47
        dec_out = self.decoder(mu_y, y_mask, mu_y, temperature=temperature, n_timesteps=n_timesteps, spk=spk, infer=True)      # :52
        return mu_y[:, :, :y_max_length], dec_out[:, :, :y_max_length], attn[:, :, :y_max_length]                             # :50,53,55


class DeXTTS(_TTSBase):
    def __init__(self, cfg):
        super().__init__()
        self.n_spks, self.n_feats = 0, int(_get(cfg, "n_feats"))                              # tts.py:18 forces n_spks = 0
        sed = int(_get(cfg, "spk_emb_dim", 64))
        tv, lf, ti = (_dict(_get(cfg, k)) for k in ("tv_encoder", "lf0_encoder", "tiv_encoder"))
        self.style = StyleEncoders(dict(tv_encoder=tv, lf0_encoder=lf, tiv_encoder=ti, dim=int(_get(_get(cfg, "decoder"), "dim"))))
        self.encoder = TextEncoder(**_dict(_get(cfg, "encoder")), n_vocab=int(_get(cfg, "n_vocab")), n_feats=self.n_feats, n_spks=0, spk_emb_dim=sed, variant="dex")
        self.decoder = DEXDiffusion(**_dict(_get(cfg, "decoder")), dit_cfg=_get(cfg, "dit"), n_feats=self.n_feats, n_spks=0, spk_emb_dim=sed)

    def load_state_dict(self, sd, strict: bool = True):
        left = self._route(sd, strict)
        style = {k: left.pop(k) for k in [k for k in left if k.split(".")[0] in ("tv_encoder", "lf0_encoder", "tiv_encoder", "conv_sty")]}
        self.style.load_state_dict(style, strict=strict)
        if strict and left:
            raise RuntimeError(f"unexpected keys {sorted(left)[:4]}")
        return self

    @torch.no_grad()
    def forward(self, x, x_lengths, ref, ref_lengths, sty, sty_lengths, lf0, lf0_lengths, n_timesteps, temperature=1.0, spk=None, length_scale=1.0):
        ref_skips, sty_dec, sty_enc = self.style(ref, ref_lengths, sty, sty_lengths, lf0, lf0_lengths)                       # tts.py:55-67
        mu_x, logw, x_mask = self.encoder(x, x_lengths, sty_enc, spk=None, length_scale=length_scale)                         # :68
        mu_y, y_mask, attn, y_lengths, y_max_length = self.encoder.align()
        dec_out = self.decoder(mu_y, y_mask, mu_y, ref_skips, ref_lengths, sty_dec, sty_lengths, temperature=temperature,
                               n_timesteps=n_timesteps, spk=spk, infer=True)                                                  # :84
        return mu_y[:, :, :y_max_length], dec_out[:, :, :y_max_length], attn[:, :, :y_max_length]
