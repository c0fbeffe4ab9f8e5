"""Host-side mirror of the reference text path: ``TextEncoder`` with the reference's constructor arguments and
state-dict names (GeDEX-TTS/model/text_encoder.py:96-146, DEX-TTS/model/text_encoder.py:93-142) and the duration / alignment
lines of the TTS forward (tts.py:37-50) as ``TextEncoder.align``.  Inference only; the arithmetic runs in libdexamd.so
(``dex_text_encode`` / ``dex_text_align``) — there is no CPU path.

    enc = TextEncoder(**cfg.encoder, n_vocab=.., n_feats=80, n_spks=.., spk_emb_dim=64)      # as tts.py:24 builds it
    enc.load_state_dict({k[len("encoder."):]: v for k, v in ckpt.items() if k.startswith("encoder.")})
    mu_x, logw, x_mask = enc(x, x_lengths, spk=spk)                 # DEX: enc(x, x_lengths, sty_enc)
    mu_x, logw, x_mask = enc(x, x_lengths, spk=spk, length_scale=1.0)   # the duration scale belongs to forward (tts.py:37)
    mu_y, y_mask, attn, y_lengths, y_max_length = enc.align()       # alignment of the LAST forward
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib


def fix_len_compatibility(length: int, num_downsamplings_in_unet: int = 2) -> int:     # model/utils.py:13-17
    while length % (2 ** num_downsamplings_in_unet):
        length += 1
    return length


def param_shapes(n_vocab, n_feats, n_channels, filter_channels, filter_channels_dp, n_heads, n_layers, kernel_size,
                 spk_emb_dim=64, n_spks=1, variant="gedex") -> Dict[str, tuple]:
    """The reference TextEncoder state dict (parameters and the two RetNetRelPos buffers), in registration order."""
    E = n_channels + (spk_emb_dim if n_spks > 1 else 0)
    o: Dict[str, tuple] = {"emb.weight": (n_vocab, n_channels)}
    for i in range(3):
        o[f"prenet.conv_layers.{i}.weight"] = (n_channels, n_channels, 5); o[f"prenet.conv_layers.{i}.bias"] = (n_channels,)
    for i in range(3):
        o[f"prenet.norm_layers.{i}.gamma"] = (n_channels,); o[f"prenet.norm_layers.{i}.beta"] = (n_channels,)
    o["prenet.proj.weight"] = (n_channels, n_channels, 1); o["prenet.proj.bias"] = (n_channels,)
    for i in range(n_layers):
        p = f"encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "g_proj", "out_proj"):
            o[f"{p}.retention.{n}.weight"] = (E, E)
        o[f"{p}.retention_layer_norm.weight"] = (E,)
        o[f"{p}.ffn.fc1.weight"] = (filter_channels, E); o[f"{p}.ffn.fc2.weight"] = (E, filter_channels); o[f"{p}.ffn.gate.weight"] = (filter_channels, E)
        o[f"{p}.final_layer_norm.weight"] = (E,)
        if variant == "dex":
            for a in ("adaln_1", "adaln_2"):
                for w in ("W_scale", "W_bias"):
                    o[f"{p}.{a}.{w}.weight"] = (E, E); o[f"{p}.{a}.{w}.bias"] = (E,)
    o["encoder.layer_norm.weight"] = (E,)
    o["encoder.retnet_rel_pos.angle"] = (E // n_heads,); o["encoder.retnet_rel_pos.decay"] = (n_heads,)
    o["proj_m.weight"] = (n_feats, E, 1); o["proj_m.bias"] = (n_feats,)
    o["proj_w.conv_1.weight"] = (filter_channels_dp, E, kernel_size); o["proj_w.conv_1.bias"] = (filter_channels_dp,)
    o["proj_w.norm_1.gamma"] = (filter_channels_dp,); o["proj_w.norm_1.beta"] = (filter_channels_dp,)
    o["proj_w.conv_2.weight"] = (filter_channels_dp, filter_channels_dp, kernel_size); o["proj_w.conv_2.bias"] = (filter_channels_dp,)
    o["proj_w.norm_2.gamma"] = (filter_channels_dp,); o["proj_w.norm_2.beta"] = (filter_channels_dp,)
    o["proj_w.proj.weight"] = (1, filter_channels_dp, 1); o["proj_w.proj.bias"] = (1,)
    return o


class TextEncoder(nn.Module):
    def __init__(self, n_vocab, n_feats, n_channels, filter_channels, filter_channels_dp, n_heads, n_layers, kernel_size, p_dropout=0.1,
                 use_softmax=True, use_decay=False, window_size=None, spk_emb_dim=64, n_spks=1, variant="gedex"):
        super().__init__()
        if not use_softmax or use_decay:
            raise NotImplementedError("only use_softmax=True / use_decay=False is built (what every shipped config sets)")
        self.n_vocab, self.n_feats, self.n_channels, self.n_heads, self.n_layers = n_vocab, n_feats, n_channels, n_heads, n_layers
        self.filter_channels, self.filter_channels_dp, self.kernel_size = filter_channels, filter_channels_dp, kernel_size
        self.spk_emb_dim, self.n_spks, self.variant = spk_emb_dim, n_spks, variant
        self.shapes = param_shapes(n_vocab, n_feats, n_channels, filter_channels, filter_channels_dp, n_heads, n_layers, kernel_size,
                                   spk_emb_dim, n_spks, variant)
        for key, shape in self.shapes.items():
            self.register_buffer(key.replace(".", "__"), torch.zeros(shape, dtype=torch.float32), persistent=False)
        self._ctx = None
        self._lib = None
        self._loaded_key = None
        self._ws = None
        self._last = None

    # ---- checkpoint surface
    def state_dict(self, *a, **k):
        return {key: getattr(self, key.replace(".", "__")) for key in self.shapes}

    def load_state_dict(self, sd, strict: bool = True):
        mine = {k: v for k, v in sd.items() if k in self.shapes}
        missing = [k for k in self.shapes if k not in mine]
        extra = [k for k in sd if k not in self.shapes]
        if strict and (missing or extra):
            raise RuntimeError(f"TextEncoder.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, v in mine.items():
            buf = getattr(self, k.replace(".", "__"))
            if tuple(v.shape) != tuple(buf.shape):
                raise RuntimeError(f"{k}: shape {tuple(v.shape)} != {tuple(buf.shape)}")
            buf.copy_(v.detach().to(buf.dtype))
        self._loaded_key = None
        return self

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.dex_text_last_error(self._ctx)
            raise RuntimeError(f"libdexamd text error {rc}: {msg.decode() if msg else '?'}")

    def _engine(self, device):
        if device.type != "cuda":
            raise RuntimeError("dex_tts_amd runs on an AMD GPU (torch device 'cuda' on ROCm); no CPU path exists")
        if self._ctx is None:
            self._lib = _lib.load()
            c = _lib.DexTextConfig(_lib.VARIANT[self.variant], self.n_vocab, self.n_feats, self.n_channels, self.filter_channels,
                                   self.filter_channels_dp, self.n_heads, self.n_layers, self.kernel_size, self.n_spks, self.spk_emb_dim, 1, 0)
            ctx = C.c_void_p()
            rc = self._lib.dex_text_create(C.byref(c), C.byref(ctx))
            self._ctx = ctx
            self._check(rc)
        bufs = [getattr(self, k.replace(".", "__")) for k in self.shapes]
        key = (str(device),) + tuple((b._version, b.data_ptr()) for b in bufs)
        if key != self._loaded_key:
            sd = self.state_dict()
            with torch.cuda.device(device):
                st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                keep = []
                for i in range(self._lib.dex_text_num_weights(self._ctx)):
                    name = C.c_char_p(); shp = (C.c_int64 * 4)(); nd = C.c_int()
                    self._check(self._lib.dex_text_weight_info(self._ctx, i, C.byref(name), shp, C.byref(nd)))
                    k = name.value.decode()
                    w = sd[k].to(device=device, dtype=torch.float32).contiguous()
                    shape = (C.c_int64 * 4)(*([int(s) for s in w.shape] + [0] * (4 - w.dim())))
                    self._check(self._lib.dex_text_load_weight_async(self._ctx, k.encode(), C.c_void_p(w.data_ptr()), shape, w.dim(), st))
                    keep.append(w)
                self._check(self._lib.dex_text_finalize(self._ctx, st))
            self._loaded_key = key

    def __del__(self):
        try:
            if self._ctx is not None and self._ctx.value:
                self._lib.dex_text_destroy(self._ctx)
        except Exception:
            pass

    def _workspace(self, need, dev):
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        return base, self._ws.numel() - (base - self._ws.data_ptr())

    @torch.no_grad()
    def forward(self, x: torch.Tensor, x_lengths: torch.Tensor, *args, spk: Optional[torch.Tensor] = None, length_scale: float = 1.0):
        """GeDEX: ``forward(x, x_lengths, spk=None)``; DEX: ``forward(x, x_lengths, sty, spk=None)`` (text_encoder.py:129 / :126).
        x [B,T] token ids, x_lengths [B] -> mu [B,n_feats,T], logw [B,1,T], x_mask [B,1,T]."""
        sty = None
        if self.variant == "dex":
            if len(args) != 1:
                raise TypeError("the DEX text encoder is called as forward(x, x_lengths, sty, spk=None)")
            sty = args[0]
        elif len(args) == 1 and spk is None:
            spk = args[0]
        elif args:
            raise TypeError("forward(x, x_lengths, spk=None)")
        dev = x.device
        self._engine(dev)
        with torch.cuda.device(dev):
            tok = x.to(device=dev, dtype=torch.int32).contiguous()
            xl = x_lengths.to(device=dev, dtype=torch.int32).contiguous()
            B, T = tok.shape
            if int(x_lengths.max()) > T or int(x_lengths.min()) < 1:
                raise ValueError("x_lengths must lie in [1, x.shape[1]]")
            lo, hi = int(tok.min()), int(tok.max())
            if lo < 0 or hi >= self.n_vocab:                 # nn.Embedding raises (text_encoder.py:130); the kernel would clamp silently
                raise IndexError(f"token id out of range: [{lo}, {hi}] outside [0, {self.n_vocab})")
            f = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
            spk, sty = f(spk if self.n_spks > 1 else None), f(sty)
            if self.n_spks > 1 and (spk is None or tuple(spk.shape) != (B, self.spk_emb_dim)):
                raise ValueError("n_spks > 1: spk must be the [B, spk_emb_dim] embedding rows (spk_emb(spk), tts.py:31)")
            if sty is not None and tuple(sty.shape) != (B, self.n_channels):
                raise ValueError("sty must be [B, n_channels]")
            mu = torch.empty(B, self.n_feats, T, dtype=torch.float32, device=dev)
            logw = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
            w_ceil = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
            y_len = torch.empty(B, dtype=torch.int32, device=dev)
            base, nbytes = self._workspace(int(self._lib.dex_text_workspace_bytes(self._ctx, B, T)), dev)
            a = _lib.DexTextArgs(B, T, tok.data_ptr(), xl.data_ptr(), spk.data_ptr() if spk is not None else None,
                                 sty.data_ptr() if sty is not None else None, float(length_scale), mu.data_ptr(), logw.data_ptr(), w_ceil.data_ptr(),
                                 y_len.data_ptr(), base, nbytes)
            self._check(self._lib.dex_text_encode(self._ctx, C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            x_mask = (torch.arange(T, device=dev)[None, :] < xl[:, None]).to(torch.float32).unsqueeze(1)       # sequence_mask: the returned tensor only
            self._last = dict(mu=mu, w_ceil=w_ceil, y_len=y_len, xl=xl, keep=(tok, spk, sty))
            return mu, logw, x_mask

    @torch.no_grad()
    def align(self, return_attn: bool = True):
        """tts.py:37-50 for the last ``forward``: -> mu_y [B,n_feats,Ty_], y_mask [B,1,Ty_], attn [B,1,T,Ty_] (or None),
        y_lengths [B] (int64), y_max_length (Ty_ = fix_len_compatibility(y_max_length); callers crop to y_max_length)."""
        if self._last is None:
            raise RuntimeError("align() follows a forward() call")
        L = self._last
        mu, w_ceil, y_len, xl = L["mu"], L["w_ceil"], L["y_len"], L["xl"]
        dev = mu.device
        with torch.cuda.device(dev):
            B, F, T = mu.shape
            y_max = int(y_len.max().item())                           # the one host read of the path (the reference does the same, tts.py:40)
            Ty = fix_len_compatibility(y_max)
            mu_y = torch.empty(B, F, Ty, dtype=torch.float32, device=dev)
            y_mask = torch.empty(B, 1, Ty, dtype=torch.float32, device=dev)
            attn = torch.empty(B, 1, T, Ty, dtype=torch.float32, device=dev) if return_attn else None
            cum = torch.empty(B * T, dtype=torch.float32, device=dev)
            a = _lib.DexAlignArgs(B, T, Ty, mu.data_ptr(), w_ceil.data_ptr(), xl.data_ptr(), y_len.data_ptr(), mu_y.data_ptr(), y_mask.data_ptr(),
                                  attn.data_ptr() if attn is not None else None, cum.data_ptr(), cum.numel() * 4)
            self._check(self._lib.dex_text_align(self._ctx, C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            self._keep = cum
            return mu_y, y_mask, attn, y_len.to(torch.int64), y_max
