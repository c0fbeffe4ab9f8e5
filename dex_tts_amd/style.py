"""Host-side mirror of the DEX style encoders and the pre-decoder lines of ``DeXTTS.forward`` (DEX-TTS/model/tts.py:26-31,
55-67; model/ref_encoder.py TVEncoder / LF0Encoder / TIVEncoder): one module holding ``tv_encoder.* / lf0_encoder.* /
tiv_encoder.* / conv_sty.*`` under the reference's state-dict names, so the matching entries of a DeXTTS checkpoint load
unchanged.  ``forward(ref, ref_lengths, sty, sty_lengths, lf0, lf0_lengths)`` returns what the reference feeds on:

    ref_skips  list of [B, c_h, Tr]      -> Diffusion.forward(ref=...)            (tts.py:67,84)
    sty_dec    [B, 2*dim, Ts]            -> Diffusion.forward(sty=...)            (tts.py:65-66)
    sty_enc    [B, c_out]                -> TextEncoder(x, x_lengths, sty_enc)    (tts.py:62-63,68)

Eval mode only (dropout off, BatchNorm on running statistics — folded into the preceding convolution here —, frozen VQ
codebook).  The arithmetic runs in libdexamd.so (``dex_style_encode``); no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

VCTK = dict(tv_encoder=dict(c_in=80, num_layer=6, c_h=128, c_out=192, c_out_g=192, commit_w=0.25, n_emb=512),
            lf0_encoder=dict(c_in=1, c_h=192, c_out=192, c_out_g=192, num_layer=2),
            tiv_encoder=dict(c_in=80, num_layer=6, c_h=128, c_out=64), dim=64)


def _basic(out, p, cin, cout, norm):
    out[f"{p}.conv.weight"] = (cout, cin, 3)
    if norm == "ln":
        out[f"{p}.ln.weight"] = (cout,); out[f"{p}.ln.bias"] = (cout,)
    elif norm == "bn":
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[f"{p}.bn.{k}"] = (cout,)
        out[f"{p}.bn.num_batches_tracked"] = ()


def _projection(out, p, cin, ch):
    out[f"{p}.conv_1.weight"] = (ch, cin, 3); out[f"{p}.conv_1.bias"] = (ch,)
    out[f"{p}.norm_1.gamma"] = (ch,); out[f"{p}.norm_1.beta"] = (ch,)
    out[f"{p}.conv_2.weight"] = (ch, ch, 3); out[f"{p}.conv_2.bias"] = (ch,)
    out[f"{p}.norm_2.gamma"] = (ch,); out[f"{p}.norm_2.beta"] = (ch,)
    out[f"{p}.proj.weight"] = (ch, ch, 1); out[f"{p}.proj.bias"] = (ch,)


def param_shapes(cfg) -> Dict[str, tuple]:
    """The reference state dict of (tv_encoder, lf0_encoder, tiv_encoder, conv_sty), in registration order."""
    tv, lf, ti = cfg["tv_encoder"], cfg["lf0_encoder"], cfg["tiv_encoder"]
    o: Dict[str, tuple] = {}
    p = "tv_encoder"
    _basic(o, f"{p}.in_conv", tv["c_in"], tv["c_h"], "ln")
    for i in range(tv["num_layer"]):
        _basic(o, f"{p}.conv_blocks.{i}.conv_block.0", tv["c_h"], tv["c_h"], "ln")
        _basic(o, f"{p}.conv_blocks.{i}.conv_block.1", tv["c_h"], tv["c_h"], "")
    _basic(o, f"{p}.out_conv", tv["c_h"], tv["c_out"], "")
    o[f"{p}.vq.embedding"] = (tv["n_emb"], tv["c_out"]); o[f"{p}.vq.ema_count"] = (tv["n_emb"],); o[f"{p}.vq.ema_weight"] = (tv["n_emb"], tv["c_out"])
    _projection(o, f"{p}.proj_0", tv["c_out"], tv["c_out_g"])
    _basic(o, f"{p}.proj_1", tv["c_out_g"], tv["c_out_g"], "bn")
    p = "lf0_encoder"
    _basic(o, f"{p}.in_conv", lf["c_in"], lf["c_h"], "ln")
    H = lf["c_h"] // 2
    for l in range(lf["num_layer"]):
        for sfx in ("", "_reverse"):
            o[f"{p}.rnn_layer.weight_ih_l{l}{sfx}"] = (3 * H, lf["c_h"]); o[f"{p}.rnn_layer.weight_hh_l{l}{sfx}"] = (3 * H, H)
            o[f"{p}.rnn_layer.bias_ih_l{l}{sfx}"] = (3 * H,); o[f"{p}.rnn_layer.bias_hh_l{l}{sfx}"] = (3 * H,)
    _basic(o, f"{p}.out_conv", lf["c_h"], lf["c_out"], "ln")
    _projection(o, f"{p}.proj", lf["c_out"], lf["c_out_g"])
    p = "tiv_encoder"
    _basic(o, f"{p}.in_conv", ti["c_in"], ti["c_h"], "bn")
    for i in range(ti["num_layer"]):
        _basic(o, f"{p}.conv_blocks.{i}.conv_block.0", ti["c_h"], ti["c_h"], "bn")
        _basic(o, f"{p}.conv_blocks.{i}.conv_block.1", ti["c_h"], ti["c_h"], "")
    _basic(o, f"{p}.out_conv", ti["c_h"], ti["c_out"], "bn")
    o["conv_sty.weight"] = (cfg["dim"] * 2, tv["c_out_g"], 1); o["conv_sty.bias"] = (cfg["dim"] * 2,)
    return o


def fold_batchnorm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """BasicConv with norm_type 'bn' in eval mode (base.py:54-58: conv without bias -> BatchNorm1d(eps 1e-5) on running
    statistics) == conv with weight * s[co] and bias beta - mean * s, s = gamma / sqrt(var + eps).  Returns the library's
    view of the state dict: '<p>.conv.weight' + '<p>.conv.bias' replace '<p>.bn.*'."""
    out = {}
    for k, v in sd.items():
        if ".bn." in k:
            continue
        out[k] = v
    for k in sd:
        if k.endswith(".bn.weight"):
            p = k[: -len(".bn.weight")]
            s = sd[f"{p}.bn.weight"].double() / torch.sqrt(sd[f"{p}.bn.running_var"].double() + 1e-5)
            out[f"{p}.conv.weight"] = (sd[f"{p}.conv.weight"].double() * s.view(-1, 1, 1)).float()
            out[f"{p}.conv.bias"] = (sd[f"{p}.bn.bias"].double() - sd[f"{p}.bn.running_mean"].double() * s).float()
    return out


class StyleEncoders(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = dict(VCTK) if cfg is None else cfg
        self.shapes = param_shapes(self.cfg)
        for key, shape in self.shapes.items():
            dt = torch.int64 if key.endswith("num_batches_tracked") else torch.float32
            self.register_buffer(key.replace(".", "__"), torch.zeros(shape, dtype=dt), persistent=False)
        self._ctx = None
        self._lib = None
        self._loaded_key = None
        self._ws = None

    def state_dict(self, *a, **k):
        return {key: getattr(self, key.replace(".", "__")) for key in self.shapes}

    def load_state_dict(self, sd, strict: bool = True):
        """Accepts the four sub-dicts of a DeXTTS checkpoint (keys tv_encoder.* / lf0_encoder.* / tiv_encoder.* / conv_sty.*;
        other keys are ignored unless ``strict``)."""
        mine = {k: v for k, v in sd.items() if k in self.shapes}
        missing = [k for k in self.shapes if k not in mine]
        if strict and (missing or len(mine) != len(sd)):
            extra = [k for k in sd if k not in self.shapes]
            raise RuntimeError(f"StyleEncoders.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, v in mine.items():
            buf = getattr(self, k.replace(".", "__"))
            if tuple(v.shape) != tuple(buf.shape):
                raise RuntimeError(f"{k}: shape {tuple(v.shape)} != {tuple(buf.shape)}")
            buf.copy_(v.detach().to(buf.dtype))
        self._loaded_key = None
        return self

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.dex_style_last_error(self._ctx)
            raise RuntimeError(f"libdexamd style error {rc}: {msg.decode() if msg else '?'}")

    def _engine(self, device):
        if device.type != "cuda":
            raise RuntimeError("dex_tts_amd runs on an AMD GPU (torch device 'cuda' on ROCm); no CPU path exists")
        if self._ctx is None:
            self._lib = _lib.load()
            tv, lf, ti = self.cfg["tv_encoder"], self.cfg["lf0_encoder"], self.cfg["tiv_encoder"]
            c = _lib.DexStyleConfig(tv["c_in"], ti["num_layer"], ti["c_h"], tv["num_layer"], tv["c_h"], tv["c_out"], tv["c_out_g"], tv["n_emb"],
                                    lf["c_h"], lf["c_out"], lf["c_out_g"], lf["num_layer"], self.cfg["dim"] * 2)
            ctx = C.c_void_p()
            rc = self._lib.dex_style_create(C.byref(c), C.byref(ctx))
            self._ctx = ctx
            self._check(rc)
        bufs = [getattr(self, k.replace(".", "__")) for k in self.shapes]
        key = (str(device),) + tuple((b._version, b.data_ptr()) for b in bufs)
        if key != self._loaded_key:
            folded = fold_batchnorm({k: v.float() if v.is_floating_point() else v for k, v in self.state_dict().items()})
            with torch.cuda.device(device):
                st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                keep = []
                for i in range(self._lib.dex_style_num_weights(self._ctx)):
                    name = C.c_char_p(); shp = (C.c_int64 * 4)(); nd = C.c_int()
                    self._check(self._lib.dex_style_weight_info(self._ctx, i, C.byref(name), shp, C.byref(nd)))
                    k = name.value.decode()
                    w = folded[k].to(device=device, dtype=torch.float32).contiguous()
                    shape = (C.c_int64 * 4)(*([int(s) for s in w.shape] + [0] * (4 - w.dim())))
                    self._check(self._lib.dex_style_load_weight_async(self._ctx, k.encode(), C.c_void_p(w.data_ptr()), shape, w.dim(), st))
                    keep.append(w)
                self._check(self._lib.dex_style_finalize(self._ctx, st))
            self._loaded_key = key

    def __del__(self):
        try:
            if self._ctx is not None and self._ctx.value:
                self._lib.dex_style_destroy(self._ctx)
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, ref: torch.Tensor, ref_lengths: torch.Tensor, sty: torch.Tensor, sty_lengths: torch.Tensor,
                lf0: torch.Tensor, lf0_lengths: torch.Tensor, return_indices: bool = False):
        """tts.py:55-67: ref [B,80,Tr] (or [B,1,80,Tr]), sty [B,80,Ts], lf0 [B,Tl] + lengths -> (ref_skips, sty_dec, sty_enc)."""
        dev = ref.device
        self._engine(dev)
        with torch.cuda.device(dev):
            f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
            i = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
            ref, sty, lf0 = f(ref.squeeze(1) if ref.dim() == 4 else ref), f(sty.squeeze(1) if sty.dim() == 4 else sty), f(lf0)
            rl, sl, ll = i(ref_lengths), i(sty_lengths), i(lf0_lengths)
            B, M, Tr = ref.shape
            Ts, Tl = sty.shape[2], lf0.shape[1]
            tv, ti = self.cfg["tv_encoder"], self.cfg["tiv_encoder"]
            if sty.shape[:2] != (B, M) or lf0.shape[0] != B or M != tv["c_in"]:
                raise ValueError("ref / sty must be [B, n_mels, T] and lf0 [B, T] with one batch size")
            skips = [torch.empty(B, ti["c_h"], Tr, dtype=torch.float32, device=dev) for _ in range(ti["num_layer"])]
            sty_dec = torch.empty(B, self.cfg["dim"] * 2, Ts, dtype=torch.float32, device=dev)
            sty_enc = torch.empty(B, tv["c_out"], dtype=torch.float32, device=dev)
            idx = torch.empty(B, Ts, dtype=torch.int32, device=dev)
            need = int(self._lib.dex_style_workspace_bytes(self._ctx, B, Tr, Ts, Tl))
            if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
                self._ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            base = (self._ws.data_ptr() + 255) // 256 * 256
            arr = (C.c_void_p * len(skips))(*[s.data_ptr() for s in skips])
            a = _lib.DexStyleArgs(B, Tr, Ts, Tl, ref.data_ptr(), rl.data_ptr(), sty.data_ptr(), sl.data_ptr(), lf0.data_ptr(), ll.data_ptr(),
                                  C.cast(arr, C.POINTER(C.c_void_p)), sty_dec.data_ptr(), sty_enc.data_ptr(), idx.data_ptr(), base,
                                  self._ws.numel() - (base - self._ws.data_ptr()))
            self._check(self._lib.dex_style_encode(self._ctx, C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            self._keep = (ref, sty, lf0, rl, sl, ll, arr)
            return (skips, sty_dec, sty_enc, idx) if return_indices else (skips, sty_dec, sty_enc)
