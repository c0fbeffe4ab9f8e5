"""Host-side mirror of the reference's HiFi-GAN ``Generator`` (GeDEX-TTS/hifigan/models.py:112-173, built by
``get_vocoder`` src/utils.py:251-281 from hifigan/config.json): same constructor (``Generator(h)`` with the config
attributes), same state-dict keys — with or without weight norm (``*.weight_g`` / ``*.weight_v`` pairs of a training
checkpoint, or plain ``*.weight`` after ``remove_weight_norm()``) — same ``forward(mel [B,80,T]) -> wav [B,1,T*256]``.
The arithmetic runs in libdexamd.so (``dex_vocode``: implicit-GEMM convolutions on the exact-fp32 MFMA path); PyTorch owns
tensors and the stream.  No CPU path."""
from __future__ import annotations

import ctypes as C
import json
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib

HIFIGAN_V1 = dict(num_mels=80, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
                  resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])


# BigVGAN-base, 22 kHz / 80 bands: the configuration src/utils.py:267 reads from bigvgan/bigvgan_base_22khz_80band/config.json (not in
# the reference tree; these are the published values of that file).  The 112 M "bigvgan_22khz_80band" model (1536 initial channels,
# six up-sampling stages down to 24 channels) is outside the implicit GEMM's 32-channel granule and is not built.
BIGVGAN_BASE = dict(num_mels=80, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
                    resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                    activation="snakebeta", snake_logscale=True)
ACTIVATION = {None: 0, "snake": 1, "snakebeta": 2}


class AttrDict(dict):
    """hifigan/__init__.py's AttrDict: config.json keys as attributes."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


def _get(h, name, default=None):
    if isinstance(h, dict):
        return h.get(name, default)
    return getattr(h, name, default)


def param_shapes(h) -> Dict[str, tuple]:
    """Generator.state_dict() after remove_weight_norm(): key -> shape (hifigan/models.py:116-148; with ``activation`` =
    'snake' / 'snakebeta' in the config: BigVGAN, bigvgan/models.py:141-184 — nested ``ups.<i>.0``, an Activation1d per ResBlock conv
    with its alpha [, beta] and the two resampling-filter buffers, ``activation_post``)."""
    c0 = int(_get(h, "upsample_initial_channel"))
    rates, ksz = list(_get(h, "upsample_rates")), list(_get(h, "upsample_kernel_sizes"))
    rk = list(_get(h, "resblock_kernel_sizes"))
    act = _get(h, "activation")
    if act not in ACTIVATION:
        raise ValueError(f"activation {act!r}: expected 'snake' or 'snakebeta' (BigVGAN) or none (HiFi-GAN)")
    ups = ".0" if act else ""
    out = {"conv_pre.weight": (c0, int(_get(h, "num_mels", 80)), 7), "conv_pre.bias": (c0,)}
    for i, k in enumerate(ksz):
        out[f"ups.{i}{ups}.weight"] = (c0 >> i, c0 >> (i + 1), k)
        out[f"ups.{i}{ups}.bias"] = (c0 >> (i + 1),)

    def activation(p, ch):
        out[f"{p}.act.alpha"] = (ch,)
        if act == "snakebeta":
            out[f"{p}.act.beta"] = (ch,)
        out[f"{p}.upsample.filter"] = (1, 1, 12); out[f"{p}.downsample.lowpass.filter"] = (1, 1, 12)
    for i in range(len(rates)):
        ch = c0 >> (i + 1)
        for j, k in enumerate(rk):
            for cs in ("convs1", "convs2"):
                for m in range(3):
                    out[f"resblocks.{i * len(rk) + j}.{cs}.{m}.weight"] = (ch, ch, k)
                    out[f"resblocks.{i * len(rk) + j}.{cs}.{m}.bias"] = (ch,)
            if act:
                for l in range(6):
                    activation(f"resblocks.{i * len(rk) + j}.activations.{l}", ch)
    if act:
        activation("activation_post", c0 >> len(rates))
    out["conv_post.weight"] = (1, c0 >> len(rates), 7)
    out["conv_post.bias"] = (1,)
    return out


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``weight = g * v / ||v||`` with the norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0 — also for the
    ConvTranspose1d layers, whose dim 0 is the INPUT channel): what remove_weight_norm() leaves (models.py:169-173)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len("_g")]
            wv = sd[base + "_v"].float()
            norm = wv.flatten(1).norm(dim=1).reshape(-1, *([1] * (wv.dim() - 1)))
            out[base] = v.float() * wv / norm
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


class Generator(nn.Module):
    def __init__(self, h=None):
        super().__init__()
        h = AttrDict(HIFIGAN_V1) if h is None else h
        if str(_get(h, "resblock", "1")) != "1":
            raise ValueError("only ResBlock / AMPBlock type '1' (hifigan/config.json V1, bigvgan base) is built")
        self.h = h
        self.shapes = param_shapes(h)
        for key, shape in self.shapes.items():           # flat parameter registry under the reference's dotted names
            self.register_buffer(key.replace(".", "__"), torch.zeros(shape), persistent=False)
        self._ctx: Optional[C.c_void_p] = None
        self._lib = None
        self._loaded_key = None
        self._ws = None

    # ---- checkpoint surface ---------------------------------------------------------------------------------------
    def state_dict(self, *a, **k):
        return {key: getattr(self, key.replace(".", "__")) for key in self.shapes}

    def load_state_dict(self, sd, strict: bool = True):
        sd = fold_weight_norm(dict(sd))
        missing = [k for k in self.shapes if k not in sd]
        extra = [k for k in sd if k not in self.shapes]
        if strict and (missing or extra):
            raise RuntimeError(f"Generator.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for key, shape in self.shapes.items():
            if key in sd:
                t = sd[key].detach().to(torch.float32)
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"{key}: shape {tuple(t.shape)} != {tuple(shape)}")
                getattr(self, key.replace(".", "__")).copy_(t)
        self._loaded_key = None
        return self

    def remove_weight_norm(self):
        """No-op: weight norm is folded at load time (the reference calls this right after loading, utils.py:278)."""
        return self

    # ---- engine -----------------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.dex_voc_last_error(self._ctx)
            raise RuntimeError(f"libdexamd vocoder error {rc}: {msg.decode() if msg else '?'}")

    def _engine(self, device):
        if device.type != "cuda":
            raise RuntimeError("dex_tts_amd runs on an AMD GPU (torch device 'cuda' on ROCm); no CPU path exists")
        if self._ctx is None:
            self._lib = _lib.load()
            c = _lib.DexVocoderConfig()
            h = self.h
            rates, ksz, rk, rd = (list(_get(h, n)) for n in ("upsample_rates", "upsample_kernel_sizes", "resblock_kernel_sizes", "resblock_dilation_sizes"))
            c.num_mels, c.upsample_initial_channel, c.n_upsamples = int(_get(h, "num_mels", 80)), int(_get(h, "upsample_initial_channel")), len(rates)
            for i, (u, k) in enumerate(zip(rates, ksz)):
                c.upsample_rates[i], c.upsample_kernel_sizes[i] = int(u), int(k)
            c.activation, c.snake_logscale = ACTIVATION[_get(h, "activation")], int(bool(_get(h, "snake_logscale", False)))
            c.n_resblock_kernels = len(rk)
            for j, k in enumerate(rk[:3]):
                c.resblock_kernel_sizes[j] = int(k)
                for m in range(3):
                    c.resblock_dilation_sizes[j][m] = int(rd[j][m])
            ctx = C.c_void_p()
            rc = self._lib.dex_voc_create(C.byref(c), C.byref(ctx))
            self._ctx = ctx
            self._check(rc)
        bufs = [getattr(self, k.replace(".", "__")) for k in self.shapes]
        key = (str(device),) + tuple((b._version, b.data_ptr()) for b in bufs)
        if key != self._loaded_key:
            # BigVGAN: the library takes ONE copy of the resampling filter (all 73 buffers are the same Kaiser-sinc constant)
            filt = [k for k in self.shapes if k.endswith(".filter")]
            if filt:
                f0 = getattr(self, "activation_post.upsample.filter".replace(".", "__"))
                for k in filt:
                    if not torch.equal(getattr(self, k.replace(".", "__")), f0):
                        raise RuntimeError(f"{k} differs from activation_post.upsample.filter: per-layer resampling filters are not supported")
            with torch.cuda.device(device):
                st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                keep = []
                for name, shape in self.shapes.items():
                    if name.endswith(".filter") and not name.startswith("activation_post."):
                        continue
                    w = getattr(self, name.replace(".", "__")).to(device=device, dtype=torch.float32).contiguous()
                    shp = (C.c_int64 * 4)(*([int(s) for s in shape] + [0] * (4 - len(shape))))
                    self._check(self._lib.dex_voc_load_weight_async(self._ctx, name.encode(), C.c_void_p(w.data_ptr()), shp, len(shape), st))
                    keep.append(w)
                self._check(self._lib.dex_voc_finalize(self._ctx, st))
            self._loaded_key = key

    def __del__(self):
        try:
            if self._ctx is not None and self._ctx.value:
                self._lib.dex_voc_destroy(self._ctx)
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """mel [B, num_mels, T] -> wav [B, 1, T * prod(upsample_rates)] (models.py:150-167)."""
        dev = x.device
        self._engine(dev)
        with torch.cuda.device(dev):
            # operand precision of the convolutions: 'fp32' (default, the parity mode), 'bf16' or 'fp16' (attribute ``precision``)
            self._check(self._lib.dex_voc_set_precision(self._ctx, _lib.PRECISION[getattr(self, "precision", "fp32")]))
            mel = x.to(dtype=torch.float32).contiguous()
            B, M, T = mel.shape
            if M != int(_get(self.h, "num_mels", 80)):
                raise ValueError(f"mel has {M} channels, the generator expects {_get(self.h, 'num_mels', 80)}")
            n = int(self._lib.dex_voc_samples(self._ctx, T))
            need = int(self._lib.dex_voc_workspace_bytes(self._ctx, B, T))
            if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
                self._ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            base = (self._ws.data_ptr() + 255) // 256 * 256
            wav = torch.empty(B, 1, n, dtype=torch.float32, device=dev)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            self._check(self._lib.dex_vocode(self._ctx, C.c_void_p(mel.data_ptr()), B, T, C.c_void_p(wav.data_ptr()), C.c_void_p(base),
                                             self._ws.numel() - (base - self._ws.data_ptr()), st))
            self._keep = mel
            return wav


def get_vocoder(config_path: Optional[str] = None, ckpt: Optional[dict] = None, device="cuda") -> Generator:
    """src/utils.py:251-281 for the 'hifigan' choice: config.json -> Generator -> load ckpt['generator'] -> eval ->
    remove_weight_norm -> device."""
    h = AttrDict(json.load(open(config_path))) if config_path else AttrDict(HIFIGAN_V1)      # a BigVGAN config.json selects BigVGAN
    g = Generator(h)
    if ckpt is not None:
        g.load_state_dict(ckpt["generator"] if "generator" in ckpt else ckpt)
    return g.eval().to(device)


BigVGAN = Generator      # bigvgan/__init__.py: ``from .models import BigVGAN as Generator`` — Generator(AttrDict(BIGVGAN_BASE))
