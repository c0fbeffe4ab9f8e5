"""ORACLE TOOLING — runs ONLY in the build container, where /root/reference exists.

Imports the real reference hot path (``model.diffusion.Diffusion`` and ``audio``) so that golden
vectors can be generated from it (oracle/make_golden.py).  Nothing from the reference is copied:
the reference modules are imported from where they lie.  Third-party packages the reference needs
but the image lacks are replaced by minimal stand-ins that restate their *published* semantics:

* ``timm.models.vision_transformer.{Attention, Mlp}`` (call sites dit.py:8,276,280; timm is unpinned
  in requirements.txt:16): qkv Linear(+bias) -> (B,N,3,H,hd) -> softmax(q*hd^-0.5 k^T) v -> proj;
  Mlp = fc1 -> act -> fc2.  State-dict names ``qkv/proj/fc1/fc2`` as in timm.
* ``librosa.util.{pad_center, tiny}``, ``librosa.filters.mel`` (librosa==0.9.2, requirements.txt:19).

These are the points where parity is "pinned by restated third-party semantics" (DESIGN.md §oracle).
"""
from __future__ import annotations

import importlib
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


class _Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, **_):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **_):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _install_timm_stub():
    if "timm" in sys.modules and hasattr(sys.modules["timm"], "_dex_stub"):
        return
    timm = types.ModuleType("timm"); timm._dex_stub = True
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Attention, vt.Mlp, vt.PatchEmbed = _Attention, _Mlp, object
    timm.models, models.vision_transformer = models, vt
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})


def _install_librosa_stub():
    from oracle.dex_oracle import slaney_mel_basis
    librosa = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    filters = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        lengths = [(0, 0)] * data.ndim
        lengths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, lengths, **kw)

    def tiny(x):
        x = np.asarray(x)
        dt = x.dtype if np.issubdtype(x.dtype, np.floating) else np.float32
        return np.finfo(dt).tiny

    def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_):
        return slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)

    util.pad_center, util.tiny, util.normalize = pad_center, tiny, (lambda x, **k: x)
    filters.mel = mel
    librosa.util, librosa.filters = util, filters
    sys.modules.update({"librosa": librosa, "librosa.util": util, "librosa.filters": filters})


def _purge(prefixes):
    for k in list(sys.modules):
        if any(k == p or k.startswith(p + ".") for p in prefixes):
            del sys.modules[k]


def import_reference(sub: str):
    """Return the reference's ``model.diffusion`` module for sub in {'GeDEX-TTS','DEX-TTS'}.

    ``model/__init__.py`` is bypassed (it pulls in tts.py -> a cp38 Cython .so and transformers-4.35
    APIs) by pre-registering an empty package whose __path__ is the reference directory."""
    sys.dont_write_bytecode = True
    _install_timm_stub()
    _purge(["model", "audio"])
    root = f"{REF_ROOT}/{sub}"
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    pkg = types.ModuleType("model"); pkg.__path__ = [f"{root}/model"]
    sys.modules["model"] = pkg
    return importlib.import_module("model.diffusion")


def import_reference_audio(sub: str = "DEX-TTS"):
    sys.dont_write_bytecode = True
    _install_librosa_stub()
    _purge(["audio"])
    root = f"{REF_ROOT}/{sub}"
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    torch.Tensor.cuda = lambda self, *a, **k: self          # stft.py:68-69 hard-codes .cuda()
    import audio  # noqa
    return importlib.import_module("audio.stft"), importlib.import_module("audio.tools")


def import_reference_text(sub: str):
    """The reference's ``model.text_encoder`` and ``model.utils`` modules (TextEncoder + RetNet, generate_path).

    Stand-ins for what the image lacks or has moved on from (requirements.txt pins transformers==4.35.2, timm unpinned):
    * ``timm.models.layers.drop_path`` (retention.py:9,403): stochastic depth; identity in eval mode, which is all that runs;
    * ``transformers.top_k_top_p_filtering`` (retention.py:12): imported, never called on this path; absent from transformers 5;
    * ``PretrainedConfig`` of transformers 5 no longer stores ``use_cache`` / ``output_hidden_states``: the attributes RetNetModel.forward
      reads (retnet.py:74-78) are set on the instance to the values transformers 4.35 would have stored (True / False / False)."""
    sys.dont_write_bytecode = True
    import transformers                                   # before the timm stand-in: its availability probe needs real import specs
    from transformers.modeling_utils import PreTrainedModel  # noqa: F401
    _install_timm_stub()
    timm = sys.modules["timm"]
    layers = types.ModuleType("timm.models.layers")

    def drop_path(x, drop_prob: float = 0.0, training: bool = False, scale_by_keep: bool = True):
        if drop_prob == 0.0 or not training:
            return x
        raise RuntimeError("drop_path stand-in: eval mode only")

    layers.drop_path = drop_path
    timm.models.layers = layers
    sys.modules["timm.models.layers"] = layers
    if not hasattr(transformers, "top_k_top_p_filtering"):
        transformers.top_k_top_p_filtering = None
    _purge(["model"])
    root = f"{REF_ROOT}/{sub}"
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    pkg = types.ModuleType("model"); pkg.__path__ = [f"{root}/model"]
    sys.modules["model"] = pkg
    return importlib.import_module("model.text_encoder"), importlib.import_module("model.utils")


def build_reference_text_encoder(sub: str, kwargs: dict, weights: dict = None):
    te, utils = import_reference_text(sub)
    enc = te.TextEncoder(**kwargs).eval()
    c = enc.encoder.config
    for k, v in dict(use_cache=True, output_retentions=False, output_hidden_states=False).items():
        if not hasattr(c, k):
            setattr(c, k, v)
    if weights is not None:
        enc.load_state_dict({k: torch.as_tensor(v) for k, v in weights.items()}, strict=True)
    return enc, utils


class AttrDict(dict):
    """dit_cfg must support attribute assignment (diffusion.py:151-152) and ** unpacking."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def build_reference_diffusion(cfg, weights: dict):
    """Instantiate the reference ``Diffusion`` for a ScoreNetConfig and load fixture weights into it."""
    sub = "DEX-TTS" if cfg.variant == "dex" else "GeDEX-TTS"
    mod = import_reference(sub)
    t = cfg.dit
    dit_cfg = AttrDict(in_channels=t.in_channels, patch_size=t.patch_size, stride_size=t.stride_size,
                       overlap=t.overlap, hidden_size=t.hidden_size, depth=t.depth, num_heads=t.num_heads,
                       mlp_ratio=t.mlp_ratio, out_channels=t.out_channels, conv_pos=t.conv_pos,
                       conv_pos_groups=t.conv_pos_groups, use_decoder=t.use_decoder, mask_type=t.mask_type)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = mod.Diffusion(n_feats=cfg.n_feats, dim=cfg.dim, dit_cfg=dit_cfg, dim_mults=tuple(cfg.dim_mults),
                          n_spks=cfg.n_spks, spk_emb_dim=cfg.spk_emb_dim, pe_scale=cfg.pe_scale,
                          model_type='dit', precond='edm', loss_type='base')
    sd = m.denoise_fn.state_dict()
    missing = set(sd) - set(weights)
    extra = set(weights) - set(sd)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    m.denoise_fn.load_state_dict({k: torch.as_tensor(v) for k, v in weights.items()}, strict=True)
    return m.eval()
