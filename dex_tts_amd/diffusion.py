"""Drop-in replacement for the reference ``model.diffusion.Diffusion`` inference path.

Same constructor arguments, same ``forward`` signatures, same checkpoint keys (every denoiser tensor
appears under ``denoise_fn.*`` and ``precond_model.model.*`` — reference diffusion.py:213-214), same
RNG draws.  All arithmetic of ``forward(..., infer=True)`` runs in libdexamd.so (hand-written gfx950
kernels); PyTorch only owns the tensors and the stream.

    GeDEX: Diffusion.forward(x, mask, mu, n_timesteps, spk, infer, temperature, mask_ratio)
           — GeDEX-TTS/model/diffusion.py:220-229
    DEX:   Diffusion.forward(x, mask, mu, ref, ref_lengths, sty, sty_lengths, n_timesteps, spk, infer,
           temperature, mask_ratio) — DEX-TTS/model/diffusion.py:250-259
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

from .config import DiTConfig, ScoreNetConfig, param_shapes
from .edm import EDMLoss
from .engine import ScoreNetEngine


class _Params(nn.Module):
    """Parameter container reproducing the reference's module-tree names (no forward)."""

    def __init__(self):
        super().__init__()


def _build_tree(shapes: Dict[str, tuple]) -> _Params:
    root = _Params()
    for key, shape in shapes.items():
        parts = key.split(".")
        node = root
        for name in parts[:-1]:
            if name not in node._modules:
                node.add_module(name, _Params())
            node = node._modules[name]
        zero = (key.endswith(".fn.g") or ".adaLN_modulation.1." in key or key.startswith("vit.final_layer.linear")
                or key == "vit.freq_new_pos_embed" or key.endswith(".bias"))
        if zero:      # reference zero-inits (dit.py:404-413, diffusion.py:35); biases start at 0 here
            t = torch.zeros(shape)
        elif key.endswith("block.1.weight"):
            t = torch.ones(shape)
        else:
            fan_in = int(math.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
            t = torch.empty(shape).uniform_(-1.0, 1.0) * (1.0 / math.sqrt(max(fan_in, 1)))
        node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
    return root


class _Precond(nn.Module):
    """Holds ``model`` so that state_dict exposes ``precond_model.model.*`` (edm.py:74-86)."""

    def __init__(self, model: nn.Module, sigma_data: float = 0.5):
        super().__init__()
        self.model = model
        self.sigma_min, self.sigma_max, self.sigma_data = 0, float("inf"), sigma_data
        self._owner = None           # weakref to the owning Diffusion, set by it (and re-set on deepcopy / unpickle)

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_owner"] = None                   # (weakrefs do not pickle: the owner binds again in its __setstate__ / before every use)
        return st

    def _denoise_once(self, x, sigma, mask, mu, **kw):
        """One EDMPrecond.forward on the OWNER's engine (EDMLoss): resolved at call time, so a deep copy of the model evaluates with
        its own weights and engine, and the module holds no closure (torch.save / pickle work, no reference cycle)."""
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("precond_model has no live owner (it belongs to a dex_tts_amd.diffusion.Diffusion)")
        return owner.engine(x.device).denoise_once(x, sigma, mask, mu, **kw)


def _cfg_get(obj, name, default):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


class Diffusion(nn.Module):
    def __init__(self, n_feats, dim, dit_cfg, loss_type="base", precond="edm", model_type="dit", dim_mults=(1, 2),
                 n_spks=1, spk_emb_dim=64, pe_scale=1000, variant="gedex"):
        super().__init__()
        if model_type != "dit" or precond != "edm":
            raise ValueError("only model_type='dit' / precond='edm' exist in the reference configs")
        d = DiTConfig()
        dit = DiTConfig(**{k: _cfg_get(dit_cfg, k, getattr(d, k)) for k in DiTConfig.__dataclass_fields__})
        self.cfg = ScoreNetConfig(variant=variant, n_feats=n_feats, dim=dim, dim_mults=tuple(dim_mults),
                                  n_spks=n_spks if n_spks is not None else 1, spk_emb_dim=spk_emb_dim,
                                  pe_scale=float(pe_scale), dit=dit)
        self.loss_type = loss_type
        self.denoise_fn = _build_tree(param_shapes(self.cfg))
        self.precond_model = _Precond(self.denoise_fn)
        self.loss_fn = EDMLoss(n_feats=n_feats, loss_type=loss_type)       # diffusion.py:215: forward-only here (dex_tts_amd/edm.py)
        self._bind_owner()
        self.precision = "fp32"
        self.use_graph = False
        self.solver = "euler"                    # the reference wires 'euler' (diffusion.py:216); 'heun' = edm.py:207-214
        self.rng_parity = True                   # replay the reference's per-step randn_like draws (edm.py:196)
        # ablation_sampler's stochastic settings (edm.py:109,194-196); the reference wires the defaults (S_churn = 0)
        self.S_churn, self.S_min, self.S_max, self.S_noise = 0.0, 0.0, float("inf"), 1.0
        self._engine = None
        self._engine_key = None
        self._make_sampler()

    def _bind_owner(self):
        import weakref
        self.precond_model._owner = weakref.ref(self)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_engine", "_engine_key", "sampler"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._engine, new._engine_key = None, None
        new._make_sampler()
        new._bind_owner()
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"], st["_engine_key"] = None, None
        st.pop("sampler", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._make_sampler()
        self._bind_owner()

    def _make_sampler(self):
        if self.cfg.variant == "dex":
            self.sampler = lambda z, mask, mu, ref, ref_lengths, sty, sty_lengths, spk, steps: self._sample(
                z, mask, mu, steps, spk, ref, sty, sty_lengths)
        else:
            self.sampler = lambda z, mask, mu, spk, steps: self._sample(z, mask, mu, steps, spk)

    # ---- engine management ---------------------------------------------------------------------
    def _weights(self):
        return {k: v for k, v in self.denoise_fn.state_dict().items()}

    def engine(self, device: torch.device) -> ScoreNetEngine:
        params = list(self.denoise_fn.parameters())
        key = (str(device), tuple(p._version for p in params), tuple(p.data_ptr() for p in params), self.precision)
        if self._engine is None or self._engine.device != device:
            self._engine = ScoreNetEngine(self.cfg, device)
            self._engine_key = None
        if key != self._engine_key:
            self._engine.load_weights(self._weights())
            self._engine.set_precision(self.precision)
            self._engine_key = key
        return self._engine

    def _sample(self, z, mask, mu, steps, spk=None, ref=None, sty=None, sty_lengths=None):
        eng = self.engine(z.device)
        noise = None
        if self.S_churn > 0:
            # the draws the reference makes inside its loop, one randn_like(x_cur) per step (edm.py:196), in that order
            noise = torch.stack([torch.randn_like(z) for _ in range(int(steps))])
        return eng.sample(z, mask, mu, int(steps), spk=spk, ref=ref, sty=sty, sty_lengths=sty_lengths,
                          use_graph=self.use_graph, solver=self.solver, noise=noise, S_churn=self.S_churn, S_min=self.S_min,
                          S_max=self.S_max, S_noise=self.S_noise)

    def _advance_rng(self, like: torch.Tensor, n: int):
        """The reference draws ``randn_like(x_cur)`` once per Euler step and multiplies it by 0
        (edm.py:196); only the generator state matters.  Advance the Philox offset by the same amount."""
        if not self.rng_parity or n <= 0 or self.S_churn > 0:      # (with S_churn > 0 the draws were really made, in _sample)
            return
        gen = torch.cuda.default_generators[like.device.index if like.device.index is not None else torch.cuda.current_device()]
        if n > 1 and hasattr(gen, "get_offset") and hasattr(gen, "set_offset"):
            o0 = gen.get_offset()
            torch.randn_like(like)                       # one real draw measures the per-draw offset increment
            gen.set_offset(o0 + n * (gen.get_offset() - o0))
        else:                                            # no offset API: draw exactly n times like the reference
            for _ in range(n):
                torch.randn_like(like)

    # ---- reference call surface ----------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, mask, mu, *args, n_timesteps=1, spk=None, infer=False, temperature=1.0, mask_ratio=0, **kw):
        names = ["ref", "ref_lengths", "sty", "sty_lengths"] if self.cfg.variant == "dex" else []
        names += ["n_timesteps", "spk", "infer", "temperature", "mask_ratio"]
        vals = {"n_timesteps": n_timesteps, "spk": spk, "infer": infer, "temperature": temperature, "mask_ratio": mask_ratio}
        vals.update(kw)
        if len(args) > len(names):
            raise TypeError("too many positional arguments")
        for n, v in zip(names, args):
            vals[n] = v
        if not vals["infer"]:                     # diffusion.py:222-224 / :252-254: the EDM training-loss VALUE (no backward on this path)
            dex = (vals["ref"], vals["ref_lengths"], vals["sty"], vals["sty_lengths"]) if self.cfg.variant == "dex" else ()
            self._bind_owner()                    # (an unpickled module restores its sub-modules' state after its own)
            return self.loss_fn(self.precond_model, x, mask, mu, *dex, spk=vals["spk"], mask_ratio=vals["mask_ratio"])
        shape = (mu.shape[0], 80, mu.shape[2])
        z = torch.randn(shape, device=x.device) / vals["temperature"] + mu            # diffusion.py:227
        if self.cfg.variant == "dex":
            out = self.sampler(z, mask, mu, vals["ref"], vals["ref_lengths"], vals["sty"], vals["sty_lengths"],
                               vals["spk"], vals["n_timesteps"])
        else:
            out = self.sampler(z, mask, mu, vals["spk"], vals["n_timesteps"])
        self._advance_rng(z, int(vals["n_timesteps"]))
        return out


def from_config(cfg: ScoreNetConfig) -> Diffusion:
    t = cfg.dit
    dit = {k: getattr(t, k) for k in DiTConfig.__dataclass_fields__}
    return Diffusion(cfg.n_feats, cfg.dim, dit, dim_mults=cfg.dim_mults, n_spks=cfg.n_spks,
                     spk_emb_dim=cfg.spk_emb_dim, pe_scale=cfg.pe_scale, variant=cfg.variant)


class GeDEXDiffusion(Diffusion):
    """Exact constructor surface of GeDEX-TTS/model/diffusion.py:210."""

    def __init__(self, n_feats, dim, dit_cfg, loss_type="base", precond="edm", model_type="dit", dim_mults=(1, 2),
                 n_spks=1, spk_emb_dim=64, pe_scale=1000):
        super().__init__(n_feats, dim, dit_cfg, loss_type, precond, model_type, dim_mults, n_spks, spk_emb_dim,
                         pe_scale, variant="gedex")


class DEXDiffusion(Diffusion):
    """Exact constructor surface of DEX-TTS/model/diffusion.py:239 (its default model_type is 'vit', which
    builds no bottleneck at all in the reference; every shipped config passes 'dit')."""

    def __init__(self, n_feats, dim, dit_cfg, loss_type="base", precond="edm", model_type="dit", dim_mults=(1, 2),
                 n_spks=1, spk_emb_dim=64, pe_scale=1000):
        super().__init__(n_feats, dim, dit_cfg, loss_type, precond, model_type, dim_mults, n_spks, spk_emb_dim,
                         pe_scale, variant="dex")
