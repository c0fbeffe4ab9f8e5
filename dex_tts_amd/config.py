"""Score-network configuration and parameter inventory.

Mirrors the YAML keys the reference feeds to ``Diffusion(**cfg.decoder, dit_cfg=cfg.dit)``
(reference: GeDEX-TTS/config/LJSpeech/base.yaml:41-62, DEX-TTS/config/VCTK/base.yaml:63-85,
GeDEX-TTS/model/diffusion.py:119-166, GeDEX-TTS/model/dit.py:339-402).

``param_shapes`` enumerates the state-dict keys (relative to ``denoise_fn.``) and shapes of the
score network, so the drop-in module can expose the same checkpoint surface without any of the
reference's module classes.  It is validated against manifests dumped from the real reference
(tests/golden/manifest_*.json).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, Tuple, List


@dataclass
class DiTConfig:
    patch_size: int = 7
    stride_size: int = 4
    hidden_size: int = 256
    depth: int = 4
    num_heads: int = 2
    mlp_ratio: float = 2
    conv_pos: int = 16
    conv_pos_groups: int = 8
    overlap: bool = True
    use_decoder: bool = False
    # accepted-and-ignored reference keys (dit.py:349-354): in_channels/out_channels are
    # overwritten with mid_dim at construction (diffusion.py:151-152); mask_type is training only.
    in_channels: int = 3
    out_channels: int = 1
    mask_type: str = "time_random"


@dataclass
class ScoreNetConfig:
    variant: str = "gedex"          # "gedex" (text->mel) | "dex" (adds TV/TIV style adaptors)
    n_feats: int = 80
    dim: int = 64
    dim_mults: Tuple[int, ...] = (1, 2)
    n_spks: int = 1
    spk_emb_dim: int = 64
    pe_scale: float = 1000.0
    groups: int = 8                 # GroupNorm groups (diffusion.py:42)
    lin_heads: int = 4              # LinearAttention heads / dim_head (diffusion.py:75)
    lin_dim_head: int = 32
    dit: DiTConfig = field(default_factory=DiTConfig)

    # ---- derived geometry -------------------------------------------------------------
    @property
    def in_planes(self) -> int:
        return 2 + (1 if self.n_spks > 1 else 0)

    @property
    def stage_dims(self) -> List[int]:
        return [self.dim * m for m in self.dim_mults]

    @property
    def mid_dim(self) -> int:
        return self.stage_dims[-1]

    @property
    def mid_h(self) -> int:
        return self.n_feats // (2 ** (len(self.dim_mults) - 1))

    @property
    def grid_h(self) -> int:        # DiT token rows (dit.py:51 with img_size[0] = mid_h)
        return self.mid_h // self.dit.stride_size

    def token_cols(self, w_mid: int) -> int:
        """Token columns for a bottleneck of width w_mid (dit.py:442-447 + conv arithmetic of :57)."""
        p, s = self.dit.patch_size, self.dit.stride_size
        wp = w_mid if w_mid % p == 0 else w_mid + (p - w_mid % p)
        return (wp + 2 * (p // 2) - p) // s + 1

    def token_rows(self) -> int:
        p, s = self.dit.patch_size, self.dit.stride_size
        return (self.mid_h + 2 * (p // 2) - p) // s + 1

    def to_dict(self):
        d = asdict(self)
        d["dim_mults"] = list(self.dim_mults)
        return d


def gedex_lj() -> ScoreNetConfig:
    return ScoreNetConfig(variant="gedex", dit=DiTConfig(patch_size=7, stride_size=4))


def gedex_vctk() -> ScoreNetConfig:
    return ScoreNetConfig(variant="gedex", n_spks=108, dit=DiTConfig(patch_size=7, stride_size=4))


def dex_vctk() -> ScoreNetConfig:
    return ScoreNetConfig(variant="dex", dit=DiTConfig(patch_size=3, stride_size=2))


def dex_libritts() -> ScoreNetConfig:
    """DEX-TTS/config/LibriTTS/base.yaml:63-85: dim 128 (stages 128 / 256), DiT hidden 384 = 2 x 192."""
    return ScoreNetConfig(variant="dex", dim=128, dit=DiTConfig(patch_size=3, stride_size=2, hidden_size=384))


PRESETS = {"gedex_lj": gedex_lj, "gedex_vctk": gedex_vctk, "dex_vctk": dex_vctk, "dex_esd": dex_vctk, "dex_libritts": dex_libritts}


def from_reference_yaml(decoder: dict, dit: dict, variant: str, n_spks: int = 1,
                        spk_emb_dim: int = 64, n_feats: int = 80) -> ScoreNetConfig:
    """Build a config from the reference's ``model.decoder`` / ``model.dit`` YAML dicts."""
    known = {k: dit[k] for k in DiTConfig.__dataclass_fields__ if k in dit}
    return ScoreNetConfig(variant=variant, n_feats=n_feats, dim=int(decoder["dim"]),
                          dim_mults=tuple(decoder["dim_mults"]), n_spks=int(n_spks),
                          spk_emb_dim=int(spk_emb_dim), pe_scale=float(decoder.get("pe_scale", 1000)),
                          dit=DiTConfig(**known))


# ----------------------------------------------------------------------------------------------
def _resnet(prefix: str, cin: int, cout: int, tdim: int, out: Dict[str, Tuple[int, ...]]):
    out[f"{prefix}.mlp.1.weight"] = (cout, tdim)
    out[f"{prefix}.mlp.1.bias"] = (cout,)
    for blk, ci in (("block1", cin), ("block2", cout)):
        out[f"{prefix}.{blk}.block.0.weight"] = (cout, ci, 3, 3)
        out[f"{prefix}.{blk}.block.0.bias"] = (cout,)
        out[f"{prefix}.{blk}.block.1.weight"] = (cout,)
        out[f"{prefix}.{blk}.block.1.bias"] = (cout,)
    if cin != cout:
        out[f"{prefix}.res_conv.weight"] = (cout, cin, 1, 1)
        out[f"{prefix}.res_conv.bias"] = (cout,)


def _linattn(prefix: str, c: int, cfg: ScoreNetConfig, out):
    hid = cfg.lin_heads * cfg.lin_dim_head
    out[f"{prefix}.fn.g"] = (1,)
    out[f"{prefix}.fn.fn.to_qkv.weight"] = (3 * hid, c, 1, 1)
    out[f"{prefix}.fn.fn.to_out.weight"] = (c, hid, 1, 1)
    out[f"{prefix}.fn.fn.to_out.bias"] = (c,)


def param_shapes(cfg: ScoreNetConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict keys (relative to ``denoise_fn.``) → shapes, in registration order."""
    d, out = cfg.dim, {}
    out["mlp.0.weight"] = (4 * d, d); out["mlp.0.bias"] = (4 * d,)
    out["mlp.2.weight"] = (d, 4 * d); out["mlp.2.bias"] = (d,)
    if cfg.variant == "dex":
        for name in ("mlp_adap", "mlp_adap_sty"):
            out[f"{name}.0.weight"] = (d, d); out[f"{name}.0.bias"] = (d,)
            out[f"{name}.2.weight"] = (2 * d, d); out[f"{name}.2.bias"] = (2 * d,)
    if cfg.n_spks > 1:
        e = cfg.spk_emb_dim
        out["spk_mlp.0.weight"] = (4 * e, e); out["spk_mlp.0.bias"] = (4 * e,)
        out["spk_mlp.2.weight"] = (cfg.n_feats, 4 * e); out["spk_mlp.2.bias"] = (cfg.n_feats,)
    dims = [cfg.in_planes] + cfg.stage_dims
    in_out = list(zip(dims[:-1], dims[1:]))
    for i, (ci, co) in enumerate(in_out):
        _resnet(f"downs.{i}.0", ci, co, d, out)
        _resnet(f"downs.{i}.1", co, co, d, out)
        _linattn(f"downs.{i}.2", co, cfg, out)
        if i < len(in_out) - 1:
            out[f"downs.{i}.3.conv.weight"] = (co, co, 3, 3)
            out[f"downs.{i}.3.conv.bias"] = (co,)
    # DiT bottleneck (dit.py:369-401)
    t, hid, mid = cfg.dit, cfg.dit.hidden_size, cfg.mid_dim
    out["vit.freq_new_pos_embed"] = (1, hid, cfg.grid_h, 1)
    out["vit.x_embedder.proj.0.weight"] = (mid, 1, t.patch_size, t.patch_size)
    out["vit.x_embedder.proj.0.bias"] = (mid,)
    out["vit.x_embedder.proj.2.weight"] = (hid, mid, 1, 1)
    out["vit.x_embedder.proj.2.bias"] = (hid,)
    out["vit.t_embedder.mlp.0.weight"] = (hid, 256); out["vit.t_embedder.mlp.0.bias"] = (hid,)
    out["vit.t_embedder.mlp.2.weight"] = (hid, hid); out["vit.t_embedder.mlp.2.bias"] = (hid,)
    out["vit.pos_conv.0.weight"] = (hid, hid // t.conv_pos_groups, t.conv_pos, t.conv_pos)
    out["vit.pos_conv.0.bias"] = (hid,)
    mh = int(hid * t.mlp_ratio)
    for k in range(t.depth):
        p = f"vit.blocks.{k}"
        out[f"{p}.attn.qkv.weight"] = (3 * hid, hid); out[f"{p}.attn.qkv.bias"] = (3 * hid,)
        out[f"{p}.attn.proj.weight"] = (hid, hid); out[f"{p}.attn.proj.bias"] = (hid,)
        out[f"{p}.mlp.fc1.weight"] = (mh, hid); out[f"{p}.mlp.fc1.bias"] = (mh,)
        out[f"{p}.mlp.fc2.weight"] = (hid, mh); out[f"{p}.mlp.fc2.bias"] = (hid,)
        out[f"{p}.adaLN_modulation.1.weight"] = (6 * hid, hid)
        out[f"{p}.adaLN_modulation.1.bias"] = (6 * hid,)
    out["vit.final_layer.linear.weight"] = (t.stride_size ** 2 * mid, hid)
    out["vit.final_layer.linear.bias"] = (t.stride_size ** 2 * mid,)
    out["vit.final_layer.adaLN_modulation.1.weight"] = (2 * hid, hid)
    out["vit.final_layer.adaLN_modulation.1.bias"] = (2 * hid,)
    if cfg.variant == "dex":
        for w in ("w_q", "w_k", "w_v", "linear"):
            out[f"tv_adaptor.{w}.weight"] = (mid, mid)
        for s in ("mean_sap", "std_sap"):
            out[f"tiv_adaptor.{s}.W.weight"] = (1, mid)
            out[f"tiv_adaptor.{s}.W.bias"] = (1,)
    for j, (ci, co) in enumerate(reversed(in_out[1:])):
        _resnet(f"ups.{j}.0", co * 2, ci, d, out)
        _resnet(f"ups.{j}.1", ci, ci, d, out)
        _linattn(f"ups.{j}.2", ci, cfg, out)
        out[f"ups.{j}.3.conv.weight"] = (ci, ci, 4, 4)
        out[f"ups.{j}.3.conv.bias"] = (ci,)
    out["final_block.block.0.weight"] = (d, d, 3, 3); out["final_block.block.0.bias"] = (d,)
    out["final_block.block.1.weight"] = (d,); out["final_block.block.1.bias"] = (d,)
    out["final_conv.weight"] = (1, d, 1, 1); out["final_conv.bias"] = (1,)
    return out


def fix_len_compatibility(length: int, num_downsamplings: int = 2) -> int:
    """Pad a frame count up to a multiple of 2**n (reference model/utils.py:13-17)."""
    q = 2 ** num_downsamplings
    return ((int(length) + q - 1) // q) * q
