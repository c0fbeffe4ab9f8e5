"""Counterpart of the reference's synthesize flow around the decoder (GeDEX-TTS/synthesize.py:15-45 ->
GeDEXTTS.forward, model/tts.py:34-55; DEX-TTS/synthesize.py:88-110 -> DeXTTS.forward, model/tts.py:53-73), for the part
this library owns: everything from the aligned text-encoder output ``mu_y`` to the mel.

    seed_init(seed)                                   src/utils.py:94-103 (torch / numpy / random seeds)
    model = Diffusion(**cfg.model.decoder, dit_cfg=cfg.model.dit, n_feats, n_spks, spk_emb_dim)      tts.py:25 / :30
    model.load_state_dict(ckpt['ema' | 'state_dict'] entries under 'decoder.')                         synthesize.py:20-24
    y_max_length_ = fix_len_compatibility(y_max_length)                                              tts.py:40
    y_mask = sequence_mask(y_lengths, y_max_length_).unsqueeze(1)                                    tts.py:43
    dec_out = decoder(mu_y, y_mask, mu_y, [ref, ref_lengths, sty, sty_lengths,] n_timesteps, temperature, spk, infer=True)
    dec_out = dec_out[:, :, :y_max_length]                                                            tts.py:54

The text encoder / duration model / vocoder are not part of this library: ``mu_y`` comes in as an
array (e.g. dumped from the reference, or synthetic).  CLI:

    python -m dex_tts_amd.synthesize --config <reference base.yaml> --mu mu_y.npy [--lengths 210,180] [--ckpt model.pth]
           [--n_timesteps 50] [--temperature 1.5] [--seed 100] [--precision fp32|bf16|fp16] --out mel.npy
"""
from __future__ import annotations

import argparse
import random
from typing import Optional, Sequence

import numpy as np
import torch

from .config import ScoreNetConfig, fix_len_compatibility, from_reference_yaml
from .diffusion import Diffusion, from_config


def seed_init(seed: int = 100):
    """src/utils.py:94-103."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def sequence_mask(length: torch.Tensor, max_length: Optional[int] = None) -> torch.Tensor:
    """model/utils.py:6-10."""
    if max_length is None:
        max_length = int(length.max())
    x = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def config_from_model_section(model: dict, variant: str) -> ScoreNetConfig:
    """``cfg.model`` of a reference base.yaml -> ScoreNetConfig (tts.py:25: Diffusion(n_feats, **decoder, dit_cfg=dit,
    n_spks=..., spk_emb_dim=...)); DEX configs carry n_spks = 0, which Diffusion treats like 1 (no speaker plane)."""
    return from_reference_yaml(model["decoder"], model["dit"], variant, n_spks=max(int(model.get("n_spks") or 1), 1),
                               spk_emb_dim=int(model.get("spk_emb_dim", 64)), n_feats=int(model.get("n_feats", 80)))


def load_reference_config(path: str) -> ScoreNetConfig:
    import yaml
    with open(path) as f:
        model = yaml.safe_load(f)["model"]
    variant = "dex" if "tv_encoder" in model else "gedex"
    return config_from_model_section(model, variant)


def decoder_state_dict(ckpt: dict, ema: bool = True) -> dict:
    """The decoder's entries of a reference checkpoint (synthesize.py:20-24), without the 'decoder.' prefix."""
    sd = ckpt["ema"] if (ema and "ema" in ckpt) else ckpt.get("state_dict", ckpt)
    return {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}


def prepare(mu_y: torch.Tensor, y_lengths: torch.Tensor, n_stages: int = 2):
    """tts.py:39-43,49: pad mu_y to the U-Net-compatible length and build y_mask.  Returns (mu_y_padded, y_mask [B,1,T'],
    y_max_length)."""
    y_max_length = int(y_lengths.max())
    t_pad = fix_len_compatibility(y_max_length, n_stages)
    B, F, T = mu_y.shape
    if T < t_pad:
        mu_y = torch.nn.functional.pad(mu_y, (0, t_pad - T))
    mu_y = mu_y[:, :, :t_pad]
    y_mask = sequence_mask(y_lengths, t_pad).unsqueeze(1).to(mu_y.dtype)
    return mu_y * y_mask, y_mask, y_max_length              # the aligned mu_y is zero past each utterance's length (attn path is masked)


@torch.no_grad()
def decode(model: Diffusion, mu_y: torch.Tensor, y_lengths: torch.Tensor, n_timesteps: int = 50, temperature: float = 1.5,
           spk: Optional[torch.Tensor] = None, ref: Optional[Sequence[torch.Tensor]] = None, ref_lengths=None,
           sty: Optional[torch.Tensor] = None, sty_lengths=None) -> torch.Tensor:
    """mu_y [B,80,T] (aligned encoder output) -> mel [B,80,y_max_length], exactly the decoder leg of GeDEXTTS.forward /
    DeXTTS.forward (tts.py:49-55 / :66-73)."""
    mu_p, y_mask, y_max = prepare(mu_y, y_lengths, len(model.cfg.dim_mults))
    if model.cfg.variant == "dex":
        out = model(mu_p, y_mask, mu_p, ref, ref_lengths, sty, sty_lengths, n_timesteps=n_timesteps, temperature=temperature,
                    spk=spk, infer=True)
    else:
        out = model(mu_p, y_mask, mu_p, n_timesteps=n_timesteps, temperature=temperature, spk=spk, infer=True)
    return out[:, :, :y_max]


MAX_VALUE = 32768.0          # synthesize.py:13-14


@torch.no_grad()
def synthesize_tokens(model, vocoder, x: torch.Tensor, x_lengths: torch.Tensor, n_timesteps: int = 50, temperature: float = 1.5,
                      spk: Optional[torch.Tensor] = None, length_scale: float = 1.0, style: Optional[dict] = None):
    """synthesize.py:31-38 from the token sequence on (the text front-end - cleaners, CMU dictionary, ``intersperse`` - is the
    reference's and stays on the host): ``model`` a ``dex_tts_amd.tts.GeDEXTTS`` / ``DeXTTS``, ``vocoder`` a
    ``dex_tts_amd.vocoder.Generator``; DEX takes ``style = dict(ref, ref_lengths, sty, sty_lengths, lf0, lf0_lengths)``
    (DEX-TTS/synthesize.py:47-95).  Returns (list of int16 waveforms cut to each utterance's length, y_dec [B,80,Ty], attn)."""
    if style is not None:
        y_enc, y_dec, attn = model(x, x_lengths, style["ref"], style["ref_lengths"], style["sty"], style["sty_lengths"], style["lf0"],
                                   style["lf0_lengths"], n_timesteps=n_timesteps, temperature=temperature, spk=spk, length_scale=length_scale)
    else:
        y_enc, y_dec, attn = model(x, x_lengths, n_timesteps=n_timesteps, temperature=temperature, spk=spk, length_scale=length_scale)
    wav = vocoder(y_dec).squeeze(1).clamp(-1, 1)                                          # [B, Ty * hop]
    hop = wav.shape[-1] // y_dec.shape[-1]
    y_len = model.encoder._last["y_len"].to(torch.int64).cpu()
    audio = (wav.cpu().numpy() * MAX_VALUE).astype(np.int16)
    return [audio[b, : int(y_len[b]) * hop] for b in range(audio.shape[0])], y_dec, attn


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", required=True, help="a reference base.yaml (GeDEX-TTS/config/*/base.yaml, DEX-TTS/config/*/base.yaml)")
    ap.add_argument("--mu", required=True, help=".npy with mu_y [B,80,T] (or [80,T])")
    ap.add_argument("--lengths", default=None, help="comma-separated valid frame counts (default: T for every utterance)")
    ap.add_argument("--ckpt", default=None, help="reference checkpoint (model-train-best.pth); default: portable synthetic weights")
    ap.add_argument("--style", default=None, help="DEX: .npz with ref (6 x [B,mid,Tr]), ref_lengths, sty [B,mid,Ts], sty_lengths")
    ap.add_argument("--n_timesteps", type=int, default=50)
    ap.add_argument("--temperature", type=float, default=1.5)
    ap.add_argument("--seed", type=int, default=100)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)

    seed_init(a.seed)
    cfg = load_reference_config(a.config)
    model = from_config(cfg)
    if a.ckpt:
        sd = decoder_state_dict(torch.load(a.ckpt, map_location="cpu"))
        if not sd:
            raise SystemExit(f"{a.ckpt}: no 'decoder.*' entries — not a DEX-TTS / GeDEX-TTS model checkpoint")
        model.load_state_dict(sd, strict=True)             # as the reference does (synthesize.py:20-24): a partial load must not pass silently
    else:
        from . import synth
        from .config import param_shapes
        w = synth.make_weights(param_shapes(cfg))
        sd = {}
        for k, v in w.items():
            sd[f"denoise_fn.{k}"] = torch.from_numpy(v)
            sd[f"precond_model.model.{k}"] = torch.from_numpy(v)
        model.load_state_dict(sd, strict=True)
    model = model.to(a.device).eval()
    model.precision = a.precision
    mu = torch.from_numpy(np.load(a.mu).astype(np.float32))
    if mu.dim() == 2:
        mu = mu[None]
    lengths = torch.tensor([int(v) for v in a.lengths.split(",")] if a.lengths else [mu.shape[2]] * mu.shape[0])
    kw = {}
    if cfg.variant == "dex":
        if not a.style:
            raise SystemExit("DEX configs need --style (reference-wav encoder outputs)")
        s = np.load(a.style)
        kw = dict(ref=[torch.from_numpy(r).to(a.device) for r in s["ref"]], ref_lengths=torch.from_numpy(s["ref_lengths"]).to(a.device),
                  sty=torch.from_numpy(s["sty"]).to(a.device), sty_lengths=torch.from_numpy(s["sty_lengths"]).to(a.device))
    mel = decode(model, mu.to(a.device), lengths.to(a.device), a.n_timesteps, a.temperature, **kw)
    np.save(a.out, mel.cpu().numpy())
    print(f"{a.out}: mel {tuple(mel.shape)}")


if __name__ == "__main__":
    main()
