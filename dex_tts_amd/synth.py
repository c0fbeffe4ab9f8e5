"""Portable synthetic weights and inputs (numpy only, counter-based, bit-reproducible anywhere).

No trained checkpoints exist offline, and the reference's default init zeroes
every DiT adaLN/final projection and every ``Rezero.g`` (dit.py:404-413, diffusion.py:35), which
would make the DiT and linear-attention branches dead.  This generator therefore overwrites *all*
parameters with non-degenerate values derived from a splitmix64 stream keyed by the parameter name,
so the oracle container (where goldens are made from the real reference) and the GPU box (where the
HIP path is checked) regenerate identical tensors without shipping ~31 MB of weights.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1) from splitmix64(fnv1a(key) ^ seed-mix); element i depends only on i."""
    base = np.uint64((_fnv1a64(key) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def symmetric(key: str, shape, scale: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return ((uniform01(key, n, seed) * 2.0 - 1.0) * scale).astype(np.float32).reshape(shape)


def normalish(key: str, shape, seed: int = 0) -> np.ndarray:
    """Approximately N(0,1): sum of 4 uniforms, variance-normalised (portable, no Box-Muller libm)."""
    n = int(np.prod(shape))
    u = sum(uniform01(f"{key}#{j}", n, seed) for j in range(4))
    return ((u - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def make_weights(shapes: dict, seed: int = 0) -> dict:
    """name -> float32 ndarray for every entry of ``config.param_shapes``.

    >=2-D tensors: U(-a, a) with a = gain*sqrt(3 / fan_in) (gain 1, or 0.35 on the paths no norm follows);
    GroupNorm scales: 1 + 0.1 u;
    Rezero gates ``*.fn.g``: 0.4 + 0.1 u; every other 1-D tensor (biases): 0.1 u;
    ``vit.freq_new_pos_embed``: 0.1 u.
    """
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith("block.1.weight"):
            w = 1.0 + symmetric(name, shape, 0.1, seed)
        elif name.endswith(".fn.g"):
            w = 0.4 + symmetric(name, shape, 0.1, seed)
        elif name == "vit.freq_new_pos_embed":
            w = symmetric(name, shape, 0.1, seed)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            if name.endswith("to_qkv.weight") or name.endswith("res_conv.weight") or ".3.conv.weight" in name:
                gain = 0.35        # keeps q/k/v and the un-normalised shortcut / resampling paths O(1)
            w = symmetric(name, shape, gain * float(np.sqrt(3.0 / fan_in)), seed)
        else:
            w = symmetric(name, shape, 0.1, seed)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_inputs(B: int, T: int, lengths=None, seed: int = 1234, temperature: float = 1.5):
    """Synthetic (mu, mask, z) of SURVEY §8-d: mel-like prior clipped to [-11.5, 2.5], float mask
    from lengths, and the latent z = randn / temperature + mu (diffusion.py:227) drawn portably."""
    mu = np.clip(normalish("mu", (B, 80, T), seed) - 5.0, -11.5, 2.5).astype(np.float32)
    if lengths is None:
        lengths = [T] * B
    lengths = np.asarray(lengths, dtype=np.int64)
    mask = (np.arange(T)[None, :] < lengths[:, None]).astype(np.float32)[:, None, :]
    z = (normalish("z", (B, 80, T), seed + 100) / np.float32(temperature) + mu).astype(np.float32)
    return mu, mask, z, lengths


def make_dex_style(B: int, Tr: int, Ts: int, mid: int = 128, n_skips: int = 6, seed: int = 77,
                   ref_lengths=None, sty_lengths=None):
    """Synthetic style-encoder outputs for DEX (stand-ins for TIVEncoder skips / TVEncoder tokens,
    DEX-TTS/model/tts.py:42-51): 6 skips [B,mid,Tr], style tokens [B,mid,Ts], and their lengths."""
    ref = [normalish(f"ref{j}", (B, mid, Tr), seed) * np.float32(0.7 + 0.1 * j) + np.float32(0.05 * j)
           for j in range(n_skips)]
    sty = normalish("sty", (B, mid, Ts), seed)
    ref_lengths = np.asarray(ref_lengths if ref_lengths is not None else [Tr] * B, dtype=np.int64)
    sty_lengths = np.asarray(sty_lengths if sty_lengths is not None else [Ts] * B, dtype=np.int64)
    return ref, ref_lengths, sty.astype(np.float32), sty_lengths


def make_vocoder_weights(shapes: dict, seed: int = 0) -> dict:
    """Portable non-degenerate HiFi-GAN generator weights (keys of ``vocoder.param_shapes``): Conv1d U(-a, a) with
    a = gain*sqrt(3/(cin*k)) (gain 0.5 on the residual branches, 0.06 on conv_post so tanh stays out of saturation),
    ConvTranspose1d a = sqrt(3/cin) (two taps reach an output sample), biases 0.05 u."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith(".filter"):          # BigVGAN resampling filter: Kaiser-windowed sinc, cutoff 0.25, half-width 0.3, 12 taps (filter.py:30-60)
            beta = 0.1102 * ((2.285 * 5 * np.pi * 1.2 + 7.95) - 8.7)
            t = np.arange(-6, 6) + 0.5
            f = 0.5 * np.kaiser(12, beta) * np.sinc(0.5 * t)
            w = (f / f.sum()).reshape(shape)
        elif name.endswith((".act.alpha", ".act.beta")):       # log-scale frequency / magnitude of the periodic activation
            w = symmetric("voc." + name, shape, 0.4, seed)
        elif name.endswith(".bias"):
            w = symmetric("voc." + name, shape, 0.05, seed)
        elif name.startswith("ups."):
            cin, cout, k = shape
            w = symmetric("voc." + name, shape, float(np.sqrt(3.0 * 2.0 / (cin * 2))), seed)
        else:
            cout, cin, k = shape
            post = 0.015 if "activation_post.act.alpha" in shapes else 0.06      # (BigVGAN's periodic activations add energy: keep tanh out of saturation)
            gain = 0.5 if name.startswith("resblocks.") else (post if name.startswith("conv_post") else 1.0)
            w = symmetric("voc." + name, shape, gain * float(np.sqrt(3.0 / (cin * k))), seed)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_style_weights(shapes: dict, seed: int = 0) -> dict:
    """Portable non-degenerate weights for the DEX style encoders (keys of ``style.param_shapes``: the reference's
    ``tv_encoder.* / lf0_encoder.* / tiv_encoder.* / conv_sty.*`` state-dict entries): conv / GRU / linear weights
    U(-a, a) with a = sqrt(3 / fan_in); norm scales 1 + 0.1 u; BatchNorm running_mean 0.1 u, running_var 1 + 0.3 u;
    the VQ codebook 0.5 u; every other 1-D tensor 0.1 u."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        key = "sty." + name
        if name.endswith("num_batches_tracked"):
            w = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            w = 1.0 + symmetric(key, shape, 0.3, seed)
        elif name.endswith("running_mean"):
            w = symmetric(key, shape, 0.1, seed)
        elif name.endswith(("ln.weight", "bn.weight", ".gamma")):
            w = 1.0 + symmetric(key, shape, 0.1, seed)
        elif name.endswith("vq.embedding"):
            w = symmetric(key, shape, 0.5, seed)
        elif name.endswith(("vq.ema_count", "vq.ema_weight")):
            w = np.zeros(shape, dtype=np.float32)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            w = symmetric(key, shape, float(np.sqrt(3.0 / fan_in)), seed)
        else:
            w = symmetric(key, shape, 0.1, seed)
        out[name] = np.ascontiguousarray(w, dtype=np.int64 if name.endswith("num_batches_tracked") else np.float32).reshape(shape)
    return out


def make_style_inputs(B: int, T: int, lengths=None, seed: int = 99):
    """Synthetic reference-utterance features: mel [B,80,T] (mel-like), normalised log-f0 [B,T] with unvoiced zeros, lengths."""
    mel = np.clip(normalish("sty_mel", (B, 80, T), seed) * 1.5 - 5.0, -11.5, 2.5).astype(np.float32)
    lf0 = normalish("sty_lf0", (B, T), seed + 1).astype(np.float32)
    lf0[uniform01("sty_uv", B * T, seed).reshape(B, T) < 0.3] = 0.0
    lengths = np.asarray(lengths if lengths is not None else [T] * B, dtype=np.int64)
    return mel, lf0, lengths


def make_text_weights(shapes: dict, seed: int = 0) -> dict:
    """Portable non-degenerate weights for the text encoder (keys of ``text.param_shapes`` = the reference TextEncoder
    state dict): Linear / Conv1d U(-a, a) with a = sqrt(3 / fan_in) (x 0.7 for the q / k projections, so the softmax is
    neither flat nor one-hot), norm scales 1 + 0.1 u, biases and norm shifts 0.1 u, the embedding 0.6 u (x sqrt(192) at
    use), the AdaptiveLayerNorm scale bias 1 + 0.1 u; the duration head is scaled and shifted so durations spread over
    1..8 frames; ``retnet_rel_pos.angle`` / ``.decay`` are the reference's constants (retention.py:76-85)."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        key = "txt." + name
        if name.endswith("retnet_rel_pos.angle"):
            half = shape[0] // 2
            a = 1.0 / (10000.0 ** np.linspace(0.0, 1.0, half, dtype=np.float32))
            w = np.repeat(a.astype(np.float32), 2)
        elif name.endswith("retnet_rel_pos.decay"):
            w = np.log(1.0 - 2.0 ** (-5.0 - np.arange(shape[0], dtype=np.float32))).astype(np.float32)
        elif name == "emb.weight":
            w = symmetric(key, shape, 0.6, seed)
        elif name == "proj_w.proj.weight":
            w = symmetric(key, shape, 0.12, seed)
        elif name == "proj_w.proj.bias":
            w = np.full(shape, 1.0, dtype=np.float32)
        elif name.endswith((".gamma", "layer_norm.weight")) or name.endswith("W_scale.bias"):
            w = 1.0 + symmetric(key, shape, 0.1, seed)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.7 if name.endswith(("q_proj.weight", "k_proj.weight")) else (0.3 if ".adaln_" in name else 1.0)
            w = symmetric(key, shape, gain * float(np.sqrt(3.0 / fan_in)), seed)
        else:
            w = symmetric(key, shape, 0.1, seed)
        out[name] = np.ascontiguousarray(w, dtype=np.float32).reshape(shape)
    return out


def make_text_inputs(B: int, L: int, lengths=None, n_vocab: int = 149, seed: int = 321):
    """Token ids [B, L] int64 (0 beyond each length), lengths [B] int64."""
    lengths = np.asarray(lengths if lengths is not None else [L] * B, dtype=np.int64)
    tok = (uniform01("tokens", B * L, seed) * n_vocab).astype(np.int64).reshape(B, L) % n_vocab
    for b in range(B):
        tok[b, lengths[b]:] = 0
    return tok, lengths


# ---- pinned whole-job cases: the inputs of the long sampler jobs whose CPU-oracle outputs are committed under tests/golden/oracle_jobs/
# (written by oracle/make_oracle_jobs.py; data, not code).  The GPU tests compare against them (tests/gpu_util.py) and bench.py measures
# the `abs_err` fields of its configs blocks against them in the run itself.  Nothing here touches oracle/.
ORACLE_VERSION = 1          # bump when oracle/dex_oracle.py changes what it computes: every stored job is then recomputed
_WKEY = {}


def make_case(cfg, B, T, lengths=None, Tr=40, Ts=40, sty_lengths=None, seed=1234):
    mu, mask, z, lengths = make_inputs(B, T, lengths, seed=seed)
    case = {"mu": mu, "mask": mask, "z": z, "eps": normalish("eps", (B, 80, T), seed + 5)}
    if cfg.variant == "dex":
        ref, rl, sty, sl = make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sty_lengths)
        case.update(ref=np.stack(ref), ref_lengths=rl, sty=sty, sty_lengths=sl)
    if cfg.n_spks > 1:
        case["spk"] = normalish("spk", (B, cfg.spk_emb_dim), 9)
    return case


def case_key(case):
    import zlib
    parts = []
    for k in sorted(case):
        a = np.ascontiguousarray(case[k])
        parts.append((k, a.shape, zlib.crc32(a.tobytes())))
    return tuple(parts)


def weights_key(name):
    """crc of the packed synthetic weights + the preset's config: a changed make_weights / PRESETS entry misses the store"""
    if name not in _WKEY:
        import zlib
        from . import config as C
        cfg = C.PRESETS[name]()
        w = make_weights(C.param_shapes(cfg))
        crc = 0
        for k in sorted(w):
            crc = zlib.crc32(np.ascontiguousarray(w[k]).tobytes(), zlib.crc32(k.encode(), crc))
        _WKEY[name] = (crc, repr(cfg))
    return _WKEY[name]


def stored_job_path(root, name, case, n_steps, solver="euler"):
    import hashlib
    import os
    h = hashlib.sha1(repr((name, int(n_steps), solver, case_key(case), weights_key(name), ORACLE_VERSION)).encode()).hexdigest()[:16]
    return os.path.join(root, "tests", "golden", "oracle_jobs", f"{name}_n{int(n_steps)}_{solver}_{h}.npy")


def pinned_job_case(name):
    """(case, n_steps) of the whole job bench.py / tests/test_gpu_full_jobs.py pin for a preset: BASELINE.json configs[2] (dex_vctk),
    the per-GPU share of configs[3] (dex_esd), configs[4] (gedex_lj: long form)"""
    from . import config as C
    cfg = C.PRESETS[name]()
    if name == "gedex_lj":
        return make_case(cfg, B=1, T=4000), 50
    case = make_case(cfg, B=32, T=256, lengths=[256 - 3 * i for i in range(32)], Tr=348, Ts=348, sty_lengths=[348 - 5 * i for i in range(32)])
    return case, (100 if name == "dex_esd" else 50)


def kernel_sources_sha(root):
    """sha1 over the kernel sources (dex_tts_amd/csrc: .hip / .h / .inc, sorted by name): what a committed rocprof summary was measured on
    (tools/collect_profiles.py writes it to profiles/profiles_head.json, bench.py compares it with the tree it runs from)"""
    import hashlib
    import os
    d = os.path.join(root, "dex_tts_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]
