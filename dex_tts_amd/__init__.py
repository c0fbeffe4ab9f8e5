"""dex_tts_amd — MI355X-native reverse-diffusion sampler for DEX-TTS / GeDEX-TTS (hot path only).

Layout: csrc/ (hand-written gfx950 kernels + C ABI), _lib.py (ctypes binding), engine.py (context owner),
diffusion.py (drop-in ``Diffusion`` module), audio.py (STFT/mel front-end), dist.py (utterance sharding).
"""
from .config import ScoreNetConfig, DiTConfig, param_shapes, PRESETS, fix_len_compatibility  # noqa: F401

__all__ = ["ScoreNetConfig", "DiTConfig", "param_shapes", "PRESETS", "fix_len_compatibility"]
