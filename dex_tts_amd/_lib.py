"""ctypes binding of libdexamd.so (C ABI declared in include/dex_amd.h).  No CPU fallback: if the
library is missing this module raises, and the product path fails loudly."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DEX_AMD_LIB: A/B tooling only (two builds of the same library side by side); still no fallback if it is absent
LIB_PATH = os.environ.get("DEX_AMD_LIB") or os.path.join(HERE, "lib", "libdexamd.so")

DEX_OK = 0
DEX_ERR_HANDOFF, DEX_ERR_HANDOFF_XCD = -5, -6        # include/dex_amd.h: dex_call_status
DEX_PENDING = 1                                   # dex_call_status_poll(wait = 0)
VARIANT = {"gedex": 0, "dex": 1}
PRECISION = {"fp32": 0, "bf16": 1, "fp16": 2, "fp16x2": 3}
SOLVER = {"euler": 0, "heun": 1}


class DexConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_feats", C.c_int32), ("dim", C.c_int32), ("n_stages", C.c_int32),
                ("dim_mults", C.c_int32 * 4), ("n_spks", C.c_int32), ("spk_emb_dim", C.c_int32),
                ("pe_scale", C.c_float),
                ("dit_patch", C.c_int32), ("dit_stride", C.c_int32), ("dit_hidden", C.c_int32),
                ("dit_depth", C.c_int32), ("dit_heads", C.c_int32), ("dit_mlp_ratio", C.c_float),
                ("dit_conv_pos", C.c_int32), ("dit_conv_pos_groups", C.c_int32)]


class DexSampleArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("n_steps", C.c_int32),
                ("z_dev", C.c_void_p), ("mu_dev", C.c_void_p), ("mask_dev", C.c_void_p), ("sigmas_dev", C.c_void_p),
                ("spk_dev", C.c_void_p),
                ("ref_skips_dev", C.POINTER(C.c_void_p)), ("n_ref", C.c_int32), ("Tr", C.c_int32),
                ("sty_dev", C.c_void_p), ("sty_lengths_dev", C.c_void_p), ("Ts", C.c_int32),
                ("out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("use_graph", C.c_int32), ("solver", C.c_int32),
                ("noise_dev", C.c_void_p), ("S_churn", C.c_float), ("S_min", C.c_float), ("S_max", C.c_float), ("S_noise", C.c_float)]


class DexVocoderConfig(C.Structure):
    _fields_ = [("num_mels", C.c_int32), ("upsample_initial_channel", C.c_int32), ("n_upsamples", C.c_int32),
                ("upsample_rates", C.c_int32 * 6), ("upsample_kernel_sizes", C.c_int32 * 6), ("n_resblock_kernels", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * 3), ("resblock_dilation_sizes", (C.c_int32 * 3) * 3),
                ("activation", C.c_int32), ("snake_logscale", C.c_int32)]


class DexStyleConfig(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("tiv_layers", C.c_int32), ("tiv_ch", C.c_int32),
                ("tv_layers", C.c_int32), ("tv_ch", C.c_int32), ("tv_cout", C.c_int32), ("tv_cout_g", C.c_int32), ("tv_n_emb", C.c_int32),
                ("lf0_ch", C.c_int32), ("lf0_cout", C.c_int32), ("lf0_cout_g", C.c_int32), ("lf0_layers", C.c_int32), ("sty_out", C.c_int32)]


class DexStyleArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("Tr", C.c_int32), ("Ts", C.c_int32), ("Tl", C.c_int32),
                ("ref_mel_dev", C.c_void_p), ("ref_lengths_dev", C.c_void_p), ("sty_mel_dev", C.c_void_p), ("sty_lengths_dev", C.c_void_p),
                ("lf0_dev", C.c_void_p), ("lf0_lengths_dev", C.c_void_p), ("ref_skips_out_dev", C.POINTER(C.c_void_p)),
                ("sty_dec_out_dev", C.c_void_p), ("sty_enc_out_dev", C.c_void_p), ("vq_idx_out_dev", C.c_void_p),
                ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class DexTextConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_vocab", C.c_int32), ("n_feats", C.c_int32), ("n_channels", C.c_int32),
                ("filter_channels", C.c_int32), ("filter_channels_dp", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32),
                ("kernel_size", C.c_int32), ("n_spks", C.c_int32), ("spk_emb_dim", C.c_int32), ("use_softmax", C.c_int32), ("use_decay", C.c_int32)]


class DexTextArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("tokens_dev", C.c_void_p), ("lengths_dev", C.c_void_p), ("spk_dev", C.c_void_p),
                ("sty_dev", C.c_void_p), ("length_scale", C.c_float), ("mu_out_dev", C.c_void_p), ("logw_out_dev", C.c_void_p),
                ("w_ceil_out_dev", C.c_void_p), ("y_lengths_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class DexAlignArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("Ty", C.c_int32), ("mu_x_dev", C.c_void_p), ("w_ceil_dev", C.c_void_p),
                ("x_lengths_dev", C.c_void_p), ("y_lengths_dev", C.c_void_p), ("mu_y_out_dev", C.c_void_p), ("y_mask_out_dev", C.c_void_p),
                ("attn_out_dev", C.c_void_p), ("workspace_dev", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class DexDenoiseArgs(C.Structure):
    _fields_ = [("s", DexSampleArgs), ("x_dev", C.c_void_p)]


# every symbol include/dex_amd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("dex_ctx_create", C.c_int, [C.POINTER(DexConfig), C.POINTER(C.c_void_p)]),
    ("dex_ctx_destroy", None, [C.c_void_p]),
    ("dex_last_error", C.c_char_p, [C.c_void_p]),
    ("dex_version", C.c_char_p, []),
    ("dex_num_evals", C.c_int, [C.c_int, C.c_int]),
    ("dex_ctx_num_weights", C.c_int, [C.c_void_p]),
    ("dex_ctx_weight_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dex_ctx_load_weight", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    ("dex_ctx_load_weight_async", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    ("dex_ctx_finalize", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_ctx_set_precision", C.c_int, [C.c_void_p, C.c_int]),
    ("dex_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dex_sample", C.c_int, [C.c_void_p, C.POINTER(DexSampleArgs), C.c_void_p]),
    ("dex_denoise_once", C.c_int, [C.c_void_p, C.POINTER(DexDenoiseArgs), C.c_void_p]),
    ("dex_edm_sigmas", C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    ("dex_num_taps", C.c_int, [C.c_void_p]),
    ("dex_tap_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("dex_tap_info", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dex_tap_copy", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("dex_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("dex_profile_num", C.c_int, [C.c_void_p]),
    ("dex_profile_get", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dex_voc_create", C.c_int, [C.POINTER(DexVocoderConfig), C.POINTER(C.c_void_p)]),
    ("dex_voc_destroy", None, [C.c_void_p]),
    ("dex_voc_last_error", C.c_char_p, [C.c_void_p]),
    ("dex_voc_num_weights", C.c_int, [C.c_void_p]),
    ("dex_voc_weight_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dex_voc_load_weight_async", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    ("dex_voc_finalize", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_voc_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    ("dex_voc_samples", C.c_int, [C.c_void_p, C.c_int]),
    ("dex_voc_set_precision", C.c_int, [C.c_void_p, C.c_int]),
    ("dex_vocode", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("dex_style_create", C.c_int, [C.POINTER(DexStyleConfig), C.POINTER(C.c_void_p)]),
    ("dex_style_destroy", None, [C.c_void_p]),
    ("dex_style_last_error", C.c_char_p, [C.c_void_p]),
    ("dex_style_num_weights", C.c_int, [C.c_void_p]),
    ("dex_style_weight_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dex_style_load_weight_async", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    ("dex_style_finalize", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_style_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dex_style_encode", C.c_int, [C.c_void_p, C.POINTER(DexStyleArgs), C.c_void_p]),
    ("dex_text_create", C.c_int, [C.POINTER(DexTextConfig), C.POINTER(C.c_void_p)]),
    ("dex_text_destroy", None, [C.c_void_p]),
    ("dex_text_last_error", C.c_char_p, [C.c_void_p]),
    ("dex_text_num_weights", C.c_int, [C.c_void_p]),
    ("dex_text_weight_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dex_text_load_weight_async", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    ("dex_text_finalize", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_text_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    ("dex_text_encode", C.c_int, [C.c_void_p, C.POINTER(DexTextArgs), C.c_void_p]),
    ("dex_text_align", C.c_int, [C.c_void_p, C.POINTER(DexAlignArgs), C.c_void_p]),
    ("dex_call_status", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_call_status_begin", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_call_status_poll", C.c_int, [C.c_void_p, C.c_int]),
    ("dex_debug_handoff_timeouts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dex_debug_xcd_local", C.c_int, []),
    ("dex_mel_frames", C.c_int, [C.c_int]),
    ("dex_mel_from_wav", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("dex_mel_create", C.c_int, [C.POINTER(C.c_void_p)]),
    ("dex_mel_destroy", None, [C.c_void_p]),
    ("dex_mel_last_error", C.c_char_p, [C.c_void_p]),
    ("dex_mel_workspace_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("dex_mel_spectrogram", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("dex_lf0_normalize", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
]

_lib = None


def load():
    """dlopen libdexamd.so and type its entry points.  Raises if the library is absent — build it with
    `python -m dex_tts_amd.build` (hipcc, gfx950)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP extension is required (python -m dex_tts_amd.build); "
                           "there is no CPU fallback in the product path")
    # torch (when installed) goes first: it ships its own HIP runtime, and a process that binds /opt/rocm's through this library
    # before importing torch ends up with a torch that sees no device
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def make_config(cfg) -> DexConfig:
    c = DexConfig()
    c.variant = VARIANT[cfg.variant]
    c.n_feats, c.dim, c.n_stages = cfg.n_feats, cfg.dim, len(cfg.dim_mults)
    for i, m in enumerate(cfg.dim_mults):
        c.dim_mults[i] = int(m)
    c.n_spks, c.spk_emb_dim, c.pe_scale = cfg.n_spks, cfg.spk_emb_dim, float(cfg.pe_scale)
    t = cfg.dit
    c.dit_patch, c.dit_stride, c.dit_hidden, c.dit_depth, c.dit_heads = t.patch_size, t.stride_size, t.hidden_size, t.depth, t.num_heads
    c.dit_mlp_ratio, c.dit_conv_pos, c.dit_conv_pos_groups = float(t.mlp_ratio), t.conv_pos, t.conv_pos_groups
    return c
