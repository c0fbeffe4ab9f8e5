/*
 * dex_amd.h — C ABI of libdexamd.so: MI355X (gfx950) reverse-diffusion sampler for DEX-TTS / GeDEX-TTS.
 *
 * Drop-in boundary for the reference's Diffusion.forward(..., infer=True) hot path
 *   GeDEX-TTS/model/diffusion.py:220-229, DEX-TTS/model/diffusion.py:250-259
 *   -> ablation_sampler (euler/edm/linear/none)   GeDEX-TTS/model/edm.py:109-216 (DEX :104-211)
 *   -> EDMPrecond.forward                          model/edm.py:88-98
 *   -> DiffusionDenoiser.forward (+DiTMask, TV/TIV adaptors)
 *                                                  GeDEX diffusion.py:168-207, DEX :190-236, model/dit.py:485-525,
 *                                                  DEX-TTS/model/ref_encoder.py:142-179,255-273
 * and for the STFT/mel front-end audio/tools.py:8-15 -> audio/stft.py:159-178,52-81.
 *
 * Conventions: every pointer named *_dev is a DEVICE pointer (HBM) to contiguous fp32 unless stated;
 * all work is enqueued asynchronously on the caller's HIP stream; the library never synchronises the
 * stream inside dex_sample/dex_denoise_once.  Functions return 0 on success and a negative DexStatus
 * otherwise; nothing throws across the ABI; dex_last_error() gives the message.
 * Ownership: the caller owns inputs, outputs and the workspace; the library owns the context, its
 * packed weight copies (hipMalloc at dex_ctx_finalize) and captured hipGraphs.
 */
#ifndef DEX_AMD_H
#define DEX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DexCtx DexCtx;
typedef void* dex_stream_t; /* hipStream_t */

typedef enum {
    DEX_PENDING = 1,       /* dex_call_status_poll(wait = 0): the stream has not reached the call's status copy yet */
    DEX_OK = 0,
    DEX_ERR_ARG = -1,      /* bad argument / shape (e.g. T % 4 != 0, unknown key, wrong weight shape) */
    DEX_ERR_STATE = -2,    /* call order (weights missing, not finalized) */
    DEX_ERR_HIP = -3,      /* a HIP runtime call failed */
    DEX_ERR_WORKSPACE = -4,/* workspace too small */
    DEX_ERR_HANDOFF = -5,  /* dex_call_status: an in-launch hand-off of the small-batch DiT block timed out; the call's outputs are NaN */
    DEX_ERR_HANDOFF_XCD = -6 /* dex_call_status: a hand-off met its peer on another XCD (outputs NaN); the XCD-local form is now off for the
                              * device, so REPEATING the call succeeds */
} DexStatus;

typedef enum { DEX_VARIANT_GEDEX = 0, DEX_VARIANT_DEX = 1 } DexVariant;
/* Arithmetic of the contractions: fp32 = exact-fp32 MFMA (the reference's arithmetic, the parity mode); bf16 / fp16 = operands
 * rounded to that type on the MFMA, fp32 accumulation, fp32 norms / softmax state / residual streams. */
typedef enum { DEX_PREC_FP32 = 0, DEX_PREC_BF16 = 1, DEX_PREC_FP16 = 2,
               DEX_PREC_FP16X2 = 3    /* fp16 operands with every weight as hi + lo (two MFMAs per product): the fast mode inside the
                                       * fp32-grade sampler bound */
} DexPrecision;
/* ablation_sampler's solver argument (edm.py:107). */
typedef enum { DEX_SOLVER_EULER = 0, DEX_SOLVER_HEUN = 1 } DexSolver;

/* Mirrors Diffusion(**cfg.decoder, dit_cfg=cfg.dit): GeDEX diffusion.py:210, dit.py:339-356. */
typedef struct {
    int32_t variant;        /* DexVariant */
    int32_t n_feats;        /* 80 (hard-coded at diffusion.py:226) */
    int32_t dim;            /* decoder.dim (64) */
    int32_t n_stages;       /* len(dim_mults) (2) */
    int32_t dim_mults[4];
    int32_t n_spks;         /* >1 adds the speaker plane (GeDEX-VCTK) */
    int32_t spk_emb_dim;
    float   pe_scale;       /* 1000 */
    int32_t dit_patch, dit_stride, dit_hidden, dit_depth, dit_heads;
    float   dit_mlp_ratio;
    int32_t dit_conv_pos, dit_conv_pos_groups;
} DexConfig;

/* One Diffusion.forward(infer=True) call == ablation_sampler with a given latent z. */
typedef struct {
    int32_t B, T;               /* batch, padded mel frames (T % 4 == 0, model/utils.py:13-17) */
    int32_t n_steps;            /* n_timesteps >= 2 */
    const float* z_dev;         /* [B,80,T] latent = randn/temperature + mu (diffusion.py:227); drawn by the caller */
    const float* mu_dev;        /* [B,80,T] */
    const float* mask_dev;      /* [B,T]   float 0/1 (reference shape [B,1,T]) */
    const float* sigmas_dev;    /* [n_steps+1] fp32 noise levels t_0..t_{N-1}, t_N=0 (edm.py:157,184-185) */
    const float* spk_dev;       /* [B,spk_emb_dim] or NULL */
    /* DEX only (NULL/0 otherwise): */
    const float* const* ref_skips_dev; /* HOST array of 6 device pointers, each [B,mid,Tr] */
    int32_t n_ref, Tr;
    const float* sty_dev;       /* [B,mid,Ts] */
    const int32_t* sty_lengths_dev; /* [B] int32 */
    int32_t Ts;
    float* out_dev;             /* [B,80,T] x_N (unmasked, like edm.py:216) */
    void*  workspace_dev;       /* >= dex_workspace_bytes(...) bytes, 256-B aligned */
    size_t workspace_bytes;
    int32_t use_graph;          /* 1: the whole call (conditioning tables + every network evaluation + the final copy) is captured
                                 * once into a hipGraph, cached under (shapes, solver, precision, stream, every device pointer
                                 * above) and replayed with ONE hipGraphLaunch; needs a non-default stream */
    int32_t solver;             /* DexSolver; 0 = Euler, what Diffusion wires (diffusion.py:216) */
    /* Stochastic sampler, ablation_sampler's S_churn / S_min / S_max / S_noise (edm.py:109,194-196).  Zero-initialised
     * fields = the deterministic sampler the reference wires (S_churn = 0: its per-step randn_like is multiplied by 0). */
    const float* noise_dev;     /* [n_steps][B,80,T]: step i's randn_like(x_cur) draw (the caller owns the RNG); required
                                 * when S_churn > 0, ignored otherwise */
    float S_churn, S_min, S_max, S_noise;   /* S_max <= 0 means +inf; S_noise is used as given when S_churn > 0 */
} DexSampleArgs;

/* One EDMPrecond.forward call (edm.py:88-98): out = c_skip*x + c_out*F(c_in*x, mask, mu, ln(sigma)/4). */
typedef struct {
    DexSampleArgs s;            /* z_dev is ignored; n_steps ignored; sigmas_dev[0] = sigma */
    const float* x_dev;         /* [B,80,T] */
} DexDenoiseArgs;

int  dex_ctx_create(const DexConfig* cfg, DexCtx** out);
void dex_ctx_destroy(DexCtx* ctx);
const char* dex_last_error(const DexCtx* ctx);
const char* dex_version(void);

/* Number of state-dict tensors the context expects, and the i-th key (relative to "denoise_fn.") + shape. */
int  dex_ctx_num_weights(const DexCtx* ctx);
int  dex_ctx_weight_info(const DexCtx* ctx, int i, const char** key, int64_t shape[4], int* ndim);
/* Hand over one tensor in the REFERENCE layout (fp32, contiguous, device memory).  The library copies it into its own
 * HBM before returning control of the pointer:
 *   dex_ctx_load_weight        synchronous — the copy runs on the legacy null stream and has completed on return; the
 *                              caller must have completed whatever produced w_dev (it is NOT ordered against
 *                              non-blocking streams);
 *   dex_ctx_load_weight_async  the copy is enqueued on `stream`, i.e. ordered after the kernels that produced w_dev on
 *                              that stream; w_dev must stay valid until the stream reaches the copy (dex_ctx_finalize on
 *                              the same stream synchronises it). */
int  dex_ctx_load_weight(DexCtx* ctx, const char* key, const float* w_dev, const int64_t* shape, int ndim);
int  dex_ctx_load_weight_async(DexCtx* ctx, const char* key, const float* w_dev, const int64_t* shape, int ndim,
                               dex_stream_t stream);
/* Pack all weights into kernel layouts (library-owned HBM) on `stream` and wait for it on the host before returning:
 * after dex_ctx_finalize the packed weights are complete for work on ANY stream. */
int  dex_ctx_finalize(DexCtx* ctx, dex_stream_t stream);
int  dex_ctx_set_precision(DexCtx* ctx, int precision /* DexPrecision */);

/* n_evals = number of network evaluations of the run: dex_num_evals(n_steps, solver). */
size_t dex_workspace_bytes(const DexCtx* ctx, int B, int T, int Tr, int Ts, int n_evals);
/* Euler: n_steps.  Heun (edm.py:202-214): 2*n_steps - 1 (no corrector on the last step). */
int  dex_num_evals(int n_steps, int solver /* DexSolver */);
int  dex_sample(DexCtx* ctx, const DexSampleArgs* args, dex_stream_t stream);
int  dex_denoise_once(DexCtx* ctx, const DexDenoiseArgs* args, dex_stream_t stream);

/* EDM rho=7 schedule in fp32 on the host (edm.py:157): writes n_steps+1 values, last one 0. */
int  dex_edm_sigmas(int n_steps, float* sigmas_host);

/* Debug taps: copy a named intermediate of the LAST dex_denoise_once/dex_sample call out of the
 * workspace (device->device, async on stream).  Names: see dex_tap_name(i).  Layout NHWC fp32. */
int  dex_num_taps(const DexCtx* ctx);
const char* dex_tap_name(const DexCtx* ctx, int i);
int  dex_tap_info(const DexCtx* ctx, const char* name, int64_t shape[4], int* ndim);
int  dex_tap_copy(DexCtx* ctx, const char* name, float* dst_dev, size_t dst_bytes, dex_stream_t stream);

/* Per-kernel timing of the last dex_sample with profiling enabled (HIP events on the launch stream). */
int  dex_profile_enable(DexCtx* ctx, int on);
int  dex_profile_num(const DexCtx* ctx);
int  dex_profile_get(const DexCtx* ctx, int i, const char** name, int* calls, double* total_ms,
                     double* flops, double* bytes);

/* Status of the LAST dex_sample / dex_denoise_once call of this context on `stream`.  Small-grid DiT blocks run as clusters of
 * co-operating workgroups whose in-launch hand-offs are bounded waits: a lost hand-off cannot hang the GPU, it turns every output of
 * the call into NaN and sets a device word.  This call waits for the stream and reads that word: DEX_OK, DEX_ERR_HANDOFF (a
 * time-out) or DEX_ERR_HANDOFF_XCD (a peer on another XCD) - the outputs are NaN in both.  In the second case the XCD-local form is
 * switched off for the device - for every context - so repeating the call succeeds.  A hipGraph replay is checked like an eager call
 * (the graph entry remembers the word its captured launches write).  Calls that used no hand-offs return DEX_OK without waiting.  The Python mirror
 * (ScoreNetEngine.sample) checks every call that could use hand-offs and raises. */
int  dex_call_status(DexCtx* ctx, dex_stream_t stream);
/* The same verdict WITHOUT blocking the host (the sampler call is asynchronous: a caller overlaps the vocoder of utterance i with the
 * sampler of utterance i + 1).  _begin enqueues, behind the last dex_sample / dex_denoise_once on `stream`, a copy of the call's
 * hand-off word into a pinned host word of the context and an event; it returns at once (calls without hand-offs: nothing is enqueued).
 * _poll returns the verdict of that call - DEX_OK / DEX_ERR_HANDOFF / DEX_ERR_HANDOFF_XCD - once the event has passed; before that
 * DEX_PENDING (wait = 0), or it waits for that event alone (wait != 0; never for later work on the stream).  One check is in flight per
 * context: a second _begin first resolves the pending one (and returns its error if it failed).  The Python mirror's default
 * (ScoreNetEngine.check_handoffs = "deferred") begins a check after every call and reads it at the next call or at .status(). */
int  dex_call_status_begin(DexCtx* ctx, dex_stream_t stream);
int  dex_call_status_poll(DexCtx* ctx, int wait);
/* Debug: 1 if a workgroup hand-off of the LAST dex_sample / dex_denoise_once call on this context timed out (small-grid DiT
 * blocks run as clusters of co-operating workgroups; a wait is bounded so a lost hand-off cannot hang the GPU), 0 if none did or
 * the call used no hand-offs, < 0 on a HIP error.  Synchronises the stream; the call's workspace must still be alive. */
int  dex_debug_handoff_timeouts(DexCtx* ctx, dex_stream_t stream);
/* Debug: 1 if workgroup b of a launch runs on XCD b % 8 on this device (probed once per process with launches that record
 * HW_REG_XCC_ID), which lets the clusters above keep their hand-offs inside one XCD's L2; 0 if not (or DEX_DIT_CLUSTER_LOCAL=0):
 * the hand-offs then go through memory.  Every hand-off of the XCD-local form re-checks its peers' XCC ids; a mismatch poisons the
 * call like a time-out (dex_debug_handoff_timeouts returns 2 and switches the form off for the process). */
int  dex_debug_xcd_local(void);

/* STFT/mel front-end (audio/tools.py:8-15): wav [L] fp32 in [-1,1] (clipped here) -> mel [80,frames],
 * energy [frames]; frames = L/256 + 1.  n_fft=1024, hop=256, 80 mels, 22050 Hz, fmin 0, fmax 8000. */
int  dex_mel_frames(int n_samples);
int  dex_mel_from_wav(DexCtx* ctx, const float* wav_dev, int n_samples, float* mel_dev, float* energy_dev,
                      dex_stream_t stream);

/* The same front-end without a score-network context and for a batch — replaces TacotronSTFT.mel_spectrogram(y [B,L])
 * (audio/stft.py:159-178, called by preprocess/preprocessor/preprocessor.py:100 and audio/tools.py:8-15): B equally long rows
 * in ONE pass (pad/clip kernel, one batched windowed-DFT GEMM, one magnitude/mel/log kernel).  wav [B,L] fp32 (clipped to
 * [-1,1] here) -> mel [B,80,frames], energy [B,frames]; the caller owns the workspace (dex_mel_workspace_bytes). */
typedef struct DexMel DexMel;
int  dex_mel_create(DexMel** out);
void dex_mel_destroy(DexMel* mel);
const char* dex_mel_last_error(const DexMel* mel);
size_t dex_mel_workspace_bytes(int B, int n_samples);
int  dex_mel_spectrogram(DexMel* mel, const float* wav_dev, int B, int n_samples, float* mel_dev, float* energy_dev,
                         void* workspace_dev, size_t workspace_bytes, dex_stream_t stream);

/* Deterministic tail of the DEX f0 front-end (DEX-TTS/synthesize.py:26-38,55-58): f0 [B,T] in Hz (0 = unvoiced; from the
 * host's DIO/StoneMask, a third-party CPU algorithm that stays on the host) -> lf0 [B,T] = normalize_lf0(log f0), what
 * dex_style_encode takes as lf0_dev.  lengths_dev [B] int32 or NULL (= T); positions past an utterance's length are 0.
 * T <= 16382 frames (190 s of audio at hop 256): DEX_ERR_ARG beyond. */
int  dex_lf0_normalize(const float* f0_dev, const int* lengths_dev, int B, int T, float* lf0_dev, dex_stream_t stream);

/* ---- Vocoder: HiFi-GAN generator (GeDEX-TTS/hifigan/models.py:112-173, built by src/utils.py:251-281 from
 * hifigan/config.json) — the step right after the sampler: mel [B,80,T] -> waveform [B, T * prod(upsample_rates)].
 * A separate context: it shares nothing with the score network. */
typedef struct DexVoc DexVoc;
typedef struct {
    int32_t num_mels;                   /* 80 */
    int32_t upsample_initial_channel;   /* 512 (V1) */
    int32_t n_upsamples;                /* len(upsample_rates), <= 6 */
    int32_t upsample_rates[6];          /* [8,8,2,2] */
    int32_t upsample_kernel_sizes[6];   /* [16,16,4,4] */
    int32_t n_resblock_kernels;         /* len(resblock_kernel_sizes), must be 3 (the stage average is xs / 3) */
    int32_t resblock_kernel_sizes[3];   /* [3,7,11] */
    int32_t resblock_dilation_sizes[3][3]; /* [[1,3,5]]*3 (ResBlock "1": three dilated + three plain convs each) */
    /* 0: HiFi-GAN (leaky_relu in front of every conv).  1 / 2: BigVGAN (DEX-TTS/bigvgan/models.py:138-211, AMPBlock1) with the
     * anti-aliased Snake / SnakeBeta activation (alias_free_torch/act.py, activations.py) in front of every ResBlock conv and of
     * conv_post, and no activation in front of the transposed convs. */
    int32_t activation;
    int32_t snake_logscale;             /* BigVGAN: alpha / beta are stored as logarithms (config "snake_logscale") */
} DexVocoderConfig;

int  dex_voc_create(const DexVocoderConfig* cfg, DexVoc** out);
void dex_voc_destroy(DexVoc* voc);
const char* dex_voc_last_error(const DexVoc* voc);
/* Generator.state_dict() keys AFTER remove_weight_norm() (models.py:169-173: "conv_pre.weight", "ups.0.bias",
 * "resblocks.4.convs1.2.weight", "conv_post.weight", ...), reference shapes; the host folds weight_g / weight_v pairs.
 * BigVGAN: "ups.<i>.0.weight" (nested ModuleList), "resblocks.<n>.activations.<l>.act.alpha" [+ ".beta"],
 * "activation_post.act.alpha" [+ ".beta"], and ONE copy of the two (identical) 12-tap resampling filters,
 * "activation_post.upsample.filter" / "activation_post.downsample.lowpass.filter" [1,1,12]. */
int  dex_voc_num_weights(const DexVoc* voc);
int  dex_voc_weight_info(const DexVoc* voc, int i, const char** key, int64_t shape[4], int* ndim);
int  dex_voc_load_weight_async(DexVoc* voc, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream);
int  dex_voc_finalize(DexVoc* voc, dex_stream_t stream);
size_t dex_voc_workspace_bytes(const DexVoc* voc, int B, int T);
int  dex_voc_samples(const DexVoc* voc, int T);          /* T * prod(upsample_rates) */
/* Operand precision of the generator's convolutions (DexPrecision): DEX_PREC_FP32 (default; exact-fp32 MFMA, the parity mode) or
 * DEX_PREC_BF16 / DEX_PREC_FP16 (operands rounded while staged, fp32 accumulation, fp32 activations in HBM, weights packed for
 * both at dex_voc_finalize).  Takes effect at the next dex_vocode. */
int  dex_voc_set_precision(DexVoc* voc, int precision);
/* Generator.forward (models.py:150-167): mel_dev [B,num_mels,T] fp32 -> wav_dev [B, dex_voc_samples(T)] fp32 in [-1,1].
 * Exact-fp32 MFMA contractions (the reference's arithmetic).  Asynchronous on `stream`. */
int  dex_vocode(DexVoc* voc, const float* mel_dev, int B, int T, float* wav_dev, void* workspace_dev, size_t workspace_bytes,
                dex_stream_t stream);

/* ---- DEX style encoders (DEX-TTS/model/ref_encoder.py TVEncoder :110-140 + VQEmbeddingEMA :199-237, LF0Encoder
 * :36-55, TIVEncoder :83-108, DeXTTS.conv_sty tts.py:31) and the part of DeXTTS.forward that feeds the decoder (tts.py:55-66):
 * the step right before the sampler for the DEX configs, once per reference utterance.  Eval mode (dropout off, BatchNorm on
 * running statistics — folded into the convolutions by the caller —, frozen codebook).  A separate context. */
typedef struct DexStyle DexStyle;
typedef struct {
    int32_t n_mels;                                             /* 80 */
    int32_t tiv_layers, tiv_ch;                                 /* tiv_encoder: num_layer 6, c_h 128 (skips [B,c_h,Tr]) */
    int32_t tv_layers, tv_ch, tv_cout, tv_cout_g, tv_n_emb;     /* tv_encoder: 6, 128, 192, 192, 512 */
    int32_t lf0_ch, lf0_cout, lf0_cout_g, lf0_layers;           /* lf0_encoder: 192, 192, 192, 2 (GRU hidden = lf0_ch / 2 = 96) */
    int32_t sty_out;                                            /* conv_sty output channels = 2 * decoder.dim (128) */
} DexStyleConfig;
typedef struct {
    int32_t B, Tr, Ts, Tl;
    const float* ref_mel_dev;  const int32_t* ref_lengths_dev;  /* [B,n_mels,Tr], [B]  -> TIVEncoder */
    const float* sty_mel_dev;  const int32_t* sty_lengths_dev;  /* [B,n_mels,Ts], [B]  -> TVEncoder  */
    const float* lf0_dev;      const int32_t* lf0_lengths_dev;  /* [B,Tl], [B]         -> LF0Encoder (normalised log-f0, 0 = unvoiced) */
    float* const* ref_skips_out_dev;   /* HOST array of tiv_layers device pointers, each [B,tiv_ch,Tr]: Diffusion.forward's `ref` */
    float* sty_dec_out_dev;            /* [B,sty_out,Ts]: Diffusion.forward's `sty` (conv_sty(z_dec + mean lf0_dec)) */
    float* sty_enc_out_dev;            /* [B,tv_cout]: pooled style vector for the text encoder (tts.py:62-63) */
    int32_t* vq_idx_out_dev;           /* optional [B,Ts]: the chosen codebook rows */
    void* workspace_dev; size_t workspace_bytes;
} DexStyleArgs;

int  dex_style_create(const DexStyleConfig* cfg, DexStyle** out);
void dex_style_destroy(DexStyle* sty);
const char* dex_style_last_error(const DexStyle* sty);
/* Keys = the reference state-dict names under tv_encoder.* / lf0_encoder.* / tiv_encoder.* / conv_sty.*; a conv followed by
 * BatchNorm is handed over FOLDED as "<p>.conv.weight" + "<p>.conv.bias" (dex_tts_amd/style.py does it). */
int  dex_style_num_weights(const DexStyle* sty);
int  dex_style_weight_info(const DexStyle* sty, int i, const char** key, int64_t shape[4], int* ndim);
int  dex_style_load_weight_async(DexStyle* sty, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream);
int  dex_style_finalize(DexStyle* sty, dex_stream_t stream);
size_t dex_style_workspace_bytes(const DexStyle* sty, int B, int Tr, int Ts, int Tl);
int  dex_style_encode(DexStyle* sty, const DexStyleArgs* args, dex_stream_t stream);

/* ---- Text encoder + durations + alignment: TextEncoder.forward (GeDEX-TTS/model/text_encoder.py:129-146, DEX
 * :126-142: embedding, ConvReluNorm prenet, RetNet in its parallel form with use_softmax = True / use_decay = False — the only
 * setting the shipped configs use —, proj_m, DurationPredictor) and the lines of the TTS forward between the encoder and the
 * decoder (tts.py:37-50: w_ceil, y_lengths, generate_path, mu_y).  Once per utterance; exact-fp32 arithmetic.  Eval mode. */
typedef struct DexText DexText;
typedef struct {
    int32_t variant;                  /* DEX_VARIANT_GEDEX, or DEX_VARIANT_DEX: AdaptiveLayerNorm(sty) after both residual sums of every layer */
    int32_t n_vocab, n_feats, n_channels, filter_channels, filter_channels_dp, n_heads, n_layers, kernel_size;
    int32_t n_spks, spk_emb_dim;      /* n_spks > 1: the speaker embedding is concatenated to the prenet output (RetNet width n_channels + spk_emb_dim) */
    int32_t use_softmax, use_decay;   /* must be 1, 0 */
} DexTextConfig;
typedef struct {
    int32_t B, T;
    const int32_t* tokens_dev;        /* [B,T] token ids */
    const int32_t* lengths_dev;       /* [B], each in [1, T] */
    const float* spk_dev;             /* [B,spk_emb_dim] speaker embedding rows (spk_emb(spk), tts.py:31) when n_spks > 1, else NULL */
    const float* sty_dev;             /* [B,n_channels] pooled style vector (dex_style_encode's sty_enc) for DEX_VARIANT_DEX, else NULL */
    float length_scale;               /* tts.py:38 */
    float* mu_out_dev;                /* [B,n_feats,T]  mu_x */
    float* logw_out_dev;              /* [B,T]          log durations */
    float* w_ceil_out_dev;            /* [B,T]          ceil(exp(logw) * mask) * length_scale */
    int32_t* y_lengths_out_dev;       /* [B]            clamp_min(sum w_ceil, 1) as integers: read these to size the alignment */
    void* workspace_dev; size_t workspace_bytes;
} DexTextArgs;
typedef struct {
    int32_t B, T, Ty;                 /* Ty = fix_len_compatibility(max y_lengths) */
    const float* mu_x_dev;            /* [B,n_feats,T] */
    const float* w_ceil_dev;          /* [B,T] */
    const int32_t* x_lengths_dev;     /* [B] */
    const int32_t* y_lengths_dev;     /* [B] */
    float* mu_y_out_dev;              /* [B,n_feats,Ty] = attn^T mu_x  (the decoder's mu) */
    float* y_mask_out_dev;            /* [B,Ty] */
    float* attn_out_dev;              /* optional [B,T,Ty] 0/1 alignment (generate_path) */
    void* workspace_dev; size_t workspace_bytes;     /* >= B*T floats */
} DexAlignArgs;

int  dex_text_create(const DexTextConfig* cfg, DexText** out);
void dex_text_destroy(DexText* txt);
const char* dex_text_last_error(const DexText* txt);
/* Keys = the reference TextEncoder state-dict names (emb.weight, prenet.*, encoder.layers.<i>.*, encoder.layer_norm.weight,
 * encoder.retnet_rel_pos.angle, proj_m.*, proj_w.*); encoder.retnet_rel_pos.decay is not used (use_decay = 0). */
int  dex_text_num_weights(const DexText* txt);
int  dex_text_weight_info(const DexText* txt, int i, const char** key, int64_t shape[4], int* ndim);
int  dex_text_load_weight_async(DexText* txt, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream);
int  dex_text_finalize(DexText* txt, dex_stream_t stream);
size_t dex_text_workspace_bytes(const DexText* txt, int B, int T);
int  dex_text_encode(DexText* txt, const DexTextArgs* args, dex_stream_t stream);
int  dex_text_align(DexText* txt, const DexAlignArgs* args, dex_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEX_AMD_H */
