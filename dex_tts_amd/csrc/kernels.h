// kernels.h — launch interfaces of the hand-written gfx950 kernels behind libdexamd.so.
// Activations are channels-last fp32: element (b,h,w,c) at base[b*bstride + (h*W + w)*ld + coff + c].
// "step" is the index of the network evaluation inside one sampler call: per-step conditioning tables are indexed as
// table[step*stride + ...].  It is a by-value launch argument (the whole call is one hipGraph with a node per launch; round 1
// replayed a one-step graph and read the index from device memory - a dependent load at the head of every kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dex {

// ---- run-time knobs (A/B switches, test hooks): DEX_* environment variables.  The public entry points take ONE snapshot of every
// registered knob per call (dex_api.hip: KnobSnapshot) and install it for the duration of the enqueue; knob() reads that snapshot,
// so a call sees one consistent setting, nothing calls getenv per launch, and the snapshot as a whole is part of the graph-cache key
// (no hand-kept list of "knobs that matter to captured graphs").  Outside a call (bench tools that launch kernels directly) knob()
// reads the environment.  KNOB_UNSET = the variable is not set.
constexpr int KNOB_UNSET = -2147483647 - 1;
int knob(const char* name);                                  // atoi of the value, or KNOB_UNSET
inline int knob_or(const char* name, int dflt) { const int v = knob(name); return v == KNOB_UNSET ? dflt : v; }
inline bool knob_off(const char* name) { return knob(name) == 0; }          // set to 0: the optional form is switched off
inline bool knob_set(const char* name) { return knob(name) != KNOB_UNSET; }
// compute units of the current device (256 on MI355X); function-local static of an inline function: initialised once, thread-safely
// (three launchers kept their own unsynchronised `static int ncu`)
inline int device_cus() {
    static const int n = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
    return n;
}

// GroupNorm / InstanceNorm statistics: [B][groups][GN_SLOTS][2] partial (mean, mean-of-squares) contributions as 64-bit
// FIXED-POINT integers (2^-36 resolution, see bf16_util.h gn_fix): integer addition is associative, so the native L2
// atomics that accumulate them give the same bits whatever order the workgroups arrive in — every sampler call is bitwise
// reproducible (the first version accumulated fp32 partial sums and differed run to run in the last bits, which bf16
// rounding then amplified to 7e-3).  Slots spread the same-address atomic traffic (~170 ns each on one address).
constexpr int GN_SLOTS = 32;
typedef long long gnfix_t;

// Profiling aid: launch functions that pick one of several template instantiations leave the chosen SYMBOL here, so the
// library's event profile (dex_profile_get) names rows like rocprofv3's kernel trace does instead of by kernel class.
extern thread_local const char* g_last_symbol;

// Precision modes (== DexPrecision of include/dex_amd.h).  The reduced-precision kernels exist twice — namespace dex::bf16
// and dex::f16, the same sources compiled for either operand type (lp_config.h) — and the launch functions below that
// take a `precision` pick the namespace (lp_dispatch.hip).
constexpr int PREC_FP32 = 0, PREC_BF16 = 1, PREC_FP16 = 2, PREC_FP16X2 = 3;      // FP16X2: fp16 operands, weights as hi + lo (lp_config.h)
inline bool prec_wsplit(int precision) { return precision == PREC_FP16X2; }
// the mode of the C-ABI call being built on this thread (set by dex_api.hip): the shape predicates of lp_dispatch.hip answer for it
extern thread_local bool g_lp_wsplit;
inline bool prec_is_lp(int precision) { return precision != PREC_FP32; }

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / linear:  C[m, n] = epi( sum_k gather(A)[m,k] * W[k,n] )
//   m = (ho, wo) in an Ho x Wo grid per batch; k = tap*Cin + c; tap=(kh,kw);
//   hi = ho*sh + off_h + kh*step_h, wi = wo*sw + off_w + kw*step_w (zero outside [0,Hi)x[0,Wi)).
//   blockIdx.z = (b*groups + g)*ksplit + s.
struct IGemmP {
    const float* A; int lda; long a_bstride; int a_coff;
    int Hi, Wi, Cin;
    int KH, KW, sh, sw, off_h, off_w, step_h, step_w;
    int Ho, Wo;
    const float* W; long w_bstride; long w_gstride;   // packed [K][N] (per group), optional per-batch
    const void* Wbf;                                   // bf16 copy, packed [N][K] (K contiguous), or null
    const void* Wfrag = nullptr;                       // the same 16-bit weights in MFMA fragment order (launch_pack_lp_frag of the [K][N] matrix), or null:
                                                       // the column walker's LDS-DMA form (igemm_bf16.hip); its lo half sits lo_off(Wfrag) elements behind (unused there)
    long w_lo_off = 0;                                 // split-weight mode (PREC_FP16X2): elements from a weight of Wbf to its lo half, 0 = none
    int N, K, ksplit, groups;
    const float* bias; long bias_bstride;              // [groups*N] or null; optional per-batch stride
    float* C; int ldc; long c_bstride; long c_sstride; int c_coff;
    int OHf, OWf, osh, osw, oh0, ow0;
    const float* inmask; int inmask_ws;
    const float* outmask; int outmask_ws;
    long mask_bstride;
    int act;                                           // 0 none, 1 GELU(erf), 2 ReLU
    float act_in_slope;                                // != 0: leaky_relu(x, slope) on the gathered A elements (fp32 kernel and the looped lp kernel:
                                                       // the vocoder's "x = leaky_relu(x); x = conv(x)", hifigan/models.py:98-103)
    const float* gate; int gate_nstride; long gate_step_stride;
    const float* res; int ldres; long res_bstride; int res_coff;
    int step;
    int unpatch_s, unpatch_C;                          // >0: scatter rows (f,w) x cols (p1,p2,c) -> NHWC image
    int parity;                                        // 1: blockIdx.z = b*4 + (ph*2+pw): ConvTranspose2d(4,2,1) as four 2x2-tap
                                                       // sub-convolutions in ONE launch (off/oh0/ow0 = parity, weights += par*K*N)
    gnfix_t* gn_stats; int gn_groups, gn_cpg;          // fused GroupNorm partial statistics of (acc + bias), or null
    int stats_final;                                   // 1: the statistics are of the STORED value (after activation, gate,
                                                       // residual, mask) - InstanceNorm of the next adaptor (cpg = 1)
    const float* ln_shift; const float* ln_scale; long ln_step_stride;   // fused LayerNorm(eps 1e-6)+modulate on the A rows
                                                       // (single-shot bf16 kernel only; needs K == Cin == row length)
    int B;
    // reduced-precision tensors in HBM (1 = bf16, 2 = fp16; 0 = fp32).  a_lp: A holds 16-bit elements (same lda / strides, in
    // elements); c_lp: C is stored as 16-bit elements.  Only the looped reduced-precision kernel implements them (igemm_bf16.hip);
    // used for activations whose every consumer rounds them to the MFMA operand type anyway (dex_api.hip, lp_inter).
    int a_lp, c_lp;
};
void launch_igemm(const IGemmP& p, int precision, hipStream_t st);

// Upsample = ConvTranspose2d(64, 64, 4, 2, 1) on x * mask as a strip-walking kernel (convt_up.hip, reduced-precision modes).
// X [B][H][W][ldx] (fp32, or 16-bit when a_lp), Y [B][2H][2W][ldy] (fp32, or 16-bit when c_lp); strides / offsets in elements.
// Wfrag[par] = the parity's matrix [K = (th*2+tw)*64 + ci][64 co] (pack_convt_kernel) in MFMA fragment order (launch_pack_lp_frag).
struct ConvTUpP { const void* X; int a_lp; int ldx; long xb; int x_coff; int H, W;
                  const void* Wfrag[4]; const float* bias;
                  void* Y; int c_lp; int ldy; int y_coff;
                  const float* inmask; int inmask_ws; long mask_bstride; int B;
                  int nseg, rows_per_wg; };               // filled by the launcher
bool convt_up_supported(int C, int H, int W, int ldx, int ldy);
// Downsample = Conv2d(64, 64, 3, 2, 1) on x * mask as a strip-walking kernel (conv_down.hip; reduced-precision modes, 16-bit input).
// X [B][H][W][ldx] (fp32, or 16-bit when a_lp; + x_coff), Y [B][H/2][W/2][ldy] (fp32, or 16-bit when c_lp); strides / offsets in elements.
// Wfrag = the weight matrix [K = (kh*3+kw)*64 + ci][64 co] in MFMA fragment order (launch_pack_lp_frag).
struct ConvDownP { const void* X; int a_lp; int ldx; long xb; int x_coff; int H, W;
                   const void* Wfrag; const float* bias;
                   void* Y; int c_lp; int ldy; int y_coff;
                   const float* inmask; int inmask_ws; long mask_bstride; int B;
                   int nseg, rows_per_wg; };              // filled by the launcher
bool conv_down_supported(int C, int H, int W, int ldx, int ldy, int x_coff);
void launch_conv_down(const ConvDownP& p, int precision, hipStream_t st);
void launch_convt_up(const ConvTUpP& p, int precision, hipStream_t st);

// Patch-staged 3x3/s1/p1 Block convolution on bf16 MFMA (conv3x3_bf16.hip).  Input transform while staging:
// x*mask, or mask*(Mish(GroupNorm(x)) + tadd[step]) when pro_stats != null (the fused tail of block1).
struct Conv3P {
    const float* X; int ldx; int x_coff; int H, W, Cin, Cout;
    const void* Wbf; const float* bias; float* Y;            // bf16 [Cout][9*Cin]; Y is [B,H,W,Cout] contiguous
    const void* Wfrag;                                       // the same weights in MFMA fragment order ([K = 9*Cin][Cout] through launch_pack_lp_frag), or null
    const float* mask; int mask_ws; long mask_bstride;
    const gnfix_t* pro_stats; const float* pro_gamma; const float* pro_beta; const float* pro_tadd;   // tadd may be null
    // second prologue form (pro_res != null): the input is the preceding ResnetBlock's tail,
    //   x = mask * Mish(GN(X)) + pro_res   (res_conv shortcut, diffusion.py:67-71), [H*W][Cin] like X;
    // the kernel also writes x for its own output pixels to pro_xout ([H*W][Cin]) for the later consumers.
    const float* pro_res; float* pro_xout;
    int xout_lp;                                             // pro_xout is written in the mode's 16-bit type (strip forms only: its reader is LinKvCtxP::res_lp)
    // pro_res recomputed instead of read (res2_w != null, pro_res == null; Cin == 64): the shortcut of the U-Net's FIRST
    // ResnetBlock is res_conv((mu, c_in*x[, spk]) * mask), a 1x1 conv of 2-3 input planes (diffusion.py:70,171-175) - two or
    // three FMAs per value from planes that stay in cache, against 4 B written by the first conv and read back here.
    // res2_w = fp32 [planes][64] (FirstConvP::W1), res2_b [64]; planes [B][H][W] (spk: [B][H]); c_in = res2_scal[step*stride+2].
    const float *res2_w, *res2_b, *res2_mu, *res2_x, *res2_spk, *res2_scal; int res2_scal_stride, res2_planes;
    // fused 1x1 shortcut of the ResnetBlock (res_conv(x * mask), diffusion.py:70): a second output computed from the
    // patch's centre tap; res_w = bf16 [Cout][Cin], res_y = [H*W][Cout] fp32
    const void* res_w; const float* res_b; float* res_y;
    const void* res_wfrag;                                   // res_w in MFMA fragment order ([K = Cin][Cout] through launch_pack_lp_frag), or null
    int step; gnfix_t* gn_stats; int B;
    long long* dbg;                                          // optional phase timestamps (tools/kbench)
    // raw conv outputs that only a GroupNorm prologue reads next (h1, h2 of a ResnetBlock) may live in HBM as bf16:
    // x_bf16: X is bf16 [.. ldx] (PRO / PRO2 forms only); y_bf16: Y is written as bf16.  Statistics stay fp32.
    int x_bf16, y_bf16;
    long w_lo_off = 0, res_lo_off = 0;                       // split-weight mode (PREC_FP16X2): elements from a weight of Wbf / res_w to its lo half
    int skip_dead = 0;                                       // set by the launchers (DEX_CONV_SKIP_DEAD): tiles whose whole input patch lies in an utterance's padding
                                                             // (mask 0 in every column) skip loads, prologue and MFMAs - conv(0) + bias is what they would compute
};
bool conv3x3_bf16_supported(int Cin, int Cout);
bool conv3x3_bf16_tail_supported(int C);     // pro_res form (Cin == Cout == C)
bool conv3x3_bf16_res_supported(int Cin, int Cout);   // res_w form (fused 1x1 shortcut)
bool conv3x3_bf16_xb_supported(int Cin, int Cout);    // x_bf16 form (bf16 input under a GroupNorm prologue)
bool conv3x3_plain_lp_in_supported(int H, int W, int B, int Cin, int Cout);   // a plain (no prologue) conv of this shape can read a 16-bit input
bool igemm_nwalk_form(const IGemmP& p);                // the column-walking unpatchify GEMM takes this launch (implements IGemmP::c_lp for the scatter)
bool conv3x3_cat_lp_in_supported(int H, int W, int B, int Cin, int Cout);   // the up path's fused-shortcut conv has a 16-bit-input form at this grid
bool conv3x3_strip_form(const Conv3P& p);              // this launch runs on one of the strip-walking throughput forms (the ones that implement Conv3P::xout_lp)
bool conv3x3_res2_form(int H, int W, int B);           // a 64 -> 64 fused-tail conv of this grid runs on the form that implements res2_*
void launch_conv3x3_lp(const Conv3P& p, int precision, hipStream_t st);   // picks the strip-streaming form (conv3x3_stream.hip) for large grids

// First ResnetBlock of the U-Net: 3x3 conv and 1x1 res_conv straight from the stacked input planes
// (mu, c_in*x[, spk]) * mask  (diffusion.py:171-175,185; edm.py:96).
struct FirstConvP {
    const float* mu; const float* x; const float* spk;    // [B,80,T]; spk plane [B,80] or null
    const float* mask; int B, H, T, planes, C;
    const float* W3; const float* b3;                     // [planes*9][C], [C]
    const float* W1; const float* b1;                     // [planes][C], [C]
    const float* scal; int scal_stride; int step;  // per-step scalars; scal[step*stride + 2] = c_in
    float* h1; float* res;                                // [B,H,T,C] each
    int h1_bf16;                                          // h1 stored as bf16 (1) / fp16 (2): its only reader is the next conv's GN prologue
    gnfix_t* gn_stats;                                    // fused GroupNorm partials of h1 (8 groups, slot-spread) or null
};
void launch_first_conv(const FirstConvP& p, hipStream_t st);

// GroupNorm statistics: per (b, group) sum / sum-of-squares in fp64 (biased variance later).
struct GnStatsP { const float* X; int ld; long bstride; int npix; int C; int groups; gnfix_t* stats; int B; };
void launch_gn_stats(const GnStatsP& p, hipStream_t st);

// y = mask*(Mish(GN(x)) + tadd[c]) + res   (Block / ResnetBlock tails, diffusion.py:41-50,66-71)
struct GnApplyP {
    const float* X; int ldx; long xb;
    float* Y; int ldy; long yb; int y_coff;
    int npix, W, C, groups; const gnfix_t* stats; const float* gamma; const float* beta;
    const float* mask; int mask_ws; long mask_bstride;
    const float* tadd; long tadd_step_stride; int step;
    const float* res; int ldres; long resb; int res_under_mask;   // 1: y = mask*(mish + tadd + res)
    int B;
};
void launch_gn_apply(const GnApplyP& p, hipStream_t st);

// final_block tail + final_conv + EDM combine + Euler update (diffusion.py:204-207, edm.py:97,202-208)
struct FinalP {
    const float* X; long xb; int npix, W, C, groups; const gnfix_t* stats; const float* gamma; const float* beta;
    const float* mask; long mask_bstride;
    const float* wfc; const float* bfc;                   // final_conv weight [C], bias [1]
    const float* xcur;                                    // [B,80,T] current sampler state (x_hat)
    float* denoised;                                      // optional D_x out
    float* xnext;                                         // optional Euler update out (may alias xcur)
    const float* scal; int scal_stride; int step;
    float* zero_ptr; long zero_n;                         // optional: clear the OTHER parity's GroupNorm statistics arena
    int B;
    // Heun (edm.py:207-214).  mode 0: Euler update with h = sigma_next - sigma (or htab[step] when given);
    // mode 1: predictor - also stores the slope d_cur in dbuf (xnext receives x' = x_hat + h d_cur);
    // mode 2: corrector - xcur is x', xhat the state the step started from: xnext = xhat + h (0.5 d_cur + 0.5 d').
    int mode; const float* htab; float* dbuf; const float* xhat;
    int x_bf16;                                           // X (the final conv's raw output) is bf16 (1) / fp16 (2)
    const int* poison;                                    // optional device word: non-zero (a workgroup hand-off of this call timed out) -> the outputs are NaN
};
void launch_final(const FinalP& p, hipStream_t st);
// "Increase noise temporarily" tables (edm.py:194-196, schedule 'linear', scaling 'none'): for the schedule t_0..t_N
//   t_hat[i] = t_i + gamma_i t_i,  gamma_i = min(S_churn / n, sqrt(2) - 1) if S_min <= t_i <= S_max else 0  (t_hat[n] = 0),
//   h[i] = t_{i+1} - t_hat[i],  ncoef[i] = sqrt(max(t_hat^2 - t_i^2, 0)) * S_noise      — fp32, in the reference's op order.
void launch_churn_tables(const float* sigmas, int n, float S_churn, float S_min, float S_max, float S_noise,
                         float* t_hat, float* h, float* ncoef, hipStream_t st);
// x_hat = x + ncoef[i] * noise (two rounded fp32 ops like the reference's mul then add)
void launch_add_noise(float* x, const float* noise, const float* ncoef_i, long n, hipStream_t st);
// Heun evaluation tables from t_hat_0..t_hat_{n-1} and the step sizes h_i = t_{i+1} - t_hat_i:
// sig[2i] = t_hat_i, sig[2i+1] = t_hat_i + h_i (i < n-1), sig[2n-1] = 0; hout[2i] = hout[2i+1] = h_i.
void launch_heun_expand(const float* t_hat, const float* h, int n, float* sig, float* hout, hipStream_t st);

// Linear attention (diffusion.py:82-92): qkv [B,n,3*heads*32]; softmax over positions on k.
struct LinAttnCtxP { const float* qkv; int ld; long bstride; int n; int heads; int chunk; int nchunks;
                     float* part_m; float* part_s; float* part_c; int B; };
void launch_linattn_ctx(const LinAttnCtxP& p, hipStream_t st);
struct LinAttnCombineP { const float* part_m; const float* part_s; const float* part_c; int nchunks; int heads;
                         const float* Wout;   // [C][heads*32] reference layout (to_out.weight)
                         const float* g;      // Rezero scalar
                         int C; float* Weff;  // [B][heads*32][C] : g * sum_e ctx[d,e] * Wout[c, h*32+e]
                         int B; };
void launch_linattn_combine(const LinAttnCombineP& p, hipStream_t st);

// Fused linear attention for the bf16 mode (linattn_fused.hip): k/v + online softmax + ctx without q,k,v in HBM
struct LinKvCtxP { const float* X; int ldx; int x_coff; long xb; int npix; int C; const void* Wkv; int nsub; int nblk;
                   float* part_m; float* part_s; float* part_c; int B;
                   // optional fused tail of the preceding ResnetBlock (H2 != null): x = mask*(Mish(GN(H2)) [+ res]) [+ res]
                   // is computed while the tile is loaded, written to Xout ([npix][C]) for the later consumers, and X is
                   // not read (diffusion.py:49,67-71).  gn_stats: slot-spread partials of H2; W/mask_ws: mask column map.
                   const float* H2; const gnfix_t* gn_stats; const float* gamma; const float* beta;
                   const float* res; int ldres; long resb; int res_under_mask;
                   const float* mask; int mask_ws; long mask_bstride; int W; float* Xout;
                   int h2_bf16;                             // H2 is bf16 [npix][C]
                   int res_lp;                              // res is stored in the mode's 16-bit type [npix][C] (written by Conv3P::xout_lp)
                   int xout_lp;     long wkv_lo_off = 0;        // split-weight mode: elements from a weight of Wkv to its lo half
};                          // Xout is written in the mode's 16-bit type (its one reader, the tail kernel, takes LinOut2P::x_lp)
void launch_linattn_kvctx(const LinKvCtxP& p, int precision, hipStream_t st);
struct LinMergeP { const float* part_m; const float* part_s; const float* part_c; int nblk;
                   const float* Wout; const float* g; int C; void* W2; int B; };       // Wout fp32 [C][128]
void launch_linattn_merge(const LinMergeP& p, int precision, hipStream_t st);
struct LinOut2P { const float* X; int ldx; int x_coff; long xb; int npix; int C; const void* Wq; const void* W2;
                  const float* bias; float* Y; int ldy; int y_coff; long yb; int B; // Wq bf16 in MFMA fragment order (launch_pack_lp_frag_nk)
                  int y_lp;         // 1: Y is stored in the mode's 16-bit type (throughput form only: linattn_out2_lp_out_supported)
                  void* Y2; int ldy2; int y2_coff; long y2b;   // optional second copy of Y in the mode's 16-bit type (throughput form only): the skip half of the up path's concatenation buffer
                  int x_lp;     long wq_lo_off = 0;         // split-weight mode: elements from a weight of Wq to its lo half
                  };      // 1: X is stored in the mode's 16-bit type [npix][C] (throughput form only; written by LinKvCtxP::xout_lp)
void launch_linattn_out2(const LinOut2P& p, int precision, hipStream_t st);
bool linattn_out2_lp_out_supported(int npix, int B);
bool linattn_fused_supported(int C);          // the fused linear attention exists for this mode (lp_dispatch.hip: false in a build without a split-weight form)

// Depthwise patch-embed conv + SiLU (dit.py:57-58), channels-last, zero padding incl. right pad to patch multiple.
struct DwConvP { const float* X; int ldx; long xb; int Hi, Wi, C; int k, s, pad; const float* Wd; const float* bd;
                 const float* mask; int mask_ws; long mask_bstride;
                 float* Y; int Hf, Wt; int B;
                 const float* aff = nullptr; };        // optional per-(utterance, channel) affine on the input, [B][2][C] (scale row, shift row): the DEX TIV
                                                       // adaptor's y = IN2d(x) * s + m folded into the load (launch_tiv_coef) - zero padding applies to y
void launch_dwconv_silu(const DwConvP& p, hipStream_t st);
// the depthwise conv + SiLU + pointwise GEMM of PatchEmbed2D as ONE launch (reduced-precision modes, small grids: patch_embed.hip);
// Wb = the pointwise weight's 16-bit [hidden][C] twin
bool patch_embed_fused_supported(int k, int C, int hid, long ntok);
void launch_patch_embed_fused(const DwConvP& p, const void* Wb, const float* bias, float* emb, int hid, int precision, hipStream_t st);

// pos-conv tail: sum split-K partials + bias -> GELU -> mean over freq -> tokens = emb + pos + freq_pos (dit.py:450-454)
// Direct grouped 16x16 positional convolution (pos_conv.hip): X = patch embedding [B][Hf][Wt][hid] fp32, Wf = bf16
// weights in MFMA fragment order per group ([g][tap][ks][lane][8], launch_pack_lp_frag of each group's [K][32]
// matrix), Y = raw convolution sums [B][Hf*Wt][hid].
struct PosConvP { const float* X; const void* Wf; float* Y; int Hf, Wt, hid, G, B;
                  int ncol; };      // filled by the launcher: 1 = the last column is computed by column workgroups (pos_conv.hip)
bool pos_conv_direct_supported(int hid, int groups, int kernel, int Hf);
void launch_pos_conv_direct(const PosConvP& p, int precision, hipStream_t st);
struct PosFinishP { const float* part; int nsplit; long split_stride; const float* bias; const float* emb;
                    const float* freq_pos; float* tok; int Hf, Wt, D; int B;
                    int cg, cg_pad; };      // > 0: the partials are [rows][G][cg_pad] with cg live channels per group (padded groups)
void launch_pos_finish(const PosFinishP& p, hipStream_t st);
// [rows][G][cg] -> [rows][G][cg_pad], zero-filled tail (grouped pos-conv whose groups are no multiple of 32 channels wide)
void launch_group_pad(const float* src, float* dst, long rows, int G, int cg, int cg_pad, hipStream_t st);

// LayerNorm(eps 1e-6, no affine) + modulate (dit.py:78-79,288-289,330)
struct LnModP { const float* X; float* Y; int rows_per_batch; int D; const float* shift; const float* scale;
                long step_stride; int step; int B; };
void launch_ln_mod(const LnModP& p, hipStream_t st);

// Row-local remainder of a DiT block + the next block's qkv projection in one launch (dit_rowchain.hip; bf16 mode,
// hidden 256 / mlp 512).  Weights are bf16 in MFMA fragment order (launch_pack_lp_frag).
struct DitChainP { const float* O; int ksplit; long o_sstride; const float* ml; int heads; int rows_per_batch;   // attention partials
                   float* X; const void *Wp, *W1, *W2, *Wq; const float *bp, *b1, *b2, *bq;
                   const float* ada;                       // this block's [n_steps][6*hidden] adaLN table
                   const float *next_shift, *next_scale; long next_step_stride;   // null: no qkv stage (last block)
                   void *Qh, *Kh, *Vt; int Npad; float qscale;          // bf16 q, k, v^T in fragment order (attention_direct.hip): written
                   const void *Qin, *Kin, *Vin;                          // attn_inline: operands READ by the in-kernel attention — a different
                                                                        // buffer set than the one written (workgroups of one launch overlap)
                   int qkv_only;                                        // 1: only LN+modulate+qkv of X (first block)
                   int attn_inline;                                     // 1: the attention core runs inside this launch (O / ml unused)
                   int step; int M; int B; long long* dbg;     // M = B * rows_per_batch
                   // cluster form (small grids, dit_rowchain_cluster_kernel): exchange slabs, flags (zeroed once per call) and the
                   // launch's epoch (unique within the call, never 0); err: device word set when a hand-off wait timed out
                   float* xslab; unsigned* xflag; unsigned epoch; int* xerr;
                   int o_lp;                                // O is stored in the mode's 16-bit type [M][hidden] (ksplit == 1, 64-row form; AttnDirectP::o_lp)
                   int xcd_map;                             // set by the launcher: row tiles of batch element b run on XCD b % 8 (in-kernel attention, B % 8 == 0)
                   int xlocal;                              // 1: the members of a cluster share an XCD (hand-offs through its L2; grid padded to rounds of 8 clusters)
                   int xdrop;
                   int tail_row0 = 0, tail_ks = 0;
                   int xcds = 8; };            // xlocal: the clusters are dealt to the first `xcds` XCDs only (workgroups of the others leave at once): fewer L2s fetch the weights and K / V^T      // 64-row form: rows from tail_row0 on are merged from tail_ks fp32 partials in O slots 1.. (AttnDirectP::tail_g); 0 / 1 = off                            // tests only (DEX_DEBUG_DROP_HANDOFF): 1 member 3 never raises its flags -> the peers' waits time out; 2 L2-scope hand-offs across XCDs
// cluster form of the row chain: workgroups per 32-row tile, bytes of exchange slab / flag words per tile, and whether a launch
// of B x N rows takes it (all workgroups co-resident: <= one per CU)
constexpr int DIT_CLUSTER = 4;
constexpr size_t DIT_CLUSTER_SLAB_FLOATS = 2 * DIT_CLUSTER * (32 * 256 + 256);
constexpr size_t DIT_CLUSTER_FLAG_WORDS = 2 * DIT_CLUSTER;
bool dit_rowchain64_form(int rows_per_batch, int B, int attn_inline);     // the 64-row batch form takes this launch (the one that implements DitChainP::o_lp)
bool dit_rowchain_cluster_form(int rows_per_batch, int B);
bool dit_rowchain_cluster_local_fits(int rows_per_batch, int B);
int dit_rowchain_cluster_xcds(int rows_per_batch, int B);          // XCDs the XCD-local clusters are dealt to (DitChainP::xcds): 8 unless DEX_DIT_XCDS packs them
bool dit_rowchain_supported(int hidden, int mlp_hidden);
// Softmax attention on the row chain's bf16 operands (2 heads x 128): no staging, K / V^T / Q fragments are read
// straight from global memory.  O: fp32 [ksplit][B][N][256] partials + ml (merged by the next row chain launch).
struct AttnDirectP { const void *Qh, *Kh, *Vt; int N, Npad, B; float* O; long o_sstride; float* ml; int ksplit; long long* dbg;
                     int o_lp;              // O (ksplit == 1, shared-ring kernel) is written in the mode's 16-bit type: its reader, the 64-row chain, rounds it so anyway (DitChainP::o_lp)
                     int xcd_map;           // set by the launcher (shared-ring kernel): 1-D grid, the query groups of one (element, split, head) share an XCD
                     int tail_g = 0, tail_ks = 0;      // 64-query form, ksplit == 1: query groups (256 rows) from tail_g on are split tail_ks ways (fp32 slots 1.., ml); 0 / 1 = off
                     int half_g = 0, half_n = 0; };    // 64-query form, set by its launcher: half_g whole query groups per (element, head), then half_n HALF units (4 waves x one 32-query block); half_n == 0 = off
void launch_attention_direct(const AttnDirectP& p, int precision, hipStream_t st);
bool attention_direct_batch_regime(int N, int B);     // shared-ring kernel (many query tiles) vs key-splitting waves (few)
int attention_direct_ksplit(int N, int B);             // key split the batch regime wants for an even load
void launch_attention_q64(const AttnDirectP& p, int precision, hipStream_t st);     // attention_q64.hip: the batch / long-form form (64 queries per wave)
int attention_q64_ksplit(int N, int B, int max_split);
void attention_q64_plan(int N, int B, int max_split, int* ks, int* tail_g, int* tail_ks);
bool attention_q64_half_plan(int N, int B, int* half_g, int* half_n);     // whole units + a shorter round of half units (no merge anywhere); false: whole units only
void launch_dit_rowchain(const DitChainP& p, int precision, hipStream_t st);
void launch_pack_lp_frag(const float* src, void* dst, int K, int N, int precision, hipStream_t st);
void launch_pack_lp_frag_nk(const float* src, void* dst, int K, int N, int precision, hipStream_t st);   // source [N][K]

// Softmax attention, head_dim 128, no key mask except kv_len (timm Attention core / TVAdaptor core).
struct AttnP { const float* Q; int ldq; long qb; const float* K; int ldk; long kb; const float* V; int ldv; long vb;
               float* O; int ldo; long ob; int Nq, Nk; const int* kv_len; int kv_len_add; int heads; float scale; int B;
               // key-split partials (bf16 split kernel only; 0/1 = off): split s handles a contiguous range of key
               // tiles, writes its normalised O to O + s*o_sstride and (running max, row sum) to
               // ml[((s*B + b)*heads + h)*Nq + q][2]; the consumer merges (dit_rowchain.hip)
               int ksplit; long o_sstride; float* ml;
               long long* dbg;                               // DEX_TIMING builds only
               int head_dim;                                 // 0 = 128 (the tuned kernels); 64 / 192 / 256 run the generic fp32 kernel
               int force_generic;                            // tests: the generic kernel at head_dim 128 too
               int o_lp; };                                  // O is written in the mode's 16-bit type (shared-K/V reduced-precision form only: attention_lp_shared_form)
// DiT blocks with 512 <= N < 1024 tokens at batch size (GeDEX B = 32: N = 650): the 64-query attention as its own launch + the 64-row
// generated streams take the block when the B x ceil(N / 64) row tiles fill at least 65 % of their rounds of 256 persistent workgroups -
// measured at 352 tiles (69 %): +0.9 % end to end over attention fused into the 32-row chain (profiles/round5_gedex_b32_separate_attention_ab.txt),
// and the gap grows with the fill.  (Round 6: the split-weight build has generated streams too and takes the same route.)
inline bool dit_sep64_small_n(int N, int B) {
    if (N < 512 || N >= 1024) return false;
    const long t = (long)B * ((N + 63) / 64), r = (t + 255) / 256;
    return t >= 256 && t * 100 >= r * 256 * 65;
}
bool attention_lp_shared_form(int Nq, int heads, int B, int ksplit);
bool attention_head_dim_supported(int hd);
void launch_attention(const AttnP& p, int precision, hipStream_t st);

// y[r, n] = act_out( bias[n] + sum_k act_in(x[r,k]) * W[n,k] )   (tiny conditioning MLPs; W in reference layout)
struct SmallLinP { const float* X; int ldx; int rows; int K; const float* W; const float* bias; int N;
                   float* Y; int ldy; int act_in; int act_out; };   // act: 0 none, 1 mish, 2 silu
void launch_small_linear(const SmallLinP& p, hipStream_t st);

// Per-step EDM scalars + sinusoidal features from the sigma table (edm.py:90-94, diffusion.py:110-117, dit.py:250-254)
struct CondPrepP { const float* sigmas; int n; float pe_scale; int dim; float* scal; int scal_stride;
                   float* t_unet; float* t_dit; };
void launch_cond_prep(const CondPrepP& p, hipStream_t st);

// misc elementwise
void launch_scale_copy(const float* src, float* dst, long n, const float* scal_ptr, hipStream_t st); // dst = src * *scal_ptr
void launch_step_reset(int* step, hipStream_t st);
void launch_step_inc(int* step, hipStream_t st);
void launch_iota(int* dst, int n, hipStream_t st);
void launch_permute4(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                     hipStream_t st);   // dst = src.permute(p0..p3).contiguous()
void launch_f32_residual_lp(const float* src, float* dst, long n, int precision, hipStream_t st);   // src - float(lp(src)): the lo half of a split weight
void launch_f32_to_lp(const float* src, void* dst, long n, int precision, hipStream_t st);        // fp32 -> bf16 / fp16, round to nearest even
void launch_pack_lp_nk(const float* src, void* dst, int K, int N, int precision, hipStream_t st);  // fp32 [K][N] -> low precision [N][K]
void launch_spk_plane(const float* spk_out, float* plane, int B, int F, hipStream_t st);

// DEX style adaptors -------------------------------------------------------------------------
// per-row mean / sqrt(unbiased var + eps) over the last dim (InstanceNorm1D.cal_stats, base.py:72-78)
void launch_row_stats(const float* X, int B, int C, int len, float eps, float* mean, float* std, long out_bstride, hipStream_t st);
// per-(b,c) sum / sumsq over pixels (InstanceNorm2D statistics, base.py:95-103), fp64 atomics
// statistics: [B][C][IN_SLOTS][2] fixed-point (mean, mean-of-squares) contributions, native L2 integer atomics spread over
// the slots (the first version used fp64 LDS + global atomics: 170 ns each on one address, 14 us per call at B=1)
constexpr int IN_SLOTS = GN_SLOTS;     // same layout as the GroupNorm partials: a GEMM epilogue can produce them (cpg = 1)
struct InStatsP { const float* X; int ld; long bstride; int npix; int C; gnfix_t* stats; int B;
                  const float* mask; int mask_ws; long mask_bstride; int W; };   // optional x*mask on load
void launch_in_stats(const InStatsP& p, hipStream_t st);
// SelfAttentionPooling for all steps (ref_encoder.py:246-253): out[step][b][C]
struct SapP { const float* t_tok; int t_ld; int t_coff; int nsteps; const float* stats; int L; int C;
              const float* w; const float* bias; float* out; int B; };
void launch_sap(const SapP& p, hipStream_t st);
// TV: fold InstanceNorm2D into w_q:  Weff[b][k][n] = rstd[b,k]*Wq[n,k];  beff[b][n] = -sum_k mean*rstd*Wq[n,k]
struct InFoldP { const gnfix_t* stats; int npix; float eps; const float* Wq; int C; float* Weff; float* beff; int B;
                 void* Wbf; int lp; int split = 0; }; // split = 1 (lp = 2): also the lo halves, fp16(v - fp16(v)), B * C * C elements behind (PREC_FP16X2).  reduced-precision modes: write rstd[k] * Wq[n][k] as bf16 (lp = 1) / fp16 (lp = 2) [B][n][k]
                                       // (the MFMA GEMM's weight layout) instead of Weff
void launch_in_fold(const InFoldP& p, hipStream_t st);
// TIV: y = IN2d(x)*s + m  (ref_encoder.py:271); s,m indexed [step][b][C]
struct TivApplyP { const float* X; int ld; long xb; float* Y; int ldy; long yb; int npix; int C; const gnfix_t* stats;
                   float eps; const float* s_tab; const float* m_tab; int step; int B; };
void launch_tiv_apply(const TivApplyP& p, hipStream_t st);
// the same transform as per-channel coefficients for a consumer that applies it on load: aff [B][2][C] (y = x * aff[b][0][c] + aff[b][1][c]; p.X / p.Y unused)
void launch_tiv_coef(const TivApplyP& p, float* aff, hipStream_t st);
// write per-step time-token rows into K/V row 0 (ref_encoder.py:157)
struct TvRow0P { const float* k0; const float* v0; int step; float* K; float* V; long kvb; int C; int B;
                 float* zero_ptr; long zero_n;             // optional: clear the IN2d statistics for their next use
                 void* Kp = nullptr; void* VTp = nullptr; int NkPad = 0; int lp_kind = 0; };   // optional: row 0 of the one-launch adaptor's 16-bit operands too (TvKvPrepP layouts, C = 128)
void launch_tv_row0(const TvRow0P& p, hipStream_t st);
// The TV adaptor as ONE launch in the batch regime (attention_bf16.hip, reduced-precision modes; ref_encoder.py:154-179):
//   out = mask * (x + linear(softmax((IN2d(x) W_q^T / sqrt(C)) K^T) V))   with the InstanceNorm folded into a per-utterance W_eff, b_eff
// (launch_in_fold) - q projection, attention over the Ts + 1 style keys, output projection, residual, mask and the TIV adaptor's
// InstanceNorm statistics of the result per 128-pixel workgroup; q and the attention output never exist in HBM.
// TvKvPrepP: K / V (fp32 [B][Nk][C]) -> 16-bit operands in MFMA FRAGMENT order, as the chain's LDS-DMA ring takes them (per utterance and
// 64-key tile sixteen 1-KB pieces each): Kp piece (st, ks), lane (i, hh) = key tile * 64 + st * 32 + i, positions ks * 16 + hh * 8 .. + 8 of
// the key row with the channels of every 16-group in accumulator order; VTp piece (t, q), lane (i, hh), element e = channel t * 32 + i of key
// tile * 64 + (q / 2) * 32 + (q % 2) * 16 + key_pos(hh * 8 + e); keys >= Nk are zeros.  C = 128.
struct TvKvPrepP { const float* K; const float* V; long kvb; int Nk; int NkPad; void* Kp; void* VTp; int B; };
struct TvChainP { const float* X; int ldx; int x_coff; long x_bstride; int npix; int Wm;
                  const float* mask; int mask_ws; long mask_bstride;
                  const void* Weff; long weff_lo_off; const float* beff;      // 16-bit [B][C][C] ([n][k]; lo halves weff_lo_off elements behind, 0 = none), fp32 [B][C]
                  const void* Wl; long wl_lo_off;                             // 16-bit [C][C] ([n][k])
                  const void* Kp; const void* VTp; int NkPad; int Nk; const int* kv_len; int kv_len_add; float scale;
                  float* out; gnfix_t* stats; int B;
                  // folded form (tv_chain_fold_kernel; TvFold2P): xmean != null -> Kp holds K' and VTp holds V'^T, both in MFMA FRAGMENT order (per
                  // utterance and 64-key tile sixteen 1-KB pieces: K' piece (st, ks) lane (i, hh) = K'[tile * 64 + st * 32 + i][ks * 16 + hh * 8 .. + 8];
                  // V'^T piece (t, q) lane (i, hh) element e = V'[key tile * 64 + (q / 2) * 32 + (q % 2) * 16 + key_pos(hh * 8 + e)][t * 32 + i]), x is
                  // centred with xmean [B][C] on load, Weff / beff / Wl are not read; zero_ptr: zero_n floats (even) cleared by the launch
                  const float* xmean = nullptr; float* zero_ptr = nullptr; long zero_n = 0; };
// Per-step operands of the folded one-launch TV adaptor (dex_elem.hip tv_fold2_kernel; replaces launch_in_fold + launch_tv_row0 there):
//   Kp[b][key][k]  = rstd[b,k] * scale * G[b][key][k]   (key 0: g0[step][k]; keys >= Nk: 0)   G = K W_q   (fp32 [B][Nk][C], row 0 unused)
//   VTp: key 0     = v0p[step][c]                        (the time token's column of V'^T = (V W_l^T)^T; the style columns are written once per
//                                                         call by launch_tv_vfrag_prep on V'); both operands in fragment order (TvChainP)
//   xmean[b][k]    = mean[b,k]
// and clears zero_n floats at zero_ptr (the TIV statistics: their last reader was the previous step's launch_tiv_coef).  C = 128.
struct TvFold2P { const gnfix_t* stats; int npix; float eps; const float* G; long gb; const float* g0; const float* v0p; int step;
                  int Nk; int NkPad; int C; float scale; void* Kp; void* VTp; float* xmean; float* zero_ptr; long zero_n; int lp_kind; int B; };
void launch_tv_fold2(const TvFold2P& p, hipStream_t st);
bool tv_chain_form(int npix, int C, int B);
void launch_tv_kv_prep(const TvKvPrepP& p, int precision, hipStream_t st);
void launch_tv_vfrag_prep(const TvKvPrepP& p, int precision, hipStream_t st);     // the folded form's V'^T operand (p.V = V'; p.K / p.Kp unused)
void launch_tv_chain(const TvChainP& p, int precision, hipStream_t st);
// transpose [B,C,L] -> [B, L(+row_off), C]
void launch_transpose_cl(const float* src, float* dst, int B, int C, int L, int row_off, long dst_bstride, hipStream_t st);

// HiFi-GAN generator pieces (dex_vocoder.hip; reference hifigan/models.py:112-173) --------------------
// mel [B,80,T] -> channels-last [B,T,ldc] with the channels past 80 zeroed (the implicit GEMM wants Cin % 32 == 0)
void launch_mel_to_cl(const float* mel, float* out, int B, int C, int T, int ldc, hipStream_t st);
// ConvTranspose1d(k, stride u, padding (k-u)/2) after its GEMM Y[l][j*Cout + co] = sum_ci x[l][ci] w[ci][co][j]:
// out[t][co] = bias[co] + sum_{j = (t+pad) mod u, +u, .. < k} Y[(t+pad-j)/u][j][co]   (0 <= (t+pad-j)/u < L)
struct ConvTFoldP { const float* Y; const float* bias; float* out; int L, Cout, k, u, pad, B; };
void launch_convt_fold(const ConvTFoldP& p, hipStream_t st);
// x = (a + b + c) * (1/3)  (the three ResBlocks of a stage, models.py:158-164); n floats
void launch_avg3(const float* a, const float* b, const float* c, float* out, long n, hipStream_t st);
// wav[t] = tanh(bias + sum_{tap<7} sum_{c<C} w[tap][c] * leaky_relu(x[t+tap-3][c], 0.01))   (models.py:165-167)
struct ConvPostP { const float* X; const float* W; const float* bias; float* wav; int L, C, B; float slope; };   // leaky_relu slope on the input (1 = none)
// BigVGAN anti-aliased periodic activation (alias_free_torch/act.py:23-28): y = downsample2(snake(upsample2(x))) per channel, channels-last
// [B][L][C]; a[c] = frequency, inv_b[c] = 1 / (magnitude + 1e-9) (already exponentiated when the checkpoint is log-scale); filt = the 12-tap
// Kaiser-sinc low-pass of both resamplers
struct AaSnakeP { const float* X; float* Y; int L, C, B; const float* a; const float* inv_b; const float* filt; };
void launch_aa_snake(const AaSnakeP& p, hipStream_t st);
void launch_snake_coeffs(const float* alpha, const float* beta, float* a, float* inv_b, int C, int logscale, hipStream_t st);
void launch_conv_post_tanh(const ConvPostP& p, hipStream_t st);

// DEX style encoders (dex_style.hip; reference DEX-TTS/model/ref_encoder.py:8-140,199-237) ------------
// LayerNorm over the channels of channels-last rows (nn.LayerNorm / base.LayerNorm: biased variance), affine, then * mask[b][t]
struct LnClP { const float* X; float* Y; long rows; int C; const float* gamma; const float* beta; float eps;
               const float* mask; int T; };      // mask [B][T] (row = b*T + t) or null
void launch_ln_cl(const LnClP& p, hipStream_t st);
// InstanceNorm1D over the full padded length (unbiased variance, base.py:72-88), output * mask: X, Y [B][T][C]
void launch_inorm_cl(const float* X, float* Y, const float* mask, int B, int T, int C, float eps, hipStream_t st);
// out[b][c] = sum_t X[b][t][c] / sum_t mask[b][t]   (tts.py:62)
void launch_masked_mean_cl(const float* X, const float* mask, float* out, int B, int T, int C, hipStream_t st);
// X[b][t][c] += v[b][c]
void launch_add_bcast_cl(float* X, const float* v, int B, int T, int C, hipStream_t st);
// lf0 [B][T] * mask -> channels-last [B][T][ldc], channel 0 = value, the rest 0
void launch_lf0_to_cl(const float* lf0, const float* mask, float* out, int B, int T, int ldc, hipStream_t st);
// channels-last [B][T][C] -> channel-first [B][C][T]
void launch_cl_to_cf(const float* X, float* out, int B, int T, int C, hipStream_t st);
// sequence mask [B][T] (float 0/1) from int32 lengths
void launch_len_mask(const int* lengths, float* mask, int B, int T, hipStream_t st);
// VQEmbeddingEMA eval lookup (ref_encoder.py:205-216): dots [R][M] = x . e^T;  idx = argmin_m (|e_m|^2 + |x|^2) - 2 dots;
// out[r][:] = e[idx] * mask[r]
struct VqP { const float* X; const float* dots; const float* emb; const float* e2; const float* mask; float* out; int* idx;
             long rows; int M, D; };
void launch_vq_lookup(const VqP& p, hipStream_t st);
void launch_row_sumsq(const float* X, float* out, long rows, int D, hipStream_t st);
// one direction of one bidirectional GRU layer per (direction, batch element) workgroup (nn.GRU, hidden H = 96):
// gi [B][T][2][3H] = W_ih x + b_ih (precomputed), Whh [2][3H][H], bhh [2][3H]; out [B][T][2H] (forward | reverse)
struct GruP { const float* gi; const float* Whh; const float* bhh; float* out; int B, T, H; };
void launch_gru_layer(const GruP& p, hipStream_t st);

// Text encoder path (text_elem.hip; dex_text.hip) -------------------------------------------------
void launch_embed(const int* tok, const float* emb, float* out, long rows, int C, int ld, float scale, int n_vocab, hipStream_t st);
void launch_bcast_cols(float* X, int ld, int coff, const float* v, int B, int T, int n, hipStream_t st);
// per-row norm over C <= 256 channels.  mode 0: LayerNorm * gamma + beta; 1: RMSNorm * gamma (gamma may be null);
// 2: LayerNorm * gamma[b] + beta[b] (per-utterance [B][C] tables, rows = b*T + t).  Then optional ReLU, then * mask[row].
struct RowNormP { const float* X; int ldx; float* Y; int ldy; long rows; int C; int mode; const float* gamma; const float* beta; float eps;
                  int relu; const float* mask; int T; };
void launch_row_norm(const RowNormP& p, hipStream_t st);
// qkvg [rows][ld] = [q | k | v | g] (E each) -> Q, K, V [rows][ldp = heads*128]: head h at columns h*128.., kd real + zero padding;
// q, k rotated by angle[d] * t (xPos theta_shift), k pre-scaled
struct RetRotP { const float* qkvg; int ld; long rows; int T, heads, kd, E; const float* angle; float kscale; float *Q, *K, *V; int ldp; };
void launch_ret_rotate(const RetRotP& p, hipStream_t st);
struct RetGateP { const float* O; int ldo; const float* qkvg; int ld; float* out; int ldout; long rows; int heads, hd, E; float eps; };
void launch_ret_gate(const RetGateP& p, hipStream_t st);
void launch_glu(const float* GF, float* out, long rows, int F, hipStream_t st);
void launch_cl_to_cf_mask(const float* X, int ldx, const float* mask, float* out, int B, int T, int C, hipStream_t st);
void launch_durations(const float* logw, const float* mask, float length_scale, float* w_ceil, float* cum, int* y_len, int B, int T, hipStream_t st);
void launch_cumsum_rows(const float* X, float* out, int B, int T, hipStream_t st);
struct AlignP { const float* mu_x; const float* cum; const int* x_len; const int* y_len; int B, T, Ty, F; float* mu_y; float* y_mask; float* attn; };
void launch_align(const AlignP& p, hipStream_t st);

// STFT / mel -------------------------------------------------------------------------------------
// clip to [-1,1] + reflect-pad n_fft/2 on both sides (stft.py:60-66, tools.py:9)
void launch_wav_pad(const float* wav, int n, int pad, float* out, int out_len, hipStream_t st, int B = 1, long out_bstride = 0);
// spec [frames][ld] holds re at cols [0,513) and im at cols [im_off, im_off+513) ->
// mel[j][f] = log(max(sum_k melW[j][k]*|spec|, 1e-5)), energy[f] = ||mag||_2   (stft.py:172-176)
struct MagMelP { const float* spec; int ld; int im_off; int frames; int nbins; const float* melW; int nmel;
                 float* mel; float* energy; };
void launch_magmel(const MagMelP& p, hipStream_t st, int B = 1);
// DEX style front-end: log-f0 + per-utterance normalisation (DEX-TTS/synthesize.py:26-38,55-58); lengths may be null
void launch_lf0_normalize(const float* f0, const int* lengths, int B, int T, float* out, hipStream_t st);

}  // namespace dex
