// dit_rowchain.hip — everything of a DiT block that is row-local, in ONE launch (bf16-MFMA mode).
//
// After the attention core, the rest of a DiTBlock (reference model/dit.py:73-84) touches one token row at a time:
//     x1  = x  + gate_msa * (O Wproj + b)                        attention projection + gated residual
//     h   = GELU(modulate(LN(x1), shift_mlp, scale_mlp) W1 + b1) Mlp.fc1
//     x2  = x1 + gate_mlp * (h W2 + b2)                          Mlp.fc2 + gated residual
//     qkv = modulate(LN(x2), shift_msa', scale_msa') Wqkv' + b'  the NEXT block's qkv projection
// At B=1 (650 token rows) these four GEMMs are pure latency — separately they cost 4 launches of 7-13 us each for
// ~0.3 GFLOP.  Here a workgroup of 8 waves owns a 32-row tile and walks the whole chain with the activations in
// LDS (bf16 MFMA A operands, fp32 residual stream); each wave owns 32-column slices of every weight matrix and
// streams them from L2 straight into MFMA B-operand registers.  Weights are host-packed in fragment order
// (`pack_lp_frag_kernel`: one 1-KB contiguous read per wave per K-step of 16) and double-buffered one 16-KB
// tile ahead, across stage boundaries — weights never depend on the data.
//
// Accumulator layout (32x32x16 bf16): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {

constexpr int RC_H = 256, RC_MLP = 512, RC_ROWS = 32, RC_NW = 8;
constexpr int A_LD = RC_H + 8;       // bf16 elements; row stride 528 B: b128 reads of 16 rows hit 64 distinct banks
constexpr int H_LD = RC_MLP + 8;
constexpr int X_LD = RC_H + 4;       // floats
constexpr int AT_LD = 132;           // attention partial O tile row stride (floats)
constexpr size_t RC_LDS_ATTN = (size_t)(RC_NW * 32 * AT_LD + RC_NW * 64) * sizeof(float);
constexpr size_t RC_LDS = (size_t)RC_ROWS * (A_LD + H_LD) * sizeof(u16) + (size_t)RC_ROWS * X_LD * sizeof(float) + 4 * RC_H * sizeof(float);

// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the bf16 rounding the value
// gets next): ~15 VALU ops instead of the ~45 of libm erff, 32 of them per lane in the fc1 epilogue
__device__ __forceinline__ float gelu_erf_rc(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to the IEEE divide
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f); pl = fmaf(pl, t, -0.284496736f); pl = fmaf(pl, t, 0.254829592f);
    const float er = 1.0f - pl * t * __expf(-z * z);           // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(er, x));
}

// Split-weight build (DEX_LP_WSPLIT, lp_config.h): every weight matrix in fragment order is followed by its lo half (what the fp16
// rounding of the weight lost, same layout), K * N elements behind; a product is two MFMAs, hi then lo, into one accumulator.
// WT16: the hi fragments ride in registers one stage ahead as before, the lo fragments are fetched inside the MFMA chain (eight at
// a time, under the eight hi MFMAs of their half): two more register tiles do not fit next to wa / wb.  WT8 (cluster form): both halves
// ride ahead.  LO_* = the distance to the lo half in 16-byte units.
constexpr long LO_WP = (long)RC_H * RC_H / 8, LO_W1 = (long)RC_H * RC_MLP / 8, LO_W2 = (long)RC_MLP * RC_H / 8, LO_WQ = (long)RC_H * 3 * RC_H / 8;
#ifdef DEX_LP_WSPLIT
struct WT16 { uint4 h[16]; const uint4* lo; };
struct WT8 { uint4 h[8]; uint4 l[8]; };
#else
struct WT16 { uint4 h[16]; };
struct WT8 { uint4 h[8]; };
#endif
// one weight tile = 32 output columns x 256 K = 16 K-steps x (64 lanes x 16 B)
__device__ __forceinline__ void wload(WT16& w, const void* W, long lo_u4, int ksteps_total, int nt, int ks0, int lane) {
    const uint4* src = reinterpret_cast<const uint4*>(W) + ((long)nt * ksteps_total + ks0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) w.h[j] = src[j * 64];
#ifdef DEX_LP_WSPLIT
    w.lo = src + lo_u4;
#endif
    __builtin_amdgcn_sched_barrier(0);     // all 16 loads issue HERE (the scheduler otherwise drips them into the MFMA chain below)
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// acc += A(32 rows x 256 K, bf16 in LDS) x W(16 register fragments).  The A fragments are read from LDS in two
// batches of 8 (all ds_read_b128 of a batch in flight together, 32 VGPRs) instead of one right before each MFMA,
// which exposed an LDS latency per link of the dependent MFMA chain.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma16(f32x16& acc, const WT16& w, const u16* a_lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 a[8];
#ifdef DEX_LP_WSPLIT
        uint4 l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = w.lo[(h * 8 + j) * 64];
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const u32x4*>(a_lane + (h * 8 + j) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, a[j]), __builtin_bit_cast(lp8, w.h[h * 8 + j]), acc, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, a[j]), __builtin_bit_cast(lp8, l[j]), acc, 0, 0, 0);
#endif
    }
}

// The same product TRANSPOSED (weights as the A operand, activations as B): acc[r] = C[feature (r&3) + 8 (r>>2) + 4 hh][token lane&31].
// Same products, same K order per output element: the values are those of mma16, only their placement over lanes / registers differs
// - it is the placement from which the q / k fragment layouts can be written without a transpose (store_qkv_tile_d).
__device__ __forceinline__ void mma16t(f32x16& acc, const WT16& w, const u16* a_lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 a[8];
#ifdef DEX_LP_WSPLIT
        uint4 l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = w.lo[(h * 8 + j) * 64];
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const u32x4*>(a_lane + (h * 8 + j) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, w.h[h * 8 + j]), __builtin_bit_cast(lp8, a[j]), acc, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, l[j]), __builtin_bit_cast(lp8, a[j]), acc, 0, 0, 0);
#endif
    }
}

// LayerNorm(eps 1e-6, no affine) + modulate of the 32 fp32 rows in X1 -> bf16 A tile.  16 threads per row, each
// owning four float4 at columns q*64 + seg*4 (64 contiguous floats per 16 lanes: conflict-free).
__device__ __forceinline__ void ln_to_A(const float* X1, u16* As, const float* shift, const float* scale, int tid) {
    const int row = tid >> 4, seg = tid & 15;
    const float* xr = X1 + row * X_LD + seg * 4;
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(xr + q * 64);
    float4 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = *reinterpret_cast<const float4*>(scale + q * 64 + seg * 4);
        sh[q] = *reinterpret_cast<const float4*>(shift + q * 64 + seg * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.f / RC_H);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q].x -= mean; v[q].y -= mean; v[q].z -= mean; v[q].w -= mean;
        ss += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss * (1.f / RC_H) + 1e-6f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint2 o;
        o.x = pack2_lp(v[q].x * rstd * (1.f + sc[q].x) + sh[q].x, v[q].y * rstd * (1.f + sc[q].y) + sh[q].y);
        o.y = pack2_lp(v[q].z * rstd * (1.f + sc[q].z) + sh[q].z, v[q].w * rstd * (1.f + sc[q].w) + sh[q].w);
        *reinterpret_cast<uint2*>(As + row * A_LD + q * 64 + seg * 4) = o;
    }
}

// One 32x32 tile of the next block's qkv projection -> the attention kernel's operand layouts (attention_direct.hip):
//   q (pre-scaled), k and v^T as bf16 in MFMA fragment order per (batch, head), rows padded to Npad (see
//   attention_direct.hip); v^T key positions have index bits 2/3 swapped inside each 32-key block (the order in
//   which a P^T accumulator lane holds its keys).
// `scr` is a wave-private LDS scratch of 32 x QK_LD bf16.  Row tiles never cross an utterance (the grid is per batch
// element) and start at a multiple of 32 tokens, so a tile is exactly one 32-row block of the fragment layouts and
// every global store is a full 16-byte chunk: 2 per lane instead of 16 scattered 2-byte stores (which cost 1.4 us
// per tile, measured).  Rows past N land in the padding rows of the operand buffers (finite duplicates; masked /
// multiplied by exact zeros in the attention kernel).
constexpr int QK_LD = 40;
__device__ __forceinline__ void store_qkv_tile(const DitChainP& p, const f32x16& acc, int nt, float bias, int b, int n0, int lane, u16* scr) {
    const int i = lane & 31, hh = lane >> 5;
    const int kind = nt >> 3, head = (nt >> 2) & 1, d0 = (nt & 3) * 32;
    const float sc = kind == 0 ? p.qscale : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        scr[row * QK_LD + i] = (u16)(pack2_lp((acc[r] + bias) * sc, 0.f) & 0xffffu);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's scratch writes landed
    __builtin_amdgcn_wave_barrier();
    u16* base = (kind == 0 ? reinterpret_cast<u16*>(p.Qh) : kind == 1 ? reinterpret_cast<u16*>(p.Kh) : reinterpret_cast<u16*>(p.Vt))
                + ((long)b * 2 + head) * p.Npad * 128 + (long)(n0 >> 5) * 4096;
    if (kind < 2) {
        // q / k: [tile][ks = d/16][lane = (d/8 & 1)*32 + row][8 d]: chunk (row, c8) -> 8 consecutive d
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = lane + 64 * c, row = id >> 2, c8 = id & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + row * QK_LD + c8 * 8);
            const int d = d0 + c8 * 8;
            *reinterpret_cast<uint4*>(base + (((d >> 4) * 64 + ((d >> 3) & 1) * 32 + row) << 3)) = v;
        }
    } else {
        // v^T: [tile][t = d/32][k2 = pos/16 & 1][lane = (pos/8 & 1)*32 + d%32][8 positions], position = row with bits 2/3
        // swapped: chunk (d, position group pg) gathers rows swz(pg*8 + j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = lane + 64 * c, dl = id & 31, pg = id >> 5;
            unsigned w[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const int p0 = pg * 8 + 2 * j2, p1 = p0 + 1;
                const int r0 = (p0 & ~12) | ((p0 & 4) << 1) | ((p0 & 8) >> 1), r1 = (p1 & ~12) | ((p1 & 4) << 1) | ((p1 & 8) >> 1);
                w[j2] = (unsigned)scr[r0 * QK_LD + dl] | ((unsigned)scr[r1 * QK_LD + dl] << 16);
            }
            const int t = d0 >> 5;
            *reinterpret_cast<uint4*>(base + (((t * 2 + (pg >> 1)) * 64 + (pg & 1) * 32 + dl) << 3)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __builtin_amdgcn_wave_barrier();                     // scratch is reused by the next tile
}


// Round 4: the same tile WITHOUT the LDS round trip (the scatter of 16 two-byte LDS writes, two wave barriers and the gather back were a
// large part of the row chain's 12 VALU instructions per MFMA, profiles/round3_dex_b32_diag_counters.txt).
//   KIND 0 / 1 (q / k): `acc` comes from mma16t (features in registers, token = lane & 31).  A lane packs its 16 features into four
//     2-dword pieces (features 8 j + 4 hh .. + 3); one v_permlane32_swap per dword pairs the pieces of the two halves of a token into
//     whole 8-feature chunks - chunk 2 pr + hh ends up in lane (hh, token), which is exactly lane `lane` of K-step (d0 / 16 + pr) of
//     the fragment layout: two lane-linear 1-KB stores.
//   KIND 2 (v^T): `acc` comes from mma16 (feature = lane & 31, tokens in registers).  The v^T layout's position order (bits 2 / 3 of the
//     key index swapped) IS the accumulator's row order: registers 8 half .. + 7 of lane (hh, d) are position group hh + 2 half, in
//     order: pack and store, nothing to exchange.
// Values are bit-identical to store_qkv_tile's ((acc + bias) * scale, rounded once).
struct QkvBias { float4 f[4]; };      // KIND 0 / 1: features 8 j + 4 hh .. + 3 of the tile (j = 0..3); KIND 2: f[0].x = the lane's feature
template <int KIND>
__device__ __forceinline__ QkvBias qkv_bias(const DitChainP& p, int nt, int lane) {
    QkvBias q;
    const int i = lane & 31, hh = lane >> 5;
    if constexpr (KIND < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) q.f[j] = *reinterpret_cast<const float4*>(p.bq + nt * 32 + 8 * j + 4 * hh);
    } else {
        q.f[0] = make_float4(p.bq[nt * 32 + i], 0.f, 0.f, 0.f);
        q.f[1] = q.f[2] = q.f[3] = q.f[0];
    }
    return q;
}
template <int KIND>
__device__ __forceinline__ void store_qkv_tile_d(const DitChainP& p, const f32x16& acc, const QkvBias& qb, int nt, int b, int n0, int lane) {
    const int i = lane & 31, hh = lane >> 5;
    const int head = (nt >> 2) & 1, d0 = (nt & 3) * 32;
    u16* base = (KIND == 0 ? reinterpret_cast<u16*>(p.Qh) : KIND == 1 ? reinterpret_cast<u16*>(p.Kh) : reinterpret_cast<u16*>(p.Vt))
                + ((long)b * 2 + head) * p.Npad * 128 + (long)(n0 >> 5) * 4096;
    if constexpr (KIND < 2) {
        const float sc = KIND == 0 ? p.qscale : 1.f;
        unsigned D[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bj = qb.f[j];
            D[j][0] = pack2_lp((acc[4 * j + 0] + bj.x) * sc, (acc[4 * j + 1] + bj.y) * sc);
            D[j][1] = pack2_lp((acc[4 * j + 2] + bj.z) * sc, (acc[4 * j + 3] + bj.w) * sc);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
            const u32x2v s0 = __builtin_amdgcn_permlane32_swap(D[2 * pr][0], D[2 * pr + 1][0], false, false);
            const u32x2v s1 = __builtin_amdgcn_permlane32_swap(D[2 * pr][1], D[2 * pr + 1][1], false, false);
            const int ks = (d0 >> 4) + pr;
            *reinterpret_cast<uint4*>(base + ((ks * 64 + lane) << 3)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    } else {
        const float bias = qb.f[0].x;
        const int t = d0 >> 5;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack2_lp(acc[8 * half + 2 * j] + bias, acc[8 * half + 2 * j + 1] + bias);
            *reinterpret_cast<uint4*>(base + (((t * 2 + half) * 64 + lane) << 3)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
}  // namespace

// ATTN: the attention core of the block runs INSIDE this launch (no attention kernel, no partials in HBM): waves 0-3 take
// head 0, waves 4-7 head 1, each group splits the key tiles four ways (attention_direct.hip's wave body), the partial
// (m, l, O) are merged through LDS straight into the bf16 A tile of the projection.
// Merge of the key-split attention partials of token n of element b for this thread's columns (seg * 4 .. + 3 of each 64-column
// block q; block q belongs to head q / 2, head_dim 128):  O = sum_s w_s O_s / sum_s w_s,  w_s = l_s 2^(m_s - max m)  (log2-domain
// maxima).  Up to RC_MAX_SPLITS splits, four at a time (16 loads in flight per round; one round = the <= 4 splits of the 32-query
// attention forms, same operation order as before; the 64-query form takes up to 8 at long-form shapes).
constexpr int RC_MAX_SPLITS = 8;
__device__ __forceinline__ void rc_merge_splits(const DitChainP& p, const float* src, int b, int n, float4 (&v)[4], int ksplit) {
    float w[2][RC_MAX_SPLITS], inv[2];
    {
        float2 st[RC_MAX_SPLITS][2];
#pragma unroll
        for (int s_ = 0; s_ < RC_MAX_SPLITS; ++s_)
            if (s_ < ksplit) {
#pragma unroll
                for (int hd = 0; hd < 2; ++hd)
                    st[s_][hd] = *reinterpret_cast<const float2*>(p.ml + ((((long)s_ * p.B + b) * p.heads + hd) * p.rows_per_batch + n) * 2);
            }
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            float mx = -INFINITY;
#pragma unroll
            for (int s_ = 0; s_ < RC_MAX_SPLITS; ++s_) if (s_ < ksplit) mx = fmaxf(mx, st[s_][hd].x);
            float wsum = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < RC_MAX_SPLITS; ++s_) {
                w[hd][s_] = 0.f;
                if (s_ < ksplit) { w[hd][s_] = st[s_][hd].y * exp2f(st[s_][hd].x - mx); wsum += w[hd][s_]; }
            }
            inv[hd] = 1.f / wsum;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r0 = 0; r0 < RC_MAX_SPLITS; r0 += 4) {
        if (r0 < ksplit) {                                  // (uniform)
            float4 pv[4][4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (r0 + s_ < ksplit) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) pv[s_][q] = *reinterpret_cast<const float4*>(src + (long)(r0 + s_) * p.o_sstride + q * 64);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    if (r0 + s_ < ksplit) {
                        const float ws = w[q >> 1][r0 + s_];
                        v[q].x = fmaf(ws, pv[s_][q].x, v[q].x); v[q].y = fmaf(ws, pv[s_][q].y, v[q].y);
                        v[q].z = fmaf(ws, pv[s_][q].z, v[q].z); v[q].w = fmaf(ws, pv[s_][q].w, v[q].w);
                    }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float iv = inv[q >> 1]; v[q] = make_float4(v[q].x * iv, v[q].y * iv, v[q].z * iv, v[q].w * iv); }
}

template <bool ATTN>
__global__ __launch_bounds__(RC_NW * 64) void dit_rowchain_kernel(const DitChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rc[];
    u16* As = reinterpret_cast<u16*>(smem_rc);                 // [32][A_LD]  bf16 A operand (O, then LN outputs)
    u16* Hs = As + RC_ROWS * A_LD;                             // [32][H_LD]  bf16 GELU(fc1)
    float* X1 = reinterpret_cast<float*>(Hs + RC_ROWS * H_LD); // [32][X_LD]  fp32 residual stream
    float* LNp = X1 + RC_ROWS * X_LD;                          // [4][256]   shift_mlp, scale_mlp, next shift, next scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    // row tiles are per batch element (never across utterances): tile t of element b covers tokens [32 t, 32 t + 32)
    const int N = p.rows_per_batch, tpb = (N + RC_ROWS - 1) / RC_ROWS;
    // p.xcd_map (launcher: in-kernel attention, B a multiple of 8): workgroup L runs on XCD L % 8, so element b's row tiles go to
    // XCD b % 8 - the K / V^T of an utterance (665 KB at N = 650) are fetched into ONE L2 instead of all eight (PMC at GeDEX B = 32:
    // 286 MB per launch against 113 MB algorithmic before)
    int b, n0;
    if (p.xcd_map) { const int slot = (int)blockIdx.x >> 3; b = ((int)blockIdx.x & 7) + 8 * (slot / tpb); n0 = (slot % tpb) * RC_ROWS; }
    else { b = blockIdx.x / tpb; n0 = (blockIdx.x - b * tpb) * RC_ROWS; }
    const long mb = (long)b * N;                               // first global row of this batch element
    const int step = p.step;
    const float* ada = p.ada + (long)step * 6 * RC_H;
    const bool has_q = p.next_shift != nullptr;

#ifdef DEX_TIMING
    long long tst[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
    tst[0] = wall_clock64();
#endif
    float2 lnv;
    {   // both LayerNorm parameter sets -> LDS (a dependent L2 round trip inside each LN otherwise)
        const int which = tid >> 7, c2 = (tid & 127) * 2;       // 512 threads x 2 floats = 4 x 256
        const float* src = which == 0 ? ada + 3 * RC_H : which == 1 ? ada + 4 * RC_H
                         : which == 2 ? (has_q ? p.next_shift + (long)step * p.next_step_stride : ada)
                                      : (has_q ? p.next_scale + (long)step * p.next_step_stride : ada);
        lnv = *reinterpret_cast<const float2*>(src + c2);
        if (!ATTN || p.qkv_only) *reinterpret_cast<float2*>(LNp + which * RC_H + c2) = lnv;     // ATTN: the LDS is attention scratch first
    }
    WT16 wa, wb;
    const int col = wave * 32 + i;
    const u16* a_lane = As + i * A_LD + hh * 8;
    const u16* h_lane = Hs + i * H_LD + hh * 8;
    f32x16 acc;
    if (p.qkv_only) {
        // first block: just LN + modulate + qkv of the incoming token rows
        wload(wb, p.Wq, LO_WQ, 16, wave, 0, lane);
        const int row = tid >> 4, seg = tid & 15;
        const float* src = p.X + (mb + min(n0 + row, N - 1)) * RC_H + seg * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(X1 + row * X_LD + q * 64 + seg * 4) = *reinterpret_cast<const float4*>(src + q * 64);
    } else {
    f32x16 ao[4];
    float am = -INFINITY, al = 0.f;
    if constexpr (ATTN) {
        union DFr { uint4 u; lp8 v; };
        const int head = wave >> 2, part = wave & 3;
        const long hbq = ((long)b * 2 + head) * p.Npad * 16;
        const uint4* Qg = reinterpret_cast<const uint4*>(p.Qin) + hbq + lane;
        const uint4* Kg = reinterpret_cast<const uint4*>(p.Kin) + hbq + lane;
        const uint4* Vg = reinterpret_cast<const uint4*>(p.Vin) + hbq + lane;
        const int ntiles = (N + 31) / 32;
        DFr qf[8], kf[8], vf[4][2];
        int kt = part;
        {
            const uint4* qp = Qg + (long)(n0 >> 5) * 512;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[ks].u = qp[ks * 64];
            const uint4* kp = Kg + (long)min(kt, ntiles - 1) * 512;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
            const uint4* vp = Vg + (long)min(kt, ntiles - 1) * 512;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[t][r] = 0.f;
        while (kt < ntiles) {
            const int k0 = kt * 32, kn = kt + 4;
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) sc = DEX_MFMA_LP(kf[ks].v, qf[ks].v, sc, 0, 0, 0);
            if (kn < ntiles) {
                const uint4* kp = Kg + (long)kn * 512;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
            }
            if (k0 + 32 > N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sc[r] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (__builtin_amdgcn_ballot_w64(mx > am + 8.f) != 0) {   // lazy rescale (attention_direct.hip): reference max moves rarely
                const float m_new = fmaxf(am, mx);
                const float alpha = exp2f(am - m_new);           // scores are in the log2 domain (q scale carries log2 e)
                al *= alpha;
                am = m_new;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ao[t][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - am); psum += sc[r]; }
            al += psum;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                DFr pb;
                pb.u.x = pack2_lp(sc[8 * k2 + 0], sc[8 * k2 + 1]); pb.u.y = pack2_lp(sc[8 * k2 + 2], sc[8 * k2 + 3]);
                pb.u.z = pack2_lp(sc[8 * k2 + 4], sc[8 * k2 + 5]); pb.u.w = pack2_lp(sc[8 * k2 + 6], sc[8 * k2 + 7]);
#pragma unroll
                for (int t = 0; t < 4; ++t) ao[t] = DEX_MFMA_LP(vf[t][k2].v, pb.v, ao[t], 0, 0, 0);
            }
            if (kn < ntiles) {
                const uint4* vp = Vg + (long)kn * 512;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
            }
            kt = kn;
        }
#ifdef DEX_TIMING
        tst[8] = wall_clock64();
#endif
        al += __shfl_xor(al, 32);
        // partial (m, l, O[query][d]) of this wave -> LDS scratch (the chain's buffers are not live yet)
        float* scr = reinterpret_cast<float*>(smem_rc);
        float* oS = scr + wave * (32 * AT_LD);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(oS + i * AT_LD + t * 32 + 8 * rq + 4 * hh) =
                    make_float4(ao[t][rq * 4 + 0], ao[t][rq * 4 + 1], ao[t][rq * 4 + 2], ao[t][rq * 4 + 3]);
        if (hh == 0) { scr[RC_NW * 32 * AT_LD + (wave * 2 + 0) * 32 + i] = am; scr[RC_NW * 32 * AT_LD + (wave * 2 + 1) * 32 + i] = al; }
    }
    wload(wa, p.Wp, LO_WP, 16, wave, 0, lane);
    // residual rows of this lane's output column + the attention output tile
    float xres[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nn = min(n0 + (r & 3) + 8 * (r >> 2) + 4 * hh, N - 1);
        xres[r] = p.X[(mb + nn) * RC_H + col];
    }
    {
        const int row = tid >> 4, seg = tid & 15;
        const int n = min(n0 + row, N - 1);
        const float* src = p.O + (mb + n) * RC_H + seg * 4;
        float4 v[4];
        if constexpr (ATTN) {
#ifdef DEX_TIMING
            tst[9] = wall_clock64();
#endif
            lds_barrier();                                   // every wave's partial is in the scratch
#ifdef DEX_TIMING
            tst[10] = wall_clock64();
#endif
            const float* scr = reinterpret_cast<const float*>(smem_rc);
            const float* stat = scr + RC_NW * 32 * AT_LD;
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                float mw[4], M = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) { mw[w] = stat[((hd * 4 + w) * 2) * 32 + row]; M = fmaxf(M, mw[w]); }
                float L = 0.f, f[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) { f[w] = (mw[w] == -INFINITY) ? 0.f : exp2f(mw[w] - M); L += f[w] * stat[((hd * 4 + w) * 2 + 1) * 32 + row]; }
                const float inv = 1.f / L;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float4 t4 = *reinterpret_cast<const float4*>(scr + ((hd * 4 + w) * 32 + row) * AT_LD + qq * 64 + seg * 4);
                        a.x = fmaf(f[w], t4.x, a.x); a.y = fmaf(f[w], t4.y, a.y); a.z = fmaf(f[w], t4.z, a.z); a.w = fmaf(f[w], t4.w, a.w);
                    }
                    v[hd * 2 + qq] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
                }
            }
            lds_barrier();                                   // scratch fully read: the chain's buffers may now be written
#ifdef DEX_TIMING
            tst[11] = wall_clock64();
#endif
            const int which = tid >> 7, c2 = (tid & 127) * 2;
            *reinterpret_cast<float2*>(LNp + which * RC_H + c2) = lnv;
        } else if (p.ksplit <= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(src + q * 64);
        } else {
            rc_merge_splits(p, src, b, n, v, p.ksplit);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint2 o;
            o.x = pack2_lp(v[q].x, v[q].y); o.y = pack2_lp(v[q].z, v[q].w);
            *reinterpret_cast<uint2*>(As + row * A_LD + q * 64 + seg * 4) = o;
        }
    }
    const float b_p = p.bp[col], g_msa = ada[2 * RC_H + col];
    const float b_1a = p.b1[col], b_1b = p.b1[col + 256];
    const float b_2 = p.b2[col], g_mlp = ada[5 * RC_H + col];
    lds_barrier();
#ifdef DEX_TIMING
    tst[1] = wall_clock64();
#endif

    // ---- x1 = x + gate_msa * (O Wproj + b)
    wload(wb, p.W1, LO_W1, 16, wave, 0, lane);
    acc = zero16();
    mma16(acc, wa, a_lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        X1[row * X_LD + col] = xres[r] + g_msa * (acc[r] + b_p);
    }
    lds_barrier();
#ifdef DEX_TIMING
    tst[2] = wall_clock64();
#endif
    ln_to_A(X1, As, LNp, LNp + RC_H, tid);
    lds_barrier();
#ifdef DEX_TIMING
    tst[3] = wall_clock64();
#endif


    // ---- h = GELU(A W1 + b1): column tiles `wave` and `wave + 8`
    wload(wa, p.W1, LO_W1, 16, wave + 8, 0, lane);
    acc = zero16();
    mma16(acc, wb, a_lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        Hs[row * H_LD + col] = (u16)(pack2_lp(gelu_erf_rc(acc[r] + b_1a), 0.f) & 0xffffu);
    }
    wload(wb, p.W2, LO_W2, 32, wave, 0, lane);
    acc = zero16();
    mma16(acc, wa, a_lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        Hs[row * H_LD + col + 256] = (u16)(pack2_lp(gelu_erf_rc(acc[r] + b_1b), 0.f) & 0xffffu);
    }
    lds_barrier();
#ifdef DEX_TIMING
    tst[4] = wall_clock64();
#endif

    // ---- x2 = x1 + gate_mlp * (h W2 + b2)
    wload(wa, p.W2, LO_W2, 32, wave, 16, lane);
    acc = zero16();
    mma16(acc, wb, h_lane);
    if (has_q) wload(wb, p.Wq, LO_WQ, 16, wave, 0, lane);
    mma16(acc, wa, h_lane + 256);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float x2 = X1[row * X_LD + col] + g_mlp * (acc[r] + b_2);
        X1[row * X_LD + col] = x2;
        if (n0 + row < N) p.X[(mb + n0 + row) * RC_H + col] = x2;
    }
    if (!has_q) return;
    }   // !qkv_only
    lds_barrier();
#ifdef DEX_TIMING
    tst[5] = wall_clock64();
#endif

    ln_to_A(X1, As, LNp + 2 * RC_H, LNp + 3 * RC_H, tid);
    lds_barrier();


#ifdef DEX_TIMING
    tst[6] = wall_clock64();
#endif
    // ---- qkv of the next block: column tiles wave (q), wave+8 (k), wave+16 (v)
    const float bq0 = p.bq[col], bq1 = p.bq[col + 256], bq2 = p.bq[col + 512];
    wload(wa, p.Wq, LO_WQ, 16, wave + 8, 0, lane);
#ifdef DEX_TIMING
    long long u0 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // wb (this tile's weights) landed; wa still in flight
    long long u1 = wall_clock64();
#endif
    acc = zero16();
    QkvBias qb = qkv_bias<0>(p, wave, lane);
    mma16t(acc, wb, a_lane);
#ifdef DEX_TIMING
    asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[15])); long long u2 = wall_clock64();
#endif
    store_qkv_tile_d<0>(p, acc, qb, wave, b, n0, lane);
#ifdef DEX_TIMING
    long long u3 = wall_clock64();
    if (p.dbg && tid == 0) { p.dbg[256 + blockIdx.x * 4 + 0] = u1 - u0; p.dbg[256 + blockIdx.x * 4 + 1] = u2 - u1; p.dbg[256 + blockIdx.x * 4 + 2] = u3 - u2; }
#endif
    wload(wb, p.Wq, LO_WQ, 16, wave + 16, 0, lane);
    acc = zero16();
    qb = qkv_bias<1>(p, wave + 8, lane);
    mma16t(acc, wa, a_lane);
    store_qkv_tile_d<1>(p, acc, qb, wave + 8, b, n0, lane);
    acc = zero16();
    qb = qkv_bias<2>(p, wave + 16, lane);
    mma16(acc, wb, a_lane);
    store_qkv_tile_d<2>(p, acc, qb, wave + 16, b, n0, lane);
#ifdef DEX_TIMING
    if (p.dbg && tid == 0) { tst[7] = wall_clock64(); for (int k = 0; k < 8; ++k) p.dbg[blockIdx.x * 8 + k] = tst[k]; for (int k = 8; k < 12; ++k) p.dbg[1024 + blockIdx.x * 4 + k - 8] = tst[k]; }
#endif
}

// (A 128-row form — four row tiles per weight fragment, residual rows through the token buffer instead of LDS — was built and
// measured in round 3: 124 vs 106 us at B=32, N=1300 and 64.5 vs 62.1 us at N=650 (profiles/round3_rowchain_128row_form_negative.txt).
// Per 128-row workgroup 57 us against 18 us of MFMA: the LayerNorm / GELU / staging / qkv-layout phases between the GEMMs are
// what a workgroup spends its time in, all eight waves are in the same phase at the same time, and halving the weight traffic
// per row does not touch them.  Dropped.  The follow-up — the same 64 rows on FOUR waves with a 72 KB footprint, so that TWO
// workgroups share a CU and one's MFMA chains run under the other's row-wise phases — was built too (244 VGPRs, no spills, parity
// green) and measured +0.7 % end to end at DEX B=32 (48.7k vs 48.4k frames/s, profiles/round3_rowchain_four_wave_form.txt): phase
// lock-step is not what bounds the launch either.  Dropped as well.)
// ---- 64-row form (batch regime, attention as its own launch).  At batch size every workgroup of the kernel above streams the
// block's 1.57 MB of weights for 32 token rows and the launch is bound by that L2 -> CU traffic (DEX B=32, N=1300: 1300 workgroups,
// 97 us for 65 GFLOP).  Here a workgroup owns 64 rows: every weight tile a wave fetches feeds TWO 32-row MFMA tiles, so the weight
// traffic per row halves.  The 64 x 512 GELU tile would not fit next to the fp32 residual rows, so the MLP runs in two halves of
// 256 hidden columns: fc1 half -> LDS -> fc2 partial (K half) accumulated in registers.  LDS 135 KB, one workgroup per CU as before.
constexpr size_t RC64_LDS = (size_t)64 * (A_LD + A_LD) * sizeof(u16) + (size_t)64 * X_LD * sizeof(float) + 4 * RC_H * sizeof(float);

__global__ __launch_bounds__(RC_NW * 64) void dit_rowchain64_kernel(const DitChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rc[];
    u16* As = reinterpret_cast<u16*>(smem_rc);                 // [64][A_LD]
    u16* Hs = As + 64 * A_LD;                                  // [64][A_LD]  one 256-column half of GELU(fc1)
    float* X1 = reinterpret_cast<float*>(Hs + 64 * A_LD);      // [64][X_LD]
    float* LNp = X1 + 64 * X_LD;                               // [4][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int N = p.rows_per_batch, tpb = (N + 63) / 64;
    const int b = blockIdx.x / tpb, n0 = (blockIdx.x - b * tpb) * 64;
    const long mb = (long)b * N;
    const int step = p.step;
    const float* ada = p.ada + (long)step * 6 * RC_H;
    const bool has_q = p.next_shift != nullptr;
#ifdef DEX_TIMING
    long long tst[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    tst[0] = wall_clock64();
#endif
    {
        const int which = tid >> 7, c2 = (tid & 127) * 2;
        const float* src = which == 0 ? ada + 3 * RC_H : which == 1 ? ada + 4 * RC_H
                         : which == 2 ? (has_q ? p.next_shift + (long)step * p.next_step_stride : ada)
                                      : (has_q ? p.next_scale + (long)step * p.next_step_stride : ada);
        *reinterpret_cast<float2*>(LNp + which * RC_H + c2) = *reinterpret_cast<const float2*>(src + c2);
    }
    WT16 wa, wb;
    const int col = wave * 32 + i;
    const u16* a_lane = As + i * A_LD + hh * 8;
    const u16* h_lane = Hs + i * A_LD + hh * 8;
    f32x16 acc;
    if (p.qkv_only) {
        wload(wb, p.Wq, LO_WQ, 16, wave, 0, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = (tid >> 4) + 32 * m, seg = tid & 15;
            const float* src = p.X + (mb + min(n0 + row, N - 1)) * RC_H + seg * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(X1 + row * X_LD + q * 64 + seg * 4) = *reinterpret_cast<const float4*>(src + q * 64);
        }
    } else {
        // the attention output rows (merged from the key-split partials when there are several) -> bf16 A tile
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = (tid >> 4) + 32 * m, seg = tid & 15;
            const int n = min(n0 + row, N - 1);
            const float* src = p.O + (mb + n) * RC_H + seg * 4;
            float4 v[4];
            const bool tail = p.tail_ks > 1 && n0 >= p.tail_row0;        // (uniform) rows of the attention's tail-split query groups
            if (p.o_lp && !tail) {     // (uniform) already in the operand type: straight into the A tile
                const u16* sh_ = reinterpret_cast<const u16*>(p.O) + (mb + n) * RC_H + seg * 4;
                uint2 o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o4[q] = *reinterpret_cast<const uint2*>(sh_ + q * 64);
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(As + row * A_LD + q * 64 + seg * 4) = o4[q];
                continue;
            }
            if (tail) {
                rc_merge_splits(p, src + p.o_sstride, b, n, v, p.tail_ks);        // partial slots 1 .. tail_ks
            } else if (p.ksplit <= 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(src + q * 64);
            } else {
                rc_merge_splits(p, src, b, n, v, p.ksplit);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o;
                o.x = pack2_lp(v[q].x, v[q].y); o.y = pack2_lp(v[q].z, v[q].w);
                *reinterpret_cast<uint2*>(As + row * A_LD + q * 64 + seg * 4) = o;
            }
        }
        wload(wa, p.Wp, LO_WP, 16, wave, 0, lane);                  // (after the merge: its 64 partial registers and this tile do not fit together)
        const float b_p = p.bp[col], g_msa = ada[2 * RC_H + col];
        const float b_1a = p.b1[col], b_1b = p.b1[col + 256];
        const float b_2 = p.b2[col], g_mlp = ada[5 * RC_H + col];
        lds_barrier();
#ifdef DEX_TIMING
    tst[1] = wall_clock64();
#endif

        // ---- x1 = x + gate_msa * (O Wproj + b)
        wload(wb, p.W1, LO_W1, 16, wave, 0, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float xres[16];                                   // residual rows of this lane's column (one half at a time: registers)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = min(n0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh, N - 1);
                xres[r] = p.X[(mb + nn) * RC_H + col];
            }
            acc = zero16();
            mma16(acc, wa, a_lane + m * 32 * A_LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
                X1[row * X_LD + col] = xres[r] + g_msa * (acc[r] + b_p);
            }
        }
        lds_barrier();
#ifdef DEX_TIMING
    tst[2] = wall_clock64();
#endif

        ln_to_A(X1, As, LNp, LNp + RC_H, tid);
        ln_to_A(X1 + 32 * X_LD, As + 32 * A_LD, LNp, LNp + RC_H, tid);
        lds_barrier();
#ifdef DEX_TIMING
    tst[3] = wall_clock64();
#endif

        // ---- MLP, hidden columns 0..255: h = GELU(A W1[:, :256] + b1), partial x2 += h W2[:256, :]
        wload(wa, p.W2, LO_W2, 32, wave, 0, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc = zero16();
            mma16(acc, wb, a_lane + m * 32 * A_LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Hs[row * A_LD + col] = (u16)(pack2_lp(gelu_erf_rc(acc[r] + b_1a), 0.f) & 0xffffu);
            }
        }
        lds_barrier();
#ifdef DEX_TIMING
    tst[4] = wall_clock64();
#endif

        wload(wb, p.W1, LO_W1, 16, wave + 8, 0, lane);
        f32x16 acc2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) { acc2[m] = zero16(); mma16(acc2[m], wa, h_lane + m * 32 * A_LD); }
        lds_barrier();                                        // every wave is done with this half of h
#ifdef DEX_TIMING
    tst[5] = wall_clock64();
#endif

        // ---- hidden columns 256..511
        wload(wa, p.W2, LO_W2, 32, wave, 16, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc = zero16();
            mma16(acc, wb, a_lane + m * 32 * A_LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Hs[row * A_LD + col] = (u16)(pack2_lp(gelu_erf_rc(acc[r] + b_1b), 0.f) & 0xffffu);
            }
        }
        lds_barrier();
#ifdef DEX_TIMING
    tst[6] = wall_clock64();
#endif

        if (has_q) wload(wb, p.Wq, LO_WQ, 16, wave, 0, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            mma16(acc2[m], wa, h_lane + m * 32 * A_LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float x2 = X1[row * X_LD + col] + g_mlp * (acc2[m][r] + b_2);
                X1[row * X_LD + col] = x2;
                if (n0 + row < N) p.X[(mb + n0 + row) * RC_H + col] = x2;
            }
        }
        if (!has_q) return;
    }
    lds_barrier();
#ifdef DEX_TIMING
    tst[7] = wall_clock64();
#endif

    ln_to_A(X1, As, LNp + 2 * RC_H, LNp + 3 * RC_H, tid);
    ln_to_A(X1 + 32 * X_LD, As + 32 * A_LD, LNp + 2 * RC_H, LNp + 3 * RC_H, tid);
    lds_barrier();
#ifdef DEX_TIMING
    tst[8] = wall_clock64();
#endif

    // ---- qkv of the next block: column tiles wave (q), wave+8 (k), wave+16 (v), two row halves each; q and k are computed transposed
    // so that their fragment layouts leave the accumulators without an LDS transpose (store_qkv_tile_d)
    wload(wa, p.Wq, LO_WQ, 16, wave + 8, 0, lane);
    QkvBias qb = qkv_bias<0>(p, wave, lane);
#pragma unroll
    // (the operand buffers hold ceil(N / 32) row tiles per (utterance, head): a second half that lies wholly past them is skipped)
    for (int m = 0; m < 2; ++m) { if (n0 + 32 * m >= p.Npad) break; acc = zero16(); mma16t(acc, wb, a_lane + m * 32 * A_LD); store_qkv_tile_d<0>(p, acc, qb, wave, b, n0 + 32 * m, lane); }
    wload(wb, p.Wq, LO_WQ, 16, wave + 16, 0, lane);
    qb = qkv_bias<1>(p, wave + 8, lane);
#pragma unroll
    for (int m = 0; m < 2; ++m) { if (n0 + 32 * m >= p.Npad) break; acc = zero16(); mma16t(acc, wa, a_lane + m * 32 * A_LD); store_qkv_tile_d<1>(p, acc, qb, wave + 8, b, n0 + 32 * m, lane); }
    qb = qkv_bias<2>(p, wave + 16, lane);
#pragma unroll
    for (int m = 0; m < 2; ++m) { if (n0 + 32 * m >= p.Npad) break; acc = zero16(); mma16(acc, wb, a_lane + m * 32 * A_LD); store_qkv_tile_d<2>(p, acc, qb, wave + 16, b, n0 + 32 * m, lane); }
#ifdef DEX_TIMING
    if (p.dbg && tid == 0) { tst[9] = wall_clock64(); for (int q_ = 0; q_ < 10; ++q_) p.dbg[(long)blockIdx.x * 16 + q_] = tst[q_]; }
#endif
}

// ---- 64-row form, round 5: GENERATED INSTRUCTION STREAMS ("a" form, dit_rowchain64a_kernel; tools/gen_rowchain_a.py ->
// dit_rowchain_a_core.inc).  Four waves, one per SIMD, each owning 64 features x 64 tokens of every stage; the weights stream through a
// 56-fragment ring in the accumulation file, the residual rows stay in registers, every epilogue sits in the MFMA gaps of the next
// pass - the generator's header has the register map and the schedule.  This shell stages the parameter rows (LDS) and hands the
// statement its buffer descriptors; the statement owns v0..v249, a0..a255 and the rest of LDS.  Taken when the attention output
// arrives as 16-bit rows (o_lp: one key split, no tail split) - the other cases keep the kernel above.
// Round 6: the split-weight build (dex::f16w) has streams of its own (tools/gen_rowchain_a.py WS: every weight fragment is followed by
// its lo fragment in the same ring, hi then lo MFMA into one accumulator) - the round-3 kernel it ran before took 173 us at DEX B = 32.
#ifndef RCA_CORE_INC
#define RCA_CORE_INC "dit_rowchain_a_core.inc"
#endif
#include RCA_CORE_INC
#ifdef DEX_LP_F16
#define RCA_MFMA "v_mfma_f32_32x32x16_f16"
#define RCA_PK "v_cvt_pk_f16_f32"
#else
#define RCA_MFMA "v_mfma_f32_32x32x16_bf16"
#define RCA_PK "v_cvt_pk_bf16_f32"
#endif
__global__ __launch_bounds__(256) void dit_rowchain64a_kernel(const DitChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rc[];
    float* PRM = reinterpret_cast<float*>(smem_rc + RCA_LDS_PRM);
    const int tid = threadIdx.x;
#ifdef RCA_TIMING
    const long long t_entry = clock64();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.rows_per_batch, tpb = (N + 63) / 64, ntiles = p.B * tpb;
    const int step = p.step;
    const float* ada = p.ada + (long)step * 6 * RC_H;
    const bool has_q = p.next_shift != nullptr;
    const unsigned long long wp = reinterpret_cast<unsigned long long>(p.Wp), w1 = reinterpret_cast<unsigned long long>(p.W1);
    const unsigned long long w2 = reinterpret_cast<unsigned long long>(p.W2), wq = reinterpret_cast<unsigned long long>(p.Wq);
    const unsigned qs = __float_as_uint(p.qscale);
    // descriptors of a tile (64 rows of one utterance): the residual rows and the O rows of its utterance (rows >= N are out of range:
    // stores dropped), this wave's head of q / k / v^T (feature tiles 2w, 2w + 1: head w >> 1); `on` = false: zero records (a prefetch
    // past the workgroup's last tile reads zeros)
#define RCA_DESC(SFX, TILE, ON)                                                                                                                         \
    const int b##SFX = (TILE) / tpb, n0##SFX = ((TILE) - b##SFX * tpb) * 64;                                                                           \
    const auto rx##SFX = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.X + (long)b##SFX * N * RC_H), 0, (ON) ? N * RC_H * 4 : 0, 0x00020000); \
    const auto ro##SFX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.O)) + (long)b##SFX * N * RC_H * 2, 0, \
                                                           (ON) && !p.qkv_only ? N * RC_H * 2 : 0, 0x00020000);
#define RCA_OPERANDS                                                                                                                     \
        : : [tid] "v"(tid), [w] "s"(wave), [n0] "s"(n0), [nm1] "s"(N - 1), [qscale] "s"(qs), [rx] "s"(rx), [ro] "s"(ro), [rq] "s"(rq), [rk] "s"(rk), [rv] "s"(rv), \
            [n0_n] "s"(n0_n), [rx_n] "s"(rx_n), [ro_n] "s"(ro_n),                                                                         \
            [wp_lo] "s"((unsigned)wp), [wp_hi] "s"((unsigned)(wp >> 32)), [w1_lo] "s"((unsigned)w1), [w1_hi] "s"((unsigned)(w1 >> 32)),  \
            [w2_lo] "s"((unsigned)w2), [w2_hi] "s"((unsigned)(w2 >> 32)), [wq_lo] "s"((unsigned)wq), [wq_hi] "s"((unsigned)(wq >> 32)),  \
            [dbg] "s"(dbg)                                                                                                                 \
        : RCA_CLOBBER
    int tile = blockIdx.x;
    {   // ---- statement 1: the requests of the first tile that depend on nothing (O rows, residual rows, first weight fragments) leave first
        RCA_DESC(, tile, true)
        const int n0_n = n0; const auto rx_n = rx; const auto ro_n = ro;
        const auto rq = rx, rk = rx, rv = rx;                       // (unused by this statement)
        const long long* dbg = nullptr;
        if (p.qkv_only) { asm volatile(RCA_ASM_PRE_QKV RCA_OPERANDS); }
        else { asm volatile(RCA_ASM_PRE RCA_OPERANDS); }
    }
    // ---- parameter rows: 0 shift_mlp, 1 1 + scale_mlp, 2 next shift_msa, 3 1 + next scale_msa, 4 gate_msa, 5 b_proj, 6 gate_mlp, 7 b_fc2, then
    // b_fc1 (2 rows) and b_qkv (3 rows): 13 rows of 256 floats, wave w takes rows w + 4 k - every load first, then the LDS writes (a
    // loop of load -> store was 6 serial round trips, 8k cycles).  The last block has no qkv stage (bq, next_* null), the first launch
    // nothing but it: such rows load a dummy (ada) and are not written.  (Nothing between the statements may touch the accumulation
    // file - it holds requests in flight: tools/audit_rowchain_a.py checks the build.)
    {
        float4 v[4]; bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wave + 4 * k;
            const bool need_q = row >= 10 || row == 2 || row == 3;
            ok[k] = row < 13 && (need_q ? has_q : !p.qkv_only);
            const float* src = !ok[k] ? ada
                             : row == 0 ? ada + 3 * RC_H : row == 1 ? ada + 4 * RC_H
                             : row == 2 ? p.next_shift + (long)step * p.next_step_stride
                             : row == 3 ? p.next_scale + (long)step * p.next_step_stride
                             : row == 4 ? ada + 2 * RC_H : row == 5 ? p.bp : row == 6 ? ada + 5 * RC_H : row == 7 ? p.b2
                             : row < 10 ? p.b1 + (row - 8) * RC_H : p.bq + (row - 10) * RC_H;
            v[k] = reinterpret_cast<const float4*>(src)[tid & 63];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wave + 4 * k;
            if (row == 1 || row == 3) { v[k].x += 1.f; v[k].y += 1.f; v[k].z += 1.f; v[k].w += 1.f; }
            if (ok[k]) reinterpret_cast<float4*>(PRM + row * RC_H)[tid & 63] = v[k];
        }
    }
    // ---- one statement per tile: the chain; its q pass requests the next tile's rows (the workgroup walks tiles blockIdx.x, + gridDim.x, ..)
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;
        RCA_DESC(, tile, true)
        RCA_DESC(_n, more ? next : tile, more)
        const long hb = ((long)b * 2 + (wave >> 1)) * p.Npad * 256;
        const int hsz = has_q ? p.Npad * 256 : 0;
        const auto rq = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.Qh) + hb, 0, hsz, 0x00020000);
        const auto rk = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.Kh) + hb, 0, hsz, 0x00020000);
        const auto rv = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.Vt) + hb, 0, hsz, 0x00020000);
        const long long* dbg = p.dbg ? p.dbg + ((long)tile * 4 + wave) * 32 : nullptr;          // (timing builds of the streams: one row of stamps per wave and tile)
#ifdef RCA_TIMING
        if (dbg && (tid & 63) == 0) const_cast<long long*>(dbg)[31] = tile == (int)blockIdx.x ? t_entry : 0;
#endif
        if (p.qkv_only) { asm volatile(RCA_ASM_QKV RCA_OPERANDS); }
        else if (has_q) { asm volatile(RCA_ASM_FULL RCA_OPERANDS); }
        else { asm volatile(RCA_ASM_LAST RCA_OPERANDS); }
        if (!more) break;
        tile = next;
    }
#undef RCA_OPERANDS
#undef RCA_DESC
}

// ---- cluster form (small grids: B x ceil(N / 32) row tiles <= 64).  At B = 1 the kernels above run 21 workgroups, each streaming
// the K / V^T of both heads and every weight matrix of the block (1.7 MB) through ONE CU's L2 -> register path: 22 us of which no
// phase is bound by anything but that path.  Here a 32-row tile belongs to a CLUSTER of DIT_CLUSTER = 4 workgroups
// (member c = blockIdx & 3) and every stage is cut so that a member streams a quarter of its bytes:
//     attention   head c >> 1, key half c & 1 (8 waves split its key tiles; merged in LDS, normalised by the member's own row sum)
//     proj        K-split by head: P_c = O_c Wp[128 (c >> 1) .. +128, :]                      -> fp32 partial [32, 256]
//       exchange 0  every member publishes P_c + its softmax (max, sum); all read the other three:
//                   x1 = x + gate_msa (sum_c w_c P_c + b),  w_c = the flash merge weight of (head, half) c          (redundantly: 4x)
//     LN + modulate (redundantly)
//     fc1 + GELU  hidden columns 128 c .. +128 (4 column tiles x 2 K halves over the 8 waves, halves summed through LDS)
//     fc2         K-split: its OWN 128 hidden columns, all 256 outputs: P2_c                 -> fp32 partial [32, 256]
//       exchange 1  x2 = x1 + gate_mlp (sum_c P2_c + b2); member c writes columns 64 c .. +64 of X
//     LN + modulate (redundantly), next block's qkv: column tiles 6 c .. 6 c + 5 of the 24 (waves 0-5)
// Two hand-offs per block instead of none, but 0.45 MB instead of 1.7 MB through each CU.  A hand-off follows the guide's R1 form:
// 16-byte write-through (sc1) payload stores, every storing wave drains (vmcnt(0)), workgroup barrier, ONE lane stores the flag
// (= the launch's epoch, unique within the call; the flag words are zeroed by a memset node at the start of every call); consumers
// poll with relaxed agent-scope loads from one lane per peer, then read the peers' slabs with sc1 loads.  Results do not depend on
// placement or timing: every sum runs in a fixed member order.  A wait is bounded (~50 ms): on time-out the launch finishes with
// garbage and sets *xerr instead of hanging the GPU; the call's final kernel then turns every output into NaN (FinalP::poison), so a
// lost hand-off can never pass as a plausible mel.
namespace {
constexpr int CL_AH_LD = 128 + 8;     // attention-output / GELU tile row stride (bf16): 272 B
constexpr int CL_RED_LD = 33;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wload8(WT8& w, const void* W, long lo_u4, int ksteps_total, int nt, int ks0, int lane) {
    const uint4* src = reinterpret_cast<const uint4*>(W) + ((long)nt * ksteps_total + ks0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) w.h[j] = src[j * 64];
#ifdef DEX_LP_WSPLIT
#pragma unroll
    for (int j = 0; j < 8; ++j) w.l[j] = src[lo_u4 + j * 64];
#endif
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mma8(f32x16& acc, const WT8& w, const u16* a_lane) {
    u32x4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const u32x4*>(a_lane + j * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, a[j]), __builtin_bit_cast(lp8, w.h[j]), acc, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, a[j]), __builtin_bit_cast(lp8, w.l[j]), acc, 0, 0, 0);
#endif
}
// LayerNorm + modulate of a row held in registers (16 threads per row, four float4 each at columns q*64 + seg*4) -> bf16 A tile:
// ln_to_A without the LDS round trip and its barrier (the row sums run over the 16 lanes of the row)
__device__ __forceinline__ void ln_regs_to_A(float4 (&v)[4], u16* As, const float* shift, const float* scale, int row, int seg) {
    float4 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = *reinterpret_cast<const float4*>(scale + q * 64 + seg * 4);
        sh[q] = *reinterpret_cast<const float4*>(shift + q * 64 + seg * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.f / RC_H);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q].x -= mean; v[q].y -= mean; v[q].z -= mean; v[q].w -= mean;
        ss += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss * (1.f / RC_H) + 1e-6f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint2 o;
        o.x = pack2_lp(v[q].x * rstd * (1.f + sc[q].x) + sh[q].x, v[q].y * rstd * (1.f + sc[q].y) + sh[q].y);
        o.y = pack2_lp(v[q].z * rstd * (1.f + sc[q].z) + sh[q].z, v[q].w * rstd * (1.f + sc[q].w) + sh[q].w);
        *reinterpret_cast<uint2*>(As + row * A_LD + q * 64 + seg * 4) = o;
    }
}
// hand-off scope.  LOCAL = false: the members of a cluster may sit on different XCDs (non-coherent L2s): payload stores are
// write-through (sc1), loads and flags go to memory (agent scope).  LOCAL = true: the launcher placed the four members of a
// cluster on ONE XCD (workgroup b runs on XCD b % 8; checked by a probe launch when the context is created and again here: every
// flag carries its writer's XCC id, a mismatch sets *xerr and poisons the call like a time-out): the XCD's L2 is the coherence
// point, payload stores are plain (the L1 is write-through), loads and flag polls bypass only the L1 (sc0) - an L2 round trip
// instead of a fabric one on each of the three legs of a hand-off (drain, poll, peer loads).
template <bool LOCAL> struct ClScope {
    static constexpr int ST_AUX = LOCAL ? 0 : 16, LD_AUX = LOCAL ? 1 : 16;
    // flag poll.  LOCAL: a workgroup-scope LOAD may be served by this CU's L1 (the coherence point of a workgroup), which would
    // keep returning the first value it fetched; a read-modify-write executes in the L2: fetch_or with 0 is the L2-scope load
    static __device__ __forceinline__ unsigned poll(unsigned* p) {
        if constexpr (LOCAL) {      // (as assembly: the compiler folds an idempotent fetch_or back into a load; sc0 on an atomic = return the old value)
            unsigned v;
            asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0u) : "memory");
            return v;
        } else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void st(unsigned* p, unsigned v) {
        if constexpr (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ unsigned long long ld64(const unsigned long long* p) {
        if constexpr (LOCAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v) {
        if constexpr (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
// flag word = epoch (24 bits) | writer's XCC id << 24
// one lane per peer polls that peer's flag until it shows this launch's epoch (relaxed loads + s_sleep)
template <bool LOCAL>
__device__ __forceinline__ void cluster_wait(unsigned* flags, int member, unsigned epoch, unsigned my_xcc, int tid, int* xerr) {
    if (tid < DIT_CLUSTER && tid != member) {
        long long t0 = wall_clock64(), last = t0;
        unsigned v;
        while (((v = ClScope<LOCAL>::poll(flags + tid)) & 0xffffffu) != epoch) {
            __builtin_amdgcn_s_sleep(LOCAL ? 1 : 2);
            const long long now = wall_clock64();
            // a poll-to-poll gap of > 1 ms means THIS wave was off the CU (queue pre-empted, CWSR, debugger): the deadline measures
            // waiting, not being descheduled, so it restarts (a healthy launch must not be poisoned by a 50 ms preemption)
            if (now - last > 100000LL) t0 = now;
            last = now;
            if (now - t0 > 5000000LL) {                                       // 50 ms at 100 MHz of actual polling: never in a healthy launch
                if (((v = ClScope<LOCAL>::poll(flags + tid)) & 0xffffffu) == epoch) break;      // one last look before giving up
                if (xerr) *xerr = 1;
                v = epoch | (my_xcc << 24);
                break;
            }
        }
        if (LOCAL && (v >> 24) != my_xcc && xerr) *xerr = 2;          // a peer on another XCD: its payload is not in this L2
    }
    __syncthreads();
}
}  // namespace

// QKV_ONLY: the first block's launch (LN + modulate + qkv of the incoming rows, no hand-off) as its own symbol
template <bool QKV_ONLY, bool LOCAL = false>
__global__ __launch_bounds__(RC_NW * 64) void dit_rowchain_cluster_kernel(const DitChainP p) {
    using Sc = ClScope<LOCAL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rc[];
    // chain buffers (they overlay the attention scratch, which is dead once the partials are merged)
    float* X1 = reinterpret_cast<float*>(smem_rc);                       // [32][X_LD] fp32: partial staging, then the residual stream
    float* P2 = X1 + RC_ROWS * X_LD;                                     // [32][X_LD] fp32: second partial staging
    float* PRM = P2 + RC_ROWS * X_LD;                                    // [8][256]: shift_mlp, scale_mlp, next shift, next scale, gate_msa, b_proj, gate_mlp, b_fc2
    float* RED = PRM + 8 * RC_H;                                         // [4][32][CL_RED_LD] fc1 K-half partials
    u16* As = reinterpret_cast<u16*>(RED + 4 * RC_ROWS * CL_RED_LD);     // [32][A_LD] bf16 A operand (LN outputs)
    u16* Ah = As + RC_ROWS * A_LD;                                       // [32][CL_AH_LD] attention output of this member's head, then its GELU(fc1) slice
    u16* QS = Ah + RC_ROWS * CL_AH_LD;                                   // [8][32][QK_LD] wave-private qkv store scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int N = p.rows_per_batch, tpb = (N + RC_ROWS - 1) / RC_ROWS;
    // p.xlocal: the four members of a cluster share blockIdx % 8, i.e. their XCD (clusters are dealt to the XCDs in rounds of 8;
    // the grid is padded to whole rounds and the clusters past the last tile leave here)
    // p.xcds < 8: only the first xcds XCDs take clusters (every XCD's L2 that takes part fetches the block's 1.5 MB of weights and the
    // K / V^T of its clusters from HBM once per launch: 8 L2s for 21 clusters was 27 MB of traffic for 4.4 MB of algorithmic bytes)
    if (p.xlocal && (int)(blockIdx.x & 7) >= p.xcds) return;
    const int cluster = p.xlocal ? (int)(blockIdx.x & 7) + p.xcds * (int)(blockIdx.x / (8 * DIT_CLUSTER)) : (int)blockIdx.x / DIT_CLUSTER;
    const int member = p.xlocal ? (int)(blockIdx.x >> 3) % DIT_CLUSTER : (int)blockIdx.x % DIT_CLUSTER;
    if (cluster >= p.B * tpb) return;
    const unsigned my_xcc = LOCAL ? xcc_id() : 0u;
    const unsigned flagv = p.epoch | (my_xcc << 24);
    const int b = cluster / tpb, n0 = (cluster - b * tpb) * RC_ROWS;
    const long mb = (long)b * N;
    const int step = p.step;
    const float* ada = p.ada + (long)step * 6 * RC_H;
    const bool has_q = p.next_shift != nullptr;
    float* slab = p.xslab + (long)cluster * DIT_CLUSTER_SLAB_FLOATS;     // [e][member][32*256 + 256]
    unsigned* flags = p.xflag + (long)cluster * DIT_CLUSTER_FLAG_WORDS;   // [e][member]
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (unsigned)(DIT_CLUSTER_SLAB_FLOATS * 4), 0x00020000);
    constexpr unsigned SLAB_B = (32 * 256 + 256) * 4;
    const int row = tid >> 4, seg = tid & 15;                            // the row-wise phases: 16 threads per row, 4 float4 each
    const int nrow = min(n0 + row, N - 1);

#ifdef DEX_TIMING
    long long cts[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    cts[0] = wall_clock64();
#endif
    float4 prm;                                                          // the parameter table entry this thread fetches
    {
        const int which = tid >> 6, c4 = (tid & 63) * 4;
        const float* src = which == 0 ? ada + 3 * RC_H : which == 1 ? ada + 4 * RC_H
                         : which == 2 ? (has_q ? p.next_shift + (long)step * p.next_step_stride : ada)
                         : which == 3 ? (has_q ? p.next_scale + (long)step * p.next_step_stride : ada)
                         : which == 4 ? ada + 2 * RC_H : which == 5 ? (QKV_ONLY ? ada : p.bp) : which == 6 ? ada + 5 * RC_H : (QKV_ONLY ? ada : p.b2);
        prm = *reinterpret_cast<const float4*>(src + c4);
    }
    float4 xr[4];                                                        // residual rows x of this thread's 16 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[q] = *reinterpret_cast<const float4*>(p.X + (mb + nrow) * RC_H + q * 64 + seg * 4);

    WT8 w8;
    f32x16 acc;
    if constexpr (!QKV_ONLY) {
        const int head = member >> 1, half = member & 1;
        // ---- attention of (head, key half): the 8 waves split the half's key tiles (attention_direct.hip's wave body)
        f32x16 ao[4];
        float am = -INFINITY, al = 0.f;
        {
            union DFr { uint4 u; lp8 v; };
            const long hbq = ((long)b * 2 + head) * p.Npad * 16;
            const uint4* Qg = reinterpret_cast<const uint4*>(p.Qin) + hbq + lane;
            const uint4* Kg = reinterpret_cast<const uint4*>(p.Kin) + hbq + lane;
            const uint4* Vg = reinterpret_cast<const uint4*>(p.Vin) + hbq + lane;
            const int ntiles = (N + 31) / 32;
            const int t_lo = half ? (ntiles + 1) / 2 : 0, t_hi = half ? ntiles : (ntiles + 1) / 2;
            DFr qf[8], kf[8], vf[4][2];
            int kt = t_lo + wave;
            {
                const uint4* qp = Qg + (long)(n0 >> 5) * 512;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) qf[ks].u = qp[ks * 64];
                const long t0 = min(kt, ntiles - 1);
                const uint4* kp = Kg + t0 * 512;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
                const uint4* vp = Vg + t0 * 512;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ao[t][r] = 0.f;
            while (kt < t_hi) {
                const int k0 = kt * 32, kn = kt + RC_NW;
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) sc = DEX_MFMA_LP(kf[ks].v, qf[ks].v, sc, 0, 0, 0);
                if (kn < t_hi) {
                    const uint4* kp = Kg + (long)kn * 512;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
                }
                if (k0 + 32 > N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sc[r] = -INFINITY;
                }
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (__builtin_amdgcn_ballot_w64(mx > am + 8.f) != 0) {       // lazy rescale (attention_direct.hip)
                    const float m_new = fmaxf(am, mx);
                    const float alpha = exp2f(am - m_new);
                    al *= alpha;
                    am = m_new;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ao[t][r] *= alpha;
                }
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - am); psum += sc[r]; }
                al += psum;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    DFr pb;
                    pb.u.x = pack2_lp(sc[8 * k2 + 0], sc[8 * k2 + 1]); pb.u.y = pack2_lp(sc[8 * k2 + 2], sc[8 * k2 + 3]);
                    pb.u.z = pack2_lp(sc[8 * k2 + 4], sc[8 * k2 + 5]); pb.u.w = pack2_lp(sc[8 * k2 + 6], sc[8 * k2 + 7]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) ao[t] = DEX_MFMA_LP(vf[t][k2].v, pb.v, ao[t], 0, 0, 0);
                }
                if (kn < t_hi) {
                    const uint4* vp = Vg + (long)kn * 512;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
                }
                kt = kn;
            }
            al += __shfl_xor(al, 32);
        }
#ifdef DEX_TIMING
    cts[1] = wall_clock64();
#endif
        wload8(w8, p.Wp, LO_WP, 16, wave, 8 * head, lane);               // proj weights: output tile `wave`, the head's K half
        // partial (m, l, O[query][d]) of this wave -> LDS scratch
        float* scr = reinterpret_cast<float*>(smem_rc);
        {
            float* oS = scr + wave * (32 * AT_LD);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<float4*>(oS + i * AT_LD + t * 32 + 8 * rq + 4 * hh) =
                        make_float4(ao[t][rq * 4 + 0], ao[t][rq * 4 + 1], ao[t][rq * 4 + 2], ao[t][rq * 4 + 3]);
            if (hh == 0) { scr[RC_NW * 32 * AT_LD + (wave * 2 + 0) * 32 + i] = am; scr[RC_NW * 32 * AT_LD + (wave * 2 + 1) * 32 + i] = al; }
        }
        lds_barrier();
        // merge the 8 wave partials of this member: O_c = sum_w f_w O_w / L_c (normalised by the member's own sum), (M_c, L_c) kept
        float myM, myL;
        float4 ov[2];
        {
            const float* stat = scr + RC_NW * 32 * AT_LD;
            float mw[RC_NW], M = -INFINITY;
#pragma unroll
            for (int w = 0; w < RC_NW; ++w) { mw[w] = stat[(w * 2) * 32 + row]; M = fmaxf(M, mw[w]); }
            float L = 0.f, f[RC_NW];
#pragma unroll
            for (int w = 0; w < RC_NW; ++w) { f[w] = (mw[w] == -INFINITY) ? 0.f : exp2f(mw[w] - M); L += f[w] * stat[(w * 2 + 1) * 32 + row]; }
            const float inv = L > 0.f ? 1.f / L : 0.f;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < RC_NW; ++w) {
                    const float4 t4 = *reinterpret_cast<const float4*>(scr + (w * 32 + row) * AT_LD + qq * 64 + seg * 4);
                    a.x = fmaf(f[w], t4.x, a.x); a.y = fmaf(f[w], t4.y, a.y); a.z = fmaf(f[w], t4.z, a.z); a.w = fmaf(f[w], t4.w, a.w);
                }
                ov[qq] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
            }
            myM = M; myL = L;
        }
        lds_barrier();                                   // scratch fully read: the chain's buffers may now be written
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            uint2 o;
            o.x = pack2_lp(ov[qq].x, ov[qq].y); o.y = pack2_lp(ov[qq].z, ov[qq].w);
            *reinterpret_cast<uint2*>(Ah + row * CL_AH_LD + qq * 64 + seg * 4) = o;
        }
        *reinterpret_cast<float4*>(PRM + (tid >> 6) * RC_H + (tid & 63) * 4) = prm;
        lds_barrier();

#ifdef DEX_TIMING
    cts[2] = wall_clock64();
#endif
        // ---- proj partial: P_c = O_c Wp[head rows, :]  (raw: bias, gate and residual are applied after the exchange)
        acc = zero16();
        mma8(acc, w8, Ah + i * CL_AH_LD + hh * 8);
        wload8(w8, p.W1, LO_W1, 16, 4 * member + (wave & 3), 8 * (wave >> 2), lane);      // fc1: column tile wave & 3 of this member's four, K half wave >> 2
        const int col = wave * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) X1[((r & 3) + 8 * (r >> 2) + 4 * hh) * X_LD + col] = acc[r];
        lds_barrier();

#ifdef DEX_TIMING
    cts[3] = wall_clock64();
#endif
        // ---- exchange 0: publish P_c and (M_c, L_c)
        {
            const unsigned base = (0 * DIT_CLUSTER + member) * SLAB_B;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(X1 + row * X_LD + q * 64 + seg * 4);
                const u32x4v u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(u, srs, base + (unsigned)(q * 512 + tid) * 16u, 0, Sc::ST_AUX);      // aux 16 = sc1: write-through
            }
            if (seg == 0) {
                const unsigned long long ml = (unsigned long long)__float_as_uint(myM) | ((unsigned long long)__float_as_uint(myL) << 32);
                Sc::st64(reinterpret_cast<unsigned long long*>(slab + (0 * DIT_CLUSTER + member) * (SLAB_B / 4) + 32 * 256) + row, ml);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0 && !(p.xdrop == 1 && member == 3)) Sc::st(flags + 0 * DIT_CLUSTER + member, flagv);
        }

#ifdef DEX_TIMING
    cts[4] = wall_clock64();
#endif
        cluster_wait<LOCAL>(flags + 0 * DIT_CLUSTER, member, p.epoch, my_xcc, tid, p.xerr);
#ifdef DEX_TIMING
    cts[5] = wall_clock64();
#endif

        {
            // every peer's partial of this thread's 16 columns + its row statistics: all loads first
            u32x4v pv[DIT_CLUSTER][4];
            float pm[DIT_CLUSTER], pl[DIT_CLUSTER];
#pragma unroll
            for (int c = 0; c < DIT_CLUSTER; ++c) {
                if (c == member) continue;
                const unsigned base = (0 * DIT_CLUSTER + c) * SLAB_B;
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[c][q] = __builtin_amdgcn_raw_buffer_load_b128(srs, base + (unsigned)(q * 512 + tid) * 16u, 0, Sc::LD_AUX);
                const unsigned long long ml = Sc::ld64(reinterpret_cast<const unsigned long long*>(slab + (0 * DIT_CLUSTER + c) * (SLAB_B / 4) + 32 * 256) + row);
                pm[c] = __uint_as_float((unsigned)ml); pl[c] = __uint_as_float((unsigned)(ml >> 32));
            }
            pm[member] = myM; pl[member] = myL;
            // flash merge weights of the two key halves of each head (fixed member order: results do not depend on arrival)
            float wgt[DIT_CLUSTER];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float M = fmaxf(pm[2 * h], pm[2 * h + 1]);
                const float fa = pl[2 * h] > 0.f ? pl[2 * h] * exp2f(pm[2 * h] - M) : 0.f, fb = pl[2 * h + 1] > 0.f ? pl[2 * h + 1] * exp2f(pm[2 * h + 1] - M) : 0.f;
                const float inv = 1.f / (fa + fb);
                wgt[2 * h] = fa * inv; wgt[2 * h + 1] = fb * inv;
            }
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 own = *reinterpret_cast<const float4*>(X1 + row * X_LD + q * 64 + seg * 4);
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < DIT_CLUSTER; ++c) {
                    const float4 pc = c == member ? own : make_float4(__uint_as_float(pv[c][q].x), __uint_as_float(pv[c][q].y), __uint_as_float(pv[c][q].z), __uint_as_float(pv[c][q].w));
                    sacc.x = fmaf(wgt[c], pc.x, sacc.x); sacc.y = fmaf(wgt[c], pc.y, sacc.y); sacc.z = fmaf(wgt[c], pc.z, sacc.z); sacc.w = fmaf(wgt[c], pc.w, sacc.w);
                }
                const float4 g = *reinterpret_cast<const float4*>(PRM + 4 * RC_H + q * 64 + seg * 4);
                const float4 bb = *reinterpret_cast<const float4*>(PRM + 5 * RC_H + q * 64 + seg * 4);
                v[q] = make_float4(xr[q].x + g.x * (sacc.x + bb.x), xr[q].y + g.y * (sacc.y + bb.y), xr[q].z + g.z * (sacc.z + bb.z), xr[q].w + g.w * (sacc.w + bb.w));
                *reinterpret_cast<float4*>(X1 + row * X_LD + q * 64 + seg * 4) = v[q];      // x1 (in place: this thread owns these elements)
            }
            ln_regs_to_A(v, As, PRM, PRM + RC_H, row, seg);      // LN + modulate straight from the registers
        }

#ifdef DEX_TIMING
    cts[6] = wall_clock64();
#endif
        lds_barrier();
#ifdef DEX_TIMING
    cts[7] = wall_clock64();
#endif

        // ---- fc1 + GELU for hidden columns 128 member .. +128: (column tile ct, K half kh) per wave
        {
            const int ct = wave & 3, kh = wave >> 2;
            acc = zero16();
            mma8(acc, w8, As + i * A_LD + hh * 8 + kh * 128);
            wload8(w8, p.W2, LO_W2, 32, wave, 8 * member, lane);                 // fc2: output tile `wave`, K = this member's hidden slice
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) RED[(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * CL_RED_LD + i] = acc[r];
            }
            lds_barrier();
            if (kh == 0) {
                const float b1v = p.b1[128 * member + ct * 32 + i];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    Ah[rr * CL_AH_LD + ct * 32 + i] = (u16)(pack2_lp(gelu_erf_rc(acc[r] + RED[(ct * 32 + rr) * CL_RED_LD + i] + b1v), 0.f) & 0xffffu);
                }
            }
            lds_barrier();
        }

#ifdef DEX_TIMING
    cts[8] = wall_clock64();
#endif
        // ---- fc2 partial over this member's hidden slice
        acc = zero16();
        mma8(acc, w8, Ah + i * CL_AH_LD + hh * 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) P2[((r & 3) + 8 * (r >> 2) + 4 * hh) * X_LD + col] = acc[r];

#ifdef DEX_TIMING
    cts[9] = wall_clock64();
#endif
        lds_barrier();
        {
            const unsigned base = (1 * DIT_CLUSTER + member) * SLAB_B;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(P2 + row * X_LD + q * 64 + seg * 4);
                const u32x4v u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(u, srs, base + (unsigned)(q * 512 + tid) * 16u, 0, Sc::ST_AUX);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !(p.xdrop == 1 && member == 3)) Sc::st(flags + 1 * DIT_CLUSTER + member, flagv);
        }

#ifdef DEX_TIMING
    cts[10] = wall_clock64();
#endif
        cluster_wait<LOCAL>(flags + 1 * DIT_CLUSTER, member, p.epoch, my_xcc, tid, p.xerr);
#ifdef DEX_TIMING
    cts[11] = wall_clock64();
#endif

        {
            u32x4v pv[DIT_CLUSTER][4];
#pragma unroll
            for (int c = 0; c < DIT_CLUSTER; ++c) {
                if (c == member) continue;
                const unsigned base = (1 * DIT_CLUSTER + c) * SLAB_B;
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[c][q] = __builtin_amdgcn_raw_buffer_load_b128(srs, base + (unsigned)(q * 512 + tid) * 16u, 0, Sc::LD_AUX);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 own = *reinterpret_cast<const float4*>(P2 + row * X_LD + q * 64 + seg * 4);
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < DIT_CLUSTER; ++c) {
                    const float4 pc = c == member ? own : make_float4(__uint_as_float(pv[c][q].x), __uint_as_float(pv[c][q].y), __uint_as_float(pv[c][q].z), __uint_as_float(pv[c][q].w));
                    sacc.x += pc.x; sacc.y += pc.y; sacc.z += pc.z; sacc.w += pc.w;
                }
                const float4 g = *reinterpret_cast<const float4*>(PRM + 6 * RC_H + q * 64 + seg * 4);
                const float4 bb = *reinterpret_cast<const float4*>(PRM + 7 * RC_H + q * 64 + seg * 4);
                const float4 x1 = *reinterpret_cast<const float4*>(X1 + row * X_LD + q * 64 + seg * 4);
                const float4 x2 = make_float4(x1.x + g.x * (sacc.x + bb.x), x1.y + g.y * (sacc.y + bb.y), x1.z + g.z * (sacc.z + bb.z), x1.w + g.w * (sacc.w + bb.w));
                xr[q] = x2;                                                 // (the residual registers are free: they now carry x2 into the LayerNorm)
                if (q == member && n0 + row < N) *reinterpret_cast<float4*>(p.X + (mb + n0 + row) * RC_H + q * 64 + seg * 4) = x2;    // member c owns columns 64 c .. +64 of X
            }
        }

#ifdef DEX_TIMING
    cts[12] = wall_clock64();
#endif
        if (!has_q) return;
    } else {
        // first block: LN + modulate + qkv of the incoming token rows only
        *reinterpret_cast<float4*>(PRM + (tid >> 6) * RC_H + (tid & 63) * 4) = prm;
        lds_barrier();                                   // the parameter table is complete
    }
    // ---- next block's qkv: column tiles 6 member .. 6 member + 5 on waves 0-5 (full K)
    WT16 wq;
    const int nt = 6 * member + wave;
    if (wave < 6) wload(wq, p.Wq, LO_WQ, 16, nt, 0, lane);
    ln_regs_to_A(xr, As, PRM + 2 * RC_H, PRM + 3 * RC_H, row, seg);      // xr = x2 (or the incoming rows of the first block)
    lds_barrier();

#ifdef DEX_TIMING
    cts[13] = wall_clock64();
#endif
    if (wave < 6) {
        acc = zero16();
        if (nt < 16) {                 // (wave-uniform) q / k tiles: transposed product, fragment layout straight from the accumulators
            const QkvBias qb = qkv_bias<0>(p, nt, lane);
            mma16t(acc, wq, As + i * A_LD + hh * 8);
            if (nt < 8) store_qkv_tile_d<0>(p, acc, qb, nt, b, n0, lane); else store_qkv_tile_d<1>(p, acc, qb, nt, b, n0, lane);
        } else {
            const QkvBias qb = qkv_bias<2>(p, nt, lane);
            mma16(acc, wq, As + i * A_LD + hh * 8);
            store_qkv_tile_d<2>(p, acc, qb, nt, b, n0, lane);
        }
    }
#ifdef DEX_TIMING
    cts[14] = wall_clock64();
    if (p.dbg && tid == 0) for (int q = 0; q < 16; ++q) p.dbg[blockIdx.x * 16 + q] = cts[q];
#endif
}
constexpr size_t RC_LDS_CLUSTER_CHAIN = (size_t)(2 * RC_ROWS * X_LD + 8 * RC_H + 4 * RC_ROWS * CL_RED_LD) * sizeof(float)
                                        + (size_t)(RC_ROWS * A_LD + RC_ROWS * CL_AH_LD + RC_NW * RC_ROWS * QK_LD) * sizeof(u16);
constexpr size_t RC_LDS_CLUSTER = RC_LDS_CLUSTER_CHAIN > RC_LDS_ATTN ? RC_LDS_CLUSTER_CHAIN : RC_LDS_ATTN;

// the cluster form needs every workgroup of the launch resident at once (<= one per CU, 256 CUs): B x tiles x 4 <= 256
bool dit_rowchain_cluster_form(int rows_per_batch, int B) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_RCC)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    const int on = knob_or("DEX_DIT_CLUSTER", 1);       // (A/B tests flip it)
    return on && (long)B * ((rows_per_batch + RC_ROWS - 1) / RC_ROWS) * DIT_CLUSTER <= 256;
}

// XCD-local clusters (DitChainP::xlocal): the grid is padded to whole rounds of 8 clusters, still one workgroup per CU at most
bool dit_rowchain_cluster_local_fits(int rows_per_batch, int B) {
    const int on = knob_or("DEX_DIT_CLUSTER_LOCAL", 1);
    const long tiles = (long)B * ((rows_per_batch + RC_ROWS - 1) / RC_ROWS);
    return on && dit_rowchain_cluster_form(rows_per_batch, B) && (tiles + 7) / 8 * 8 * DIT_CLUSTER <= 256;
}
// how many XCDs the XCD-local clusters of this launch are dealt to: all 8 by default (packing them onto fewer XCDs cuts the L2
// re-fetches but measured no faster: profiles/round4_cluster_xcd_packing_ab.txt); DEX_DIT_XCDS = 1..7 packs them, raised to the
// fewest XCDs that still give every workgroup a CU of its own (32 CUs per XCD).  tests/test_gpu_cluster.py covers xcds = 3.
int dit_rowchain_cluster_xcds(int rows_per_batch, int B) {
    const long tiles = (long)B * ((rows_per_batch + RC_ROWS - 1) / RC_ROWS);
    const int forced = knob_or("DEX_DIT_XCDS", 0);
    int x = forced >= 1 && forced <= 8 ? forced : 8;
    while (x < 8 && (tiles + x - 1) / x * DIT_CLUSTER > 32) ++x;          // (a forced value that does not fit is raised)
    return x;
}

bool dit_rowchain_supported(int hidden, int mlp_hidden) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_RC)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return hidden == RC_H && mlp_hidden == RC_MLP; }
bool dit_rowchain64_form(int rows_per_batch, int B, int attn_inline) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_RC64)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    const int m64 = knob_or("DEX_ROWCHAIN64", 1);
    const long wg32 = (long)B * ((rows_per_batch + RC_ROWS - 1) / RC_ROWS);
    const bool small_n = dit_sep64_small_n(rows_per_batch, B);       // (kernels.h: GeDEX B = 32)
    return m64 && !attn_inline && (m64 == 2 || wg32 >= 768 || small_n);
}

void launch_dit_rowchain(const DitChainP& p, hipStream_t st) {
    if (p.xslab && (p.attn_inline || p.qkv_only) && dit_rowchain_cluster_form(p.rows_per_batch, p.B)) {
        static bool attrc = false;
        if (!attrc) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_cluster_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS_CLUSTER);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_cluster_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS_CLUSTER);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_cluster_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS_CLUSTER);
            attrc = true;
        }
        const int tiles = p.B * ((p.rows_per_batch + RC_ROWS - 1) / RC_ROWS);
        const int xc = p.xlocal ? (p.xcds >= 1 && p.xcds <= 8 ? p.xcds : 8) : 8;
        const dim3 gridc((p.xlocal ? (tiles + xc - 1) / xc * 8 : tiles) * DIT_CLUSTER);
        if (p.qkv_only) hipLaunchKernelGGL((dit_rowchain_cluster_kernel<true, false>), gridc, dim3(RC_NW * 64), RC_LDS_CLUSTER, st, p);
        else if (p.xdrop == 2) {        // tests only: L2-scope hand-offs between members dealt to DIFFERENT XCDs - must end as an error, never as a mel
            DitChainP q = p; q.xlocal = 0;
            hipLaunchKernelGGL((dit_rowchain_cluster_kernel<false, true>), dim3(tiles * DIT_CLUSTER), dim3(RC_NW * 64), RC_LDS_CLUSTER, st, q);
        } else if (p.xlocal) {
            g_last_symbol = "dit_rowchain_cluster_kernel<false,true>";
            hipLaunchKernelGGL((dit_rowchain_cluster_kernel<false, true>), gridc, dim3(RC_NW * 64), RC_LDS_CLUSTER, st, p);
        } else {
            g_last_symbol = "dit_rowchain_cluster_kernel<false,false>";
            hipLaunchKernelGGL((dit_rowchain_cluster_kernel<false, false>), gridc, dim3(RC_NW * 64), RC_LDS_CLUSTER, st, p);
        }
        return;
    }
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(RC_LDS_ATTN > RC_LDS ? RC_LDS_ATTN : RC_LDS));
        attr = true;
    }
    {   // batch regime with the attention as its own launch: the 64-row form (DEX_ROWCHAIN64=0: never, 2: whenever attention is separate)
        if (dit_rowchain64_form(p.rows_per_batch, p.B, p.attn_inline)) {
            static bool attr64 = false;
            if (!attr64) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC64_LDS);
                attr64 = true;
            }
            if (knob_or("DEX_ROWCHAIN64A", 1) && (p.qkv_only || (p.o_lp && p.ksplit <= 1 && p.tail_ks <= 1))) {   // the generated streams (0: the round-3 kernel)
                static bool attr64a = false;
                if (!attr64a) {
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain64a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCA_LDS_BYTES);
                    attr64a = true;
                }
                g_last_symbol = "dit_rowchain64a_kernel";
                const int ncu = device_cus();
                const int ntiles = p.B * ((p.rows_per_batch + 63) / 64);
                hipLaunchKernelGGL(dit_rowchain64a_kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), RCA_LDS_BYTES, st, p);      // persistent: one workgroup per CU walks its tiles
                return;
            }
            g_last_symbol = "dit_rowchain64_kernel";
            hipLaunchKernelGGL(dit_rowchain64_kernel, dim3(p.B * ((p.rows_per_batch + 63) / 64)), dim3(RC_NW * 64), RC64_LDS, st, p);
            return;
        }
    }
    dim3 grid(p.B * ((p.rows_per_batch + RC_ROWS - 1) / RC_ROWS));
    if (p.attn_inline && !p.qkv_only) {
        DitChainP q = p;
        q.xcd_map = (p.B % 8 == 0 && !knob_off("DEX_XCD_MAP")) ? 1 : 0;      // 0: the plain b-major order
        hipLaunchKernelGGL(dit_rowchain_kernel<true>, grid, dim3(RC_NW * 64), RC_LDS_ATTN > RC_LDS ? RC_LDS_ATTN : RC_LDS, st, q);
    }
    else
        hipLaunchKernelGGL(dit_rowchain_kernel<false>, grid, dim3(RC_NW * 64), RC_LDS, st, p);
}

// fp32 [K][N] -> bf16 in MFMA B-fragment order: dst[((nt * K/16 + ks) * 64 + lane) * 8 + j] =
// W[k = ks*16 + (lane >> 5)*8 + j][n = nt*32 + (lane & 31)]
__global__ void pack_lp_frag_kernel(const float* __restrict__ src, u16* __restrict__ dst, int K, int N) {
    const long total = (long)K * N;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int j = (int)(o & 7), lane = (int)((o >> 3) & 63);
        const long t = o >> 9;
        const int ks = (int)(t % (K / 16)), nt = (int)(t / (K / 16));
        const int k = ks * 16 + (lane >> 5) * 8 + j, n = nt * 32 + (lane & 31);
        dst[o] = (u16)(pack2_lp(src[(long)k * N + n], 0.f) & 0xffffu);
    }
}
// same fragment order from a row-major [N][K] source (nn.Linear / 1x1-conv weight layout)
__global__ void pack_lp_frag_nk_kernel(const float* __restrict__ src, u16* __restrict__ dst, int K, int N) {
    const long total = (long)K * N;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int j = (int)(o & 7), lane = (int)((o >> 3) & 63);
        const long t = o >> 9;
        const int ks = (int)(t % (K / 16)), nt = (int)(t / (K / 16));
        const int k = ks * 16 + (lane >> 5) * 8 + j, n = nt * 32 + (lane & 31);
        dst[o] = (u16)(pack2_lp(src[(long)n * K + k], 0.f) & 0xffffu);
    }
}
void launch_pack_lp_frag_nk(const float* src, void* dst, int K, int N, hipStream_t st) {
    hipLaunchKernelGGL(pack_lp_frag_nk_kernel, dim3(256), dim3(256), 0, st, src, reinterpret_cast<u16*>(dst), K, N);
}
void launch_pack_lp_frag(const float* src, void* dst, int K, int N, hipStream_t st) {
    hipLaunchKernelGGL(pack_lp_frag_kernel, dim3(256), dim3(256), 0, st, src, reinterpret_cast<u16*>(dst), K, N);
}

}  // namespace DEX_LP_NS
}  // namespace dex
