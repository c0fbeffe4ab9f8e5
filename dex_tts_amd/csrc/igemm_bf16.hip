// igemm_bf16.hip — implicit-GEMM convolution / linear with bf16 operands on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate, fp32 activations in HBM converted to bf16 while staging; weights pre-packed bf16 [N][K]).
// Same gather semantics and the same epilogue as igemm.hip.  LDS tiles are row-major [rows][BK+8] bf16:
// the 16-byte pad makes every ds_read_b128 / ds_write_b128 lane group hit 64 distinct banks.
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"
#include "igemm_epilogue.h"

namespace dex {
namespace DEX_LP_NS {

typedef unsigned short u16;


// one K tile of global loads into a register slot: raw fp32 activations (addresses clamped, validity folded into
// the mask factor) and bf16 weights; nothing here waits on memory
template <int BN, int BK, bool PT, int AP, int RPP, int BP, bool ALP = false>
__device__ __forceinline__ void igemm_load_tile(const IGemmP& p, float4 (&fa)[AP][2], float (&fm)[AP], float (&fo)[AP], uint4& rb0, uint4& rb1,
#ifdef DEX_LP_WSPLIT
                                                uint4& rl0, uint4& rl1,      // the lo halves of the same weights (p.w_lo_off elements behind)
#endif
                                                int k0, bool live, int tk8, int trow,
                                                const int (&bh)[AP], const int (&bw)[AP], unsigned mvbits,
                                                const float* Ab, const float* mrow, int mws, const u16* Wb, int n0) {
    const int kthr = PT ? k0 + tk8 : k0;                     // tap resolution per thread or per tile
    const int tap = kthr / p.Cin, c0 = kthr - tap * p.Cin + (PT ? 0 : tk8);
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int hi = bh[j] + kh * p.step_h, wi = bw[j] + kw * p.step_w;
        const bool ok = ((mvbits >> j) & 1u) && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
        const int hc = ok ? hi : 0, wc = ok ? wi : 0;
        const long eo = ((long)hc * p.Wi + wc) * p.lda + c0;
        if constexpr (ALP) fa[j][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const u16*>(Ab) + eo);      // 8 raw 16-bit values
        else {
            fa[j][0] = *reinterpret_cast<const float4*>(Ab + eo);
            fa[j][1] = *reinterpret_cast<const float4*>(Ab + eo + 4);
        }
        fm[j] = mrow[wc * mws];                // raw mask value (a dummy in-bounds load when there is no mask): NO op on it here
        fo[j] = (ok && live) ? 1.f : 0.f;      // validity from indices only; tiles past the K range (ring padding) give exact zeros
    }
    if (BN >= RPP || trow < BN) rb0 = *reinterpret_cast<const uint4*>(Wb + (long)(n0 + trow) * p.K + k0 + tk8);
    if constexpr (BP > 1) rb1 = *reinterpret_cast<const uint4*>(Wb + (long)(n0 + trow + RPP) * p.K + k0 + tk8);
#ifdef DEX_LP_WSPLIT
    if (BN >= RPP || trow < BN) rl0 = *reinterpret_cast<const uint4*>(Wb + p.w_lo_off + (long)(n0 + trow) * p.K + k0 + tk8);
    if constexpr (BP > 1) rl1 = *reinterpret_cast<const uint4*>(Wb + p.w_lo_off + (long)(n0 + trow + RPP) * p.K + k0 + tk8);
#endif
}

// PT: the 8-element chunk of every thread resolves its own tap (k -> (kh,kw,c)); needed when Cin < BK, e.g. the
// grouped 16x16 pos-conv (Cin = 32 per group) with BK = 128: 8 K iterations instead of 32 exposed round trips.
// D: K tiles of global loads kept in flight ahead of the MFMAs (raw fp32 in registers, converted when they are
// stored to LDS).  At B=1 a K tile is ~0.1 us of MFMA behind a ~1.3 us round trip: D=3 cuts the exposed trips 3x.
// NKT > 0: the K-tile count is a compile-time constant and the ring loop is fully unrolled — only in straight-line
// code does the compiler keep exact vmcnt(N) waits (inside a loop it falls back to vmcnt(0) at every LDS store,
// which serialises the ring: measured 19 -> 24.5 us on the stride-2 downsample conv before unrolling).
// ALP / CLP: the A / C tensors hold 16-bit elements (IGemmP::a_lp / c_lp).  Compile-time: as runtime flags the extra branches
// cost the ring its exact vmcnt bookkeeping (the Downsample conv went from 140 to 279 us at B=32).
// LR: leaky_relu(x, IGemmP::act_in_slope) on the gathered A elements (the vocoder's activation in front of every convolution),
// compile-time for the same reason.
template <int BM, int BN, int BK, bool PT = false, int D = 1, int NKT = 0, bool ALP = false, bool CLP = false, bool LR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void igemm_lp_kernel(const IGemmP p) {
    constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (WM * 32);
    constexpr int TPR = BK / 8;                 // threads per tile row (8 elements each)
    constexpr int RPP = 256 / TPR;              // rows per pass
    constexpr int AP = BM / RPP;                // A passes
    constexpr int BP = (BN + RPP - 1) / RPP;    // B passes
    constexpr int LDS_LD = BK + 8;              // bf16 elements per LDS row
    static_assert(MT >= 1 && AP >= 1, "tile");
    __shared__ __attribute__((aligned(16))) u16 As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) u16 Bs[BN * LDS_LD];
#ifdef DEX_LP_WSPLIT
    __shared__ __attribute__((aligned(16))) u16 Bl[BN * LDS_LD];       // lo halves of the weight tile: a second MFMA per product
    const bool has_lo = p.w_lo_off != 0;                               // (a 16-bit operand built at run time has none: one MFMA)
#define WS_RL(d) , rl0[d], rl1[d]
#else
#define WS_RL(d)
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int z = blockIdx.z, par = 0;
    if (p.parity) { par = z & 3; z >>= 2; }
    const int s = z % p.ksplit;
    const int g = (z / p.ksplit) % p.groups;
    const int b = z / (p.ksplit * p.groups);
    const int off_h = p.parity ? (par >> 1) : p.off_h, off_w = p.parity ? (par & 1) : p.off_w;
    const int oh0 = p.parity ? (par >> 1) : p.oh0, ow0 = p.parity ? (par & 1) : p.ow0;
    const int M = p.Ho * p.Wo;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch), each XCD has a private L2, and
    // neighbouring pixel tiles share their 3x3 halo -> give every XCD one contiguous range of tiles.
    int mtile = blockIdx.x;
    if ((gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = mtile * BM, n0 = blockIdx.y * BN;
    const int Kper = p.K / p.ksplit, kbeg = s * Kper, nkt = NKT ? NKT : Kper / BK;

    // (a_lp: 16-bit elements behind the same pointer - halve the element offsets' byte scale)
    const float* Ab = ALP ? reinterpret_cast<const float*>(reinterpret_cast<const u16*>(p.A) + (long)b * p.a_bstride + p.a_coff + g * p.Cin)
                          : p.A + (long)b * p.a_bstride + p.a_coff + g * p.Cin;
    // both arms are kernel-argument (global) pointers: a select against a __device__ constant would degrade the
    // mask loads to FLAT, and an outstanding FLAT load forces vmcnt(0) waits everywhere
    const float* mrow = p.inmask ? p.inmask + (long)b * p.mask_bstride : p.A;
    const int mws = p.inmask ? p.inmask_ws : 0;
    const bool has_mask = p.inmask != nullptr;
    const int trow = tid / TPR, tk8 = (tid % TPR) * 8;
    int bh[AP], bw[AP];
    unsigned mvbits = 0;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int m = m0 + trow + RPP * j;
        mvbits |= (m < M ? 1u : 0u) << j;
        const int ho = m / p.Wo, wo = m - ho * p.Wo;
        bh[j] = ho * p.sh + off_h;
        bw[j] = wo * p.sw + off_w;
    }
    // bf16 weights [N][K] (per group / per batch strides are in elements of the fp32 [K][N] pack: same count)
    const u16* Wb = reinterpret_cast<const u16*>(p.Wbf) + (long)b * p.w_bstride + (long)g * p.w_gstride + (long)par * p.K * p.N;

    float4 fa[D][AP][2];
    float fm[D][AP], fo[D][AP];
    uint4 rb0[D], rb1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { rb0[d] = make_uint4(0, 0, 0, 0); rb1[d] = rb0[d]; }
#ifdef DEX_LP_WSPLIT
    uint4 rl0[D], rl1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { rl0[d] = make_uint4(0, 0, 0, 0); rl1[d] = rl0[d]; }
#endif
    static_assert(BP <= 2, "B passes");
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // The ring runs over ceil(nkt / D) * D tiles with NO data-dependent control flow (tiles past the end re-read the
    // last tile with a zero mask): the waitcnt bookkeeping stays exact (vmcnt = loads of the D-1 younger tiles)
    // instead of collapsing to vmcnt(0) at every conditional.
    const int nkt_pad = (nkt + D - 1) / D * D;
#pragma unroll
    for (int d = 0; d < D; ++d)
        igemm_load_tile<BN, BK, PT, AP, RPP, BP, ALP>(p, fa[d], fm[d], fo[d], rb0[d], rb1[d] WS_RL(d), kbeg + min(d, nkt - 1) * BK, d < nkt, tk8, trow, bh, bw, mvbits, Ab, mrow, mws, Wb, n0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll(NKT ? (NKT + D - 1) / D : 1)
    for (int kt0 = 0; kt0 < nkt_pad; kt0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kt = kt0 + d;
            lds_barrier();                  // previous tile's MFMAs are done with the LDS tiles
            __builtin_amdgcn_sched_barrier(0);     // keep the conversion (and its vmcnt wait) of this slot below the barrier
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const float mk = has_mask ? mul_pinned(fm[d][j], fo[d][j]) : fo[d][j];
                float4 f0 = fa[d][j][0], f1 = fa[d][j][1];
                if constexpr (ALP) {       // the 16-bit values widened exactly; mask 0 / 1 leaves them on the operand grid
                    const unsigned u0 = __float_as_uint(f0.x), u1 = __float_as_uint(f0.y), u2 = __float_as_uint(f0.z), u3 = __float_as_uint(f0.w);
                    f0 = make_float4(lp_lo(u0), lp_hi(u0), lp_lo(u1), lp_hi(u1));
                    f1 = make_float4(lp_lo(u2), lp_hi(u2), lp_lo(u3), lp_hi(u3));
                }
                if constexpr (LR) {        // leaky_relu = max(x, slope x) for 0 < slope < 1
                    const float sl = p.act_in_slope;
                    f0 = make_float4(fmaxf(f0.x, f0.x * sl), fmaxf(f0.y, f0.y * sl), fmaxf(f0.z, f0.z * sl), fmaxf(f0.w, f0.w * sl));
                    f1 = make_float4(fmaxf(f1.x, f1.x * sl), fmaxf(f1.y, f1.y * sl), fmaxf(f1.z, f1.z * sl), fmaxf(f1.w, f1.w * sl));
                }
                uint4 v;
                v.x = pack2_mul_lp_pinned(f0.x, f0.y, mk); v.y = pack2_mul_lp_pinned(f0.z, f0.w, mk);
                v.z = pack2_mul_lp_pinned(f1.x, f1.y, mk); v.w = pack2_mul_lp_pinned(f1.z, f1.w, mk);
                *reinterpret_cast<uint4*>(As + (trow + RPP * j) * LDS_LD + tk8) = v;
            }
            if (BN >= RPP || trow < BN) *reinterpret_cast<uint4*>(Bs + trow * LDS_LD + tk8) = rb0[d];
            if constexpr (BP > 1) *reinterpret_cast<uint4*>(Bs + (trow + RPP) * LDS_LD + tk8) = rb1[d];
#ifdef DEX_LP_WSPLIT
            // (no branch inside the MFMA stream: an operand without a lo half gets a zero lo tile)
            if (BN >= RPP || trow < BN) *reinterpret_cast<uint4*>(Bl + trow * LDS_LD + tk8) = has_lo ? rl0[d] : make_uint4(0, 0, 0, 0);
            if constexpr (BP > 1) *reinterpret_cast<uint4*>(Bl + (trow + RPP) * LDS_LD + tk8) = has_lo ? rl1[d] : make_uint4(0, 0, 0, 0);
#endif
            lds_barrier();
            igemm_load_tile<BN, BK, PT, AP, RPP, BP, ALP>(p, fa[d], fm[d], fo[d], rb0[d], rb1[d] WS_RL(d), kbeg + min(kt + D, nkt - 1) * BK, kt + D < nkt, tk8, trow, bh, bw, mvbits, Ab, mrow, mws, Wb, n0);
            __builtin_amdgcn_sched_barrier(0);     // the scheduler otherwise sinks these loads down to their first use
            const u16* ap = As + (wm * (MT * 32) + i) * LDS_LD + hh * 8;
            const u16* bp = Bs + (wn * 32 + i) * LDS_LD + hh * 8;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const lp8 bf = *reinterpret_cast<const lp8*>(bp + ks * 16);
#ifdef DEX_LP_WSPLIT
                const lp8 bl = *reinterpret_cast<const lp8*>(bp + (Bl - Bs) + ks * 16);
#endif
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const lp8 af = *reinterpret_cast<const lp8*>(ap + t * 32 * LDS_LD + ks * 16);
                    acc[t] = DEX_MFMA_LP(af, bf, acc[t], 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                    acc[t] = DEX_MFMA_LP(af, bl, acc[t], 0, 0, 0);
#endif
                }
            }
        }
    }
#undef WS_RL
    igemm_epilogue<MT, CLP>(p, acc, m0, n0, wm * (MT * 32), wn * 32, lane, b, g, s, M, oh0, ow0);
}

// ---- single-shot variant (K <= 512): the whole K extent of the A and B tiles is staged at once, so a
// workgroup makes ONE global round trip before its MFMA chain instead of one per K tile (at B=1 the DiT linears
// are pure latency: 4-8 exposed round trips at ~1.2 us each).  Optional fused prologue on the A rows:
// LayerNorm(eps 1e-6, no affine) + adaLN modulate (dit.py:78-79,288-289) when the row IS the K extent.
// ALP: A holds 16-bit elements (IGemmP::a_lp; no LayerNorm staging): one 16-byte load per item, rounded once at its producer.
template <int BM, int BN, int K, bool ALP = false>
__global__ __launch_bounds__(256) void igemm_lp_ss_kernel(const IGemmP p) {
    constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (WM * 32);
    constexpr int LDS_LD = K + 8, KC = K / 8;               // KC: 8-element chunks per row (power of two <= 64)
    constexpr int AIT = BM * KC / 256, BIT = BN * KC / 256; // items per thread
    constexpr int ABATCH = AIT > 8 ? 8 : AIT;               // A items in flight per batch (2 float4 each)
    static_assert(MT >= 1 && AIT >= 1 && BIT >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) u16 smem_ss[];
    u16* As = smem_ss;
    u16* Bs = smem_ss + BM * LDS_LD;
#ifdef DEX_LP_WSPLIT
    const bool has_lo = p.w_lo_off != 0;                     // (a 16-bit operand built at run time has no lo half: one MFMA per product)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int b = blockIdx.z, par = 0;
    if (p.parity) { par = b & 3; b >>= 2; }
    const int off_h = p.parity ? (par >> 1) : p.off_h, off_w = p.parity ? (par & 1) : p.off_w;
    const int oh0 = p.parity ? (par >> 1) : p.oh0, ow0 = p.parity ? (par & 1) : p.ow0;
    const int M = p.Ho * p.Wo;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const float* Ab = p.A + (long)b * p.a_bstride + p.a_coff;
    const float* mrow = p.inmask ? p.inmask + (long)b * p.mask_bstride : nullptr;
    const u16* Wb = reinterpret_cast<const u16*>(p.Wbf) + (long)b * p.w_bstride + (long)par * K * p.N;
    const int step = p.step;
    const float* lsh = p.ln_shift ? p.ln_shift + (long)step * p.ln_step_stride : nullptr;
    const float* lsc = p.ln_scale ? p.ln_scale + (long)step * p.ln_step_stride : nullptr;

    // ---- issue: all B loads, then the first batch of A loads (everything in flight together)
    uint4 br[BIT];
#pragma unroll
    for (int j = 0; j < BIT; ++j) {
        const int it = tid + 256 * j;
        const int n = it / KC, c8 = (it % KC) * 8;
        br[j] = *reinterpret_cast<const uint4*>(Wb + (long)(n0 + n) * K + c8);
    }
#ifdef DEX_LP_WSPLIT
    // the lo halves of this wave's weight columns go straight into B-operand registers (lane = column i, K half hh: 16 contiguous
    // bytes of a weight row per K-step) - a second LDS tile would halve the workgroups per CU (measured: 14 -> 32 us at B = 1)
    uint4 bl_[K / 16];
    {
        const u16* wl = Wb + p.w_lo_off + (long)(n0 + wn * 32 + i) * K + hh * 8;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) bl_[ks] = has_lo ? *reinterpret_cast<const uint4*>(wl + ks * 16) : make_uint4(0, 0, 0, 0);
    }
#endif
#pragma unroll
    for (int a0 = 0; a0 < AIT; a0 += ABATCH) {
        float4 f0[ABATCH], f1[ABATCH];
        float mk[ABATCH];
#pragma unroll
        for (int j = 0; j < ABATCH; ++j) {
            const int it = tid + 256 * (a0 + j);
            const int row = it / KC, k8 = (it % KC) * 8;
            const int m = m0 + row;
            const int tap = k8 / p.Cin, c0 = k8 - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int mm = m < M ? m : 0;
            const int ho = mm / p.Wo, wo = mm - ho * p.Wo;
            const int hi = ho * p.sh + off_h + kh * p.step_h, wi = wo * p.sw + off_w + kw * p.step_w;
            const bool ok = m < M && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            const int hc = ok ? hi : 0, wc = ok ? wi : 0;
            if constexpr (ALP) {
                const u16* srch = reinterpret_cast<const u16*>(p.A) + (long)b * p.a_bstride + p.a_coff + ((long)hc * p.Wi + wc) * p.lda + c0;
                f0[j] = *reinterpret_cast<const float4*>(srch);          // 8 raw 16-bit values
                f1[j] = f0[j];
            } else {
                const float* src = Ab + ((long)hc * p.Wi + wc) * p.lda + c0;
                f0[j] = *reinterpret_cast<const float4*>(src);
                f1[j] = *reinterpret_cast<const float4*>(src + 4);
            }
            const float mv = mrow ? mrow[wc * p.inmask_ws] : 1.f;
            mk[j] = ok ? mv : 0.f;
        }
        if (a0 == 0) {
#pragma unroll
            for (int j = 0; j < BIT; ++j) {
                const int it = tid + 256 * j;
                const int n = it / KC, c8 = (it % KC) * 8;
                *reinterpret_cast<uint4*>(Bs + n * LDS_LD + c8) = br[j];
            }
        }
#pragma unroll
        for (int j = 0; j < ABATCH; ++j) {
            const int it = tid + 256 * (a0 + j);
            const int row = it / KC, k8 = (it % KC) * 8;
            float4 a = f0[j], c = f1[j];
            if constexpr (ALP) {
                const unsigned u0 = __float_as_uint(f0[j].x), u1 = __float_as_uint(f0[j].y), u2 = __float_as_uint(f0[j].z), u3 = __float_as_uint(f0[j].w);
                a = make_float4(lp_lo(u0), lp_hi(u0), lp_lo(u1), lp_hi(u1)); c = make_float4(lp_lo(u2), lp_hi(u2), lp_lo(u3), lp_hi(u3));
            }
            if (lsh) {
                // row statistics over the KC lanes that share this row (lanes contiguous, KC a power of two)
                float s = (a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w);
#pragma unroll
                for (int o = 1; o < KC; o <<= 1) s += __shfl_xor(s, o);
                const float mean = s * (1.f / (float)K);
                a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean; c.x -= mean; c.y -= mean; c.z -= mean; c.w -= mean;
                float q = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
#pragma unroll
                for (int o = 1; o < KC; o <<= 1) q += __shfl_xor(q, o);
                const float rstd = rsqrtf(q * (1.f / (float)K) + 1e-6f);
                const float4 sc0 = *reinterpret_cast<const float4*>(lsc + k8), sc1 = *reinterpret_cast<const float4*>(lsc + k8 + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(lsh + k8), sh1 = *reinterpret_cast<const float4*>(lsh + k8 + 4);
                a.x = a.x * rstd * (1.f + sc0.x) + sh0.x; a.y = a.y * rstd * (1.f + sc0.y) + sh0.y;
                a.z = a.z * rstd * (1.f + sc0.z) + sh0.z; a.w = a.w * rstd * (1.f + sc0.w) + sh0.w;
                c.x = c.x * rstd * (1.f + sc1.x) + sh1.x; c.y = c.y * rstd * (1.f + sc1.y) + sh1.y;
                c.z = c.z * rstd * (1.f + sc1.z) + sh1.z; c.w = c.w * rstd * (1.f + sc1.w) + sh1.w;
            }
            const float m_ = mk[j];
            uint4 v;
            v.x = pack2_lp(a.x * m_, a.y * m_); v.y = pack2_lp(a.z * m_, a.w * m_);
            v.z = pack2_lp(c.x * m_, c.y * m_); v.w = pack2_lp(c.z * m_, c.w * m_);
            *reinterpret_cast<uint4*>(As + row * LDS_LD + k8) = v;
        }
    }
    __syncthreads();
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const u16* ap = As + (wm * (MT * 32) + i) * LDS_LD + hh * 8;
    const u16* bp = Bs + (wn * 32 + i) * LDS_LD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
        const lp8 bf = *reinterpret_cast<const lp8*>(bp + ks * 16);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const lp8 af = *reinterpret_cast<const lp8*>(ap + t * 32 * LDS_LD + ks * 16);
            acc[t] = DEX_MFMA_LP(af, bf, acc[t], 0, 0, 0);
#ifdef DEX_LP_WSPLIT
            acc[t] = DEX_MFMA_LP(af, __builtin_bit_cast(lp8, bl_[ks]), acc[t], 0, 0, 0);      // (zero fragments when the operand has no lo half)
#endif
        }
    }
    igemm_epilogue<MT>(p, acc, m0, n0, wm * (MT * 32), wn * 32, lane, b, 0, 0, M, oh0, ow0);
}

// ---- column-walking variant of the single-shot kernel (batched synthesis).  The DiT FinalLayer + unpatchify GEMM has K = 256 and
// N = stride^2 * C = 2048: as 64 x 64 single-shot tiles the LayerNorm + modulate staging of a 64-row A tile is redone by each of
// the 32 column-tile workgroups, and the launch spends its time there (245 us at B=32 for 22 GFLOP and a 170 MB output whose
// HBM floor is ~40 us).  Here a workgroup stages its A rows ONCE and walks all the column tiles; the next weight tile is
// prefetched into registers under the MFMAs and the epilogue of the current one (single LDS tile: two workgroups per CU).
// Its epilogue is the unpatchify scatter only (bias, output mask, crop): everything that depends on the token row is computed
// once for the whole walk.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// BM rows per workgroup, BM / 16 waves (a wave = 32 rows x 32 of the tile's 64 columns).  BM = 128 (round 6): every 64-column weight
// tile a workgroup stages serves 128 rows instead of 64 (650 -> 360 MB of L2 -> LDS weight traffic per launch at DEX B = 32) - measured
// no faster, so it is opt-in (launch_nwalk).
// DMA (round 6; 64-row workgroups, not the split-weight build): the weight tiles come from the FRAGMENT-ordered twin (IGemmP::Wfrag: a
// 64-column tile = 32 contiguous 1-KB pieces, piece (wn, ks) = the A fragment of columns wn * 32 .. + 32, K-step ks) through a two-slot
// LDS-DMA ring (global_load_lds_dwordx4, eight pieces per wave and tile): no staging registers, no ds_write pass, ONE barrier per tile
// instead of two, unpadded conflict-free fragment reads.  The A tile (LayerNorm staging, dead once every wave holds its rows as
// fragments) aliases slot 1, so tile 0 lands under the staging.  A wave waits for its own pieces with vmcnt(number of scatter stores
// it issued behind them) - loads and stores leave the counter in issue order on this target - so the stores of a tile stay in flight.
template <int K, int BM = 64, bool DMA = false>
__global__ __launch_bounds__(BM * 4) void igemm_lp_nwalk_kernel(const IGemmP p) {
    constexpr int BN = 64, NTHR = BM * 4;
    constexpr int WN = BN / 32, WM = BM / 32, MT = 1;
    constexpr int LDS_LD = K + 8, KC = K / 8;
    constexpr int AIT = BM * KC / NTHR, BIT = BN * KC / NTHR;
    constexpr int ABATCH = AIT > 8 ? 8 : AIT;
    constexpr int TILE = BN * K;                      // u16 per fragment-ordered weight tile (32 KB)
    static_assert(!DMA || BM == 64, "DMA form: 64-row workgroups");
    extern __shared__ __attribute__((aligned(16))) u16 smem_ss[];
    u16* As = DMA ? smem_ss + TILE : smem_ss;         // DMA: [slot 0 | slot 1 = A tile (+ its padding past the slot)]
    u16* Bs = smem_ss + BM * LDS_LD;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    const int M = p.Ho * p.Wo;
    const int m0 = blockIdx.x * BM;
    const int ntile_all = p.N / BN;
    const int ntile = ntile_all / (int)gridDim.y, nt_first = blockIdx.y * ntile;      // this workgroup's share of the column tiles
    const float* Ab = p.A + (long)b * p.a_bstride + p.a_coff;
    const float* mrow = p.inmask ? p.inmask + (long)b * p.mask_bstride : nullptr;
    const u16* Wb = reinterpret_cast<const u16*>(p.Wbf) + (long)b * p.w_bstride + (long)nt_first * BN * K;
    const int step = p.step;
    const float* lsh = p.ln_shift ? p.ln_shift + (long)step * p.ln_step_stride : nullptr;
    const float* lsc = p.ln_scale ? p.ln_scale + (long)step * p.ln_step_stride : nullptr;

    const u16* Wf = DMA ? reinterpret_cast<const u16*>(p.Wfrag) + (long)b * p.w_bstride + (long)nt_first * TILE : nullptr;
#define NW_DMA(nt_)                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
        const int piece = 8 * wave + j;                                                                             \
        __builtin_amdgcn_global_load_lds(Wf + (long)(nt_) * TILE + piece * 512 + lane * 8,                          \
                                         (lds_ptr_t)(smem_ss + ((nt_) & 1) * TILE + piece * 512), 16, 0, 0);        \
    }
    // DMA: bias (N floats) and this workgroup's output-mask values (64 token rows x unpatch_s column offsets) go to LDS once - with
    // LDS-DMA pieces in flight hipcc turns the wait of ANY ordinary global load into vmcnt(0), so the walk must not issue one
    float* sbias = reinterpret_cast<float*>(smem_ss + TILE + BM * LDS_LD);        // [N]
    float* smk = sbias + p.N;                                                      // [64 rows][8]
    if constexpr (DMA) {
        NW_DMA(0)
        const float* biasg = p.bias ? p.bias + (long)b * p.bias_bstride : nullptr;
        for (int idx = tid; idx < p.N; idx += NTHR) sbias[idx] = biasg ? biasg[idx] : 0.f;
        const float* omg = p.outmask ? p.outmask + (long)b * p.mask_bstride : nullptr;
        const int row = tid >> 2, mrow_ = m0 + row, mm_ = mrow_ < M ? mrow_ : 0, uw_ = (mm_ % p.Wo) * p.unpatch_s;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int sft = (tid & 3) + 4 * e;
            smk[row * 8 + sft] = omg ? omg[min(uw_ + sft, p.OWf - 1) * p.outmask_ws] : 1.f;
        }
    }
    u32x4 br[BIT];                          // (native vectors + macros, not lambdas over an array: those ended up in scratch)
#define NW_LOAD_B(nt_)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < BIT; ++j) {                                                              \
        const int it = tid + NTHR * j;                                                                              \
        br[j] = *reinterpret_cast<const u32x4*>(Wb + (long)((nt_) * BN + it / KC) * K + (it % KC) * 8);            \
    }
#define NW_STORE_B()                                                                                                \
    _Pragma("unroll") for (int j = 0; j < BIT; ++j) {                                                              \
        const int it = tid + NTHR * j;                                                                              \
        *reinterpret_cast<u32x4*>(Bs + (it / KC) * LDS_LD + (it % KC) * 8) = br[j];                                 \
    }
#ifdef DEX_LP_WSPLIT
    // split weights (round 6): the lo tile travels like the hi tile - coalesced 16-byte loads one column tile ahead, then LDS - into the
    // bytes of the A tile, which is dead once every wave holds its token rows as fragments (afr below).  Before, the lo fragments of a
    // wave's 32 columns were loaded straight into B-operand registers: 16 bytes per lane from 32 weight rows per request, the pattern
    // that costs the CU's address path 4x a contiguous one, and 128 registers of ring - 124.5 us against the plain build's 52.6 at DEX B = 32.
    const bool has_lo = p.w_lo_off != 0;
    u16* Bl = As;
    u32x4 brl[BIT];
#define NW_LOAD_LO(nt_)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < BIT; ++j) {                                                              \
        const int it = tid + NTHR * j;                                                                              \
        const u32x4 v_ = *reinterpret_cast<const u32x4*>(Wb + (has_lo ? p.w_lo_off : 0) + (long)((nt_) * BN + it / KC) * K + (it % KC) * 8); \
        brl[j] = has_lo ? v_ : u32x4{0u, 0u, 0u, 0u};                                                               \
    }
#define NW_STORE_LO()                                                                                               \
    _Pragma("unroll") for (int j = 0; j < BIT; ++j) {                                                              \
        const int it = tid + NTHR * j;                                                                              \
        *reinterpret_cast<u32x4*>(Bl + (it / KC) * LDS_LD + (it % KC) * 8) = brl[j];                                \
    }
    NW_LOAD_LO(0)
#endif
    if constexpr (!DMA) { NW_LOAD_B(0) }
#pragma unroll
    for (int a0 = 0; a0 < AIT; a0 += ABATCH) {
        float4 f0[ABATCH], f1[ABATCH];
        float mk[ABATCH];
#pragma unroll
        for (int j = 0; j < ABATCH; ++j) {
            const int it = tid + NTHR * (a0 + j);
            const int row = it / KC, k8 = (it % KC) * 8;
            const int m = m0 + row;
            const int mm = m < M ? m : 0;                    // (1 x 1 taps only: row m of the A matrix)
            const float* src = Ab + (long)mm * p.lda + k8;
            f0[j] = *reinterpret_cast<const float4*>(src);
            f1[j] = *reinterpret_cast<const float4*>(src + 4);
            const float mv = mrow ? mrow[(mm % p.Wi) * p.inmask_ws] : 1.f;
            mk[j] = m < M ? mv : 0.f;
        }
        if constexpr (!DMA) { if (a0 == 0) { NW_STORE_B() } }
#pragma unroll
        for (int j = 0; j < ABATCH; ++j) {
            const int it = tid + NTHR * (a0 + j);
            const int row = it / KC, k8 = (it % KC) * 8;
            float4 a = f0[j], c = f1[j];
            if (lsh) {
                float s = (a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w);
#pragma unroll
                for (int o = 1; o < KC; o <<= 1) s += __shfl_xor(s, o);
                const float mean = s * (1.f / (float)K);
                a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean; c.x -= mean; c.y -= mean; c.z -= mean; c.w -= mean;
                float q = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
#pragma unroll
                for (int o = 1; o < KC; o <<= 1) q += __shfl_xor(q, o);
                const float rstd = rsqrtf(q * (1.f / (float)K) + 1e-6f);
                const float4 sc0 = *reinterpret_cast<const float4*>(lsc + k8), sc1 = *reinterpret_cast<const float4*>(lsc + k8 + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(lsh + k8), sh1 = *reinterpret_cast<const float4*>(lsh + k8 + 4);
                a.x = a.x * rstd * (1.f + sc0.x) + sh0.x; a.y = a.y * rstd * (1.f + sc0.y) + sh0.y;
                a.z = a.z * rstd * (1.f + sc0.z) + sh0.z; a.w = a.w * rstd * (1.f + sc0.w) + sh0.w;
                c.x = c.x * rstd * (1.f + sc1.x) + sh1.x; c.y = c.y * rstd * (1.f + sc1.y) + sh1.y;
                c.z = c.z * rstd * (1.f + sc1.z) + sh1.z; c.w = c.w * rstd * (1.f + sc1.w) + sh1.w;
            }
            const float m_ = mk[j];
            uint4 v;
            v.x = pack2_lp(a.x * m_, a.y * m_); v.y = pack2_lp(a.z * m_, a.w * m_);
            v.z = pack2_lp(c.x * m_, c.y * m_); v.w = pack2_lp(c.z * m_, c.w * m_);
            *reinterpret_cast<uint4*>(As + row * LDS_LD + k8) = v;
        }
    }
    __syncthreads();
    const u16* ap = As + (wm * (MT * 32) + i) * LDS_LD + hh * 8;
    const u16* bp = Bs + (wn * 32 + i) * LDS_LD + hh * 8;            // (DMA: the ring slot's piece (wn, 0), set per tile)
    // unpatchify scatter without activation / gate / residual (the FinalLayer): everything that depends on the ROW only - token
    // (f, w), its pixel base, validity - is computed once for the whole column walk (the shared epilogue redoes two divisions
    // and a 64-bit address per element and column tile: the launch was bound by that integer work and by 16 dependent mask
    // loads per tile, not by its 170 MB of output)
    static_assert(MT == 1, "one 32-row tile per wave");
    // Round 5: the product is computed TRANSPOSED (weight fragment = A operand): a lane owns ONE token row (lane & 31) and 16 of the
    // tile's 32 channels, four runs of four (8 q + 4 hh ..).  The scatter then writes 16 bytes per lane - two stores of a tile in the
    // 16-bit form (one v_permlane32_swap per dword pairs the two halves of a token into 8-channel chunks), four in fp32 - where the
    // plain product wrote one 2- / 4-byte element per lane and register: 16 store instructions of 128 / 256 bytes per tile, and the
    // launch was bound by their number (5.4k store instructions per CU at DEX B = 32: 66 us for a 9 us GEMM with a 20 us output).
    // Same products in the same K order, same (acc + bias) * mask: the values are bit-identical, only their placement changes.
    // the wave's 32 token rows stay in registers for the whole walk (16 fragments: they were re-read from LDS for every column tile -
    // two LDS fragment reads per MFMA)
    lp8 afr[K / 16];
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) afr[ks] = *reinterpret_cast<const lp8*>(ap + ks * 16);
#ifdef DEX_LP_WSPLIT
    __syncthreads();                                      // every wave holds its rows: the A tile's bytes take the lo weight tiles from here on
    NW_STORE_LO()
    __syncthreads();
    const u16* blp = Bl + (wn * 32 + i) * LDS_LD + hh * 8;
#endif
    int u_pix, u_f, u_w;
    {
        const int m = m0 + wm * 32 + i;
        const bool v = m < M;
        const int mm = v ? m : 0;
        const int f = mm / p.Wo, w = mm - f * p.Wo;
        u_f = v ? f * p.unpatch_s : 0x40000000;                    // an invalid row fails the height test below
        u_w = w * p.unpatch_s;
        u_pix = f * p.unpatch_s * p.OWf + w * p.unpatch_s;
    }
    const float* omask = p.outmask ? p.outmask + (long)b * p.mask_bstride : nullptr;
    const float* biasb = p.bias ? p.bias + (long)b * p.bias_bstride : nullptr;
    // DMA: the scatter of a tile is DEFERRED to the head of the next iteration, behind the next DMA group and in front of the MFMA chain:
    // waiting for a wave's pieces is vmcnt(0) (hipcc drains the counter in front of any LDS read that may alias a piece in flight), and a
    // tile's stores issued at its end would be waited for a few instructions later; issued here they have a whole MFMA chain to drain
    float4 dst4[4]; char* d_ptr = nullptr; bool d_ok = false, d_all = false, d_have = false;
#define NW_FLUSH()                                                                                                  \
    if (d_have) {                                                                                                   \
        if (p.c_lp) {                                                                                               \
            u16* c_ = reinterpret_cast<u16*>(d_ptr);                                                                \
            if (d_all) { *reinterpret_cast<float4*>(c_) = dst4[0]; *reinterpret_cast<float4*>(c_ + 16) = dst4[1]; } \
            else if (d_ok) { *reinterpret_cast<float4*>(c_) = dst4[0]; *reinterpret_cast<float4*>(c_ + 16) = dst4[1]; } \
        } else {                                                                                                    \
            float* c_ = reinterpret_cast<float*>(d_ptr);                                                            \
            if (d_all) { _Pragma("unroll") for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(c_ + 8 * q) = dst4[q]; } \
            else if (d_ok) { _Pragma("unroll") for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(c_ + 8 * q) = dst4[q]; } \
        }                                                                                                           \
    }
    for (int nt = 0; nt < ntile; ++nt) {
        if constexpr (DMA) {
            __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): this wave's pieces of tile nt have landed ...
            __syncthreads();                          // ... everybody's: tile nt visible, slot (nt + 1) & 1 (tile nt - 1 / the A tile) free
            bp = smem_ss + (nt & 1) * TILE + (wn * 16 * 64 + lane) * 8;
        } else {
            if (nt + 1 < ntile) { NW_LOAD_B(nt + 1) }
        }
#ifdef DEX_LP_WSPLIT
        if (nt + 1 < ntile) { NW_LOAD_LO(nt + 1) }
#endif
        const int ng0 = (nt_first + nt) * BN + wn * 32, pp = ng0 / p.unpatch_C;
        const int u_c0 = ng0 - pp * p.unpatch_C;                // first channel of this wave's 32 (a multiple of 32)
        const int u_p1 = pp / p.unpatch_s, u_p2 = pp - u_p1 * p.unpatch_s;
        float u_mk;
        float4 b4[4];
        if constexpr (DMA) {
            u_mk = smk[(wm * 32 + i) * 8 + u_p2];
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const float4*>(sbias + ng0 + 8 * q + 4 * hh);
            if (nt + 1 < ntile) { NW_DMA(nt + 1) }
            NW_FLUSH()
        } else {
            u_mk = omask ? omask[min(u_w + u_p2, p.OWf - 1) * p.outmask_ws] : 1.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = biasb ? *reinterpret_cast<const float4*>(biasb + ng0 + 8 * q + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            const lp8 bf = *reinterpret_cast<const lp8*>(bp + ks * (DMA ? 512 : 16));
            acc[0] = DEX_MFMA_LP(bf, afr[ks], acc[0], 0, 0, 0);
#ifdef DEX_LP_WSPLIT
            acc[0] = DEX_MFMA_LP(*reinterpret_cast<const lp8*>(blp + ks * 16), afr[ks], acc[0], 0, 0, 0);
#endif
        }
        {
            // values first (the mask / bias loads are consumed here, once), then the stores: a store under a per-lane branch that still
            // depends on a load gets an s_waitcnt vmcnt(0) of its own, which also drains every earlier STORE
            float val[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                val[4 * q + 0] = (acc[0][4 * q + 0] + b4[q].x) * u_mk; val[4 * q + 1] = (acc[0][4 * q + 1] + b4[q].y) * u_mk;
                val[4 * q + 2] = (acc[0][4 * q + 2] + b4[q].z) * u_mk; val[4 * q + 3] = (acc[0][4 * q + 3] + b4[q].w) * u_mk;
            }
            const bool ok = u_f + u_p1 < p.OHf && u_w + u_p2 < p.OWf;
            const long e0 = (long)b * p.c_bstride + p.c_coff + (long)(u_p1 * p.OWf + u_p2 + u_pix) * p.ldc + u_c0;     // element offset of (token, channel u_c0)
            __builtin_amdgcn_sched_barrier(0);
            if (p.c_lp) {             // (uniform) C in the mode's 16-bit type: its reader rounds it so anyway (the up path's 3x3 conv)
                typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
                unsigned D[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) { D[q][0] = pack2_lp(val[4 * q], val[4 * q + 1]); D[q][1] = pack2_lp(val[4 * q + 2], val[4 * q + 3]); }
                uint4 ch[2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const u32x2n s0 = __builtin_amdgcn_permlane32_swap(D[2 * pr][0], D[2 * pr + 1][0], false, false);
                    const u32x2n s1 = __builtin_amdgcn_permlane32_swap(D[2 * pr][1], D[2 * pr + 1][1], false, false);
                    ch[pr] = make_uint4(s0[0], s1[0], s0[1], s1[1]);                 // channels u_c0 + 16 pr + 8 hh .. + 7 of this lane's token
                }
                // (passing the workgroup's 64 x 64 outputs through an LDS tile so that a request writes 8 tokens x 128 contiguous bytes was
                // measured too: 55.4 vs 54.3 us at DEX B = 32, 77.4 vs 74.1 at GeDEX B = 32 - the stores no longer bound the launch)
                u16* cph = reinterpret_cast<u16*>(p.C) + e0 + 8 * hh;
                if constexpr (DMA) {
                    dst4[0] = __builtin_bit_cast(float4, ch[0]); dst4[1] = __builtin_bit_cast(float4, ch[1]);
                    d_ptr = reinterpret_cast<char*>(cph); d_ok = ok; d_all = __builtin_amdgcn_ballot_w64(!ok) == 0; d_have = true;
                } else
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) {                        // the common case: nothing of this tile is cropped
                    *reinterpret_cast<uint4*>(cph) = ch[0];
                    *reinterpret_cast<uint4*>(cph + 16) = ch[1];
                } else if (ok) {
                    *reinterpret_cast<uint4*>(cph) = ch[0];
                    *reinterpret_cast<uint4*>(cph + 16) = ch[1];
                }
            } else {
                float* cp = p.C + e0 + 4 * hh;
                if constexpr (DMA) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst4[q] = make_float4(val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]);
                    d_ptr = reinterpret_cast<char*>(cp); d_ok = ok; d_all = __builtin_amdgcn_ballot_w64(!ok) == 0; d_have = true;
                } else
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(cp + 8 * q) = make_float4(val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]);
                } else if (ok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(cp + 8 * q) = make_float4(val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]);
                }
            }
        }
        if (!DMA && nt + 1 < ntile) {
            __syncthreads();                              // every wave is done with this weight tile
            NW_STORE_B()
#ifdef DEX_LP_WSPLIT
            NW_STORE_LO()
#endif
            __syncthreads();
        }
    }
    if constexpr (DMA) { NW_FLUSH() }
}

static bool nwalk_eligible(const IGemmP& p) {
    if (p.a_lp || p.act_in_slope != 0.f) return false;         // (c_lp: implemented in the column walker's scatter)
    if (p.K != 256 || p.Cin != 256 || p.KH != 1 || p.KW != 1 || p.parity || p.ksplit != 1 || p.groups != 1 || (p.N % 64) != 0) return false;
    if (p.sh != 1 || p.sw != 1 || p.off_h != 0 || p.off_w != 0 || p.Ho != p.Hi || p.Wo != p.Wi || p.gn_stats) return false;
    // the kernel's epilogue is the unpatchify scatter and nothing else: bias, output mask, crop
    if (p.unpatch_s <= 0 || (p.unpatch_C % 32) != 0 || p.gate || p.res || p.act != 0 || p.stats_final) return false;
    if ((p.ldc % 8) != 0 || (p.c_coff % 8) != 0 || (p.c_bstride % 8) != 0) return false;          // 16-byte scatter stores (8 x 16 bit / 4 x fp32 per lane)
    const int mode = knob_or("DEX_GEMM_NWALK", 1);       // 0: never, 2: whenever the shape allows (tests)
    if (mode == 0) return false;
    const long wgs = (long)((p.Ho * p.Wo + 63) / 64) * p.B;
    // small grids too (round 4): with one column tile per workgroup (nsplit = N / 64 below) the walker is the single-shot kernel with
    // the cheap unpatchify scatter - row-only terms once instead of two divisions and a 64-bit address per element: 14.3 -> 12 us at
    // B = 1, +0.4 % end to end
    // (round 5: the grids in between as well - the long form's 79 row tiles ran the single-shot kernel at 73 us, the walker takes 31)
    (void)wgs;
    return p.N / 64 >= 8;
}
bool igemm_nwalk_form(const IGemmP& p) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_NWALK)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return nwalk_eligible(p); }
static void launch_nwalk(const IGemmP& p, hipStream_t st) {
    constexpr int K = 256;
    const size_t lds = (size_t)(64 + 64) * (K + 8) * sizeof(u16), lds128 = (size_t)(128 + 64) * (K + 8) * sizeof(u16);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lp_nwalk_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lp_nwalk_kernel<K, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
#ifndef DEX_LP_WSPLIT
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lp_nwalk_kernel<K, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
#endif
        attr = true;
    }
    // 128-row workgroups (eight waves, one workgroup per CU): half the weight stream per row.  Built, bit-identical, and measured NO faster
    // (profiles/round6_unpatchify_gemm_128_row_workgroups_negative.txt: 53.8 vs 56.4 us at DEX B = 32, 73.3 vs 74.7 at GeDEX B = 32) - the
    // L2 -> LDS weight stream the round-5 notes blamed is not what bounds the launch; what is left is the scatter itself (256- / 512-byte
    // runs at 2.3 - 3.1 TB/s of writes).  OPT-IN: DEX_NWALK_BM=128 (tests keep the form alive).
    {
        const long wgs128 = (long)((p.Ho * p.Wo + 127) / 128) * p.B;
        const int bm_env = knob_or("DEX_NWALK_BM", 0);
        int ns = 1;
        while (ns < 4 && wgs128 * ns < 640 && (p.N / 64) % (ns * 2) == 0) ns *= 2;
        const bool big = bm_env == 128;
        if (big) {
            const int fs = knob_or("DEX_NWALK_SPLIT", 0);
            if (fs > 0 && (p.N / 64) % fs == 0) ns = fs;
            g_last_symbol = "igemm_lp_nwalk_kernel<256, 128>";
            hipLaunchKernelGGL((igemm_lp_nwalk_kernel<K, 128>), dim3((p.Ho * p.Wo + 127) / 128, ns, p.B), dim3(512), lds128, st, p);
            return;
        }
    }
#ifndef DEX_LP_WSPLIT
    if (p.Wfrag && p.w_bstride == 0 && p.N <= 2048 && p.unpatch_s <= 8 && knob_or("DEX_NWALK_DMA", 0) != 0) {       // the LDS-DMA form (fragment-ordered twin of the weight): OPT-IN - built, bit-identical, measured no faster (see below)
        const size_t lds_dma = (size_t)64 * K * 2 + (size_t)64 * (K + 8) * 2 + (size_t)p.N * 4 + 64 * 8 * 4;       // ring slot 0 | slot 1 = A tile | bias | mask table
        const long wgs = (long)((p.Ho * p.Wo + 63) / 64) * p.B;
        int nsplit = 1;
        const int fs = knob_or("DEX_NWALK_SPLIT", 0);
        while (nsplit < 4 && wgs * nsplit < 1280 && (p.N / 64) % (nsplit * 2) == 0) nsplit *= 2;
        if (wgs <= 16) nsplit = p.N / 64;
        if (fs > 0 && (p.N / 64) % fs == 0) nsplit = fs;
        g_last_symbol = "igemm_lp_nwalk_kernel<256,64,1>";
        hipLaunchKernelGGL((igemm_lp_nwalk_kernel<K, 64, true>), dim3((p.Ho * p.Wo + 63) / 64, nsplit, p.B), dim3(256), lds_dma, st, p);
        return;
    }
#endif
    g_last_symbol = "igemm_lp_nwalk_kernel<256>";
    // split the walk over 1 / 2 / 4 workgroups so that the grid is at least ~2.5 rounds of the chip's 512 slots: a workgroup's tiles
    // run back to back behind each other's store drain, more of them in flight hide it (measured at B=32: 115 -> ? us)
    const long wgs = (long)((p.Ho * p.Wo + 63) / 64) * p.B;
    int nsplit = 1;
    const int fs = knob_or("DEX_NWALK_SPLIT", 0);
    while (nsplit < 4 && wgs * nsplit < 1280 && (p.N / 64) % (nsplit * 2) == 0) nsplit *= 2;
    if (wgs <= 16) nsplit = p.N / 64;              // small grid: one column tile per workgroup
    if (fs > 0 && (p.N / 64) % fs == 0) nsplit = fs;
    dim3 grid((p.Ho * p.Wo + 63) / 64, nsplit, p.B);
    hipLaunchKernelGGL((igemm_lp_nwalk_kernel<K>), grid, dim3(256), lds, st, p);
}

template <int K>
static void launch_ss(const IGemmP& p, hipStream_t st) {
    const size_t lds = (size_t)(64 + 64) * (K + 8) * sizeof(u16);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lp_ss_kernel<64, 64, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lp_ss_kernel<64, 64, K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid((p.Ho * p.Wo + 63) / 64, p.N / 64, p.B * (p.parity ? 4 : 1));
    if (p.a_lp) hipLaunchKernelGGL((igemm_lp_ss_kernel<64, 64, K, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((igemm_lp_ss_kernel<64, 64, K>), grid, dim3(256), lds, st, p);
}

static bool ss_eligible(const IGemmP& p) {
    if (p.c_lp || p.act_in_slope != 0.f) return false;
    if (p.a_lp && (p.ln_shift || p.KH != 1 || p.KW != 1 || (p.Cin % 8) != 0)) return false;      // 16-bit A: plain 1x1 rows only
    if (p.ksplit != 1 || p.groups != 1 || (p.N % 64) != 0) return false;
    if (p.K != 64 && p.K != 128 && p.K != 256 && p.K != 512) return false;
    const long blocks = (long)((p.Ho * p.Wo + 63) / 64) * (p.N / 64) * p.B;
    // (K <= 128 - the DEX TV adaptor's 1x1 convs at batch size: one round trip instead of a two-tile loop, 124 -> 114 us for the pair)
    return blocks <= (p.K <= 128 ? 16384 : 4096) || p.ln_shift != nullptr;
}

bool igemm_lp_io_supported(int Cin, int K, int N, int ksplit) { return Cin % 64 == 0 && (K / (ksplit > 0 ? ksplit : 1)) % 64 == 0 && N % 64 == 0; }

void launch_igemm_lp(const IGemmP& p, hipStream_t st) {
    const int M = p.Ho * p.Wo;
    if (DEX_LP_NS::igemm_nwalk_form(p)) { launch_nwalk(p, st); return; }
    if (ss_eligible(p)) {
        if (p.K == 64) launch_ss<64>(p, st);
        else if (p.K == 128) launch_ss<128>(p, st);
        else if (p.K == 256) launch_ss<256>(p, st);
        else launch_ss<512>(p, st);
        return;
    }
    const int zdim = p.B * p.groups * p.ksplit * (p.parity ? 4 : 1);
    const bool k64 = (p.Cin % 64 == 0) && ((p.K / p.ksplit) % 64 == 0);
    if (p.act_in_slope != 0.f) {       // vocoder convolutions: leaky_relu while the A tile is staged (looped kernel, three tilings)
        if (p.N % 64 == 0) {
            const long blocks128 = (long)((M + 127) / 128) * (p.N / 64) * zdim;
            if (blocks128 < 1024) {
                dim3 grid((M + 63) / 64, p.N / 64, zdim);
                if (k64) hipLaunchKernelGGL((igemm_lp_kernel<64, 64, 64, false, 1, 0, false, false, true>), grid, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((igemm_lp_kernel<64, 64, 32, false, 1, 0, false, false, true>), grid, dim3(256), 0, st, p);
            } else {
                dim3 grid((M + 127) / 128, p.N / 64, zdim);
                if (k64) hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 64, false, 1, 0, false, false, true>), grid, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 32, false, 1, 0, false, false, true>), grid, dim3(256), 0, st, p);
            }
        } else {
            dim3 grid((M + 127) / 128, p.N / 32, zdim);
            hipLaunchKernelGGL((igemm_lp_kernel<128, 32, 32, false, 1, 0, false, false, true>), grid, dim3(256), 0, st, p);
        }
        return;
    }
    if (p.a_lp || p.c_lp) {          // 16-bit A / C tensors: the batch tile only (igemm_lp_io_supported)
        dim3 grid((M + 127) / 128, p.N / 64, zdim);
        if (p.a_lp && p.c_lp) hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 64, false, 1, 0, true, true>), grid, dim3(256), 0, st, p);
        else if (p.a_lp) hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 64, false, 1, 0, true, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 64, false, 1, 0, false, true>), grid, dim3(256), 0, st, p);
        return;
    }
    if (p.N % 64 == 0) {
        const long blocks128 = (long)((M + 127) / 128) * (p.N / 64) * zdim;
        if (blocks128 < 1024) {
            dim3 grid((M + 63) / 64, p.N / 64, zdim);
            if (k64 && p.K / p.ksplit == 576) hipLaunchKernelGGL((igemm_lp_kernel<64, 64, 64, false, 3, 9>), grid, dim3(256), 0, st, p);   // 3x3 x 64ch
            else if (k64) hipLaunchKernelGGL((igemm_lp_kernel<64, 64, 64>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_lp_kernel<64, 64, 32>), grid, dim3(256), 0, st, p);
        } else {
            dim3 grid((M + 127) / 128, p.N / 64, zdim);
            if (k64) hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 64>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_lp_kernel<128, 64, 32>), grid, dim3(256), 0, st, p);
        }
    } else {
        dim3 grid((M + 127) / 128, p.N / 32, zdim);
        if (p.K / p.ksplit == 1024 && p.Cin % 8 == 0) hipLaunchKernelGGL((igemm_lp_kernel<128, 32, 128, true, 2, 8>), grid, dim3(256), 0, st, p);   // pos-conv split
        else if ((p.K / p.ksplit) % 128 == 0 && p.Cin % 8 == 0) hipLaunchKernelGGL((igemm_lp_kernel<128, 32, 128, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_lp_kernel<128, 32, 32>), grid, dim3(256), 0, st, p);
    }
}

}  // namespace DEX_LP_NS
}  // namespace dex
