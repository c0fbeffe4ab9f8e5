// pos_conv.hip — the DiT positional convolution (reference model/dit.py pos_conv: Conv2d(hid, hid, 16, padding 8,
// groups 8) + SamePad) as a patch-staged direct convolution on v_mfma_f32_32x32x16_bf16 (bf16-MFMA mode).
//
// As an implicit GEMM this layer gathers every input element 256 times (once per tap): 170 MB of L2 reads per
// launch at B=1 and 5.4 GB at B=32 — it was L2-bandwidth-bound (23 us / 409 us).  Here a workgroup owns one output
// row ho of one (batch, group) and a chunk of 32*CT columns: the (<=16 rows) x (chunk + 15 columns) x 32-channel
// input patch is staged ONCE in LDS as bf16 and every tap reads it with a shifted pointer; rows outside the token
// grid are simply not part of the tap list (for a 10-row grid 96 of the 160 (row, kh) pairs exist).
// The four waves split the tap list (tap t -> wave t % 4) so every weight fragment is fetched by exactly one wave,
// straight from L2 into MFMA B-operand registers through a 4-deep register ring (weights are host-packed in
// fragment order, 1 KB contiguous per wave per K-step); the waves' accumulators are summed through LDS at the end.
// Output: raw convolution sums [B][N][hid]; bias + GELU + the mean over frequency stay in pos_finish (dit_elem.hip).
//
// Column workgroups (p.ncol > 0).  The token grids of the shipped configs are 65 columns wide (T/patch + 1): as 32-column
// tiles every row needs a third tile for ONE column, and as row chunks a whole extra workgroup that streams the row's 400 KB
// of weights for it.  With ncol = 1 the row workgroups cover the first Wt - 1 columns and one extra workgroup per (batch,
// group) computes the LAST column of all rows as a single tile whose 32 rows are the token rows ho (lane stride = one patch
// row of 17 positions: the chunk swizzle below stays conflict-free); its taps are the 16 x 9 with a column inside the grid.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

namespace {
constexpr int KP = 16, PAD = 8, CG = 32, LDP = CG;         // LDS pixel = 64 B, no padding: the four 16-B channel chunks of
                                                            // pixel px sit at slot c ^ ((px >> 2) & 3) — 16 consecutive pixels
                                                            // then cover all 64 banks for any fixed chunk (conflict-free b128)
}

template <int CT, bool COL>
__device__ __forceinline__ void pos_conv_body(const PosConvP& p, int rest) {
    constexpr int CW = 32 * CT, PW = CW + KP - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pc[];
    u16* patch = reinterpret_cast<u16*>(smem_pc);          // [nrows][PW][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    // XCD-aware decomposition: workgroup id % 8 selects the XCD (round-robin dispatch) and each XCD has a private L2, so
    // the group index lives in the low bits — an XCD then only ever fetches ONE group's 512 KB of weights (with the
    // group in grid.z every XCD pulled all 4 MB: PMC FETCH_SIZE showed 38 MB per launch at B=1).
    const int Hf = p.Hf, Wt = p.Wt;
    const int Wrow = Wt - p.ncol;                                           // columns covered by the row workgroups
    const int nchunk = (Wrow + CW - 1) / CW;
    const int g = blockIdx.x % p.G;
    constexpr bool col = COL;                                               // a column workgroup (see the header)
    constexpr int CPW = KP + 1, CKW = PAD + 1;                              // its patch row stride (positions) and live kw count
    int w0, ho, b;
    if (col) { b = rest - nchunk * Hf * p.B; ho = 0; w0 = Wt - 1; }
    else { w0 = (rest % nchunk) * CW; rest /= nchunk; ho = rest % Hf; b = rest / Hf; }
    const int hi_lo = col ? -PAD : max(0, ho - PAD), hi_hi = col ? Hf + PAD - 1 : min(Hf, ho + PAD);   // input rows [hi_lo, hi_hi)
    const int nrows = hi_hi - hi_lo;
    const int kh_lo = col ? 0 : hi_lo - ho + PAD;                           // kh of input row hi: hi - ho + PAD
    const int ntaps = col ? KP * CKW : nrows * KP;                          // multiple of 16: every wave gets ntaps/4
    constexpr int tdiv = col ? CKW : KP;                                    // tap t = (kh_lo + t / tdiv, t % tdiv)
    constexpr int pw = col ? CPW : PW;                                      // patch row stride

    // ---- weight ring: taps wave, wave+4, wave+8, wave+12 in flight before the patch is staged
    const u32x4* Wf = reinterpret_cast<const u32x4*>(p.Wf) + (long)g * (KP * KP * 2 * 64) + lane;
    u32x4 wr[4][2];
#ifdef DEX_LP_WSPLIT
    u32x4 wl[4][2];                                         // split weights: the lo halves ride the same ring, one whole pack (G groups) behind
    const long lo_u4 = (long)p.G * (KP * KP * 2 * 64);
#define PC_LOAD_LO(d, tap) do { wl[d][0] = Wf[lo_u4 + ((tap) * 2 + 0) * 64]; wl[d][1] = Wf[lo_u4 + ((tap) * 2 + 1) * 64]; } while (0)
#else
#define PC_LOAD_LO(d, tap) do { } while (0)
#endif
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int t = wave + 4 * d;                         // < 16 <= ntaps
        const int tap = (kh_lo + t / tdiv) * KP + (t % tdiv);
        wr[d][0] = Wf[(tap * 2 + 0) * 64];
        wr[d][1] = Wf[(tap * 2 + 1) * 64];
        PC_LOAD_LO(d, tap);
    }
    // ---- stage the patch: item = (row, patch column, 8-channel chunk); zero outside the token grid
    {
        const float* X = p.X + ((long)b * Hf * Wt) * p.hid + g * CG;
        const int items = nrows * pw * 4;
        for (int it0 = 0; it0 < items; it0 += 256 * 9) {
            float4 f0[9], f1[9];
            bool ok[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int it = min(it0 + tid + 256 * q, items - 1);
                const int c8 = (it & 3) * 8, px = it >> 2;
                const int pc = px % pw, pr = px / pw;
                const int wi = w0 + pc - PAD, hi = hi_lo + pr;
                ok[q] = (unsigned)wi < (unsigned)Wt && (unsigned)hi < (unsigned)Hf;
                const float* src = X + ((long)(ok[q] ? hi : 0) * Wt + (ok[q] ? wi : 0)) * p.hid + c8;
                f0[q] = *reinterpret_cast<const float4*>(src);
                f1[q] = *reinterpret_cast<const float4*>(src + 4);
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int it = it0 + tid + 256 * q;
                if (it < items) {
                    const float m = ok[q] ? 1.f : 0.f;
                    uint4 v;
                    v.x = pack2_lp(f0[q].x * m, f0[q].y * m); v.y = pack2_lp(f0[q].z * m, f0[q].w * m);
                    v.z = pack2_lp(f1[q].x * m, f1[q].y * m); v.w = pack2_lp(f1[q].z * m, f1[q].w * m);
                    const int px = it >> 2;
                    *reinterpret_cast<uint4*>(patch + px * LDP + (((it & 3) ^ ((px >> 2) & 3)) * 8)) = v;
                }
            }
        }
    }
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    lds_barrier();

    // ---- this wave's taps: t = wave + 4*n, n = 0 .. ntaps/4 - 1 (a multiple of 4: the ring slot index is static)
    const int nmine = ntaps / 4;
    for (int n0 = 0; n0 < nmine; n0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int t = wave + 4 * (n0 + d);
            const int pr = t / tdiv, kw = t % tdiv;
            // patch pixel of column tile 0 (tile ct: + 32*ct, same swizzle phase); column workgroup: lane = token row
            const int px = col ? (i + pr) * CPW + kw : pr * PW + i + kw;
            const lp8 b0 = __builtin_bit_cast(lp8, wr[d][0]);
            const lp8 b1 = __builtin_bit_cast(lp8, wr[d][1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct > 0 && col) break;                     // (one tile)
                const int sw = (px >> 2) & 3;                 // (px + 32*ct) >> 2 has the same low two bits
                const u16* ap = patch + (long)(px + 32 * ct) * LDP;
                const lp8 a0 = *reinterpret_cast<const lp8*>(ap + ((hh ^ sw) * 8));
                const lp8 a1 = *reinterpret_cast<const lp8*>(ap + (((2 + hh) ^ sw) * 8));
                acc[ct] = DEX_MFMA_LP(a0, b0, acc[ct], 0, 0, 0);
                acc[ct] = DEX_MFMA_LP(a1, b1, acc[ct], 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                acc[ct] = DEX_MFMA_LP(a0, __builtin_bit_cast(lp8, wl[d][0]), acc[ct], 0, 0, 0);
                acc[ct] = DEX_MFMA_LP(a1, __builtin_bit_cast(lp8, wl[d][1]), acc[ct], 0, 0, 0);
#endif
            }
            const int tn = min(t + 16, ntaps - 4 + wave);   // refill (clamped re-read at the tail, never consumed)
            const int tap = (kh_lo + tn / tdiv) * KP + (tn % tdiv);
            wr[d][0] = Wf[(tap * 2 + 0) * 64];
            wr[d][1] = Wf[(tap * 2 + 1) * 64];
            PC_LOAD_LO(d, tap);
        }
    }
#undef PC_LOAD_LO
    // ---- sum the four waves' accumulators through LDS (the patch is dead), store raw sums
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem_pc);          // [4][CT][32 rows][33]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * CT + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + i] = acc[ct][r];
    lds_barrier();
    float* Y = p.Y + ((long)b * Hf * Wt + (long)ho * Wt) * p.hid + g * CG;
    if (col) {                                              // tile row = token row ho, one column
        for (int idx = tid; idx < 1024; idx += 256) {
            const int n = idx & 31, row = idx >> 5;
            if (row < Hf)
                Y[((long)row * Wt + w0) * p.hid + n] = (red[((0 * CT) * 32 + row) * 33 + n] + red[((1 * CT) * 32 + row) * 33 + n]) +
                                                        (red[((2 * CT) * 32 + row) * 33 + n] + red[((3 * CT) * 32 + row) * 33 + n]);
        }
        return;
    }
    for (int idx = tid; idx < CT * 1024; idx += 256) {
        const int n = idx & 31, row = (idx >> 5) & 31, ct = idx >> 10;
        const int w = w0 + ct * 32 + row;
        if (w < Wrow) {
            const float v = (red[((0 * CT + ct) * 32 + row) * 33 + n] + red[((1 * CT + ct) * 32 + row) * 33 + n]) +
                            (red[((2 * CT + ct) * 32 + row) * 33 + n] + red[((3 * CT + ct) * 32 + row) * 33 + n]);
            Y[(long)w * p.hid + n] = v;
        }
    }
}

template <int CT>
__global__ __launch_bounds__(256) void pos_conv_direct_kernel(const PosConvP p) {
    const int rest = blockIdx.x / p.G;
    const int nrow_wgs = ((p.Wt - p.ncol + 32 * CT - 1) / (32 * CT)) * p.Hf * p.B;
    if (rest >= nrow_wgs) pos_conv_body<CT, true>(p, rest);                // (workgroup-uniform)
    else pos_conv_body<CT, false>(p, rest);
}

bool pos_conv_direct_supported(int hid, int groups, int kernel, int Hf) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_POS)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    return groups > 0 && hid / groups == CG && kernel == KP && Hf >= 1;
}

template <int CT>
static void launch_pc(const PosConvP& p, hipStream_t st) {
    constexpr int PW = 32 * CT + KP - 1;
    const int nrows = p.Hf < 2 * PAD ? p.Hf : 2 * PAD;
    size_t lds = (size_t)nrows * PW * LDP * sizeof(u16);
    const size_t lds_red = (size_t)4 * CT * 32 * 33 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    const size_t lds_col = p.ncol ? (size_t)(32 + 2 * PAD - 1) * (KP + 1) * LDP * sizeof(u16) : 0;     // (lanes past the last token row read, and discard, up to row 31 + 15)
    if (lds < lds_col) lds = lds_col;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pos_conv_direct_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    dim3 grid((unsigned)((((p.Wt - p.ncol + 32 * CT - 1) / (32 * CT)) * p.Hf + (p.ncol ? 1 : 0)) * p.B * p.G));
    hipLaunchKernelGGL(pos_conv_direct_kernel<CT>, grid, dim3(256), lds, st, p);
}

void launch_pos_conv_direct(const PosConvP& p0, hipStream_t st) {
    PosConvP p = p0;
    // one live column in the last 32-column tile (Wt = 65 in every shipped config at the BASELINE shapes) and enough row
    // workgroups that the extra column workgroups are noise: the column form (see the header).  DEX_POS_COL=0 disables it.
    const bool col_off = knob_off("DEX_POS_COL");
    const long col_min = knob_or("DEX_POS_COL_MIN", 512);
    p.ncol = (!col_off && p.Wt > 32 && p.Wt % 32 == 1 && p.Hf <= 32 && (long)p.Hf * p.B * p.G >= col_min) ? 1 : 0;
    // widest chunk (fewest halo columns) whose patch still lets two workgroups share a CU (three when the grid is
    // large), but never so wide that a small batch leaves CUs idle.  Measured (us, CT = 3 / 2 / 1):
    // GeDEX B=32 (10 rows) 94 / 124 / 132;  DEX B=32 (20 rows) 381 / 348 / 321;  DEX B=1 39 / 27 / 28.
    const int tiles = (p.Wt - p.ncol + 31) / 32;
    const long rows = (long)p.Hf * p.B * p.G;
    const int nrows = p.Hf < 2 * PAD ? p.Hf : 2 * PAD;
    auto lds_of = [&](int c) { return (long)nrows * (32 * c + KP - 1) * LDP * 2; };
    auto wgs_of = [&](int c) { return rows * ((tiles + c - 1) / c); };
    int ct = tiles >= 3 ? 3 : tiles;
    while (ct > 1 && lds_of(ct) > 80 * 1024) --ct;
    // (with column workgroups the two remaining tiles of a row stay ONE workgroup - half the weight streams - although its patch
    // then allows two, not three, workgroups per CU: DEX B=32 166 vs 210 us, GeDEX B=32 66 vs 85)
    if (wgs_of(ct) >= 2048 && !(p.ncol && tiles == 2)) while (ct > 1 && lds_of(ct) > 53 * 1024) --ct;
    while (ct > 1 && wgs_of(ct) < 192) --ct;
    const int ct_env = knob_or("DEX_POS_CT", 0);
    if (ct_env >= 1 && ct_env <= 3 && ct_env <= tiles) ct = ct_env;
    if (ct == 3) launch_pc<3>(p, st);
    else if (ct == 2) launch_pc<2>(p, st);
    else launch_pc<1>(p, st);
}

}  // namespace DEX_LP_NS
}  // namespace dex
