// convt_up.hip — Upsample = ConvTranspose2d(64, 64, 4, 2, 1) on x * mask (diffusion.py:13-19, call site diffusion.py:211-216)
// as a strip-walking kernel (reduced-precision MFMA modes).
//
// Each output parity (oh & 1, ow & 1) of the transposed convolution is a 2x2-tap convolution of the input (dex_api.hip,
// pack_convt_kernel).  As four implicit GEMMs (igemm_bf16.hip) every workgroup wrote 64 B (16-bit output) or 128 B pieces of
// pixels two apart - the other parity fills the gaps from another workgroup at another time - and re-gathered its 2x2
// neighbourhood per parity: 228 us at B = 32 for 43 GFLOP and ~210 MB.  Here:
//   * wave w of a workgroup IS parity w: its whole weight matrix [64 co][4 taps x 64 ci] (32 KB) stays in registers as the
//     MFMA A operand for the lifetime of the workgroup (128 VGPRs; transposed product: C rows = co, columns = pixels);
//   * the workgroup walks a strip of 32*MT input columns down its rows; input rows (x * mask, 16-bit) live in a four-slot
//     LDS ring - every input row is fetched once per strip, a row ahead of its use - and are the B operand of all 4 waves;
//   * the 2 x (2*32*MT) output pixels of one input row are assembled in LDS (lane = pixel after the transposed product, so 4
//     consecutive channels pack into one LDS write) and leave as 16 B per lane: each output row segment is one contiguous run.
// Bias enters as the initial accumulator.  K order = the igemm's ((th*2+tw)*64 + ci), so the sums match it term by term.
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {

constexpr int CU_C = 64;           // channels (in == out)
constexpr int CU_PXB = 144;        // bytes per ring pixel: 64 x 16 bit + 16 B (conflict-free 16 B reads at a 144 B lane stride)

template <int MT, bool ALP, bool CLP>
struct CuGeom {
    static constexpr int MPX = 32 * MT, PC = MPX + 2, ROWB = PC * CU_PXB, RING = 4 * ROWB;
    static constexpr int OPX = 2 * MPX, SPB = CLP ? 144 : 272, STG = 2 * OPX * SPB;
    static constexpr int NL = (PC * 8 + 255) / 256;            // 16 B ring chunks per thread per input row
    static constexpr int OCH = CLP ? 8 : 16;                   // 16 B chunks per staged output pixel
    static constexpr int NS = 2 * OPX * OCH / 256;             // output chunks per thread per tile
    static constexpr int NB = NS < 8 ? NS : 8;                 // ... moved per batch (LDS reads first, then the stores)
    static constexpr int LDS = RING + STG + 256;
};

// one input row of the strip: issue the global loads (addresses clamped, `m` = mask or 0 outside the image)
template <int NL, int PC, bool ALP>
__device__ __forceinline__ void cu_row_load(const ConvTUpP& p, const void* Xb, const float* mrow, int row, int iw0, int tid,
                                            uint4 (&ra)[NL], uint4 (&rb)[ALP ? 1 : NL], float (&m)[NL]) {
    const bool rok = (unsigned)row < (unsigned)p.H;
    const int rc = rok ? row : 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int q = tid + 256 * j;
        const int px = q >> 3, c8 = (q & 7) * 8;
        const int iw = iw0 - 1 + px;
        const bool ok = rok && q < PC * 8 && (unsigned)iw < (unsigned)p.W;
        const int wc = ok ? iw : 0;
        const long e = ((long)rc * p.W + wc) * p.ldx + c8;
        if constexpr (ALP) ra[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(Xb) + e);
        else {
            ra[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(Xb) + e);
            rb[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(Xb) + e + 4);
        }
        const float mv = mrow[wc * p.inmask_ws];
        m[j] = ok ? mv : 0.f;
    }
}
template <int NL, int PC, bool ALP>
__device__ __forceinline__ void cu_row_store(unsigned char* slot, int tid, const uint4 (&ra)[NL], const uint4 (&rb)[ALP ? 1 : NL], const float (&m)[NL]) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int q = tid + 256 * j;
        const int px = q >> 3, c8 = (q & 7) * 8;
        uint4 v;
        if constexpr (ALP) {       // widened exactly; a 0 / 1 mask leaves the values on the operand grid
            v.x = pack2_lp(lp_lo(ra[j].x) * m[j], lp_hi(ra[j].x) * m[j]); v.y = pack2_lp(lp_lo(ra[j].y) * m[j], lp_hi(ra[j].y) * m[j]);
            v.z = pack2_lp(lp_lo(ra[j].z) * m[j], lp_hi(ra[j].z) * m[j]); v.w = pack2_lp(lp_lo(ra[j].w) * m[j], lp_hi(ra[j].w) * m[j]);
        } else {
            v.x = pack2_lp(__uint_as_float(ra[j].x) * m[j], __uint_as_float(ra[j].y) * m[j]);
            v.y = pack2_lp(__uint_as_float(ra[j].z) * m[j], __uint_as_float(ra[j].w) * m[j]);
            v.z = pack2_lp(__uint_as_float(rb[j].x) * m[j], __uint_as_float(rb[j].y) * m[j]);
            v.w = pack2_lp(__uint_as_float(rb[j].z) * m[j], __uint_as_float(rb[j].w) * m[j]);
        }
        if (q < PC * 8) *reinterpret_cast<uint4*>(slot + px * CU_PXB + c8 * 2) = v;
    }
}

}  // namespace

// grid (nseg * nchunk, B); 256 threads; wave = output parity (ph = wave >> 1, pw = wave & 1)
template <int MT, bool ALP, bool CLP>
__global__ __launch_bounds__(256) void convt_up_kernel(const ConvTUpP p) {
    using G = CuGeom<MT, ALP, CLP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cu[];
    unsigned char* ring = smem_cu;
    unsigned char* stage = smem_cu + G::RING;
    float* bs = reinterpret_cast<float*>(smem_cu + G::RING + G::STG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ph = wave >> 1, pw = wave & 1;
    const int seg = blockIdx.x % p.nseg, chunk = blockIdx.x / p.nseg, b = blockIdx.y;
    const int iw0 = seg * G::MPX;
    const int r0 = chunk * p.rows_per_wg, r1 = min(p.H, r0 + p.rows_per_wg);
    const void* Xb = ALP ? static_cast<const void*>(reinterpret_cast<const u16*>(p.X) + (long)b * p.xb + p.x_coff)
                         : static_cast<const void*>(reinterpret_cast<const float*>(p.X) + (long)b * p.xb + p.x_coff);
    const float* mrow = p.inmask + (long)b * p.mask_bstride;
    const int OW = 2 * p.W;

    // ---- everything the first tile needs goes out together: this parity's weights, three input rows, the bias
    uint4 wr[2][16];
    {
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag[wave]);      // [ct][ks][lane] x 16 B: 1 KB contiguous per wave load
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) wr[ct][ks] = Wf[(ct * 16 + ks) * 64 + lane];
    }
#ifdef DEX_LP_WSPLIT
    uint4 wl[2][16];                     // split weights: the lo halves of this parity's matrix (its pack of 2 x 16 x 64 fragments behind the hi one)
    {
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag[wave]) + 2L * 16 * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) wl[ct][ks] = Wf[(ct * 16 + ks) * 64 + lane];
    }
#endif
    uint4 ra[G::NL], rb[ALP ? 1 : G::NL];
    float rm[G::NL];
    {
        uint4 pa[3][G::NL], pb[3][ALP ? 1 : G::NL];
        float pm[3][G::NL];
#pragma unroll
        for (int d = 0; d < 3; ++d) cu_row_load<G::NL, G::PC, ALP>(p, Xb, mrow, r0 + d - 1, iw0, tid, pa[d], pb[d], pm[d]);
        const float bv = tid < CU_C ? p.bias[tid] : 0.f;
        cu_row_load<G::NL, G::PC, ALP>(p, Xb, mrow, r0 + 2 <= r1 ? r0 + 2 : -1, iw0, tid, ra, rb, rm);      // ring-written under tile r0
#pragma unroll
        for (int d = 0; d < 3; ++d) cu_row_store<G::NL, G::PC, ALP>(ring + ((r0 + d) & 3) * G::ROWB, tid, pa[d], pb[d], pm[d]);
        if (tid < CU_C) bs[tid] = bv;
    }
    lds_barrier();

    for (int ih = r0; ih < r1; ++ih) {
        // ---- MFMA phase: acc[ct][pt] (co tile x pixel tile) = bias + sum over the 4 taps of this parity
        f32x16 acc[2][MT];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bs + ct * 32 + 8 * g + 4 * hh);
#pragma unroll
                for (int pt = 0; pt < MT; ++pt) { acc[ct][pt][4 * g] = b4.x; acc[ct][pt][4 * g + 1] = b4.y; acc[ct][pt][4 * g + 2] = b4.z; acc[ct][pt][4 * g + 3] = b4.w; }
            }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int tap = ks >> 2, th = tap >> 1, tw = tap & 1;
            // input pixel (ih + ph - th, iw + pw - tw); ring slot of row r is (r + 1) & 3, ring column of iw is iw - iw0 + 1
            const unsigned char* xr = ring + ((ih + ph - th + 1) & 3) * G::ROWB + (i + 1 + pw - tw) * CU_PXB + ((ks & 3) * 16 + hh * 8) * 2;
#pragma unroll
            for (int pt = 0; pt < MT; ++pt) {
                const lp8 xb = *reinterpret_cast<const lp8*>(xr + pt * 32 * CU_PXB);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[ct][pt] = DEX_MFMA_LP(__builtin_bit_cast(lp8, wr[ct][ks]), xb, acc[ct][pt], 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                    acc[ct][pt] = DEX_MFMA_LP(__builtin_bit_cast(lp8, wl[ct][ks]), xb, acc[ct][pt], 0, 0, 0);
#endif
                }
            }
        }
        lds_barrier();              // every wave is past the previous tile's reads of the output stage, and past this tile's reads of the ring
        // ---- accumulators -> output stage: lane = pixel, 4 consecutive channels per register group
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < MT; ++pt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned char* d = stage + (ph * G::OPX + 2 * (pt * 32 + i) + pw) * G::SPB + (ct * 32 + 8 * g + 4 * hh) * (CLP ? 2 : 4);
                    if constexpr (CLP) *reinterpret_cast<uint2*>(d) = make_uint2(pack2_lp(acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1]), pack2_lp(acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]));
                    else *reinterpret_cast<float4*>(d) = make_float4(acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
                }
        // ---- input row ih + 2 (in flight since the previous tile) -> ring; its slot held row ih - 2
        cu_row_store<G::NL, G::PC, ALP>(ring + ((ih + 3) & 3) * G::ROWB, tid, ra, rb, rm);
        lds_barrier();
        // the loads of row ih + 3 go out BEFORE this tile's stores: the wait for them at the next ring write then does not
        // wait for the stores (vmcnt retires in order)
        cu_row_load<G::NL, G::PC, ALP>(p, Xb, mrow, ih + 3 <= r1 ? ih + 3 : -1, iw0, tid, ra, rb, rm);
        __builtin_amdgcn_sched_barrier(0);
        // ---- output stage -> HBM, 16 B per lane: consecutive lanes = consecutive bytes of an output row segment
#pragma unroll
        for (int h0 = 0; h0 < G::NS; h0 += G::NB) {
            uint4 ov[G::NB];
#pragma unroll
            for (int j = 0; j < G::NB; ++j) {
                const int q = tid + 256 * (h0 + j);
                const int row = q / (G::OPX * G::OCH), rem = q % (G::OPX * G::OCH);
                ov[j] = *reinterpret_cast<const uint4*>(stage + (row * G::OPX + rem / G::OCH) * G::SPB + (rem % G::OCH) * 16);
            }
#pragma unroll
            for (int j = 0; j < G::NB; ++j) {
                const int q = tid + 256 * (h0 + j);
                const int row = q / (G::OPX * G::OCH), rem = q % (G::OPX * G::OCH);
                const int ow = 2 * iw0 + rem / G::OCH;
                const long e = (((long)b * 2 * p.H + 2 * ih + row) * OW + ow) * p.ldy + p.y_coff + (rem % G::OCH) * (CLP ? 8 : 4);
                if (ow < OW) {
                    if constexpr (CLP) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.Y) + e) = ov[j];
                    else *reinterpret_cast<uint4*>(reinterpret_cast<float*>(p.Y) + e) = ov[j];
                }
            }
        }
    }
}

bool convt_up_supported(int C, int H, int W, int ldx, int ldy) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_UP)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return C == CU_C && (ldx % 8) == 0 && (ldy % 8) == 0 && H >= 1 && W >= 1; }

template <int MT, bool ALP, bool CLP>
static void cu_launch(ConvTUpP p, hipStream_t st) {
    using G = CuGeom<MT, ALP, CLP>;
    static_assert(G::NS % G::NB == 0, "output chunks");
    p.nseg = (p.W + G::MPX - 1) / G::MPX;
    // ~2 workgroups per CU; a workgroup re-reads two halo rows per row chunk, so chunks stay as long as the grid allows
    const long tiles = (long)p.H * p.nseg * p.B;
    const int target = knob_or("DEX_CONVT_WGS", 512);
    int R = (int)((tiles + target - 1) / target);      // (rounded UP: the long form's 2 520 row segments were 630 workgroups on 512 slots with R = 4; R = 5 is 504)
    if (R < 1) R = 1;
    if (R > p.H) R = p.H;
    const int nchunk = (p.H + R - 1) / R;
    p.rows_per_wg = R;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_up_kernel<MT, ALP, CLP>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS); attr = true; }
    hipLaunchKernelGGL((convt_up_kernel<MT, ALP, CLP>), dim3(p.nseg * nchunk, p.B), dim3(256), G::LDS, st, p);
}

void launch_convt_up(const ConvTUpP& p, hipStream_t st) {
    g_last_symbol = "convt_up_kernel";
    // 32-column strips (two workgroups per CU, 222 registers) beat 64-column ones (one per CU, 292) at every grid measured:
    // 60.8 vs 72.4 us (GeDEX B = 32), 36.7 vs 40.2 us (DEX B = 32); DEX_CONVT_MT=2 keeps the wide form reachable
    const bool wide = knob_or("DEX_CONVT_MT", 0) == 2;
#define CU_GO(MT) do { if (p.a_lp && p.c_lp) cu_launch<MT, true, true>(p, st); else if (p.a_lp) cu_launch<MT, true, false>(p, st); \
                       else if (p.c_lp) cu_launch<MT, false, true>(p, st); else cu_launch<MT, false, false>(p, st); } while (0)
    if (wide) CU_GO(2); else CU_GO(1);
#undef CU_GO
}

}  // namespace DEX_LP_NS
}  // namespace dex
