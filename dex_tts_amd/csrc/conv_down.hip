// conv_down.hip — Downsample = Conv2d(64, 64, 3, stride 2, pad 1) on x * mask (diffusion.py:22-28, call site diffusion.py:189)
// as a strip-walking kernel (reduced-precision MFMA modes; 16-bit input at batch size, fp32 input otherwise).
//
// As an implicit GEMM (igemm_bf16.hip) every workgroup re-gathers the nine taps of its 128 output pixels from L2 and the A tile
// goes global -> registers -> LDS per K tile: 102 us at B = 32 for 210 MB of HBM traffic (~2 TB/s).  Here, as in convt_up.hip:
//   * wave (ct, pt) of a workgroup owns output-channel tile ct x pixel tile pt of a 64-column output strip and keeps its whole
//     weight matrix [32 co][9 taps x 64 ci] (36 K-steps, 144 VGPRs) in registers as the MFMA A operand;
//   * the workgroup walks the strip down the output rows; the three input rows of an output row (x * mask, 16-bit) live in a
//     three-slot LDS ring with the even and the odd input columns in separate planes, so that the stride-2 pixel operand of a tap
//     is a unit-stride LDS read; output row oh + 1 needs two new input rows, which are fetched one output row ahead and replace
//     the two dead ones between the barriers of a tile;
//   * the 64 output pixels of a tile are assembled in LDS (lane = pixel: four consecutive channels per LDS write) and leave as
//     16 B per lane, one contiguous run.
// K order = the igemm's ((kh*3 + kw)*64 + ci) and the bias is added to the finished sum as there: bit-identical results.
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {
constexpr int CD_C = 64, CD_MPX = 64, CD_PXB = 144;                 // channels; output columns per strip; bytes per ring pixel (64 x 16 bit + 16)
constexpr int CD_PL = CD_MPX + 1, CD_ROWB = 2 * CD_PL * CD_PXB;      // pixels per column-parity plane (input columns 2*ow0 - 1 .. 2*ow0 + 127), bytes per ring row
constexpr int CD_NIT = (2 * CD_MPX + 1) * 8, CD_NL = (CD_NIT + 255) / 256;     // 16 B items per input row / per thread
}

template <bool ALP> struct CdRow;                                   // registers of one input row in flight
template <> struct CdRow<true> { uint4 a[CD_NL]; float m[CD_NL]; };
template <> struct CdRow<false> { uint4 a[CD_NL]; uint4 c[CD_NL]; float m[CD_NL]; };

// grid (nseg * nchunk, B); 256 threads
template <bool ALP, bool CLP>
#ifdef DEX_LP_WSPLIT
#define CD_WAVES_ATTR                // (twice the weight registers: one wave per SIMD)
#else
#define CD_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(ALP ? 2 : 1, ALP ? 2 : 1)))
#endif
__global__ __launch_bounds__(256) CD_WAVES_ATTR void conv_down_kernel(const ConvDownP p) {
    constexpr int SPB = CLP ? 144 : 272, STG = CD_MPX * SPB, OCH = CLP ? 8 : 16, NS = CD_MPX * OCH / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cd[];
    unsigned char* ring = smem_cd;                                   // [3 slots][2 planes][CD_PL] pixels
    unsigned char* stage = smem_cd + 3 * CD_ROWB;
    float* bs = reinterpret_cast<float*>(stage + STG);
    const int DUMMY = 3 * CD_ROWB + STG + CD_C * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ct = wave & 1, pt = wave >> 1;
    const int seg = blockIdx.x % p.nseg, chunk = blockIdx.x / p.nseg, b = blockIdx.y;
    const int ow0 = seg * CD_MPX, Ho = p.H / 2, Wo = p.W / 2;
    const int r0 = chunk * p.rows_per_wg, r1 = min(Ho, r0 + p.rows_per_wg);
    const bool full_strip = ow0 + CD_MPX <= Wo;
    const u16* Xb = reinterpret_cast<const u16*>(p.X) + (long)b * p.xb + p.x_coff;                 // ALP: 16-bit input
    const float* Xf = reinterpret_cast<const float*>(p.X) + (long)b * p.xb + p.x_coff;
    const float* mrow = p.inmask + (long)b * p.mask_bstride;

    // per-thread constants of the row loads: element offset inside an image row (column clamped), column mask, ring offset
    int coff[CD_NL], roff[CD_NL]; float cmask[CD_NL];
#pragma unroll
    for (int j = 0; j < CD_NL; ++j) {
        const int q = tid + 256 * j;
        const int pc = q >> 3, c8 = (q & 7) * 8;                     // patch column (input column 2*ow0 - 1 + pc), channel chunk
        const int wi = 2 * ow0 - 1 + pc;
        const bool ok = q < CD_NIT && (unsigned)wi < (unsigned)p.W;
        const int wc = ok ? wi : 0;
        coff[j] = wc * p.ldx + c8;
        cmask[j] = ok ? mrow[wc * p.inmask_ws] : 0.f;
        roff[j] = q < CD_NIT ? ((pc & 1) * CD_PL + (pc >> 1)) * CD_PXB + c8 * 2 : -1;
    }
    using Row = CdRow<ALP>;
    auto row_load = [&](int row, Row& R) __attribute__((always_inline)) {
        const bool rok = (unsigned)row < (unsigned)p.H;
        const int rc = __builtin_amdgcn_readfirstlane(rok ? row : 0);
#pragma unroll
        for (int j = 0; j < CD_NL; ++j) {
            if constexpr (ALP) R.a[j] = *reinterpret_cast<const uint4*>(Xb + (long)rc * p.W * p.ldx + coff[j]);
            else { const float* xf = Xf + (long)rc * p.W * p.ldx + coff[j]; R.a[j] = *reinterpret_cast<const uint4*>(xf); R.c[j] = *reinterpret_cast<const uint4*>(xf + 4); }
            R.m[j] = rok ? cmask[j] : 0.f;
        }
    };
    auto row_store = [&](int row, const Row& R) __attribute__((always_inline)) {      // ring slot of input row r: (r + 1) mod 3
        const int slot = __builtin_amdgcn_readfirstlane((row + 1) % 3) * CD_ROWB;
#pragma unroll
        for (int j = 0; j < CD_NL; ++j) {
            const float m = R.m[j];          // the 16-bit values widened exactly; a 0 / 1 mask leaves them on the operand grid
            uint4 v;
            if constexpr (ALP) {
                v.x = pack2_lp(lp_lo(R.a[j].x) * m, lp_hi(R.a[j].x) * m); v.y = pack2_lp(lp_lo(R.a[j].y) * m, lp_hi(R.a[j].y) * m);
                v.z = pack2_lp(lp_lo(R.a[j].z) * m, lp_hi(R.a[j].z) * m); v.w = pack2_lp(lp_lo(R.a[j].w) * m, lp_hi(R.a[j].w) * m);
            } else {
                v.x = pack2_lp(__uint_as_float(R.a[j].x) * m, __uint_as_float(R.a[j].y) * m); v.y = pack2_lp(__uint_as_float(R.a[j].z) * m, __uint_as_float(R.a[j].w) * m);
                v.z = pack2_lp(__uint_as_float(R.c[j].x) * m, __uint_as_float(R.c[j].y) * m); v.w = pack2_lp(__uint_as_float(R.c[j].z) * m, __uint_as_float(R.c[j].w) * m);
            }
            *reinterpret_cast<uint4*>(smem_cd + (roff[j] >= 0 ? slot + roff[j] : DUMMY)) = v;
        }
    };

    // ---- prologue: this wave's weights, the three input rows of the first output row, the next two in flight
    Row Ra, Rb;
    row_load(2 * r0 - 1, Ra);
    row_load(2 * r0, Rb);
    uint4 wr[36];
    {
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag) + (long)ct * 36 * 64 + lane;      // [ct][tap * 4 + ks][lane] x 16 B
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) wr[ks] = Wf[ks * 64];
    }
#ifdef DEX_LP_WSPLIT
    uint4 wl[36];                        // split weights: the lo halves (the whole [2 ct][36][64 lanes] pack behind the hi one)
    {
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag) + 2L * 36 * 64 + (long)ct * 36 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) wl[ks] = Wf[ks * 64];
    }
#endif
    if (tid < CD_C) bs[tid] = p.bias[tid];
    row_store(2 * r0 - 1, Ra);
    row_store(2 * r0, Rb);
    row_load(2 * r0 + 1, Ra);
    row_store(2 * r0 + 1, Ra);
    row_load(r0 + 1 < r1 ? 2 * r0 + 2 : -1, Ra);
    row_load(r0 + 1 < r1 ? 2 * r0 + 3 : -1, Rb);
    lds_barrier();

    for (int oh = r0; oh < r1; ++oh) {
        // ---- MFMA chain: 32 channels x 32 pixels = bias + 9 taps x 64 ci
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const unsigned char* rowp = ring + __builtin_amdgcn_readfirstlane((2 * oh + kh) % 3) * CD_ROWB;      // input row 2*oh - 1 + kh
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                // input column 2*(ow0 + pt*32 + i) - 1 + kw = patch column 2*(pt*32 + i) + kw
                const unsigned char* xr = rowp + ((kw & 1) * CD_PL + pt * 32 + i + (kw >> 1)) * CD_PXB + hh * 16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const lp8 xv = *reinterpret_cast<const lp8*>(xr + ks * 32);
                    acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, wr[(kh * 3 + kw) * 4 + ks]), xv, acc, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                    acc = DEX_MFMA_LP(__builtin_bit_cast(lp8, wl[(kh * 3 + kw) * 4 + ks]), xv, acc, 0, 0, 0);
#endif
                }
            }
        }
        lds_barrier();              // every wave is past its reads of the ring and of the previous tile's output stage
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned char* d = stage + (pt * 32 + i) * SPB + (ct * 32 + 8 * g + 4 * hh) * (CLP ? 2 : 4);
            const float4 b4 = *reinterpret_cast<const float4*>(bs + ct * 32 + 8 * g + 4 * hh);      // (sum first, then the bias: the igemm's order)
            const float v0 = acc[4 * g] + b4.x, v1 = acc[4 * g + 1] + b4.y, v2 = acc[4 * g + 2] + b4.z, v3 = acc[4 * g + 3] + b4.w;
            if constexpr (CLP) *reinterpret_cast<uint2*>(d) = make_uint2(pack2_lp(v0, v1), pack2_lp(v2, v3));
            else *reinterpret_cast<float4*>(d) = make_float4(v0, v1, v2, v3);
        }
        // input rows 2*oh + 2, 2*oh + 3 (in flight since the previous tile) replace rows 2*oh - 1, 2*oh
        row_store(2 * oh + 2, Ra);
        row_store(2 * oh + 3, Rb);
        lds_barrier();
        // the loads of the next tile's new rows go out BEFORE this tile's stores (vmcnt retires in order)
        row_load(oh + 2 < r1 ? 2 * oh + 4 : -1, Ra);
        row_load(oh + 2 < r1 ? 2 * oh + 5 : -1, Rb);
        __builtin_amdgcn_sched_barrier(0);
        {
            uint4 ov[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) { const int q = tid + 256 * j; ov[j] = *reinterpret_cast<const uint4*>(stage + (q / OCH) * SPB + (q % OCH) * 16); }
            const long rowbase = (((long)b * Ho + oh) * Wo) * p.ldy + p.y_coff;          // wave-uniform
#define CD_OUT(pred_) _Pragma("unroll") for (int j = 0; j < NS; ++j) { \
                const int q = tid + 256 * j, wo = ow0 + q / OCH, e = wo * p.ldy + (q % OCH) * (CLP ? 8 : 4); \
                if (pred_) { if constexpr (CLP) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.Y) + rowbase + e) = ov[j]; \
                             else *reinterpret_cast<uint4*>(reinterpret_cast<float*>(p.Y) + rowbase + e) = ov[j]; } }
            if (full_strip) { CD_OUT(true) } else { CD_OUT(wo < Wo) }
#undef CD_OUT
        }
    }
}

bool conv_down_supported(int C, int H, int W, int ldx, int ldy, int x_coff) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_DOWN)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    return C == CD_C && (ldx % 8) == 0 && (ldy % 8) == 0 && (x_coff % 8) == 0 && (H % 2) == 0 && (W % 2) == 0 && H >= 2 && W >= 2;
}

void launch_conv_down(const ConvDownP& p0, hipStream_t st) {
    g_last_symbol = "conv_down_kernel";
    ConvDownP p = p0;
    const int Ho = p.H / 2, Wo = p.W / 2;
    p.nseg = (Wo + CD_MPX - 1) / CD_MPX;
    // ~2 workgroups per CU (a workgroup re-reads one halo row per row chunk)
    const int target = knob_or("DEX_CONV_DOWN_WGS", 512);
    int nchunk = (target + p.nseg * p.B - 1) / (p.nseg * p.B);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > Ho) nchunk = Ho;
    p.rows_per_wg = (Ho + nchunk - 1) / nchunk;
    nchunk = (Ho + p.rows_per_wg - 1) / p.rows_per_wg;
    const int lds = 3 * CD_ROWB + CD_MPX * (p.c_lp ? 144 : 272) + CD_C * 4 + 16;
    static bool attr = false;
    if (!attr) {
        constexpr int LMAX = 3 * CD_ROWB + CD_MPX * 272 + CD_C * 4 + 16;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_down_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LMAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_down_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LMAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_down_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LMAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_down_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LMAX);
        attr = true;
    }
    const dim3 grid(p.nseg * nchunk, p.B);
    if (p.a_lp) { if (p.c_lp) hipLaunchKernelGGL((conv_down_kernel<true, true>), grid, dim3(256), lds, st, p); else hipLaunchKernelGGL((conv_down_kernel<true, false>), grid, dim3(256), lds, st, p); }
    else { if (p.c_lp) hipLaunchKernelGGL((conv_down_kernel<false, true>), grid, dim3(256), lds, st, p); else hipLaunchKernelGGL((conv_down_kernel<false, false>), grid, dim3(256), lds, st, p); }
}

}  // namespace DEX_LP_NS
}  // namespace dex
