// patch_embed.hip — PatchEmbed2D of the DiT bottleneck (dit.py:57-58: overlapping depthwise k x k conv, SiLU, pointwise C -> hidden) as
// ONE launch for small grids (reduced-precision modes).  At B = 1 the two separate kernels are 7.0 + 8.4 us of pure latency for 2 MFLOP
// of depthwise work and a 650 x 128 x 256 GEMM; here a workgroup owns EIGHT tokens: its 256 threads compute the depthwise outputs of
// the 8 x C/4 (token, channel quad) items (all 49 taps of an item unconditional loads in flight together, as in dwconv_silu_direct_kernel),
// round SiLU(.) to the MFMA operand type into an LDS A tile [32 rows, 8 live][C] and the four waves run the pointwise GEMM on it —
// the weight fragments (requested at kernel entry, independent of the depthwise phase) come straight from the packed [hidden][C]
// 16-bit twin.  Same arithmetic in the same order as dwconv_silu + igemm_lp_ss (depthwise in fp32, one rounding, K order 0..C):
// bit-identical output.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {
constexpr int PE_TOK = 8;
__device__ __forceinline__ float silu_pe(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
}

template <int KS, int C>
__global__ __launch_bounds__(256) void patch_embed_fused_kernel(const DwConvP p, const u16* __restrict__ Wb, const float* __restrict__ bias, float* __restrict__ emb, int hid) {
    constexpr int C4 = C / 4, LD = C + 8, KSTEPS = C / 16;
    constexpr int IPT = PE_TOK * C4 / 256;                 // items per thread (C = 128: 1, C = 64: the upper half of the threads idles)
    __shared__ __attribute__((aligned(16))) u16 As[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const long ntok = (long)p.B * p.Hf * p.Wt;
    const long tok0 = (long)blockIdx.x * PE_TOK;
    // pointwise weight fragments of this wave's column tiles: B operand lane = (column i, K half hh), 8 consecutive k
    const int ntile = hid / 32, per_wave = ntile / 4;       // hid % 128 == 0
    union Fr { uint4 u; lp8 v; };
    Fr wf[2][KSTEPS];                                       // up to two column tiles in flight per pass (hidden 256: all of them)
#ifdef DEX_LP_WSPLIT
    Fr wl[2][KSTEPS];                                       // split weights: the lo halves, hid * C elements behind (the twin's second pack)
#endif
    auto wload_tile = [&](int slot, int nt) __attribute__((always_inline)) {
        const u16* src = Wb + (long)(nt * 32 + i) * C + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) wf[slot][ks].u = *reinterpret_cast<const uint4*>(src + ks * 16);
    };
    // (the lo halves are requested after the depthwise phase: next to its 49 taps in flight they would not fit the register file)
    auto wload_lo = [&](int slot, int nt) __attribute__((always_inline)) {
#ifdef DEX_LP_WSPLIT
        const u16* src = Wb + (long)hid * C + (long)(nt * 32 + i) * C + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) wl[slot][ks].u = *reinterpret_cast<const uint4*>(src + ks * 16);
#endif
    };
    wload_tile(0, wave * per_wave);
    if (per_wave > 1) wload_tile(1, wave * per_wave + 1);
    // zero rows 8..31 of the A tile (the MFMA reads 32 rows)
    for (int q = tid; q < (32 - PE_TOK) * LD / 8; q += 256) *reinterpret_cast<uint4*>(As + PE_TOK * LD + q * 8) = make_uint4(0, 0, 0, 0);
    // ---- depthwise conv + SiLU of this workgroup's tokens
    if (tid < PE_TOK * C4) {
        const int cq = tid % C4, tl = tid / C4;
        const long tok = min(tok0 + tl, ntok - 1);
        const int wt = (int)(tok % p.Wt);
        const int f = (int)((tok / p.Wt) % p.Hf);
        const int b = (int)(tok / ((long)p.Wt * p.Hf));
        const float* X = p.X + (long)b * p.xb;
        const float* mrow = p.mask ? p.mask + (long)b * p.mask_bstride : nullptr;
        float4 acc = *reinterpret_cast<const float4*>(p.bd + cq * 4);
        // (DwConvP::aff: the DEX TIV adaptor's per-channel affine applied on load, as in dit_elem.hip)
        const float4 a4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 0) * p.C + cq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 c4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 1) * p.C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int hi = f * p.s + kh - p.pad;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int wi = wt * p.s + kw - p.pad;
                const bool inb = (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
                const int hc = inb ? hi : 0, wc = inb ? wi : 0;              // clamped: loads are unconditional
                float4 v = *reinterpret_cast<const float4*>(X + ((long)hc * p.Wi + wc) * p.ldx + cq * 4);
                if (p.aff) { v.x = fmaf(v.x, a4.x, c4.x); v.y = fmaf(v.y, a4.y, c4.y); v.z = fmaf(v.z, a4.z, c4.z); v.w = fmaf(v.w, a4.w, c4.w); }
                const float4 w = *reinterpret_cast<const float4*>(p.Wd + (kh * KS + kw) * p.C + cq * 4);
                float mk = mrow ? mrow[wc * p.mask_ws] : 1.f;
                mk = inb ? mk : 0.f;
                acc.x = fmaf(v.x * mk, w.x, acc.x); acc.y = fmaf(v.y * mk, w.y, acc.y);
                acc.z = fmaf(v.z * mk, w.z, acc.z); acc.w = fmaf(v.w * mk, w.w, acc.w);
            }
        }
        uint2 o;
        o.x = pack2_lp(silu_pe(acc.x), silu_pe(acc.y)); o.y = pack2_lp(silu_pe(acc.z), silu_pe(acc.w));
        *reinterpret_cast<uint2*>(As + tl * LD + cq * 4) = o;
    }
    wload_lo(0, wave * per_wave);
    if (per_wave > 1) wload_lo(1, wave * per_wave + 1);
    __syncthreads();
    // ---- pointwise GEMM: this wave's column tiles, K = C
    const u16* a_lane = As + i * LD + hh * 8;
    for (int t0 = 0; t0 < per_wave; t0 += 2) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            if (t0 + s_ >= per_wave) break;
            const int nt = wave * per_wave + t0 + s_;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const lp8 af = *reinterpret_cast<const lp8*>(a_lane + ks * 16);
                acc = DEX_MFMA_LP(af, wf[s_][ks].v, acc, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                acc = DEX_MFMA_LP(af, wl[s_][ks].v, acc, 0, 0, 0);
#endif
            }
            const float bv = bias ? bias[nt * 32 + i] : 0.f;
            // accumulator rows (r & 3) + 8 (r >> 2) + 4 hh: rows 0..7 are r = 0..3 of both lane halves
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowl = r + 4 * hh;
                if (tok0 + rowl < ntok) emb[(tok0 + rowl) * hid + nt * 32 + i] = acc[r] + bv;
            }
        }
        if (t0 + 2 < per_wave) {                           // hidden > 256: next pair of column tiles
            wload_tile(0, wave * per_wave + t0 + 2); wload_lo(0, wave * per_wave + t0 + 2);
            if (t0 + 3 < per_wave) { wload_tile(1, wave * per_wave + t0 + 3); wload_lo(1, wave * per_wave + t0 + 3); }
        }
    }
}

bool patch_embed_fused_supported(int k, int C, int hid, long ntok) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_PE)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    // the small-grid regime: measured (fused vs depthwise kernel + GEMM, us) 1 015 tokens (T = 800) 14.1 vs 16.1, 5 010 tokens (T = 4000) 41.6 vs 35.0,
    // 20 800 tokens (B = 32) 153 vs 94 - the one launch pays off up to a few thousand tokens
    return (k == 3 || k == 7) && (C == 64 || C == 128) && hid % 128 == 0 && (ntok * (C / 4) + 255) / 256 < 320;
}

void launch_patch_embed_fused(const DwConvP& p, const void* Wb, const float* bias, float* emb, int hid, hipStream_t st) {
    const long ntok = (long)p.B * p.Hf * p.Wt;
    const dim3 grid((unsigned)((ntok + PE_TOK - 1) / PE_TOK));
    const u16* w = reinterpret_cast<const u16*>(Wb);
    g_last_symbol = "patch_embed_fused_kernel";
    if (p.k == 7 && p.C == 128) hipLaunchKernelGGL((patch_embed_fused_kernel<7, 128>), grid, dim3(256), 0, st, p, w, bias, emb, hid);
    else if (p.k == 7) hipLaunchKernelGGL((patch_embed_fused_kernel<7, 64>), grid, dim3(256), 0, st, p, w, bias, emb, hid);
    else if (p.C == 128) hipLaunchKernelGGL((patch_embed_fused_kernel<3, 128>), grid, dim3(256), 0, st, p, w, bias, emb, hid);
    else hipLaunchKernelGGL((patch_embed_fused_kernel<3, 64>), grid, dim3(256), 0, st, p, w, bias, emb, hid);
}

}  // namespace DEX_LP_NS
}  // namespace dex
