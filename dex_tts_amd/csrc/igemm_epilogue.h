// igemm_epilogue.h — shared epilogue of the implicit-GEMM kernels (fp32 and bf16 MFMA variants).
// Accumulator layout of a 32x32 MFMA tile (dtype-independent on gfx950):
//   col = lane & 31,  row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5),  r in [0,16).
// bias -> [GroupNorm partial statistics of the raw conv output] -> act -> adaLN gate -> +residual -> mask -> store
//
// Written in phases per 32x32 tile (indices, then ALL residual/mask loads with clamped addresses, then math,
// then predicated stores) so the 16 dependent global loads of a lane are in flight together instead of
// serialising behind per-element branches.
#pragma once
#include "kernels.h"
#include "bf16_util.h"

namespace dex {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// rows [row0, row0 + MT*32) x cols [col0, col0+32) of the block tile belong to this wave.
template <int MT, bool CLP = false>       // CLP: C holds 16-bit elements (IGemmP::c_lp = the type)
__device__ __forceinline__ void igemm_epilogue(const IGemmP& p, f32x16 (&acc)[MT], int m0, int n0, int row0, int col0,
                                               int lane, int b, int g, int s, int M, int oh0, int ow0) {
    const int i = lane & 31, hh = lane >> 5;
    const int n = n0 + col0 + i;                 // column within the group
    const int ng = g * p.N + n;                  // global output channel
    const int step = p.step;
    const float bias = p.bias ? p.bias[(long)b * p.bias_bstride + ng] : 0.f;
    const float gate = p.gate ? p.gate[(long)step * p.gate_step_stride + (long)ng * p.gate_nstride] : 1.f;
    const float* omask = p.outmask ? p.outmask + (long)b * p.mask_bstride : nullptr;
    // (c_lp: the tensor holds 16-bit elements; the same element offsets on a 16-bit pointer, kept in a float* variable)
    float* Cb = CLP ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(p.C) + (long)b * p.c_bstride + (long)s * p.c_sstride + p.c_coff)
                       : p.C + (long)b * p.c_bstride + (long)s * p.c_sstride + p.c_coff;
    const float* Rb = p.res ? p.res + (long)b * p.res_bstride + p.res_coff : nullptr;
    const bool unp = p.unpatch_s > 0;
    int up_c = 0, up_p1 = 0, up_p2 = 0;
    if (unp) {
        const int pp = ng / p.unpatch_C;
        up_c = ng - pp * p.unpatch_C;
        up_p1 = pp / p.unpatch_s;
        up_p2 = pp - up_p1 * p.unpatch_s;
    }
    const int col_out = unp ? up_c : ng;
    float gs = 0.f, gss = 0.f;                   // GroupNorm partials of this lane's column
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        // Fast path: the 32 rows of this tile are 32 consecutive pixels of ONE output row, all in range (every layer of
        // the network at widths that are multiples of 32, and every Linear).  One division per tile, one 64-bit base per
        // tensor, 32-bit row offsets — the generic path below spends ~40 integer instructions per element.
        const int mt0 = m0 + row0 + t * 32;
        const int ho_t = mt0 / p.Wo, wo_t = mt0 - ho_t * p.Wo;
        if (!unp && mt0 + 32 <= M && wo_t + 32 <= p.Wo) {
            const int oh = ho_t * p.osh + oh0, ow0_ = (wo_t + 4 * hh) * p.osw + ow0;
            const long pix0 = (long)oh * p.OWf + ow0_;
            float* cp = Cb + pix0 * p.ldc + col_out;
            const float* rp = Rb ? Rb + pix0 * p.ldres + ng : nullptr;
            const float* mp = omask ? omask + (long)ow0_ * p.outmask_ws : nullptr;
            const int cs = p.osw * p.ldc, rs = p.osw * p.ldres, ms = p.osw * p.outmask_ws;
            float rv[16], mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rp ? rp[((r & 3) + 8 * (r >> 2)) * rs] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = mp ? mp[((r & 3) + 8 * (r >> 2)) * ms] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r] + bias;
                if (!p.stats_final) { gs += v; gss = fmaf(v, v, gss); }
                if (p.act == 1) v = gelu_erf(v); else if (p.act == 2) v = fmaxf(v, 0.f);
                v = (v * gate + rv[r]) * mk[r];
                if (p.stats_final) { gs += v; gss = fmaf(v, v, gss); }
                if constexpr (CLP) reinterpret_cast<unsigned short*>(Cb)[(pix0 + ((r & 3) + 8 * (r >> 2)) * p.osw) * p.ldc + col_out] =
                                (unsigned short)(pack2_kind(v, 0.f, p.c_lp) & 0xffffu);
                else cp[((r & 3) + 8 * (r >> 2)) * cs] = v;
            }
            continue;
        }
        int opix[16], ow_[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const int m = m0 + row;
            bool v = m < M;
            const int mm = v ? m : 0;
            const int ho = mm / p.Wo, wo = mm - ho * p.Wo;
            int oh, ow;
            if (unp) { oh = ho * p.unpatch_s + up_p1; ow = wo * p.unpatch_s + up_p2; v = v && oh < p.OHf && ow < p.OWf; }
            else { oh = ho * p.osh + oh0; ow = wo * p.osw + ow0; }
            ok[r] = v;
            opix[r] = v ? oh * p.OWf + ow : 0;
            ow_[r] = v ? ow : 0;
        }
        float rv[16], mk[16];
        if (Rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = Rb[(long)opix[r] * p.ldres + ng];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        }
        if (omask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = omask[ow_[r] * p.outmask_ws];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = 1.f;
        }
        // values first - every residual / mask load is consumed here -, then the stores: a store inside a per-lane branch that
        // still depends on a load gets its own s_waitcnt vmcnt(0), which also waits for every EARLIER STORE to complete, i.e. the
        // 16 stores of a tile went out one write latency at a time (seen in the ISA; the unpatchify GEMM took 245 us at B=32)
        float val[16];
        bool all_ok = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + bias;
            const float vs = ok[r] ? v : 0.f;
            if (!p.stats_final) { gs += vs; gss = fmaf(vs, vs, gss); }
            if (p.act == 1) v = gelu_erf(v); else if (p.act == 2) v = fmaxf(v, 0.f);
            v = (v * gate + rv[r]) * mk[r];
            if (p.stats_final && ok[r]) { gs += v; gss = fmaf(v, v, gss); }
            val[r] = v;
            all_ok = all_ok && ok[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (__builtin_amdgcn_ballot_w64(!all_ok) == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (CLP) reinterpret_cast<unsigned short*>(Cb)[(long)opix[r] * p.ldc + col_out] = (unsigned short)(pack2_kind(val[r], 0.f, p.c_lp) & 0xffffu);
                else Cb[(long)opix[r] * p.ldc + col_out] = val[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r]) {
                    if constexpr (CLP) reinterpret_cast<unsigned short*>(Cb)[(long)opix[r] * p.ldc + col_out] = (unsigned short)(pack2_kind(val[r], 0.f, p.c_lp) & 0xffffu);
                    else Cb[(long)opix[r] * p.ldc + col_out] = val[r];
                }
        }
    }
    if (p.gn_stats) {
        // channels-per-group cpg in {8,16,32}: reduce over the cpg lanes of a group and over both half-waves,
        // then ONE fixed-point atomic pair per group per wave into slot (blockIdx.x % GN_SLOTS) — slots spread the
        // same-address atomic traffic; integer adds commute, so the totals do not depend on the arrival order.
        const int cpg = p.gn_cpg;
        for (int o = 1; o < cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
        gs += __shfl_xor(gs, 32); gss += __shfl_xor(gss, 32);
        if (hh == 0 && (i & (cpg - 1)) == 0) {
            const int grp = ng / cpg;
            gnfix_t* dst = p.gn_stats + (((long)b * p.gn_groups + grp) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2;
            const double inv_n = 1.0 / ((double)p.Ho * p.Wo * cpg);
            gn_add(dst, gn_fix(gs, inv_n));
            gn_add(dst + 1, gn_fix(gss, inv_n));
        }
    }
}

}  // namespace dex
