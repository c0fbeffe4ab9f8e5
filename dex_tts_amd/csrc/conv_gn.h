// conv_gn.h — the GroupNorm-prologue coefficient table shared by the 3x3 convolution kernels (conv3x3_bf16.hip,
// conv3x3_regw.hip): the producer's statistics, affine and time bias folded to  y = Mish(x * coef0 + coef1) + coef2.
#pragma once
#include "kernels.h"
#include "bf16_util.h"

namespace dex {

// Prologue coefficients of the producer's GroupNorm, per input channel:  y = Mish(x * coef0 + coef1) + coef2.
// Thread (group g = tid/32, k = tid%32) of the first 256: one slot of group g's fp32 partials; after the xor-reduction every
// lane of the group holds the sums, so lane k < Cin/8 finishes channel g*Cin/8 + k itself.  The loads (statistics, gamma,
// beta, time bias: ONE value per thread) are issued before the weight/patch loads so they return first; the table lives in
// LDS.  (The first version had every thread fetch gamma/beta/time bias of its 8 channels: 24 KB of redundant L1 traffic per
// workgroup in the phase where the TA is the bottleneck - the load-issue phase measured 4.4k cycles per workgroup at B=1.)
struct CvGnLoads { unsigned s1l, s1h, s2l, s2h; float ga, be, ta; };
__device__ __forceinline__ CvGnLoads cv_gn_issue(const Conv3P& p, int b, int tid, int step) {
    CvGnLoads l{0u, 0u, 0u, 0u, 0.f, 0.f, 0.f};
    if (tid < 8 * GN_SLOTS) {
        const int g = tid / GN_SLOTS, k = tid % GN_SLOTS, cpg = p.Cin / 8;
        const uint4 v = *reinterpret_cast<const uint4*>(p.pro_stats + (((long)b * 8 + g) * GN_SLOTS + k) * 2);   // one 16-byte load
        l.s1l = v.x; l.s1h = v.y; l.s2l = v.z; l.s2h = v.w;
        if (k < cpg) {
            const int c = g * cpg + k;
            l.ga = p.pro_gamma[c]; l.be = p.pro_beta[c];
            l.ta = p.pro_tadd ? p.pro_tadd[(long)step * p.Cin + c] : 0.f;
        }
    }
    return l;
}
template <int CN>
__device__ __forceinline__ void cv_gn_finish(const Conv3P& p, const CvGnLoads& l, int tid, float (*coef)[CN]) {
    if (tid >= 8 * GN_SLOTS) return;                  // 512-thread workgroups: whole waves 4..7 sit this out
    const int g = tid / GN_SLOTS, k = tid % GN_SLOTS, cpg = p.Cin / 8;
    // first touch of the loaded partials through a pinned instruction: as a plain conversion it is hoisted to right behind
    // the load, and the s_waitcnt vmcnt(0) that comes with it stalls the workgroup a full round trip BEFORE the weight
    // and patch loads are even issued (seen in the ISA of the first version of this function)
    long long s1 = (long long)(((unsigned long long)mov_pinned(l.s1h) << 32) | mov_pinned(l.s1l));
    long long s2 = (long long)(((unsigned long long)mov_pinned(l.s2h) << 32) | mov_pinned(l.s2l));
    gn_slots_reduce<GN_SLOTS>(s1, s2);
    float rstd, mu;
    gn_moments(s1, s2, 1e-5, mu, rstd);
    if (k < cpg) {
        const int c = g * cpg + k;
        coef[0][c] = rstd * l.ga;
        coef[1][c] = l.be - mu * rstd * l.ga;
        coef[2][c] = l.ta;
    }
}

}  // namespace dex
