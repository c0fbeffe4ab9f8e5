// tools/attnw.hip — round-3 candidates for the batch-regime DiT attention (attention_direct.hip's ring kernel): NW waves of 32
// queries share each staged K / V^T tile (NW = 4 is the shipped shape; 8 and 12 cut the LDS-DMA refill + barrier per MFMA to a half
// and a third), optionally with the score accumulator initialised to -m (no subtraction in the softmax) and v_max3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dex_tts_amd/csrc -I include tools/attnw.hip -L dex_tts_amd/lib -ldexamd \
//         -Wl,-rpath,$PWD/dex_tts_amd/lib -o tools/attnw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <functional>
#include <algorithm>
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

using namespace dex;
using namespace dex::bf16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
union DFrag { uint4 u; lp8 v; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int HD = 128;

template <int NW, int OPT>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 3 : NW == 8 ? 2 : 3)) void attn_ringw_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    const int qt = min((int)blockIdx.x * NW + wave, ntiles - 1);
    const bool live_wave = (int)blockIdx.x * NW + wave < ntiles;
    const int q0 = qt * 32;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // the 16 one-KB pieces of a (K tile, V^T tile) pair are dealt to the waves: piece q < 8 = K piece q, else V piece q - 8
    auto dma = [&](int ktile, int kslot, int vtile, int vslot, bool with_v) __attribute__((always_inline)) {
        const long tk = min(ktile, t_hi - 1), tv = min(vtile, t_hi - 1);
#pragma unroll
        for (int q = 0; q < 16; q += NW) {
            const int pc = q + wave;
            if (pc < 8) __builtin_amdgcn_global_load_lds(Kg + tk * 512 + pc * 64, (lds_ptr)&kS[kslot][pc * 64], 16, 0, 0);
            else if (pc < 16 && with_v) __builtin_amdgcn_global_load_lds(Vg + tv * 512 + (pc - 8) * 64, (lds_ptr)&vS[vslot][(pc - 8) * 64], 16, 0, 0);
        }
    };
    dma(t_lo, 0, t_lo, 0, true); dma(t_lo + 1, 1, t_lo + 1, 1, true); dma(t_lo + 2, 2, 0, 0, false);
    DFrag qf[8];
    {
        const uint4* qp = Qg + (long)qt * 512;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks].u = qp[ks * 64];
    }
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = (OPT & 1) ? -1e30f : -INFINITY, l_run = 0.f;
    auto qk = [&](const uint4* kbuf, float init) __attribute__((always_inline)) -> f32x16 {
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = init;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { DFrag k0; k0.u = kbuf[ks * 64 + lane]; s0 = DEX_MFMA_LP(k0.v, qf[ks].v, s0, 0, 0, 0); }
        return s0;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // OPT & 1: the score accumulator starts at -m_ref, so the MFMA chain delivers s - m_ref and the softmax has no subtraction; the
    // first tile has no reference yet: it starts at 0 and the (then certain) rescale branch sets the reference
    float m_in = 0.f;                    // the reference the pending score tile was computed against
    f32x16 s = qk(kS[0], 0.f);
    lds_barrier();
    int s0 = 0, s1 = 1, s2 = 2;
    constexpr int CNT = (16 + NW - 1) / NW;            // pieces per wave and iteration (NW = 12: waves 0-3 two, the others one)
    for (int kt = t_lo; kt < t_hi; ++kt) {
        const int k0 = kt * 32;
        dma(kt + 3, s0, kt + 2, s2, true);
        const float m_next = (OPT & 1) ? (kt == t_lo ? 0.f : m_run) : 0.f;
        const f32x16 sn = qk(kS[s1], (OPT & 1) ? -m_next : 0.f);
        if (k0 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
        }
        float mx;
        if (OPT & 1) {
            mx = __builtin_fmaxf(__builtin_fmaxf(s[0], s[1]), s[2]);          // v_max3_f32 chains
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[r]), s[r + 1]);
            mx = fmaxf(mx, s[15]);
        } else {
            mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (OPT & 1) {
            // s holds (score - m_in); the running reference is m_run.  Normally m_in == m_run.
            const float rel = mx + (m_in - m_run);                       // tile maximum relative to the running reference
            if (__builtin_amdgcn_ballot_w64(rel > 8.f || m_in != m_run) != 0) {
                const float m_new = rel > 8.f ? m_run + rel : m_run;      // (first tile: m_run = -1e30 -> m_new = tile max)
                const float alpha = exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                const float shift = m_in - m_new;                         // bring the pending tile onto the new reference
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += shift;
                m_run = m_new;
            }
        } else if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        float psum = 0.f;
        if (OPT & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r]); psum += s[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); psum += s[r]; }
        }
        l_run += psum;
        const uint4* vcur = vS[s0];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pb;
            pb.u.x = pack2_lp(s[8 * k2 + 0], s[8 * k2 + 1]); pb.u.y = pack2_lp(s[8 * k2 + 2], s[8 * k2 + 3]);
            pb.u.z = pack2_lp(s[8 * k2 + 4], s[8 * k2 + 5]); pb.u.w = pack2_lp(s[8 * k2 + 6], s[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                DFrag vf; vf.u = vcur[(t * 2 + k2) * 64 + lane];
                o[t] = DEX_MFMA_LP(vf.v, pb.v, o[t], 0, 0, 0);
            }
        }
        // the group issued LAST iteration has landed; this iteration's flies on
        if (NW == 12) { if (wave < 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else if (CNT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        lds_barrier();
        s = sn; m_in = m_next;
        const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
    l_run += __shfl_xor(l_run, 32);
    if (live_wave && q0 + i < N) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        float* op = p.O + (long)sp * p.o_sstride + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) =
                    make_float4(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv, o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv);
        if (p.ml && hh == 0) {
            float* ml = p.ml + ((((long)sp * p.B + b) * 2 + h) * N + q0 + i) * 2;
            ml[0] = m_run; ml[1] = l_run;
        }
    }
}
template <int NW, int OPT>
static void launch_ringw(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + NW - 1) / NW, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL((attn_ringw_kernel<NW, OPT>), grid, dim3(64 * NW), 0, 0, p);
}

// ------------------------------------------------------------------------------------------------------------------
// q64: ONE wave per SIMD, 64 queries per wave = two 32-query tiles A and B that run HALF AN ITERATION APART, so that every MFMA of
// the wave has the softmax work of exactly one score of the OTHER tile in its shadow, placed by hand (source order + scheduling
// fences; the compiler's own order of the 460-register round-2 attempt ran at 0.135):
//   phase X(t): MFMAs  PV_B(t-1) [8], S_A(t+1) [8]      VALU  exp / sum / pack of tile t for A, tile maximum of B(t)
//   phase Y(t): MFMAs  PV_A(t)   [8], S_B(t+1) [8]      VALU  exp / sum / pack of tile t for B, tile maximum of A(t+1)
// The O accumulators (128 registers) live in the ACCUMULATION file (inline-asm MFMA, "+a"); the score accumulators start at -m (no
// subtraction in the softmax); the rescale decision taken from a tile maximum is applied half an iteration later, when the PV MFMAs
// that still use the old reference have been issued.  K ring 3 slots, V^T ring 4 slots (V(t-1) is still read in X(t)).
#define MFMA_ACC(acc, a, b) acc = DEX_MFMA_LP(a, b, acc, 0, 0, 0)
template <int VER>
__global__ __launch_bounds__(256, 1) void attn_q64h_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[4][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    const int qtA = min(((int)blockIdx.x * 4 + wave) * 2, ntiles - 1), qtB = min(qtA + 1, ntiles - 1);
    const bool liveA = ((int)blockIdx.x * 4 + wave) * 2 < ntiles, liveB = ((int)blockIdx.x * 4 + wave) * 2 + 1 < ntiles;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma_k = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, ntiles - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_global_load_lds(Kg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&kS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    auto dma_v = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, ntiles - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_global_load_lds(Vg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&vS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    // tile t lives in K slot t % 3 and V slot t % 4
    dma_k(0, 0); dma_v(0, 0); dma_k(1, 1); dma_v(1, 1); dma_k(2, 2);
    DFrag qa[8], qb[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { qa[ks].u = Qg[(long)qtA * 512 + ks * 64]; qb[ks].u = Qg[(long)qtB * 512 + ks * 64]; }
    f32x16 oa[4], ob[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oa[t][r] = 0.f; ob[t][r] = 0.f; }
    float mA, lA = 0.f, mB, lB = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto tile_max = [&](const f32x16& s) __attribute__((always_inline)) -> float {
        float mx = __builtin_fmaxf(__builtin_fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[r]), s[r + 1]);
        mx = fmaxf(mx, s[15]);
        return fmaxf(mx, __shfl_xor(mx, 32));
    };
    auto mask_tail = [&](f32x16& s, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
    };
    // prologue: S_A(0), S_B(0) with a zero start, their maxima become the first references
    f32x16 sA, sB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        DFrag k0; k0.u = kS[0][ks * 64 + lane];
        sA = DEX_MFMA_LP(k0.v, qa[ks].v, sA, 0, 0, 0);
        sB = DEX_MFMA_LP(k0.v, qb[ks].v, sB, 0, 0, 0);
    }
    if (32 > N) { mask_tail(sA, 0); mask_tail(sB, 0); }
    mA = tile_max(sA); mB = tile_max(sB);
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[r] -= mA; sB[r] -= mB; }
    bool overflow = false;                 // (experiment: fixed reference = first-tile maximum, no rescale in the loop)
    u32x4_t pA[2], pB[2];
    pB[0] = u32x4_t{0, 0, 0, 0}; pB[1] = pB[0];
    lds_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * 32;
        dma_k(kt + 3, (kt + 3) % 3); dma_v(kt + 2, (kt + 2) & 3);
        const uint4* kn = kS[(kt + 1) % 3];
        const uint4* vp = vS[(kt + 3) & 3];            // V(t-1)   (t = 0: any landed slot, P_B = 0)
        const uint4* vc = vS[kt & 3];                  // V(t)
        // ---- phase X(t)
        f32x16 nA;
#pragma unroll
        for (int r = 0; r < 16; ++r) nA[r] = -mA;
        // fragment j of the phase: j < 8: V(t-1)[tt*2+k2] (k2 = j >> 2, tt = j & 3), else K(t+1)[j - 8]; four at a time, one group ahead
        DFrag fr[2][4];
        auto fragX = [&](int j) __attribute__((always_inline)) -> uint4 { return j < 8 ? vp[(((j & 3) * 2) + (j >> 2)) * 64 + lane] : kn[(j - 8) * 64 + lane]; };
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[0][q].u = fragX(q);
        float mxB = -INFINITY;
        if (k0 + 32 > N) mask_tail(sB, k0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fr[(g + 1) & 1][q].u = fragX(4 * g + 4 + q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * g + q;
                if (j < 8) { const int k2 = j >> 2, tt = j & 3; MFMA_ACC(ob[tt], fr[g & 1][q].v, __builtin_bit_cast(lp8, pB[k2])); }
                else nA = DEX_MFMA_LP(fr[g & 1][q].v, qa[j - 8].v, nA, 0, 0, 0);
                // VALU shadow: score j of A(t)
                { float e = __builtin_amdgcn_exp2f(sA[j]); asm volatile("" : "+v"(e)); lA += e; sA[j] = e; }      // (pinned: the sinking pass otherwise moves it to its use in the next phase)
                if (j & 1) { unsigned d = pack2_lp(sA[j - 1], sA[j]); asm volatile("" : "+v"(d)); pA[j >> 3][(j >> 1) & 3] = d; }
                if (j & 1) mxB = __builtin_fmaxf(__builtin_fmaxf(mxB, sB[j - 1]), sB[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mxB = fmaxf(mxB, __shfl_xor(mxB, 32));
        overflow |= __builtin_amdgcn_ballot_w64(mxB > 100.f) != 0;
        // ---- phase Y(t)
        f32x16 nB;
#pragma unroll
        for (int r = 0; r < 16; ++r) nB[r] = -mB;
        auto fragY = [&](int j) __attribute__((always_inline)) -> uint4 { return j < 8 ? vc[(((j & 3) * 2) + (j >> 2)) * 64 + lane] : kn[(j - 8) * 64 + lane]; };
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[0][q].u = fragY(q);
        float mxA = -INFINITY;
        if (k0 + 64 > N) mask_tail(nA, k0 + 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fr[(g + 1) & 1][q].u = fragY(4 * g + 4 + q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * g + q;
                if (j < 8) { const int k2 = j >> 2, tt = j & 3; MFMA_ACC(oa[tt], fr[g & 1][q].v, __builtin_bit_cast(lp8, pA[k2])); }
                else nB = DEX_MFMA_LP(fr[g & 1][q].v, qb[j - 8].v, nB, 0, 0, 0);
                { float e = __builtin_amdgcn_exp2f(sB[j]); asm volatile("" : "+v"(e)); lB += e; sB[j] = e; }
                if (j & 1) { unsigned d = pack2_lp(sB[j - 1], sB[j]); asm volatile("" : "+v"(d)); pB[j >> 3][(j >> 1) & 3] = d; }
                if (j & 1) mxA = __builtin_fmaxf(__builtin_fmaxf(mxA, nA[j - 1]), nA[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mxA = fmaxf(mxA, __shfl_xor(mxA, 32));
        overflow |= __builtin_amdgcn_ballot_w64(mxA > 100.f) != 0;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        lds_barrier();
        sA = nA; sB = nB;
    }
    // tail: PV_B(ntiles - 1)
    {
        const uint4* vp = vS[(ntiles - 1) & 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k2 = j >> 2, tt = j & 3; DFrag f; f.u = vp[(tt * 2 + k2) * 64 + lane]; MFMA_ACC(ob[tt], f.v, __builtin_bit_cast(lp8, pB[k2])); }
    }
    lA += __shfl_xor(lA, 32); lB += __shfl_xor(lB, 32);
    if (overflow) { lA = __builtin_nanf(""); lB = lA; }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const bool live = which ? liveB : liveA;
        const int q0 = (which ? qtB : qtA) * 32;
        if (live && q0 + i < N) {
            const float inv = 1.f / (which ? lB : lA);
            float* op = p.O + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16& o = which ? ob[t] : oa[t];
                    *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) = make_float4(o[rq * 4 + 0] * inv, o[rq * 4 + 1] * inv, o[rq * 4 + 2] * inv, o[rq * 4 + 3] * inv);
                }
        }
    }
}
template <int VER>
static void launch_q64h(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + 7) / 8, 2, p.B);
    hipLaunchKernelGGL((attn_q64h_kernel<VER>), grid, dim3(256), 0, 0, p);
}

static double timeit(const char* name, int iters, double flops, std::function<void()> f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    printf("%-58s %9.2f us  %8.1f TF/s  (%.3f of 2.5 PF, %.3f of the 1.77 PF sustained)\n", name, us, flops / us * 1e-6, flops / us * 1e-6 / 2500.0, flops / us * 1e-6 / 1772.0);
    return us;
}
static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    struct AC { int B, N; };
    std::vector<AC> cases = {{32, 650}, {32, 1300}, {8, 2580}, {1, 5010}};
    const int only_case = argc > 1 ? atoi(argv[1]) : -1;
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        if (only_case >= 0 && (int)ci != only_case) continue;
        const AC c = cases[ci];
        const int Npad = (c.N + 31) / 32 * 32;
        const size_t el = (size_t)c.B * 2 * Npad * 128;
        unsigned short *q, *k, *v; hipMalloc(&q, el * 2); hipMalloc(&k, el * 2); hipMalloc(&v, el * 2);
        std::vector<unsigned short> hq(el), hk(el), hv(el);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
        for (size_t j = 0; j < el; ++j) { hq[j] = f2bf(rnd() * 0.35f); hk[j] = f2bf(rnd() * 0.9f); hv[j] = f2bf(rnd()); }
        hipMemcpy(q, hq.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), el * 2, hipMemcpyHostToDevice);
        const size_t on = (size_t)c.B * c.N * 256;
        float *O, *O2; hipMalloc(&O, on * 4); hipMalloc(&O2, on * 4);
        AttnDirectP a{q, k, v, c.N, Npad, c.B, O, (long)on, nullptr, 1, nullptr};
        AttnDirectP a2 = a; a2.O = O2;
        const double fl = 4.0 * c.B * c.N * (double)c.N * 256;
        printf("---- B=%d N=%d (%.1f GFLOP)\n", c.B, c.N, fl * 1e-9);
        timeit("shipped attn_direct (library)", 20, fl, [&] { dex::bf16::launch_attention_direct(a, 0); });
        auto check = [&](const char* what) {
            hipDeviceSynchronize();
            std::vector<float> r(on), g(on);
            hipMemcpy(r.data(), O, on * 4, hipMemcpyDeviceToHost); hipMemcpy(g.data(), O2, on * 4, hipMemcpyDeviceToHost);
            double mx = 0, ref = 0; for (size_t j = 0; j < on; ++j) { mx = std::max(mx, (double)fabsf(r[j] - g[j])); ref = std::max(ref, (double)fabsf(r[j])); }
            printf("      %s vs shipped: max|d| = %.3e (|O|max %.3f)\n", what, mx, ref);
        };
#define CAND(NW, OPT, label) hipMemset(O2, 0, on * 4); timeit(label, 20, fl, [&] { launch_ringw<NW, OPT>(a2); }); check(label);
        CAND(4, 0, "ringw: 4 waves, plain")
        CAND(4, 1, "ringw: 4 waves, acc = -m, max3")
        CAND(8, 0, "ringw: 8 waves, plain")
        CAND(8, 1, "ringw: 8 waves, acc = -m, max3")
        CAND(12, 0, "ringw: 12 waves, plain")
        CAND(12, 1, "ringw: 12 waves, acc = -m, max3")
        hipMemset(O2, 0, on * 4); timeit("q64h: 64 q per wave, hand-placed softmax shadows", 20, fl, [&] { launch_q64h<0>(a2); }); check("q64h");
        hipFree(q); hipFree(k); hipFree(v); hipFree(O); hipFree(O2);
    }
    return 0;
}
