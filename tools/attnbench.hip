// attnbench: development bench of the batch-regime DiT attention kernel (attention_direct.hip) on fragment-ordered bf16
// operands.  Links libdexamd.so for the shipped kernel (baseline + correctness reference) and carries candidate kernels
// with switchable pieces so that one gpurun call separates MFMA / softmax-VALU / LDS+barrier costs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dex_tts_amd/csrc -I include tools/attnbench.hip -L dex_tts_amd/lib -ldexamd \
//         -Wl,-rpath,$PWD/dex_tts_amd/lib -o tools/attnbench
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
constexpr int HD = 128;

static double timeit(const char* name, int iters, double flops, std::function<void()> f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    printf("%-58s %9.2f us  %8.1f TF/s  (%.3f of 2.5 PF)\n", name, us, flops / us * 1e-6, flops / us * 1e-6 / 2500.0);
    return us;
}

// ------------------------------------------------------------------------------------------------------------------
// Candidate: 4 waves x 32 queries per workgroup, K / V^T tiles through a 3-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: fragment-ordered tiles are contiguous, one wave instruction = 1 KB), ONE barrier per key
// tile, S^T(k+1) issued before softmax(k).
//   MODE 0: full   1: no softmax VALU (P = S)   2: no LDS refill (ring filled once; no DMA, no barrier in the loop)
//   MODE 3: MFMAs only (K / V fragments = the Q registers, no LDS reads, no softmax, no refill)   4: MODE 2 without softmax
template <int MODE, bool ONE_CHAIN, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_ring_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];          // [slot][ks*64 + lane]
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];          // [slot][(t*2+k2)*64 + lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    const int qt = min((int)blockIdx.x * 4 + wave, ntiles - 1);
    const bool live_wave = (int)blockIdx.x * 4 + wave < ntiles;
    const int q0 = qt * 32;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // wave w copies K-steps / V fragments 2w, 2w+1 of a tile (2 KB of each)
    auto dma_k = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, ntiles - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Kg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&kS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    auto dma_v = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, ntiles - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Vg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&vS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    dma_k(0, 0); dma_v(0, 0); dma_k(1, 1);
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
    float m_run = -INFINITY, l_run = 0.f;
    auto qk = [&](const uint4* kbuf) __attribute__((always_inline)) -> f32x16 {
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
        if constexpr (ONE_CHAIN) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { DFrag k0; if (MODE == 3) k0 = qf[(ks + 1) & 7]; else k0.u = kbuf[ks * 64 + lane]; s0 = DEX_MFMA_LP(k0.v, qf[ks].v, s0, 0, 0, 0); }
        } else {
            f32x16 s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2) {
                DFrag k0; k0.u = kbuf[ks * 64 + lane];
                DFrag k1; k1.u = kbuf[(ks + 1) * 64 + lane];
                s0 = DEX_MFMA_LP(k0.v, qf[ks].v, s0, 0, 0, 0);
                s1 = DEX_MFMA_LP(k1.v, qf[ks + 1].v, s1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] += s1[r];
        }
        return s0;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s = qk(kS[0]);
    int sk = 1, sv = 0;                 // ring slot of K(kt+1), V(kt)
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * 32;
        const int sk2 = sk == 2 ? 0 : sk + 1, sv1 = sv == 2 ? 0 : sv + 1;
        if (MODE < 2) { dma_k(kt + 2, sk2); dma_v(kt + 1, sv1); }
        const f32x16 sn = qk(kS[MODE >= 2 ? (kt & 1) : sk]);
        if (k0 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
        }
        if (MODE != 1 && MODE < 3) {
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
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
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); psum += s[r]; }
            l_run += psum;
        } else {
            l_run += 1.f;
        }
        const uint4* vcur = vS[MODE >= 2 ? 0 : sv];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pb;
            pb.u.x = pack2_lp(s[8 * k2 + 0], s[8 * k2 + 1]); pb.u.y = pack2_lp(s[8 * k2 + 2], s[8 * k2 + 3]);
            pb.u.z = pack2_lp(s[8 * k2 + 4], s[8 * k2 + 5]); pb.u.w = pack2_lp(s[8 * k2 + 6], s[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                DFrag vf; if (MODE == 3) vf = qf[t * 2 + k2]; else vf.u = vcur[(t * 2 + k2) * 64 + lane];
                o[t] = DEX_MFMA_LP(vf.v, pb.v, o[t], 0, 0, 0);
            }
        }
        if (MODE < 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces landed ...
            lds_barrier();                                         // ... and everyone's; everyone is done reading the old slots
        }
        s = sn; sk = sk2; sv = sv1;
    }
    if (p.dbg && tid == 0 && blockIdx.x == 1 && blockIdx.y == 0) {
        p.dbg[blockIdx.z * 2] = clock64() - c0; p.dbg[blockIdx.z * 2 + 1] = wall_clock64() - w0;
    }
    l_run += __shfl_xor(l_run, 32);
    if (live_wave && q0 + i < N) {
        const float inv = 1.f / l_run;
        float* op = p.O + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) =
                    make_float4(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv, o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv);
    }
}

template <int MODE, bool ONE_CHAIN, int MINW>
static void launch_ring(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + 3) / 4, 2, p.B);
    hipLaunchKernelGGL((attn_ring_kernel<MODE, ONE_CHAIN, MINW>), grid, dim3(256), 0, 0, p);
}


// ------------------------------------------------------------------------------------------------------------------
// Candidate V3: 64 keys per iteration (two 32-key sub-tiles).  K and V^T pairs live in separate 2-slot LDS rings filled by
// LDS-DMA; a slot is refilled right after its last reader, so every DMA group has a FULL iteration to land:
//   iteration kp:  S^T(kp+1) from Kslot[(kp+1)&1]  ||  softmax(kp)  ->  PV(kp) from Vslot[kp&1]
//                  -> vmcnt(0) [K(kp+2), V(kp+1): issued one iteration ago] -> ONE barrier -> issue K(kp+3), V(kp+2)
// NWQ waves x 32 queries per workgroup share the rings (64 KB).
template <int NWQ, int MINW>
__global__ __launch_bounds__(64 * NWQ, MINW) void attn_k64_kernel(const AttnDirectP p) {
    extern __shared__ __attribute__((aligned(16))) uint4 ring[];       // K slots: [0,1024) [1024,2048); V slots: [2048,3072) [3072,4096)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = p.N;
    const int ntiles = (N + 31) / 32, npairs = (ntiles + 1) / 2;
    const int qt = min((int)blockIdx.x * NWQ + wave, ntiles - 1);
    const bool live_wave = (int)blockIdx.x * NWQ + wave < ntiles;
    const int q0 = qt * 32;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int PER = 16 / NWQ;                 // 1-KB pieces of one 64-key K (or V) pair per wave
    auto dma = [&](const uint4* G, int pair, int slot_base) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int j = wave * PER + q;         // piece: sub-tile j/8, fragment j%8
            const long tile = min(pair * 2 + (j >> 3), ntiles - 1);
            __builtin_amdgcn_global_load_lds(G + tile * 512 + (j & 7) * 64, (lds_ptr)&ring[slot_base + j * 64], 16, 0, 0);
        }
    };
    dma(Kg, 0, 0); dma(Vg, 0, 2048); dma(Kg, 1, 1024); dma(Vg, 1, 3072);
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
    float m_run = -INFINITY, l_run = 0.f;
    auto qk2 = [&](const uint4* st, f32x16& sa, f32x16& sb) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            DFrag ka, kb; ka.u = st[ks * 64 + lane]; kb.u = st[512 + ks * 64 + lane];
            sa = DEX_MFMA_LP(ka.v, qf[ks].v, sa, 0, 0, 0);
            sb = DEX_MFMA_LP(kb.v, qf[ks].v, sb, 0, 0, 0);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sa, sb;
    qk2(&ring[0], sa, sb);
    lds_barrier();                                 // everyone has read K(0): its slot takes K(2)
    dma(Kg, 2, 0);
    for (int kp = 0; kp < npairs; ++kp) {
        const int k0 = kp * 64;
        f32x16 na, nb;
        qk2(&ring[((kp + 1) & 1) * 1024], na, nb);            // (past the end: a clamped re-read, never consumed)
        if (k0 + 64 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (k0 + kr >= N) sa[r] = -INFINITY;
                if (k0 + 32 + kr >= N) sb[r] = -INFINITY;
            }
        }
        float mx = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sb[0], sb[1]));
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(fmaxf(sa[r], sa[r + 1]), fmaxf(sb[r], sb[r + 1])));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = __builtin_amdgcn_exp2f(sa[r] - m_run); ps0 += sa[r]; sb[r] = __builtin_amdgcn_exp2f(sb[r] - m_run); ps1 += sb[r]; }
        l_run += ps0 + ps1;
        const uint4* vcur = &ring[2048 + (kp & 1) * 1024];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pa, pb;
            pa.u.x = pack2_lp(sa[8 * k2 + 0], sa[8 * k2 + 1]); pa.u.y = pack2_lp(sa[8 * k2 + 2], sa[8 * k2 + 3]);
            pa.u.z = pack2_lp(sa[8 * k2 + 4], sa[8 * k2 + 5]); pa.u.w = pack2_lp(sa[8 * k2 + 6], sa[8 * k2 + 7]);
            pb.u.x = pack2_lp(sb[8 * k2 + 0], sb[8 * k2 + 1]); pb.u.y = pack2_lp(sb[8 * k2 + 2], sb[8 * k2 + 3]);
            pb.u.z = pack2_lp(sb[8 * k2 + 4], sb[8 * k2 + 5]); pb.u.w = pack2_lp(sb[8 * k2 + 6], sb[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                DFrag va, vb; va.u = vcur[(t * 2 + k2) * 64 + lane]; vb.u = vcur[512 + (t * 2 + k2) * 64 + lane];
                o[t] = DEX_MFMA_LP(va.v, pa.v, o[t], 0, 0, 0);
                o[t] = DEX_MFMA_LP(vb.v, pb.v, o[t], 0, 0, 0);
            }
        }
        // K(kp+2) and V(kp+1) were issued one iteration ago; everyone has now read K(kp+1) and V(kp)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (kp + 1 < npairs) { dma(Kg, kp + 3, ((kp + 1) & 1) * 1024); dma(Vg, kp + 2, 2048 + (kp & 1) * 1024); }
        sa = na; sb = nb;
    }
    l_run += __shfl_xor(l_run, 32);
    if (live_wave && q0 + i < N) {
        const float inv = 1.f / l_run;
        float* op = p.O + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) =
                    make_float4(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv, o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv);
    }
}
template <int NWQ, int MINW>
static void launch_k64(const AttnDirectP& p) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_k64_kernel<NWQ, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
    dim3 grid(((p.N + 31) / 32 + NWQ - 1) / NWQ, 2, p.B);
    hipLaunchKernelGGL((attn_k64_kernel<NWQ, MINW>), grid, dim3(64 * NWQ), 65536, 0, p);
}


// ------------------------------------------------------------------------------------------------------------------
// Candidate V4: ONE wave per SIMD, 64 queries per wave (two 32-query tiles A and B share every K / V^T fragment read):
// half the LDS reads, DMA bytes and barriers per MFMA; the two tiles give the in-order wave independent work to put
// beside each MFMA (softmax of A next to the MFMAs of B).  4 waves = 256 queries per workgroup; K / V^T tiles of 32 keys
// through 3-slot LDS rings filled by LDS-DMA one iteration ahead, ONE barrier per key tile.
//   SCHED: interleave hint pattern (0 = compiler's own order)
template <int SCHED>
__global__ __launch_bounds__(256, 1) void attn_q64_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];
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
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Kg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&kS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    auto dma_v = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, ntiles - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Vg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&vS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    dma_k(0, 0); dma_v(0, 0); dma_k(1, 1);
    DFrag qa[8], qb[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { qa[ks].u = Qg[(long)qtA * 512 + ks * 64]; qb[ks].u = Qg[(long)qtB * 512 + ks * 64]; }
    f32x16 oa[4], ob[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oa[t][r] = 0.f; ob[t][r] = 0.f; }
    float mA = -INFINITY, lA = 0.f, mB = -INFINITY, lB = 0.f;
    auto qk = [&](const uint4* kbuf, f32x16& sa, f32x16& sb) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            DFrag k0; k0.u = kbuf[ks * 64 + lane];
            sa = DEX_MFMA_LP(k0.v, qa[ks].v, sa, 0, 0, 0);
            sb = DEX_MFMA_LP(k0.v, qb[ks].v, sb, 0, 0, 0);
        }
    };
    auto softmax = [&](f32x16& s, float& m_run, float& l_run, f32x16 (&o)[4], int k0) __attribute__((always_inline)) {
        if (k0 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); psum += s[r]; }
        l_run += psum;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sa, sb;
    qk(kS[0], sa, sb);
    int sk = 1, sv = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * 32;
        const int sk2 = sk == 2 ? 0 : sk + 1, sv1 = sv == 2 ? 0 : sv + 1;
        dma_k(kt + 2, sk2); dma_v(kt + 1, sv1);
        f32x16 na, nb;
        qk(kS[sk], na, nb);                       // 16 MFMAs: S^T(kt+1) of both query tiles
        softmax(sa, mA, lA, oa, k0);              // VALU, independent of the MFMAs above
        softmax(sb, mB, lB, ob, k0);
        const uint4* vcur = vS[sv];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pa, pb;
            pa.u.x = pack2_lp(sa[8 * k2 + 0], sa[8 * k2 + 1]); pa.u.y = pack2_lp(sa[8 * k2 + 2], sa[8 * k2 + 3]);
            pa.u.z = pack2_lp(sa[8 * k2 + 4], sa[8 * k2 + 5]); pa.u.w = pack2_lp(sa[8 * k2 + 6], sa[8 * k2 + 7]);
            pb.u.x = pack2_lp(sb[8 * k2 + 0], sb[8 * k2 + 1]); pb.u.y = pack2_lp(sb[8 * k2 + 2], sb[8 * k2 + 3]);
            pb.u.z = pack2_lp(sb[8 * k2 + 4], sb[8 * k2 + 5]); pb.u.w = pack2_lp(sb[8 * k2 + 6], sb[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                DFrag vf; vf.u = vcur[(t * 2 + k2) * 64 + lane];
                oa[t] = DEX_MFMA_LP(vf.v, pa.v, oa[t], 0, 0, 0);
                ob[t] = DEX_MFMA_LP(vf.v, pb.v, ob[t], 0, 0, 0);
            }
        }
        if constexpr (SCHED == 1) {
            // 16 S^T MFMAs with the softmax VALU in their shadows, then 16 PV MFMAs with the pack / address VALU
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // 6 VALU
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        sa = na; sb = nb; sk = sk2; sv = sv1;
    }
    lA += __shfl_xor(lA, 32); lB += __shfl_xor(lB, 32);
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
template <int SCHED>
static void launch_q64(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + 7) / 8, 2, p.B);
    hipLaunchKernelGGL((attn_q64_kernel<SCHED>), grid, dim3(256), 0, 0, p);
}


// ------------------------------------------------------------------------------------------------------------------
// Candidate V5 = the ring kernel with the DMA TWO key tiles ahead (same 3 slots: K(kt+3) goes into the slot K(kt) left
// after iteration kt-1, V(kt+2) into V(kt-1)'s) and an optional key split (blockIdx.z = b * ksplit + sp: split sp walks
// key tiles [t_lo, t_hi) and writes its normalised O and (m, l), merged by the row chain).
template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_ring2_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    const int qt = min((int)blockIdx.x * 4 + wave, ntiles - 1);
    const bool live_wave = (int)blockIdx.x * 4 + wave < ntiles;
    const int q0 = qt * 32;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma_k = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, t_hi - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Kg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&kS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    auto dma_v = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, t_hi - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Vg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&vS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    // ring slot of tile t = (t - t_lo) % 3
    dma_k(t_lo, 0); dma_v(t_lo, 0); dma_k(t_lo + 1, 1); dma_v(t_lo + 1, 1); dma_k(t_lo + 2, 2);
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
    float m_run = -INFINITY, l_run = 0.f;
    auto qk = [&](const uint4* kbuf) __attribute__((always_inline)) -> f32x16 {
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { DFrag k0; k0.u = kbuf[ks * 64 + lane]; s0 = DEX_MFMA_LP(k0.v, qf[ks].v, s0, 0, 0, 0); }
        return s0;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s = qk(kS[0]);
    lds_barrier();                      // K(t_lo) has been read by everyone: its slot may take K(t_lo + 3)
    int s0 = 0, s1 = 1, s2 = 2;         // slots of tiles kt, kt+1, kt+2 (relative): K(kt+1) in s1, V(kt) in s0
    for (int kt = t_lo; kt < t_hi; ++kt) {
        const int k0 = kt * 32;
        dma_k(kt + 3, s0); dma_v(kt + 2, s2);          // K(kt) was read last iteration, V(kt-1) too (slot s2 == slot of tile kt-1)
        const f32x16 sn = qk(kS[s1]);
        if (k0 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); psum += s[r]; }
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
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the group issued LAST iteration (K(kt+2), V(kt+1)) landed; this one's flies on
        lds_barrier();
        s = sn;
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
template <int MINW>
static void launch_ring2(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + 3) / 4, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL((attn_ring2_kernel<MINW>), grid, dim3(256), 0, 0, p);
}


// ------------------------------------------------------------------------------------------------------------------
// Candidate V6 = ring2 with a branch-free main body: the lazy-rescale test for tile kt+1 runs at the END of iteration kt
// (on the finished S^T(kt+1)), the key mask exists only in a peeled last iteration, so S^T MFMAs, softmax VALU and PV
// MFMAs of one iteration are ONE basic block the scheduler can interleave (SCHED = 1 adds explicit group hints).
template <int MINW, int SCHED>
__global__ __launch_bounds__(256, MINW) void attn_ring3_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    const int qt = min((int)blockIdx.x * 4 + wave, ntiles - 1);
    const bool live_wave = (int)blockIdx.x * 4 + wave < ntiles;
    const int q0 = qt * 32;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma_k = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, t_hi - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Kg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&kS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    auto dma_v = [&](int tile, int slot) __attribute__((always_inline)) {
        const long t = min(tile, t_hi - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(Vg + t * 512 + (2 * wave + j) * 64, (lds_ptr)&vS[slot][(2 * wave + j) * 64], 16, 0, 0);
    };
    dma_k(t_lo, 0); dma_v(t_lo, 0); dma_k(t_lo + 1, 1); dma_v(t_lo + 1, 1); dma_k(t_lo + 2, 2);
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
    float m_run = -INFINITY, l_run = 0.f;
    auto qk = [&](const uint4* kbuf) __attribute__((always_inline)) -> f32x16 {
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { DFrag k0; k0.u = kbuf[ks * 64 + lane]; s0 = DEX_MFMA_LP(k0.v, qf[ks].v, s0, 0, 0, 0); }
        return s0;
    };
    auto mask_tail = [&](f32x16& s, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
    };
    // reference maximum for the scores in s: moves (rarely) when some query's maximum exceeds it by more than 2^8
    auto fix_max = [&](const f32x16& s) __attribute__((always_inline)) {
        float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s = qk(kS[0]);
    if (t_lo * 32 + 32 > N) mask_tail(s, t_lo * 32);
    fix_max(s);
    lds_barrier();
    int s0 = 0, s1 = 1, s2 = 2;
    for (int kt = t_lo; kt < t_hi; ++kt) {
        dma_k(kt + 3, s0); dma_v(kt + 2, s2);
        // ---- one basic block: S^T(kt+1) MFMAs | exp / sum / pack of tile kt | PV(kt) MFMAs
        f32x16 sn = qk(kS[s1]);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_run); psum += s[r]; }
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
        if constexpr (SCHED == 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // K fragment read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // S^T MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);      // sub / exp / add of the current tile
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // V^T fragment read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // PV MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
        // ---- end of the block: next tile's mask (last tile only) and reference maximum
        if ((kt + 1) * 32 + 32 > N) mask_tail(sn, (kt + 1) * 32);
        fix_max(sn);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        lds_barrier();
        s = sn;
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
template <int MINW, int SCHED>
static void launch_ring3(const AttnDirectP& p) {
    dim3 grid(((p.N + 31) / 32 + 3) / 4, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL((attn_ring3_kernel<MINW, SCHED>), grid, dim3(256), 0, 0, p);
}

static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    struct AC { int B, N; };
    std::vector<AC> cases = {{32, 650}, {32, 1300}, {8, 2580}, {2, 5010}, {1, 5010}, {1, 2580}};
    const int only_case = argc > 1 ? atoi(argv[1]) : -1;       // attnbench [case index] [profile: 1 = one launch of each mode, no timing loops]
    const bool prof = argc > 2 && atoi(argv[2]) == 1;
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        if (only_case >= 0 && (int)ci != only_case) continue;
        const AC c = cases[ci];
        const int Npad = (c.N + 31) / 32 * 32;
        const size_t el = (size_t)c.B * 2 * Npad * 128;
        unsigned short *q, *k, *v; hipMalloc(&q, el * 2); hipMalloc(&k, el * 2); hipMalloc(&v, el * 2);
        std::vector<unsigned short> hq(el), hk(el), hv(el);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
        for (size_t j = 0; j < el; ++j) { hq[j] = f2bf(rnd() * 0.35f); hk[j] = f2bf(rnd() * 0.9f); hv[j] = f2bf(rnd()); }   // scores: std ~ 2 (log2 domain)
        hipMemcpy(q, hq.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), el * 2, hipMemcpyHostToDevice);
        const size_t on = (size_t)c.B * c.N * 256;
        float *O, *O2; hipMalloc(&O, on * 4); hipMalloc(&O2, on * 4);
        AttnDirectP a{q, k, v, c.N, Npad, c.B, O, (long)on, nullptr, 1, nullptr};
        AttnDirectP a2 = a; a2.O = O2;
        const double fl = 4.0 * c.B * c.N * (double)c.N * 256;
        char nm[96];
        printf("---- B=%d N=%d (%.1f GFLOP)\n", c.B, c.N, fl * 1e-9);
        long long* dbg; hipMalloc(&dbg, 1024); hipMemset(dbg, 0, 1024);
        auto clk = [&](const char* what, std::function<void(const AttnDirectP&)> launch) {
            AttnDirectP ad = a2; ad.dbg = dbg;
            for (int r = 0; r < 5; ++r) launch(ad);
            hipDeviceSynchronize();
            long long h[64]; hipMemcpy(h, dbg, 512, hipMemcpyDeviceToHost);
            double cy = 0, wl = 0; int n = 0;
            for (int z = 0; z < std::min(c.B, 32); ++z) if (h[2 * z + 1] > 0) { cy += h[2 * z]; wl += h[2 * z + 1]; ++n; }
            printf("      %-40s key loop of one wave: %.0f shader cycles, %.2f us -> %.2f GHz; %.0f cycles per key tile (MFMA floor 512)\n", what, cy / n, wl / n * 0.01,
                   cy / (wl * 10.0), cy / n / ((c.N + 31) / 32));
        };
        if (prof) {
            launch_ring<0, true, 3>(a2); launch_ring<1, true, 3>(a2); launch_ring<2, true, 3>(a2); launch_ring<3, true, 3>(a2); launch_ring<4, true, 3>(a2);
            dex::bf16::launch_attention_direct(a, 0);
            hipDeviceSynchronize();
            continue;
        }
        snprintf(nm, 96, "shipped attn_direct (library)"); timeit(nm, 20, fl, [&] { dex::bf16::launch_attention_direct(a, 0); });
        auto check = [&](const char* what) {
            hipDeviceSynchronize();
            std::vector<float> r(on), g(on);
            hipMemcpy(r.data(), O, on * 4, hipMemcpyDeviceToHost); hipMemcpy(g.data(), O2, on * 4, hipMemcpyDeviceToHost);
            double mx = 0, ref = 0; for (size_t j = 0; j < on; ++j) { mx = std::max(mx, (double)fabsf(r[j] - g[j])); ref = std::max(ref, (double)fabsf(r[j])); }
            printf("      %s vs shipped: max|d| = %.3e (|O|max %.3f)\n", what, mx, ref);
        };
        hipMemset(O2, 0, on * 4);
        timeit("ring full, one S chain,  >=2 waves/SIMD", 20, fl, [&] { launch_ring<0, true, 2>(a2); }); check("ring<0,true,2>");
        hipMemset(O2, 0, on * 4);
        timeit("ring full, one S chain,  >=3 waves/SIMD", 20, fl, [&] { launch_ring<0, true, 3>(a2); }); check("ring<0,true,3>");
        hipMemset(O2, 0, on * 4);
        timeit("k64: 4 waves x 32 q, 64 keys/iter, >=2 waves/SIMD", 20, fl, [&] { launch_k64<4, 2>(a2); }); check("k64<4,2>");
        hipMemset(O2, 0, on * 4);
        timeit("k64: 8 waves x 32 q, 64 keys/iter, 2 waves/SIMD", 20, fl, [&] { launch_k64<8, 2>(a2); }); check("k64<8,2>");
        hipMemset(O2, 0, on * 4);
        timeit("ring2: DMA two tiles ahead,       >=3 waves/SIMD", 20, fl, [&] { launch_ring2<3>(a2); }); check("ring2<3>");
        hipMemset(O2, 0, on * 4);
        timeit("ring3: branch-free body, compiler order   >=3", 20, fl, [&] { launch_ring3<3, 0>(a2); }); check("ring3<3,0>");
        hipMemset(O2, 0, on * 4);
        timeit("ring3: branch-free body, sched groups     >=3", 20, fl, [&] { launch_ring3<3, 1>(a2); }); check("ring3<3,1>");
        hipMemset(O2, 0, on * 4);
        timeit("ring3: branch-free body, sched groups     >=2", 20, fl, [&] { launch_ring3<2, 1>(a2); }); check("ring3<2,1>");
        for (int ks : {2, 4}) {
            float *Os, *ml; hipMalloc(&Os, on * 4 * ks); hipMalloc(&ml, (size_t)ks * c.B * 2 * c.N * 2 * 4);
            AttnDirectP as = a; as.O = Os; as.ksplit = ks; as.o_sstride = (long)on; as.ml = ml;
            snprintf(nm, 96, "ring2: key split %d (partials merged by the consumer)", ks);
            timeit(nm, 20, fl, [&] { launch_ring2<3>(as); });
            hipDeviceSynchronize();
            std::vector<float> ho((size_t)on * ks), hml((size_t)ks * c.B * 2 * c.N * 2), r(on);
            hipMemcpy(ho.data(), Os, ho.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hml.data(), ml, hml.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(r.data(), O, on * 4, hipMemcpyDeviceToHost);
            double mxd = 0;
            for (int bb = 0; bb < c.B; ++bb) for (int n = 0; n < c.N; n += 7) for (int hd = 0; hd < 2; ++hd) {
                float M = -INFINITY; for (int sidx = 0; sidx < ks; ++sidx) M = fmaxf(M, hml[((((size_t)sidx * c.B + bb) * 2 + hd) * c.N + n) * 2]);
                double wsum = 0; std::vector<double> w(ks);
                for (int sidx = 0; sidx < ks; ++sidx) { const float* q2 = &hml[((((size_t)sidx * c.B + bb) * 2 + hd) * c.N + n) * 2]; w[sidx] = q2[1] * exp2(q2[0] - M); wsum += w[sidx]; }
                for (int d = 0; d < 128; d += 5) {
                    double acc = 0; for (int sidx = 0; sidx < ks; ++sidx) acc += w[sidx] * ho[(size_t)sidx * on + ((size_t)bb * c.N + n) * 256 + hd * 128 + d];
                    mxd = std::max(mxd, fabs(acc / wsum - r[((size_t)bb * c.N + n) * 256 + hd * 128 + d]));
                }
            }
            printf("      ring2 ksplit=%d merged vs shipped: max|d| = %.3e\n", ks, mxd);
            hipFree(Os); hipFree(ml);
        }
        hipMemset(O2, 0, on * 4);
        timeit("q64: 1 wave/SIMD, 64 q per wave, compiler order", 20, fl, [&] { launch_q64<0>(a2); }); check("q64<0>");
        hipMemset(O2, 0, on * 4);
        timeit("q64: 1 wave/SIMD, 64 q per wave, sched groups", 20, fl, [&] { launch_q64<1>(a2); }); check("q64<1>");
        timeit("  .. no softmax VALU (P = S)        >=3", 20, fl, [&] { launch_ring<1, true, 3>(a2); });
        timeit("  .. no LDS refill / barrier        >=3", 20, fl, [&] { launch_ring<2, true, 3>(a2); });
        timeit("  .. no refill, no softmax          >=3", 20, fl, [&] { launch_ring<4, true, 3>(a2); });
        timeit("  .. MFMAs only (no LDS reads)      >=3", 20, fl, [&] { launch_ring<3, true, 3>(a2); });
        timeit("  .. MFMAs only (no LDS reads)      >=2", 20, fl, [&] { launch_ring<3, true, 2>(a2); });
        clk("full", [](const AttnDirectP& q) { launch_ring<0, true, 3>(q); });
        clk("no softmax", [](const AttnDirectP& q) { launch_ring<1, true, 3>(q); });
        clk("no refill", [](const AttnDirectP& q) { launch_ring<2, true, 3>(q); });
        clk("no refill, no softmax", [](const AttnDirectP& q) { launch_ring<4, true, 3>(q); });
        clk("MFMA only", [](const AttnDirectP& q) { launch_ring<3, true, 3>(q); });
        hipFree(dbg);
        hipFree(q); hipFree(k); hipFree(v); hipFree(O); hipFree(O2);
    }
    return 0;
}
