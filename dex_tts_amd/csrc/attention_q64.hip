// attention_q64.hip — batch / long-form softmax attention of the DiT blocks (timm Attention core, reference model/dit.py:276,
// GeDEX-TTS/model/dit.py:268-290 is the DiTBlock around it) on fragment-ordered 16-bit operands (see attention_direct.hip for
// the layouts the row chain writes): the round-4 form with 64 QUERIES PER WAVE.
//
//   workgroup = 4 waves, ONE wave per SIMD, each wave owns 64 queries (two 32-query blocks A and B) and the whole 512-entry
//   register file: O^T (2 x 4 x 16 = 128), Q (2 x 8 fragments = 64) and the K / V^T fragment staging (2 x 4 fragments = 32) live
//   in the ACCUMULATION file (inline-asm MFMAs with "a" operands; the fragments are ds_read straight into it), the score tiles
//   (2 x 64), packed P (32), the -m accumulator seeds (32) and the softmax state in the arch VGPRs.
//   Every K / V^T fragment read from LDS feeds TWO MFMAs (blocks A and B): 0.5 ds_read_b128 per MFMA (the 32-query forms: 1).
//   Key tiles are 64 keys (two of the producer's 32-key tiles = 16 KB of K + 16 KB of V^T), brought in by LDS-DMA
//   (buffer_load ... lds, 1 KB per wave instruction, 4 + 4 pieces per wave and tile) into a 4-slot K ring and a 3-slot V^T ring,
//   two tiles ahead; ONE barrier per 64-key tile = per 64 MFMAs of a wave.
//   Per tile a wave runs two phases of 32 MFMAs, software-pipelined over tiles:
//     phase A(i): S(i+1) = K(i+1) Q^T - m   (accumulators SEEDED with -m: no subtraction in the softmax)
//                 VALU shadow: second half of the exp2 / row sums of tile i, all 32 packs of P(i)
//     phase B(i): O^T += V^T(i) P(i)^T
//                 VALU shadow: maxima of S(i+1), the (rare) reference-maximum move, first half of the exp2 / row sums of tile i+1
//   with the fillers of every MFMA gap written out by hand (<= 5 single-issue instructions per 32-cycle gap, fenced with
//   sched_barrier so they stay where they are written).  The reference maximum moves lazily (threshold 2^8, scores are in
//   the log2 domain as the producer folds log2 e into q), exactly as in the 32-query kernels; the move rescales O^T in the
//   accumulation file after the tile's PV MFMAs.
//   Work units (batch element, head, group of 4 x 64 queries, key split) are walked by persistent workgroups; unit u runs on
//   XCD u % 8 and all units of one (element, head) share an XCD (its K and V^T cross the fabric once).
// Output: normalised O per key split (+ (m, l) for the consumer's merge), the format attn_direct_ring_kernel writes.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int HD = 128;
constexpr int KSLOTS = 4, VSLOTS = 3, TILE_BYTES = 16384;
constexpr int V_RING = KSLOTS * TILE_BYTES;
constexpr int LDS_BYTES = (KSLOTS + VSLOTS) * TILE_BYTES;

#ifdef DEX_LP_F16
#define Q64_MFMA "v_mfma_f32_32x32x16_f16"
#else
#define Q64_MFMA "v_mfma_f32_32x32x16_bf16"
#endif

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); }
}

// S^T tile MFMAs: D (and the seed C) in arch VGPRs, K fragment (A) and Q fragment (B) in the accumulation file
__device__ __forceinline__ void mfma_s0(f32x16& d, const u32x4& k, const u32x4& q) { asm volatile(Q64_MFMA " %0, %1, %2, 0" : "=&v"(d) : "a"(k), "a"(q)); }
__device__ __forceinline__ void mfma_sc(f32x16& d, const u32x4& k, const u32x4& q, const f32x16& c) { asm volatile(Q64_MFMA " %0, %1, %2, %3" : "=&v"(d) : "a"(k), "a"(q), "v"(c)); }
__device__ __forceinline__ void mfma_s(f32x16& d, const u32x4& k, const u32x4& q) { asm volatile(Q64_MFMA " %0, %1, %2, %0" : "+v"(d) : "a"(k), "a"(q)); }
// O^T MFMAs: accumulator and V^T fragment (A) in the accumulation file, P (B) in arch VGPRs
__device__ __forceinline__ void mfma_o(f32x16& o, const u32x4& v, const u32x4& p) { asm volatile(Q64_MFMA " %0, %1, %2, %0" : "+a"(o) : "a"(v), "v"(p)); }
// one fragment (16 B per lane) from LDS into the accumulation file; NOT counted by the compiler: q64_lgkm0() before the first use
#ifdef Q64_NO_LDS
template <int OFF> __device__ __forceinline__ void lds_read_a(u32x4& f, unsigned addr) { asm volatile("" : "=a"(f) : "v"(addr)); }
#else
template <int OFF> __device__ __forceinline__ void lds_read_a(u32x4& f, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(f) : "v"(addr), "n"(OFF)); }
#endif
__device__ __forceinline__ void q64_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The shadow work of an MFMA gap is VOLATILE asm, one instruction per statement: volatile statements keep their order among
// themselves (the MFMAs are volatile too), so every filler stays in the gap it is written in - plain C++ arithmetic was sunk /
// hoisted across the MFMAs by the IR passes whatever the scheduling fences said (12 v_exp_f32 in one gap, none in the next 19).
__device__ __forceinline__ float max3(float a, float b, float c) { float r; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max2(float a, float b) { float r; asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float exp2_v(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ void add_v(float& acc, float x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
__device__ __forceinline__ unsigned pack_v(float lo, float hi) {
    unsigned r;
#ifdef DEX_LP_F16
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#else
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
}
#ifdef DEX_LP_F16
#define Q64_PK "v_cvt_pk_f16_f32"
#else
#define Q64_PK "v_cvt_pk_bf16_f32"
#endif
// ONE statement per gap (hipcc pads every asm statement whose output the next instruction reads with an s_nop: one statement = none)
// gap of phase A: e = exp2(s) in place, one pack of two finished values, l += e   (the pack separates v_exp_f32 from its reader:
// gfx940 trans forwarding hazard, 1 wait state)
__device__ __forceinline__ void gap_exp_pack_sum(float& s, float& l, unsigned& w, float lo, float hi) {
    asm volatile("v_exp_f32 %0, %0\n\t" Q64_PK " %2, %3, %4\n\tv_add_f32 %1, %1, %0" : "+v"(s), "+v"(l), "=v"(w) : "v"(lo), "v"(hi));
}
// last gap of phase A: the pack takes the value exponentiated in this very gap as its second input
__device__ __forceinline__ void gap_exp_sum_pack(float& s, float& l, unsigned& w, float lo) {
    asm volatile("v_exp_f32 %0, %0\n\ts_nop 0\n\tv_add_f32 %1, %1, %0\n\t" Q64_PK " %2, %3, %0" : "+v"(s), "+v"(l), "=v"(w) : "v"(lo));
}
// gap of phase B: two exponentials in place + their row sums
__device__ __forceinline__ void gap_exp2_sum2(float& s0, float& s1, float& l) {
    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %2, %2, %1" : "+v"(s0), "+v"(s1), "+v"(l));
}
// four independent running maxima, first / later step
__device__ __forceinline__ void gap_max_first(float& c0, float& c1, float& c2, float& c3, float a0, float a1, float a2, float b0, float b1, float b2,
                                              float d0, float d1, float d2, float e0, float e1, float e2) {
    asm volatile("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2), "v"(d0), "v"(d1), "v"(d2), "v"(e0), "v"(e1), "v"(e2));
}
__device__ __forceinline__ void gap_max_next(float& c0, float& c1, float& c2, float& c3, float a0, float a1, float b0, float b1, float d0, float d1, float e0, float e1) {
    asm volatile("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9\n\tv_max3_f32 %3, %3, %10, %11"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(d0), "v"(d1), "v"(e0), "v"(e1));
}
// the two block maxima from the four chains + the four values the chains left out
__device__ __forceinline__ void gap_max_last(float& mxA, float& mxB, float c0, float c1, float c2, float c3, float a, float b, float d, float e) {
    asm volatile("v_max3_f32 %0, %2, %3, %6\n\tv_max3_f32 %1, %4, %5, %8\n\tv_max_f32 %0, %0, %7\n\tv_max_f32 %1, %1, %9"
                 : "=&v"(mxA), "=&v"(mxB) : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a), "v"(b), "v"(d), "v"(e));
}
// max over the two 32-lane halves (the two key halves of a query's column), result in every lane
__device__ __forceinline__ float xhalf_max(float x) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// wait states between the last MFMA of a chain and a non-MFMA reader / writer of its result (inline-asm MFMAs are opaque to hipcc)
__device__ __forceinline__ void fence_v(f32x16& a, f32x16& b, f32x16& c, f32x16& d) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void fence_a(f32x16& a, f32x16& b, f32x16& c, f32x16& d) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a), "+a"(b), "+a"(c), "+a"(d)); }
}  // namespace

#ifdef Q64_STAMP
#define Q64_T(k) do { if (p.dbg) tst[k] = __builtin_readcyclecounter(); } while (0)
#else
#define Q64_T(k) do { } while (0)
#endif

// ng = groups of 256 queries per (element, head); nunits = 2 B ng ksplit; xcd_mode: unit u -> XCD u % 8 owns (element, head) pairs
__global__ __launch_bounds__(256, 1) void attn_q64_kernel(const AttnDirectP p, const int ng, const int nunits, const int xcd_mode) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char q64_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int N = p.N, ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nt32 = (N + 31) >> 5, nT = (nt32 + 1) >> 1;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)q64_smem;
    const unsigned lane16 = lds_base + lane * 16;
    const long bh_bytes = (long)p.Npad * (HD * 2);            // one (element, head) operand: Npad rows x 128 x 16 bit
#ifdef Q64_STAMP
    long long tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        // ---- which (element, head, query group, key split)
        int bh, g, sp;
        {
            const int per = ng * ks;
            int u = unit;
            if (xcd_mode) { const int x = u & 7, s = u >> 3; bh = x + 8 * (s / per); u = s % per; }
            else { bh = u / per; u = u % per; }
            g = u / ks; sp = u % ks;
        }
        const int b = bh >> 1, h = bh & 1;
        const int T_lo = (int)((long)nT * sp / ks), T_hi = (int)((long)nT * (sp + 1) / ks);
        const int nt = T_hi - T_lo;
        const int q64 = g * 4 + wave;                          // this wave's 64-query block
        const bool liveA = q64 * 2 < nt32, liveB = q64 * 2 + 1 < nt32;
        const int qtA = min(q64 * 2, nt32 - 1), qtB = min(q64 * 2 + 1, nt32 - 1);
        Q64_T(0);

        const unsigned char* Kb = reinterpret_cast<const unsigned char*>(p.Kh) + bh * bh_bytes;
        const unsigned char* Vb = reinterpret_cast<const unsigned char*>(p.Vt) + bh * bh_bytes;
        const auto rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Kb), 0, (int)bh_bytes, 0x00020000);
        const auto rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Vb), 0, (int)bh_bytes, 0x00020000);
        // piece j (0..3) of this wave for 64-key tile T (absolute): 32-key tile 2T + (wave >> 1), 1-KB pieces (wave & 1) * 4 + j
        auto dma_k = [&](int T, int j) __attribute__((always_inline)) {
            const int t32 = min(2 * T + (wave >> 1), nt32 - 1), pc = (wave & 1) * 4 + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (lds_ptr)(q64_smem + (T & 3) * TILE_BYTES + (wave * 4 + j) * 1024), 16, lane * 16,
                                                     t32 * 8192 + pc * 1024, 0, 0);
        };
        auto dma_v = [&](int T, int vslot, int j) __attribute__((always_inline)) {
            const int t32 = min(2 * T + (wave >> 1), nt32 - 1), pc = (wave & 1) * 4 + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_ptr)(q64_smem + V_RING + vslot * TILE_BYTES + (wave * 4 + j) * 1024), 16, lane * 16,
                                                     t32 * 8192 + pc * 1024, 0, 0);
        };
        // K slot of tile T = T & 3 (absolute index); V^T slot of tile T = (T - T_lo) % 3, tracked by rotation
        // ---- prologue: K(0), Q, V(0), K(1), V(1), K(2), K(3)   (tiles beyond the split's range are simply not fetched)
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_k(T_lo, j);
        u32x4 QA[8], QB[8];
        {
            const uint4* Qg = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.Qh) + bh * bh_bytes) + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint4 a = Qg[(long)qtA * 512 + s * 64], c = Qg[(long)qtB * 512 + s * 64];
                QA[s] = u32x4{a.x, a.y, a.z, a.w}; QB[s] = u32x4{c.x, c.y, c.z, c.w};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_v(T_lo, 0, j);
        if (nt > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_k(T_lo + 1, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_v(T_lo + 1, 1, j);
        }
        if (nt > 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_k(T_lo + 2, j);
        }
        if (nt > 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_k(T_lo + 3, j);
        }
        f32x16 OA[4], OB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { OA[t][r] = 0.f; OB[t][r] = 0.f; }
        // everything but K(3) has landed (K(0), Q first)
        if (nt > 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        Q64_T(1);

        f32x16 S0[4], S1[4];                 // [A kb0, A kb1, B kb0, B kb1] of the current / the next tile (roles alternate)
        u32x4 PA[4], PB[4];
        u32x4 fr[2][4];
        f32x16 negmA, negmB;
        float mA, mB, lA = 0.f, lB = 0.f;

        // masks the keys >= N of 64-key tile T (only the last tile of the sequence can have any)
        auto mask_tile = [&](f32x16 (&S)[4], int T) __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (T * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) { S[kb][r] = -INFINITY; S[2 + kb][r] = -INFINITY; }
        };
        // the 32 S^T MFMAs of a tile whose K fragments sit in the K ring at kaddr; group 0's four fragments are already requested
        // into fr[0].  FIRST: seed 0 (prologue tile), no shadow work.  fill(n) = the VALU / DMA shadow of gap n.  last4(q): the read
        // issued in gaps 24..27 (the next phase's group 0).
        auto qk_phase = [&](f32x16 (&D)[4], unsigned kaddr, auto first_, auto&& fill, auto&& last4) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_)::value != 0;
            sfor<0, 32>([&](auto n_) {
                constexpr int n = decltype(n_)::value;
                constexpr int s = n >> 2, kb = (n >> 1) & 1, x = n & 1;
                constexpr int j = s * 2 + kb, grp = j >> 2, q = j & 3;
                if constexpr ((n & 7) == 0) q64_lgkm0();
                f32x16& acc = D[x * 2 + kb];
                if constexpr (s == 0) {
                    if constexpr (FIRST) mfma_s0(acc, fr[grp & 1][q], x ? QB[s] : QA[s]);
                    else mfma_sc(acc, fr[grp & 1][q], x ? QB[s] : QA[s], x ? negmB : negmA);
                } else mfma_s(acc, fr[grp & 1][q], x ? QB[s] : QA[s]);
                if constexpr ((n & 7) < 4) {
                    if constexpr (grp < 3) {
                        constexpr int j2 = (grp + 1) * 4 + (n & 7), s2 = j2 >> 1, kb2 = j2 & 1;
                        lds_read_a<(kb2 * 8 + s2) * 1024>(fr[(grp + 1) & 1][n & 7], kaddr);
                    } else last4(IC<(n & 7)>{});
                }
                fill(n_);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // group 0 of a K tile: fragments (kb, s) = (0,0) (1,0) (0,1) (1,1)
        auto k_group0 = [&](unsigned kaddr) __attribute__((always_inline)) {
            lds_read_a<0 * 1024>(fr[0][0], kaddr); lds_read_a<8 * 1024>(fr[0][1], kaddr);
            lds_read_a<1 * 1024>(fr[0][2], kaddr); lds_read_a<9 * 1024>(fr[0][3], kaddr);
        };
        // V^T fragment (kk, td) of a 64-key tile: 32-key tile kb = kk >> 1, K-step k2 = kk & 1
#define Q64_VOFF(kk, td) ((((kk) >> 1) * 8 + (td) * 2 + ((kk) & 1)) * 1024)
        // the 32 O^T MFMAs of a tile (V^T fragments at vaddr; group 0 already requested into fr[0])
        auto pv_phase = [&](unsigned vaddr, auto&& fill, auto&& last4) __attribute__((always_inline)) {
            sfor<0, 32>([&](auto n_) {
                constexpr int n = decltype(n_)::value;
                constexpr int kk = n >> 3, td = (n >> 1) & 3, x = n & 1;
                if constexpr ((n & 7) == 0) q64_lgkm0();
                mfma_o(x ? OB[td] : OA[td], fr[kk & 1][td], x ? PB[kk] : PA[kk]);
                if constexpr ((n & 7) < 4) {
                    if constexpr (kk < 3) lds_read_a<Q64_VOFF(kk + 1, n & 7)>(fr[(kk + 1) & 1][n & 7], vaddr);
                    else last4(IC<(n & 7)>{});
                }
                fill(n_);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto v_group0 = [&](unsigned vaddr) __attribute__((always_inline)) {
            lds_read_a<Q64_VOFF(0, 0)>(fr[0][0], vaddr); lds_read_a<Q64_VOFF(0, 1)>(fr[0][1], vaddr);
            lds_read_a<Q64_VOFF(0, 2)>(fr[0][2], vaddr); lds_read_a<Q64_VOFF(0, 3)>(fr[0][3], vaddr);
        };
        // softmax pieces.  Value n (0..31) of a key block: query block X = n >> 4, register n & 15; pack pair k (0..15) of key block kb:
        // block X = k >> 3, pair pi = k & 7 -> P_X[2 kb + (pi >> 2)][pi & 3].  Gap n of phase A: exponential of second-half value n,
        // one pack (gaps 0..15: the first-half pairs, 16..31: the second-half pairs, each after its two exponentials), the row sum.
        auto fill_a = [&](f32x16 (&C)[4], auto n_) __attribute__((always_inline)) {
            constexpr int n = decltype(n_)::value, X = n >> 4, r = n & 15;
            constexpr int kbp = n < 16 ? 0 : 1, k = n < 16 ? n : n - 16, XP = k >> 3, pi = k & 7;     // this gap's pack
            float sv = C[X * 2 + 1][r];
            unsigned w;
            if constexpr (n < 31) {
                if constexpr (X) gap_exp_pack_sum(sv, lB, w, C[XP * 2 + kbp][2 * pi], C[XP * 2 + kbp][2 * pi + 1]);
                else gap_exp_pack_sum(sv, lA, w, C[XP * 2 + kbp][2 * pi], C[XP * 2 + kbp][2 * pi + 1]);
            } else gap_exp_sum_pack(sv, lB, w, C[XP * 2 + kbp][2 * pi]);       // n = 31: pair (B, kb 1, 14 | 15), 15 is this gap's value
            C[X * 2 + 1][r] = sv;
            if constexpr (XP) PB[2 * kbp + (pi >> 2)][pi & 3] = w; else PA[2 * kbp + (pi >> 2)][pi & 3] = w;
        };
        // first-half exponentials, two per statement: values 2 n, 2 n + 1 of key block 0 (n = 0..15)
        auto fill_b = [&](f32x16 (&S)[4], auto n_) __attribute__((always_inline)) {
            constexpr int n = decltype(n_)::value, X = n >> 3, r = (2 * n) & 15;
            float s0 = S[X * 2][r], s1 = S[X * 2][r + 1];
            if constexpr (X) gap_exp2_sum2(s0, s1, lB); else gap_exp2_sum2(s0, s1, lA);
            S[X * 2][r] = s0; S[X * 2][r + 1] = s1;
        };

        // ---- tile 0: scores with a zero seed, first reference maximum, first-half exponentials
        {
            const unsigned ka = lane16 + (T_lo & 3) * TILE_BYTES;
            k_group0(ka);
            qk_phase(S0, ka, IC<1>{}, [&](auto) {}, [&](auto) {});
            fence_v(S0[0], S0[1], S0[2], S0[3]);
            if (T_lo * 64 + 64 > N) mask_tile(S0, T_lo);
            float c0, c1, c2, c3;
            gap_max_first(c0, c1, c2, c3, S0[0][0], S0[0][1], S0[0][2], S0[1][0], S0[1][1], S0[1][2], S0[2][0], S0[2][1], S0[2][2], S0[3][0], S0[3][1], S0[3][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) gap_max_next(c0, c1, c2, c3, S0[0][r], S0[0][r + 1], S0[1][r], S0[1][r + 1], S0[2][r], S0[2][r + 1], S0[3][r], S0[3][r + 1]);
            gap_max_last(mA, mB, c0, c1, c2, c3, S0[0][15], S0[1][15], S0[2][15], S0[3][15]);
            mA = xhalf_max(mA); mB = xhalf_max(mB);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                S0[0][r] -= mA; S0[1][r] -= mA; S0[2][r] -= mB; S0[3][r] -= mB;
                negmA[r] = -mA; negmB[r] = -mB;
            }
            sfor<0, 16>([&](auto n_) { fill_b(S0, n_); });
        }
        asm volatile("s_barrier" ::: "memory");      // every wave has read K(0): its slot may take K(4)
        Q64_T(2);

        // ---- steady state: iteration it (tile T = T_lo + it is current, T + 1 is next), it + 1 < nt
        int vs0 = 0, vs1 = 1, vs2 = 2;       // V^T slots of tiles it, it + 1, it + 2
        bool pend = false;
        float alA = 1.f, alB = 1.f;
        auto iter = [&](f32x16 (&C)[4], f32x16 (&Nx)[4], int it) __attribute__((always_inline)) {
            const int T = T_lo + it;
            const unsigned ka = lane16 + ((T + 1) & 3) * TILE_BYTES;
            const unsigned va = lane16 + V_RING + vs0 * TILE_BYTES;
            const unsigned ka2 = lane16 + ((T + 2) & 3) * TILE_BYTES;
            const bool has_k = it + 4 < nt, has_v = it + 2 < nt;
            // phase A: S(T+1) | second-half exponentials + all packs of tile T | K(T+4) DMA
            qk_phase(Nx, ka, IC<0>{},
                     [&](auto n_) {
                         constexpr int n = decltype(n_)::value;
#ifndef Q64_NO_FILL
                         fill_a(C, n_);
#endif
#ifndef Q64_NO_DMA
                         if constexpr ((n & 7) == 5) { if (has_k) dma_k(T + 4, n >> 3); }
#endif
                     },
                     [&](auto q_) { constexpr int q = decltype(q_)::value; lds_read_a<Q64_VOFF(0, q)>(fr[0][q], va); });
            // phase B: O += V(T) P(T) | maxima of S(T+1), reference move, first-half exponentials of tile T+1 | V(T+2) DMA
            float c0, c1, c2, c3, mxA, mxB;
            pv_phase(va,
                     [&](auto n_) {
                         constexpr int n = decltype(n_)::value;
#ifndef Q64_NO_DMA
                         if constexpr (n == 0 || n == 1 || n == 10 || n == 11) { if (has_v) dma_v(T + 2, vs2, n < 2 ? n : n - 8); }
#endif
#ifndef Q64_NO_FILL
                         if constexpr (n == 2) {
                             if ((T + 1) * 64 + 64 > N) { asm volatile("s_nop 7" ::: "memory"); mask_tile(Nx, T + 1); }
                             gap_max_first(c0, c1, c2, c3, Nx[0][0], Nx[0][1], Nx[0][2], Nx[1][0], Nx[1][1], Nx[1][2],
                                           Nx[2][0], Nx[2][1], Nx[2][2], Nx[3][0], Nx[3][1], Nx[3][2]);
                         }
                         if constexpr (n >= 3 && n <= 8) {
                             constexpr int r = 2 * (n - 3) + 3;
                             gap_max_next(c0, c1, c2, c3, Nx[0][r], Nx[0][r + 1], Nx[1][r], Nx[1][r + 1], Nx[2][r], Nx[2][r + 1], Nx[3][r], Nx[3][r + 1]);
                         }
                         if constexpr (n == 9) gap_max_last(mxA, mxB, c0, c1, c2, c3, Nx[0][15], Nx[1][15], Nx[2][15], Nx[3][15]);
                         if constexpr (n == 10) { mxA = xhalf_max(mxA); mxB = xhalf_max(mxB); }
                         if constexpr (n == 11) {
                             if (__builtin_amdgcn_ballot_w64(max2(mxA, mxB) > 8.f) != 0) {
                                 // the reference maximum moves: scores of tile T+1 (already relative to the old one), the seeds, the
                                 // row sums now; O^T after this phase's MFMAs (they still use P(T), which is relative to the old one)
                                 const float dA = fmaxf(mxA, 0.f), dB = fmaxf(mxB, 0.f);
                                 mA += dA; mB += dB;
                                 alA = __builtin_amdgcn_exp2f(-dA); alB = __builtin_amdgcn_exp2f(-dB);
                                 lA *= alA; lB *= alB;
#pragma unroll
                                 for (int r = 0; r < 16; ++r) {
                                     Nx[0][r] -= dA; Nx[1][r] -= dA; Nx[2][r] -= dB; Nx[3][r] -= dB;
                                     negmA[r] = -mA; negmB[r] = -mB;
                                 }
                                 pend = true;
                             }
                         }
                         if constexpr (n >= 12 && n < 28) fill_b(Nx, IC<n - 12>{});
#endif
                     },
                     [&](auto q_) {
                         constexpr int q = decltype(q_)::value;      // group 0 of K(T+2): (kb, s) = (q & 1, q >> 1)
                         lds_read_a<((q & 1) * 8 + (q >> 1)) * 1024>(fr[0][q], ka2);
                     });
            if (pend) {
                fence_a(OA[0], OA[1], OA[2], OA[3]); fence_a(OB[0], OB[1], OB[2], OB[3]);
#pragma unroll
                for (int t = 0; t < 4; ++t) { OA[t] *= alA; OB[t] *= alB; }
                pend = false;
            }
            // everything issued before this iteration has landed; this iteration's pieces fly on
            if (has_k) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (has_v) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef Q64_NO_BAR
            asm volatile("s_barrier" ::: "memory");
#endif
            const int tmp = vs0; vs0 = vs1; vs1 = vs2; vs2 = tmp;
        };
        {
            if (nt > 1) k_group0(lane16 + ((T_lo + 1) & 3) * TILE_BYTES);
            int it = 0;
            while (it + 1 < nt) {
                iter(S0, S1, it); ++it;
                if (it + 1 >= nt) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) S0[a] = S1[a];
                    break;
                }
                iter(S1, S0, it); ++it;
            }
        }
        Q64_T(3);
        // ---- last tile: second-half exponentials + packs (no MFMAs left to hide them), then its O^T MFMAs
        {
            sfor<0, 32>([&](auto n_) { fill_a(S0, n_); });
            const unsigned va = lane16 + V_RING + vs0 * TILE_BYTES;
            v_group0(va);
            pv_phase(va, [&](auto) {}, [&](auto) {});
        }
        asm volatile("s_barrier" ::: "memory");      // the rings are free for the next unit's prologue
        fence_a(OA[0], OA[1], OA[2], OA[3]); fence_a(OB[0], OB[1], OB[2], OB[3]);
        Q64_T(4);
        // ---- epilogue: normalise, store (fp32, or the mode's 16-bit type), (m, l) for the consumer's merge of key splits
        lA = xhalf_sum(lA); lB = xhalf_sum(lB);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const bool live = which ? liveB : liveA;
            const int q0 = (which ? qtB : qtA) * 32;
            const float l = which ? lB : lA, m = which ? mB : mA;
            if (live && q0 + i < N) {
                const float inv = l > 0.f ? 1.f / l : 0.f;
                if (p.o_lp) {
                    unsigned short* oh = reinterpret_cast<unsigned short*>(p.O) + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const f32x16& o = which ? OB[t] : OA[t];
                            *reinterpret_cast<uint2*>(oh + t * 32 + 8 * rq) =
                                make_uint2(pack2_lp(o[rq * 4 + 0] * inv, o[rq * 4 + 1] * inv), pack2_lp(o[rq * 4 + 2] * inv, o[rq * 4 + 3] * inv));
                        }
                } else {
                    float* op = p.O + (long)sp * p.o_sstride + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const f32x16& o = which ? OB[t] : OA[t];
                            *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) =
                                make_float4(o[rq * 4 + 0] * inv, o[rq * 4 + 1] * inv, o[rq * 4 + 2] * inv, o[rq * 4 + 3] * inv);
                        }
                }
                if (p.ml && hh == 0) {
                    float* ml = p.ml + ((((long)sp * p.B + b) * 2 + h) * N + q0 + i) * 2;
                    ml[0] = m; ml[1] = l;
                }
            }
        }
        Q64_T(5);
#ifdef Q64_STAMP
        if (p.dbg && lane == 0) {
            long long* d = p.dbg + ((long)unit * 4 + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = tst[k];
            d[6] = nt; d[7] = __builtin_amdgcn_s_getreg(0x14 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID
        }
#endif
    }
}

// Key split for the 64-query form: pick the split (1..max_split) that minimises  rounds of workgroups x (key tiles + overhead)
int attention_q64_ksplit(int N, int B, int max_split) {
    const int nt32 = (N + 31) / 32, nT = (nt32 + 1) / 2, ng = (nt32 + 7) / 8;
    int best = 1; double best_cost = 1e30;
    for (int ks = 1; ks <= max_split && ks <= nT; ++ks) {
        const long units = 2L * B * ng * ks;
        const long rounds = (units + 255) / 256;
        const double cost = rounds * ((nT + ks - 1) / ks + 2.5) + 0.75 * (ks - 1);       // (+ the consumer's merge of every extra partial)
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    return best;
}

void launch_attention_q64(const AttnDirectP& p, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr = true;
    }
    const int nt32 = (p.N + 31) / 32, ng = (nt32 + 7) / 8, ks = p.ksplit > 1 ? p.ksplit : 1;
    const int xcd_mode = ((2 * p.B) % 8 == 0) ? 1 : 0;
    const int nunits = 2 * p.B * ng * ks;
    const int grid = nunits < 256 ? nunits : 256;
    hipLaunchKernelGGL(attn_q64_kernel, dim3(grid), dim3(256), LDS_BYTES, st, p, ng, nunits, xcd_mode);
}

}  // namespace DEX_LP_NS
}  // namespace dex
