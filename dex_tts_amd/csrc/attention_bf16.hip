// attention_bf16.hip — softmax attention core on v_mfma_f32_32x32x16_bf16 (fp32 Q/K/V in HBM, converted to
// bf16 while staging; fp32 scores, softmax state and accumulation).  head_dim 128.
//
// Transposed flash formulation (every per-query quantity lane-local, no cross-lane traffic for P):
//     S^T = K Q^T    A = K rows [key][d] from LDS (16-B reads), B = Q^T fragments kept in registers (pre-scaled)
//     O^T = V^T P^T  A = V^T from LDS laid out [d][pos(key)], B = P^T = the S^T accumulator registers packed
//                    to bf16 — the contraction order over keys is the accumulator order, so V^T is stored with
//                    key bits 2 and 3 swapped (pos(key)) and P needs no shuffle at all.
// C-layout of a 32x32 tile: col = lane&31 (query), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//
// Two workgroup shapes:
//   SPLIT=false: 4 waves = 4 query tiles (128 queries) sharing double-buffered K/V tiles of 64 keys
//                (one barrier per tile, next tile prefetched into registers during the MFMAs);
//   SPLIT=true : NW waves share ONE 32-query tile and split the keys (wave-private 32-key tiles, partial
//                (m, l, O) merged through LDS) — keeps small token counts (N=650 at B=1) spread over the chip.
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
constexpr int AHD = 128;
constexpr int K_LD = AHD + 8;             // bf16 elements per K row in LDS (272 B)

__device__ __forceinline__ int key_pos(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

union Frag { uint4 u; lp8 v; };

// Stage KT keys starting at k0 (rows clamped to Nk-1; masked later through the scores) with GS cooperating
// threads (index t): K -> kS[key][K_LD], V -> vT[d][KT+8] (transposed, key positions permuted).
template <int KT, int GS>
struct Stager {
    static constexpr int KI = KT * AHD / 8 / GS;          // K items (8 consecutive d of one key) per thread
    static constexpr int VI = (KT / 2) * (AHD / 4) / GS;  // V items (key pair x 4 consecutive d) per thread
    float4 kr[KI][2];
    float4 vr[VI][2];
    // V item -> (key pair, 4 channels): consecutive threads walk the 128 channels of one key pair (coalesced
    // 512-B rows).  The transposed LDS dword writes of a wave then collide 16-way; a key-pair-major map is
    // conflict-free but measured slower (uncoalesced 64-B row segments cost more than the LDS replays).
    static __device__ __forceinline__ int v_kp(int it) { return it / (AHD / 4); }
    static __device__ __forceinline__ int v_d4(int it) { return (it % (AHD / 4)) * 4; }
    __device__ __forceinline__ void load(const float* Kb, int ldk, const float* Vb, int ldv, int k0, int Nk, int t) {
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int it = t + GS * j;
            const int key = it / (AHD / 8), d8 = (it % (AHD / 8)) * 8;
            const float* src = Kb + (long)min(k0 + key, Nk - 1) * ldk + d8;
            kr[j][0] = *reinterpret_cast<const float4*>(src);
            kr[j][1] = *reinterpret_cast<const float4*>(src + 4);
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int kp = v_kp(t + GS * j), d4 = v_d4(t + GS * j);
            vr[j][0] = *reinterpret_cast<const float4*>(Vb + (long)min(k0 + 2 * kp, Nk - 1) * ldv + d4);
            vr[j][1] = *reinterpret_cast<const float4*>(Vb + (long)min(k0 + 2 * kp + 1, Nk - 1) * ldv + d4);
        }
    }
    __device__ __forceinline__ void store(u16* kS, u16* vT, int t) {
        constexpr int V_LD = KT + 8;
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int it = t + GS * j;
            const int key = it / (AHD / 8), d8 = (it % (AHD / 8)) * 8;
            uint4 u;
            u.x = pack2_lp_asm(kr[j][0].x, kr[j][0].y); u.y = pack2_lp_asm(kr[j][0].z, kr[j][0].w);
            u.z = pack2_lp_asm(kr[j][1].x, kr[j][1].y); u.w = pack2_lp_asm(kr[j][1].z, kr[j][1].w);
            *reinterpret_cast<uint4*>(kS + key * K_LD + d8) = u;
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int kp = v_kp(t + GS * j), d4 = v_d4(t + GS * j);
            unsigned* dst = reinterpret_cast<unsigned*>(vT + key_pos(2 * kp));      // even position: dword aligned
            dst[((d4 + 0) * V_LD) >> 1] = pack2_lp_asm(vr[j][0].x, vr[j][1].x);
            dst[((d4 + 1) * V_LD) >> 1] = pack2_lp_asm(vr[j][0].y, vr[j][1].y);
            dst[((d4 + 2) * V_LD) >> 1] = pack2_lp_asm(vr[j][0].z, vr[j][1].z);
            dst[((d4 + 3) * V_LD) >> 1] = pack2_lp_asm(vr[j][0].w, vr[j][1].w);
        }
    }
};

// One KV tile of KT keys for one wave's 32 queries: scores, online softmax, P.V.
template <int KT>
__device__ __forceinline__ void attn_tile(const u16* kS, const u16* vT, const Frag (&qf)[8], f32x16 (&o)[4],
                                          float& m_run, float& l_run, int k0, int Nk, int lane) {
    constexpr int V_LD = KT + 8, NS = KT / 32;
    const int i = lane & 31, hh = lane >> 5;
    f32x16 s[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
        const u16* ka = kS + (st * 32 + i) * K_LD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            Frag a; a.u = *reinterpret_cast<const uint4*>(ka + ks * 16);
            s[st] = DEX_MFMA_LP(a.v, qf[ks].v, s[st], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Nk) s[st][r] = -INFINITY;
            mx = fmaxf(mx, s[st][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[st][r] = __expf(s[st][r] - m_new); psum += s[st][r]; }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            Frag pb;
            pb.u.x = pack2_lp_asm(s[st][8 * k2 + 0], s[st][8 * k2 + 1]); pb.u.y = pack2_lp_asm(s[st][8 * k2 + 2], s[st][8 * k2 + 3]);
            pb.u.z = pack2_lp_asm(s[st][8 * k2 + 4], s[st][8 * k2 + 5]); pb.u.w = pack2_lp_asm(s[st][8 * k2 + 6], s[st][8 * k2 + 7]);
            const u16* va = vT + i * V_LD + (st * 2 + k2) * 16 + hh * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Frag a; a.u = *reinterpret_cast<const uint4*>(va + t * 32 * V_LD);
                o[t] = DEX_MFMA_LP(a.v, pb.v, o[t], 0, 0, 0);
            }
        }
}

__device__ __forceinline__ void load_q(Frag (&qf)[8], const float* Qb, int ldq, int qrow, int Nq, float scale, int hh) {
    // branch-free (row clamped, out-of-range rows scaled to zero): all 16 loads issue back to back
    const float sc = qrow < Nq ? scale : 0.f;
    const float* qp = Qb + (long)min(qrow, Nq - 1) * ldq + hh * 8;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 4) {          // two batches of 8 loads: 32 staging VGPRs instead of 64
        float4 a[4], c[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a[ks] = *reinterpret_cast<const float4*>(qp + (k0 + ks) * 16);
            c[ks] = *reinterpret_cast<const float4*>(qp + (k0 + ks) * 16 + 4);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[k0 + ks].u.x = pack2_lp_asm(a[ks].x * sc, a[ks].y * sc); qf[k0 + ks].u.y = pack2_lp_asm(a[ks].z * sc, a[ks].w * sc);
            qf[k0 + ks].u.z = pack2_lp_asm(c[ks].x * sc, c[ks].y * sc); qf[k0 + ks].u.w = pack2_lp_asm(c[ks].z * sc, c[ks].w * sc);
        }
    }
}

// ---- SPLIT=false: NWS waves = NWS query tiles sharing the K/V tiles of 64 keys ---------------------------
// (NWS = 8: the staging of a K/V tile - 64 KB of fp32 converted and transposed into LDS - is shared by 256 queries instead of 128;
// with few keys per utterance, the DEX TV adaptor's <= 349, that staging is most of the kernel)
template <int NWS>
__global__ __launch_bounds__(64 * NWS) void attn_lp_shared_kernel(const AttnP p) {
    constexpr int KT = 64, V_LD = KT + 8;
    constexpr int KBUF = KT * K_LD, VBUF = AHD * V_LD;
    extern __shared__ __attribute__((aligned(16))) u16 smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * (32 * NWS) + wave * 32, h = blockIdx.y, b = blockIdx.z;
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const float* Qb = p.Q + (long)b * p.qb + h * AHD;
    const float* Kb = p.K + (long)b * p.kb + h * AHD;
    const float* Vb = p.V + (long)b * p.vb + h * AHD;
    Frag qf[8];
    load_q(qf, Qb, p.ldq, q0 + i, p.Nq, p.scale, hh);
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (Nk + KT - 1) / KT;
    Stager<KT, 64 * NWS> sg;
    sg.load(Kb, p.ldk, Vb, p.ldv, 0, Nk, tid);
    sg.store(smem_b, smem_b + 2 * KBUF, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();                                   // tile kt visible; buffer (kt+1)&1 free
        const int cur = kt & 1;
        if (kt + 1 < ntiles) sg.load(Kb, p.ldk, Vb, p.ldv, (kt + 1) * KT, Nk, tid);
        attn_tile<KT>(smem_b + cur * KBUF, smem_b + 2 * KBUF + cur * VBUF, qf, o, m_run, l_run, kt * KT, Nk, lane);
        if (kt + 1 < ntiles) sg.store(smem_b + (cur ^ 1) * KBUF, smem_b + 2 * KBUF + (cur ^ 1) * VBUF, tid);
    }
    l_run += __shfl_xor(l_run, 32);
    const int qrow = q0 + i;
    if (qrow < p.Nq) {
        const float inv = 1.f / l_run;
        float* op = p.O + (long)b * p.ob + (long)qrow * p.ldo + h * AHD;
        if (p.o_lp) {          // (uniform) O in the mode's 16-bit type: its one reader, a GEMM, rounds it so anyway
            u16* oh = reinterpret_cast<u16*>(p.O) + (long)b * p.ob + (long)qrow * p.ldo + h * AHD;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<uint2*>(oh + t * 32 + 8 * rq + 4 * hh) =
                        make_uint2(pack2_lp(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv), pack2_lp(o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<float4*>(op + t * 32 + 8 * rq + 4 * hh) =
                        make_float4(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv, o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv);
        }
    }
}

// ---- SPLIT=true: NW waves share one 32-query tile and split the keys ----------------------------------
constexpr int SP_KT = 32;
constexpr int SP_WAVE_U16 = SP_KT * K_LD + AHD * (SP_KT + 8);   // 4352 + 5120 = 9472 u16 = 18944 B >= 32*132*4
constexpr int SP_O_LD = 132;

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_lp_split_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    u16* kS = smem_b + wave * SP_WAVE_U16;
    u16* vT = kS + SP_KT * K_LD;
    float* stat = reinterpret_cast<float*>(smem_b + NW * SP_WAVE_U16);     // [NW][2][32]
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const float* Qb = p.Q + (long)b * p.qb + h * AHD;
    const float* Kb = p.K + (long)b * p.kb + h * AHD;
    const float* Vb = p.V + (long)b * p.vb + h * AHD;
#ifdef DEX_TIMING
    long long tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tst[0] = wall_clock64();
#endif
    Frag qf[8];
    load_q(qf, Qb, p.ldq, q0 + i, p.Nq, p.scale, hh);
    const int ntiles = (Nk + SP_KT - 1) / SP_KT;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    if constexpr (NW >= 8) {
        // 2 waves per SIMD leave 256 VGPRs: no room for a second tile's fp32 staging registers
        for (int kt = t_lo + wave; kt < t_hi; kt += NW) {
            Stager<SP_KT, 64> sg;
            sg.load(Kb, p.ldk, Vb, p.ldv, kt * SP_KT, Nk, lane);
            sg.store(kS, vT, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): wave-private tile written
            __builtin_amdgcn_wave_barrier();
            attn_tile<SP_KT>(kS, vT, qf, o, m_run, l_run, kt * SP_KT, Nk, lane);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        int kt = t_lo + wave;
        Stager<SP_KT, 64> sg;
        if (kt < t_hi) sg.load(Kb, p.ldk, Vb, p.ldv, kt * SP_KT, Nk, lane);
        while (kt < t_hi) {
            sg.store(kS, vT, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int kn = kt + NW;
            if (kn < t_hi) sg.load(Kb, p.ldk, Vb, p.ldv, kn * SP_KT, Nk, lane);   // next tile in flight under the MFMAs
            attn_tile<SP_KT>(kS, vT, qf, o, m_run, l_run, kt * SP_KT, Nk, lane);
            __builtin_amdgcn_wave_barrier();
            kt = kn;
        }
    }
    l_run += __shfl_xor(l_run, 32);
#ifdef DEX_TIMING
    asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[3][15])); tst[4] = wall_clock64();
#endif
    __syncthreads();
#ifdef DEX_TIMING
    tst[5] = wall_clock64();
#endif
    float* oS = reinterpret_cast<float*>(smem_b + wave * SP_WAVE_U16);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<float4*>(oS + i * SP_O_LD + t * 32 + 8 * rq + 4 * hh) =
                make_float4(o[t][rq * 4 + 0], o[t][rq * 4 + 1], o[t][rq * 4 + 2], o[t][rq * 4 + 3]);
    if (hh == 0) { stat[(wave * 2 + 0) * 32 + i] = m_run; stat[(wave * 2 + 1) * 32 + i] = l_run; }
    __syncthreads();
    const int d4 = (tid & 31) * 4;
    for (int q = tid >> 5; q < 32; q += NW * 2) {
        if (q0 + q >= p.Nq) continue;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, stat[(w * 2) * 32 + q]);
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = stat[(w * 2) * 32 + q];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += f * stat[(w * 2 + 1) * 32 + q];
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem_b + w * SP_WAVE_U16) + q * SP_O_LD + d4);
            acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
        }
        const float inv = L > 0.f ? 1.f / L : 0.f;
        float* op = p.O + (long)sp * p.o_sstride + (long)b * p.ob + (long)(q0 + q) * p.ldo + h * AHD + d4;
        *reinterpret_cast<float4*>(op) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (ksplit > 1 && d4 == 0) {
            float* ml = p.ml + ((((long)sp * p.B + b) * p.heads + h) * p.Nq + q0 + q) * 2;
            ml[0] = M; ml[1] = L;
        }
    }
#ifdef DEX_TIMING
    if (p.dbg && lane == 0 && (wave == 0 || wave == NW - 1)) {
        tst[6] = wall_clock64();
        long long* d = p.dbg + (((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) * 2 + (wave ? 1 : 0)) * 8;
        for (int k = 0; k < 8; ++k) d[k] = tst[k];
    }
#endif
}

template <int NW>
static void launch_split(const AttnP& p, hipStream_t st) {
    const size_t lds = (size_t)NW * SP_WAVE_U16 * sizeof(u16) + NW * 64 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_lp_split_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid((p.Nq + 31) / 32, p.heads, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL((attn_lp_split_kernel<NW>), grid, dim3(NW * 64), lds, st, p);
}

// the shared-K/V form (the only one that implements AttnP::o_lp) takes a launch of this size
bool attention_lp_shared_form(int Nq, int heads, int B, int ksplit) { return (long)((Nq + 127) / 128) * heads * B >= 256 && ksplit <= 1; }

void launch_attention_lp(const AttnP& p, hipStream_t st) {
    const long blocks128 = (long)((p.Nq + 127) / 128) * p.heads * p.B;
    if (blocks128 >= 256 && p.ksplit <= 1) {
        constexpr int KT = 64;
        const size_t lds = (size_t)(2 * KT * K_LD + 2 * AHD * (KT + 8)) * sizeof(u16);
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_lp_shared_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_lp_shared_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        const int w8 = knob_or("DEX_ATTN_SHARED_W8", 1);
        if (w8 && (long)((p.Nq + 255) / 256) * p.heads * p.B >= 512) {
            hipLaunchKernelGGL(attn_lp_shared_kernel<8>, dim3((p.Nq + 255) / 256, p.heads, p.B), dim3(512), lds, st, p);
            return;
        }
        dim3 grid((p.Nq + 127) / 128, p.heads, p.B);
        hipLaunchKernelGGL(attn_lp_shared_kernel<4>, grid, dim3(256), lds, st, p);
        return;
    }
    const long blocks32 = (long)((p.Nq + 31) / 32) * p.heads * p.B;
    const int ntiles = ((p.Nk + SP_KT - 1) / SP_KT + (p.ksplit > 1 ? p.ksplit - 1 : 0)) / (p.ksplit > 1 ? p.ksplit : 1);
    int nw = 8;
    if (blocks32 * 4 >= 2048 || ntiles < 8) nw = 4;
    if (blocks32 * 2 >= 2048 || ntiles < 4) nw = 2;
    if (nw == 8) launch_split<8>(p, st);
    else if (nw == 4) launch_split<4>(p, st);
    else launch_split<2>(p, st);
}


// ------------------------------------------------------------------------------------------------------------------
// The DEX TV adaptor as ONE launch (batch regime; ref_encoder.py:154-179, kernels.h TvChainP).  As separate launches the adaptor is
// a 1x1 GEMM (q, 84 MB written at B = 32), the shared-K/V attention above (q read back, every workgroup converting and transposing
// the same <= 349 fp32 keys / values of its utterance, 88 us) and a second 1x1 GEMM (output projection + residual, 4-byte stores
// per lane): 206 us of a 2.4 ms Euler step for 40 GFLOP and 168 MB of tensors that HAVE to move (x in, the result out).  Here a
// workgroup of four waves owns 128 pixels from x to the result:
//   * x rows arrive as coalesced 512-byte rows, are masked, rounded once and staged in LDS next to this utterance's folded W_eff;
//     q^T = W_eff x^T comes out of the MFMA with lane = pixel and the channels of a 16-group in accumulator order, which IS the
//     B-operand order of the score MFMA once the keys' channels are stored in the same order (launch_tv_kv_prep) - as P^T needs
//     no shuffle in front of V^T (attn_tile), q needs none in front of K;
//   * the K / V^T tiles are 16-byte copies of operands prepared once per step (no conversion, no transposed LDS writes);
//   * O^T (lane = pixel) is normalised, rounded and fed to out^T = W_l O^T the same way (W_l staged with permuted columns);
//   * the result leaves through an LDS stage as whole 256-byte row segments, where the residual is added, the mask applied and the
//     per-channel statistics of the TIV adaptor's InstanceNorm accumulate with lane = channel (fixed order inside a wave,
//     fixed-point integer adds across waves and workgroups: deterministic, kernels.h GN_SLOTS).
// Roundings are the ones of the separate launches (x, q / sqrt(C), P, O in the operand type; fp32 everywhere else).
constexpr int TVC_KT = 64, TVC_VLD = TVC_KT + 8;
constexpr int TVC_KBUF = TVC_KT * K_LD, TVC_VBUF = AHD * TVC_VLD;
constexpr int TVC_U16 = 2 * TVC_KBUF + 2 * TVC_VBUF;            // 35840 u16 = 71680 B: the K / V rings; W_eff + x tile and W_l + output stage alias them
constexpr int TVC_FBUF = TVC_KT * AHD;                          // u16 per fragment-ordered K or V^T tile (16 KB: sixteen 1-KB pieces)
constexpr int TVC_SLD = 68;                                     // floats per pixel of the output stage (64 channels + 4: conflict-free 16 B accesses)
static_assert(2 * 128 * K_LD <= TVC_U16 && 128 * K_LD * 2 + 128 * TVC_SLD * 4 <= TVC_U16 * 2, "aliases fit the rings");

// One tile of 64 style keys for a wave's 32 pixels: attn_tile's products in attn_tile's order with less VALU work around them - the
// softmax is the instruction stream that bounds this kernel (32 MFMAs against ~340 VALU instructions per tile in attn_tile):
//   * key masking (compare + select per score) only in the tile that holds keys >= Nk;
//   * exp(s - m) as ONE fma into the exp2 domain + v_exp_f32 (m2 = m * log2 e per lane);
//   * LAZY rescaling: the reference maximum m moves - and the 64 accumulator multiplies run - only when some pixel's tile maximum
//     exceeds it by more than TVC_TAU; until then P = exp(s - m) is formed against the old m (<= e^TAU, no overflow in either operand
//     type), which is the same softmax once O is divided by the l accumulated against the same m.
constexpr float TVC_TAU = 6.f, TVC_LOG2E = 1.4426950408889634f;
// FRAG (the folded form below): the tiles lie in LDS in MFMA fragment order - piece (st, ks) of K', piece (t, st * 2 + k2) of V'^T, 1 KB
// each, lane l's 16 bytes at l * 16 - as the LDS-DMA ring delivers them from operands their producers write in that order (linear,
// unpadded, conflict-free); otherwise padded row-major tiles (K_LD / TVC_VLD).
// LAST: only the last tile of an utterance can hold keys >= Nk, and only it carries the masking code - as a (uniform) branch inside one
// tile function hipcc if-converted it into 30 compare + select pairs (+ their index arithmetic and hazard nops: ~110 of a tile's ~390
// instructions) in EVERY tile.
template <bool FRAG = false, bool LAST = true>
__device__ __forceinline__ void tvc_tile(const u16* kS, const u16* vT, const Frag (&qf)[8], f32x16 (&o)[4], float& m_run, float& l_run, int k0, int Nk, int lane) {
    constexpr int KT = TVC_KT, V_LD = TVC_VLD, NS = KT / 32;
    const int i = lane & 31, hh = lane >> 5;
    f32x16 s[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
        const u16* ka = FRAG ? kS + (st * 8 * 64 + lane) * 8 : kS + (st * 32 + i) * K_LD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            Frag a; a.u = *reinterpret_cast<const uint4*>(ka + ks * (FRAG ? 512 : 16));
            s[st] = DEX_MFMA_LP(a.v, qf[ks].v, s[st], 0, 0, 0);
        }
    }
    if constexpr (LAST) {
        if (k0 + KT > Nk) {                                                 // (uniform) the tile that holds keys >= Nk
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= Nk) s[st][r] = -INFINITY;
                }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_run + TVC_TAU) != 0) {          // (uniform) some pixel's maximum moved: new reference for every pixel
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    }
    const float m2 = -m_run * TVC_LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[st][r] = __builtin_amdgcn_exp2f(fmaf(s[st][r], TVC_LOG2E, m2)); psum += s[st][r]; }
    l_run += psum;
    // (software-pipelining the fragment reads one MFMA chain ahead with sched_group_barrier pairs was measured and is slower:
    // 108.5 against 99.9 us per launch at B = 32 - the tile is bound by LDS bandwidth and VALU issue at two waves per SIMD, not by
    // exposed read latency)
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            Frag pb;
            pb.u.x = pack2_lp_asm(s[st][8 * k2 + 0], s[st][8 * k2 + 1]); pb.u.y = pack2_lp_asm(s[st][8 * k2 + 2], s[st][8 * k2 + 3]);
            pb.u.z = pack2_lp_asm(s[st][8 * k2 + 4], s[st][8 * k2 + 5]); pb.u.w = pack2_lp_asm(s[st][8 * k2 + 6], s[st][8 * k2 + 7]);
            const u16* va = FRAG ? vT + ((st * 2 + k2) * 64 + lane) * 8 : vT + i * V_LD + (st * 2 + k2) * 16 + hh * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Frag a; a.u = *reinterpret_cast<const uint4*>(va + t * (FRAG ? 4 * 512 : 32 * V_LD));
                o[t] = DEX_MFMA_LP(a.v, pb.v, o[t], 0, 0, 0);
            }
        }
}

// K (fp32 [B][Nk][C]) -> the 16-bit K operand in MFMA fragment order, the channels of every 16-group in accumulator order (what q^T's
// accumulators are as a B operand): piece (st, ks) of 64-key tile t, lane (i, hh), elements 0..3 = channels 16 ks + 4 hh .. + 3,
// elements 4..7 = channels 16 ks + 4 hh + 8 .. + 11 of key t * 64 + st * 32 + i; keys >= Nk are zeros.  (V: tv_vfrag_prep_kernel below.)
__global__ __launch_bounds__(256) void tv_kv_prep_kernel(const TvKvPrepP p) {
    const long nk = (long)p.B * p.NkPad * (AHD / 8);             // K chunks: 8 positions of one key
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u16* Kp = reinterpret_cast<u16*>(p.Kp);
    if (idx < nk) {
        const int c = (int)(idx % (AHD / 8)), key = (int)((idx / (AHD / 8)) % p.NkPad), b = (int)(idx / ((long)(AHD / 8) * p.NkPad));
        const int d0 = 16 * (c >> 1) + 4 * (c & 1);              // positions 8c .. 8c + 7 hold channels d0 .. d0 + 3, d0 + 8 .. d0 + 11
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), e = a;
        if (key < p.Nk) {
            const float* src = p.K + (long)b * p.kvb + (long)key * AHD + d0;
            a = *reinterpret_cast<const float4*>(src); e = *reinterpret_cast<const float4*>(src + 8);
        }
        *reinterpret_cast<uint4*>(Kp + (long)b * p.NkPad * AHD + ((long)((key >> 6) * 16 + ((key >> 5) & 1) * 8 + (c >> 1)) * 64 + (c & 1) * 32 + (key & 31)) * 8) =
            make_uint4(pack2_lp(a.x, a.y), pack2_lp(a.z, a.w), pack2_lp(e.x, e.y), pack2_lp(e.z, e.w));
    }
}

#ifndef TVC_SKIP
#define TVC_SKIP 0             // tools/tvchainbench: phases left out (anatomy builds; results are wrong): 1 attention tiles, 2 result I/O, 4 x loads
#endif
__global__ __launch_bounds__(256, 2) void tv_chain_kernel(const TvChainP p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_b[];
    __shared__ long long red[2 * AHD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, row0 = blockIdx.x * 128;
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const int ntiles = (TVC_SKIP & 1) ? 0 : (Nk + TVC_KT - 1) / TVC_KT;
    const float* Xb = p.X + (long)b * p.x_bstride + p.x_coff;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const u16* Wg = reinterpret_cast<const u16*>(p.Weff) + (long)b * AHD * AHD;
    const u16* Kg = reinterpret_cast<const u16*>(p.Kp) + (long)b * p.NkPad * AHD;
    const u16* Vg = reinterpret_cast<const u16*>(p.VTp) + (long)b * AHD * p.NkPad;
    red[tid] = 0;

    // ---- K / V^T tile kt: 16 KB + 16 KB of ready operands in MFMA fragment order, sixteen + sixteen 1-KB pieces by LDS-DMA into ring half
    // kt & 1 ([K 0 | V^T 0 | K 1 | V^T 1]: four + four pieces per wave).  Round 6: the tiles travelled through 8 x 16 bytes of registers per
    // thread and a ds_write pass into padded rows - registers the kernel does not have: hipcc kept them in SCRATCH across every tile
    // (16 + 10 scratch instructions in the loop).  122 -> 75 us at configs[2] for the folded form below, which shares the ring.
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto kv_dma = [&](int kt) __attribute__((always_inline)) {
        u16* dst = smem_b + (kt & 1) * 2 * TVC_FBUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = 4 * wave + j;
            __builtin_amdgcn_global_load_lds(Kg + (long)kt * TVC_FBUF + piece * 512 + lane * 8, (lds_ptr)(dst + piece * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Vg + (long)kt * TVC_FBUF + piece * 512 + lane * 8, (lds_ptr)(dst + TVC_FBUF + piece * 512), 16, 0, 0);
        }
    };

    // ---- prologue: W_eff and the 128 x rows (masked, rounded once) into LDS
    u16* Ws = smem_b;                        // [128 n][K_LD]
    u16* Xs = smem_b + 128 * K_LD;           // [128 pixels][K_LD]
    {
        uint4 wr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int id = tid + 256 * j; wr[j] = *reinterpret_cast<const uint4*>(Wg + (id >> 4) * AHD + (id & 15) * 8); }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            float4 xa[8];
            float mk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = tid + 256 * (jb * 8 + j);
                const int pix = row0 + (id >> 5);
                const bool ok = pix < p.npix;
                const int pc = ok ? pix : p.npix - 1;
                xa[j] = (TVC_SKIP & 4) ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(Xb + (long)pc * p.ldx + (id & 31) * 4);
                const float mv = mrow[(pc % p.Wm) * p.mask_ws];
                mk[j] = ok ? mv : 0.f;
            }
            if (jb == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int id = tid + 256 * j; *reinterpret_cast<uint4*>(Ws + (id >> 4) * K_LD + (id & 15) * 8) = wr[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = tid + 256 * (jb * 8 + j);
                *reinterpret_cast<uint2*>(Xs + (id >> 5) * K_LD + (id & 31) * 4) =
                    make_uint2(pack2_lp(xa[j].x * mk[j], xa[j].y * mk[j]), pack2_lp(xa[j].z * mk[j], xa[j].w * mk[j]));
            }
        }
    }
    float4 bq[4][2][2];                      // b_eff of this lane's accumulator rows: requested before the barrier, back by the end of the first MFMA chain
    {
        const float* be = p.beff + (long)b * AHD;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                bq[t][k2][0] = *reinterpret_cast<const float4*>(be + t * 32 + 16 * k2 + 4 * hh);
                bq[t][k2][1] = *reinterpret_cast<const float4*>(be + t * 32 + 16 * k2 + 8 + 4 * hh);
            }
    }
    __syncthreads();

    // ---- q^T = W_eff x^T (+ b_eff), scaled, rounded: the B operand of the score MFMAs
    Frag qf[8];
    {
        Frag xf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[ks].u = *reinterpret_cast<const uint4*>(Xs + (wave * 32 + i) * K_LD + ks * 16 + hh * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                Frag a; a.u = *reinterpret_cast<const uint4*>(Ws + (t * 32 + i) * K_LD + ks * 16 + hh * 8);
                acc = DEX_MFMA_LP(a.v, xf[ks].v, acc, 0, 0, 0);
            }
#ifdef DEX_LP_WSPLIT
            if (p.weff_lo_off) {             // (uniform) lo halves straight from global: 32 KB per utterance, cache resident
                const u16* wl = Wg + p.weff_lo_off + (long)(t * 32 + i) * AHD + hh * 8;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    Frag a; a.u = *reinterpret_cast<const uint4*>(wl + ks * 16);
                    acc = DEX_MFMA_LP(a.v, xf[ks].v, acc, 0, 0, 0);
                }
            }
#endif
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float4 b0 = bq[t][k2][0], b1 = bq[t][k2][1];
                const float sc = p.scale;
                Frag& q = qf[2 * t + k2];
                q.u.x = pack2_lp((acc[8 * k2 + 0] + b0.x) * sc, (acc[8 * k2 + 1] + b0.y) * sc);
                q.u.y = pack2_lp((acc[8 * k2 + 2] + b0.z) * sc, (acc[8 * k2 + 3] + b0.w) * sc);
                q.u.z = pack2_lp((acc[8 * k2 + 4] + b1.x) * sc, (acc[8 * k2 + 5] + b1.y) * sc);
                q.u.w = pack2_lp((acc[8 * k2 + 6] + b1.z) * sc, (acc[8 * k2 + 7] + b1.w) * sc);
            }
        }
    }
    __syncthreads();                         // W_eff / x tile read by every wave: the rings may overwrite them

    // ---- attention over the style keys
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    if (ntiles > 0) kv_dma(0);               // (the W_eff / x tiles filled the ring's bytes: the first tile cannot land earlier)
    // (W_l's 32 KB are requested under the last tile, in the registers the K / V prefetch no longer needs)
    const u16* Wl = reinterpret_cast<const u16*>(p.Wl);
    uint2 la[8], lb[8];
    auto wl_load = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = tid + 256 * j;
            const int n = id >> 4, c = id & 15, d0 = 16 * (c >> 1) + 4 * (c & 1);
            la[j] = *reinterpret_cast<const uint2*>(Wl + n * AHD + d0);
            lb[j] = *reinterpret_cast<const uint2*>(Wl + n * AHD + d0 + 8);
        }
    };
    for (int kt = 0; kt + 1 < ntiles; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of tile kt have landed
        __syncthreads();                     // ... everybody's: tile kt visible; ring half (kt + 1) & 1 free
        kv_dma(kt + 1);
        tvc_tile<true, false>(smem_b + (kt & 1) * 2 * TVC_FBUF, smem_b + (kt & 1) * 2 * TVC_FBUF + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    wl_load();
    if (ntiles > 0) {
        const int kt = ntiles - 1;
        tvc_tile<true>(smem_b + (kt & 1) * 2 * TVC_FBUF, smem_b + (kt & 1) * 2 * TVC_FBUF + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
    }
    l_run += __shfl_xor(l_run, 32);
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    __syncthreads();                         // rings free

    // ---- out^T = W_l O^T: W_l with the columns of every 16-group in accumulator order
    u16* Ls = smem_b;                        // [128 n][K_LD]
    float* stage = reinterpret_cast<float*>(smem_b + 128 * K_LD);          // [128 pixels][TVC_SLD]
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = tid + 256 * j;
            *reinterpret_cast<uint4*>(Ls + (id >> 4) * K_LD + (id & 15) * 8) = make_uint4(la[j].x, la[j].y, lb[j].x, lb[j].y);
        }
    }
    Frag of[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            Frag& f = of[2 * t + k2];
            f.u.x = pack2_lp(o[t][8 * k2 + 0] * inv, o[t][8 * k2 + 1] * inv); f.u.y = pack2_lp(o[t][8 * k2 + 2] * inv, o[t][8 * k2 + 3] * inv);
            f.u.z = pack2_lp(o[t][8 * k2 + 4] * inv, o[t][8 * k2 + 5] * inv); f.u.w = pack2_lp(o[t][8 * k2 + 6] * inv, o[t][8 * k2 + 7] * inv);
        }
    __syncthreads();
    float gs[2][4], gq[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[h][e] = 0.f; gq[h][e] = 0.f; }
    float* myst = stage + wave * 32 * TVC_SLD;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * h + u;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                Frag a; a.u = *reinterpret_cast<const uint4*>(Ls + (t * 32 + i) * K_LD + ks * 16 + hh * 8);
                acc = DEX_MFMA_LP(a.v, of[ks].v, acc, 0, 0, 0);
            }
#ifdef DEX_LP_WSPLIT
            if (p.wl_lo_off) {
                const u16* wl = Wl + p.wl_lo_off + (long)(t * 32 + i) * AHD;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int d0 = 16 * ks + 4 * hh;
                    const uint2 x0 = *reinterpret_cast<const uint2*>(wl + d0), x1 = *reinterpret_cast<const uint2*>(wl + d0 + 8);
                    Frag a; a.u = make_uint4(x0.x, x0.y, x1.x, x1.y);
                    acc = DEX_MFMA_LP(a.v, of[ks].v, acc, 0, 0, 0);
                }
            }
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(myst + i * TVC_SLD + u * 32 + 8 * g + 4 * hh) = make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's stage rows written
        __builtin_amdgcn_wave_barrier();
        float4 sv[8], rv[8];
        float mk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane;
            const int px = id >> 4, c4 = (id & 15) * 4;
            const int pix = row0 + wave * 32 + px;
            const bool ok = pix < p.npix;
            const int pc = ok ? pix : p.npix - 1;
            rv[j] = (TVC_SKIP & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(Xb + (long)pc * p.ldx + h * 64 + c4);
            const float mv = mrow[(pc % p.Wm) * p.mask_ws];
            mk[j] = ok ? mv : 0.f;
            sv[j] = *reinterpret_cast<const float4*>(myst + px * TVC_SLD + c4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane;
            const int px = id >> 4, c4 = (id & 15) * 4;
            const int pix = row0 + wave * 32 + px;
            float4 v;
            v.x = (sv[j].x + rv[j].x) * mk[j]; v.y = (sv[j].y + rv[j].y) * mk[j];
            v.z = (sv[j].z + rv[j].z) * mk[j]; v.w = (sv[j].w + rv[j].w) * mk[j];
            if (pix < p.npix) {
                if (!(TVC_SKIP & 2) || v.x == 12345.f) *reinterpret_cast<float4*>(p.out + ((long)b * p.npix + pix) * AHD + h * 64 + c4) = v;
                gs[h][0] += v.x; gs[h][1] += v.y; gs[h][2] += v.z; gs[h][3] += v.w;
                gq[h][0] = fmaf(v.x, v.x, gq[h][0]); gq[h][1] = fmaf(v.y, v.y, gq[h][1]);
                gq[h][2] = fmaf(v.z, v.z, gq[h][2]); gq[h][3] = fmaf(v.w, v.w, gq[h][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();               // the stage rows are rewritten by the next half
    }
    if (p.stats) {
        const double inv_n = 1.0 / (double)p.npix;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = gs[h][e], q = gq[h][e];
                a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
                a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
                if (lane < 16) {
                    const int ch = h * 64 + lane * 4 + e;
                    gn_add(&red[ch * 2], gn_fix(a, inv_n)); gn_add(&red[ch * 2 + 1], gn_fix(q, inv_n));
                }
            }
        __syncthreads();
        const long long v = red[tid];
        if (v != 0) gn_add(p.stats + (((long)b * AHD + (tid >> 1)) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2 + (tid & 1), v);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// FOLDED form (round 6; bf16 / fp16 modes): w_q and `linear` live inside the style operands (dex_elem.hip tv_fold2_kernel per step,
// tv_vfrag_prep_kernel below per call; dex_api.hip prepare()):
//   scores = (x * mask - mean) . K'^T,   K'[key][k] = rstd[k] / sqrt(C) * sum_n K[key][n] W_q[n][k]      (IN2d and w_q inside the keys)
//   result = mask * (x + softmax(scores) V'),   V'[key][c] = sum_d V[key][d] W_l[c][d]                      (`linear` inside the values)
// so a workgroup runs NO projection: the centred x tile is the score MFMAs' B operand as it lies in LDS, the normalised O^T accumulators
// go straight to the output stage - per wave 64 of 256 MFMAs, both weight stagings (2 x 32 KB per workgroup from L2) and three
// workgroup barriers less.  And the K' / V'^T operands are written by their producers in MFMA FRAGMENT order (a 64-key tile = sixteen
// 1-KB pieces each), so the ring is filled by LDS-DMA (global_load_lds_dwordx4, four + four pieces per wave and tile): no staging
// registers (the register-staged ring above keeps its 8 x 16 bytes per thread in SCRATCH across every tile - 16 + 10 scratch
// instructions in the loop), no ds_write pass, unpadded conflict-free fragment reads.
// LDS: [K'0 | V'^T0 | K'1 | V'^T1], 4 x 16 KB; the x tile and the output stage alias the second half (+ 2.8 KB), so tile 0 lands under
// the x rows.  The residual rows of the first output half are requested under the last tile, the second half's before the first is
// processed (the registers the projection form spends on W_l's prefetch).
static_assert(128 * TVC_SLD * 4 <= 128 * K_LD * 2, "the output stage fits the x tile's bytes");
__global__ __launch_bounds__(256, 2) void tv_chain_fold_kernel(const TvChainP p) {
    // The two ring halves are two DISTINCT static arrays and the tile loop is unrolled over them: hipcc puts an s_waitcnt vmcnt(0) in front
    // of every LDS read that MAY alias an LDS-DMA piece in flight, and with one array (one underlying object) that was the first fragment
    // read behind the next tile's requests - the ring prefetched nothing.  Reads of ring A provably do not alias pieces landing in ring B.
    __shared__ __attribute__((aligned(16))) u16 ringA[2 * TVC_FBUF];                                   // tiles 0, 2, 4: [K' | V'^T]
    __shared__ __attribute__((aligned(16))) u16 ringB[2 * TVC_FBUF > 128 * K_LD ? 2 * TVC_FBUF : 128 * K_LD];   // tiles 1, 3, 5; the x tile / output stage (34.8 KB) alias it
    __shared__ long long red[2 * AHD];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, row0 = blockIdx.x * 128;
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const int ntiles = (TVC_SKIP & 1) ? 0 : (Nk + TVC_KT - 1) / TVC_KT;
    const float* Xb = p.X + (long)b * p.x_bstride + p.x_coff;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const u16* Kg = reinterpret_cast<const u16*>(p.Kp) + (long)b * p.NkPad * AHD;
    const u16* Vg = reinterpret_cast<const u16*>(p.VTp) + (long)b * AHD * p.NkPad;
    red[tid] = 0;
#define TVF_DMA(ring, kt)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
        const int piece = 4 * wave + j;                                                                             \
        __builtin_amdgcn_global_load_lds(Kg + (long)(kt) * TVC_FBUF + piece * 512 + lane * 8, (lds_ptr)(&ring[piece * 512]), 16, 0, 0);              \
        __builtin_amdgcn_global_load_lds(Vg + (long)(kt) * TVC_FBUF + piece * 512 + lane * 8, (lds_ptr)(&ring[TVC_FBUF + piece * 512]), 16, 0, 0);   \
    }
    if (ntiles > 0) { TVF_DMA(ringA, 0) }    // first K' / V'^T tile in flight under the x rows

    // ---- prologue: the 128 x rows, masked and centred (x * mask - mean: what IN2d subtracts; its 1 / std lives in K'), rounded once
    u16* Xs = ringB;                         // [128 pixels][K_LD]
    Frag qf[8];
    {
        const float4 mu = *reinterpret_cast<const float4*>(p.xmean + (long)b * AHD + (tid & 31) * 4);
        if (p.zero_ptr) {                    // the TV input statistics' next use is the next step: cleared here, after their last reader (launch_tv_fold2)
            const long zstep = (long)gridDim.x * gridDim.y * 512;
            for (long zi = ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid) * 2; zi < p.zero_n; zi += zstep)       // (zero_n even, 8-byte aligned)
                *reinterpret_cast<float2*>(p.zero_ptr + zi) = make_float2(0.f, 0.f);
        }
        // all sixteen row chunks of a thread in flight together: one round trip
        float4 xa[16];
        float mk[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int id = tid + 256 * j;
            const int pix = row0 + (id >> 5);
            const bool ok = pix < p.npix;
            const int pc = ok ? pix : p.npix - 1;
            xa[j] = (TVC_SKIP & 4) ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(Xb + (long)pc * p.ldx + (id & 31) * 4);
            const float mv = mrow[(pc % p.Wm) * p.mask_ws];
            mk[j] = ok ? mv : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int id = tid + 256 * j;
            *reinterpret_cast<uint2*>(Xs + (id >> 5) * K_LD + (id & 31) * 4) =
                make_uint2(pack2_lp(fmaf(xa[j].x, mk[j], -mu.x), fmaf(xa[j].y, mk[j], -mu.y)), pack2_lp(fmaf(xa[j].z, mk[j], -mu.z), fmaf(xa[j].w, mk[j], -mu.w)));
        }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks].u = *reinterpret_cast<const uint4*>(Xs + (wave * 32 + i) * K_LD + ks * 16 + hh * 8);
    // (the loop's first barrier - or the one in front of the last tile - separates these reads from tile 1's DMA into the same bytes)

    // ---- attention over the style keys
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    float4 rpre[2][8];
    float mkp[8];
    auto res_load = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane;
            const int px = id >> 4, c4 = (id & 15) * 4;
            const int pix = row0 + wave * 32 + px;
            const int pc = pix < p.npix ? pix : p.npix - 1;
            rpre[h][j] = (TVC_SKIP & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(Xb + (long)pc * p.ldx + h * 64 + c4);
            if (h == 0) { const float mv = mrow[(pc % p.Wm) * p.mask_ws]; mkp[j] = pix < p.npix ? mv : 0.f; }
        }
    };
    int kt = 0;
    while (kt + 1 < ntiles) {                // (tile kt is not the last one; even tiles live in ring A)
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of tile kt have landed (requested a whole tile ago)
        __syncthreads();                     // ... everybody's: tile kt visible; ring B (tile kt - 1 / the x tile) free
        TVF_DMA(ringB, kt + 1)
        tvc_tile<true, false>(ringA, ringA + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
        ++kt;
        if (kt + 1 >= ntiles) break;
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        TVF_DMA(ringA, kt + 1)
        tvc_tile<true, false>(ringB, ringB + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
        ++kt;
    }
#undef TVF_DMA
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    res_load(0);
    if (ntiles > 0) {
        if (kt & 1) tvc_tile<true>(ringB, ringB + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
        else tvc_tile<true>(ringA, ringA + TVC_FBUF, qf, o, m_run, l_run, kt * TVC_KT, Nk, lane);
    }
    l_run += __shfl_xor(l_run, 32);
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    __syncthreads();                         // rings free

    // ---- result: O^T / l through the LDS stage as 256-byte row segments, + residual, * mask, the TIV statistics (as the projection form)
    float* stage = reinterpret_cast<float*>(ringB);          // [128 pixels][TVC_SLD]
    float gs[2][4], gq[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[h][e] = 0.f; gq[h][e] = 0.f; }
    float* myst = stage + wave * 32 * TVC_SLD;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * h + u;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(myst + i * TVC_SLD + u * 32 + 8 * g + 4 * hh) =
                    make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's stage rows written
        __builtin_amdgcn_wave_barrier();
        if (h == 0) res_load(1);
        float4 sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane;
            const int px = id >> 4, c4 = (id & 15) * 4;
            sv[j] = *reinterpret_cast<const float4*>(myst + px * TVC_SLD + c4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane;
            const int px = id >> 4, c4 = (id & 15) * 4;
            const int pix = row0 + wave * 32 + px;
            const float4 rv = rpre[h][j];
            float4 v;
            v.x = (sv[j].x + rv.x) * mkp[j]; v.y = (sv[j].y + rv.y) * mkp[j];
            v.z = (sv[j].z + rv.z) * mkp[j]; v.w = (sv[j].w + rv.w) * mkp[j];
            if (pix < p.npix) {
                if (!(TVC_SKIP & 2) || v.x == 12345.f) *reinterpret_cast<float4*>(p.out + ((long)b * p.npix + pix) * AHD + h * 64 + c4) = v;
                gs[h][0] += v.x; gs[h][1] += v.y; gs[h][2] += v.z; gs[h][3] += v.w;
                gq[h][0] = fmaf(v.x, v.x, gq[h][0]); gq[h][1] = fmaf(v.y, v.y, gq[h][1]);
                gq[h][2] = fmaf(v.z, v.z, gq[h][2]); gq[h][3] = fmaf(v.w, v.w, gq[h][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();               // the stage rows are rewritten by the next half
    }
    if (p.stats) {
        const double inv_n = 1.0 / (double)p.npix;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = gs[h][e], q = gq[h][e];
                a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
                a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
                if (lane < 16) {
                    const int ch = h * 64 + lane * 4 + e;
                    gn_add(&red[ch * 2], gn_fix(a, inv_n)); gn_add(&red[ch * 2 + 1], gn_fix(q, inv_n));
                }
            }
        __syncthreads();
        const long long v = red[tid];
        if (v != 0) gn_add(p.stats + (((long)b * AHD + (tid >> 1)) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2 + (tid & 1), v);
    }
}

// V' (fp32 [B][Nk][C], row 0 = the time token: rewritten per step by tv_fold2_kernel) -> the 16-bit V'^T operand in fragment order:
// VTp[b][tile][t][q][lane (i, hh)][e] = V'[b][key = tile * 64 + (q >> 1) * 32 + (q & 1) * 16 + key_pos(hh * 8 + e)][ch = t * 32 + i], 0 for keys >= Nk
__global__ __launch_bounds__(256) void tv_vfrag_prep_kernel(const TvKvPrepP p) {
    const long n = (long)p.B * (p.NkPad / 64) * 16 * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), q = (int)((idx >> 6) & 3), t = (int)((idx >> 8) & 3);
    const long bt = idx >> 10;
    const int tile = (int)(bt % (p.NkPad / 64)), b = (int)(bt / (p.NkPad / 64));
    const int i = lane & 31, hh = lane >> 5, ch = t * 32 + i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int key = tile * 64 + (q >> 1) * 32 + (q & 1) * 16 + key_pos(hh * 8 + e);
        v[e] = key < p.Nk ? p.V[(long)b * p.kvb + (long)key * AHD + ch] : 0.f;
    }
    reinterpret_cast<uint4*>(p.VTp)[idx] = make_uint4(pack2_lp(v[0], v[1]), pack2_lp(v[2], v[3]), pack2_lp(v[4], v[5]), pack2_lp(v[6], v[7]));
}

// The one-launch TV adaptor takes every grid: measured against the three launches (tools/tv_chain_small_batches.py, 10-step calls, bf16)
// B = 1: +1.3 ... 3.5 % end to end (T = 128 ... 512), B = 2 ... 12: +2.4 ... 5.8 %, B = 32: +4.5 %.  DEX_TV_CHAIN=0: the three launches.
bool tv_chain_form(int npix, int C, int B) {
    (void)npix; (void)B;
    return C == AHD && knob_or("DEX_TV_CHAIN", 1) != 0;
}
void launch_tv_vfrag_prep(const TvKvPrepP& p, hipStream_t st);
void launch_tv_kv_prep(const TvKvPrepP& p, hipStream_t st) {
    const long n = (long)p.B * p.NkPad * (AHD / 8);
    hipLaunchKernelGGL(tv_kv_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    launch_tv_vfrag_prep(p, st);
}
void launch_tv_vfrag_prep(const TvKvPrepP& p, hipStream_t st) {
    const long n = (long)p.B * (p.NkPad / 64) * 16 * 64;
    hipLaunchKernelGGL(tv_vfrag_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
}
void launch_tv_chain(const TvChainP& p, hipStream_t st) {
    const size_t lds = (size_t)TVC_U16 * sizeof(u16);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tv_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const dim3 grid((p.npix + 127) / 128, p.B);
    if (p.xmean) {                     // the folded form (kernels.h TvChainP): static LDS (two ring arrays)
        g_last_symbol = "tv_chain_fold_kernel";
        hipLaunchKernelGGL(tv_chain_fold_kernel, grid, dim3(256), 0, st, p);
        return;
    }
    g_last_symbol = "tv_chain_kernel";
    hipLaunchKernelGGL(tv_chain_kernel, grid, dim3(256), lds, st, p);
}

}  // namespace DEX_LP_NS
}  // namespace dex
