// attention.hip — softmax attention core, head_dim 128 (timm Attention inside DiTBlock, dit.py:276,288;
// TVAdaptor cross-attention, ref_encoder.py:166-174).  No score matrix is materialised.
//
// fp32 mode, "transposed" flash formulation on v_mfma_f32_32x32x2_f32 so that every per-query quantity is
// lane-local:   S^T = K Q^T   (A = K tile from LDS, B = Q^T kept in registers, pre-scaled)
//               O^T = V^T P^T (A = V read straight from global — coalesced along d, B = P^T = the S^T
//                              accumulator registers themselves, no data movement: the contraction order
//                              over keys is permuted identically for A and B)
// C-layout of a 32x32 tile: col = lane&31 (query), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// A workgroup = NW waves sharing one 32-query tile; wave w takes key tiles w, w+NW, ...; partial (m, l, O)
// are merged through LDS at the end (split-KV), which keeps small token counts (N=650 at B=1) spread over CUs.
#include "kernels.h"

namespace dex {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD = 128;
constexpr int KT_LD = 33;                 // K^T tile [128 d][32 keys + 1]
constexpr int O_LD = 132;                 // merged O tile [32 queries][128 d + 4]
constexpr int WAVE_LDS = HD * KT_LD;      // 4224 floats == 32 * 132
constexpr int Q_LD = HD + 1;              // shared Q tile row stride (odd: conflict-free column reads)

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_f32_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    float* kT = smem + wave * WAVE_LDS;
    float* stat = smem + NW * WAVE_LDS;                     // [NW][2][32]
    float* qS = stat + NW * 64;                             // [32 queries][Q_LD]  pre-scaled Q tile, shared by all waves
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const float* Qb = p.Q + (long)b * p.qb + h * HD;
    const float* Kb = p.K + (long)b * p.kb + h * HD;
    const float* Vb = p.V + (long)b * p.vb + h * HD;

    // Q tile -> LDS (keeps 64 VGPRs free for the K prefetch; the B operand of S^T is one ds_read_b32 per MFMA)
    for (int it = tid; it < 32 * (HD / 4); it += NW * 64) {
        const int qi = it / (HD / 4), d4 = (it % (HD / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + qi < p.Nq) v = *reinterpret_cast<const float4*>(Qb + (long)(q0 + qi) * p.ldq + d4);
        float* d = qS + qi * Q_LD + d4;
        d[0] = v.x * p.scale; d[1] = v.y * p.scale; d[2] = v.z * p.scale; d[3] = v.w * p.scale;
    }
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // l_run: this lane's half of the row sum

    const int ntiles = (Nk + 31) / 32;
    // K tile gather map: per instruction a 32-lane half covers 4 keys x 32 d (128-B coalesced rows); the LDS image
    // is K^T[d][key] with bank = (d + key) % 32 -> conflict-free writes and reads.
    float4 kreg[16];
    auto kload = [&](int kt) {
        const int k0 = kt * 32;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            kreg[it] = *reinterpret_cast<const float4*>(Kb + (long)min(k0 + key, Nk - 1) * p.ldk + dd);
        }
    };
    if (wave < ntiles) kload(wave);
    __syncthreads();                                        // Q tile visible
    for (int kt = wave; kt < ntiles; kt += NW) {
        const int k0 = kt * 32;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            float* d = kT + dd * KT_LD + key;
            d[0] = kreg[it].x; d[KT_LD] = kreg[it].y; d[2 * KT_LD] = kreg[it].z; d[3 * KT_LD] = kreg[it].w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes landed (wave-private tile)
        __builtin_amdgcn_wave_barrier();
        if (kt + NW < ntiles) kload(kt + NW);               // prefetch the next tile of this wave behind the MFMAs
        // ---- S^T[key][query] = sum_d K[key][d] * Qs[query][d]
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = 0.f;
        const float* ka = kT + (hh * 64) * KT_LD + i;
        const float* qa = qS + i * Q_LD + hh * 64;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk)
            sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk * KT_LD], qa[kk], sT, 0, 0, 0);
        // ---- online softmax over this lane's 16 keys (+ partner half via xor 32)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Nk) sT[r] = -INFINITY;
            mx = fmaxf(mx, sT[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);          // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = __expf(sT[r] - m_new); psum += sT[r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // ---- O^T[d][query] += sum_key V[key][d] * P[query][key];  step s uses key(s,hh) on both operands
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int key = min(k0 + (s & 3) + 8 * (s >> 2) + 4 * hh, Nk - 1);
            const float* vp = Vb + (long)key * p.ldv + i;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * 32], sT[s], o[t], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- merge the NW partials through LDS (reuse the K^T region as O[query][d])
    l_run += __shfl_xor(l_run, 32);
    __syncthreads();                                       // every wave is done with its K^T tile
    float* oS = smem + wave * WAVE_LDS;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int d = t * 32 + 8 * rq + 4 * hh;
            *reinterpret_cast<float4*>(oS + i * O_LD + d) =
                make_float4(o[t][rq * 4 + 0], o[t][rq * 4 + 1], o[t][rq * 4 + 2], o[t][rq * 4 + 3]);
        }
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
            const float4 v = *reinterpret_cast<const float4*>(smem + w * WAVE_LDS + q * O_LD + d4);
            acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
        }
        const float inv = 1.f / L;
        float* op = p.O + (long)b * p.ob + (long)(q0 + q) * p.ldo + h * HD + d4;
        *reinterpret_cast<float4*>(op) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

void launch_attention_bf16(const AttnP& p, hipStream_t st);   // attention_bf16.hip

template <int NW>
static void launch_attn_nw(const AttnP& p, hipStream_t st) {
    const size_t lds = (size_t)(NW * WAVE_LDS + NW * 64 + 32 * Q_LD) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.Nq + 31) / 32, p.heads, p.B);
    hipLaunchKernelGGL((attn_f32_kernel<NW>), grid, dim3(NW * 64), lds, st, p);
}

void launch_attention(const AttnP& p, int precision, hipStream_t st) {
    if (precision == 1) { launch_attention_bf16(p, st); return; }
    const long blocks = (long)((p.Nq + 31) / 32) * p.heads * p.B;
    const int ntiles = (p.Nk + 31) / 32;
    // enough waves to cover ~1024 SIMDs, but never more waves than key tiles
    // 8 waves per workgroup = 2 per SIMD: one wave's K staging / softmax hides behind the other's MFMA chain
    // (measured at B=32 N=1300: 69 -> 78 TF/s vs 2 or 4 waves at one wave per SIMD)
    int nw = 8;
    if (ntiles < 8) nw = 4;
    if (ntiles < 4) nw = 2;
    (void)blocks;
    if (nw == 8) launch_attn_nw<8>(p, st);
    else if (nw == 4) launch_attn_nw<4>(p, st);
    else launch_attn_nw<2>(p, st);
}

}  // namespace dex
