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
constexpr int V_LD = HD + 8;               // V tile [32 keys][128 d + 8]: rows 4 apart land 32 banks apart (the two half-waves of a
                                          // PV step read keys 4 apart), 544-B rows keep ds_write_b128 aligned
constexpr int WAVE_LDS = 32 * V_LD;       // 4352 floats >= HD * KT_LD (4224) >= 32 * O_LD: K^T, then V, then the merged O tile
constexpr int Q_LD = HD + 1;              // shared Q tile row stride (odd: conflict-free column reads)

// QREG: the pre-scaled Q fragments live in 64 VGPRs instead of a shared LDS tile — the 4-wave variant then needs
// 70 KB of LDS and 2 workgroups share a CU, so one workgroup's prologue / partial merge hides under the other's MFMAs.
template <int NW, bool QREG>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(QREG ? 2 : 1, 8))) void attn_f32_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    float* kT = smem + wave * WAVE_LDS;
    float* stat = smem + NW * WAVE_LDS;                     // [NW][2][32]
    float* qS = stat + NW * 64;                             // [32 queries][Q_LD]  pre-scaled Q tile, shared by all waves (!QREG)
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const float* Qb = p.Q + (long)b * p.qb + h * HD;
    const float* Kb = p.K + (long)b * p.kb + h * HD;
    const float* Vb = p.V + (long)b * p.vb + h * HD;

    float qreg[QREG ? 64 : 1];
    if constexpr (QREG) {
        // B operand of S^T: lane (query i, half hh) holds Q[q0 + i][hh*64 + kk] * scale, kk = 0..63
        const float sc = q0 + i < p.Nq ? p.scale : 0.f;
        const float* qp = Qb + (long)min(q0 + i, p.Nq - 1) * p.ldq + hh * 64;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + k4 * 4);
            qreg[k4 * 4 + 0] = v.x * sc; qreg[k4 * 4 + 1] = v.y * sc; qreg[k4 * 4 + 2] = v.z * sc; qreg[k4 * 4 + 3] = v.w * sc;
        }
    } else {
        // Q tile -> LDS (the B operand of S^T is one ds_read_b32 per MFMA)
        for (int it = tid; it < 32 * (HD / 4); it += NW * 64) {
            const int qi = it / (HD / 4), d4 = (it % (HD / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + qi < p.Nq) v = *reinterpret_cast<const float4*>(Qb + (long)(q0 + qi) * p.ldq + d4);
            float* d = qS + qi * Q_LD + d4;
            d[0] = v.x * p.scale; d[1] = v.y * p.scale; d[2] = v.z * p.scale; d[3] = v.w * p.scale;
        }
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
    // One 64-register staging buffer carries K, then V, then the next K: the V loads fly under the S^T MFMAs, the
    // next K loads under the PV MFMAs.  Both tiles go through the wave-private LDS region (K transposed, V as is):
    // reading V straight from global, one dword per lane per MFMA, made the PV phase 2.7x slower than S^T.
    typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vectors: arrays of HIP's float4 struct can defeat SROA
    f32x4 kreg[16];
    // gather map of a 32x128 tile: per instruction a 32-lane half covers 4 rows x 32 d (128-B coalesced row segments)
    {   // (unconditional: a wave without tiles loads a clamped tile it never uses — keeps the array in registers)
        const int k0 = min(wave, max(ntiles - 1, 0)) * 32;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            kreg[it] = *reinterpret_cast<const f32x4*>(Kb + (long)min(k0 + key, Nk - 1) * p.ldk + dd);
        }
    }
    __syncthreads();                                        // Q tile visible
#ifdef DEX_TIMING
    long long tA = 0, tB = 0, tC = 0, tD = 0, t0_ = clock64(), w0_ = wall_clock64();
#endif
    for (int kt = wave; kt < ntiles; kt += NW) {
        const int k0 = kt * 32;
        const int kn = min(kt + NW, ntiles - 1) * 32;       // next tile of this wave (clamped re-read at the tail)
#ifdef DEX_TIMING
        long long c0 = clock64();
#endif
        // K -> K^T[d][key] (bank = (d + key) % 32: conflict-free writes and reads)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            float* d = kT + dd * KT_LD + key;
            d[0] = kreg[it].x; d[KT_LD] = kreg[it].y; d[2 * KT_LD] = kreg[it].z; d[3 * KT_LD] = kreg[it].w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes landed (wave-private tile)
        __builtin_amdgcn_wave_barrier();
        // V tile of the same keys into the staging registers, behind the S^T MFMAs
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            kreg[it] = *reinterpret_cast<const f32x4*>(Vb + (long)min(k0 + key, Nk - 1) * p.ldv + dd);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DEX_TIMING
        long long c1 = clock64(); tA += c1 - c0;
#endif
        // ---- S^T[key][query] = sum_d K[key][d] * Qs[query][d]
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = 0.f;
        const float* ka = kT + (hh * 64) * KT_LD + i;
        const float* qa = qS + i * Q_LD + hh * 64;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk)
            sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk * KT_LD], QREG ? qreg[kk] : qa[kk], sT, 0, 0, 0);
#ifdef DEX_TIMING
        asm volatile("s_nop 0" :: "v"(sT[0]), "v"(sT[15])); long long c2 = clock64(); tB += c2 - c1;
#endif
        // ---- online softmax over this lane's 16 keys (+ partner half via xor 32)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Nk) sT[r] = -INFINITY;
            mx = fmaxf(mx, sT[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // lazy rescale: the reference maximum moves only when some query exceeds it by more than 8 (P <= e^8 stays exact in
        // fp32; the partial merge and the final division use the same reference, so the result is unchanged)
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);      // first tile: exp(-inf) = 0
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = __expf(sT[r] - m_run); psum += sT[r]; }
        l_run += psum;
        // V -> LDS as V[key][d] over the K^T image (S^T is done with it), 16-byte stores
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            *reinterpret_cast<f32x4*>(kT + key * V_LD + dd) = kreg[it];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int key = (it >> 2) * 8 + hh * 4 + (i >> 3);
            const int dd = (it & 3) * 32 + (i & 7) * 4;
            kreg[it] = *reinterpret_cast<const f32x4*>(Kb + (long)min(kn + key, Nk - 1) * p.ldk + dd);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DEX_TIMING
        asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(sT[15])); long long c3 = clock64(); tC += c3 - c2;
#endif
        // ---- O^T[d][query] += sum_key V[key][d] * P[query][key];  step s uses key(s,hh) on both operands
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* vp = kT + ((s & 3) + 8 * (s >> 2) + 4 * hh) * V_LD + i;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * 32], sT[s], o[t], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#ifdef DEX_TIMING
        asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[3][15])); tD += clock64() - c3;
#endif
    }
#ifdef DEX_TIMING
    if (p.dbg && lane == 0 && wave == 0) { long long* d = p.dbg + ((long)blockIdx.x + gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) * 8; d[0] = tA; d[1] = tB; d[2] = tC; d[3] = tD; d[4] = clock64() - t0_; d[5] = wall_clock64() - w0_; }
#endif
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

void launch_attention_lp(const AttnP& p, int precision, hipStream_t st);

// ------------------------------------------------------------------------------------------------------------------
// Generic head_dim (a multiple of 64, <= 256): DEX-TTS/config/LibriTTS/base.yaml has hidden 384 = 2 x 192 in the DiT and a
// 256-channel TVAdaptor.  Same transposed formulation, walked in 64-wide slices of the head dimension so the wave-private
// LDS tile stays 9 KB whatever the head_dim: S^T accumulates over the slices of K, O^T keeps HD/32 accumulator tiles, the
// NW partials are merged slice by slice.  No software pipelining — this shape is a correctness path, not a tuned one.
constexpr int GC = 64;                      // head-dim slice
constexpr int GKT_LD = 33;                  // K^T slice [64 d][32 keys + 1]
constexpr int GV_LD = GC + 8;               // V slice [32 keys][64 d + 8];  also the merged O slice [32 queries][64 d + 8]
constexpr int GWAVE_LDS = 32 * GV_LD;       // 2304 floats >= 64 * 33
template <int HDG, int NW>
__global__ __launch_bounds__(NW * 64) void attn_f32_generic_kernel(const AttnP p) {
    constexpr int NCH = HDG / GC, QLD = HDG + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    float* wt = smem + wave * GWAVE_LDS;                    // wave-private tile
    float* stat = smem + NW * GWAVE_LDS;                    // [NW][2][32]
    float* qS = stat + NW * 64;                             // [32 queries][QLD] pre-scaled Q
    int Nk = p.Nk;
    if (p.kv_len) Nk = min(p.Nk, p.kv_len[b] + p.kv_len_add);
    const float* Qb = p.Q + (long)b * p.qb + h * HDG;
    const float* Kb = p.K + (long)b * p.kb + h * HDG;
    const float* Vb = p.V + (long)b * p.vb + h * HDG;
    for (int it = tid; it < 32 * (HDG / 4); it += NW * 64) {
        const int qi = it / (HDG / 4), d4 = (it % (HDG / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + qi < p.Nq) v = *reinterpret_cast<const float4*>(Qb + (long)(q0 + qi) * p.ldq + d4);
        float* d = qS + qi * QLD + d4;
        d[0] = v.x * p.scale; d[1] = v.y * p.scale; d[2] = v.z * p.scale; d[3] = v.w * p.scale;
    }
    f32x16 o[HDG / 32];
#pragma unroll
    for (int t = 0; t < HDG / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (Nk + 31) / 32;
    __syncthreads();
    for (int kt = wave; kt < ntiles; kt += NW) {
        const int k0 = kt * 32;
        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = 0.f;
        for (int c = 0; c < NCH; ++c) {
            // K slice [32 keys][64 d] -> K^T[d][key]: 8 float4 per lane, a 32-lane half covers 2 keys x 64 d... (16 lanes per key row)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int key = it * 4 + (lane >> 4), dd = (lane & 15) * 4;
                const float4 v = *reinterpret_cast<const float4*>(Kb + (long)min(k0 + key, Nk - 1) * p.ldk + c * GC + dd);
                float* d = wt + dd * GKT_LD + key;
                d[0] = v.x; d[GKT_LD] = v.y; d[2 * GKT_LD] = v.z; d[3 * GKT_LD] = v.w;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const float* ka = wt + (hh * 32) * GKT_LD + i;
            const float* qa = qS + i * QLD + c * GC + hh * 32;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kk * GKT_LD], qa[kk], sT, 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Nk) sT[r] = -INFINITY;
            mx = fmaxf(mx, sT[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {         // lazy rescale, as in the tuned kernel
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < HDG / 32; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = __expf(sT[r] - m_run); psum += sT[r]; }
        l_run += psum;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int key = it * 4 + (lane >> 4), dd = (lane & 15) * 4;
                *reinterpret_cast<float4*>(wt + key * GV_LD + dd) = *reinterpret_cast<const float4*>(Vb + (long)min(k0 + key, Nk - 1) * p.ldv + c * GC + dd);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float* vp = wt + ((s & 3) + 8 * (s >> 2) + 4 * hh) * GV_LD + i;
#pragma unroll
                for (int t = 0; t < 2; ++t) o[c * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * 32], sT[s], o[c * 2 + t], 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 32);
    if (hh == 0) { stat[(wave * 2 + 0) * 32 + i] = m_run; stat[(wave * 2 + 1) * 32 + i] = l_run; }
    // merge the NW partials, one 64-wide slice of the head dimension at a time
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();                                    // previous slice consumed / every wave done with its tile
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(wt + i * GV_LD + t * 32 + 8 * rq + 4 * hh) =
                    make_float4(o[c * 2 + t][rq * 4 + 0], o[c * 2 + t][rq * 4 + 1], o[c * 2 + t][rq * 4 + 2], o[c * 2 + t][rq * 4 + 3]);
        __syncthreads();
        const int d4 = (tid & 15) * 4;
        for (int q = tid >> 4; q < 32; q += NW * 4) {
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
                const float4 v = *reinterpret_cast<const float4*>(smem + w * GWAVE_LDS + q * GV_LD + d4);
                acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
            }
            const float inv = 1.f / L;
            float* op = p.O + (long)b * p.ob + (long)(q0 + q) * p.ldo + h * HDG + c * GC + d4;
            *reinterpret_cast<float4*>(op) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        }
    }
}
template <int HDG>
static void launch_attn_generic(const AttnP& p, hipStream_t st) {
    constexpr int NW = 4;
    const size_t lds = (size_t)(NW * GWAVE_LDS + NW * 64 + 32 * (HDG + 1)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_generic_kernel<HDG, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.Nq + 31) / 32, p.heads, p.B);
    hipLaunchKernelGGL((attn_f32_generic_kernel<HDG, NW>), grid, dim3(NW * 64), lds, st, p);
}
bool attention_head_dim_supported(int hd) { return hd == 64 || hd == 128 || hd == 192 || hd == 256; }
   // lp_dispatch.hip -> attention_bf16.hip (bf16 / fp16 build)

template <int NW, bool QREG>
static void launch_attn_nw(const AttnP& p, hipStream_t st) {
    const size_t lds = (size_t)(NW * WAVE_LDS + NW * 64 + (QREG ? 0 : 32 * Q_LD)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel<NW, QREG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.Nq + 31) / 32, p.heads, p.B);
    hipLaunchKernelGGL((attn_f32_kernel<NW, QREG>), grid, dim3(NW * 64), lds, st, p);
}

void launch_attention(const AttnP& p, int precision, hipStream_t st) {
    const int hd = p.head_dim ? p.head_dim : HD;
    if (hd != HD || p.force_generic) {          // other head dims (DEX-LibriTTS: 192 / 256): the generic fp32 kernel in every mode
        if (hd == 64) launch_attn_generic<64>(p, st);
        else if (hd == 128) launch_attn_generic<128>(p, st);
        else if (hd == 192) launch_attn_generic<192>(p, st);
        else launch_attn_generic<256>(p, st);
        return;
    }
    if (prec_is_lp(precision)) { launch_attention_lp(p, precision, st); return; }
    const long blocks = (long)((p.Nq + 31) / 32) * p.heads * p.B;
    const int ntiles = (p.Nk + 31) / 32;
    // Many query tiles (batch): 4 key-splitting waves with Q in registers, two workgroups per CU (2 waves per SIMD as
    // well, but one workgroup's prologue and partial merge overlap the other's MFMA loop).  Few query tiles: 8 waves
    // per workgroup split the keys further; never more waves than key tiles.
    if (blocks > 256 && ntiles >= 8) { launch_attn_nw<4, true>(p, st); return; }
    int nw = 8;
    if (ntiles < 8) nw = 4;
    if (ntiles < 4) nw = 2;
    if (nw == 8) launch_attn_nw<8, false>(p, st);
    else if (nw == 4) launch_attn_nw<4, false>(p, st);
    else launch_attn_nw<2, false>(p, st);
}

}  // namespace dex
