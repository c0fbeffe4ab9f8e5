// attention_direct.hip — softmax attention of the DiT blocks (timm Attention core, reference model/dit.py:73-76)
// on operands the producer already laid out for the matrix cores (bf16-MFMA mode, 2 heads x 128).
//
// The row-chain kernel (dit_rowchain.hip) writes q (pre-scaled), k as bf16 [B][2][N][128] and v TRANSPOSED as bf16
// [B][2][128][Npad] with the key index bits 2/3 swapped inside each 32-key block.  With those layouts every MFMA
// operand of the transposed flash formulation is one 16-byte global load per lane:
//     S^T[key][query] = K Q^T      A = K rows  (lane = key,  8 consecutive d)     B = Q rows (lane = query)
//     O^T[d][query]  += V^T P^T    A = V^T rows (lane = d,   8 consecutive key positions)
//                                  B = P^T, taken from the S^T accumulator registers in place: a lane holds keys
//                                  (j&3) + 8*(2*k2 + (j>>2)) + 4*hh for slot j of K-step k2 — exactly the
//                                  bit-2/3-swapped position order v^T was stored in.
// No LDS staging, no fp32->bf16 conversion, no transposes: a wave's K tile is 32 VGPRs and its V^T tile 32 VGPRs,
// so the next tile's loads are issued at the point of last use of the current registers and fly under the MFMAs.
// NW waves share one 32-query tile and split the key tiles (and, at small batch, KSPLIT workgroups split them
// further: one or two tiles per wave = one global round trip); the waves' partials are merged through LDS and
// each split writes normalised O plus (max, sum) for the consumer to merge.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "bf16_util.h"

namespace dex {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {
constexpr int HD = 128, O_LD = 132, NWD = 8;
union DFrag { uint4 u; bf16x8 v; };
}  // namespace

__global__ __launch_bounds__(NWD * 64) void attn_direct_kernel(const AttnDirectP p) {
    extern __shared__ __attribute__((aligned(16))) float smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    const int N = p.N;
    const long hb = (long)b * 2 + h;
    const u16* Qg = reinterpret_cast<const u16*>(p.Qh) + hb * N * HD;
    const u16* Kg = reinterpret_cast<const u16*>(p.Kh) + hb * N * HD;
    const u16* Vg = reinterpret_cast<const u16*>(p.Vt) + hb * HD * p.Npad;
#ifdef DEX_TIMING
    long long tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tst[0] = wall_clock64();
#endif
    const int ntiles = (N + 31) / 32;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    int kt = t_lo + wave;

    // all first-round loads back to back: Q fragments, then this wave's first K and V^T tiles
    DFrag qf[8], kf[8], vf[4][2];
    {
        const u16* qp = Qg + (long)min(q0 + i, N - 1) * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks].u = *reinterpret_cast<const uint4*>(qp + ks * 16);
    }
    if (kt < t_hi) {
        const u16* kp = Kg + (long)min(kt * 32 + i, N - 1) * HD + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks].u = *reinterpret_cast<const uint4*>(kp + ks * 16);
        const u16* vp = Vg + (long)i * p.Npad + kt * 32 + hh * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = *reinterpret_cast<const uint4*>(vp + (long)t * 32 * p.Npad + k2 * 16);
    }
    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
#ifdef DEX_TIMING
    tst[1] = wall_clock64();
#endif
    while (kt < t_hi) {
        const int k0 = kt * 32, kn = kt + NWD;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks].v, qf[ks].v, s, 0, 0, 0);
        if (kn < t_hi) {                                   // next K tile into the registers just consumed
            const u16* kp = Kg + (long)min(kn * 32 + i, N - 1) * HD + hh * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks].u = *reinterpret_cast<const uint4*>(kp + ks * 16);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= N) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_new); psum += s[r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pb;
            pb.u.x = pack2_bf16(s[8 * k2 + 0], s[8 * k2 + 1]); pb.u.y = pack2_bf16(s[8 * k2 + 2], s[8 * k2 + 3]);
            pb.u.z = pack2_bf16(s[8 * k2 + 4], s[8 * k2 + 5]); pb.u.w = pack2_bf16(s[8 * k2 + 6], s[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t][k2].v, pb.v, o[t], 0, 0, 0);
        }
        if (kn < t_hi) {                                   // next V^T tile
            const u16* vp = Vg + (long)i * p.Npad + kn * 32 + hh * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = *reinterpret_cast<const uint4*>(vp + (long)t * 32 * p.Npad + k2 * 16);
        }
        kt = kn;
    }
    l_run += __shfl_xor(l_run, 32);
#ifdef DEX_TIMING
    asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[3][15])); tst[4] = wall_clock64();
#endif
    // ---- merge the NW partials through LDS: oS[wave][query][d], stat[wave][2][32]
    float* oS = smem_d + wave * 32 * O_LD;
    float* stat = smem_d + NWD * 32 * O_LD;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<float4*>(oS + i * O_LD + t * 32 + 8 * rq + 4 * hh) =
                make_float4(o[t][rq * 4 + 0], o[t][rq * 4 + 1], o[t][rq * 4 + 2], o[t][rq * 4 + 3]);
    if (hh == 0) { stat[(wave * 2 + 0) * 32 + i] = m_run; stat[(wave * 2 + 1) * 32 + i] = l_run; }
    __syncthreads();
#ifdef DEX_TIMING
    tst[5] = wall_clock64();
#endif
    const int d4 = (tid & 31) * 4;
#pragma unroll
    for (int q = tid >> 5; q < 32; q += NWD * 2) {
        if (q0 + q >= N) continue;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWD; ++w) M = fmaxf(M, stat[(w * 2) * 32 + q]);
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NWD; ++w) {
            const float mw = stat[(w * 2) * 32 + q];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            L += f * stat[(w * 2 + 1) * 32 + q];
            const float4 v = *reinterpret_cast<const float4*>(smem_d + (w * 32 + q) * O_LD + d4);
            acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
        }
        const float inv = L > 0.f ? 1.f / L : 0.f;
        float* op = p.O + (long)sp * p.o_sstride + ((long)b * N + q0 + q) * (2 * HD) + h * HD + d4;
        *reinterpret_cast<float4*>(op) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (p.ml && d4 == 0) {
            float* ml = p.ml + ((((long)sp * p.B + b) * 2 + h) * N + q0 + q) * 2;
            ml[0] = M; ml[1] = L;
        }
    }
#ifdef DEX_TIMING
    if (p.dbg && lane == 0 && (wave == 0 || wave == NWD - 1)) {
        tst[6] = wall_clock64();
        long long* d = p.dbg + (((long)blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)) * 2 + (wave ? 1 : 0)) * 8;
        for (int k = 0; k < 8; ++k) d[k] = tst[k];
    }
#endif
}

void launch_attention_direct(const AttnDirectP& p, hipStream_t st) {
    const size_t lds = (size_t)(NWD * 32 * O_LD + NWD * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid((p.N + 31) / 32, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL(attn_direct_kernel, grid, dim3(NWD * 64), lds, st, p);
}

}  // namespace dex
