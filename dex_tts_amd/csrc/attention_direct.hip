// attention_direct.hip — softmax attention of the DiT blocks (timm Attention core, reference model/dit.py:73-76)
// on operands the producer already laid out for the matrix cores (bf16-MFMA mode, 2 heads x 128).
//
// The row-chain kernel (dit_rowchain.hip) writes q (pre-scaled), k and v^T as bf16 in MFMA FRAGMENT order per
// (batch, head): q/k as [32-row tile][K-step ks][lane = hh*32 + row][8 d], v^T as [32-key tile][d tile t][K-step k2]
// [lane = hh*32 + d][8 key positions], key positions with index bits 2/3 swapped inside each 32-key block.  Every
// MFMA operand of the transposed flash formulation is then one 16-byte load per lane and every wave-level load
// instruction reads 1 KB of CONTIGUOUS memory (the first version used row-major q/k: each instruction touched 32
// rows x 32 B and L1 lines were evicted between the 4 instructions that share them):
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
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {
constexpr int HD = 128, O_LD = 132, NWD = 8;
union DFrag { uint4 u; lp8 v; };
}  // namespace

__global__ __launch_bounds__(NWD * 64) void attn_direct_kernel(const AttnDirectP p) {
    extern __shared__ __attribute__((aligned(16))) float smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z / ksplit, sp = blockIdx.z % ksplit;
    const int N = p.N;
    const long hb = (long)b * 2 + h;
    const uint4* Qg = reinterpret_cast<const uint4*>(p.Qh) + hb * p.Npad * (HD / 8) + lane;    // fragment-tiled: [tile][ks][lane]
    const uint4* Kg = reinterpret_cast<const uint4*>(p.Kh) + hb * p.Npad * (HD / 8) + lane;
    const uint4* Vg = reinterpret_cast<const uint4*>(p.Vt) + hb * p.Npad * (HD / 8) + lane;    // [tile][t][k2][lane]
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
        const uint4* qp = Qg + (long)(q0 >> 5) * 8 * 64;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks].u = qp[ks * 64];
    }
    if (kt < t_hi) {
        const uint4* kp = Kg + (long)kt * 8 * 64;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
        const uint4* vp = Vg + (long)kt * 8 * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
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
        for (int ks = 0; ks < 8; ++ks) s = DEX_MFMA_LP(kf[ks].v, qf[ks].v, s, 0, 0, 0);
        if (kn < t_hi) {                                   // next K tile into the registers just consumed
            const uint4* kp = Kg + (long)kn * 8 * 64;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks].u = kp[ks * 64];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= N) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.f) != 0) {     // lazy rescale (see attn_direct_ring_kernel)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);        // scores are in the log2 domain (q scale carries log2 e)
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
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            DFrag pb;
            pb.u.x = pack2_lp(s[8 * k2 + 0], s[8 * k2 + 1]); pb.u.y = pack2_lp(s[8 * k2 + 2], s[8 * k2 + 3]);
            pb.u.z = pack2_lp(s[8 * k2 + 4], s[8 * k2 + 5]); pb.u.w = pack2_lp(s[8 * k2 + 6], s[8 * k2 + 7]);
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = DEX_MFMA_LP(vf[t][k2].v, pb.v, o[t], 0, 0, 0);
        }
        if (kn < t_hi) {                                   // next V^T tile
            const uint4* vp = Vg + (long)kn * 8 * 64;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) vf[t][k2].u = vp[(t * 2 + k2) * 64];
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
            const float f = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
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

// ---- batch regime: 4 waves = 4 consecutive 32-query tiles of one (batch, head) walk the key tiles together; K and V^T
// tiles are shared through 3-slot LDS rings that keep the fragment order of the global layout and are filled by LDS-DMA
// (global_load_lds_dwordx4: a fragment-ordered tile is 8 contiguous 1-KB pieces, two per wave) TWO tiles ahead: K(kt+3)
// goes into the slot K(kt) left after iteration kt-1, V(kt+2) into V(kt-1)'s, so a DMA group has two iterations to land
// and the loop waits with vmcnt(4) for the previous group only.  ONE barrier per key tile.  161 VGPRs, no AGPR traffic:
// three waves per SIMD (the first version staged through registers, used 256 VGPRs + accvgpr moves and ran two).
// Inside a wave the S^T MFMAs of tile kt+1 are issued before the softmax VALU work of tile kt; the S^T chain is ONE
// accumulator (a dependent 32x32x16 chain issues back to back; the two-accumulator form cost 16 adds and 16 registers).
// Scores are in the log2 domain (log2 e is folded into the q scale by the producer): p = exp2(s - m); the running-max
// rescale of O is lazy (see the loop).  Optional key split (blockIdx.z = b * ksplit + sp) for grids that would leave CUs
// idle or unevenly loaded: split sp walks key tiles [t_lo, t_hi) and writes its normalised O and (m, l); the row chain
// merges the partials.
// Measured on MI355X (tools/attnbench.hip, B=32 N=1300): 0.255 of the nominal 2.5 PF vs 0.20 before.  The same loop with
// the MFMAs alone (no LDS reads, no softmax, no refill) reaches 0.42: with random operands the chip clocks down to
// ~1.5 GHz under dense bf16 MFMA load (clock64 / wall_clock64 inside the kernel), and 704 workgroups on 256 CUs lose 8 %.
__global__ __launch_bounds__(256, 3) void attn_direct_ring_kernel(const AttnDirectP p) {
    __shared__ __attribute__((aligned(16))) uint4 kS[3][512];
    __shared__ __attribute__((aligned(16))) uint4 vS[3][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    // p.xcd_map (launcher: 1-D grid, 2 B ksplit a multiple of 8): workgroup L runs on XCD L % 8; all query groups of one
    // (batch element, split, head) go to ONE XCD, so its K and V^T (665 KB at N = 1300) are fetched into one L2 instead of all
    // eight (PMC at DEX B = 32: 411 MB per launch against ~ 90 MB algorithmic before)
    int h, z, qg;
    if (p.xcd_map) {
        const int nq = (ntiles + 3) / 4, slot = (int)blockIdx.x >> 3, g = ((int)blockIdx.x & 7) + 8 * (slot / nq);
        qg = slot % nq; h = g & 1; z = g >> 1;
    } else { qg = blockIdx.x; h = blockIdx.y; z = blockIdx.z; }
    const int b = z / ksplit, sp = z % ksplit;
    const int t_lo = (int)((long)ntiles * sp / ksplit), t_hi = (int)((long)ntiles * (sp + 1) / ksplit);
    const int qt = min(qg * 4 + wave, ntiles - 1);
    const bool live_wave = qg * 4 + wave < ntiles;
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
        // Lazy rescale: the reference maximum only moves when some query's true maximum exceeds it by more than 2^8
        // (scores are log2-domain), so P = 2^(s - m_ref) <= 256 stays exact in fp32 sums and well inside bf16 / fp16 range,
        // the result is mathematically unchanged (the final division uses the same reference), and the 64-register
        // rescale of O — which with 32 queries per wave fired on nearly every tile — runs a handful of times.
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
        if (p.o_lp) {          // (uniform; ksplit == 1) the projection GEMM of the row chain rounds O to the operand type: store it so
            unsigned short* oh = reinterpret_cast<unsigned short*>(p.O) + ((long)b * N + q0 + i) * (2 * HD) + h * HD + 4 * hh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<uint2*>(oh + t * 32 + 8 * rq) =
                        make_uint2(pack2_lp(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv), pack2_lp(o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<float4*>(op + t * 32 + 8 * rq) =
                        make_float4(o[t][rq * 4 + 0] * inv, o[t][rq * 4 + 1] * inv, o[t][rq * 4 + 2] * inv, o[t][rq * 4 + 3] * inv);
        }
        if (p.ml && hh == 0) {
            float* ml = p.ml + ((((long)sp * p.B + b) * 2 + h) * N + q0 + i) * 2;
            ml[0] = m_run; ml[1] = l_run;
        }
    }
}

// Many (query tile, head, batch) items: the shared-ring kernel; few: 8 key-splitting waves per 32-query tile.
bool attention_direct_batch_regime(int N, int B) { return (long)((N + 31) / 32) * 2 * B >= 1024; }
// Key split of the batch regime: enough workgroups to load 256 CUs evenly (>= 2 per CU), never below 8 key tiles per split.
int attention_direct_ksplit(int N, int B) {
    const int ntiles = (N + 31) / 32;
    const long wgs = (long)((ntiles + 3) / 4) * 2 * B;
    int ks = 1;
    while (ks < 4 && wgs * ks < 384 && ntiles / (ks * 2) >= 8) ks *= 2;       // (the consumer pays for every extra partial it merges)
    return ks;
}

void launch_attention_direct(const AttnDirectP& p, hipStream_t st) {
    const size_t lds = (size_t)(NWD * 32 * O_LD + NWD * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (attention_direct_batch_regime(p.N, p.B)) {          // batch regime (the caller sized ksplit with attention_direct_ksplit)
        dim3 grid(((p.N + 31) / 32 + 3) / 4, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
        AttnDirectP q = p;
        q.xcd_map = ((2 * grid.z) % 8 == 0 && !knob_off("DEX_XCD_MAP")) ? 1 : 0;      // 0: the plain 3-D grid
        if (q.xcd_map) grid = dim3(grid.x * 2 * grid.z);
        hipLaunchKernelGGL(attn_direct_ring_kernel, grid, dim3(256), 0, st, q);
        return;
    }
    dim3 grid((p.N + 31) / 32, 2, p.B * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL(attn_direct_kernel, grid, dim3(NWD * 64), lds, st, p);
}

}  // namespace DEX_LP_NS
}  // namespace dex
