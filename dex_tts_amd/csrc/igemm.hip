// igemm.hip — implicit-GEMM convolution / linear on the CDNA4 matrix cores.
//
// One kernel serves every dense contraction of the score network (reference call sites:
// Block conv3x3 diffusion.py:44, res_conv :62, to_qkv/to_out :79-80, Downsample :25, Upsample :16 (as four
// 2x2-tap parity sub-convolutions), patch-embed pointwise dit.py:59, grouped 16x16 pos-conv dit.py:82-88
// (split-K), every nn.Linear of the DiT blocks dit.py:281-284 + timm Attention/Mlp, FinalLayer :322).
//
// fp32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate — bitwise an fmaf chain), A tile
// staged K-major in LDS ([BK][BM+1], conflict-free b32 reads and writes), B tile [BK][BN].
// 256 threads = 4 waves; each wave owns MT 32x32 accumulator tiles.  Register-staged prefetch of the next
// K tile overlaps the global gather with the MFMA chain.
#include "kernels.h"
#include "igemm_epilogue.h"

namespace dex {


template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IGemmP p) {
    constexpr int BK = 32;
    constexpr int WN = BN / 32, WM = 4 / WN, MT = BM / (WM * 32);
    constexpr int AP = BM / 32;                 // A-gather passes (32 rows x 8 float4 per pass)
    constexpr int LDAS = BM + 1;
    constexpr int BTR = BN / 4;                 // threads per B row
    constexpr int BROWS = 256 / BTR;            // B rows per pass
    constexpr int BP = BK / BROWS;
    static_assert(MT >= 1 && BP >= 1, "tile");
    __shared__ float As[BK * LDAS];
    __shared__ float Bs[BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int z = blockIdx.z, par = 0;
    if (p.parity) { par = z & 3; z >>= 2; }
    const int s = z % p.ksplit;
    const int g = (z / p.ksplit) % p.groups;
    const int b = z / (p.ksplit * p.groups);
    const int off_h = p.parity ? (par >> 1) : p.off_h, off_w = p.parity ? (par & 1) : p.off_w;
    const int oh0 = p.parity ? (par >> 1) : p.oh0, ow0 = p.parity ? (par & 1) : p.ow0;
    const int M = p.Ho * p.Wo;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch), each XCD has a private L2, and
    // neighbouring pixel tiles share their 3x3 halo -> give every XCD one contiguous range of tiles.
    int mtile = blockIdx.x;
    if ((gridDim.x & 7) == 0) mtile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = mtile * BM, n0 = blockIdx.y * BN;
    const int Kper = p.K / p.ksplit, kbeg = s * Kper, nkt = Kper / BK;

    const float* Ab = p.A + (long)b * p.a_bstride + p.a_coff + g * p.Cin;
    const float* mrow = p.inmask ? p.inmask + (long)b * p.mask_bstride : nullptr;
    const int arow = tid >> 3, ak4 = (tid & 7) * 4;
    int bh[AP], bw[AP];
    bool mv[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        int m = m0 + arow + 32 * j;
        mv[j] = m < M;
        int ho = m / p.Wo, wo = m - ho * p.Wo;
        bh[j] = ho * p.sh + off_h;
        bw[j] = wo * p.sw + off_w;
    }
    const int brow = tid / BTR, bc4 = (tid % BTR) * 4;
    const float* Wb = p.W + (long)b * p.w_bstride + (long)g * p.w_gstride + (long)par * p.K * p.N + n0 + bc4;

    float4 ra[AP];
    float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0;
    static_assert(BP <= 2, "B passes");
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // kt = -1 is the prologue (gather tile 0 only); afterwards: stage tile kt to LDS, prefetch tile kt+1
    // into registers, run the MFMA chain on tile kt.
    for (int kt = -1; kt < nkt; ++kt) {
        if (kt >= 0) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                float* d = As + ak4 * LDAS + arow + 32 * j;
                d[0] = ra[j].x; d[LDAS] = ra[j].y; d[2 * LDAS] = ra[j].z; d[3 * LDAS] = ra[j].w;
            }
            *reinterpret_cast<float4*>(Bs + brow * BN + bc4) = rb0;
            if constexpr (BP > 1) *reinterpret_cast<float4*>(Bs + (brow + BROWS) * BN + bc4) = rb1;
            __syncthreads();
        }
        if (kt + 1 < nkt) {
            const int k0 = kbeg + (kt + 1) * BK;
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const int hi = bh[j] + kh * p.step_h, wi = bw[j] + kw * p.step_w;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mv[j] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) {
                    v = *reinterpret_cast<const float4*>(Ab + ((long)hi * p.Wi + wi) * p.lda + c0 + ak4);
                    if (mrow) {
                        const float mk = mrow[wi * p.inmask_ws];
                        v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
                    }
                    if (p.act_in_slope != 0.f) {              // leaky_relu on load (vocoder)
                        const float sl = p.act_in_slope;
                        v.x = v.x > 0.f ? v.x : v.x * sl; v.y = v.y > 0.f ? v.y : v.y * sl;
                        v.z = v.z > 0.f ? v.z : v.z * sl; v.w = v.w > 0.f ? v.w : v.w * sl;
                    }
                }
                ra[j] = v;
            }
            rb0 = *reinterpret_cast<const float4*>(Wb + (long)(k0 + brow) * p.N);
            if constexpr (BP > 1) rb1 = *reinterpret_cast<const float4*>(Wb + (long)(k0 + brow + BROWS) * p.N);
        }
        if (kt >= 0) {
            const float* ap = As + hh * LDAS + wm * (MT * 32) + i;
            const float* bp = Bs + hh * BN + wn * 32 + i;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float bf = bp[kk * 2 * BN];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float af = ap[kk * 2 * LDAS + t * 32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
    }

    igemm_epilogue<MT>(p, acc, m0, n0, wm * (MT * 32), wn * 32, lane, b, g, s, M, oh0, ow0);
}

void launch_igemm_lp(const IGemmP& p, int precision, hipStream_t st);   // lp_dispatch.hip -> igemm_bf16.hip (bf16 / fp16 build)

void launch_igemm(const IGemmP& p, int precision, hipStream_t st) {
    if (prec_is_lp(precision) && p.Wbf != nullptr) { launch_igemm_lp(p, precision, st); return; }
    const int M = p.Ho * p.Wo;
    const int zdim = p.B * p.groups * p.ksplit * (p.parity ? 4 : 1);
    if (p.N % 64 == 0) {
        // small-M problems (DiT tokens at B=1) use the 64-row tile to put more workgroups on the chip
        const long blocks128 = (long)((M + 127) / 128) * (p.N / 64) * zdim;
        if (blocks128 < 512) {
            dim3 grid((M + 63) / 64, p.N / 64, zdim);
            hipLaunchKernelGGL((igemm_f32_kernel<64, 64>), grid, dim3(256), 0, st, p);
        } else {
            dim3 grid((M + 127) / 128, p.N / 64, zdim);
            hipLaunchKernelGGL((igemm_f32_kernel<128, 64>), grid, dim3(256), 0, st, p);
        }
    } else {  // N % 32 == 0 (grouped pos-conv: 32 channels per group)
        dim3 grid((M + 127) / 128, p.N / 32, zdim);
        hipLaunchKernelGGL((igemm_f32_kernel<128, 32>), grid, dim3(256), 0, st, p);
    }
}

}  // namespace dex
