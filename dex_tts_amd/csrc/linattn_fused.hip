// linattn_fused.hip — LinearAttention (diffusion.py:74-92) without ever materialising q, k or v (bf16-MFMA mode).
//
// Algebra: q enters linearly (out[e,n] = sum_d ctx[d,e] q[d,n], q = Wq x), so the whole Residual(Rezero(
// LinearAttention)) collapses to   y = x + M_b x + g*b_out   with a per-utterance C x C matrix
//     M_b = g * Wout * blockdiag_h(ctx_h^T) * Wq,      ctx_h[d,e] = sum_n softmax_n(k)[d,n] v[e,n].
// Kernel 1 (this file) streams x once: per 32-pixel sub-tile a wave computes k and v (8 tiles of 32x32) with
// v_mfma_f32_32x32x16_bf16, keeps an online per-channel max, and accumulates ctx^T with a SECOND MFMA whose A
// and B operands are the k/v accumulator registers themselves (C-layout: lane = channel, registers = pixels ->
// exactly the A[e][px] / B[px][d] fragment shape; both use the same pixel order, so no shuffle and no LDS).
// Kernel 2 merges the workgroup partials (flash-style) into normalised ctx; kernel 3 folds ctx, Wq, Wout and g
// into M_b (fp32 [ci][co] and bf16 [co][ci]); the tail is one GEMM with K = C and a residual.
// HBM traffic per call: x read twice + y written, instead of writing q,k,v (6x the size of x) and re-reading them.
#include "kernels.h"
#include "bf16_util.h"

namespace dex {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
union LFrag { uint4 u; bf16x8 v; };


// grid (nblk, B); 256 threads; each wave owns `nsub` consecutive 32-pixel sub-tiles.
// part_m/part_s: [B][4][nblk][32], part_c: [B][4][nblk][32 d][32 e]   (same layout linattn_combine reads)
template <int C>
__global__ __launch_bounds__(256) void linattn_kvctx_kernel(const LinKvCtxP p) {
    constexpr int LDW = C + 8, KS = C / 16;
    extern __shared__ __attribute__((aligned(16))) u16 smem_la[];
    u16* Ws = smem_la;                                       // [256][LDW]  rows: k(4x32) then v(4x32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int blk = blockIdx.x, b = blockIdx.y;
    const u16* Wg = reinterpret_cast<const u16*>(p.Wkv);     // bf16 [256][C]
    {   // all weight loads in flight together, then the LDS stores (a load->store loop serialises 8-16 round trips)
        uint4 wr[C / 8];
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            wr[j] = *reinterpret_cast<const uint4*>(Wg + (long)(it / (C / 8)) * C + (it % (C / 8)) * 8);
        }
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            *reinterpret_cast<uint4*>(Ws + (it / (C / 8)) * LDW + (it % (C / 8)) * 8) = wr[j];
        }
    }
    const float* X = p.X + (long)b * p.xb + p.x_coff;
    const int px_base = (blk * 4 + wave) * p.nsub * 32;

    f32x16 ctxT[4];
    float m_run[4], s_run[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        m_run[h] = -INFINITY; s_run[h] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT[h][r] = 0.f;
    }
    __syncthreads();
    for (int sub = 0; sub < p.nsub; ++sub) {
        const int px0 = px_base + sub * 32;
        if (px0 >= p.npix) break;
        // A fragments of x: lane (pixel i, half hh) holds x[px][ks*16 + hh*8 .. +8]
        const int pxr = min(px0 + i, p.npix - 1);
        const float* xr = X + (long)pxr * p.ldx + hh * 8;
        LFrag af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + ks * 16);
            const float4 c = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
            af[ks].u.x = pack2_bf16(a.x, a.y); af[ks].u.y = pack2_bf16(a.z, a.w);
            af[ks].u.z = pack2_bf16(c.x, c.y); af[ks].u.w = pack2_bf16(c.z, c.w);
        }
        f32x16 kv[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) kv[nt][r] = 0.f;
            const u16* bp = Ws + (nt * 32 + i) * LDW + hh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                LFrag bf; bf.u = *reinterpret_cast<const uint4*>(bp + ks * 16);
                kv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks].v, bf.v, kv[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            // column (channel d = lane&31) max over the 32 pixels of the sub-tile
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = px0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (px >= p.npix) kv[h][r] = -INFINITY;
                mx = fmaxf(mx, kv[h][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m_run[h], mx);
            const float alpha = __expf(m_run[h] - mn);
            m_run[h] = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { kv[h][r] = __expf(kv[h][r] - mn); ps += kv[h][r]; }
            s_run[h] = s_run[h] * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctxT[h][r] *= alpha;
            // ctx^T[e][d] += sum_px v[px][e] * p[px][d]   (A = v tile regs, B = p tile regs, same pixel order)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                LFrag va, pb;
                va.u.x = pack2_bf16(kv[4 + h][8 * k2 + 0], kv[4 + h][8 * k2 + 1]); va.u.y = pack2_bf16(kv[4 + h][8 * k2 + 2], kv[4 + h][8 * k2 + 3]);
                va.u.z = pack2_bf16(kv[4 + h][8 * k2 + 4], kv[4 + h][8 * k2 + 5]); va.u.w = pack2_bf16(kv[4 + h][8 * k2 + 6], kv[4 + h][8 * k2 + 7]);
                pb.u.x = pack2_bf16(kv[h][8 * k2 + 0], kv[h][8 * k2 + 1]); pb.u.y = pack2_bf16(kv[h][8 * k2 + 2], kv[h][8 * k2 + 3]);
                pb.u.z = pack2_bf16(kv[h][8 * k2 + 4], kv[h][8 * k2 + 5]); pb.u.w = pack2_bf16(kv[h][8 * k2 + 6], kv[h][8 * k2 + 7]);
                ctxT[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.v, pb.v, ctxT[h], 0, 0, 0);
            }
        }
    }
    // ---- merge the 4 waves of the workgroup through LDS, write one partial per head
    __syncthreads();                                          // weights no longer needed
    float* mS = reinterpret_cast<float*>(smem_la);            // [4 waves][4 heads][32 d][33]  ctx as [d][e]
    float* mm = mS + 4 * 4 * 32 * 33;                         // [4][4][32] m
    float* ms = mm + 4 * 4 * 32;                              // [4][4][32] s
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float st = s_run[h] + __shfl_xor(s_run[h], 32);
        float* dst = mS + ((wave * 4 + h) * 32 + i) * 33;     // row d = lane&31
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2) + 4 * hh] = ctxT[h][r];   // column e
        if (hh == 0) { mm[(wave * 4 + h) * 32 + i] = m_run[h]; ms[(wave * 4 + h) * 32 + i] = st; }
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * 1024; idx += 256) {
        const int h = idx >> 10, d = (idx >> 5) & 31, e = idx & 31;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, mm[(w * 4 + h) * 32 + d]);
        float acc = 0.f, s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = mm[(w * 4 + h) * 32 + d];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            acc = fmaf(f, mS[((w * 4 + h) * 32 + d) * 33 + e], acc);
            s = fmaf(f, ms[(w * 4 + h) * 32 + d], s);
        }
        const long pidx = ((long)b * 4 + h) * p.nblk + blk;
        p.part_c[pidx * 1024 + d * 32 + e] = acc;
        if (e == 0) { p.part_m[pidx * 32 + d] = M; p.part_s[pidx * 32 + d] = s; }
    }
}

void launch_linattn_kvctx(const LinKvCtxP& p, hipStream_t st) {
    const size_t lds_w = (size_t)256 * (p.C + 8) * sizeof(u16);
    const size_t lds_m = (size_t)(4 * 4 * 32 * 33 + 2 * 4 * 4 * 32) * sizeof(float);
    const size_t lds = lds_w > lds_m ? lds_w : lds_m;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    dim3 grid(p.nblk, p.B);
    if (p.C == 64) hipLaunchKernelGGL(linattn_kvctx_kernel<64>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(linattn_kvctx_kernel<128>, grid, dim3(256), lds, st, p);
}

// grid (4 heads, B, 32 rows d): merge the workgroup partials of ONE context row -> normalised ctx[b][h][d][:].
// thread = (column e = tid%32, partial lane pl = tid/32): 8 lanes stride the partial list with independent loads.
__global__ __launch_bounds__(256) void linattn_merge_kernel(const LinMergeP p) {
    __shared__ float redm[8], reda[8][32], reds[8];
    const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y, d = blockIdx.z;
    const long pbase = ((long)b * 4 + h) * p.nblk;
    const int e = tid & 31, pl = tid >> 5;
    float m = -INFINITY;
#pragma unroll 8
    for (int c = pl; c < p.nblk; c += 8) m = fmaxf(m, p.part_m[(pbase + c) * 32 + d]);
    if (e == 0) redm[pl] = m;
    __syncthreads();
    m = redm[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) m = fmaxf(m, redm[k]);
    float acc = 0.f, s = 0.f;
#pragma unroll 8
    for (int c = pl; c < p.nblk; c += 8) {
        const float w = __expf(p.part_m[(pbase + c) * 32 + d] - m);
        acc = fmaf(w, p.part_c[(pbase + c) * 1024 + d * 32 + e], acc);
        s = fmaf(w, p.part_s[(pbase + c) * 32 + d], s);
    }
    reda[pl][e] = acc;
    if (e == 0) reds[pl] = s;
    __syncthreads();
    if (pl == 0) {
        float a = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += reda[k][e]; ss += reds[k]; }
        p.ctx[(((long)b * 4 + h) * 32 + d) * 32 + e] = a / ss;
    }
}
void launch_linattn_merge(const LinMergeP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_merge_kernel, dim3(4, p.B, 32), dim3(256), 0, st, p);
}

// grid (C/16 column slabs, B): M_b = g * Wout * blockdiag(ctx^T) * Wq for 16 input columns ci:
//   T[h*32+e][ci] = sum_d ctx[h][d][e] * Wq[h*32+d][ci];   M[co][ci] = g * sum_{he} Wout[co][he] * T[he][ci]
// written as Mt fp32 [ci][co] and Mbf bf16 [co][ci].  Everything is staged in LDS with bulk coalesced loads first.
__global__ __launch_bounds__(256) void linattn_fold_kernel(const LinFoldP p) {
    extern __shared__ float smem_f[];
    const int C = p.C, tid = threadIdx.x, b = blockIdx.y, ci0 = blockIdx.x * 16;
    float* cs = smem_f;                     // [4][32][33] ctx
    float* wq = cs + 4 * 32 * 33;           // [128][17]   Wq[:, ci0:ci0+16]
    float* T = wq + 128 * 17;               // [128][17]
    float* wo = T + 128 * 17;               // [C][129]    Wout
    {   // bulk loads first (registers), LDS stores after: one global round trip for the whole staging
        float c_[16], q_[8];
        float4 o_[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) c_[j] = p.ctx[(long)b * 4096 + tid + 256 * j];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int idx = tid + 256 * j; q_[j] = p.Wq[(long)(idx >> 4) * C + ci0 + (idx & 15)]; }
        const int nwo = C * 32 / 256;                       // float4 items per thread (8 for C=64, 16 for C=128)
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < nwo) o_[j] = *reinterpret_cast<const float4*>(p.Wout + (long)(tid + 256 * j) * 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int idx = tid + 256 * j; cs[(idx >> 5) * 33 + (idx & 31)] = c_[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int idx = tid + 256 * j; wq[(idx >> 4) * 17 + (idx & 15)] = q_[j]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < nwo) {
            const int idx = (tid + 256 * j) * 4;
            float* d = wo + (idx >> 7) * 129 + (idx & 127);
            d[0] = o_[j].x; d[1] = o_[j].y; d[2] = o_[j].z; d[3] = o_[j].w;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 128 * 16; idx += 256) {
        const int he = idx >> 4, cl = idx & 15, h = he >> 5, e = he & 31;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) a = fmaf(cs[(h * 32 + d) * 33 + e], wq[(h * 32 + d) * 17 + cl], a);
        T[he * 17 + cl] = a;
    }
    __syncthreads();
    const float g = p.g[0];
    for (int idx = tid; idx < C * 16; idx += 256) {
        const int co = idx >> 4, cl = idx & 15;
        float a = 0.f;
#pragma unroll 16
        for (int he = 0; he < 128; ++he) a = fmaf(wo[co * 129 + he], T[he * 17 + cl], a);
        a *= g;
        const int ci = ci0 + cl;
        p.Mt[(long)b * C * C + (long)ci * C + co] = a;
        unsigned u = __float_as_uint(a);
        u += 0x7FFFu + ((u >> 16) & 1u);
        reinterpret_cast<u16*>(p.Mbf)[(long)b * C * C + (long)co * C + ci] = (u16)(u >> 16);
    }
}
void launch_linattn_fold(const LinFoldP& p, hipStream_t st) {
    const size_t lds = (size_t)(4 * 32 * 33 + 2 * 128 * 17 + p.C * 129) * sizeof(float);
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024); attr = true; }
    hipLaunchKernelGGL(linattn_fold_kernel, dim3(p.C / 16, p.B), dim3(256), lds, st, p);
}

}  // namespace dex
