// linattn.hip — LinearAttention of the U-Net stages (diffusion.py:74-92): softmax over ALL positions on k
// (no mask, no scaling), ctx[d,e] = sum_n k[d,n] v[e,n], out[e,n] = sum_d ctx[d,e] q[d,n].
// One pass over k,v per 128-position chunk with a local max (flash-style partials), then a small combine that
// also folds ctx, the Rezero gate g and to_out into one per-batch [128 x C] matrix, so the tail is a plain
// GEMM on q.  Both kernels issue all their global loads up front (they are latency-, not bandwidth-bound).
#include "kernels.h"

namespace dex {

constexpr int LA_POS = 128;   // positions per workgroup (== LA_CHUNK in dex_api.hip)

// grid (nchunks, heads, B), 256 threads
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const LinAttnCtxP p) {
    __shared__ __attribute__((aligned(16))) float Ks[LA_POS * 32];
    __shared__ __attribute__((aligned(16))) float Vs[LA_POS * 32];
    __shared__ float red[8 * 32];
    __shared__ float mloc[32];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hid = p.heads * 32;
    const int n0 = chunk * LA_POS;
    const int cnt = min(LA_POS, p.n - n0);
    const float* base = p.qkv + (long)b * p.bstride + (long)n0 * p.ld;
    const int koff = hid + h * 32, voff = 2 * hid + h * 32;
    // stage k and v tiles: 128 positions x 32 channels each; thread -> (position r = tid/8 + 32*j, float4 c4)
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;
    float4 kv[4], vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + 32 * j;
        const int rr = r < cnt ? r : 0;
        kv[j] = *reinterpret_cast<const float4*>(base + (long)rr * p.ld + koff + c4);
        vv[j] = *reinterpret_cast<const float4*>(base + (long)rr * p.ld + voff + c4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + 32 * j;
        if (r >= cnt) { kv[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); vv[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
        *reinterpret_cast<float4*>(Ks + r * 32 + c4) = kv[j];
        *reinterpret_cast<float4*>(Vs + r * 32 + c4) = vv[j];
    }
    __syncthreads();
    // per-channel max over the chunk
    const int dl = tid & 31, part = tid >> 5;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Ks[(part * 16 + r) * 32 + dl]);
    red[part * 32 + dl] = mx;
    __syncthreads();
    if (tid < 32) {
        float m = red[tid];
#pragma unroll
        for (int r = 1; r < 8; ++r) m = fmaxf(m, red[r * 32 + tid]);
        mloc[tid] = m;
    }
    __syncthreads();
    {   // p = exp(k - m) in place (padding rows hold -inf -> 0)
        const float m = mloc[dl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = (part * 16 + r) * 32 + dl;
            Ks[idx] = __expf(Ks[idx] - m);
        }
    }
    __syncthreads();
    const int d = tid >> 3, e4 = (tid & 7) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float ssum = 0.f;
#pragma unroll 16
    for (int r = 0; r < LA_POS; ++r) {
        const float pv = Ks[r * 32 + d];
        const float4 v = *reinterpret_cast<const float4*>(Vs + r * 32 + e4);
        acc.x = fmaf(pv, v.x, acc.x); acc.y = fmaf(pv, v.y, acc.y);
        acc.z = fmaf(pv, v.z, acc.z); acc.w = fmaf(pv, v.w, acc.w);
        ssum += pv;
    }
    const long pidx = ((long)b * p.heads + h) * p.nchunks + chunk;
    *reinterpret_cast<float4*>(p.part_c + pidx * 1024 + d * 32 + e4) = acc;
    if ((tid & 7) == 0) p.part_s[pidx * 32 + d] = ssum;
    if (tid < 32) p.part_m[pidx * 32 + tid] = mloc[tid];
}
void launch_linattn_ctx(const LinAttnCtxP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_ctx_kernel, dim3(p.nchunks, p.heads, p.B), dim3(256), 0, st, p);
}

// grid (heads, B).  Weff[b][h*32+d][c] = g * sum_e ctx[d][e] * Wout[c][h*32+e]
__global__ __launch_bounds__(256) void linattn_combine_kernel(const LinAttnCombineP p) {
    __shared__ float ctx[32 * 33];
    __shared__ float red[8 * 32];
    __shared__ float gm[32], gs[32];
    __shared__ float wsl[256 * 33];                       // Wout[:, h*32 : h*32+32] for up to 256 output channels
    const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const long pbase = ((long)b * p.heads + h) * p.nchunks;
    const int hid = p.heads * 32;
    for (int idx = tid; idx < p.C * 32; idx += 256) {     // coalesced 128-B rows
        const int c = idx >> 5, e = idx & 31;
        wsl[c * 33 + e] = p.Wout[(long)c * hid + h * 32 + e];
    }
    const int dl = tid & 31, part = tid >> 5;
    float m = -INFINITY;
    for (int c = part; c < p.nchunks; c += 8) m = fmaxf(m, p.part_m[(pbase + c) * 32 + dl]);
    red[part * 32 + dl] = m;
    __syncthreads();
    if (tid < 32) {
        float mm = red[tid];
#pragma unroll
        for (int r = 1; r < 8; ++r) mm = fmaxf(mm, red[r * 32 + tid]);
        gm[tid] = mm;
    }
    __syncthreads();
    float s = 0.f;
    const float gmd = gm[dl];
    for (int c = part; c < p.nchunks; c += 8)
        s = fmaf(__expf(p.part_m[(pbase + c) * 32 + dl] - gmd), p.part_s[(pbase + c) * 32 + dl], s);
    __syncthreads();
    red[part * 32 + dl] = s;
    __syncthreads();
    if (tid < 32) {
        float ss = red[tid];
#pragma unroll
        for (int r = 1; r < 8; ++r) ss += red[r * 32 + tid];
        gs[tid] = ss;
    }
    const int d = tid >> 3, e4 = (tid & 7) * 4;
    const float gmx = gm[d];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = 0; c < p.nchunks; ++c) {
        const float w = __expf(p.part_m[(pbase + c) * 32 + d] - gmx);
        const float4 v = *reinterpret_cast<const float4*>(p.part_c + (pbase + c) * 1024 + d * 32 + e4);
        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    __syncthreads();
    const float inv = 1.f / gs[d];
    ctx[d * 33 + e4 + 0] = acc.x * inv; ctx[d * 33 + e4 + 1] = acc.y * inv;
    ctx[d * 33 + e4 + 2] = acc.z * inv; ctx[d * 33 + e4 + 3] = acc.w * inv;
    __syncthreads();
    const float g = p.g[0];
    float* We = p.Weff + ((long)b * hid + h * 32) * p.C;
    for (int idx = tid; idx < 32 * p.C; idx += 256) {
        const int dd = idx / p.C, c = idx - dd * p.C;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) a = fmaf(ctx[dd * 33 + e], wsl[c * 33 + e], a);
        We[(long)dd * p.C + c] = g * a;
    }
}
void launch_linattn_combine(const LinAttnCombineP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_combine_kernel, dim3(p.heads, p.B), dim3(256), 0, st, p);
}

}  // namespace dex
