// linattn.hip — LinearAttention of the U-Net stages (diffusion.py:74-92): softmax over ALL positions on k
// (no mask, no scaling), ctx[d,e] = sum_n k[d,n] v[e,n], out[e,n] = sum_d ctx[d,e] q[d,n].
// One pass over k,v per 128-position chunk with a local max (flash-style partials), then a small combine that
// also folds ctx, the Rezero gate g and to_out into one per-batch [128 x C] matrix, so the tail is a plain
// GEMM on q.  Both kernels issue all their global loads up front (they are latency-, not bandwidth-bound).
#include "kernels.h"

namespace dex {

constexpr int LA_POS = 128;   // positions per staged sub-tile
constexpr int LA_SUBT = 4;    // sub-tiles per workgroup (LA_CHUNK = LA_POS * LA_SUBT in dex_api.hip)

// grid (nchunks, heads, B), 256 threads.  A workgroup walks LA_SUBT sub-tiles of 128 positions with a running
// per-channel max (online softmax), prefetching the next sub-tile into registers while the current one is reduced.
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const LinAttnCtxP p) {
    __shared__ __attribute__((aligned(16))) float Ks[LA_POS * 32];
    __shared__ __attribute__((aligned(16))) float Vs[LA_POS * 32];
    __shared__ float red[8 * 32];
    __shared__ float mrun[32], mscale[32];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hid = p.heads * 32;
    const int nbeg = chunk * p.chunk;
    const int nend = min(p.n, nbeg + p.chunk);
    const float* base = p.qkv + (long)b * p.bstride;
    const int koff = hid + h * 32, voff = 2 * hid + h * 32;
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;
    const int dl = tid & 31, part = tid >> 5;
    const int d = tid >> 3, e4 = (tid & 7) * 4;
    if (tid < 32) mrun[tid] = -INFINITY;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float ssum = 0.f;
    float4 kv[4], vv[4];
    auto gload = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + r0 + 32 * j;
            const int nn = n < nend ? n : nbeg;
            kv[j] = *reinterpret_cast<const float4*>(base + (long)nn * p.ld + koff + c4);
            vv[j] = *reinterpret_cast<const float4*>(base + (long)nn * p.ld + voff + c4);
            if (n >= nend) { kv[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); vv[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
    };
    gload(nbeg);
    for (int n0 = nbeg; n0 < nend; n0 += LA_POS) {
        __syncthreads();                                   // previous sub-tile fully consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 32 * j;
            *reinterpret_cast<float4*>(Ks + r * 32 + c4) = kv[j];
            *reinterpret_cast<float4*>(Vs + r * 32 + c4) = vv[j];
        }
        __syncthreads();
        if (n0 + LA_POS < nend) gload(n0 + LA_POS);        // prefetch
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Ks[(part * 16 + r) * 32 + dl]);
        red[part * 32 + dl] = mx;
        __syncthreads();
        if (tid < 32) {
            float m = red[tid];
#pragma unroll
            for (int r = 1; r < 8; ++r) m = fmaxf(m, red[r * 32 + tid]);
            const float mo = mrun[tid], mn = fmaxf(mo, m);
            mscale[tid] = __expf(mo - mn);                 // first sub-tile: exp(-inf) = 0
            mrun[tid] = mn;
        }
        __syncthreads();
        {
            const float m = mrun[dl];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = (part * 16 + r) * 32 + dl;
                Ks[idx] = __expf(Ks[idx] - m);             // padding rows hold -inf -> 0
            }
        }
        __syncthreads();
        const float sc = mscale[d];
        acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc; ssum *= sc;
#pragma unroll 16
        for (int r = 0; r < LA_POS; ++r) {
            const float pv = Ks[r * 32 + d];
            const float4 v = *reinterpret_cast<const float4*>(Vs + r * 32 + e4);
            acc.x = fmaf(pv, v.x, acc.x); acc.y = fmaf(pv, v.y, acc.y);
            acc.z = fmaf(pv, v.z, acc.z); acc.w = fmaf(pv, v.w, acc.w);
            ssum += pv;
        }
    }
    const long pidx = ((long)b * p.heads + h) * p.nchunks + chunk;
    *reinterpret_cast<float4*>(p.part_c + pidx * 1024 + d * 32 + e4) = acc;
    if ((tid & 7) == 0) p.part_s[pidx * 32 + d] = ssum;
    __syncthreads();
    if (tid < 32) p.part_m[pidx * 32 + tid] = mrun[tid];
}
void launch_linattn_ctx(const LinAttnCtxP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_ctx_kernel, dim3(p.nchunks, p.heads, p.B), dim3(256), 0, st, p);
}

// grid (heads, B, 4): each workgroup owns 8 of the 32 context rows d of one head (the combine is bandwidth-/
// latency-bound on the partials, so it is spread over 4x more CUs).
//   Weff[b][h*32+d][c] = g * sum_e ctx[d][e] * Wout[c][h*32+e]
__global__ __launch_bounds__(256) void linattn_combine_kernel(const LinAttnCombineP p) {
    __shared__ float ctx[8 * 33];
    __shared__ float red[8 * 32];
    __shared__ float wsl[256 * 33];                       // Wout[:, h*32 : h*32+32] for up to 256 output channels
    const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y, ds = blockIdx.z;
    const long pbase = ((long)b * p.heads + h) * p.nchunks;
    const int hid = p.heads * 32;
    for (int idx = tid; idx < p.C * 32; idx += 256) {     // coalesced 128-B rows
        const int c = idx >> 5, e = idx & 31;
        wsl[c * 33 + e] = p.Wout[(long)c * hid + h * 32 + e];
    }
    // thread = (local row dlr = tid/32, column e = tid%32); chunk loop with independent loads
    const int dlr = tid >> 5, e = tid & 31, d = ds * 8 + dlr;
    float m = -INFINITY;
#pragma unroll 8
    for (int c = 0; c < p.nchunks; ++c) m = fmaxf(m, p.part_m[(pbase + c) * 32 + d]);
    float acc = 0.f, s = 0.f;
#pragma unroll 8
    for (int c = 0; c < p.nchunks; ++c) {
        const float w = __expf(p.part_m[(pbase + c) * 32 + d] - m);
        acc = fmaf(w, p.part_c[(pbase + c) * 1024 + d * 32 + e], acc);
        s = fmaf(w, p.part_s[(pbase + c) * 32 + d], s);
    }
    ctx[dlr * 33 + e] = acc / s;
    __syncthreads();
    const float g = p.g[0];
    float* We = p.Weff + ((long)b * hid + h * 32 + ds * 8) * p.C;
    for (int idx = tid; idx < 8 * p.C; idx += 256) {
        const int dd = idx / p.C, c = idx - dd * p.C;
        float a = 0.f;
#pragma unroll
        for (int ee = 0; ee < 32; ++ee) a = fmaf(ctx[dd * 33 + ee], wsl[c * 33 + ee], a);
        We[(long)dd * p.C + c] = g * a;
    }
}
void launch_linattn_combine(const LinAttnCombineP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_combine_kernel, dim3(p.heads, p.B, 4), dim3(256), 0, st, p);
}

}  // namespace dex
