// style_elem.hip — the small kernels of the DEX style encoders around their implicit-GEMM convolutions
// (reference DEX-TTS/model/ref_encoder.py, model/base.py): channel LayerNorm, InstanceNorm1D, masked means, the VQ codebook
// lookup, the GRU recurrence, layout changes.  Activations are channels-last fp32 [B][T][C]; T is a few hundred frames of
// ONE reference utterance per batch element, so everything here is latency, not bandwidth.
#include "kernels.h"

namespace dex {

// ---- LayerNorm over channels: one wave per row, C <= 256
__global__ __launch_bounds__(256) void ln_cl_kernel(const LnClP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.X + row * p.C;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; v[j] = c < p.C ? x[c] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = (lane + 64 * j < p.C) ? v[j] - mean : 0.f; q = fmaf(d, d, q); }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)p.C + p.eps);
    const float mk = p.mask ? p.mask[row] : 1.f;          // mask is [B][T] and rows are b*T + t: the same linear index
    float* y = p.Y + row * p.C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < p.C) y[c] = ((v[j] - mean) * rstd * p.gamma[c] + p.beta[c]) * mk;
    }
}
void launch_ln_cl(const LnClP& p, hipStream_t st) {
    hipLaunchKernelGGL(ln_cl_kernel, dim3((unsigned)((p.rows + 3) / 4)), dim3(256), 0, st, p);
}

// ---- InstanceNorm1D: workgroup = (64 channels, b); 4 time lanes x 64 channels, two passes over T
__global__ __launch_bounds__(256) void inorm_cl_kernel(const float* X, float* Y, const float* mask, int T, int C, float eps) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x, cl = tid & 63, tl = tid >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    const bool act = c < C;
    const float* x = X + (long)b * T * C + (act ? c : 0);
    float s = 0.f;
    for (int t = tl; t < T; t += 4) s += x[(long)t * C];
    red[tl][cl] = s;
    __syncthreads();
    const float mean = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)T;
    __syncthreads();
    float q = 0.f;
    for (int t = tl; t < T; t += 4) { const float d = x[(long)t * C] - mean; q = fmaf(d, d, q); }
    red[tl][cl] = q;
    __syncthreads();
    const float var = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)(T - 1);      // unbiased (torch.var default)
    const float std = sqrtf(var + eps);
    if (!act) return;
    float* y = Y + (long)b * T * C + c;
    const float* mrow = mask + (long)b * T;
    for (int t = tl; t < T; t += 4) y[(long)t * C] = (x[(long)t * C] - mean) / std * mrow[t];
}
void launch_inorm_cl(const float* X, float* Y, const float* mask, int B, int T, int C, float eps, hipStream_t st) {
    hipLaunchKernelGGL(inorm_cl_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, X, Y, mask, T, C, eps);
}

__global__ __launch_bounds__(256) void masked_mean_cl_kernel(const float* X, const float* mask, float* out, int T, int C) {
    __shared__ float red[4][64], mred[4];
    const int tid = threadIdx.x, cl = tid & 63, tl = tid >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    const float* x = X + (long)b * T * C + (c < C ? c : 0);
    float s = 0.f, m = 0.f;
    for (int t = tl; t < T; t += 4) { s += x[(long)t * C]; m += mask[(long)b * T + t]; }
    red[tl][cl] = s;
    if (cl == 0) mred[tl] = m;
    __syncthreads();
    if (tl == 0 && c < C) out[(long)b * C + c] = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (mred[0] + mred[1] + mred[2] + mred[3]);
}
void launch_masked_mean_cl(const float* X, const float* mask, float* out, int B, int T, int C, hipStream_t st) {
    hipLaunchKernelGGL(masked_mean_cl_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, X, mask, out, T, C);
}

__global__ void add_bcast_cl_kernel(float* X, const float* v, int T, int C, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long)T * C));
        X[i] += v[(long)b * C + c];
    }
}
void launch_add_bcast_cl(float* X, const float* v, int B, int T, int C, hipStream_t st) {
    const long total = (long)B * T * C;
    long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_bcast_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, v, T, C, total);
}

__global__ void lf0_to_cl_kernel(const float* lf0, const float* mask, float* out, long rows, int ldc) {
    const long total = rows * ldc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ldc;
        out[i] = (i % ldc) == 0 ? lf0[r] * mask[r] : 0.f;
    }
}
void launch_lf0_to_cl(const float* lf0, const float* mask, float* out, int B, int T, int ldc, hipStream_t st) {
    const long rows = (long)B * T;
    long blocks = (rows * ldc + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lf0_to_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, st, lf0, mask, out, rows, ldc);
}

__global__ void cl_to_cf_kernel(const float* X, float* out, int T, int C, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long bc = i / T;
        const int c = (int)(bc % C), b = (int)(bc / C);
        out[i] = X[((long)b * T + t) * C + c];
    }
}
void launch_cl_to_cf(const float* X, float* out, int B, int T, int C, hipStream_t st) {
    const long total = (long)B * T * C;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cl_to_cf_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, out, T, C, total);
}

__global__ void len_mask_kernel(const int* lengths, float* mask, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * T) mask[i] = (i % T) < lengths[i / T] ? 1.f : 0.f;
}
void launch_len_mask(const int* lengths, float* mask, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(len_mask_kernel, dim3((B * T + 255) / 256), dim3(256), 0, st, lengths, mask, B, T);
}

// ---- VQ: one wave per row.  distances = (|e|^2 + |x|^2) + (-2) * dot, argmin (lowest index among equal minima)
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* X, float* out, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = X[row * D + c]; s = fmaf(v, v, s); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = s;
}
void launch_row_sumsq(const float* X, float* out, long rows, int D, hipStream_t st) {
    hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, X, out, rows, D);
}
__global__ __launch_bounds__(256) void vq_lookup_kernel(const VqP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    float x2 = 0.f;
    for (int c = lane; c < p.D; c += 64) { const float v = p.X[row * p.D + c]; x2 = fmaf(v, v, x2); }
    for (int o = 32; o > 0; o >>= 1) x2 += __shfl_xor(x2, o);
    float best = INFINITY; int bi = 0x7fffffff;
    for (int m = lane; m < p.M; m += 64) {
        const float d = (p.e2[m] + x2) + (-2.f) * p.dots[row * p.M + m];
        if (d < best) { best = d; bi = m; }              // ascending m per lane: the first minimum of the lane
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const float mk = p.mask[row];
    for (int c = lane; c < p.D; c += 64) p.out[row * p.D + c] = p.emb[(long)bi * p.D + c] * mk;
    if (p.idx && lane == 0) p.idx[row] = bi;
}
void launch_vq_lookup(const VqP& p, hipStream_t st) {
    hipLaunchKernelGGL(vq_lookup_kernel, dim3((unsigned)((p.rows + 3) / 4)), dim3(256), 0, st, p);
}

// ---- GRU recurrence (torch.nn.GRU gate order r | z | n):  r = s(gi_r + W_hr h + b_hr), z = s(gi_z + W_hz h + b_hz),
// n = tanh(gi_n + r * (W_hn h + b_hn)), h' = (1 - z) * n + z * h.  Workgroup = (direction, b); thread j < 3H owns row j of
// W_hh in registers; h lives in LDS.  H = 96.
constexpr int GRU_H = 96;
__global__ __launch_bounds__(320) void gru_layer_kernel(const GruP p) {
    __shared__ float hs[GRU_H], gh[3 * GRU_H];
    const int tid = threadIdx.x, dir = blockIdx.x, b = blockIdx.y;
    const int H = GRU_H;
    float w[GRU_H];
    float bh = 0.f;
    if (tid < 3 * H) {
        const float* wr = p.Whh + ((long)dir * 3 * H + tid) * H;
#pragma unroll
        for (int k = 0; k < GRU_H; ++k) w[k] = wr[k];
        bh = p.bhh[dir * 3 * H + tid];
    }
    if (tid < H) hs[tid] = 0.f;
    __syncthreads();
    for (int s = 0; s < p.T; ++s) {
        const int t = dir ? p.T - 1 - s : s;
        if (tid < 3 * H) {
            float acc = bh;
#pragma unroll
            for (int k = 0; k < GRU_H; ++k) acc = fmaf(w[k], hs[k], acc);
            gh[tid] = acc;
        }
        __syncthreads();
        if (tid < H) {
            const float* g = p.gi + (((long)b * p.T + t) * 2 + dir) * 3 * H;
            const float r = 1.f / (1.f + expf(-(g[tid] + gh[tid])));
            const float z = 1.f / (1.f + expf(-(g[H + tid] + gh[H + tid])));
            const float n = tanhf(g[2 * H + tid] + r * gh[2 * H + tid]);
            const float hn = (1.f - z) * n + z * hs[tid];
            hs[tid] = hn;
            p.out[((long)b * p.T + t) * 2 * H + dir * H + tid] = hn;
        }
        __syncthreads();
    }
}
void launch_gru_layer(const GruP& p, hipStream_t st) {
    hipLaunchKernelGGL(gru_layer_kernel, dim3(2, p.B), dim3(320), 0, st, p);
}

}  // namespace dex
