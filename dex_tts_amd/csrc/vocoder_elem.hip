// vocoder_elem.hip — the HBM-bound pieces of the HiFi-GAN generator (reference hifigan/models.py:112-173) around its
// implicit-GEMM convolutions: mel layout change, ConvTranspose1d overlap-add, ResBlock average, conv_post + tanh.
// Activations are channels-last fp32 [B][L][C].
#include "kernels.h"

namespace dex {

__global__ __launch_bounds__(256) void mel_to_cl_kernel(const float* mel, float* out, int B, int C, int T, int ldc) {
    const long total = (long)B * T * ldc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % ldc);
        const long bt = i / ldc;
        const int t = (int)(bt % T), b = (int)(bt / T);
        out[i] = c < C ? mel[((long)b * C + c) * T + t] : 0.f;
    }
}
void launch_mel_to_cl(const float* mel, float* out, int B, int C, int T, int ldc, hipStream_t st) {
    long blocks = ((long)B * T * ldc + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mel_to_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mel, out, B, C, T, ldc);
}

// one thread = 4 consecutive output channels of one output position
__global__ __launch_bounds__(256) void convt_fold_kernel(const ConvTFoldP p) {
    const int C4 = p.Cout >> 2;
    const long Lo = (long)p.L * p.u;
    const long total = (long)p.B * Lo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const long bt = i / C4;
        const long t = bt % Lo;
        const int b = (int)(bt / Lo);
        float4 acc = *reinterpret_cast<const float4*>(p.bias + c);
        const float* Yb = p.Y + (long)b * p.L * p.k * p.Cout;
        // taps in ascending j, the order torch's col2im accumulates them in is not specified; two terms: commutative
        for (int j = (int)((t + p.pad) % p.u); j < p.k; j += p.u) {
            const long l = (t + p.pad - j) / p.u;
            if (l >= 0 && l < p.L) {
                const float4 y = *reinterpret_cast<const float4*>(Yb + (l * p.k + j) * p.Cout + c);
                acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
            }
        }
        *reinterpret_cast<float4*>(p.out + ((long)b * Lo + t) * p.Cout + c) = acc;
    }
}
void launch_convt_fold(const ConvTFoldP& p, hipStream_t st) {
    long blocks = ((long)p.B * p.L * p.u * (p.Cout / 4) + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(convt_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

__global__ __launch_bounds__(256) void avg3_kernel(const float4* a, const float4* b, const float4* c, float4* out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], z = c[i];
        // xs = r0(x); xs += r1(x); xs += r2(x); x = xs / 3   (models.py:158-164): the same op order
        out[i] = make_float4(((x.x + y.x) + z.x) / 3.f, ((x.y + y.y) + z.y) / 3.f, ((x.z + y.z) + z.z) / 3.f, ((x.w + y.w) + z.w) / 3.f);
    }
}
void launch_avg3(const float* a, const float* b, const float* c, float* out, long n, hipStream_t st) {
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avg3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                       reinterpret_cast<const float4*>(c), reinterpret_cast<float4*>(out), n4);
}

// C <= 64 channels, 7 taps; one thread per output sample (x rows are 128 B: L1 serves the 7-tap overlap)
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const ConvPostP p) {
    __shared__ float ws[7 * 64];
    for (int k = threadIdx.x; k < 7 * p.C; k += 256) ws[k] = p.W[k];
    __syncthreads();
    const long total = (long)p.B * p.L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long t = i % p.L;
        const int b = (int)(i / p.L);
        float acc = p.bias[0];
        for (int tap = 0; tap < 7; ++tap) {
            const long tt = t + tap - 3;
            if (tt < 0 || tt >= p.L) continue;
            const float* x = p.X + ((long)b * p.L + tt) * p.C;
            for (int c = 0; c < p.C; c += 4) {
                float4 v = *reinterpret_cast<const float4*>(x + c);
                v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;     // HiFi-GAN: F.leaky_relu default slope 0.01; BigVGAN: 1
                v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
                const float* w = ws + tap * p.C + c;
                acc = fmaf(v.x, w[0], acc); acc = fmaf(v.y, w[1], acc); acc = fmaf(v.z, w[2], acc); acc = fmaf(v.w, w[3], acc);
            }
        }
        p.wav[i] = tanhf(acc);
    }
}
void launch_conv_post_tanh(const ConvPostP& p, hipStream_t st) {
    long blocks = ((long)p.B * p.L + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(conv_post_tanh_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

// ---- BigVGAN Activation1d.  UpSample1d(2, 12): replicate-pad 5, transposed conv stride 2 with the filter, x2, crop 15 / 15:
//   up[m] = 2 * sum_j x[clamp(j - 5)] * f[m + 15 - 2 j]   (the six j with 0 <= m + 15 - 2 j <= 11),  m in [0, 2L)
// snake: s = up + inv_b * sin^2(a * up);  DownSample1d(2, 12): replicate-pad (5, 6), conv stride 2:
//   y[t] = sum_k s[clamp(2 t + k - 5, 0, 2L - 1)] * f[k].
// A block owns AS_T output positions x 64 channels: the 2 * AS_T + 11 activated samples it needs go through LDS once
// (two sin per output instead of twelve).
constexpr int AS_T = 32, AS_S = 2 * AS_T + 11;
__global__ __launch_bounds__(256) void aa_snake_kernel(const AaSnakeP p) {
    __shared__ float s[AS_S][64];
    __shared__ float f[12];
    const int tid = threadIdx.x, c = tid & 63, tl = tid >> 6;
    const int c0 = blockIdx.x * 64, t0 = blockIdx.y * AS_T, b = blockIdx.z;
    if (tid < 12) f[tid] = p.filt[tid];
    __syncthreads();
    const bool cok = c0 + c < p.C;
    const float a = cok ? p.a[c0 + c] : 0.f, ib = cok ? p.inv_b[c0 + c] : 0.f;
    const float* X = p.X + (long)b * p.L * p.C + (cok ? c0 + c : 0);
    const int L2 = 2 * p.L;
    for (int q = tl; q < AS_S; q += 4) {
        const int m = min(max(2 * t0 - 5 + q, 0), L2 - 1);            // the down-sampler's replicate padding
        float up = 0.f;
        const int j0 = (m + 5) >> 1;                                  // ceil((m + 4) / 2)
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const int j = j0 + jj, k = m + 15 - 2 * j;                // k in [0, 11] for these six j (m + 15 - 2 j0 is 10 or 11)
            const int xi = min(max(j - 5, 0), p.L - 1);
            if (k >= 0) up = fmaf(X[(long)xi * p.C], f[k], up);
        }
        up *= 2.f;
        const float sn = sinf(up * a);
        s[q][c] = up + ib * (sn * sn);
    }
    __syncthreads();
    if (!cok) return;
    float* Y = p.Y + (long)b * p.L * p.C + c0 + c;
    for (int tt = tl; tt < AS_T; tt += 4) {
        const int t = t0 + tt;
        if (t >= p.L) break;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc = fmaf(s[2 * tt + k][c], f[k], acc);
        Y[(long)t * p.C] = acc;
    }
}
void launch_aa_snake(const AaSnakeP& p, hipStream_t st) {
    hipLaunchKernelGGL(aa_snake_kernel, dim3((p.C + 63) / 64, (p.L + AS_T - 1) / AS_T, p.B), dim3(256), 0, st, p);
}
// a = alpha (exp if log-scale); inv_b = 1 / (beta + 1e-9) (activations.py:52-56 / :110-116; beta == alpha for plain Snake)
__global__ void snake_coeffs_kernel(const float* alpha, const float* beta, float* a, float* inv_b, int C, int logscale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float al = alpha[c], be = beta[c];
    if (logscale) { al = expf(al); be = expf(be); }
    a[c] = al;
    inv_b[c] = 1.0f / (be + 0.000000001f);
}
void launch_snake_coeffs(const float* alpha, const float* beta, float* a, float* inv_b, int C, int logscale, hipStream_t st) {
    hipLaunchKernelGGL(snake_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, st, alpha, beta, a, inv_b, C, logscale);
}

}  // namespace dex
