// text_elem.hip — element-wise / per-row kernels of the text encoder path (dex_text.hip): embedding gather, the three
// row norms (channel LayerNorm of text_encoder.py:13-31, RMSNorm of retention.py:48-66, AdaptiveLayerNorm of DEX
// base.py:180-194), xPos rotation of q / k into the attention kernel's padded head layout (retention.py:26-35,271-278),
// the retention output gate (group RMSNorm per head x swish(g), :285-287), the GLU product (:379-383), durations
// (tts.py:37-39) and the monotonic path / mu_y gather (utils.py:26-39, tts.py:44-50).  Activations are channels-last
// [B*T][ld] fp32.  Everything here runs once per utterance on a few hundred rows: written for clarity, not for speed.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

namespace dex {

__global__ void embed_kernel(const int* tok, const float* emb, float* out, long rows, int C, int ld, float scale, int n_vocab) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const long r = idx / C; const int c = (int)(idx % C);
    int t = tok[r]; t = t < 0 ? 0 : (t >= n_vocab ? n_vocab - 1 : t);
    out[r * ld + c] = emb[(long)t * C + c] * scale;
}
void launch_embed(const int* tok, const float* emb, float* out, long rows, int C, int ld, float scale, int n_vocab, hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, st, tok, emb, out, rows, C, ld, scale, n_vocab);
}

// X[b][t][coff + j] = v[b][j]   (spk.unsqueeze(-1).repeat(1, 1, T), text_encoder.py:138)
__global__ void bcast_cols_kernel(float* X, int ld, int coff, const float* v, int T, int n, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % n); const long row = idx / n; const long b = row / T;
    X[row * ld + coff + j] = v[b * n + j];
}
void launch_bcast_cols(float* X, int ld, int coff, const float* v, int B, int T, int n, hipStream_t st) {
    const long total = (long)B * T * n;
    hipLaunchKernelGGL(bcast_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, ld, coff, v, T, n, total);
}

// one wave per row, C <= 256
__global__ __launch_bounds__(256) void row_norm_kernel(const RowNormP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.X + row * p.ldx;
    float v[4];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; v[j] = c < p.C ? x[c] : 0.f; s += v[j]; q = fmaf(v[j], v[j], q); }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    float mean = 0.f, rstd;
    if (p.mode == 1) rstd = 1.f / sqrtf(q / (float)p.C + p.eps);               // RMSNorm: x * rsqrt(mean(x^2) + eps)
    else {
        mean = s / (float)p.C;
        float d2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = (lane + 64 * j < p.C) ? v[j] - mean : 0.f; d2 = fmaf(d, d, d2); }
        for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o);
        rstd = 1.f / sqrtf(d2 / (float)p.C + p.eps);
    }
    const float mk = p.mask ? p.mask[row] : 1.f;
    const long b = row / p.T;
    float* y = p.Y + row * p.ldy;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < p.C) {
            float o;
            if (p.mode == 1) o = v[j] * rstd * (p.gamma ? p.gamma[c] : 1.f);
            else if (p.mode == 2) o = (v[j] - mean) * rstd * p.gamma[b * p.C + c] + p.beta[b * p.C + c];    // per-utterance scale / bias
            else o = (v[j] - mean) * rstd * p.gamma[c] + p.beta[c];
            if (p.relu) o = fmaxf(o, 0.f);
            y[c] = o * mk;
        }
    }
}
void launch_row_norm(const RowNormP& p, hipStream_t st) {
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((p.rows + 3) / 4)), dim3(256), 0, st, p);
}

// q, k (x key_dim^-0.5), v of one row -> xPos-rotated, head-padded operands of the fp32 attention kernel
__global__ void ret_rotate_kernel(const RetRotP p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HP = 128;
    const long total = p.rows * p.heads * (HP / 2);
    if (idx >= total) return;
    const int j = (int)(idx % (HP / 2));
    const int h = (int)((idx / (HP / 2)) % p.heads);
    const long row = idx / ((long)(HP / 2) * p.heads);
    const int t = (int)(row % p.T);
    const long o = row * p.ldp + h * HP + 2 * j;
    if (2 * j >= p.kd) {
        p.Q[o] = 0.f; p.Q[o + 1] = 0.f; p.K[o] = 0.f; p.K[o + 1] = 0.f; p.V[o] = 0.f; p.V[o + 1] = 0.f;
        return;
    }
    const float* x = p.qkvg + row * p.ld;
    const int c = h * p.kd + 2 * j;
    const float a0 = (float)t * p.angle[2 * j], a1 = (float)t * p.angle[2 * j + 1];
    const float s0 = sinf(a0), c0 = cosf(a0), s1 = sinf(a1), c1 = cosf(a1);
    const float q0 = x[c], q1 = x[c + 1];
    const float k0 = x[p.E + c] * p.kscale, k1 = x[p.E + c + 1] * p.kscale;
    p.Q[o] = q0 * c0 + (-q1) * s0; p.Q[o + 1] = q1 * c1 + q0 * s1;              // x * cos + rotate_every_two(x) * sin
    p.K[o] = k0 * c0 + (-k1) * s0; p.K[o + 1] = k1 * c1 + k0 * s1;
    p.V[o] = x[2 * p.E + c]; p.V[o + 1] = x[2 * p.E + c + 1];
}
void launch_ret_rotate(const RetRotP& p, hipStream_t st) {
    const long total = p.rows * p.heads * 64;
    hipLaunchKernelGGL(ret_rotate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
}

// out[row][h*hd + d] = swish(g) * O / rms_h(O): one wave per (row, head), hd <= 128
__global__ __launch_bounds__(256) void ret_gate_kernel(const RetGateP p) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.rows * p.heads) return;
    const long row = item / p.heads; const int h = (int)(item % p.heads);
    const float* o = p.O + row * p.ldo + h * 128;
    float v[2], q = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int d = lane + 64 * j; v[j] = d < p.hd ? o[d] : 0.f; q = fmaf(v[j], v[j], q); }
    for (int s = 32; s > 0; s >>= 1) q += __shfl_xor(q, s);
    const float r = 1.f / sqrtf(q / (float)p.hd + p.eps);
    const float* g = p.qkvg + row * p.ld + 3 * p.E + h * p.hd;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = lane + 64 * j;
        if (d < p.hd) { const float gg = g[d]; p.out[row * p.ldout + h * p.hd + d] = (gg / (1.f + expf(-gg))) * (v[j] * r); }
    }
}
void launch_ret_gate(const RetGateP& p, hipStream_t st) {
    hipLaunchKernelGGL(ret_gate_kernel, dim3((unsigned)((p.rows * p.heads + 3) / 4)), dim3(256), 0, st, p);
}

// out[r][f] = gelu(GF[r][F + f]) * GF[r][f]     (GF = a [gate | fc1]^T)
__global__ void glu_kernel(const float* GF, float* out, long rows, int F) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * F) return;
    const long r = idx / F; const int f = (int)(idx % F);
    const float g = GF[r * 2 * F + f], x = GF[r * 2 * F + F + f];
    out[idx] = 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)) * g;
}
void launch_glu(const float* GF, float* out, long rows, int F, hipStream_t st) {
    hipLaunchKernelGGL(glu_kernel, dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, GF, out, rows, F);
}

// channels-last [B][T][ldx] (C used) * mask -> channel-first [B][C][T]
__global__ void cl_to_cf_mask_kernel(const float* X, int ldx, const float* mask, float* out, int T, int C, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int t = (int)(idx % T); const int c = (int)((idx / T) % C); const long b = idx / ((long)T * C);
    out[idx] = X[(b * T + t) * ldx + c] * mask[b * T + t];
}
void launch_cl_to_cf_mask(const float* X, int ldx, const float* mask, float* out, int B, int T, int C, hipStream_t st) {
    const long total = (long)B * T * C;
    hipLaunchKernelGGL(cl_to_cf_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, ldx, mask, out, T, C, total);
}

// w = exp(logw) * mask; w_ceil = ceil(w) * length_scale; cum = running sum, accumulated in fp64 and rounded per element (what
// torch.cumsum does for fp32 on the CPU; exact for the integer durations of length_scale = 1); y_len = max(sum, 1) truncated
__global__ void durations_kernel(const float* logw, const float* mask, float length_scale, float* w_ceil, float* cum, int* y_len, int T) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double run = 0.0;
    for (int t = 0; t < T; ++t) {
        const float w = expf(logw[(long)b * T + t]) * mask[(long)b * T + t];
        const float wc = ceilf(w) * length_scale;
        w_ceil[(long)b * T + t] = wc;
        run += (double)wc;
        cum[(long)b * T + t] = (float)run;
    }
    const float tot = fmaxf((float)run, 1.f);
    y_len[b] = (int)tot;
}
void launch_durations(const float* logw, const float* mask, float length_scale, float* w_ceil, float* cum, int* y_len, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(64), 0, st, logw, mask, length_scale, w_ceil, cum, y_len, T);
}

// running sum along each row: fp64 accumulator, fp32 outputs (torch.cumsum on the CPU)
__global__ void cumsum_rows_kernel(const float* X, float* out, int T) {
    if (threadIdx.x != 0) return;
    const long b = blockIdx.x;
    double run = 0.0;
    for (int t = 0; t < T; ++t) { run += (double)X[b * T + t]; out[b * T + t] = (float)run; }
}
void launch_cumsum_rows(const float* X, float* out, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(cumsum_rows_kernel, dim3(B), dim3(64), 0, st, X, out, T);
}

// frame t of utterance b belongs to the first token i with t < cum[i] (generate_path: the row where the step function flips);
// mu_y[b][:, t] = mu_x[b][:, i] under x_mask[i] * y_mask[t], else 0; attn (optional) [B][T][Ty] gets the 0/1 column.
__global__ void align_kernel(const AlignP p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= p.Ty) return;
    const float* cum = p.cum + (long)b * p.T;
    const float tf = (float)t;
    int lo = 0, hi = p.T;                                   // smallest i with tf < cum[i]; T if none
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tf < cum[mid]) hi = mid; else lo = mid + 1; }
    const bool ym = t < p.y_len[b];
    const bool on = lo < p.T && lo < p.x_len[b] && ym;
    if (p.y_mask) p.y_mask[(long)b * p.Ty + t] = ym ? 1.f : 0.f;
    for (int c = 0; c < p.F; ++c) p.mu_y[((long)b * p.F + c) * p.Ty + t] = on ? p.mu_x[((long)b * p.F + c) * p.T + lo] : 0.f;
    if (p.attn) for (int i = 0; i < p.T; ++i) p.attn[((long)b * p.T + i) * p.Ty + t] = (on && i == lo) ? 1.f : 0.f;
}
void launch_align(const AlignP& p, hipStream_t st) {
    hipLaunchKernelGGL(align_kernel, dim3((p.Ty + 63) / 64, p.B), dim3(64), 0, st, p);
}

}  // namespace dex
