// dex_elem.hip — DEX-TTS style adaptors (DEX-TTS/model/ref_encoder.py:142-179,239-273, model/base.py:67-114)
// around the shared attention / GEMM kernels: instance-norm statistics, SAP pooling tables, IN2d folded
// into w_q, AdaIN apply, time-token rows, layout transposes.
#include "kernels.h"
#include <algorithm>
#include "bf16_util.h"

namespace dex {

// InstanceNorm1D.cal_stats (base.py:72-78): mean and sqrt(unbiased var + eps) over the FULL padded length.
// One wave per (b, c) row of a [B, C, len] tensor; out[b*out_bstride + c].
__global__ __launch_bounds__(256) void row_stats_kernel(const float* X, int B, int C, int len, float eps,
                                                        float* mean, float* sd, long out_bstride) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * C) return;
    const float* x = X + row * len;
    double s = 0.0;
    for (int k = lane; k < len; k += 64) s += (double)x[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mu = s / len;
    double q = 0.0;
    for (int k = lane; k < len; k += 64) { const double d = (double)x[k] - mu; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) {
        const int b = (int)(row / C), c = (int)(row % C);
        mean[(long)b * out_bstride + c] = (float)mu;
        sd[(long)b * out_bstride + c] = (float)sqrt(q / (double)(len - 1) + (double)eps);
    }
}
void launch_row_stats(const float* X, int B, int C, int len, float eps, float* mean, float* sd, long out_bstride, hipStream_t st) {
    const long rows = (long)B * C;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, X, B, C, len, eps, mean, sd, out_bstride);
}

// per-(b,c) sum / sumsq over all pixels (incl. padding).  grid (chunks, B); C <= 256.
constexpr int IN_PIX = 256;     // few workgroups (few atomics: 40k global atomics cost ~5 us), 16 loads in flight per thread
__global__ __launch_bounds__(256) void in_stats_kernel(const InStatsP p, const int pix_per_wg) {
    __shared__ long long red[256][2];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int C4 = p.C >> 2;
    red[tid][0] = 0; red[tid][1] = 0;
    __syncthreads();
    const int cq = tid % C4, prow = tid / C4, rpp = 256 / C4;
    const int pbeg = blockIdx.x * pix_per_wg, pend = min(p.npix, pbeg + pix_per_wg);
    const float* X = p.X + (long)b * p.bstride;
    // rounds of 16 unconditional (clamped) loads in flight together (a plain loop serialised 32 round trips: 14-17 us).  A round's 16
    // values are summed in fp32 in a fixed order and converted to fixed point ONCE PER ROUND; the rounds of a thread add up as integers.
    // Rounds sit at absolute pixel positions (pixels per workgroup is a multiple of a round's span), so the bits of the statistics do not
    // depend on how many pixels a workgroup takes - i.e. not on the batch size (the per-thread fp32 run was 32 values at
    // B = 1 and up to 512 at batch, and an utterance's fp32-mode result was not batch-invariant).
    constexpr int RN = 16;
    const double inv_n = 1.0 / (double)p.npix;
    gnfix_t as[4] = {0, 0, 0, 0}, aq[4] = {0, 0, 0, 0};
    for (int px0 = pbeg + prow; px0 < pend; px0 += RN * rpp) {
        float4 v[RN]; float mk[RN];
#pragma unroll
        for (int k = 0; k < RN; ++k) {
            const int px = min(px0 + k * rpp, pend - 1);
            v[k] = *reinterpret_cast<const float4*>(X + (long)px * p.ld + cq * 4);
            mk[k] = p.mask ? p.mask[(long)b * p.mask_bstride + (px % p.W) * p.mask_ws] : 1.f;
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < RN; ++k) {
            const float m = (px0 + k * rpp < pend) ? mk[k] : 0.f;
            const float4 t = make_float4(v[k].x * m, v[k].y * m, v[k].z * m, v[k].w * m);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            q.x = fmaf(t.x, t.x, q.x); q.y = fmaf(t.y, t.y, q.y); q.z = fmaf(t.z, t.z, q.z); q.w = fmaf(t.w, t.w, q.w);
        }
        as[0] += gn_fix(s.x, inv_n); as[1] += gn_fix(s.y, inv_n); as[2] += gn_fix(s.z, inv_n); as[3] += gn_fix(s.w, inv_n);
        aq[0] += gn_fix(q.x, inv_n); aq[1] += gn_fix(q.y, inv_n); aq[2] += gn_fix(q.z, inv_n); aq[3] += gn_fix(q.w, inv_n);
    }
    const int c = cq * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { gn_add(&red[c + e][0], as[e]); gn_add(&red[c + e][1], aq[e]); }
    __syncthreads();
    if (tid < p.C) {
        gnfix_t* dst = p.stats + (((long)b * p.C + tid) * IN_SLOTS + (blockIdx.x % IN_SLOTS)) * 2;
        gn_add(dst, red[tid][0]);
        gn_add(dst + 1, red[tid][1]);
    }
}
void launch_in_stats(const InStatsP& p, hipStream_t st) {
    // pixels per workgroup: IN_PIX at small grids; at batch size as many as leave ~1 024 workgroups - every workgroup ends with 2 C global
    // atomics, and 1 280 workgroups x 256 of them were most of the launch at B = 32, T = 512 (52.9 us for 168 MB; 19.0 us for 84 MB at T = 256)
    int pix = IN_PIX;
    const long want = ((long)p.npix * p.B / 1024 + IN_PIX - 1) / IN_PIX * IN_PIX;
    if (want > pix) pix = (int)std::min<long>(want, 4096);
    hipLaunchKernelGGL(in_stats_kernel, dim3((p.npix + pix - 1) / pix, p.B), dim3(256), 0, st, p, pix);
}

__device__ __forceinline__ void in_mean_rstd(const gnfix_t* stats, long idx, int npix, float eps, float& mean, float& rstd) {
    long long t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < IN_SLOTS; ++k) { t1 += stats[(idx * IN_SLOTS + k) * 2]; t2 += stats[(idx * IN_SLOTS + k) * 2 + 1]; }
    const double n = (double)npix;
    const double mu = (double)t1 * (1.0 / GN_FIX_ONE);
    double var = ((double)t2 * (1.0 / GN_FIX_ONE) - mu * mu) * (n / (n - 1.0));    // unbiased (torch.var default, base.py:99)
    var = var < 0.0 ? 0.0 : var;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// SelfAttentionPooling (ref_encoder.py:246-253) for every Euler step: one wave per (step, b).  C <= 256.
__global__ __launch_bounds__(64) void sap_kernel(const SapP p) {
    const int lane = threadIdx.x, stepi = blockIdx.x, b = blockIdx.y;
    const int per = (p.C + 63) / 64;
    float logit[8], xs[8][4];
    const float* tt = p.t_tok + (long)stepi * p.t_ld + p.t_coff;
    for (int l = 0; l <= p.L; ++l) {
        float a = 0.f;
        for (int j = 0; j < per; ++j) {
            const int c = lane + 64 * j;
            float v = 0.f;
            if (c < p.C) v = (l == 0) ? tt[c] : p.stats[((long)b * p.L + (l - 1)) * p.C + c];
            xs[l][j] = v;
            if (c < p.C) a = fmaf(v, p.w[c], a);
        }
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        logit[l] = a + p.bias[0];
    }
    float mx = logit[0];
    for (int l = 1; l <= p.L; ++l) mx = fmaxf(mx, logit[l]);
    float den = 0.f;
    for (int l = 0; l <= p.L; ++l) { logit[l] = __expf(logit[l] - mx); den += logit[l]; }
    for (int j = 0; j < per; ++j) {
        const int c = lane + 64 * j;
        if (c >= p.C) continue;
        float o = 0.f;
        for (int l = 0; l <= p.L; ++l) o = fmaf(xs[l][j], logit[l] / den, o);
        p.out[((long)stepi * p.B + b) * p.C + c] = o;
    }
}
void launch_sap(const SapP& p, hipStream_t st) {
    hipLaunchKernelGGL(sap_kernel, dim3(p.nsteps, p.B), dim3(64), 0, st, p);
}

// TVAdaptor: q = w_q(IN2d(x)) folded into a per-batch weight/bias (ref_encoder.py:166):
//   Weff[b][k][n] = rstd[b,k] * Wq[n][k];   beff[b][n] = -sum_k mean[b,k] * rstd[b,k] * Wq[n][k]
// grid (C/16 + 1, B): workgroup j < C/16 writes rows k = 16j .. 16j+15 of Weff (Wq read through an LDS tile so that both
// sides are coalesced), the last one the bias.  (One workgroup per batch element took 16 us at B=1.)
__global__ __launch_bounds__(256) void in_fold_kernel(const InFoldP p) {
    __shared__ float smean[256], srstd[256];
    __shared__ float tile[16][257];
    const int tid = threadIdx.x, b = blockIdx.y, C = p.C;
    const int nslice = C / 16;
    if (blockIdx.x < nslice) {
        const int k0 = blockIdx.x * 16;
        if (tid < 16) { float m; in_mean_rstd(p.stats, (long)b * C + k0 + tid, p.npix, p.eps, m, srstd[tid]); }
        if (p.Wbf) {            // [n][k] bf16: no transpose - 64 B in, 32 B out per thread
            __syncthreads();
            unsigned short* Wo = reinterpret_cast<unsigned short*>(p.Wbf) + (long)b * C * C;
            for (int n = tid; n < C; n += 256) {
                uint4 o[2], ol[2];
                unsigned* ow = reinterpret_cast<unsigned*>(o);
                unsigned* olw = reinterpret_cast<unsigned*>(ol);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 w = *reinterpret_cast<const float4*>(p.Wq + (long)n * C + k0 + q4 * 4);
                    const float v0 = w.x * srstd[q4 * 4], v1 = w.y * srstd[q4 * 4 + 1], v2 = w.z * srstd[q4 * 4 + 2], v3 = w.w * srstd[q4 * 4 + 3];
                    ow[q4 * 2] = pack2_kind(v0, v1, p.lp);
                    ow[q4 * 2 + 1] = pack2_kind(v2, v3, p.lp);
                    if (p.split) {          // (fp16) what the rounding lost, rounded again: the lo half of the split weight
                        const unsigned a = ow[q4 * 2], c = ow[q4 * 2 + 1];
                        const float h0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(a & 0xffffu)), h1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(a >> 16));
                        const float h2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(c & 0xffffu)), h3 = (float)__builtin_bit_cast(_Float16, (unsigned short)(c >> 16));
                        olw[q4 * 2] = pack2_kind(v0 - h0, v1 - h1, p.lp);
                        olw[q4 * 2 + 1] = pack2_kind(v2 - h2, v3 - h3, p.lp);
                    }
                }
                *reinterpret_cast<uint4*>(Wo + (long)n * C + k0) = o[0];
                *reinterpret_cast<uint4*>(Wo + (long)n * C + k0 + 8) = o[1];
                if (p.split) {
                    unsigned short* Wl = reinterpret_cast<unsigned short*>(p.Wbf) + (long)p.B * C * C + (long)b * C * C;
                    *reinterpret_cast<uint4*>(Wl + (long)n * C + k0) = ol[0];
                    *reinterpret_cast<uint4*>(Wl + (long)n * C + k0 + 8) = ol[1];
                }
            }
            return;
        }
        // tile[kk][n] = Wq[n][k0 + kk]: thread reads 16 consecutive k of row n (64 B)
        for (int n = tid; n < C; n += 256) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 w = *reinterpret_cast<const float4*>(p.Wq + (long)n * C + k0 + q4 * 4);
                tile[q4 * 4 + 0][n] = w.x; tile[q4 * 4 + 1][n] = w.y; tile[q4 * 4 + 2][n] = w.z; tile[q4 * 4 + 3][n] = w.w;
            }
        }
        __syncthreads();
        float* We = p.Weff + (long)b * C * C + (long)k0 * C;
        for (int idx = tid; idx < 16 * C; idx += 256) {
            const int kk = idx / C, n = idx - kk * C;
            We[idx] = srstd[kk] * tile[kk][n];
        }
    } else {
        if (tid < C) in_mean_rstd(p.stats, (long)b * C + tid, p.npix, p.eps, smean[tid], srstd[tid]);
        __syncthreads();
        if (tid < C) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const float* wr = p.Wq + (long)tid * C;
            for (int k = 0; k < C; k += 4) {
                const float4 w = *reinterpret_cast<const float4*>(wr + k);
                a0 = fmaf(smean[k] * srstd[k], w.x, a0); a1 = fmaf(smean[k + 1] * srstd[k + 1], w.y, a1);
                a2 = fmaf(smean[k + 2] * srstd[k + 2], w.z, a2); a3 = fmaf(smean[k + 3] * srstd[k + 3], w.w, a3);
            }
            p.beff[(long)b * C + tid] = -((a0 + a1) + (a2 + a3));
        }
    }
}
void launch_in_fold(const InFoldP& p, hipStream_t st) {
    hipLaunchKernelGGL(in_fold_kernel, dim3(p.C / 16 + 1, p.B), dim3(256), 0, st, p);
}

// TIVAdaptor: y = IN2d(x) * s + m, output NOT masked (ref_encoder.py:271)
__global__ __launch_bounds__(256) void tiv_apply_kernel(const TivApplyP p) {
    __shared__ float sa[256], sb[256];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int step = p.step;
    if (tid < p.C) {
        float mean, rstd;
        in_mean_rstd(p.stats, (long)b * p.C + tid, p.npix, p.eps, mean, rstd);
        const float s = p.s_tab[((long)step * p.B + b) * p.C + tid];
        const float m = p.m_tab[((long)step * p.B + b) * p.C + tid];
        sa[tid] = rstd * s;
        sb[tid] = m - mean * rstd * s;
    }
    __syncthreads();
    const int C4 = p.C >> 2;
    const long total = (long)p.npix * C4;
    const float* X = p.X + (long)b * p.xb;
    float* Y = p.Y + (long)b * p.yb;
    if ((256 % C4) == 0) {
        // a thread's channel quad is fixed (256 threads cover whole pixels): coefficients in registers, no per-element division,
        // four loads in flight (the generic loop below ran at 2.4 TB/s: a 64-bit division and one dependent load per element)
        const int c = (tid % C4) * 4, ppb = 256 / C4;                       // pixels per block pass
        const float4 a4 = *reinterpret_cast<const float4*>(sa + c), b4 = *reinterpret_cast<const float4*>(sb + c);
        const long pstride = (long)gridDim.x * ppb;
        for (long px = (long)blockIdx.x * ppb + tid / C4; px < p.npix; px += 4 * pstride) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const float4*>(X + min(px + k * pstride, (long)p.npix - 1) * p.ld + c);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (px + k * pstride < p.npix)
                    *reinterpret_cast<float4*>(Y + (px + k * pstride) * p.ldy + c) =
                        make_float4(fmaf(x[k].x, a4.x, b4.x), fmaf(x[k].y, a4.y, b4.y), fmaf(x[k].z, a4.z, b4.z), fmaf(x[k].w, a4.w, b4.w));
        }
        return;
    }
    for (long idx = (long)blockIdx.x * 256 + tid; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        const long px = idx / C4;
        const float4 x = *reinterpret_cast<const float4*>(X + px * p.ld + c);
        float4 y;
        y.x = fmaf(x.x, sa[c], sb[c]); y.y = fmaf(x.y, sa[c + 1], sb[c + 1]);
        y.z = fmaf(x.z, sa[c + 2], sb[c + 2]); y.w = fmaf(x.w, sa[c + 3], sb[c + 3]);
        *reinterpret_cast<float4*>(Y + px * p.ldy + c) = y;
    }
}
__global__ void tiv_coef_kernel(const TivApplyP p, float* aff) {
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid < p.C) {          // (the arithmetic of tiv_apply_kernel's prologue: the consumer's fmaf(x, a, c) gives tiv_apply's bits)
        float mean, rstd;
        in_mean_rstd(p.stats, (long)b * p.C + tid, p.npix, p.eps, mean, rstd);
        const float s = p.s_tab[((long)p.step * p.B + b) * p.C + tid];
        const float m = p.m_tab[((long)p.step * p.B + b) * p.C + tid];
        aff[((long)b * 2 + 0) * p.C + tid] = rstd * s;
        aff[((long)b * 2 + 1) * p.C + tid] = m - mean * rstd * s;
    }
}
void launch_tiv_coef(const TivApplyP& p, float* aff, hipStream_t st) {
    hipLaunchKernelGGL(tiv_coef_kernel, dim3(p.B), dim3(256), 0, st, p, aff);
}
void launch_tiv_apply(const TivApplyP& p, hipStream_t st) {
    const long total = (long)p.npix * (p.C / 4);
    long blocks = (total + 1023) / 1024; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    const long cap = knob_or("DEX_TIV_CAP", 512);    // workgroups in total: each recomputes the per-channel coefficients from the slot partials (measured at DEX B=32: 5120 workgroups 60.6 us, 2048 45.1, 1024 37.9, 512 34.3)
    if (cap > 0 && blocks * p.B > cap) { blocks = cap / p.B; if (blocks < 1) blocks = 1; }
    hipLaunchKernelGGL(tiv_apply_kernel, dim3((unsigned)blocks, p.B), dim3(256), 0, st, p);
}

__global__ void tv_row0_kernel(const TvRow0P p) {
    const int b = blockIdx.x, c = threadIdx.x;
    const int step = p.step;
    if (p.zero_ptr) for (long i = (long)b * blockDim.x + c; i < p.zero_n; i += (long)gridDim.x * blockDim.x) p.zero_ptr[i] = 0.f;
    if (b < p.B && c < p.C) {
        const float kv = p.k0[(long)step * p.C + c], vv = p.v0[(long)step * p.C + c];
        p.K[(long)b * p.kvb + c] = kv;
        p.V[(long)b * p.kvb + c] = vv;
        if (p.Kp) {        // the time token's row of the 16-bit operands (the style rows are converted once per call: launch_tv_kv_prep)
            // (fragment order, TvKvPrepP: key 0 = tile 0, st 0, lane i = 0)
            const int pos = (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1);          // channel c sits at position pos of its key row (accumulator order inside every 16-group)
            reinterpret_cast<unsigned short*>(p.Kp)[(long)b * p.NkPad * p.C + ((pos >> 4) * 64 + ((pos >> 3) & 1) * 32) * 8 + (pos & 7)] =
                (unsigned short)(pack2_kind(kv, 0.f, p.lp_kind) & 0xffffu);
            reinterpret_cast<unsigned short*>(p.VTp)[(long)b * p.C * p.NkPad + (((c >> 5) * 4) * 64 + (c & 31)) * 8] = (unsigned short)(pack2_kind(vv, 0.f, p.lp_kind) & 0xffffu);
        }
    }
}
void launch_tv_row0(const TvRow0P& p, hipStream_t st) {
    // (the statistics it clears for their next use are 4 MB at B = 32: B workgroups took 7 us over them)
    const long zb = p.zero_ptr ? (p.zero_n + 4095) / 4096 : 0;
    hipLaunchKernelGGL(tv_row0_kernel, dim3((unsigned)std::max<long>(p.B, std::min<long>(zb, 1024))), dim3(256), 0, st, p);
}

// Folded TV adaptor, per step (kernels.h TvFold2P): grid (NkPad / 64, B), 256 threads.  Every workgroup reduces the utterance's 128 x IN_SLOTS
// statistics itself (64 KB from L2, all loads in flight), then scales its 64 keys of G into the 16-bit K' operand: 4 x 16 B per thread, in the
// MFMA fragment order tv_chain_fold_kernel's LDS-DMA ring takes as it is (attention_bf16.hip).
__global__ __launch_bounds__(256) void tv_fold2_kernel(const TvFold2P p) {
    __shared__ float srstd[128];
    const int tid = threadIdx.x, b = blockIdx.y, C = p.C;
    if (p.zero_ptr) {
        const long nthr = (long)gridDim.x * gridDim.y * 256;
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < p.zero_n; i += nthr) p.zero_ptr[i] = 0.f;
    }
    if (tid < C) {
        float mean, rstd;
        in_mean_rstd(p.stats, (long)b * C + tid, p.npix, p.eps, mean, rstd);
        srstd[tid] = rstd * p.scale;
        if (blockIdx.x == 0) {
            p.xmean[(long)b * C + tid] = mean;
            // key 0 of V'^T in fragment order: tile 0, piece (t = ch / 32, q = 0), lane ch % 32 (hh = 0), element 0
            reinterpret_cast<unsigned short*>(p.VTp)[(long)b * C * p.NkPad + (((tid >> 5) * 4) * 64 + (tid & 31)) * 8] =
                (unsigned short)(pack2_kind(p.v0p[(long)p.step * C + tid], 0.f, p.lp_kind) & 0xffffu);
        }
    }
    __syncthreads();
    unsigned short* Kp = reinterpret_cast<unsigned short*>(p.Kp) + (long)b * p.NkPad * C;
    const int c8 = (tid & 15) * 8;
    const float4 r0 = *reinterpret_cast<const float4*>(srstd + c8), r1 = *reinterpret_cast<const float4*>(srstd + c8 + 4);
    float4 ga[4], gc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = blockIdx.x * 64 + (tid >> 4) + 16 * j;
        const float* src = key == 0 ? p.g0 + (long)p.step * C + c8 : p.G + (long)b * p.gb + (long)min(key, p.Nk - 1) * C + c8;
        ga[j] = *reinterpret_cast<const float4*>(src); gc[j] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = blockIdx.x * 64 + (tid >> 4) + 16 * j;
        uint4 o = make_uint4(pack2_kind(ga[j].x * r0.x, ga[j].y * r0.y, p.lp_kind), pack2_kind(ga[j].z * r0.z, ga[j].w * r0.w, p.lp_kind),
                             pack2_kind(gc[j].x * r1.x, gc[j].y * r1.y, p.lp_kind), pack2_kind(gc[j].z * r1.z, gc[j].w * r1.w, p.lp_kind));
        if (key >= p.Nk) o = make_uint4(0u, 0u, 0u, 0u);
        // fragment order: tile key / 64, piece (st = (key / 32) % 2, ks = chunk / 2), lane (i = key % 32, hh = chunk % 2)
        const int c = tid & 15;
        *reinterpret_cast<uint4*>(Kp + ((long)((key >> 6) * 16 + ((key >> 5) & 1) * 8 + (c >> 1)) * 64 + (c & 1) * 32 + (key & 31)) * 8) = o;
    }
}
void launch_tv_fold2(const TvFold2P& p, hipStream_t st) {
    hipLaunchKernelGGL(tv_fold2_kernel, dim3(p.NkPad / 64, p.B), dim3(256), 0, st, p);
}

__global__ void transpose_cl_kernel(const float* src, float* dst, int B, int C, int L, int row_off, long dst_bstride) {
    const long total = (long)B * C * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int l = (int)((i / C) % L);
        const int b = (int)(i / ((long)C * L));
        dst[(long)b * dst_bstride + (long)(l + row_off) * C + c] = src[((long)b * C + c) * L + l];
    }
}
void launch_transpose_cl(const float* src, float* dst, int B, int C, int L, int row_off, long dst_bstride, hipStream_t st) {
    const long total = (long)B * C * L;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(transpose_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, B, C, L, row_off, dst_bstride);
}

}  // namespace dex
