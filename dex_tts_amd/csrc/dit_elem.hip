// dit_elem.hip — HBM-bound pieces of the DiT bottleneck (dit.py): overlapping depthwise patch-embed conv + SiLU,
// pos-conv tail (GELU, mean over frequency, add positional terms), LayerNorm + adaLN modulate.
#include "kernels.h"
#include <cstdlib>

namespace dex {

__device__ __forceinline__ float silu_d(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_d(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// PatchEmbed2D.proj[0..1] (dit.py:57-58): depthwise k x k, stride s, pad k/2; the reference right-pads the
// width to a multiple of patch_size with zeros first (dit.py:442-445) — identical to treating wi >= Wi as zero.
// The DiT input is x * mask_mid (diffusion.py:189 Identity on the last stage), applied here on load.
template <int KS>
__global__ __launch_bounds__(256) void dwconv_silu_kernel(const DwConvP p) {
    // weights [KS*KS][C] in LDS (a workgroup walks many outputs: grid-stride), column masks of a thread's KS taps loaded once per
    // output: per tap one 16 B global load instead of 16 B + 16 B (weights) + 4 B (mask) through the L1, which bounds this kernel
    extern __shared__ __attribute__((aligned(16))) float4 wsh_dw[];
    const int C4 = p.C >> 2;
    for (int q = threadIdx.x; q < KS * KS * C4; q += 256) wsh_dw[q] = *reinterpret_cast<const float4*>(p.Wd + (long)q * 4);
    __syncthreads();
    const long total = (long)p.B * p.Hf * p.Wt * C4;
    for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
        const int cq = (int)(gid % C4);
        const long tok = gid / C4;
        const int wt = (int)(tok % p.Wt);
        const int f = (int)((tok / p.Wt) % p.Hf);
        const int b = (int)(tok / ((long)p.Wt * p.Hf));
        const float* X = p.X + (long)b * p.xb;
        const float* mrow = p.mask ? p.mask + (long)b * p.mask_bstride : nullptr;
        float mkc[KS]; int wcl[KS];
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int wi = wt * p.s + kw - p.pad;
            const bool inw = (unsigned)wi < (unsigned)p.Wi;
            wcl[kw] = inw ? wi : 0;                                          // clamped: loads are unconditional
            mkc[kw] = inw ? (mrow ? mrow[wcl[kw] * p.mask_ws] : 1.f) : 0.f;
        }
        float4 acc = *reinterpret_cast<const float4*>(p.bd + cq * 4);
        const float4 a4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 0) * p.C + cq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 c4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 1) * p.C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int hi = f * p.s + kh - p.pad;
            const bool inh = (unsigned)hi < (unsigned)p.Hi;
            const int hc = inh ? hi : 0;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                float4 v = *reinterpret_cast<const float4*>(X + ((long)hc * p.Wi + wcl[kw]) * p.ldx + cq * 4);
                if (p.aff) { v.x = fmaf(v.x, a4.x, c4.x); v.y = fmaf(v.y, a4.y, c4.y); v.z = fmaf(v.z, a4.z, c4.z); v.w = fmaf(v.w, a4.w, c4.w); }
                const float4 w = wsh_dw[(kh * KS + kw) * C4 + cq];
                const float mk = inh ? mkc[kw] : 0.f;
                acc.x = fmaf(v.x * mk, w.x, acc.x); acc.y = fmaf(v.y * mk, w.y, acc.y);
                acc.z = fmaf(v.z * mk, w.z, acc.z); acc.w = fmaf(v.w * mk, w.w, acc.w);
            }
        }
        acc.x = silu_d(acc.x); acc.y = silu_d(acc.y); acc.z = silu_d(acc.z); acc.w = silu_d(acc.w);
        *reinterpret_cast<float4*>(p.Y + tok * p.C + cq * 4) = acc;
    }
}
// (the form without LDS: small grids - the weight staging above is one more dependent round trip per workgroup - and small kernels)
template <int KS>
__global__ __launch_bounds__(256) void dwconv_silu_direct_kernel(const DwConvP p) {
    const int C4 = p.C >> 2;
    const long total = (long)p.B * p.Hf * p.Wt * C4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cq = (int)(gid % C4);
    const long tok = gid / C4;
    const int wt = (int)(tok % p.Wt);
    const int f = (int)((tok / p.Wt) % p.Hf);
    const int b = (int)(tok / ((long)p.Wt * p.Hf));
    const float* X = p.X + (long)b * p.xb;
    const float* mrow = p.mask ? p.mask + (long)b * p.mask_bstride : nullptr;
    float4 acc = *reinterpret_cast<const float4*>(p.bd + cq * 4);
    const float4 a4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 0) * p.C + cq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 c4 = p.aff ? *reinterpret_cast<const float4*>(p.aff + ((long)b * 2 + 1) * p.C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
        const int hi = f * p.s + kh - p.pad;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int wi = wt * p.s + kw - p.pad;
            const bool inb = (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            const int hc = inb ? hi : 0, wc = inb ? wi : 0;              // clamped: loads are unconditional
            float4 v = *reinterpret_cast<const float4*>(X + ((long)hc * p.Wi + wc) * p.ldx + cq * 4);
            if (p.aff) { v.x = fmaf(v.x, a4.x, c4.x); v.y = fmaf(v.y, a4.y, c4.y); v.z = fmaf(v.z, a4.z, c4.z); v.w = fmaf(v.w, a4.w, c4.w); }
            const float4 w = *reinterpret_cast<const float4*>(p.Wd + (kh * KS + kw) * p.C + cq * 4);
            float mk = mrow ? mrow[wc * p.mask_ws] : 1.f;
            mk = inb ? mk : 0.f;
            acc.x = fmaf(v.x * mk, w.x, acc.x); acc.y = fmaf(v.y * mk, w.y, acc.y);
            acc.z = fmaf(v.z * mk, w.z, acc.z); acc.w = fmaf(v.w * mk, w.w, acc.w);
        }
    }
    acc.x = silu_d(acc.x); acc.y = silu_d(acc.y); acc.z = silu_d(acc.z); acc.w = silu_d(acc.w);
    *reinterpret_cast<float4*>(p.Y + tok * p.C + cq * 4) = acc;
}
void launch_dwconv_silu(const DwConvP& p, hipStream_t st) {
    const long total = (long)p.B * p.Hf * p.Wt * (p.C / 4);
    long blocks = (total + 255) / 256;
    if (blocks < 1024 || p.k < 7) {       // measured: B=1 7.3 (direct) vs 8.3 us; 3x3 at DEX B=32 20.2 vs 21.9; 7x7 at GeDEX B=32 80 vs 65
        const dim3 g1((unsigned)blocks);
        if (p.k == 7) hipLaunchKernelGGL(dwconv_silu_direct_kernel<7>, g1, dim3(256), 0, st, p);
        else if (p.k == 3) hipLaunchKernelGGL(dwconv_silu_direct_kernel<3>, g1, dim3(256), 0, st, p);
        else if (p.k == 15) hipLaunchKernelGGL(dwconv_silu_direct_kernel<15>, g1, dim3(256), 0, st, p);
        return;
    }
    const long cap = knob_or("DEX_DWCONV_CAP", 1024);     // workgroups (each stages the weights once)
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks);
    const size_t lds = (size_t)p.k * p.k * p.C * sizeof(float);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {       // (64 KB is the default limit; a 15 x 15 kernel on 128 channels needs 115 KB)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_silu_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_silu_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_silu_kernel<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_bytes = lds;
    }
    if (p.k == 7) hipLaunchKernelGGL(dwconv_silu_kernel<7>, grid, dim3(256), lds, st, p);
    else if (p.k == 3) hipLaunchKernelGGL(dwconv_silu_kernel<3>, grid, dim3(256), lds, st, p);
    else if (p.k == 15) hipLaunchKernelGGL(dwconv_silu_kernel<15>, grid, dim3(256), lds, st, p);
}

// pos = mean_f GELU(conv + bias)  (dit.py:450-451; SamePad already applied by only computing Hf x Wt outputs);
// tok[b, f*Wt + w, :] = emb + pos[w] + freq_pos[f]  (dit.py:452-454)
// One workgroup per (b, w): thread = (channel quad cq, frequency lane fl of 4); all split-K partial loads of a
// thread are independent; the mean over frequency is an LDS reduction over the 4 frequency lanes.
__global__ __launch_bounds__(256) void pos_finish_kernel(const PosFinishP p) {
    __shared__ float4 red[4][64];
    const int tid = threadIdx.x, wt = blockIdx.x, b = blockIdx.y;
    const int D4 = p.D >> 2;
    for (int c0 = 0; c0 < D4; c0 += 64) {
        const int cq = c0 + (tid & 63), fl = tid >> 6;
        const bool act = cq < D4;
        float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            const float4 bias = *reinterpret_cast<const float4*>(p.bias + cq * 4);
            // channel c = cq*4 of the partials: dense [row][D], or padded groups [row][G][cg_pad]
            const int c = cq * 4;
            const int pcol = p.cg ? (c / p.cg) * p.cg_pad + c % p.cg : c;
            const long pld = p.cg ? (long)(p.D / p.cg) * p.cg_pad : p.D;
            for (int f = fl; f < p.Hf; f += 4) {
                const long row = ((long)b * p.Hf + f) * p.Wt + wt;
                float4 a = bias;
#pragma unroll 8
                for (int s = 0; s < p.nsplit; ++s) {
                    const float4 v = *reinterpret_cast<const float4*>(p.part + (long)s * p.split_stride + row * pld + pcol);
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
                part.x += gelu_d(a.x); part.y += gelu_d(a.y); part.z += gelu_d(a.z); part.w += gelu_d(a.w);
            }
        }
        __syncthreads();
        red[fl][tid & 63] = part;
        __syncthreads();
        if (act) {
            const float inv = 1.f / (float)p.Hf;
            float4 pos;
            pos.x = (red[0][tid & 63].x + red[1][tid & 63].x + red[2][tid & 63].x + red[3][tid & 63].x) * inv;
            pos.y = (red[0][tid & 63].y + red[1][tid & 63].y + red[2][tid & 63].y + red[3][tid & 63].y) * inv;
            pos.z = (red[0][tid & 63].z + red[1][tid & 63].z + red[2][tid & 63].z + red[3][tid & 63].z) * inv;
            pos.w = (red[0][tid & 63].w + red[1][tid & 63].w + red[2][tid & 63].w + red[3][tid & 63].w) * inv;
            for (int f = fl; f < p.Hf; f += 4) {
                const long row = ((long)b * p.Hf + f) * p.Wt + wt;
                const float4 e = *reinterpret_cast<const float4*>(p.emb + row * p.D + cq * 4);
                const float4 fp = *reinterpret_cast<const float4*>(p.freq_pos + (long)f * p.D + cq * 4);
                float4 o;
                o.x = e.x + pos.x + fp.x; o.y = e.y + pos.y + fp.y; o.z = e.z + pos.z + fp.z; o.w = e.w + pos.w + fp.w;
                *reinterpret_cast<float4*>(p.tok + row * p.D + cq * 4) = o;
            }
        }
    }
}
void launch_pos_finish(const PosFinishP& p, hipStream_t st) {
    hipLaunchKernelGGL(pos_finish_kernel, dim3(p.Wt, p.B), dim3(256), 0, st, p);
}

__global__ __launch_bounds__(256) void group_pad_kernel(const float* src, float* dst, long rows, int G, int cg, int cg_pad) {
    const long total = rows * G * cg_pad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % cg_pad);
        const long rg = i / cg_pad;
        dst[i] = j < cg ? src[rg * cg + j] : 0.f;
    }
}
void launch_group_pad(const float* src, float* dst, long rows, int G, int cg, int cg_pad, hipStream_t st) {
    long blocks = (rows * G * cg_pad + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(group_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, rows, G, cg, cg_pad);
}

// LayerNorm (eps 1e-6, biased var, no affine) then x*(1+scale)+shift.  One wave per token, D <= 512, D % 64 == 0.
__global__ __launch_bounds__(256) void ln_mod_kernel(const LnModP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long rows = (long)p.B * p.rows_per_batch;
    if (row >= rows) return;
    const int step = p.step;
    const float* x = p.X + row * p.D;
    const int per = p.D >> 6;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (j < per) ? x[lane + 64 * j] : 0.f; s += v[j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)p.D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = (j < per) ? v[j] - mean : 0.f; q = fmaf(d, d, q); }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)p.D + 1e-6f);
    const float* sh = p.shift + (long)step * p.step_stride;
    const float* sc = p.scale + (long)step * p.step_stride;
    float* y = p.Y + row * p.D;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < per) {
            const int c = lane + 64 * j;
            y[c] = (v[j] - mean) * rstd * (1.f + sc[c]) + sh[c];
        }
    }
}
void launch_ln_mod(const LnModP& p, hipStream_t st) {
    const long rows = (long)p.B * p.rows_per_batch;
    hipLaunchKernelGGL(ln_mod_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
}

}  // namespace dex
