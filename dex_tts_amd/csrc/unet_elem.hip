// unet_elem.hip — HBM-bound pieces of the Grad-TTS style U-Net stages (channels-last fp32):
// first-layer direct convolution, GroupNorm statistics, GN-apply+Mish+mask(+time bias)(+residual),
// the fused final_block tail + final_conv + EDM combine + Euler update, conditioning tables.
#include "kernels.h"
#include "bf16_util.h"
#include <cstdlib>

namespace dex {

// Mish = x * tanh(softplus(x)) (diffusion.py:8-10; torch softplus threshold 20).
// tanh(log1p(e)) = n/(n+2) with n = e*(e+2), e = exp(x): one exp, one divide, no cancellation.
__device__ __forceinline__ float mish_f(float x) {
    if (x > 20.f) return x;
    const float e = __expf(x);
    const float n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));        // v_rcp_f32: 1 ulp, an order below the parity tolerances
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }


// mean / rstd of each GroupNorm group from the slot-spread fixed-point partials: thread (g = tid/GN_SLOTS, slot = tid%GN_SLOTS)
// loads one (mean, mean-of-squares) pair, GN_SLOTS-lane shuffle reduce — one global round trip instead of 32 serial ones.
__device__ __forceinline__ void gn_mean_rstd(const gnfix_t* stats, int b, int groups, float* smean, float* srstd, int tid) {
    if (tid < groups * GN_SLOTS) {
        const int g = tid / GN_SLOTS;
        const longlong2 v = *reinterpret_cast<const longlong2*>(stats + (((long)b * groups + g) * GN_SLOTS + (tid % GN_SLOTS)) * 2);
        long long s1 = v.x, s2 = v.y;
        gn_slots_reduce<GN_SLOTS>(s1, s2);
        if ((tid % GN_SLOTS) == 0) gn_moments(s1, s2, 1e-5, smean[g], srstd[g]);
    }
}

// ------------------------------------------------------------------------------------------------
// first conv (C = 64, or 128 for DEX-LibriTTS): one thread = one pixel x 16 output channels; the 3x3 and 1x1 weights sit
// in LDS (5-16 KB, broadcast reads), so the 18-27 input taps of a pixel are loaded by C/16 threads instead of C/4 and the
// GroupNorm partials are reduced by shuffles before they touch LDS (the per-thread LDS atomics of the first version were a
// 32-way serialisation).  planes <= 3.
template <int PLANES, int C>
__global__ __launch_bounds__(256) void first_conv_kernel(const FirstConvP p) {
    constexpr int TPP = C / 16;                 // threads per pixel
    constexpr int PPB = 256 / TPP;              // pixels per block
    constexpr int CPG = C / 8;                  // channels per GroupNorm group (8 or 16)
    __shared__ __attribute__((aligned(16))) float w3s[PLANES * 9 * C];
    __shared__ __attribute__((aligned(16))) float w1s[PLANES * C];
    __shared__ long long red[16];
    const int tid = threadIdx.x;
    for (int k = tid; k < PLANES * 9 * C; k += 256) w3s[k] = p.W3[k];
    for (int k = tid; k < PLANES * C; k += 256) w1s[k] = p.W1[k];
    if (tid < 16) red[tid] = 0;
    // grid (blocks per utterance, B): a block walks the PPB-pixel groups bx, bx + gridDim.x, ... of ONE utterance, so the weight
    // fill above and the statistics reduction below are paid once per block, not once per 64 pixels (at B=32: 20480 short blocks,
    // 158 us; the GroupNorm partials stay in registers across the walk)
    const int b = blockIdx.y;
    const long npu = (long)p.H * p.T;                          // pixels per utterance
    const int cq = tid % TPP;                                 // channels [cq*16, cq*16+16)
    const int step = p.step;
    const float c_in = p.scal[step * p.scal_stride + 2];
    const float* mrow = p.mask + (long)b * p.T;
    const float* pl[2] = {p.mu + (long)b * p.H * p.T, p.x + (long)b * p.H * p.T};
    constexpr int GPT = 16 / CPG;             // GroupNorm groups per thread: 2 (C = 64) or 1 (C = 128)
    float gs[GPT], gq[GPT];
#pragma unroll
    for (int g = 0; g < GPT; ++g) { gs[g] = 0.f; gq[g] = 0.f; }
    __syncthreads();
    const long ngrp = (npu + PPB - 1) / PPB;
    for (long grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
    asm volatile("" ::: "memory");      // the weights stay in LDS: without this the 300+ loop-invariant LDS reads are hoisted into registers (256 VGPRs, one wave per SIMD)
    const long pl_raw = grp * PPB + tid / TPP;
    const bool live = pl_raw < npu;
    const long pixl = live ? pl_raw : npu - 1;
    const long pix = (long)b * npu + pixl;
    const int w = (int)(pixl % p.T);
    const int h = (int)(pixl / p.T);
    float v[PLANES][9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = h + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = w + kw - 1;
            const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.T;
            const int hc = inb ? hi : h, wc = inb ? wi : w;          // clamped, always-valid address
            const float mk = inb ? mrow[wc] : 0.f;
#pragma unroll
            for (int q = 0; q < PLANES; ++q) {
                float t;
                if (q < 2) t = pl[q][(long)hc * p.T + wc] * ((q == 1) ? c_in : 1.f);
                else t = p.spk[(long)b * p.H + hc];
                v[q][kh * 3 + kw] = t * mk;
            }
        }
    }
    float4 a3[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a3[j] = *reinterpret_cast<const float4*>(p.b3 + cq * 16 + j * 4);
        a1[j] = *reinterpret_cast<const float4*>(p.b1 + cq * 16 + j * 4);
    }
#pragma unroll
    for (int q = 0; q < PLANES; ++q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float x = v[q][t];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(w3s + (q * 9 + t) * C + cq * 16 + j * 4);
                a3[j].x = fmaf(x, wv.x, a3[j].x); a3[j].y = fmaf(x, wv.y, a3[j].y); a3[j].z = fmaf(x, wv.z, a3[j].z); a3[j].w = fmaf(x, wv.w, a3[j].w);
            }
        }
        const float xc = v[q][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 w1 = *reinterpret_cast<const float4*>(w1s + q * C + cq * 16 + j * 4);
            a1[j].x = fmaf(xc, w1.x, a1[j].x); a1[j].y = fmaf(xc, w1.y, a1[j].y); a1[j].z = fmaf(xc, w1.z, a1[j].z); a1[j].w = fmaf(xc, w1.w, a1[j].w);
        }
    }
    if (live) {
        if (p.h1_bf16) {
            uint4 lo, hi;
            const int kd = p.h1_bf16;               // 1 = bf16, 2 = fp16
            lo.x = pack2_kind(a3[0].x, a3[0].y, kd); lo.y = pack2_kind(a3[0].z, a3[0].w, kd); lo.z = pack2_kind(a3[1].x, a3[1].y, kd); lo.w = pack2_kind(a3[1].z, a3[1].w, kd);
            hi.x = pack2_kind(a3[2].x, a3[2].y, kd); hi.y = pack2_kind(a3[2].z, a3[2].w, kd); hi.z = pack2_kind(a3[3].x, a3[3].y, kd); hi.w = pack2_kind(a3[3].z, a3[3].w, kd);
            unsigned short* h = reinterpret_cast<unsigned short*>(p.h1) + pix * C + cq * 16;
            *reinterpret_cast<uint4*>(h) = lo;
            *reinterpret_cast<uint4*>(h + 8) = hi;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!p.h1_bf16) *reinterpret_cast<float4*>(p.h1 + pix * C + cq * 16 + j * 4) = a3[j];
            if (p.res) *reinterpret_cast<float4*>(p.res + pix * C + cq * 16 + j * 4) = a1[j];     // null: the consumer recomputes it (Conv3P::res2_*)
        }
    }
    if (p.gn_stats) {          // GroupNorm partials of h1: CPG channels per group -> this thread feeds 16 / CPG groups
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (j * 4) / CPG;
            const float4 u = a3[j];
            if (live) {
                gs[g] += (u.x + u.y) + (u.z + u.w);
                gq[g] += (u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w);
            }
        }
    }
    }   // group walk
    if (p.gn_stats) {
        // lanes of a wave with the same cq (stride TPP) hold the same groups: xor-reduce over the pixel index bits
#pragma unroll
        for (int o = TPP; o < 64; o <<= 1) {
#pragma unroll
            for (int g = 0; g < GPT; ++g) { gs[g] += __shfl_xor(gs[g], o); gq[g] += __shfl_xor(gq[g], o); }
        }
        if ((tid & 63) < TPP) {          // one wave's sums (fixed shuffle order) -> fixed point; integer adds commute
            const double inv_n = 1.0 / ((double)p.H * p.T * CPG);
#pragma unroll
            for (int g = 0; g < GPT; ++g) { gn_add(&red[(GPT * cq + g) * 2], gn_fix(gs[g], inv_n)); gn_add(&red[(GPT * cq + g) * 2 + 1], gn_fix(gq[g], inv_n)); }
        }
        __syncthreads();
        if (tid < 16)
            gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2 + (tid & 1), red[tid]);
    }
}
// ---- matrix-core form.  The VALU kernel above spends 320 FMAs and 80 LDS weight reads per pixel quarter (150 us at B=32 for
// 168 MB of output).  The same arithmetic is a GEMM with K = planes x 9 taps: per 32 pixels x 32 channels one
// v_mfma_f32_32x32x2_f32 per tap (exact fp32 products, fp32 accumulation; the two planes ARE the K pair: lanes 0-31 feed mu,
// lanes 32-63 c_in * x of their pixel at that tap), the 1x1 shortcut one more on the centre tap.  A wave owns 32 consecutive
// pixels of an utterance x all C channels; the 18-27 weight registers per channel tile are loaded once per wave.
// Accumulator layout: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (pixel).
typedef float f32x16_fc __attribute__((ext_vector_type(16)));
template <int PLANES, int C>
__global__ __launch_bounds__(256) void first_conv_mfma_kernel(const FirstConvP p) {
    constexpr int NT = C / 32, CPG = C / 8, KS = PLANES == 3 ? 2 : 1;       // a third plane (speaker) rides a second K pair with a zero partner
    __shared__ long long red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const long npu = (long)p.H * p.T;
    const int step = p.step;
    const float c_in = p.scal[step * p.scal_stride + 2];
    const float* mrow = p.mask + (long)b * p.T;
    const float* plane = (hh == 0 ? p.mu : p.x) + (long)b * npu;           // this lane's K index = its plane
    const float psc = hh == 0 ? 1.f : c_in;
    if (tid < 16) red[tid] = 0;
    // B operands: lane (channel i of tile nt, K index hh) holds W[(plane*9 + tap)][nt*32 + i]
    float w3[KS][9][NT], w1[KS][NT], b3[NT], b1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        b3[nt] = p.b3[nt * 32 + i]; b1[nt] = p.b1[nt * 32 + i];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int pln = ks * 2 + hh;                                   // 0, 1 | 2, (3 = none)
            const bool has = pln < PLANES;
            w1[ks][nt] = has ? p.W1[pln * C + nt * 32 + i] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) w3[ks][t][nt] = has ? p.W3[(pln * 9 + t) * C + nt * 32 + i] : 0.f;
        }
    }
    float gs[NT], gq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { gs[nt] = 0.f; gq[nt] = 0.f; }
    __syncthreads();
    const long nseg = (npu + 31) / 32;
    // A operands: this lane's pixel at the nine taps, times the column mask (zero outside the image)
    auto gather = [&](long seg, float (&a)[KS][9]) __attribute__((always_inline)) {
        const long pl_raw = seg * 32 + i;
        const long pixl = pl_raw < npu ? pl_raw : npu - 1;
        const int w = (int)(pixl % p.T), h = (int)(pixl / p.T);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = h + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = w + kw - 1;
                const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.T;
                const int hc = inb ? hi : h, wc = inb ? wi : w;
                const float mk = inb ? mrow[wc] : 0.f;
                const float t0 = plane[(long)hc * p.T + wc] * psc;
                a[0][kh * 3 + kw] = t0 * mk;
                if constexpr (PLANES == 3) a[1][kh * 3 + kw] = hh == 0 ? p.spk[(long)b * p.H + hc] * mk : 0.f;
            }
        }
    };
    // Strip mode (T a multiple of 32, enough waves: the batch regime).  A wave owns the 32 columns cs*32.. of a band of rows and
    // walks DOWN it: the column masks of its three taps are loaded once, and a new output row needs only the three taps of ONE new
    // input row per plane - 3 load instructions per segment instead of 18-27 (the generic gather above is load-issue-bound: 100 us at
    // B = 32 for a 168 MB output).  The other six taps are the previous segment's, shifted.
    const long nwaves = (long)gridDim.x * 4, wv = (long)blockIdx.x * 4 + wave;
    const int nstrip = p.T / 32;
    const bool strip = (p.T % 32) == 0 && nwaves >= nstrip;
    const int cs = strip ? (int)(wv % nstrip) : 0;
    float mkc[3] = {0.f, 0.f, 0.f};
    if (strip) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { const int wi = cs * 32 + i + kw - 1; mkc[kw] = (unsigned)wi < (unsigned)p.T ? mrow[wi] : 0.f; }
    }
    auto gather_row = [&](int hi, float (&a)[KS][9], int slot) __attribute__((always_inline)) {      // taps of input row hi -> a[.][slot*3 .. +3]
        const bool rin = (unsigned)hi < (unsigned)p.H;
        const int hc = rin ? hi : 0;
        const float sp = (PLANES == 3 && hh == 0) ? p.spk[(long)b * p.H + hc] : 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = cs * 32 + i + kw - 1;
            const int wc = (unsigned)wi < (unsigned)p.T ? wi : cs * 32 + i;
            const float mk = rin ? mkc[kw] : 0.f;
            a[0][slot * 3 + kw] = (plane[(long)hc * p.T + wc] * psc) * mk;
            if constexpr (PLANES == 3) a[1][slot * 3 + kw] = sp * mk;
        }
    };
    long seg, seg_end, sstride;
    if (strip) {
        const int nrc = (int)(nwaves / nstrip), rc = (int)(wv / nstrip);
        const int R = (p.H + nrc - 1) / nrc, h0 = rc * R, h1 = rc < nrc ? min(p.H, h0 + R) : h0;
        seg = (long)h0 * nstrip + cs; seg_end = (long)h1 * nstrip; sstride = nstrip;
    } else { seg = wv; seg_end = nseg; sstride = nwaves; }
    float a[KS][9], an[KS][9];
    if (seg < seg_end) {
        if (strip) { const int h = (int)(seg / nstrip); gather_row(h - 1, a, 0); gather_row(h, a, 1); gather_row(h + 1, a, 2); }
        else gather(seg, a);
    }
    for (; seg < seg_end; seg += sstride) {
        if (seg + sstride < seg_end) {                                      // the next segment's operands fly under this one's MFMAs and stores
            if (strip) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int t = 0; t < 6; ++t) an[ks][t] = a[ks][t + 3];
                gather_row((int)(seg / nstrip) + 2, an, 2);
            } else gather(seg + sstride, an);
        }
        f32x16_fc acc[NT], accr[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[nt][r] = 0.f; accr[nt][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][t], w3[ks][t][nt], acc[nt], 0, 0, 0);
        if (p.res) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) accr[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][4], w1[ks][nt], accr[nt], 0, 0, 0);
        }
        // epilogue: + bias, statistics, stores.  Lane = one channel of 16 pixel rows; for 16-bit h1 neighbouring lanes swap half of
        // their rows so that each stores channel PAIRS (dword stores), as in the convolution epilogues.
        const long p0 = seg * 32 + 4 * hh;                      // pixel of accumulator row 0 of this lane
        const bool full = seg * 32 + 32 <= npu;
        const bool odd = (lane & 1) != 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + i;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[nt][r] + b3[nt];
                const bool ok = full || p0 + (r & 3) + 8 * (r >> 2) < npu;
                const float vs = ok ? v[r] : 0.f;
                gs[nt] += vs; gq[nt] = fmaf(vs, vs, gq[nt]);
            }
            if (p.h1_bf16 && full) {
                unsigned short* hp = reinterpret_cast<unsigned short*>(p.h1) + ((long)b * npu + p0) * C + n + (odd ? 16 * C - 1 : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float lo_r = v[j], hi_r = v[8 + j];
                    const float recv = lane_xor1(odd ? lo_r : hi_r);
                    const float mine = odd ? hi_r : lo_r;
                    const unsigned pk = odd ? pack2_kind(recv, mine, p.h1_bf16) : pack2_kind(mine, recv, p.h1_bf16);
                    *reinterpret_cast<unsigned*>(hp + ((j & 3) + 8 * (j >> 2)) * C) = pk;        // (an LDS-staged, 16 B per lane store was measured: 89 -> 93 us)
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long pr = p0 + (r & 3) + 8 * (r >> 2);
                    if (pr < npu) {
                        if (p.h1_bf16) reinterpret_cast<unsigned short*>(p.h1)[((long)b * npu + pr) * C + n] = (unsigned short)(pack2_kind(v[r], 0.f, p.h1_bf16) & 0xffffu);
                        else p.h1[((long)b * npu + pr) * C + n] = v[r];
                    }
                }
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long pr = p0 + (r & 3) + 8 * (r >> 2);
                    if (pr < npu) p.res[((long)b * npu + pr) * C + n] = accr[nt][r] + b1[nt];
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 9; ++t) a[ks][t] = an[ks][t];
    }
    if (p.gn_stats) {
        const double inv_n = 1.0 / ((double)p.H * p.T * CPG);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float a_ = gs[nt], q_ = gq[nt];
            for (int o = 1; o < CPG; o <<= 1) { a_ += __shfl_xor(a_, o); q_ += __shfl_xor(q_, o); }
            a_ += __shfl_xor(a_, 32); q_ += __shfl_xor(q_, 32);
            if (hh == 0 && (i & (CPG - 1)) == 0) {        // a wave's sums come out of a fixed order; the integer adds commute
                const int g = (nt * 32 + i) / CPG;
                gn_add(&red[g * 2], gn_fix(a_, inv_n)); gn_add(&red[g * 2 + 1], gn_fix(q_, inv_n));
            }
        }
        __syncthreads();
        if (tid < 16) gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2 + (tid & 1), red[tid]);
    }
}

void launch_first_conv(const FirstConvP& p, hipStream_t st) {
    const long npu = (long)p.H * p.T;
    const int ppb = p.C == 128 ? 32 : 64;
    long blocks = (npu + ppb - 1) / ppb;                          // one group per block at small batch ...
    const long capt = knob_or("DEX_FIRST_CAP", 4096);
    const long cap = capt / p.B > 32 ? capt / p.B : 32;           // ... about two rounds of resident blocks at large batch
    if (blocks > cap) blocks = cap;
    const int mfma = knob_or("DEX_FIRST_MFMA", 1);       // 0: the VALU form
    if (mfma) {
        long nb = (npu + 127) / 128;                               // 4 waves x 32 pixels per block pass
        const long capm = knob_or("DEX_FIRST_CAP", p.B >= 8 ? 512 : 1024);   // measured at B=32: 8192 blocks 122 us, 4096 112, 2048 103, 1024 99; round 5: 1536 / 1024 / 512 = 97.5 / 91.7 / 89.1 (one round of the 512 slots)
        const long cm = capm / p.B > 16 ? capm / p.B : 16;
        if (nb > cm) nb = cm;
        const dim3 g2((unsigned)nb, p.B);
        if (p.C == 128) { if (p.planes == 3) hipLaunchKernelGGL((first_conv_mfma_kernel<3, 128>), g2, dim3(256), 0, st, p); else hipLaunchKernelGGL((first_conv_mfma_kernel<2, 128>), g2, dim3(256), 0, st, p); }
        else { if (p.planes == 3) hipLaunchKernelGGL((first_conv_mfma_kernel<3, 64>), g2, dim3(256), 0, st, p); else hipLaunchKernelGGL((first_conv_mfma_kernel<2, 64>), g2, dim3(256), 0, st, p); }
        return;
    }
    const dim3 grid((unsigned)blocks, p.B);
    if (p.C == 128) {
        if (p.planes == 3) hipLaunchKernelGGL((first_conv_kernel<3, 128>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((first_conv_kernel<2, 128>), grid, dim3(256), 0, st, p);
        return;
    }
    if (p.planes == 3) hipLaunchKernelGGL((first_conv_kernel<3, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((first_conv_kernel<2, 64>), grid, dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics: grid (chunks, B); each block reduces GN_PIX pixels x C channels into
// per-group fixed-point (mean, mean-of-squares) contributions, block-combined in LDS, then one atomic pair per group.
constexpr int GN_PIX = 256;
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnStatsP p) {
    __shared__ long long red[32][2];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int C4 = p.C >> 2, cpg = p.C / p.groups;
    if (tid < 32) { red[tid][0] = 0; red[tid][1] = 0; }
    __syncthreads();
    const int cq = tid % C4, prow = tid / C4, rpp = 256 / C4;
    const int pbeg = blockIdx.x * GN_PIX;
    const int pend = min(p.npix, pbeg + GN_PIX);
    const float* X = p.X + (long)b * p.bstride;
    float s = 0.f, ss = 0.f;
    for (int px = pbeg + prow; px < pend; px += rpp) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long)px * p.ld + cq * 4);
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const int g = (cq * 4) / cpg;
    const double inv_n = 1.0 / ((double)p.npix * cpg);
    gn_add(&red[g][0], gn_fix(s, inv_n));          // per-thread sums in a fixed order; the integer adds commute
    gn_add(&red[g][1], gn_fix(ss, inv_n));
    __syncthreads();
    if (tid < p.groups * 2) {
        const int gg = tid >> 1, k = tid & 1;
        gn_add(p.stats + (((long)b * p.groups + gg) * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * 2 + k, red[gg][k]);
    }
}
void launch_gn_stats(const GnStatsP& p, hipStream_t st) {
    dim3 grid((p.npix + GN_PIX - 1) / GN_PIX, p.B);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnApplyP p) {
    __shared__ float smean[32], srstd[32];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int C4 = p.C >> 2, cpg = p.C / p.groups;
    gn_mean_rstd(p.stats, b, p.groups, smean, srstd, tid);
    __syncthreads();
    const int step = p.step;
    const long total = (long)p.npix * C4;
    const float* X = p.X + (long)b * p.xb;
    float* Y = p.Y + (long)b * p.yb + p.y_coff;
    const float* R = p.res ? p.res + (long)b * p.resb : nullptr;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    for (long idx = (long)blockIdx.x * 256 + tid; idx < total; idx += (long)gridDim.x * 256) {
        const int cq = (int)(idx % C4);
        const long px = idx / C4;
        const int w = (int)(px % p.W);
        const int c = cq * 4, g = c / cpg;
        const float mean = smean[g], rstd = srstd[g];
        const float4 x = *reinterpret_cast<const float4*>(X + px * p.ldx + c);
        const float4 ga = *reinterpret_cast<const float4*>(p.gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(p.beta + c);
        float4 y;
        y.x = mish_f((x.x - mean) * rstd * ga.x + be.x);
        y.y = mish_f((x.y - mean) * rstd * ga.y + be.y);
        y.z = mish_f((x.z - mean) * rstd * ga.z + be.z);
        y.w = mish_f((x.w - mean) * rstd * ga.w + be.w);
        if (p.tadd) {
            const float4 t = *reinterpret_cast<const float4*>(p.tadd + (long)step * p.tadd_step_stride + c);
            y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
        }
        const float mk = mrow[w * p.mask_ws];
        if (R && p.res_under_mask) {      // identity shortcut: h + (x * mask)
            const float4 r = *reinterpret_cast<const float4*>(R + px * p.ldres + c);
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        y.x *= mk; y.y *= mk; y.z *= mk; y.w *= mk;
        if (R && !p.res_under_mask) {     // res_conv(x * mask): bias survives in padded columns
            const float4 r = *reinterpret_cast<const float4*>(R + px * p.ldres + c);
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        *reinterpret_cast<float4*>(Y + px * p.ldy + c) = y;
    }
}
void launch_gn_apply(const GnApplyP& p, hipStream_t st) {
    const long total = (long)p.npix * (p.C / 4);
    long blocks = (total + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, p.B), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
// final: per pixel  f = mask * (b + sum_c w[c] * mask * Mish(GN(x)[c]));  D = c_skip*x + c_out*f;
//        x_next = x + (t_next - t)/t * (x - D).          scal row: [sigma, sigma_next, c_in, c_skip, c_out, c_noise]
// 8 lanes cooperate on one pixel (8 channels per lane and 64-channel block: one 16-byte load of bf16 / fp16, two of fp32).
// A block owns FIN_U groups of 32 pixels per pass and issues every load of the pass - activations, mask, sampler state -
// before the first use (the first version walked its pixels one dependent load at a time: 125 us for 168 MB at B=32);
// the per-channel GroupNorm coefficient and final_conv weight live in registers for the block's life; Mish on packed pairs.
constexpr int FIN_U = 4;
__global__ __launch_bounds__(256) void final_kernel(const FinalP p) {
    __shared__ float smean[32], srstd[32];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.groups;
    gn_mean_rstd(p.stats, b, p.groups, smean, srstd, tid);
    __syncthreads();
    const int step = p.step;
    const float* sc = p.scal + (long)step * p.scal_stride;
    const float sigma = sc[0], sigma_next = sc[1], c_skip = sc[3], c_out = sc[4];
    const float* X = p.X + (long)b * p.xb;
    const unsigned short* Xh = reinterpret_cast<const unsigned short*>(p.X) + (long)b * p.xb;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const int sub = tid & 7;
    const int nblk = p.C / 64;                       // 1, or 2 for the 128-channel geometry
    f32x2 ca[2][4], cb[2][4], cw[2][4];              // t = x * ca + cb;  out += Mish(t) * cw
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < nblk) {
            const int c = k * 64 + sub * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = c + 2 * j, c1 = c0 + 1;
                const float a0 = srstd[c0 / cpg] * p.gamma[c0], a1 = srstd[c1 / cpg] * p.gamma[c1];
                ca[k][j] = f32x2{a0, a1};
                cb[k][j] = f32x2{p.beta[c0] - smean[c0 / cpg] * a0, p.beta[c1] - smean[c1 / cpg] * a1};
                cw[k][j] = f32x2{p.wfc[c0], p.wfc[c1]};
            }
        }
    }
    const float bfc = p.bfc[0];
    // a lost workgroup hand-off earlier in this call (cluster form of the DiT block: bounded wait, dit_rowchain.hip) must not pass
    // as a plausible mel: every output of the call becomes NaN
    const bool poisoned = p.poison && *p.poison;
    const float inv = 1.f / sigma;
    const float h = p.htab ? p.htab[step] : sigma_next - sigma;
    const long stride = (long)gridDim.x * 32 * FIN_U;
    for (long base = (long)blockIdx.x * 32 * FIN_U; base < p.npix; base += stride) {
        uint4 raw[FIN_U][2][2];                      // [pass][64-channel block][fp32: two halves; bf16: [0] only]
        float mk[FIN_U], xc[FIN_U], xa[FIN_U], xd[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const long pr = base + u * 32 + (tid >> 3);
            const long px = pr < p.npix ? pr : p.npix - 1;
            const long o = (long)b * p.npix + px;
            mk[u] = mrow[(int)(px % p.W)];
            xc[u] = p.xcur[o];
            xa[u] = p.mode == 2 ? p.xhat[o] : 0.f;
            xd[u] = p.mode == 2 ? p.dbuf[o] : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nblk) {
                    if (p.x_bf16) raw[u][k][0] = *reinterpret_cast<const uint4*>(Xh + px * p.C + k * 64 + sub * 8);
                    else {
                        raw[u][k][0] = *reinterpret_cast<const uint4*>(X + px * p.C + k * 64 + sub * 8);
                        raw[u][k][1] = *reinterpret_cast<const uint4*>(X + px * p.C + k * 64 + sub * 8 + 4);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            f32x2 acc2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nblk) {
                    f32x2 x[4];
                    if (p.x_bf16) {
                        const uint4 r = raw[u][k][0];
                        x[0] = f32x2{lo_kind(r.x, p.x_bf16), hi_kind(r.x, p.x_bf16)}; x[1] = f32x2{lo_kind(r.y, p.x_bf16), hi_kind(r.y, p.x_bf16)};
                        x[2] = f32x2{lo_kind(r.z, p.x_bf16), hi_kind(r.z, p.x_bf16)}; x[3] = f32x2{lo_kind(r.w, p.x_bf16), hi_kind(r.w, p.x_bf16)};
                    } else {
                        const uint4 r0 = raw[u][k][0], r1 = raw[u][k][1];
                        x[0] = f32x2{__uint_as_float(r0.x), __uint_as_float(r0.y)}; x[1] = f32x2{__uint_as_float(r0.z), __uint_as_float(r0.w)};
                        x[2] = f32x2{__uint_as_float(r1.x), __uint_as_float(r1.y)}; x[3] = f32x2{__uint_as_float(r1.z), __uint_as_float(r1.w)};
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2 = mish2_add(x[j] * ca[k][j] + cb[k][j], f32x2{0.f, 0.f}) * cw[k][j] + acc2;
                }
            }
            float acc = acc2.x + acc2.y;
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            const long pr = base + u * 32 + (tid >> 3);
            if (sub == 0 && pr < p.npix) {
                const float f = (acc * mk[u] + bfc) * mk[u];
                const long o = (long)b * p.npix + pr;            // [B,80,T] has the same (h*T + w) linear order
                // The sampler's scalar arithmetic repeats the reference's fp32 operations ONE ROUNDING AT A TIME (edm.py:96-97,
                // 199-214: every product and sum is its own torch op; no fused multiply-add).  It matters: the Heun corrector
                // evaluates the network at sigma' = 0.002 with a step h of the previous noise level, so one ulp of D' enters
                // x_next multiplied by h / (2 sigma') ~ 1e2 (measured before: 3e-4 against the oracle at n = 4, now ~1e-5).
                const float D = poisoned ? __builtin_nanf("") : __fadd_rn(__fmul_rn(c_skip, xc[u]), __fmul_rn(c_out, f));
                if (p.denoised) p.denoised[o] = D;
                if (p.xnext) {
                    const float d = __fsub_rn(__fmul_rn(inv, xc[u]), __fmul_rn(inv, D));
                    if (p.mode == 2) {
                        p.xnext[o] = __fadd_rn(xa[u], __fmul_rn(h, __fadd_rn(__fmul_rn(0.5f, xd[u]), __fmul_rn(0.5f, d))));
                    } else {
                        if (p.mode == 1) p.dbuf[o] = d;
                        p.xnext[o] = __fadd_rn(xc[u], __fmul_rn(h, d));
                    }
                }
            }
        }
    }
    if (p.zero_ptr) {          // clear the statistics arena the NEXT Euler step accumulates into (other parity)
        const long nthreads = (long)gridDim.x * gridDim.y * 256;
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < p.zero_n; i += nthreads) p.zero_ptr[i] = 0.f;
    }
}
__global__ void churn_tables_kernel(const float* sigmas, int n, float S_churn, float S_min, float S_max, float S_noise,
                                    float* t_hat, float* h, float* ncoef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { t_hat[n] = 0.f; return; }
    const float t = sigmas[i];
    // gamma is a python float in the reference (edm.py:194); multiplying the fp32 0-dim tensor t_cur by it happens in fp32
    const double gd = fmin((double)S_churn / (double)n, sqrt(2.0) - 1.0);
    const bool on = S_churn > 0.f && (double)S_min <= (double)t && (S_max <= 0.f || (double)t <= (double)S_max);
    const float g = on ? (float)gd : 0.f;
    const float th = __fadd_rn(t, __fmul_rn(g, t));                  // t_cur + gamma * t_cur, no contraction
    t_hat[i] = th;
    h[i] = __fsub_rn(sigmas[i + 1], th);
    const float d2 = __fsub_rn(__fmul_rn(th, th), __fmul_rn(t, t));
    ncoef[i] = __fmul_rn(sqrtf(fmaxf(d2, 0.f)), S_noise);
}
void launch_churn_tables(const float* sigmas, int n, float S_churn, float S_min, float S_max, float S_noise,
                         float* t_hat, float* h, float* ncoef, hipStream_t st) {
    hipLaunchKernelGGL(churn_tables_kernel, dim3((n + 64) / 64), dim3(64), 0, st, sigmas, n, S_churn, S_min, S_max, S_noise, t_hat, h, ncoef);
}
__global__ void add_noise_kernel(float* x, const float* noise, const float* ncoef, long n) {
    const float c = *ncoef;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = __fadd_rn(x[i], __fmul_rn(c, noise[i]));
}
void launch_add_noise(float* x, const float* noise, const float* ncoef_i, long n, hipStream_t st) {
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, noise, ncoef_i, n);
}
__global__ void heun_expand_kernel(const float* t_hat, const float* hin, int n, float* sig, float* h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = t_hat[i], hh = hin[i];
    sig[2 * i] = t; h[2 * i] = hh;
    if (i < n - 1) { sig[2 * i + 1] = t + hh; h[2 * i + 1] = hh; }
    else sig[2 * i + 1] = 0.f;
}
void launch_heun_expand(const float* t_hat, const float* hin, int n, float* sig, float* h, hipStream_t st) {
    hipLaunchKernelGGL(heun_expand_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t_hat, hin, n, sig, h);
}
void launch_final(const FinalP& p, hipStream_t st) {
    // every block pays the GroupNorm-coefficient prologue (fp64 divide + sqrt behind a barrier, ~2 us): at large batch
    // keep the total near 16K blocks so each one streams several 16-pixel groups instead of one
    long blocks = (p.npix + 32 * FIN_U - 1) / (32 * FIN_U);      // one pass of FIN_U 32-pixel groups per block ...
    // ... several passes at large batch: every block pays the GroupNorm-coefficient prologue (fp64 divide + sqrt behind a barrier,
    // 72 coefficient loads), so about one round of resident blocks is best (measured at B=32: 16384 blocks 87 us, 8192 85, 4096 74,
    // 2048 68)
    // (round 5: three workgroups fit a CU - 768 slots; 768 / 1536 / 2048 / 3072 blocks = 79.5 / 69.5 / 69.8 / 72.6 us at GeDEX B = 32, 40.2 / 38.8 / 40.4 / 41.8 at DEX B = 32)
    const long capt = knob_or("DEX_FINAL_CAP", 1536);
    const long cap = capt / p.B > 32 ? capt / p.B : 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(final_kernel, dim3((unsigned)blocks, p.B), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
// Conditioning prep: thread i = Euler step i.  fp32 math in the reference's op order
// (edm.py:90-94; SinusoidalPosEmb diffusion.py:110-117; timestep_embedding dit.py:250-254).
__global__ void cond_prep_kernel(const CondPrepP p) {
    const int i = blockIdx.x;
    const int k = threadIdx.x;
    const float sigma = p.sigmas[i], sigma_next = p.sigmas[i + 1];
    const float sd = 0.5f;
    const float c_noise = logf(sigma) / 4.f;
    if (k == 0) {
        float* sc = p.scal + (long)i * p.scal_stride;
        const float s2 = sigma * sigma + sd * sd;
        sc[0] = sigma; sc[1] = sigma_next;
        sc[2] = 1.f / sqrtf(s2);                 // c_in
        sc[3] = __fmul_rn(__frcp_rn(s2), sd * sd);   // c_skip = sigma_data^2 / (..): python float / tensor = tensor.reciprocal() * float in torch (two roundings)
        sc[4] = sigma * sd / sqrtf(s2);          // c_out
        sc[5] = c_noise;
    }
    const int half = p.dim / 2;
    if (k < half) {
        const float e = (float)(log(10000.0) / (double)(half - 1));   // python double -> fp32 scalar
        const float f = expf((float)k * -e);
        const float a = p.pe_scale * c_noise * f;
        p.t_unet[(long)i * p.dim + k] = sinf(a);
        p.t_unet[(long)i * p.dim + half + k] = cosf(a);
    }
    if (k < 128) {
        const float f = expf((float)(-log(10000.0)) * (float)k / 128.f);
        const float a = c_noise * f;
        p.t_dit[(long)i * 256 + k] = cosf(a);
        p.t_dit[(long)i * 256 + 128 + k] = sinf(a);
    }
}
void launch_cond_prep(const CondPrepP& p, hipStream_t st) {
    hipLaunchKernelGGL(cond_prep_kernel, dim3(p.n), dim3(128), 0, st, p);
}

// tiny conditioning MLPs: one wave per (output feature n, group of 8 rows); the 8 row loads are independent.
__global__ __launch_bounds__(256) void small_linear_kernel(const SmallLinP p) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.N) return;
    const float* w = p.W + (long)wave * p.K;
    const float bias = p.bias ? p.bias[wave] : 0.f;
    const int r0 = blockIdx.y * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = lane; k < p.K; k += 64) {
        const float wk = w[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = min(r0 + j, p.rows - 1);
            float v = p.X[(long)r * p.ldx + k];
            if (p.act_in == 1) v = mish_f(v); else if (p.act_in == 2) v = silu_f(v);
            acc[j] = fmaf(v, wk, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = acc[j];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0 && r0 + j < p.rows) {
            float y = a + bias;
            if (p.act_out == 1) y = mish_f(y); else if (p.act_out == 2) y = silu_f(y);
            p.Y[(long)(r0 + j) * p.ldy + wave] = y;
        }
    }
}
void launch_small_linear(const SmallLinP& p, hipStream_t st) {
    hipLaunchKernelGGL(small_linear_kernel, dim3((p.N * 64 + 255) / 256, (p.rows + 7) / 8), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------------
__global__ void scale_copy_kernel(const float* src, float* dst, long n, const float* scal) {
    const float s = scal ? *scal : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i] * s;
}
void launch_scale_copy(const float* src, float* dst, long n, const float* scal_ptr, hipStream_t st) {
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n, scal_ptr);
}
__global__ void iota_kernel(int* d, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = i; }
void launch_iota(int* dst, int n, hipStream_t st) { hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, n); }
__global__ void step_reset_kernel(int* s) { *s = 0; }
__global__ void step_inc_kernel(int* s) { *s = *s + 1; }
void launch_step_reset(int* step, hipStream_t st) { hipLaunchKernelGGL(step_reset_kernel, dim3(1), dim3(1), 0, st, step); }
void launch_step_inc(int* step, hipStream_t st) { hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step); }

__global__ void permute4_kernel(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3) {
    const long n = (long)d0 * d1 * d2 * d3;
    const int sd[4] = {d0, d1, d2, d3};
    const long ss[4] = {(long)d1 * d2 * d3, (long)d2 * d3, (long)d3, 1};
    const int pm[4] = {p0, p1, p2, p3};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long rem = i, off = 0;
        for (int a = 3; a >= 0; --a) {
            const int dim = sd[pm[a]];
            const int idx = (int)(rem % dim);
            rem /= dim;
            off += idx * ss[pm[a]];
        }
        dst[i] = src[off];
    }
}
void launch_permute4(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3, hipStream_t st) {
    const long n = (long)d0 * d1 * d2 * d3;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(permute4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, d0, d1, d2, d3, p0, p1, p2, p3);
}

__global__ void f32_to_lp_kernel(const float* src, unsigned short* dst, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = (unsigned short)(pack2_kind(src[i], 0.f, kind) & 0xffffu);            // round to nearest even
}
void launch_f32_to_lp(const float* src, void* dst, long n, int precision, hipStream_t st) {
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f32_to_lp_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, (unsigned short*)dst, n, precision);
}
// what rounding to the 16-bit type loses: dst = src - float(lp(src)) (the "lo" half of a split weight, packed like the weight itself)
__global__ void f32_residual_lp_kernel(const float* src, float* dst, long n, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned h = pack2_kind(src[i], 0.f, kind) & 0xffffu;
        const float hf = kind == 2 ? (float)__builtin_bit_cast(_Float16, (unsigned short)h) : __uint_as_float(h << 16);
        dst[i] = src[i] - hf;
    }
}
void launch_f32_residual_lp(const float* src, float* dst, long n, int precision, hipStream_t st) {
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f32_residual_lp_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n, precision);
}
// fp32 [K][N] -> low precision [N][K] (the MFMA GEMMs' weight operand: K contiguous)
__global__ void pack_lp_nk_kernel(const float* src, unsigned short* dst, int K, int N, int kind) {
    const long total = (long)K * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), n = (int)(i / K);
        dst[i] = (unsigned short)(pack2_kind(src[(long)k * N + n], 0.f, kind) & 0xffffu);
    }
}
void launch_pack_lp_nk(const float* src, void* dst, int K, int N, int precision, hipStream_t st) {
    hipLaunchKernelGGL(pack_lp_nk_kernel, dim3(256), dim3(256), 0, st, src, (unsigned short*)dst, K, N, precision);
}

// speaker plane: the reference repeats spk_mlp(spk) [B,80] along time (diffusion.py:173-175); we keep [B,80].
void launch_spk_plane(const float* spk_out, float* plane, int B, int F, hipStream_t st) {
    launch_scale_copy(spk_out, plane, (long)B * F, nullptr, st);
}

}  // namespace dex
