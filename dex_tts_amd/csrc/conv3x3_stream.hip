// conv3x3_stream.hip — 3x3 convolution 64 -> 64 channels for LARGE grids (batched synthesis), bf16 MFMA, fp32 in/out.
//
// The tile kernel in conv3x3_bf16.hip is built for latency (batch 1): every workgroup loads its whole patch up front,
// converts, runs nine taps with the weights streamed through LDS, stores.  With two workgroups per CU those phases
// barely overlap, and at batch 32 the launch sits at ~2.6 TB/s with the matrix pipe 15 % busy.  This kernel is the
// throughput form of the same arithmetic:
//   * one workgroup of eight waves per CU walks DOWN a 32-pixel-wide strip of the image, eight rows per iteration
//     (one wave per image row, 64 output channels each);
//   * all nine taps of the weights (9 x 64 x 64 bf16 = 72 KB) are loaded into LDS ONCE per workgroup, so the
//     iteration has no weight traffic, no per-tap barrier and no global load other than the next rows of the strip;
//   * the rows of iteration t+1 are requested before the MFMAs of iteration t and converted after them (two
//     8-row slots in LDS: the rows iteration t no longer needs are overwritten behind a barrier), so HBM latency is
//     covered by a whole iteration of matrix work;
//   * rows are read once per strip segment instead of once per 4-row tile (halo 1.05x instead of 1.5x), which also
//     cuts the fused GroupNorm + Mish prologue by a third;
//   * GroupNorm partial sums of the output stay in registers across the strip: one reduction per workgroup.
// Prologue forms are the ones of conv3x3_bf16.hip: PRO (producer's GroupNorm + Mish + time bias, runtime flag) and
// PRO2 (the previous ResnetBlock's tail: x = mask * Mish(GN(h2)) + res, also written out for later consumers).
// Reference: Block / ResnetBlock, diffusion.py:42-71.
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"
#include <cstdlib>

namespace dex {
namespace DEX_LP_NS {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SC = 64;                 // channels in and out
constexpr int SPW = 34;                // patch width: 32 pixels + halo
constexpr int SLDP = SC + 8;           // LDS row stride (u16) of one pixel / one weight row
constexpr int STR = 8;                 // image rows per iteration == waves per workgroup
constexpr int SNT = 512;               // threads
constexpr int SITEMS = STR * SPW * (SC / 8);          // 8-channel items of one 8-row chunk (2176)
constexpr int SNI = (SITEMS + SNT - 1) / SNT;         // per thread (5; the last one only for tid < 128)
constexpr size_t S_LDS = ((size_t)2 * STR * SPW * SLDP + (size_t)9 * SC * SLDP) * sizeof(u16);   // 161280 B

__device__ __forceinline__ float st_mish(float x) {
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
// x * a + b as an opaque instruction: the first operation on a prefetched value stays where it is written (below the
// MFMAs of the current iteration) instead of being hoisted to the load together with its s_waitcnt
__device__ __forceinline__ float fma_pinned(float x, float a, float b) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    return r;
}

struct ChunkRegs {
    f32x4 a[SNI], c[SNI];       // 8 channels of one pixel (bf16 input: a holds the 8 raw values, c is unused)
    float mk[SNI];              // mask of the pixel, 0 outside the image
};

template <bool PRO, bool PRO2, bool XB>
__global__ __launch_bounds__(SNT) void conv3x3_stream64_kernel(const Conv3P p, const int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* patch = smem;                                 // [2 slots][8 rows][34][SLDP]
    u16* wts = smem + 2 * STR * SPW * SLDP;            // [9 taps][64 cout][SLDP]
    __shared__ float smean[8], srstd[8];
    __shared__ long long gnred[16];
    __shared__ __attribute__((aligned(16))) float coef[3][SC];        // GroupNorm prologue per channel: scale, shift, time bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int w0 = blockIdx.x * 32, b = blockIdx.z;
    const int ntile_total = (p.H + STR - 1) / STR;
    const int t0 = blockIdx.y * tiles_per_wg;
    const int nt = min(tiles_per_wg, ntile_total - t0);
    const int hs = t0 * STR;
    const int step = p.step;
    const float* X = p.X + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.H * p.W * p.ldx + p.x_coff;      // XB: bf16 input
    const float* R = PRO2 ? p.pro_res + (long)b * p.H * p.W * SC : nullptr;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    constexpr bool pro = PRO;
    const int c8 = (tid & 7) * 8;                      // this thread's channel group in every chunk item
#ifdef DEX_TIMING
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk0 = __builtin_readcyclecounter();
#define STAMP(k) do { const long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
    long long tlast = tk0;
#else
#define STAMP(k) do {} while (0)
#endif

    // ---- set-up loads go out together: GroupNorm partials first (oldest in the queue: their wait leaves the rest in
    // flight), all nine taps of the weights, then (below) the first two row chunks
    float gn_ga = 0.f, gn_be = 0.f, gn_ta = 0.f;
    uint4 gn_raw = make_uint4(0u, 0u, 0u, 0u);                  // one slot's fixed-point (mean, mean-of-squares) pair
    if constexpr (PRO) {
        if (tid < 8 * GN_SLOTS)
            gn_raw = *reinterpret_cast<const uint4*>(p.pro_stats + (((long)b * 8 + tid / GN_SLOTS) * GN_SLOTS + (tid % GN_SLOTS)) * 2);
        if (tid < SC) { gn_ga = p.pro_gamma[tid]; gn_be = p.pro_beta[tid]; gn_ta = p.pro_tadd ? p.pro_tadd[(long)step * SC + tid] : 0.f; }
    }
    u32x4 wr[9];
    {
        const u16* Wg = reinterpret_cast<const u16*>(p.Wbf);          // [64][9*64]
#pragma unroll
        for (int q = 0; q < 9; ++q) wr[q] = *reinterpret_cast<const u32x4*>(Wg + (long)(tid >> 3) * (9 * SC) + q * SC + c8);
    }

    // chunk c = image rows hs - 1 + 8c .. + 7, LDS slot c & 1
    auto load_chunk = [&](int c, ChunkRegs& x, ChunkRegs& r) __attribute__((always_inline)) {
        // the mask values first: their select is an ordinary instruction the compiler places right behind its load, and
        // loads retire in order - behind the big loads that wait would drain the whole prefetch before the MFMAs
#pragma unroll
        for (int q = 0; q < SNI; ++q) {
            const int pxi = min((tid >> 3) + (SNT / 8) * q, STR * SPW - 1);
            const int j = pxi / SPW, pw = pxi - j * SPW;
            const int hi = hs - 1 + STR * c + j, wi = w0 + pw - 1;
            const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float mk = mrow[(inb ? wi : 0) * p.mask_ws];
            x.mk[q] = inb ? mk : 0.f;
        }
#pragma unroll
        for (int q = 0; q < SNI; ++q) {
            const int pxi = min((tid >> 3) + (SNT / 8) * q, STR * SPW - 1);
            const int j = pxi / SPW, pw = pxi - j * SPW;
            const int hi = hs - 1 + STR * c + j, wi = w0 + pw - 1;
            const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int hc = inb ? hi : 0, wc = inb ? wi : 0;
            if constexpr (XB) {
                x.a[q] = *reinterpret_cast<const f32x4*>(Xh + ((long)hc * p.W + wc) * p.ldx + c8);     // 8 raw bf16
            } else {
                const float* src = X + ((long)hc * p.W + wc) * p.ldx + c8;
                x.a[q] = *reinterpret_cast<const f32x4*>(src);
                x.c[q] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            if constexpr (PRO2) {
                const float* rs = R + ((long)hc * p.W + wc) * SC + c8;
                r.a[q] = *reinterpret_cast<const f32x4*>(rs);
                r.c[q] = *reinterpret_cast<const f32x4*>(rs + 4);
            }
        }
    };
    auto store_chunk = [&](int c, const ChunkRegs& x, const ChunkRegs& r) __attribute__((always_inline)) {
        u16* dst = patch + (c & 1) * STR * SPW * SLDP;
        const int row_lo = hs, row_hi = min(hs + STR * nt, p.H);        // image rows this workgroup owns (PRO2 write-out)
#pragma unroll
        for (int q = 0; q < SNI; ++q) {
            const int pxi = (tid >> 3) + (SNT / 8) * q;
            const float mk = x.mk[q];
            float v[8];
            if constexpr (XB) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned u = __float_as_uint(x.a[q][k]);
                    v[2 * k] = lp_lo(u); v[2 * k + 1] = lp_hi(u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = x.a[q][k]; v[4 + k] = x.c[q][k]; }
            }
            uint4 o;
            if constexpr (PRO) {
                f32x2 sc[4], sh[4], ta[4], w[4];
                *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(&coef[0][c8]); *reinterpret_cast<float4*>(sc + 2) = *reinterpret_cast<const float4*>(&coef[0][c8 + 4]);
                *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(&coef[1][c8]); *reinterpret_cast<float4*>(sh + 2) = *reinterpret_cast<const float4*>(&coef[1][c8 + 4]);
                *reinterpret_cast<float4*>(ta) = *reinterpret_cast<const float4*>(&coef[2][c8]); *reinterpret_cast<float4*>(ta + 2) = *reinterpret_cast<const float4*>(&coef[2][c8 + 4]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2 in; in.x = v[2 * k]; in.y = v[2 * k + 1];
                    w[k] = mish2_add(pk_fma_pinned(in, sc[k], sh[k]), ta[k]);
                }
                if constexpr (PRO2) {
                    const f32x2 rr[4] = {{r.a[q][0], r.a[q][1]}, {r.a[q][2], r.a[q][3]}, {r.c[q][0], r.c[q][1]}, {r.c[q][2], r.c[q][3]}};
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = w[k] * mk + rr[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] = w[k].x; v[2 * k + 1] = w[k].y; }
                    const int j = pxi / SPW, pw = pxi - j * SPW;
                    const int hi = hs - 1 + STR * c + j;
                    if (pxi < STR * SPW && hi >= row_lo && hi < row_hi && pw >= 1 && pw <= 32 && w0 + pw - 1 < p.W) {
                        const long eo = ((long)b * p.H * p.W + (long)hi * p.W + (w0 + pw - 1)) * SC + c8;
                        if (p.xout_lp) {       // (uniform) 16-bit x for its one reader, the attention's context pass
                            *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.pro_xout) + eo) =
                                make_uint4(pack2_lp(v[0], v[1]), pack2_lp(v[2], v[3]), pack2_lp(v[4], v[5]), pack2_lp(v[6], v[7]));
                        } else {
                            float* xo = p.pro_xout + eo;
                            *reinterpret_cast<float4*>(xo) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(xo + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = w[k] * mk;
                o.x = pack2_lp(w[0].x, w[0].y); o.y = pack2_lp(w[1].x, w[1].y);
                o.z = pack2_lp(w[2].x, w[2].y); o.w = pack2_lp(w[3].x, w[3].y);
            } else {
                o.x = pack2_mul_lp_pinned(v[0], v[1], mk); o.y = pack2_mul_lp_pinned(v[2], v[3], mk);
                o.z = pack2_mul_lp_pinned(v[4], v[5], mk); o.w = pack2_mul_lp_pinned(v[6], v[7], mk);
            }
            if (pxi < STR * SPW) *reinterpret_cast<uint4*>(dst + pxi * SLDP + c8) = o;
        }
    };

    ChunkRegs cx, cr, cx1;
    load_chunk(0, cx, cr);
    if constexpr (!PRO2) load_chunk(1, cx1, cr);              // (PRO2 carries the residual rows too: one chunk at a time)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRO) {        // prologue coefficients per channel:  y = Mish(x * coef0 + coef1) + coef2
        if (tid < 8 * GN_SLOTS) {
            long long s1 = (long long)(((unsigned long long)gn_raw.y << 32) | gn_raw.x);
            long long s2 = (long long)(((unsigned long long)gn_raw.w << 32) | gn_raw.z);
            gn_slots_reduce<GN_SLOTS>(s1, s2);
            if ((tid % GN_SLOTS) == 0) gn_moments(s1, s2, 1e-5, smean[tid / GN_SLOTS], srstd[tid / GN_SLOTS]);
        }
        lds_barrier();
        if (tid < SC) {
            const float mean = smean[tid / 8], rstd = srstd[tid / 8];
            coef[0][tid] = rstd * gn_ga;
            coef[1][tid] = gn_be - mean * rstd * gn_ga;
            coef[2][tid] = gn_ta;
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) *reinterpret_cast<u32x4*>(wts + (q * SC + (tid >> 3)) * SLDP + c8) = wr[q];
    lds_barrier();                                            // coefficients visible before the first conversion
    store_chunk(0, cx, cr);
    if constexpr (PRO2) { load_chunk(1, cx, cr); store_chunk(1, cx, cr); }
    else store_chunk(1, cx1, cr);
    lds_barrier();

    STAMP(0);
    const float bias0 = p.bias[i], bias1 = p.bias[32 + i];
    float gs[2] = {0.f, 0.f}, gss[2] = {0.f, 0.f};
    const bool yb = p.y_bf16 != 0;

    // Epilogue of one wave's image row (32 pixels x 64 channels): + bias, GroupNorm partials, 32 stores from one 64-bit base
    // per lane (a lane owns ONE channel of 16 pixel rows; each store covers two full 128-byte lines).  Two alternatives
    // were measured at B=32 and lost: (a) the tile through a wave-private LDS scratch so that it leaves as 16-byte chunks,
    // 8 stores of 1 KB instead of 32 (164 vs 156 us: one more barrier, same time inside the stores); (b) the previous tile's
    // stores trickled into the next MFMA loop, one per K-step, with a second accumulator set (175 us: each store still
    // stalls its wave ~150 cycles, now inside the MFMA loop).  The epilogue waits for the CU's memory pipeline either way.
    auto emit_tile = [&](const f32x16 (&acc)[2], int t) __attribute__((always_inline)) {
        const int ho = hs + STR * t + wave;
        const long off = (long)b * p.H * p.W * SC + ((long)ho * p.W + w0 + 4 * hh) * SC + i;
        float* yf = p.Y + off; u16* yh = reinterpret_cast<u16*>(p.Y) + off;
        const bool full = ho < p.H && w0 + 32 <= p.W;
        if (full && yb) {          // channel pairs of 8 rows per lane after a DPP swap with the neighbouring lane (see cv_epilogue)
            const bool odd = (lane & 1) != 0;
            u16* yp = yh + (odd ? 16 * SC - 1 : 0);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const float bias = n2 ? bias1 : bias0;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[n2][r] + bias; gs[n2] += v; gss[n2] = fmaf(v, v, gss[n2]); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float lo_r = acc[n2][j] + bias, hi_r = acc[n2][8 + j] + bias;
                    const float recv = lane_xor1(odd ? lo_r : hi_r);
                    const float mine = odd ? hi_r : lo_r;
                    const unsigned pk = odd ? pack2_lp(recv, mine) : pack2_lp(mine, recv);
                    *reinterpret_cast<unsigned*>(yp + ((j & 3) + 8 * (j >> 2)) * SC + n2 * 32) = pk;
                }
            }
            return;
        }
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const float bias = n2 ? bias1 : bias0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = full || (ho < p.H && w0 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.W);
                const float v = acc[n2][r] + bias;
                const float vs = ok ? v : 0.f;
                gs[n2] += vs; gss[n2] = fmaf(vs, vs, gss[n2]);
                if (ok) {
                    if (yb) yh[((r & 3) + 8 * (r >> 2)) * SC + n2 * 32] = lp_bits(v);
                    else yf[((r & 3) + 8 * (r >> 2)) * SC + n2 * 32] = v;
                }
            }
        }
    };
    // iteration t: request the rows of iteration t+1, then the 72 MFMAs of this wave's image row
    auto iteration = [&](int t, f32x16 (&cur)[2]) __attribute__((always_inline)) {
        const bool more = t + 1 < nt;
        if (more) load_chunk(t + 2, cx, cr);                  // rows of the NEXT iteration; in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { cur[0][r] = 0.f; cur[1][r] = 0.f; }
        // 36 (tap, K-step) pairs, the fragments of pair s+1 are read from LDS before the MFMAs of pair s are issued
        const u16* arow[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int r = wave + kh;                                          // row of the 10-row window
            arow[kh] = patch + ((((t + (r >> 3)) & 1) * STR + (r & 7)) * SPW + i) * SLDP + hh * 8;
        }
        const u16* brow = wts + i * SLDP + hh * 8;
        lp8 af[2], b0[2], b1[2];
        af[0] = *reinterpret_cast<const lp8*>(arow[0]);
        b0[0] = *reinterpret_cast<const lp8*>(brow);
        b1[0] = *reinterpret_cast<const lp8*>(brow + 32 * SLDP);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 1 < 36) {
                const int tap = (s + 1) >> 2, ks = (s + 1) & 3, kh = tap / 3, kw = tap - kh * 3;
                af[(s + 1) & 1] = *reinterpret_cast<const lp8*>(arow[kh] + kw * SLDP + ks * 16);
                b0[(s + 1) & 1] = *reinterpret_cast<const lp8*>(brow + tap * SC * SLDP + ks * 16);
                b1[(s + 1) & 1] = *reinterpret_cast<const lp8*>(brow + (tap * SC + 32) * SLDP + ks * 16);
            }
            __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks each read to right above its MFMA)
            cur[0] = DEX_MFMA_LP(af[s & 1], b0[s & 1], cur[0], 0, 0, 0);
            cur[1] = DEX_MFMA_LP(af[s & 1], b1[s & 1], cur[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16 acc[2];
    for (int t = 0; t < nt; ++t) {
        iteration(t, acc);
        STAMP(2);
        emit_tile(acc, t);
        STAMP(3);
        if (t + 1 < nt) {
            lds_barrier();                                    // every wave is done reading the slot of chunk t
            STAMP(4);
            store_chunk(t + 2, cx, cr);
            STAMP(5);
            lds_barrier();
            STAMP(6);
        }
    }

#ifdef DEX_TIMING
    if (p.dbg && tid == 0) {
        long long* d = p.dbg + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        for (int k = 0; k < 7; ++k) d[k] = tk[k];
        d[7] = __builtin_readcyclecounter() - tk0;
    }
#endif
    if (p.gn_stats) {
        if (tid < 16) gnred[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            float a = gs[n2], q = gss[n2];
            for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            if (hh == 0 && (i & 7) == 0) {         // a wave's sums come out of a fixed order; the integer adds commute
                const int g = (n2 * 32 + i) / 8;
                const double inv_n = 1.0 / ((double)p.H * p.W * (SC / 8));
                gn_add(&gnred[g * 2], gn_fix(a, inv_n)); gn_add(&gnred[g * 2 + 1], gn_fix(q, inv_n));
            }
        }
        __syncthreads();
        if (tid < 16) {
            const int slot = (blockIdx.x + blockIdx.y * gridDim.x) % GN_SLOTS;
            gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + slot) * 2 + (tid & 1), gnred[tid]);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Ping-pong form (round 2).  The strip walker above runs its phases back to back - request rows, 72 MFMAs per wave,
// 16-32 stores per wave, convert the next rows - and with ONE workgroup per CU (the resident weights fill the LDS) nothing
// overlaps them: phase counters put the matrix pipe at ~2.6k of ~12k cycles per 8-row tile.  Here the eight waves form TWO
// groups of four (one wave per SIMD each) that walk two different strip segments, four rows per iteration, half a period
// apart: while group A runs the MFMAs of its tile, group B stores its finished tile, converts the rows it requested one
// phase earlier into its own LDS slots and vice versa; a workgroup barrier separates the phases.  Both groups share the
// resident weights.  Same LDS budget (2 groups x 2 slots x 4 rows == 2 slots x 8 rows).
#include "conv_res2.h"
constexpr int PR = 4;                                  // image rows per group iteration == waves per group
constexpr int PGT = 256;                               // threads per group
constexpr int PITEMS = PR * SPW * (SC / 8);            // 8-channel items of a 4-row chunk (1088)
constexpr int PNI = (PITEMS + PGT - 1) / PGT;          // per thread (5; the last only for gtid < 64)

struct PChunk {
    f32x4 a[PNI], c[PNI];       // 8 channels of one pixel (bf16 input: a holds the 8 raw values, c is unused)
    float mk[PNI];
};

template <bool PRO, int TAIL, bool XB>       // TAIL: 0 none, 1 fused ResnetBlock tail with a stored shortcut (pro_res), 2 shortcut recomputed (res2_*)
__global__ __launch_bounds__(SNT) void conv3x3_pp64_kernel(const Conv3P p, const int seg_tiles, const int nseg_y) {
    constexpr bool PRO2 = TAIL != 0;
    constexpr bool r2 = TAIL == 2;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* wts = smem + 2 * STR * SPW * SLDP;            // [9 taps][64 cout][SLDP]
    __shared__ float smean[2][8], srstd[2][8];
    __shared__ long long gnred[2][16];
    __shared__ __attribute__((aligned(16))) float coef[2][3][SC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int g = wave >> 2, gw = wave & 3, gtid = tid & (PGT - 1);
    u16* patch = smem + g * (2 * PR * SPW * SLDP);     // this group's [2 slots][4 rows][34][SLDP]
    // strip segment of this group: index = (b * nseg_y + y) * nstrip + x
    const int nstrip = (p.W + 31) / 32;
    const long nseg = (long)nstrip * nseg_y * p.B;
    const long sraw = (long)blockIdx.x * 2 + g;
    const bool live = sraw < nseg;
    const long sidx = live ? sraw : nseg - 1;
    const int sx = (int)(sidx % nstrip), sy = (int)((sidx / nstrip) % nseg_y), b = (int)(sidx / ((long)nstrip * nseg_y));
    const int w0 = sx * 32;
    const int ntile_total = (p.H + PR - 1) / PR;
    const int t0 = sy * seg_tiles;
    const int nt = live ? min(seg_tiles, ntile_total - t0) : 0;
    const int hs = t0 * PR;
    const int step = p.step;
    const float* X = p.X + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.H * p.W * p.ldx + p.x_coff;      // XB: bf16 input
    const float* R = (PRO2 && !r2) ? p.pro_res + (long)b * p.H * p.W * SC : nullptr;
    const float* Pmu = r2 ? p.res2_mu + (long)b * p.H * p.W : nullptr;
    const float* Px = r2 ? p.res2_x + (long)b * p.H * p.W : nullptr;
    const float* Pspk = (r2 && p.res2_planes == 3) ? p.res2_spk + (long)b * p.H : nullptr;
    // coefficient table of the recomputed shortcut: in the 16 unused pad bytes of the weight rows (row q*16 + c/4 holds
    // channels c..c+3 of table row q) - the static LDS budget has no 1 KB left
    auto r2_put = [&](int q, int c) -> float* { return reinterpret_cast<float*>(wts + (q * 16 + (c >> 2)) * SLDP + SC) + (c & 3); };
    auto r2_at = [&](int q, int c) -> float4 { return *reinterpret_cast<const float4*>(wts + (q * 16 + (c >> 2)) * SLDP + SC); };
    const float r2_cin = r2 ? p.res2_scal[(long)step * p.res2_scal_stride + 2] : 0.f;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const int c8 = (tid & 7) * 8;

    if constexpr (r2) res2_fill(p, tid, r2_put);
    float gn_ga = 0.f, gn_be = 0.f, gn_ta = 0.f;
    uint4 gn_raw = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (PRO) {
        if (gtid < 8 * GN_SLOTS)
            gn_raw = *reinterpret_cast<const uint4*>(p.pro_stats + (((long)b * 8 + gtid / GN_SLOTS) * GN_SLOTS + (gtid % GN_SLOTS)) * 2);
        if (gtid < SC) { gn_ga = p.pro_gamma[gtid]; gn_be = p.pro_beta[gtid]; gn_ta = p.pro_tadd ? p.pro_tadd[(long)step * SC + gtid] : 0.f; }
    }
    u32x4 wr[9];
    {
        const u16* Wg = reinterpret_cast<const u16*>(p.Wbf);          // [64][9*64]
#pragma unroll
        for (int q = 0; q < 9; ++q) wr[q] = *reinterpret_cast<const u32x4*>(Wg + (long)(tid >> 3) * (9 * SC) + q * SC + c8);
    }

    // chunk c = image rows hs - 1 + 4c .. + 3, slot c & 1 of this group.  Item q of a thread is a fixed (row j, column) of
    // every chunk: its column, column mask and element offset are computed once per strip segment (the first version redid
    // two divisions and a 64-bit multiply-add chain per item and chunk: ~1k VALU cycles per tile in the load-issue phase).
    int q_row[PNI], q_eo[PNI];
    float q_mk[PNI];
    bool q_wr[PNI];
#pragma unroll
    for (int q = 0; q < PNI; ++q) {
        const int pxi = min((gtid >> 3) + (PGT / 8) * q, PR * SPW - 1);
        const int j = pxi / SPW, pw = pxi - j * SPW;
        const int wi = w0 + pw - 1;
        const bool colok = (unsigned)wi < (unsigned)p.W;
        q_row[q] = colok ? j : -0x10000;                                 // a column outside the image fails every row test
        q_eo[q] = (colok ? wi : 0);
        q_mk[q] = mrow[(colok ? wi : 0) * p.mask_ws];
        q_wr[q] = (gtid >> 3) + (PGT / 8) * q < PR * SPW && pw >= 1 && pw <= 32 && colok;
    }
    auto load_chunk = [&](int c, PChunk& x, PChunk& r) __attribute__((always_inline)) {
        const int hb = hs - 1 + PR * c;
#pragma unroll
        for (int q = 0; q < PNI; ++q) {
            const int hi = hb + q_row[q];
            const bool inb = (unsigned)hi < (unsigned)p.H;
            const int pix = inb ? hi * p.W + q_eo[q] : 0;
            x.mk[q] = inb ? q_mk[q] : 0.f;
            if constexpr (XB) {
                x.a[q] = *reinterpret_cast<const f32x4*>(Xh + (long)pix * p.ldx + c8);
            } else {
                const float* src = X + (long)pix * p.ldx + c8;
                x.a[q] = *reinterpret_cast<const f32x4*>(src);
                x.c[q] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            if constexpr (PRO2) {
                if constexpr (r2) {
                    r.a[q][0] = Pmu[pix]; r.a[q][1] = Px[pix];
                    r.a[q][2] = Pspk ? Pspk[inb ? hi : 0] : 0.f;
                } else {
                    const float* rs = R + pix * SC + c8;
                    r.a[q] = *reinterpret_cast<const f32x4*>(rs);
                    r.c[q] = *reinterpret_cast<const f32x4*>(rs + 4);
                }
            }
        }
    };
    auto store_chunk = [&](int c, const PChunk& x, const PChunk& r) __attribute__((always_inline)) {
        u16* dst = patch + (c & 1) * PR * SPW * SLDP;
        const int row_lo = hs, row_hi = min(hs + PR * nt, p.H);
        float4 t2[4][2];                     // recomputed shortcut: this thread's table entries, read once per chunk
        if constexpr (r2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { t2[k][0] = r2_at(k, c8); t2[k][1] = r2_at(k, c8 + 4); }
        }
        auto t2_at = [&](int k, int c) -> float4 { return t2[k][(c >> 2) & 1]; };
        // this thread's eight channels are the same for every item: their GroupNorm coefficients are read ONCE per chunk (inside the
        // item loop the compiler has to re-read them behind every patch store - 30 LDS reads per thread and chunk on the resource the
        // matrix role of the other group is bound by)
        constexpr bool HOIST = XB || TAIL != 2;        // (the fp32-input form with the recomputed shortcut has no registers left for it)
        f32x2 sc[4], sh[4], ta[4];
        if constexpr (PRO && HOIST) {
            *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(&coef[g][0][c8]); *reinterpret_cast<float4*>(sc + 2) = *reinterpret_cast<const float4*>(&coef[g][0][c8 + 4]);
            *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(&coef[g][1][c8]); *reinterpret_cast<float4*>(sh + 2) = *reinterpret_cast<const float4*>(&coef[g][1][c8 + 4]);
            *reinterpret_cast<float4*>(ta) = *reinterpret_cast<const float4*>(&coef[g][2][c8]); *reinterpret_cast<float4*>(ta + 2) = *reinterpret_cast<const float4*>(&coef[g][2][c8 + 4]);
        }
#pragma unroll
        for (int q = 0; q < PNI; ++q) {
            const int pxi = (gtid >> 3) + (PGT / 8) * q;
            const float mk = x.mk[q];
            float v[8];
            if constexpr (XB) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned u = __float_as_uint(x.a[q][k]);
                    v[2 * k] = lp_lo(u); v[2 * k + 1] = lp_hi(u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = x.a[q][k]; v[4 + k] = x.c[q][k]; }
            }
            uint4 o;
            if constexpr (PRO) {
                f32x2 w[4];
                if constexpr (!HOIST) {
                    *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(&coef[g][0][c8]); *reinterpret_cast<float4*>(sc + 2) = *reinterpret_cast<const float4*>(&coef[g][0][c8 + 4]);
                    *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(&coef[g][1][c8]); *reinterpret_cast<float4*>(sh + 2) = *reinterpret_cast<const float4*>(&coef[g][1][c8 + 4]);
                    *reinterpret_cast<float4*>(ta) = *reinterpret_cast<const float4*>(&coef[g][2][c8]); *reinterpret_cast<float4*>(ta + 2) = *reinterpret_cast<const float4*>(&coef[g][2][c8 + 4]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2 in; in.x = v[2 * k]; in.y = v[2 * k + 1];
                    w[k] = mish2_add(pk_fma_pinned(in, sc[k], sh[k]), ta[k]);
                }
                if constexpr (PRO2) {
                    f32x2 rr[4];
                    if constexpr (r2) {
                        float e8[8];
                        res2_eval(t2_at, c8, p.res2_planes, r2_cin, r.a[q][0], r.a[q][1], r.a[q][2], mk, e8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) rr[k] = f32x2{e8[2 * k], e8[2 * k + 1]};
                    } else {
                        rr[0] = f32x2{r.a[q][0], r.a[q][1]}; rr[1] = f32x2{r.a[q][2], r.a[q][3]};
                        rr[2] = f32x2{r.c[q][0], r.c[q][1]}; rr[3] = f32x2{r.c[q][2], r.c[q][3]};
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = w[k] * mk + rr[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] = w[k].x; v[2 * k + 1] = w[k].y; }
                    const int hi = hs - 1 + PR * c + q_row[q];
                    if (q_wr[q] && hi >= row_lo && hi < row_hi) {
                        const long eo = ((long)b * p.H * p.W + hi * p.W + q_eo[q]) * SC + c8;
                        if (p.xout_lp) {       // (uniform) 16-bit x for its one reader, the attention's context pass
                            *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.pro_xout) + eo) =
                                make_uint4(pack2_lp(v[0], v[1]), pack2_lp(v[2], v[3]), pack2_lp(v[4], v[5]), pack2_lp(v[6], v[7]));
                        } else {
                            float* xo = p.pro_xout + eo;
                            *reinterpret_cast<float4*>(xo) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(xo + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = w[k] * mk;
                o.x = pack2_lp(w[0].x, w[0].y); o.y = pack2_lp(w[1].x, w[1].y);
                o.z = pack2_lp(w[2].x, w[2].y); o.w = pack2_lp(w[3].x, w[3].y);
            } else {
                o.x = pack2_mul_lp_pinned(v[0], v[1], mk); o.y = pack2_mul_lp_pinned(v[2], v[3], mk);
                o.z = pack2_mul_lp_pinned(v[4], v[5], mk); o.w = pack2_mul_lp_pinned(v[6], v[7], mk);
            }
            if (pxi < PR * SPW) *reinterpret_cast<uint4*>(dst + pxi * SLDP + c8) = o;
        }
    };

    PChunk cx, cr, cx1;
    if (nt > 0) {
        load_chunk(0, cx, cr);
        if constexpr (!PRO2) load_chunk(1, cx1, cr);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PRO) {
        if (gtid < 8 * GN_SLOTS) {
            long long s1 = (long long)(((unsigned long long)gn_raw.y << 32) | gn_raw.x);
            long long s2 = (long long)(((unsigned long long)gn_raw.w << 32) | gn_raw.z);
            gn_slots_reduce<GN_SLOTS>(s1, s2);
            if ((gtid % GN_SLOTS) == 0) gn_moments(s1, s2, 1e-5, smean[g][gtid / GN_SLOTS], srstd[g][gtid / GN_SLOTS]);
        }
        lds_barrier();
        if (gtid < SC) {
            const float mean = smean[g][gtid / 8], rstd = srstd[g][gtid / 8];
            coef[g][0][gtid] = rstd * gn_ga;
            coef[g][1][gtid] = gn_be - mean * rstd * gn_ga;
            coef[g][2][gtid] = gn_ta;
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) *reinterpret_cast<u32x4*>(wts + (q * SC + (tid >> 3)) * SLDP + c8) = wr[q];
    lds_barrier();
    if (nt > 0) {
        store_chunk(0, cx, cr);
        if constexpr (PRO2) { load_chunk(1, cx, cr); store_chunk(1, cx, cr); }
        else store_chunk(1, cx1, cr);
    }
    lds_barrier();

    const float bias0 = p.bias[i], bias1 = p.bias[32 + i];
    float gs[2] = {0.f, 0.f}, gss[2] = {0.f, 0.f};
    const bool yb = p.y_bf16 != 0;

    auto emit_tile = [&](const f32x16 (&acc)[2], int t) __attribute__((always_inline)) {
        const int ho = hs + PR * t + gw;
        const long off = (long)b * p.H * p.W * SC + ((long)ho * p.W + w0 + 4 * hh) * SC + i;
        float* yf = p.Y + off; u16* yh = reinterpret_cast<u16*>(p.Y) + off;
        const bool full = ho < p.H && w0 + 32 <= p.W;
        if (full && yb) {
            const bool odd = (lane & 1) != 0;
            u16* yp = yh + (odd ? 16 * SC - 1 : 0);
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const float bias = n2 ? bias1 : bias0;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[n2][r] + bias; gs[n2] += v; gss[n2] = fmaf(v, v, gss[n2]); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float lo_r = acc[n2][j] + bias, hi_r = acc[n2][8 + j] + bias;
                    const float recv = lane_xor1(odd ? lo_r : hi_r);
                    const float mine = odd ? hi_r : lo_r;
                    const unsigned pk = odd ? pack2_lp(recv, mine) : pack2_lp(mine, recv);
                    *reinterpret_cast<unsigned*>(yp + ((j & 3) + 8 * (j >> 2)) * SC + n2 * 32) = pk;
                }
            }
            return;
        }
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const float bias = n2 ? bias1 : bias0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = full || (ho < p.H && w0 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.W);
                const float v = acc[n2][r] + bias;
                const float vs = ok ? v : 0.f;
                gs[n2] += vs; gss[n2] = fmaf(vs, vs, gss[n2]);
                if (ok) {
                    if (yb) yh[((r & 3) + 8 * (r >> 2)) * SC + n2 * 32] = lp_bits(v);
                    else yf[((r & 3) + 8 * (r >> 2)) * SC + n2 * 32] = v;
                }
            }
        }
    };
    auto mfma_tile = [&](int t, f32x16 (&cur)[2]) __attribute__((always_inline)) {
        if (t + 1 < nt) load_chunk(t + 2, cx, cr);            // rows of the next tile: in flight under the MFMAs, converted in the next phase
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) { cur[0][r] = 0.f; cur[1][r] = 0.f; }
        const u16* arow[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int r = gw + kh;                                            // row of the 6-row window
            arow[kh] = patch + ((((t + (r >> 2)) & 1) * PR + (r & 3)) * SPW + i) * SLDP + hh * 8;
        }
        const u16* brow = wts + i * SLDP + hh * 8;
        lp8 af[2], b0[2], b1[2];
        af[0] = *reinterpret_cast<const lp8*>(arow[0]);
        b0[0] = *reinterpret_cast<const lp8*>(brow);
        b1[0] = *reinterpret_cast<const lp8*>(brow + 32 * SLDP);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 1 < 36) {
                const int tap = (s + 1) >> 2, ks = (s + 1) & 3, kh = tap / 3, kw = tap - kh * 3;
                af[(s + 1) & 1] = *reinterpret_cast<const lp8*>(arow[kh] + kw * SLDP + ks * 16);
                b0[(s + 1) & 1] = *reinterpret_cast<const lp8*>(brow + tap * SC * SLDP + ks * 16);
                b1[(s + 1) & 1] = *reinterpret_cast<const lp8*>(brow + (tap * SC + 32) * SLDP + ks * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            cur[0] = DEX_MFMA_LP(af[s & 1], b0[s & 1], cur[0], 0, 0, 0);
            cur[1] = DEX_MFMA_LP(af[s & 1], b1[s & 1], cur[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // phase ph: group (ph & 1) runs the matrix work of its tile (ph - g) / 2, the other group the memory work of the tile
    // it finished one phase earlier
    f32x16 acc[2];
    const int nphase = 2 * seg_tiles + 1;
#ifdef DEX_TIMING
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
    const long long tk0 = tlast;
#define PSTAMP(k) do { const long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define PSTAMP(k) do {} while (0)
#endif
    for (int ph = 0; ph < nphase; ++ph) {
        if ((ph & 1) == g) {
            const int t = (ph - g) >> 1;
            if (t < nt) mfma_tile(t, acc);
            PSTAMP(1);
        } else {
            const int t = (ph - g - 1) >> 1;
            if (t >= 0 && t < nt) {
                emit_tile(acc, t);
                PSTAMP(2);
                if (t + 1 < nt) store_chunk(t + 2, cx, cr);
                PSTAMP(3);
            }
        }
        lds_barrier();
        PSTAMP(4);
    }
#ifdef DEX_TIMING
    if (p.dbg && gtid == 0) {
        long long* d = p.dbg + ((long)blockIdx.x * 2 + g) * 8;
        for (int k = 0; k < 7; ++k) d[k] = tk[k];
        d[7] = __builtin_readcyclecounter() - tk0;
    }
#endif

    if (p.gn_stats) {
        if (tid < 32) gnred[tid >> 4][tid & 15] = 0;
        __syncthreads();
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            float a = gs[n2], q = gss[n2];
            for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            if (hh == 0 && (i & 7) == 0) {
                const int gg = (n2 * 32 + i) / 8;
                const double inv_n = 1.0 / ((double)p.H * p.W * (SC / 8));
                gn_add(&gnred[g][gg * 2], gn_fix(a, inv_n)); gn_add(&gnred[g][gg * 2 + 1], gn_fix(q, inv_n));
            }
        }
        __syncthreads();
        if (gtid < 16 && live) {
            const int slot = (int)(sidx % GN_SLOTS);
            gn_add(p.gn_stats + (((long)b * 8 + (gtid >> 1)) * GN_SLOTS + slot) * 2 + (gtid & 1), gnred[g][gtid]);
        }
    }
}

}  // namespace

template <bool PRO, bool PRO2, bool XB>
static void stream_go(const Conv3P& p, int tiles_per_wg, dim3 grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream64_kernel<PRO, PRO2, XB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_LDS);
        attr = true;
    }
    hipLaunchKernelGGL((conv3x3_stream64_kernel<PRO, PRO2, XB>), grid, dim3(SNT), S_LDS, st, p, tiles_per_wg);
}

// Largest strip segment (iterations per workgroup) that still fills the chip once; 0 = use the tile kernel.
static bool pp_plan(const Conv3P& p, int& seg, int& nseg_y);
// Will a 64 -> 64 fused-tail convolution of this grid run on the ping-pong form?  Only that form implements Conv3P::res2_*
// (the caller then skips storing the first ResnetBlock's shortcut).
bool conv3x3_res2_form(int H, int W, int B) {
#if defined(DEX_LP_WSPLIT)
    return false;            // no split-weight form (lp_config.h)
#endif
   
    Conv3P q{}; q.H = H; q.W = W; q.B = B; q.Cin = SC; q.Cout = SC; q.ldx = SC; q.x_bf16 = 1;
    q.pro_stats = reinterpret_cast<const gnfix_t*>(&q);      // (non-null: the prologue forms)
    int a, b2;
    return conv3x3_stream_tiles(q) != 0 && pp_plan(q, a, b2);
}
int conv3x3_stream_tiles(const Conv3P& p) {
#if defined(DEX_LP_WSPLIT)
    return 0;                // no split-weight form (lp_config.h)
#endif
   
    if (p.Cin != SC || p.Cout != SC || p.res_w || p.ldx % 8 != 0 || p.x_coff % 8 != 0) return 0;
    if (p.x_bf16 && !p.pro_stats) { int a, b2; if (!pp_plan(p, a, b2)) return 0; }       // plain 16-bit input: the ping-pong form only
    const int mode = knob_or("DEX_CONV_STREAM", 1);      // 0: never, 2: whenever the shape allows (tests), default: by grid size
    if (mode == 0) return 0;
    const int tiles = (p.H + STR - 1) / STR;
    const long strips = (long)((p.W + 31) / 32) * p.B;
    for (int tpw = tiles < 5 ? tiles : 5; tpw >= 2; --tpw)
        if (strips * ((tiles + tpw - 1) / tpw) >= 256) return tpw;
    // (one 8-row tile per workgroup at batch 1 was measured too: 16.5 us, the same as the tile kernel - not kept)
    return mode == 2 ? (tiles < 5 ? tiles : 5) : 0;
}

template <bool PRO, int TAIL, bool XB>
static void pp_go(const Conv3P& p, int seg_tiles, int nseg_y, unsigned nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp64_kernel<PRO, TAIL, XB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_LDS);
        attr = true;
    }
    hipLaunchKernelGGL((conv3x3_pp64_kernel<PRO, TAIL, XB>), dim3(nwg), dim3(SNT), S_LDS, st, p, seg_tiles, nseg_y);
}

// Ping-pong plan: strip segments of `seg` 4-row tiles, two segments per workgroup; wanted: at least one workgroup per CU.
static bool pp_plan(const Conv3P& p, int& seg, int& nseg_y) {
    const int mode = knob_or("DEX_CONV_PP", 1);          // 0: never, 2: whenever the shape allows (tests), default: by grid size
    if (mode == 0) return false;
    const int tiles = (p.H + PR - 1) / PR;
    const long cols = (long)((p.W + 31) / 32) * p.B;
    // Round 5: the split of a strip into segments by ROUNDS of the chip (one workgroup of two segments per CU).  The first rule cut
    // strips until there were >= 512 segments; the long form (T = 4000: 125 strips of 20 tiles) got 5 x 4-tile segments = 313
    // workgroups - two rounds, the second 22 % full, the 72 KB weight load amortised over four tiles - where 4 x 5-tile segments are
    // 250 workgroups in ONE round: cost = rounds x (tiles per segment + ~2 tiles' worth of prologue), among splits that keep >= 95 %
    // of the CUs busy and >= 3 tiles per segment.  The batch shapes (DEX / GeDEX B = 32: 256 workgroups) choose as before.
    const int ncu = device_cus();
    long best = -1;
    for (int k = 1; k == 1 || tiles / k >= 3; ++k) {
        const int sg = (tiles + k - 1) / k, ny = (tiles + sg - 1) / sg;
        const long wgs = (cols * ny + 1) / 2;
        if (wgs * 20 < (long)ncu * 19) continue;          // (a 3/4 bound also took DEX B = 32's half-resolution convs from the 8-row strip walker: 35 vs 31 us)
        const long cost = (wgs + ncu - 1) / ncu * (sg + 2);
        if (best < 0 || cost < best) { best = cost; seg = sg; nseg_y = ny; }
    }
    if (best >= 0) return true;
    int k = 1;                                     // (mode 2, tests: the first rule's split on grids below a round)
    while (cols * k < 512 && tiles / (k + 1) >= 3) ++k;
    seg = (tiles + k - 1) / k;
    nseg_y = (tiles + seg - 1) / seg;
    return mode == 2;
}

void launch_conv3x3_stream(const Conv3P& p, int tiles_per_wg, hipStream_t st) {
    int seg = 0, nsy = 0;
    if (pp_plan(p, seg, nsy)) {
        const long nseg = (long)((p.W + 31) / 32) * nsy * p.B;
        const unsigned nwg = (unsigned)((nseg + 1) / 2);
        const bool pro_ = p.pro_stats != nullptr;
        const bool tail_ = p.pro_res || p.res2_w;
        g_last_symbol = p.res2_w ? (p.x_bf16 ? "conv3x3_pp64_kernel<1,2,1>" : "conv3x3_pp64_kernel<1,2,0>")
                      : tail_ ? (p.x_bf16 ? "conv3x3_pp64_kernel<1,1,1>" : "conv3x3_pp64_kernel<1,1,0>")
                      : pro_ ? (p.x_bf16 ? "conv3x3_pp64_kernel<1,0,1>" : "conv3x3_pp64_kernel<1,0,0>")
                      : (p.x_bf16 ? "conv3x3_pp64_kernel<0,0,1>" : "conv3x3_pp64_kernel<0,0,0>");
        if (p.res2_w) { p.x_bf16 ? pp_go<true, 2, true>(p, seg, nsy, nwg, st) : pp_go<true, 2, false>(p, seg, nsy, nwg, st); }
        else if (tail_) { p.x_bf16 ? pp_go<true, 1, true>(p, seg, nsy, nwg, st) : pp_go<true, 1, false>(p, seg, nsy, nwg, st); }
        else if (pro_) { p.x_bf16 ? pp_go<true, 0, true>(p, seg, nsy, nwg, st) : pp_go<true, 0, false>(p, seg, nsy, nwg, st); }
        else if (p.x_bf16) pp_go<false, 0, true>(p, seg, nsy, nwg, st);
        else pp_go<false, 0, false>(p, seg, nsy, nwg, st);
        return;
    }
    const int tiles = (p.H + STR - 1) / STR;
    dim3 grid((p.W + 31) / 32, (tiles + tiles_per_wg - 1) / tiles_per_wg, p.B);
    const bool pro = p.pro_stats != nullptr;
    g_last_symbol = p.pro_res ? (p.x_bf16 ? "conv3x3_stream64_kernel<1,1,1>" : "conv3x3_stream64_kernel<1,1,0>")
                  : pro ? (p.x_bf16 ? "conv3x3_stream64_kernel<1,0,1>" : "conv3x3_stream64_kernel<1,0,0>") : "conv3x3_stream64_kernel<0,0,0>";
    if (p.pro_res) { p.x_bf16 ? stream_go<true, true, true>(p, tiles_per_wg, grid, st) : stream_go<true, true, false>(p, tiles_per_wg, grid, st); }
    else if (pro) { p.x_bf16 ? stream_go<true, false, true>(p, tiles_per_wg, grid, st) : stream_go<true, false, false>(p, tiles_per_wg, grid, st); }
    else stream_go<false, false, false>(p, tiles_per_wg, grid, st);
}

}  // namespace DEX_LP_NS
}  // namespace dex
