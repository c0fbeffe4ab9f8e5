// conv3x3_bf16.hip — 3x3 / stride 1 / pad 1 convolution of the U-Net Block (diffusion.py:41-50) as a
// patch-staged direct convolution on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 tensors in HBM).
//
// A workgroup owns TH rows x 32 columns of output pixels.  The (TH+2) x 34 input patch is staged ONCE in LDS
// as bf16 (1.6x halo overhead instead of the 9x re-gather of an im2col GEMM), with the producer's tail fused
// into the staging:  x*mask   or   mask*(Mish(GroupNorm(x)) + time_bias)   (diffusion.py:49,67-69), so the
// activated tensor of block1 never exists in HBM.  Two weight schedules:
//   RESIDENT  (few workgroups, B=1): a workgroup owns a slice of NSL output channels and keeps ALL nine taps of
//             its weights in LDS; every global load of the kernel is issued up front and the 9-tap MFMA chain
//             runs without a single barrier — at one workgroup per CU dependent global round trips are the
//             only cost that matters (measured: 9 streamed taps = 9 exposed round trips).
//   STREAMED  (many workgroups, big batches): per-tap weight slices [Cout][CC] stream through a double-buffered
//             LDS tile (small LDS footprint, 3 workgroups per CU).
// LDS rows are padded by 16 B so every ds_read_b128 / ds_write_b128 lane group covers 64 distinct banks.
// Epilogue: +bias, GroupNorm partial statistics (workgroup-combined, one fp32 atomic pair per group into a
// slot-spread buffer), fp32 channels-last store (128-B rows).
#include "kernels.h"
#include <cstdlib>
#include <cstdio>
#include "lp_util.h"
#include "kernels_lp.h"
#include "conv_gn.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: register rings of HIP's uint4 struct defeat SROA

__device__ __forceinline__ float cv_mish(float x) {           // branch-free: tanh(softplus(x)) == 1 to fp32 for x > 20
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}

// Epilogue of one wave: NT 32x32 tiles; rows = pixels (w0 + row) of image row ho, cols = channels nbase + t*32 + i.
// GroupNorm partials are combined across the workgroup's waves in LDS (gnred[8][2]) before the atomics.
template <int NT, int COUT>
__device__ __forceinline__ void cv_epilogue(const Conv3P& p, f32x16 (&acc)[NT], int b, int ho, int w0, int nbase, int lane, int tid,
                                            long long* gnred) {
    const int i = lane & 31, hh = lane >> 5;
    constexpr int cpg = COUT / 8;
    if (p.gn_stats) { if (tid < 16) gnred[tid] = 0; __syncthreads(); }
    // one 64-bit base per lane; the 16 rows of a tile are compile-time offsets from it (the first version rebuilt a
    // 64-bit address and a bounds predicate per element).  Full tiles (the common case: W % 32 == 0) store unpredicated.
    const bool full = ho < p.H && w0 + 32 <= p.W;
    const long ybase = (long)b * p.H * p.W * COUT + ((long)ho * p.W + w0 + 4 * hh) * COUT + nbase + i;
    float* yl = p.Y + ybase;
    u16* yh = reinterpret_cast<u16*>(p.Y) + ybase;            // y_bf16: same element offsets, half the bytes
    const bool yb = p.y_bf16 != 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = nbase + t * 32 + i;
        const float bias = p.bias[n];
        float gs = 0.f, gss = 0.f;
        if (full && yb) {
            // bf16 output: a lane owns ONE channel of 16 pixel rows, i.e. 2-byte stores.  Neighbouring lanes swap half of
            // their rows (DPP), so each lane stores channel PAIRS of 8 rows: 8 dword stores instead of 16 short stores
            // (the epilogue is store-issue bound: 64 addresses per instruction whatever their width).
            const bool odd = (lane & 1) != 0;
            u16* yp = yh + (odd ? 16 * COUT - 1 : 0) + t * 32;        // odd lanes: rows 16..27 of the tile, pair starts one channel left
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[t][r] + bias; gs += v; gss = fmaf(v, v, gss); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo_r = acc[t][j] + bias, hi_r = acc[t][8 + j] + bias;       // rows j and 8+j of this lane's channel
                const float recv = lane_xor1(odd ? lo_r : hi_r);                          // the partner's value for MY row set
                const float mine = odd ? hi_r : lo_r;
                const unsigned pk = odd ? pack2_lp(recv, mine) : pack2_lp(mine, recv);
                *reinterpret_cast<unsigned*>(yp + ((j & 3) + 8 * (j >> 2)) * COUT) = pk;
            }
        } else if (full) {          // fp32 output: the same swap, 8-byte stores
            const bool odd = (lane & 1) != 0;
            float* yp = yl + (odd ? 16 * COUT - 1 : 0) + t * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[t][r] + bias; gs += v; gss = fmaf(v, v, gss); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo_r = acc[t][j] + bias, hi_r = acc[t][8 + j] + bias;
                const float recv = lane_xor1(odd ? lo_r : hi_r);
                const float mine = odd ? hi_r : lo_r;
                *reinterpret_cast<float2*>(yp + ((j & 3) + 8 * (j >> 2)) * COUT) = odd ? make_float2(recv, mine) : make_float2(mine, recv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wo = w0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool ok = ho < p.H && wo < p.W;
                const float v = acc[t][r] + bias;
                const float vs = ok ? v : 0.f;
                gs += vs; gss = fmaf(vs, vs, gss);
                if (ok) {
                    if (yb) yh[((r & 3) + 8 * (r >> 2)) * COUT + t * 32] = lp_bits(v);
                    else yl[((r & 3) + 8 * (r >> 2)) * COUT + t * 32] = v;
                }
            }
        }
        if (p.gn_stats) {
            for (int o = 1; o < cpg; o <<= 1) { gs += __shfl_xor(gs, o); gss += __shfl_xor(gss, o); }
            gs += __shfl_xor(gs, 32); gss += __shfl_xor(gss, 32);
            if (hh == 0 && (i & (cpg - 1)) == 0) {      // a wave's sums come out of a fixed shuffle order; the integer adds commute
                const double inv_n = 1.0 / ((double)p.H * p.W * cpg);
                gn_add(&gnred[(n / cpg) * 2], gn_fix(gs, inv_n)); gn_add(&gnred[(n / cpg) * 2 + 1], gn_fix(gss, inv_n));
            }
        }
    }
    if (p.gn_stats) {
        __syncthreads();
        if (tid < 16) {
            const long long v = gnred[tid];
            if (v != 0) {
                const int slot = (blockIdx.x + blockIdx.y * gridDim.x) % GN_SLOTS;
                gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + slot) * 2 + (tid & 1), v);
            }
        }
    }
}

// the fused 1x1 shortcut's output tile of one wave: + bias, no statistics, same tile addressing as the main output
template <int NT, int COUT>
__device__ __forceinline__ void cv_res_store(const Conv3P& p, const f32x16 (&accr)[NT], int b, int ho, int w0, int nb, int lane) {
    const int i = lane & 31, hh = lane >> 5;
    float* yl = p.res_y + ((long)b * p.H * p.W + (long)ho * p.W + w0 + 4 * hh) * COUT + nb + i;
    const bool rfull = ho < p.H && w0 + 32 <= p.W;
    const bool odd = (lane & 1) != 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bias = p.res_b[nb + t * 32 + i];
        if (rfull) {           // channel pairs of 8 rows per lane (DPP swap with the neighbouring lane): 8-byte stores
            float* yp = yl + (odd ? 16 * COUT - 1 : 0) + t * 32;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo_r = accr[t][j] + bias, hi_r = accr[t][8 + j] + bias;
                const float recv = lane_xor1(odd ? lo_r : hi_r);
                const float mine = odd ? hi_r : lo_r;
                *reinterpret_cast<float2*>(yp + ((j & 3) + 8 * (j >> 2)) * COUT) = odd ? make_float2(recv, mine) : make_float2(mine, recv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wo = w0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ho < p.H && wo < p.W) yl[((r & 3) + 8 * (r >> 2)) * COUT + t * 32] = accr[t][r] + bias;
            }
        }
    }
}

// CC channels per chunk, output-channel slice [slice*NSL, +NSL) of COUT, TH rows (waves: TH x (4/TH)).
// grid.z = b * (COUT/NSL) + slice.
template <int CC, int COUT, int NSL, int TH, bool PRO2 = false, bool RES = false, bool XB = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv3x3_lp_kernel(const Conv3P p) {
    constexpr int PW = 34, PH = TH + 2;
    constexpr int LDP = CC + 8;
    constexpr int WN = NW / TH;                          // NW = 8: two waves per SIMD in ONE workgroup (the 128-channel layers fill the LDS with one)
    constexpr int NTHR = 64 * NW;
    constexpr int NT = NSL / 32 / WN;
    constexpr int NSLICE = COUT / NSL;
    constexpr int ITEMS = PH * PW * (CC / 8);            // 8-channel patch items
    constexpr int NI = (ITEMS + NTHR - 1) / NTHR;        // per thread
    constexpr int WPT = NSL * CC / 8 / NTHR;             // weight items (16 B) per thread per tap
    constexpr int RING = CC == 128 ? (PRO2 ? 1 : 2) : 3;  // taps of weights in flight ahead of the MFMAs
    constexpr int NPASS = CC == 128 ? (TH == 8 ? (PRO2 ? 8 : 4) : (PRO2 ? 5 : 2)) : (PRO2 ? 3 : 1);   // patch staging passes: 128-channel chunks (or two source tensors) would need >256 registers in one
    constexpr int NIP = (NI + NPASS - 1) / NPASS;        // (1 workgroup per CU instead of 2: measured 16 -> 22 us at 40x256)
#ifdef DEX_LP_WSPLIT
    // split weights: the nine taps run twice over the same patch - taps 9..17 are the lo halves of the weights, p.w_lo_off elements
    // behind their hi halves (same [COUT][9*Cin] layout); the 1x1 shortcut's lo half replaces its hi half in rbuf after the centre tap.
    // Two other routes for the lo halves were built and measured SLOWER than this second pass through the LDS ring: per-wave register
    // rings fed from the row-major twin (a lane's 16 bytes of a weight row = 64 separate lines per load instruction: B = 1 19.3 -> 16.5 k
    // frames/s, B = 32 50.8 -> 36.8 k) and from a fragment-order twin (coalesced, but every wave of a workgroup fetches the same
    // fragments from L2 again, where the LDS ring loads them once per workgroup: 19.3 -> 18.1 k, 54.5 -> 45.9 k).  A third: hi and lo rows
    // of a tap in ONE LDS tile for the 64-channel chunks (nine phases, each A fragment read once for both MFMAs): 19.3 -> 18.8 k,
    // 54.5 -> 51.4 k - the doubled weight tile costs a resident workgroup per CU.
    constexpr int NTAP = 18;
#else
    constexpr int NTAP = 9;
#endif
    static_assert(NT >= 1 && WPT >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* patch = smem;                                // [PH*PW][LDP]
    u16* wbuf = smem + PH * PW * LDP;                 // [2][NSL][LDP]
    u16* rbuf = wbuf + 2 * NSL * LDP;                 // [NSL][LDP]  1x1 shortcut weights of this chunk (RES)
    __shared__ long long gnred[16];
    constexpr int COEF_N = (TH == 8) ? 128 : 256;        // the 8-row form fills the LDS to the last KB: its table covers Cin = 128 only
    __shared__ __attribute__((aligned(16))) float coef[3][COEF_N];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wrow = wave % TH, wcol = wave / TH;
#ifdef DEX_TIMING
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk0 = __builtin_readcyclecounter();
    long long tlast = tk0;
#define CSTAMP(k) do { const long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define CSTAMP(k) do {} while (0)
#endif
    // (skip_dead: the strip index is rotated by the row-tile and batch indices.  Workgroup L runs on XCD L % 8 and the grid's x extent is
    // a multiple of 8 at the batch shapes, so unrotated a strip - and with it the padding, which sits in the high strips of every short
    // utterance - always lands on the same XCD: with 79 % of the tiles skipped three XCDs did all the work and the launch got 9 % shorter)
    const int xs = p.skip_dead ? (int)((blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x) : (int)blockIdx.x;
    const int w0 = xs * 32, h0 = blockIdx.y * TH;
    const int b = blockIdx.z / NSLICE, slice = blockIdx.z % NSLICE;
    const int step = p.step;
    const float* X = p.X + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.H * p.W * p.ldx + p.x_coff;   // XB: the input is bf16
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const int K = 9 * p.Cin;
    const u16* Wg = reinterpret_cast<const u16*>(p.Wbf) + (long)slice * NSL * K;     // [COUT][9*Cin], rows of this slice
    const bool pro = p.pro_stats != nullptr;
    // Round 6: DEAD tiles.  The batched jobs are ragged (an utterance shorter than the batch's T is padded, mask 0) and the reference
    // computes the padding: the conv sees x * mask = 0 there, so a tile whose 34 patch columns are all masked produces exactly
    // conv(0) + bias - zero accumulators into the ordinary epilogue (stores, statistics).  Such a tile skips its loads, the GroupNorm /
    // Mish prologue and every MFMA: bit-identical, 10 - 16 % of the tiles of the benchmarked batches.  (Not the fused-tail form: it also
    // writes x = mask * Mish(GN(h2)) + res for later consumers, which is not zero in the padding.)
    // (a branch of its own that ENDS the kernel: folded into the main path - a `dead` flag around the chunk loop - it changed the main
    // path's register allocation: 181 -> 240 VGPRs for the 64-channel form, one resident workgroup per SIMD pair less, +40 % launch time)
    if constexpr (!PRO2) {
        if (p.skip_dead) {
            const int wi = w0 - 1 + (lane < 34 ? lane : 33);
            const float mv = (unsigned)wi < (unsigned)p.W ? mrow[wi * p.mask_ws] : 0.f;
            if (__builtin_amdgcn_ballot_w64(mv != 0.f) == 0) {          // (every wave of the workgroup computes the same answer)
                f32x16 zacc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) zacc[t][r] = 0.f;
                if constexpr (RES) cv_res_store<NT, COUT>(p, zacc, b, h0 + wrow, w0, slice * NSL + wcol * NT * 32, lane);
                cv_epilogue<NT, COUT>(p, zacc, b, h0 + wrow, w0, slice * NSL + wcol * NT * 32, lane, tid, gnred);
                return;
            }
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x16 accr[RES ? NT : 1];
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accr[t][r] = 0.f;
    }
    // weight item j of this thread: output channel wn[j], 8 input channels at wc8[j] (same for every tap)
    int wofs[WPT], wlds[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int it = tid + NTHR * j;
        const int n = it / (CC / 8), c8 = (it % (CC / 8)) * 8;
        wofs[j] = n * K + c8;
        wlds[j] = n * LDP + c8;
    }
    const int pc8 = (tid % (CC / 8)) * 8;             // 256 % (CC/8) == 0: a thread's patch items share one channel chunk

    CvGnLoads gnl{};
    if (pro) gnl = cv_gn_issue(p, b, tid, step);
    const int nchunk = p.Cin / CC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cbase = ch * CC;
        // ---- every global load of the chunk's first round goes out back to back: the patch FIRST (loads return in order and
        // the conversion, the long phase, needs only the patch), then the weights of taps 0..RING, which land under it
        u32x4 w0r[WPT], wr[RING][WPT], rwr[RES ? WPT : 1];
#ifdef DEX_LP_WSPLIT
        u32x4 rwl[RES ? WPT : 1];
#define WS_TAPOFS(t) ((long)((t) % 9) * p.Cin + ((t) >= 9 ? p.w_lo_off : 0L))
#else
#define WS_TAPOFS(t) ((long)(t) * p.Cin)
#endif
        float4 sc0, sc1, sh0, sh1, t0, t1;                // this thread's 8 channels of the coefficient table (read after the barrier)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            float4 pf0[NIP], pf1[XB ? 1 : NIP], rf0[PRO2 ? NIP : 1], rf1[PRO2 ? NIP : 1];   // XB: pf0 holds the 8 raw bf16
            float pmk[NIP];
            bool pin[NIP];
#pragma unroll
            for (int q = 0; q < NIP; ++q) {
                const int it = min(tid + NTHR * (ps * NIP + q), ITEMS - 1);
                const int px = it / (CC / 8);
                const int pw = px % PW, ph = px / PW;
                const int hi = h0 + ph - 1, wi = w0 + pw - 1;
                const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const int hc = inb ? hi : 0, wc = inb ? wi : 0;          // clamped: every load is unconditional
                if constexpr (XB) {
                    pf0[q] = *reinterpret_cast<const float4*>(Xh + ((long)hc * p.W + wc) * p.ldx + cbase + pc8);
                } else {
                    const float* src = X + ((long)hc * p.W + wc) * p.ldx + cbase + pc8;
                    pf0[q] = *reinterpret_cast<const float4*>(src);
                    pf1[q] = *reinterpret_cast<const float4*>(src + 4);
                }
                if constexpr (PRO2) {
                    const float* rs = p.pro_res + ((long)b * p.H * p.W + (long)hc * p.W + wc) * p.Cin + cbase + pc8;
                    rf0[q] = *reinterpret_cast<const float4*>(rs);
                    rf1[q] = *reinterpret_cast<const float4*>(rs + 4);
                }
                const float mk = mrow[wc * p.mask_ws];
                pmk[q] = inb ? mk : 0.f;
                pin[q] = inb;
            }
            if (ps == 0) {
                if constexpr (RES) {
                    const u16* Rg = reinterpret_cast<const u16*>(p.res_w) + (long)slice * NSL * p.Cin;      // [COUT][Cin]
#pragma unroll
                    for (int j = 0; j < WPT; ++j) {
                        const int it = tid + NTHR * j;
                        rwr[j] = *reinterpret_cast<const u32x4*>(Rg + (long)(it / (CC / 8)) * p.Cin + cbase + (it % (CC / 8)) * 8);
#ifdef DEX_LP_WSPLIT
                        rwl[j] = *reinterpret_cast<const u32x4*>(Rg + p.res_lo_off + (long)(it / (CC / 8)) * p.Cin + cbase + (it % (CC / 8)) * 8);
#endif
                    }
                }
#pragma unroll
                for (int j = 0; j < WPT; ++j) w0r[j] = *reinterpret_cast<const u32x4*>(Wg + wofs[j] + cbase);
#pragma unroll
                for (int s = 0; s < RING; ++s)
#pragma unroll
                    for (int j = 0; j < WPT; ++j) wr[s][j] = *reinterpret_cast<const u32x4*>(Wg + wofs[j] + WS_TAPOFS(s + 1) + cbase);
                CSTAMP(0);
                if (pro && ch == 0) cv_gn_finish<COEF_N>(p, gnl, tid, coef);
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();                            // GN coefficients visible; previous chunk's MFMAs done with patch/wbuf
                CSTAMP(1);
                if (pro) {
                    const int c = cbase + pc8;
                    sc0 = *reinterpret_cast<const float4*>(&coef[0][c]); sc1 = *reinterpret_cast<const float4*>(&coef[0][c + 4]);
                    sh0 = *reinterpret_cast<const float4*>(&coef[1][c]); sh1 = *reinterpret_cast<const float4*>(&coef[1][c + 4]);
                    t0 = *reinterpret_cast<const float4*>(&coef[2][c]); t1 = *reinterpret_cast<const float4*>(&coef[2][c + 4]);
                }
            }
#pragma unroll
            for (int q = 0; q < NIP; ++q) {
                float4 f0, f1;
                if constexpr (XB) {
                    const float4 r = pf0[q];
                    const unsigned u0 = __float_as_uint(r.x), u1 = __float_as_uint(r.y), u2 = __float_as_uint(r.z), u3 = __float_as_uint(r.w);
                    f0 = make_float4(lp_lo(u0), lp_hi(u0), lp_lo(u1), lp_hi(u1));
                    f1 = make_float4(lp_lo(u2), lp_hi(u2), lp_lo(u3), lp_hi(u3));
                } else { f0 = pf0[q]; f1 = pf1[q]; }
#ifdef DEX_SCALAR_PROLOGUE
                if (pro) {
                    f0.x = cv_mish(fmaf(f0.x, sc0.x, sh0.x)) + t0.x; f0.y = cv_mish(fmaf(f0.y, sc0.y, sh0.y)) + t0.y;
                    f0.z = cv_mish(fmaf(f0.z, sc0.z, sh0.z)) + t0.z; f0.w = cv_mish(fmaf(f0.w, sc0.w, sh0.w)) + t0.w;
                    f1.x = cv_mish(fmaf(f1.x, sc1.x, sh1.x)) + t1.x; f1.y = cv_mish(fmaf(f1.y, sc1.y, sh1.y)) + t1.y;
                    f1.z = cv_mish(fmaf(f1.z, sc1.z, sh1.z)) + t1.z; f1.w = cv_mish(fmaf(f1.w, sc1.w, sh1.w)) + t1.w;
                }
#else
                if (pro) {       // GroupNorm-apply + Mish + time bias on fp32 pairs (packed VALU, bf16_util.h)
                    const f32x2 a0 = mish2_add(f32x2{f0.x, f0.y} * f32x2{sc0.x, sc0.y} + f32x2{sh0.x, sh0.y}, f32x2{t0.x, t0.y});
                    const f32x2 a1 = mish2_add(f32x2{f0.z, f0.w} * f32x2{sc0.z, sc0.w} + f32x2{sh0.z, sh0.w}, f32x2{t0.z, t0.w});
                    const f32x2 a2 = mish2_add(f32x2{f1.x, f1.y} * f32x2{sc1.x, sc1.y} + f32x2{sh1.x, sh1.y}, f32x2{t1.x, t1.y});
                    const f32x2 a3 = mish2_add(f32x2{f1.z, f1.w} * f32x2{sc1.z, sc1.w} + f32x2{sh1.z, sh1.w}, f32x2{t1.z, t1.w});
                    f0 = make_float4(a0.x, a0.y, a1.x, a1.y); f1 = make_float4(a2.x, a2.y, a3.x, a3.y);
                }
#endif
                const float mk = pmk[q];
                const int it = tid + NTHR * (ps * NIP + q);
                if constexpr (PRO2) {
                    // x = mask * Mish(GN(h2)) + res: the value the un-fused gn_apply kernel wrote; the conv then sees x * mask
                    f0.x = fmaf(f0.x, mk, rf0[q].x); f0.y = fmaf(f0.y, mk, rf0[q].y); f0.z = fmaf(f0.z, mk, rf0[q].z); f0.w = fmaf(f0.w, mk, rf0[q].w);
                    f1.x = fmaf(f1.x, mk, rf1[q].x); f1.y = fmaf(f1.y, mk, rf1[q].y); f1.z = fmaf(f1.z, mk, rf1[q].z); f1.w = fmaf(f1.w, mk, rf1[q].w);
                    const int px_ = it / (CC / 8), pw_ = px_ % PW, ph_ = px_ / PW;
                    if (slice == 0 && it < ITEMS && pin[q] && ph_ >= 1 && ph_ <= TH && pw_ >= 1 && pw_ <= 32) {
                        float* xo = p.pro_xout + ((long)b * p.H * p.W + (long)(h0 + ph_ - 1) * p.W + (w0 + pw_ - 1)) * p.Cin + cbase + pc8;
                        *reinterpret_cast<float4*>(xo) = f0;
                        *reinterpret_cast<float4*>(xo + 4) = f1;
                    }
                }
                uint4 v;
                v.x = pack2_lp(f0.x * mk, f0.y * mk); v.y = pack2_lp(f0.z * mk, f0.w * mk);
                v.z = pack2_lp(f1.x * mk, f1.y * mk); v.w = pack2_lp(f1.z * mk, f1.w * mk);
                if (it < ITEMS) *reinterpret_cast<uint4*>(patch + (it / (CC / 8)) * LDP + pc8) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) *reinterpret_cast<u32x4*>(wbuf + wlds[j]) = w0r[j];
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < WPT; ++j) *reinterpret_cast<u32x4*>(rbuf + wlds[j]) = rwr[j];
        }
        CSTAMP(2);
        // ---- nine taps; tap t's weights sit in wbuf[t & 1], taps t+1 .. t+RING are in registers / in flight
#pragma unroll
        for (int tt = 0; tt < NTAP; ++tt) {
            const int tap = tt % 9;                   // (the loop is unrolled: a constant)
            lds_barrier();                            // wbuf[tt & 1] (and the patch) visible; wbuf[(tt+1) & 1] is free
            if (tt + 1 < NTAP) {
                u16* wn = wbuf + ((tt + 1) & 1) * NSL * LDP;
#pragma unroll
                for (int j = 0; j < WPT; ++j) *reinterpret_cast<u32x4*>(wn + wlds[j]) = wr[tt % RING][j];
                if (tt + 1 + RING < NTAP) {
#pragma unroll
                    for (int j = 0; j < WPT; ++j)
                        wr[tt % RING][j] = *reinterpret_cast<const u32x4*>(Wg + wofs[j] + WS_TAPOFS(tt + 1 + RING) + cbase);
                }
                __builtin_amdgcn_sched_barrier(0);    // the refill loads issue before the MFMAs, not dripped between them
            }
#ifdef DEX_LP_WSPLIT
            if constexpr (RES) {
                if (tt == 5) {                        // every wave is past the centre tap (barrier above): the shortcut's lo half for tap 13
#pragma unroll
                    for (int j = 0; j < WPT; ++j) *reinterpret_cast<u32x4*>(rbuf + wlds[j]) = rwl[j];
                }
            }
#endif
            const u16* wb = wbuf + (tt & 1) * NSL * LDP;
            const int kh = tap / 3, kw = tap - kh * 3;
            const u16* ap = patch + ((wrow + kh) * PW + i + kw) * LDP + hh * 8;
            const u16* bp = wb + (wcol * NT * 32 + i) * LDP + hh * 8;
#pragma unroll
            for (int ks = 0; ks < CC / 16; ++ks) {
                const lp8 af = *reinterpret_cast<const lp8*>(ap + ks * 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const lp8 bf = *reinterpret_cast<const lp8*>(bp + t * 32 * LDP + ks * 16);
                    acc[t] = DEX_MFMA_LP(af, bf, acc[t], 0, 0, 0);
                }
                if constexpr (RES) {
                    if (tap == 4) {                   // centre tap: the same A fragment feeds the 1x1 shortcut
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const lp8 rf = *reinterpret_cast<const lp8*>(rbuf + (wcol * NT * 32 + t * 32 + i) * LDP + hh * 8 + ks * 16);
                            accr[t] = DEX_MFMA_LP(af, rf, accr[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (RES) cv_res_store<NT, COUT>(p, accr, b, h0 + wrow, w0, slice * NSL + wcol * NT * 32, lane);
    CSTAMP(3);
    cv_epilogue<NT, COUT>(p, acc, b, h0 + wrow, w0, slice * NSL + wcol * NT * 32, lane, tid, gnred);
#ifdef DEX_TIMING
    CSTAMP(4);
    if (p.dbg && tid == 0) {
        long long* d = p.dbg + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        for (int k = 0; k < 5; ++k) d[k] = tk[k];
        d[7] = __builtin_readcyclecounter() - tk0;
    }
#endif
}

template <int CC, int COUT, int NSL, int TH, bool PRO2 = false, bool RES = false, bool XB = false, int NW = 4>
static void launch_c3(const Conv3P& p, hipStream_t st) {
    constexpr int LDP = CC + 8;
    const size_t lds = ((size_t)(TH + 2) * 34 * LDP + (2 + (RES ? 1 : 0)) * NSL * LDP) * sizeof(u16);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lp_kernel<CC, COUT, NSL, TH, PRO2, RES, XB, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid((p.W + 31) / 32, (p.H + TH - 1) / TH, p.B * (COUT / NSL));
    Conv3P q = p;
    // padding-only tiles are looked for at batch size only: the test is one more dependent load at the head of a launch, and the small
    // grids are latency chains (B = 1, where nothing is padded anyway: 25.5 -> 25.1 k frames/s with the test on; B = 32 fp16x2: +0.5 ... 2 %)
    q.skip_dead = (p.B >= 4 && (long)grid.x * grid.y * grid.z >= 1024 && knob_or("DEX_CONV_SKIP_DEAD", 1) != 0) ? 1 : 0;
    static char sym[96];
    if (!sym[0]) snprintf(sym, sizeof sym, "conv3x3_lp_kernel<%d,%d,%d,%d,%d,%d,%d,%d>", CC, COUT, NSL, TH, (int)PRO2, (int)RES, (int)XB, NW);
    g_last_symbol = sym;
    hipLaunchKernelGGL((conv3x3_lp_kernel<CC, COUT, NSL, TH, PRO2, RES, XB, NW>), grid, dim3(64 * NW), lds, st, q);
}

bool conv3x3_bf16_tail_supported(int C) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return C == 64 || C == 128; }
bool conv3x3_bf16_xb_supported(int Cin, int Cout) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return Cin == Cout && (Cin == 64 || Cin == 128); }   // bf16 INPUT (needs pro_stats)
bool conv3x3_bf16_res_supported(int Cin, int Cout) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3_RES)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return (Cout == 128 && Cin == 64) || (Cout == 64 && (Cin == 128 || Cin == 256)); }
bool conv3x3_bf16_supported(int Cin, int Cout) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    return (Cin == 64 || Cin == 128 || Cin == 256) && (Cout == 64 || Cout == 128);
}

// plain (no GroupNorm prologue) 16-bit INPUT: only the eight-wave 64 -> 128 form with the fused shortcut is instantiated for it
// (the first conv after a Downsample at batch size; dex_api.hip lp_inter)
bool conv3x3_plain_lp_in_supported(int H, int W, int B, int Cin, int Cout) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    const int w8 = knob_or("DEX_CONV_W8", 1);
    const long tiles4 = (long)((W + 31) / 32) * ((H + 3) / 4) * B;
    return w8 && tiles4 >= 256 && Cin == 64 && Cout == 128;
}

// the up path's first conv (2C -> C with the fused 1x1 shortcut, plain input) reads a 16-bit concatenation buffer only on the
// eight-wave patch forms below (same conditions as launch_conv3x3_lp's branches for res_w, Cout == 64, Cin >= 128)
bool conv3x3_cat_lp_in_supported(int H, int W, int B, int Cin, int Cout) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_CONV3_RES)
    return false;            // no split-weight form yet (lp_config.h)
#endif
   
    if (Cout != 64 || Cin < 128 || Cin % 128 != 0) return false;
    const int w8 = knob_or("DEX_CONV_W8", 1);
    const long small_max = knob_or("DEX_CONV_SMALL_MAX", 256);
    const long tiles4 = (long)((W + 31) / 32) * ((H + 3) / 4) * B;
    return w8 && tiles4 >= small_max;
}

void launch_conv3x3_lp(const Conv3P& p, hipStream_t st) {
    if (const int tpw = conv3x3_stream_tiles(p)) { launch_conv3x3_stream(p, tpw, st); return; }   // batched synthesis
    if (conv3x3_regw_form(p)) { launch_conv3x3_regw(p, st); return; }                              // 128-channel layers, batched synthesis
    // few tiles (half resolution at small batch): 2-row tiles and 64-channel output slices put more, lighter
    // workgroups on the chip (per-tap weight traffic per workgroup halves, two workgroups fit per CU)
    const bool tail_ = p.pro_res != nullptr;     // (Conv3P::res2_* is served by the ping-pong strip kernel only: conv3x3_res2_form)
    const long tiles4 = (long)((p.W + 31) / 32) * ((p.H + 3) / 4) * p.B;
    const long small_max = knob_or("DEX_CONV_SMALL_MAX", 256);
    const bool small = tiles4 < small_max;       // (at B=32 the 4-row tiles win despite one workgroup per CU: 189 vs 218 us)
    // the 128-channel-wide forms at 4-row tiles fill the LDS with ONE workgroup per CU; as eight waves (4 rows x 2 halves
    // of the output channels) that workgroup keeps two waves per SIMD busy instead of one: +8.6 % end to end at B=32
    // (DEX_CONV_W8=0 restores the four-wave form)
    const int w8 = knob_or("DEX_CONV_W8", 1);
    // 8-row tiles for the 128 -> 128 convs at batch size: each wave owns one row x all 128 output channels, the per-tap weight
    // slices are staged once per 256 pixels instead of once per 128 and there is one barrier per 32 MFMAs of a wave instead of
    // per 16 (the patch + two weight taps fill the LDS to within 64 bytes).  DEX_CONV_TH8=0 keeps the 4-row form.
    const int th8 = knob_or("DEX_CONV_TH8", 1);
    // (measured at B=32: 156 -> 139 us; the fused-tail form needs eight staging passes at this size and loses, 207 -> 243: 4-row tiles)
    if (th8 && w8 && !small && !tail_ && p.Cin == 128 && p.Cout == 128 && !p.res_w && p.H % 8 == 0 && (long)((p.W + 31) / 32) * (p.H / 8) * p.B >= 512) {
        p.x_bf16 ? launch_c3<128, 128, 128, 8, false, false, true, 8>(p, st) : launch_c3<128, 128, 128, 8, false, false, false, 8>(p, st);
        return;
    }
    if (w8 && !small && p.Cin == 128 && p.Cout == 128 && !p.res_w) {
        if (p.x_bf16) { tail_ ? launch_c3<128, 128, 128, 4, true, false, true, 8>(p, st) : launch_c3<128, 128, 128, 4, false, false, true, 8>(p, st); }
        else { tail_ ? launch_c3<128, 128, 128, 4, true, false, false, 8>(p, st) : launch_c3<128, 128, 128, 4, false, false, false, 8>(p, st); }
        return;
    }
    if (w8 && !small && p.res_w && p.Cout == 128 && p.Cin == 64) {
        p.x_bf16 ? launch_c3<64, 128, 128, 4, false, true, true, 8>(p, st) : launch_c3<64, 128, 128, 4, false, true, false, 8>(p, st);      // (16-bit plain input: batch regime only)
        return;
    }
    if (th8 && w8 && !small && p.res_w && p.Cout == 64 && p.Cin >= 128 && !p.pro_stats && p.H % 8 == 0 && (long)((p.W + 31) / 32) * (p.H / 8) * p.B >= 512) {
        // 8-row tiles: a wave owns one row x 64 channels (32 at four rows); x_bf16: the concatenation buffer in the mode's 16-bit type
        p.x_bf16 ? launch_c3<128, 64, 64, 8, false, true, true, 8>(p, st) : launch_c3<128, 64, 64, 8, false, true, false, 8>(p, st);
        return;
    }
    if (w8 && !small && p.res_w && p.Cout == 64 && p.Cin >= 128) {
        p.x_bf16 ? launch_c3<128, 64, 64, 4, false, true, true, 8>(p, st) : launch_c3<128, 64, 64, 4, false, true, false, 8>(p, st);
        return;
    }
    if (p.x_bf16) {       // raw conv output stored as bf16: the GroupNorm-prologue forms with Cin == Cout (conv3x3_bf16_xb_supported)
        if (tail_) {
            if (p.Cin == 64) { small ? launch_c3<64, 64, 64, 2, true, false, true>(p, st) : launch_c3<64, 64, 64, 4, true, false, true>(p, st); }
            else { small ? launch_c3<128, 128, 64, 2, true, false, true>(p, st) : launch_c3<128, 128, 128, 4, true, false, true>(p, st); }
        } else {
            if (p.Cin == 64) { small ? launch_c3<64, 64, 64, 2, false, false, true>(p, st) : launch_c3<64, 64, 64, 4, false, false, true>(p, st); }
            else { small ? launch_c3<128, 128, 64, 2, false, false, true>(p, st) : launch_c3<128, 128, 128, 4, false, false, true>(p, st); }
        }
        return;
    }
    if (tail_) {      // fused ResnetBlock tail in front: only the Cin == Cout shapes of the second block's first conv
        if (p.Cin == 64 && p.Cout == 64) { small ? launch_c3<64, 64, 64, 2, true>(p, st) : launch_c3<64, 64, 64, 4, true>(p, st); }
        else { small ? launch_c3<128, 128, 64, 2, true>(p, st) : launch_c3<128, 128, 128, 4, true>(p, st); }
        return;
    }
    if (p.res_w) {        // fused 1x1 shortcut: the channel-changing first convs (64 -> 128 on the way down, 128/256 -> 64 on the way up)
        if (p.Cout == 128 && p.Cin == 64) { small ? launch_c3<64, 128, 64, 2, false, true>(p, st) : launch_c3<64, 128, 128, 4, false, true>(p, st); return; }
        if (p.Cout == 64 && p.Cin >= 128) { small ? launch_c3<128, 64, 64, 2, false, true>(p, st) : launch_c3<128, 64, 64, 4, false, true>(p, st); return; }
    }
    if (p.Cout == 64) {
        if (p.Cin == 64) { small ? launch_c3<64, 64, 64, 2>(p, st) : launch_c3<64, 64, 64, 4>(p, st); }
        else { small ? launch_c3<128, 64, 64, 2>(p, st) : launch_c3<128, 64, 64, 4>(p, st); }      // Cin 128 / 256 (two chunks)
    } else {
        if (p.Cin == 64) { small ? launch_c3<64, 128, 64, 2>(p, st) : launch_c3<64, 128, 128, 4>(p, st); }
        else { small ? launch_c3<128, 128, 64, 2>(p, st) : launch_c3<128, 128, 128, 4>(p, st); }
    }
}

}  // namespace DEX_LP_NS
}  // namespace dex
