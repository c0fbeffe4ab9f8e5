// linattn_fused.hip — LinearAttention (diffusion.py:74-92) without ever materialising q, k or v (bf16-MFMA mode).
//
// Algebra: q enters linearly (out[e,n] = sum_d ctx[d,e] q[d,n], q = Wq x), so the whole Residual(Rezero(
// LinearAttention)) collapses to   y = x + M_b x + g*b_out   with a per-utterance C x C matrix
//     M_b = g * Wout * blockdiag_h(ctx_h^T) * Wq,      ctx_h[d,e] = sum_n softmax_n(k)[d,n] v[e,n].
// Kernel 1 (this file) streams x once: per 32-pixel sub-tile a wave computes k and v (8 tiles of 32x32) with
// v_mfma_f32_32x32x16_bf16, keeps an online per-channel max, and accumulates ctx^T with a SECOND MFMA whose A
// and B operands are the k/v accumulator registers themselves (C-layout: lane = channel, registers = pixels ->
// exactly the A[e][px] / B[px][d] fragment shape; both use the same pixel order, so no shuffle and no LDS).
// Kernel 2 merges the workgroup partials (flash-style) into normalised ctx and folds it with Wout and g into
// W2 = g Wout blockdiag(ctx^T) (C x 128, bf16); kernel 3 is the tail  y = x + W2 (Wq x) + g b  (two chained MFMA
// GEMMs per 32-pixel wave tile, M_b itself is never formed).
// HBM traffic per call: x read twice + y written, instead of writing q,k,v (6x the size of x) and re-reading them.
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
union LFrag { uint4 u; lp8 v; };
constexpr float LOG2E_F = 1.4426950408889634f;


// grid (nblk, B); 256 threads; each wave owns `nsub` consecutive 32-pixel sub-tiles.
// part_m/part_s: [B][4][nblk][32], part_c: [B][4][nblk][32 d][32 e]   (same layout linattn_combine reads)
// FL >= 0 (round 5): the flags of the PRO form as COMPILE-TIME constants - bit 0 h2_bf16, bit 1 res_lp, bit 2 xout_lp, bit 3 res_under_mask.  As
// run-time (uniform) flags they became ~180 v_cndmask + both sides of every select per 32-pixel sub-tile of a VALU-bound loop; the
// launcher instantiates every combination with 16-bit h2 (the default of the reduced-precision modes) and keeps FL = -1 for the rest
// (DEX_H_BF16=0).
template <int C, bool PRO, int FL = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C == 64 ? 2 : 1, C == 64 ? 2 : 8))) void linattn_kvctx_kernel(const LinKvCtxP p) {
    constexpr int LDW = C + 8, KS = C / 16;
    extern __shared__ __attribute__((aligned(16))) u16 smem_la[];
    u16* Ws = smem_la;                                       // [256][LDW]  rows: k(4x32) then v(4x32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int blk = blockIdx.x, b = blockIdx.y;
    const u16* Wg = reinterpret_cast<const u16*>(p.Wkv);     // bf16 [256][C]
    {   // all weight loads in flight together, then the LDS stores (a load->store loop serialises 8-16 round trips)
        uint4 wr[C / 8];
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            wr[j] = *reinterpret_cast<const uint4*>(Wg + (long)(it / (C / 8)) * C + (it % (C / 8)) * 8);
        }
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            *reinterpret_cast<uint4*>(Ws + (it / (C / 8)) * LDW + (it % (C / 8)) * 8) = wr[j];
        }
#ifdef DEX_LP_WSPLIT
        // the lo halves of the same rows (p.wkv_lo_off elements behind), a second LDS image behind the first
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            wr[j] = *reinterpret_cast<const uint4*>(Wg + p.wkv_lo_off + (long)(it / (C / 8)) * C + (it % (C / 8)) * 8);
        }
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
            const int it = tid + 256 * j;
            *reinterpret_cast<uint4*>(Ws + 256 * LDW + (it / (C / 8)) * LDW + (it % (C / 8)) * 8) = wr[j];
        }
#endif
    }
    const float* X = PRO ? p.H2 + (long)b * p.npix * C : p.X + (long)b * p.xb + p.x_coff;
    const int ldx = PRO ? C : p.ldx;
    const float* R = (PRO && p.res) ? p.res + (long)b * p.resb : nullptr;
    const float* mrow = PRO ? p.mask + (long)b * p.mask_bstride : nullptr;
    const int px_base = (blk * 4 + wave) * p.nsub * 32;

    f32x16 ctxT[4];
    float m_run[4], s_run[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        m_run[h] = -INFINITY; s_run[h] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT[h][r] = 0.f;
    }
    // x rows of the first sub-tile go out before the barrier (beside the weight loads); later sub-tiles are
    // prefetched one iteration ahead.  PRO: the rows are the raw conv output + the residual, turned into x below.
    float4 xa[KS], xc[KS], ra[PRO ? KS : 1], rc[PRO ? KS : 1];
    float mkv = 1.f;
    float ga_pre = 0.f, be_pre = 0.f;                       // GroupNorm affine of channel tid, requested with the first loads
    if constexpr (PRO) { if (tid < C) { ga_pre = p.gamma[tid]; be_pre = p.beta[tid]; } }
    const bool hb = FL >= 0 ? (FL & 1) != 0 : (PRO && p.h2_bf16 != 0);      // H2 stored as bf16: xa holds the 8 raw values until they are used
    const bool rlp = FL >= 0 ? (FL & 2) != 0 : (PRO && p.res_lp != 0);      // the residual rows likewise (ra holds 8 raw 16-bit values)
    const bool xlp = FL >= 0 ? (FL & 4) != 0 : (p.xout_lp != 0);            // x leaves in the mode's 16-bit type
    const unsigned short* Rh = (PRO && p.res) ? reinterpret_cast<const unsigned short*>(p.res) + (long)b * p.resb : nullptr;
    const unsigned short* Xh = PRO ? reinterpret_cast<const unsigned short*>(p.H2) + (long)b * p.npix * C : nullptr;
    {
        const int pxr = min(px_base + i, p.npix - 1);
        const float* xr = X + (long)pxr * ldx + hh * 8;
        if (hb) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { xa[ks] = *reinterpret_cast<const float4*>(Xh + (long)pxr * C + hh * 8 + ks * 16); xc[ks] = xa[ks]; }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xa[ks] = *reinterpret_cast<const float4*>(xr + ks * 16);
                xc[ks] = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
            }
        }
        if constexpr (PRO) {
            if (R) {
                if (rlp) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) { ra[ks] = *reinterpret_cast<const float4*>(Rh + (long)pxr * p.ldres + hh * 8 + ks * 16); rc[ks] = ra[ks]; }
                } else {
                    const float* rr = R + (long)pxr * p.ldres + hh * 8;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        ra[ks] = *reinterpret_cast<const float4*>(rr + ks * 16);
                        rc[ks] = *reinterpret_cast<const float4*>(rr + ks * 16 + 4);
                    }
                }
            }
            mkv = mrow[(pxr % p.W) * p.mask_ws];
        }
    }
    __shared__ float smean[8], srstd[8];
    __shared__ __attribute__((aligned(16))) float gsc_s[PRO ? C : 4], gsh_s[PRO ? C : 4];   // GroupNorm folded to x*gsc + gsh per channel
    if constexpr (PRO) {
        {   // 8 groups x GN_SLOTS partials == 256 threads
            const int g = tid / GN_SLOTS;
            const longlong2 sv = *reinterpret_cast<const longlong2*>(p.gn_stats + (((long)b * 8 + g) * GN_SLOTS + (tid % GN_SLOTS)) * 2);
            long long s1 = sv.x, s2 = sv.y;
            gn_slots_reduce<GN_SLOTS>(s1, s2);
            if ((tid % GN_SLOTS) == 0) gn_moments(s1, s2, 1e-5, smean[g], srstd[g]);
        }
    }
    __syncthreads();
    if constexpr (PRO) {
        if (tid < C) {
            const float ga = ga_pre * srstd[tid / (C / 8)];
            gsc_s[tid] = ga;
            gsh_s[tid] = be_pre - smean[tid / (C / 8)] * ga;
        }
        __syncthreads();
    }
    for (int sub = 0; sub < p.nsub; ++sub) {
        const int px0 = px_base + sub * 32;
        if (px0 >= p.npix) break;
        if constexpr (PRO) {
            // x = mask * (Mish(GN(h2)) + r)   (identity shortcut)   or   mask * Mish(GN(h2)) + r   (res_conv shortcut)
            const bool under = FL >= 0 ? (FL & 8) != 0 : (p.res_under_mask != 0);
            const bool live = px0 + i < p.npix;
            float* xo = p.Xout + ((long)b * p.npix + min(px0 + i, p.npix - 1)) * C + hh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8] = {xa[ks].x, xa[ks].y, xa[ks].z, xa[ks].w, xc[ks].x, xc[ks].y, xc[ks].z, xc[ks].w};
                if (hb) {
                    const unsigned u0 = __float_as_uint(xa[ks].x), u1 = __float_as_uint(xa[ks].y), u2 = __float_as_uint(xa[ks].z), u3 = __float_as_uint(xa[ks].w);
                    v[0] = lp_lo(u0); v[1] = lp_hi(u0); v[2] = lp_lo(u1); v[3] = lp_hi(u1);
                    v[4] = lp_lo(u2); v[5] = lp_hi(u2); v[6] = lp_lo(u3); v[7] = lp_hi(u3);
                }
                float r_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float4 sc0 = *reinterpret_cast<const float4*>(gsc_s + ks * 16 + hh * 8), sc1 = *reinterpret_cast<const float4*>(gsc_s + ks * 16 + hh * 8 + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(gsh_s + ks * 16 + hh * 8), sh1 = *reinterpret_cast<const float4*>(gsh_s + ks * 16 + hh * 8 + 4);
                const float gsc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                const float gsh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
                if (R) {
                    if (rlp) {
                        const unsigned u0 = __float_as_uint(ra[ks].x), u1 = __float_as_uint(ra[ks].y), u2 = __float_as_uint(ra[ks].z), u3 = __float_as_uint(ra[ks].w);
                        r_[0] = lp_lo(u0); r_[1] = lp_hi(u0); r_[2] = lp_lo(u1); r_[3] = lp_hi(u1); r_[4] = lp_lo(u2); r_[5] = lp_hi(u2); r_[6] = lp_lo(u3); r_[7] = lp_hi(u3);
                    } else { r_[0] = ra[ks].x; r_[1] = ra[ks].y; r_[2] = ra[ks].z; r_[3] = ra[ks].w; r_[4] = rc[ks].x; r_[5] = rc[ks].y; r_[6] = rc[ks].z; r_[7] = rc[ks].w; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = fmaf(v[j], gsc[j], gsh[j]);
                    const float e = __expf(fminf(t, 20.f));
                    const float n = e * (e + 2.f);
                    const float y = t * (n * __builtin_amdgcn_rcpf(n + 2.f));   // Mish (rcp: 1 ulp, bf16 consumers)
                    v[j] = under ? (y + r_[j]) * mkv : fmaf(y, mkv, r_[j]);
                }
                xa[ks] = make_float4(v[0], v[1], v[2], v[3]); xc[ks] = make_float4(v[4], v[5], v[6], v[7]);
                if (live) {
                    if (xlp) {       // 16-bit x for the tail kernel: its q operand rounds x the same way, its residual term reads the rounded value
                        u16* xh = reinterpret_cast<u16*>(p.Xout) + ((long)b * p.npix + min(px0 + i, p.npix - 1)) * C + hh * 8 + ks * 16;
                        *reinterpret_cast<uint4*>(xh) = make_uint4(pack2_lp(v[0], v[1]), pack2_lp(v[2], v[3]), pack2_lp(v[4], v[5]), pack2_lp(v[6], v[7]));
                    } else {
                        *reinterpret_cast<float4*>(xo + ks * 16) = xa[ks];
                        *reinterpret_cast<float4*>(xo + ks * 16 + 4) = xc[ks];
                    }
                }
            }
        }
        // A fragments of x: lane (pixel i, half hh) holds x[px][ks*16 + hh*8 .. +8]
        LFrag af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            af[ks].u.x = pack2_lp(xa[ks].x, xa[ks].y); af[ks].u.y = pack2_lp(xa[ks].z, xa[ks].w);
            af[ks].u.z = pack2_lp(xc[ks].x, xc[ks].y); af[ks].u.w = pack2_lp(xc[ks].z, xc[ks].w);
        }
        if (sub + 1 < p.nsub) {
            const int pxr = min(px0 + 32 + i, p.npix - 1);
            const float* xr = X + (long)pxr * ldx + hh * 8;
            if (hb) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { xa[ks] = *reinterpret_cast<const float4*>(Xh + (long)pxr * C + hh * 8 + ks * 16); xc[ks] = xa[ks]; }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    xa[ks] = *reinterpret_cast<const float4*>(xr + ks * 16);
                    xc[ks] = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
                }
            }
            if constexpr (PRO) {
                if (R) {
                    if (rlp) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) { ra[ks] = *reinterpret_cast<const float4*>(Rh + (long)pxr * p.ldres + hh * 8 + ks * 16); rc[ks] = ra[ks]; }
                    } else {
                        const float* rr = R + (long)pxr * p.ldres + hh * 8;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            ra[ks] = *reinterpret_cast<const float4*>(rr + ks * 16);
                            rc[ks] = *reinterpret_cast<const float4*>(rr + ks * 16 + 4);
                        }
                    }
                }
                mkv = mrow[(pxr % p.W) * p.mask_ws];
            }
        }
        // head by head: k_h and v_h tiles (2 x 16 accumulator registers live instead of the 128 of all eight tiles at once -
        // the kernel sat at one wave per SIMD), softmax statistics, then ctx_h += v_h^T p_h from those registers
        // (round 6) SEED forms: the ragged last sub-tile's pixels >= npix enter as -inf seeds of the k accumulators, once per sub-tile,
        // instead of a compare + select per element and head (124 of the 537 vector instructions of this block); the 16 seed registers
        // fit beside a 16-bit residual prefetch (the batch forms), with an fp32 one the 64-channel form would spill (12 registers).
        constexpr bool SEED = C == 128 || (FL >= 0 && (FL & 2) != 0);
        f32x16 kseed;
#pragma unroll
        for (int r = 0; r < 16; ++r) kseed[r] = (SEED && px0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= p.npix) ? -INFINITY : 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x16 kh, vh;
#pragma unroll
            for (int r = 0; r < 16; ++r) { kh[r] = SEED ? kseed[r] : 0.f; vh[r] = 0.f; }
            const u16* bk = Ws + (h * 32 + i) * LDW + hh * 8;
            const u16* bv = Ws + ((4 + h) * 32 + i) * LDW + hh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                LFrag fk, fv; fk.u = *reinterpret_cast<const uint4*>(bk + ks * 16); fv.u = *reinterpret_cast<const uint4*>(bv + ks * 16);
                kh = DEX_MFMA_LP(af[ks].v, fk.v, kh, 0, 0, 0);
                vh = DEX_MFMA_LP(af[ks].v, fv.v, vh, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
                LFrag lk, lv; lk.u = *reinterpret_cast<const uint4*>(bk + 256 * LDW + ks * 16); lv.u = *reinterpret_cast<const uint4*>(bv + 256 * LDW + ks * 16);
                kh = DEX_MFMA_LP(af[ks].v, lk.v, kh, 0, 0, 0);
                vh = DEX_MFMA_LP(af[ks].v, lv.v, vh, 0, 0, 0);
#endif
            }
            // column (channel d = lane&31) max over the 32 pixels of the sub-tile
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (!SEED) { if (px0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= p.npix) kh[r] = -INFINITY; }
                mx = fmaxf(mx, kh[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // fp16 operands: the softmax reference sits 14 octaves BELOW the running maximum, so p = exp(k - ref) spans
            // [.., 2^14] and a position 2^-38 below the maximum still survives the fp16 rounding of the MFMA operand (over
            // n = 80*T positions nearly all of them sit far below the maximum); the partial's (m, s, ctx) are consistent
            // with that reference, so nothing downstream changes.  bf16 has fp32's exponent range: no shift.
            constexpr float KSHIFT = LP_IS_F16 ? 14.f * 0.69314718056f : 0.f;
            const float mn = fmaxf(m_run[h], mx - KSHIFT);
            const float alpha = __expf(m_run[h] - mn);
            m_run[h] = mn;
            const float nmn2 = -mn * LOG2E_F;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { kh[r] = __builtin_amdgcn_exp2f(fmaf(kh[r], LOG2E_F, nmn2)); ps += kh[r]; }
            s_run[h] = s_run[h] * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctxT[h][r] *= alpha;
            // ctx^T[e][d] += sum_px v[px][e] * p[px][d]   (A = v tile regs, B = p tile regs, same pixel order)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                LFrag va, pb;
                va.u.x = pack2_lp(vh[8 * k2 + 0], vh[8 * k2 + 1]); va.u.y = pack2_lp(vh[8 * k2 + 2], vh[8 * k2 + 3]);
                va.u.z = pack2_lp(vh[8 * k2 + 4], vh[8 * k2 + 5]); va.u.w = pack2_lp(vh[8 * k2 + 6], vh[8 * k2 + 7]);
                pb.u.x = pack2_lp(kh[8 * k2 + 0], kh[8 * k2 + 1]); pb.u.y = pack2_lp(kh[8 * k2 + 2], kh[8 * k2 + 3]);
                pb.u.z = pack2_lp(kh[8 * k2 + 4], kh[8 * k2 + 5]); pb.u.w = pack2_lp(kh[8 * k2 + 6], kh[8 * k2 + 7]);
                ctxT[h] = DEX_MFMA_LP(va.v, pb.v, ctxT[h], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the heads sequential: interleaving them brings all tiles back to life
        }
    }
    // ---- merge the 4 waves of the workgroup through LDS, write one partial per head
    __syncthreads();                                          // weights no longer needed
    float* mS = reinterpret_cast<float*>(smem_la);            // [4 waves][4 heads][32 d][33]  ctx as [d][e]
    float* mm = mS + 4 * 4 * 32 * 33;                         // [4][4][32] m
    float* ms = mm + 4 * 4 * 32;                              // [4][4][32] s
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float st = s_run[h] + __shfl_xor(s_run[h], 32);
        float* dst = mS + ((wave * 4 + h) * 32 + i) * 33;     // row d = lane&31
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2) + 4 * hh] = ctxT[h][r];   // column e
        if (hh == 0) { mm[(wave * 4 + h) * 32 + i] = m_run[h]; ms[(wave * 4 + h) * 32 + i] = st; }
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * 1024; idx += 256) {
        const int h = idx >> 10, d = (idx >> 5) & 31, e = idx & 31;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, mm[(w * 4 + h) * 32 + d]);
        float acc = 0.f, s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = mm[(w * 4 + h) * 32 + d];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            acc = fmaf(f, mS[((w * 4 + h) * 32 + d) * 33 + e], acc);
            s = fmaf(f, ms[(w * 4 + h) * 32 + d], s);
        }
        const long pidx = ((long)b * 4 + h) * p.nblk + blk;
        p.part_c[pidx * 1024 + d * 32 + e] = acc;
        if (e == 0) { p.part_m[pidx * 32 + d] = M; p.part_s[pidx * 32 + d] = s; }
    }
}

void launch_linattn_kvctx(const LinKvCtxP& p, hipStream_t st) {
#ifdef DEX_LP_WSPLIT
    const size_t lds_w = (size_t)2 * 256 * (p.C + 8) * sizeof(u16);          // hi + lo images of the k | v rows
    constexpr int LDS_MAX = 144 * 1024;
#else
    const size_t lds_w = (size_t)256 * (p.C + 8) * sizeof(u16);
    constexpr int LDS_MAX = 96 * 1024;
#endif
    const size_t lds_m = (size_t)(4 * 4 * 32 * 33 + 2 * 4 * 4 * 32) * sizeof(float);
    const size_t lds = lds_w > lds_m ? lds_w : lds_m;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
#define KVCTX_ATTR(CC, F) hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_kvctx_kernel<CC, true, F>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        KVCTX_ATTR(64, 1) KVCTX_ATTR(64, 3) KVCTX_ATTR(64, 5) KVCTX_ATTR(64, 7) KVCTX_ATTR(64, 9) KVCTX_ATTR(64, 11) KVCTX_ATTR(64, 13) KVCTX_ATTR(64, 15)
        KVCTX_ATTR(128, 1) KVCTX_ATTR(128, 3) KVCTX_ATTR(128, 5) KVCTX_ATTR(128, 7) KVCTX_ATTR(128, 9) KVCTX_ATTR(128, 11) KVCTX_ATTR(128, 13) KVCTX_ATTR(128, 15)
#undef KVCTX_ATTR
        attr = true;
    }
    dim3 grid(p.nblk, p.B);
    const bool pro = p.H2 != nullptr;
    const int fl = pro ? (p.h2_bf16 != 0 ? 1 : 0) | (p.res_lp != 0 ? 2 : 0) | (p.xout_lp != 0 ? 4 : 0) | (p.res_under_mask != 0 ? 8 : 0) : -1;
#define KVCTX_CASE(CC, F) case F: hipLaunchKernelGGL((linattn_kvctx_kernel<CC, true, F>), grid, dim3(256), lds, st, p); break;
#define KVCTX_LAUNCH(CC)                                                                                                              \
    switch (fl) {                                                                                                                      \
        KVCTX_CASE(CC, 1) KVCTX_CASE(CC, 3) KVCTX_CASE(CC, 5) KVCTX_CASE(CC, 7) KVCTX_CASE(CC, 9) KVCTX_CASE(CC, 11) KVCTX_CASE(CC, 13) KVCTX_CASE(CC, 15)   \
        default:                                                                                                                       \
            if (pro) hipLaunchKernelGGL((linattn_kvctx_kernel<CC, true>), grid, dim3(256), lds, st, p);                                \
            else hipLaunchKernelGGL((linattn_kvctx_kernel<CC, false>), grid, dim3(256), lds, st, p);                                   \
    }
    if (p.C == 64) { KVCTX_LAUNCH(64) } else { KVCTX_LAUNCH(128) }
#undef KVCTX_LAUNCH
#undef KVCTX_CASE
}

// grid (4 heads, B, 32 rows d): merge the workgroup partials of ONE context row -> normalised ctx[b][h][d][:], then
// fold it with the output projection:  W2[co][h*32+d] = g * sum_e Wout[co][h*32+e] * ctx[h][d][e]   (one column of
// W2 = g * Wout * blockdiag(ctx^T) per workgroup).  W2 is written as bf16 in the A-fragment order the tail kernel
// consumes: he index permuted to the accumulator row order of the q^T tiles (see linattn_out2_kernel).
// thread = (column e = tid%32, partial lane pl = tid/32): 8 lanes stride the partial list with independent loads.
template <int MC>
__device__ __forceinline__ void merge_fold(const LinMergeP& p, long pbase, int d, int e, int pl, float& m, float& acc, float& s) {
    for (int c0 = 0; c0 < p.nblk; c0 += 8 * MC) {
        float pm[MC], ps[MC], pc[MC];
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = min(c0 + pl + 8 * k, p.nblk - 1);           // clamped: loads are unconditional
            pm[k] = p.part_m[(pbase + c) * 32 + d];
            ps[k] = p.part_s[(pbase + c) * 32 + d];
            pc[k] = p.part_c[(pbase + c) * 1024 + d * 32 + e];
        }
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            if (c0 + pl + 8 * k < p.nblk) {
                const float mn = fmaxf(m, pm[k]);
                const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
                const float w = (pm[k] == -INFINITY) ? 0.f : __expf(pm[k] - mn);
                acc = fmaf(acc, a, w * pc[k]);
                s = fmaf(s, a, w * ps[k]);
                m = mn;
            }
        }
    }
}

__global__ __launch_bounds__(256) void linattn_merge_kernel(const LinMergeP p) {
    __shared__ float redm[8], reda[8][32], reds[8], cvec[32];
    const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y, d = blockIdx.z;
    const int C = p.C;
    // this thread's Wout row slice (co = tid): 32 floats, in flight under the partial merge
    float4 wo[8];
    if (tid < C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wo[j] = *reinterpret_cast<const float4*>(p.Wout + (long)tid * 128 + h * 32 + j * 4);
    }
    const float g = p.g[0];
    const long pbase = ((long)b * 4 + h) * p.nblk;
    const int e = tid & 31, pl = tid >> 5;
    // ONE pass: every lane first requests its whole share of the partial list (up to MC entries per round, all loads
    // in flight together), then folds them with a running maximum (flash-style).  The two-pass form (maximum, then
    // weighted sum) walked the list twice in rounds of 8 dependent-latency loads: 9.3 us at 160 partials, B=1.
    float m = -INFINITY, acc = 0.f, s = 0.f;
    const int per_lane = (p.nblk + 7) / 8;
    if (per_lane <= 8) merge_fold<8>(p, pbase, d, e, pl, m, acc, s);
    else if (per_lane <= 16) merge_fold<16>(p, pbase, d, e, pl, m, acc, s);
    else merge_fold<24>(p, pbase, d, e, pl, m, acc, s);
    if (e == 0) { redm[pl] = m; reds[pl] = s; }
    reda[pl][e] = acc;
    __syncthreads();
    if (pl == 0) {
        float M = redm[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) M = fmaxf(M, redm[k]);
        float a = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float w = (redm[k] == -INFINITY) ? 0.f : __expf(redm[k] - M);
            a = fmaf(w, reda[k][e], a); ss = fmaf(w, reds[k], ss);
        }
        cvec[e] = a / ss;
    }
    __syncthreads();
    if (tid < C) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a = fmaf(wo[j].x, cvec[j * 4 + 0], a); a = fmaf(wo[j].y, cvec[j * 4 + 1], a);
            a = fmaf(wo[j].z, cvec[j * 4 + 2], a); a = fmaf(wo[j].w, cvec[j * 4 + 3], a);
        }
        a *= g;
        // fragment slot of (co = tid, he = h*32 + d):  d = (j&3) + 8*(2*ksl + (j>>2)) + 4*hh
        const int co = tid, hh = (d >> 2) & 1, j = (d & 3) + 4 * ((d >> 3) & 1), ks = 2 * h + (d >> 4);
        const long dst = ((((long)b * (C / 32) + (co >> 5)) * 8 + ks) * 64 + hh * 32 + (co & 31)) * 8 + j;
        reinterpret_cast<u16*>(p.W2)[dst] = (u16)(pack2_lp(a, 0.f) & 0xffffu);
    }
}
void launch_linattn_merge(const LinMergeP& p, hipStream_t st) {
    hipLaunchKernelGGL(linattn_merge_kernel, dim3(4, p.B, 32), dim3(256), 0, st, p);
}

// Tail, latency form (small grids): y = x + W2 (Wq x) + g*b per pixel, as two chained MFMA GEMMs computed TRANSPOSED so that no operand ever
// needs a layout change:  q^T[he][px] = Wq[he][:] . x[px][:]  (A = Wq rows, B = x rows),  then
// y^T[co][px] = sum_he W2[co][he] q^T[he][px]  (A = W2, B = the q^T accumulators re-used in place: a lane already
// holds, for its pixel column, 8 he values per K-step — in accumulator row order, which is why W2 is stored with
// that permutation).  Each wave owns 32 pixels end to end: no LDS, no barrier, every global load issued up front
// (at B=1 the LDS-staged form below costs 1.5 % end to end: one more dependent round trip in a latency-bound chain).
// grid (ceil(npix/128), B), 256 threads.
template <int C>
__global__ __launch_bounds__(256) void linattn_out2_direct_kernel(const LinOut2P p) {
    constexpr int KS1 = C / 16, CT = C / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int px0 = (blockIdx.x * 4 + wave) * 32;
    if (px0 >= p.npix) return;
    const float* X = p.X + (long)b * p.xb + p.x_coff;
    const int px = min(px0 + i, p.npix - 1);
    // B fragments of GEMM1: lane (pixel column i, half hh) holds x[px][ks*16 + hh*8 .. +8]
    const float* xr = X + (long)px * p.ldx + hh * 8;
    float4 xa[KS1], xc[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        xa[ks] = *reinterpret_cast<const float4*>(xr + ks * 16);
        xc[ks] = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
    }
    // A fragments of GEMM2 (all of W2 for this utterance: CT x 8 K-steps)
    const uint4* w2 = reinterpret_cast<const uint4*>(p.W2) + (long)b * CT * 8 * 64 + lane;
    uint4 w2f[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w2f[ct][ks] = w2[(ct * 8 + ks) * 64];
    LFrag xf[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        xf[ks].u.x = pack2_lp(xa[ks].x, xa[ks].y); xf[ks].u.y = pack2_lp(xa[ks].z, xa[ks].w);
        xf[ks].u.z = pack2_lp(xc[ks].x, xc[ks].y); xf[ks].u.w = pack2_lp(xc[ks].z, xc[ks].w);
    }
    // ---- GEMM1: q^T, four 32-row he tiles; converted to bf16 B fragments tile by tile
    const uint4* wq = reinterpret_cast<const uint4*>(p.Wq) + lane;                  // bf16, fragment order [he tile][K-step][lane][8]
    LFrag qf[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 q;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            LFrag af; af.u = wq[(t * KS1 + ks) * 64];
            q = DEX_MFMA_LP(af.v, xf[ks].v, q, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
            LFrag al; al.u = wq[(t * KS1 + ks) * 64 + p.wq_lo_off / 8];       // the lo half of the same fragment
            q = DEX_MFMA_LP(al.v, xf[ks].v, q, 0, 0, 0);
#endif
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            qf[t * 2 + k2].u.x = pack2_lp(q[8 * k2 + 0], q[8 * k2 + 1]); qf[t * 2 + k2].u.y = pack2_lp(q[8 * k2 + 2], q[8 * k2 + 3]);
            qf[t * 2 + k2].u.z = pack2_lp(q[8 * k2 + 4], q[8 * k2 + 5]); qf[t * 2 + k2].u.w = pack2_lp(q[8 * k2 + 6], q[8 * k2 + 7]);
        }
    }
    // ---- GEMM2 + epilogue: rows co = ct*32 + (r&3) + 8*(r>>2) + 4*hh  ->  4 consecutive channels per register quad
    float* Y = p.Y + (long)b * p.yb + p.y_coff;
    const bool ok = px0 + i < p.npix;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = 0.f;
        float4 res[4], bia[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = ct * 32 + 8 * g4 + 4 * hh;
            res[g4] = *reinterpret_cast<const float4*>(X + (long)px * p.ldx + co);
            bia[g4] = *reinterpret_cast<const float4*>(p.bias + co);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            LFrag af; af.u = w2f[ct][ks];
            y = DEX_MFMA_LP(af.v, qf[ks].v, y, 0, 0, 0);
        }
        if (ok) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = ct * 32 + 8 * g4 + 4 * hh;
                float4 o;
                o.x = y[g4 * 4 + 0] + res[g4].x + bia[g4].x; o.y = y[g4 * 4 + 1] + res[g4].y + bia[g4].y;
                o.z = y[g4 * 4 + 2] + res[g4].z + bia[g4].z; o.w = y[g4 * 4 + 3] + res[g4].w + bia[g4].w;
                *reinterpret_cast<float4*>(Y + (long)(px0 + i) * p.ldy + co) = o;
            }
        }
    }
}
// Tail, throughput form (large grids: +1.2 % end to end at B=32): y = x + W2 (Wq x) + g*b per pixel, two chained MFMA GEMMs computed TRANSPOSED so that no operand ever
// needs a layout change:  q^T[he][px] = Wq[he][:] . x[px][:]  (A = Wq rows, B = x rows),  then
// y^T[co][px] = sum_he W2[co][he] q^T[he][px]  (A = W2, B = the q^T accumulators re-used in place: a lane already
// holds, for its pixel column, 8 he values per K-step — in accumulator row order, which is why W2 is stored with
// that permutation).  Each wave owns 32 pixels end to end, no workgroup barrier.
// The x tile goes through a WAVE-PRIVATE LDS image: the MFMA layouts have a lane own 8 (operand) or 4 (result) channels
// of ONE pixel, so direct loads/stores touch 32 cache lines per instruction and use 16-32 bytes of each (x: 8 loads,
// residual re-read: 4*CT, y: 4*CT stores — ~770 line touches per tile for 128 lines of data, and this kernel is bound by
// the CU's address/line rate).  Through LDS every global instruction moves 1 KB of contiguous pixels (8 lines), the
// residual comes from the image, and y is staged in place over it.
// grid (ceil(npix/128), B), 256 threads, dynamic LDS 4 * 32 * (C + 4) floats.
template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, C == 64 ? 4 : 2))) void linattn_out2_kernel(const LinOut2P p) {
    constexpr int KS1 = C / 16, CT = C / 32;
    constexpr int LDX = C + 4;                          // floats: row stride 4 mod 64 dwords -> conflict-free b128 rows
    constexpr int NCH = 32 * C / 4 / 64;                // 16-byte chunks of the tile per lane (8 for C = 64)
    extern __shared__ __attribute__((aligned(16))) float smem_o2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int px0 = (blockIdx.x * 4 + wave) * 32;
    if (px0 >= p.npix) return;
    float* xs = smem_o2 + wave * 32 * LDX;
    const float* X = p.X + (long)b * p.xb + p.x_coff;
    // coalesced tile load: chunk c = lane + 64k covers pixel c / (C/4), channels (c % (C/4)) * 4 .. +4
    float4 xin[NCH];
    const bool xlp = p.x_lp != 0;                       // 16-bit X: chunk c = lane + 64k (k < NCH / 2) covers pixel c / (C/8), channels (c % (C/8)) * 8 .. +8
    if (xlp) {
        const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.xb + p.x_coff;
#pragma unroll
        for (int k = 0; k < NCH / 2; ++k) {
            const int c = lane + 64 * k, px = c / (C / 8), ch = (c % (C / 8)) * 8;
            xin[k] = *reinterpret_cast<const float4*>(Xh + (long)min(px0 + px, p.npix - 1) * p.ldx + ch);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = lane + 64 * k, px = c / (C / 4), ch = (c % (C / 4)) * 4;
            xin[k] = *reinterpret_cast<const float4*>(X + (long)min(px0 + px, p.npix - 1) * p.ldx + ch);
        }
    }
    // A fragments of GEMM2 (all of W2 for this utterance: CT x 8 K-steps)
    const uint4* w2 = reinterpret_cast<const uint4*>(p.W2) + (long)b * CT * 8 * 64 + lane;
    uint4 w2f[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w2f[ct][ks] = w2[(ct * 8 + ks) * 64];
    if (xlp) {
#pragma unroll
        for (int k = 0; k < NCH / 2; ++k) {
            const int c = lane + 64 * k, px = c / (C / 8), ch = (c % (C / 8)) * 8;
            const unsigned u0 = __float_as_uint(xin[k].x), u1 = __float_as_uint(xin[k].y), u2 = __float_as_uint(xin[k].z), u3 = __float_as_uint(xin[k].w);
            *reinterpret_cast<float4*>(xs + px * LDX + ch) = make_float4(lp_lo(u0), lp_hi(u0), lp_lo(u1), lp_hi(u1));
            *reinterpret_cast<float4*>(xs + px * LDX + ch + 4) = make_float4(lp_lo(u2), lp_hi(u2), lp_lo(u3), lp_hi(u3));
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = lane + 64 * k, px = c / (C / 4), ch = (c % (C / 4)) * 4;
            *reinterpret_cast<float4*>(xs + px * LDX + ch) = xin[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // B fragments of GEMM1: lane (pixel column i, half hh) holds x[px][ks*16 + hh*8 .. +8]
    LFrag xf[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const float4 xa = *reinterpret_cast<const float4*>(xs + i * LDX + ks * 16 + hh * 8);
        const float4 xc = *reinterpret_cast<const float4*>(xs + i * LDX + ks * 16 + hh * 8 + 4);
        xf[ks].u.x = pack2_lp(xa.x, xa.y); xf[ks].u.y = pack2_lp(xa.z, xa.w);
        xf[ks].u.z = pack2_lp(xc.x, xc.y); xf[ks].u.w = pack2_lp(xc.z, xc.w);
    }
    // ---- GEMM1: q^T, four 32-row he tiles; converted to bf16 B fragments tile by tile
    const uint4* wq = reinterpret_cast<const uint4*>(p.Wq) + lane;                  // bf16, fragment order [he tile][K-step][lane][8]
    LFrag qf[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 q;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            LFrag af; af.u = wq[(t * KS1 + ks) * 64];
            q = DEX_MFMA_LP(af.v, xf[ks].v, q, 0, 0, 0);
#ifdef DEX_LP_WSPLIT
            LFrag al; al.u = wq[(t * KS1 + ks) * 64 + p.wq_lo_off / 8];       // the lo half of the same fragment
            q = DEX_MFMA_LP(al.v, xf[ks].v, q, 0, 0, 0);
#endif
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            qf[t * 2 + k2].u.x = pack2_lp(q[8 * k2 + 0], q[8 * k2 + 1]); qf[t * 2 + k2].u.y = pack2_lp(q[8 * k2 + 2], q[8 * k2 + 3]);
            qf[t * 2 + k2].u.z = pack2_lp(q[8 * k2 + 4], q[8 * k2 + 5]); qf[t * 2 + k2].u.w = pack2_lp(q[8 * k2 + 6], q[8 * k2 + 7]);
        }
    }
    // ---- GEMM2 + epilogue: rows co = ct*32 + (r&3) + 8*(r>>2) + 4*hh  ->  4 consecutive channels per register quad;
    // y = result + x (from the image) + bias replaces x in the image, location by location
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            LFrag af; af.u = w2f[ct][ks];
            y = DEX_MFMA_LP(af.v, qf[ks].v, y, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = ct * 32 + 8 * g4 + 4 * hh;
            float* slot = xs + i * LDX + co;
            const float4 res = *reinterpret_cast<const float4*>(slot);
            const float4 bia = *reinterpret_cast<const float4*>(p.bias + co);
            float4 o;
            o.x = y[g4 * 4 + 0] + res.x + bia.x; o.y = y[g4 * 4 + 1] + res.y + bia.y;
            o.z = y[g4 * 4 + 2] + res.z + bia.z; o.w = y[g4 * 4 + 3] + res.w + bia.w;
            *reinterpret_cast<float4*>(slot) = o;
        }
    }
    __builtin_amdgcn_wave_barrier();
    float* Y = p.Y + (long)b * p.yb + p.y_coff;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = lane + 64 * k, px = c / (C / 4), ch = (c % (C / 4)) * 4;
        const float4 o = *reinterpret_cast<const float4*>(xs + px * LDX + ch);
        if (px0 + px < p.npix) {
            if (p.y_lp) {     // every consumer of this tensor rounds it to the operand type: store it that way (half the bytes)
                u16* yh = reinterpret_cast<u16*>(p.Y) + (long)b * p.yb + p.y_coff + (long)(px0 + px) * p.ldy + ch;
                *reinterpret_cast<uint2*>(yh) = make_uint2(pack2_lp(o.x, o.y), pack2_lp(o.z, o.w));
            } else *reinterpret_cast<float4*>(Y + (long)(px0 + px) * p.ldy + ch) = o;
            if (p.Y2) {       // (uniform) a 16-bit copy for the reader that rounds it anyway (the up path's conv reads the concatenation buffer)
                u16* y2 = reinterpret_cast<u16*>(p.Y2) + (long)b * p.y2b + p.y2_coff + (long)(px0 + px) * p.ldy2 + ch;
                *reinterpret_cast<uint2*>(y2) = make_uint2(pack2_lp(o.x, o.y), pack2_lp(o.z, o.w));
            }
        }
    }
}
bool linattn_fused_supported(int C) {
#if defined(DEX_LP_WSPLIT) && !defined(DEX_WS_HAVE_LINATTN)
    return false;            // no split-weight form yet (lp_config.h)
#endif
    return C == 64 || C == 128;
}
// the throughput form below takes grids of at least this many workgroups (DEX_OUT2_MIN)
static long out2_min_wgs() { return knob_or("DEX_OUT2_MIN", 256); }
bool linattn_out2_lp_out_supported(int npix, int B) { return (long)((npix + 127) / 128) * B >= out2_min_wgs(); }       // == the throughput form below
void launch_linattn_out2(const LinOut2P& p, hipStream_t st) {
    dim3 grid((p.npix + 127) / 128, p.B);
    const size_t lds = (size_t)4 * 32 * (p.C + 4) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_out2_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)4 * 32 * 132 * sizeof(float)));
        attr = true;
    }
    if ((long)grid.x * p.B < out2_min_wgs()) {        // latency regime: the direct form
        if (p.C == 64) hipLaunchKernelGGL(linattn_out2_direct_kernel<64>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(linattn_out2_direct_kernel<128>, grid, dim3(256), 0, st, p);
        return;
    }
    if (p.C == 64) hipLaunchKernelGGL(linattn_out2_kernel<64>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(linattn_out2_kernel<128>, grid, dim3(256), lds, st, p);
}

}  // namespace DEX_LP_NS
}  // namespace dex
