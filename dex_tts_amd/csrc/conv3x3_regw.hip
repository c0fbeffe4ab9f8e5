// conv3x3_regw.hip — 3x3 / stride 1 / pad 1 Block convolution (diffusion.py:41-50) with the WEIGHTS RESIDENT IN REGISTERS
// (reduced-precision MFMA modes, batch regime, the 128-channel layers of the half-resolution stage).
//
// The patch-staged kernel (conv3x3_bf16.hip) streams the per-tap weight slices of a 128-channel layer through LDS for every
// 128-256 output pixels and feeds BOTH MFMA operands from LDS: 1.25 ds_read_b128 per MFMA.  A wave gets one ds_read_b128
// back per ~31 cycles whatever the layout (tools/ldsbench: 33 B/clk per wave, scaling with the number of waves, not with the
// batch depth), i.e. about one per MFMA: those kernels are bound by the operand fetch, 24 % of the MFMA peak at B = 32.
// Here a workgroup keeps the whole weight matrix in registers for its lifetime and only the pixels come from LDS:
//   * four waves, one per SIMD (the weights need the whole register file): wave (cp, kh) owns the output-channel PAIR of
//     tiles cp (2 x 32 channels) x the K half kh (input channels 64*kh..+64 of every tap) as the MFMA A operand - 288
//     registers - and walks a strip of 64 output columns down the image; every pixel fragment fetched from LDS feeds TWO
//     MFMAs (0.5 ds_read_b128 per MFMA: 288 KB per tile against 4.8k cycles' worth at 1:1, measured in tools/rwbench);
//   * input rows, already transformed (x * mask, or the fused producer tail mask * (Mish(GroupNorm(x)) + time bias) [+ res],
//     diffusion.py:49,67-71), live in a four-slot LDS ring as 16-bit rows; each input row is fetched and transformed ONCE per
//     strip, one row ahead of its use, and is the B operand of all four waves (transposed product: C rows = output
//     channels, columns = pixels);
//   * no weight traffic and no barrier inside the 144-MFMA chain of a tile; the two K halves of a channel pair swap one pixel
//     tile's partial sums through LDS (each wave finishes 64 channels x 32 pixels), two workgroup barriers per tile;
//   * outputs leave through an LDS stage (lane = pixel: four consecutive channels per LDS write) as 16 B per lane, a
//     contiguous run per row segment; GroupNorm partial statistics accumulate per lane over the whole strip and are reduced
//     once per workgroup (fixed order inside a wave, integer fixed-point adds across waves / workgroups: deterministic).
#define DEX_NO_PK            // single-value prologue arithmetic (bf16_util.h); the file is also built with -fno-slp-vectorize (build.py)
#include "kernels.h"
#include <cstdlib>
#include "lp_util.h"
#include "kernels_lp.h"
#include "conv_gn.h"

namespace dex {
namespace DEX_LP_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

namespace {

#ifndef RW_PAD
#define RW_PAD 16
#endif
#ifndef RW_VALU_PER_MFMA
#define RW_VALU_PER_MFMA 12
#endif

template <int CIN, int COUT, bool YB>
struct RwGeom {
    static constexpr int NTHR = 256, MPX = 64, PC = MPX + 2;
    static constexpr int PXB = CIN * 2 + RW_PAD, ROWB = PC * PXB, RING = 4 * ROWB;        // +16 B: conflict-free 16 B reads at the lane stride
    static constexpr int SPB = COUT * (YB ? 2 : 4) + 16, STG = MPX * SPB;
    static constexpr int KSH = CIN / 32, KST = 9 * KSH;                                 // K steps of 16 per tap / per tile of ONE K half
    static constexpr int NCT = COUT / 32;                                               // channel tiles: two per wave
    static constexpr int EXCH = 4 * 8192;                                               // per wave: two 32 x 32 fp32 partial tiles
    static constexpr int CH = CIN / 8, NL = (PC * CH + NTHR - 1) / NTHR;                // 16 B ring chunks per pixel / per thread per row
    static constexpr int OCH = COUT * (YB ? 2 : 4) / 16, NS = MPX * OCH / NTHR;         // 16 B output chunks per pixel / per thread per tile
    static constexpr int TAIL = 3 * CIN * 4 + COUT * 4 + 16 * 8;                        // coefficient table, bias, statistics
    static constexpr int LDS = RING + STG + EXCH + TAIL + 16;                          // + a dummy slot for ring writes of lanes without a column
    static_assert(NCT == 4 && NTHR % CH == 0 && NS >= 1 && KSH == 4, "geometry");
};

// The MFMA as inline assembly: the weight operand is REQUIRED in an accumulation register ("a").  Left to the register
// allocator the 288 weight registers of a wave are arch VGPRs spilled to AGPRs, copied back (4 x v_accvgpr_read_b32 into one
// reused quad + a hazard nop) in front of every MFMA: ~50 cycles per MFMA instead of 27.  With the constraint they LIVE in the
// 256 AGPRs (the MFMA reads srcA from either file); the last KV K-steps' worth stays in arch VGPRs.  The compiler does not see
// an MFMA here: rw_mfma_fence() below supplies the wait states between the last MFMA of a chain and the first read of its result.
typedef unsigned rw_u32x4 __attribute__((ext_vector_type(4)));
#ifdef DEX_LP_F16
#define RW_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define RW_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
#ifndef RW_ACC_C
#define RW_ACC_C "+a"          // accumulators in the accumulation file: the C read / D write of an MFMA stay off the arch-VGPR ports
#endif
#define RW_ACC_D "=a"
#ifndef RW_KA
#define RW_KA 24
#endif
__device__ __forceinline__ void rw_mfma_a(f32x16& acc, const rw_u32x4& w, const lp8& x) { asm volatile(RW_MFMA_OP " %0, %1, %2, %0" : RW_ACC_C(acc) : "a"(w), "v"(x)); }
__device__ __forceinline__ void rw_mfma_v(f32x16& acc, const rw_u32x4& w, const lp8& x) { asm volatile(RW_MFMA_OP " %0, %1, %2, %0" : RW_ACC_C(acc) : "v"(w), "v"(x)); }
// first MFMA of an accumulator: C = 0 (inline constant), no initialisation of the 16 registers
__device__ __forceinline__ void rw_mfma_a0(f32x16& acc, const rw_u32x4& w, const lp8& x) { asm volatile(RW_MFMA_OP " %0, %1, %2, 0" : RW_ACC_D(acc) : "a"(w), "v"(x)); }
__device__ __forceinline__ void rw_mfma_v0(f32x16& acc, const rw_u32x4& w, const lp8& x) { asm volatile(RW_MFMA_OP " %0, %1, %2, 0" : RW_ACC_D(acc) : "v"(w), "v"(x)); }
__device__ __forceinline__ void rw_mfma_fence(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile("s_nop 15\n\ts_nop 15" : RW_ACC_C(a), RW_ACC_C(b), RW_ACC_C(c), RW_ACC_C(d));
}

// (a VALU write needs wait states before an MFMA reads the register: the accumulators' initial values)
__device__ __forceinline__ void rw_mfma_init_fence(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile("s_nop 4" : RW_ACC_C(a), RW_ACC_C(b), RW_ACC_C(c), RW_ACC_C(d));
}

// registers of one input row in flight
template <int NL, bool XB, bool RES2>
struct RwRow { uint4 a[NL]; uint4 b[XB ? 1 : NL]; float4 r0[RES2 ? NL : 1], r1[RES2 ? NL : 1]; float m[NL]; };

}  // namespace

// PROF: 0 plain x * mask, 1 producer tail mask * (Mish(GN(x)) + tadd) (pro_stats), 2 resnet tail x' = mask * Mish(GN(x)) + pro_res
// (x' also written to pro_xout for this workgroup's pixels), then x' * mask.
// XOL: x' (PROF == 2) leaves in the mode's 16-bit type (Conv3P::xout_lp).
template <int CIN, int COUT, int PROF, bool XB, bool YB, bool XOL = false>
__global__ __launch_bounds__(256) void conv3x3_rw_kernel(const Conv3P p, const int nseg, const int rows_per_wg) {
    using G = RwGeom<CIN, COUT, YB>;
    using Row = RwRow<G::NL, XB, false>;             // (the residual of the PROF == 2 form travels separately: res_ring below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rw[];
    unsigned char* ring = smem_rw;
    unsigned char* stage = smem_rw + G::RING;
    unsigned char* exch = smem_rw + G::RING + G::STG;                          // [4 waves][8][64 lanes] x 16 B
    float* coef = reinterpret_cast<float*>(smem_rw + G::RING + G::STG + G::EXCH);   // [3][CIN]
    float* bs = coef + 3 * CIN;                                                 // [COUT]
    long long* gnred = reinterpret_cast<long long*>(bs + COUT);                 // [8][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int cp = wave & 1, kh2 = wave >> 1;                                   // this wave's channel pair (tiles 2cp, 2cp+1) and K half; it FINISHES pixel tile kh2
    const int seg = blockIdx.x % nseg, chunk = blockIdx.x / nseg, b = blockIdx.y;
    const int iw0 = seg * G::MPX;
    const int r0 = chunk * rows_per_wg, r1 = min(p.H, r0 + rows_per_wg);
    const bool full_strip = iw0 + G::MPX <= p.W;                                // workgroup-uniform
    const int step = p.step;
    const float* X = p.X + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const int c8 = (tid % G::CH) * 8, px0 = tid / G::CH;                        // this thread's channel chunk; ring column of its item 0

    // Per-thread constants of the row loads: element offset of item j inside an image row (column clamped into the image), and
    // its column mask (0 outside the image / past the strip's halo).  A row load is then a wave-uniform row base + these offsets:
    // no per-row address arithmetic or mask loads in the MFMA loop.
    int coff[G::NL]; float cmask[G::NL];
#pragma unroll
    for (int j = 0; j < G::NL; ++j) {
        const int px = px0 + j * (G::NTHR / G::CH);
        const int wi = iw0 - 1 + px;
        const bool ok = px < G::PC && (unsigned)wi < (unsigned)p.W;
        const int wc = ok ? wi : 0;
        coff[j] = wc * p.ldx + c8;
        cmask[j] = ok ? mrow[wc * p.mask_ws] : 0.f;
    }
    auto row_load = [&](int row, Row& R) __attribute__((always_inline)) {
        const bool rok = (unsigned)row < (unsigned)p.H;
        const int rc = __builtin_amdgcn_readfirstlane(rok ? row : 0);
        const u16* xh = Xh + (long)rc * p.W * p.ldx;
        const float* xf = X + (long)rc * p.W * p.ldx;
#pragma unroll
        for (int j = 0; j < G::NL; ++j) {
            if constexpr (XB) R.a[j] = *reinterpret_cast<const uint4*>(xh + coff[j]);
            else { R.a[j] = *reinterpret_cast<const uint4*>(xf + coff[j]); R.b[j] = *reinterpret_cast<const uint4*>(xf + coff[j] + 4); }
            R.m[j] = rok ? cmask[j] : 0.f;
        }
    };
    // Transform + ring write of a loaded row (its ring slot is (row + 1) & 3), in SLICES of four values: slice s = item s / 2,
    // channels c8 + 4 * (s & 1) .. +4.  In the row loop one slice rides in each MFMA batch (~30 plain VALU instructions
    // among 8 MFMAs: inside the ~5 issue slots an MFMA leaves free).
    struct Coef { float4 sc, sh, t; };
    auto coef_load = [&](int hq) __attribute__((always_inline)) {
        Coef c{};
        if constexpr (PROF != 0) {
            c.sc = *reinterpret_cast<const float4*>(coef + c8 + 4 * hq);
            c.sh = *reinterpret_cast<const float4*>(coef + CIN + c8 + 4 * hq);
            c.t = *reinterpret_cast<const float4*>(coef + 2 * CIN + c8 + 4 * hq);
        }
        return c;
    };
    auto f4 = [](const float4& v, int q) __attribute__((always_inline)) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };
    struct Slice { uint2 v; float4 f; };
    // the arithmetic of one value in two steps (each rides behind ONE MFMA in the row loop): t = GN(x), e = exp(min(t, 20)) | y
    struct Half { float t, e; };
    auto val_a = [&](float x, float sc, float sh) __attribute__((always_inline)) {
        Half hv;
#ifndef RW_NO_TRANSFORM          // (tools/rwbench experiments)
        if constexpr (PROF != 0) { hv.t = fmaf(x, sc, sh); hv.e = __builtin_amdgcn_exp2f(fminf(hv.t * 1.44269504088896f, 28.8539008f)); }
        else
#endif
        { hv.t = x; hv.e = 0.f; }
        return hv;
    };
    auto val_b = [&](const Half& hv, float add, float mk, float res) __attribute__((always_inline)) {
        float y = hv.t;
#ifndef RW_NO_TRANSFORM
        if constexpr (PROF != 0) { const float n = hv.e * (hv.e + 2.f); y = fmaf(hv.t, n * __builtin_amdgcn_rcpf(n + 2.f), add); }   // Mish(t) + time bias (mish1_add, bf16_util.h)
#endif
        if constexpr (PROF == 2) y = fmaf(y, mk, res);
        return y;
    };
    auto slice_x = [&](const Row& R, int s_, int q) __attribute__((always_inline)) {      // raw value q of slice s_
        const int j = s_ >> 1, hq = s_ & 1;
        if constexpr (XB) {
            const unsigned u = (q < 2) ? (hq ? R.a[j].z : R.a[j].x) : (hq ? R.a[j].w : R.a[j].y);
            return (q & 1) ? lp_hi(u) : lp_lo(u);
        } else {
            const uint4 u = hq ? R.b[j] : R.a[j];
            return __uint_as_float(q == 0 ? u.x : q == 1 ? u.y : q == 2 ? u.z : u.w);
        }
    };
    // PROF == 2: the fp32 residual of slice s_ of a row, four values (ldres == CIN == ldx: the row-load offsets).  In the row loop
    // it is fetched RES_AHEAD batches before its slice into a small register ring - a whole row in flight would be 40 registers.
    auto res_load = [&](int row, int s_) __attribute__((always_inline)) {
        const int rc = __builtin_amdgcn_readfirstlane((unsigned)row < (unsigned)p.H ? row : 0);
        return *reinterpret_cast<const float4*>(p.pro_res + ((long)b * p.H + rc) * p.W * CIN + coff[s_ >> 1] + 4 * (s_ & 1));
    };
    auto row_calc = [&](const Row& R, int s_, const Coef& cf, const float4& res) __attribute__((always_inline)) {      // whole slice at once (prologue rows)
        const float mk = R.m[s_ >> 1];
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = val_b(val_a(slice_x(R, s_, q), f4(cf.sc, q), f4(cf.sh, q)), f4(cf.t, q), mk, f4(res, q));
        return Slice{make_uint2(pack2_lp(y[0] * mk, y[1] * mk), pack2_lp(y[2] * mk, y[3] * mk)), make_float4(y[0], y[1], y[2], y[3])};
    };
    // The stores of a slice are BRANCH-FREE.  Behind a per-lane branch the x' store, whose data depends on the residual load,
    // gets an s_waitcnt vmcnt(0) of its own: every slice then waits for the loads issued ahead of it (the residual ring, the
    // next input row) - the PROF == 2 row loop ran 4.4k cycles per row longer than the PROF == 1 one.  Lanes without a ring
    // column write a dummy LDS slot; lanes that do not own their pixel (halo columns / rows, image border) give the x' store an
    // out-of-range offset, which a raw buffer store drops.
    const __amdgpu_buffer_rsrc_t xo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(PROF == 2 ? (XOL ? reinterpret_cast<const float*>(reinterpret_cast<const u16*>(p.pro_xout) + (long)b * p.H * p.W * CIN) : p.pro_xout + (long)b * p.H * p.W * CIN) : p.bias),
        0, PROF == 2 ? (unsigned)((long)p.H * p.W * CIN * (XOL ? 2 : 4)) : 0u, 0x00020000);
    int xoff[G::NL];                     // byte offset of item j inside an image row of x', or "out of range" for columns this thread does not own
#pragma unroll
    for (int j = 0; j < G::NL; ++j) {
        const int px = px0 + j * (G::NTHR / G::CH), wi = iw0 - 1 + px;
        xoff[j] = (px >= 1 && px <= G::MPX && wi < p.W) ? (wi * CIN + c8) * 4 : 0x40000000;
    }
    auto row_put = [&](int row, int s_, const Slice& sl) __attribute__((always_inline)) {
        const int j = s_ >> 1, hq = s_ & 1;
        const int px = px0 + j * (G::NTHR / G::CH);
        const int slot_off = ((row + 1) & 3) * G::ROWB + px * G::PXB + c8 * 2 + hq * 8;
        *reinterpret_cast<uint2*>(ring + ((j < G::NL - 1 || px < G::PC) ? slot_off : G::RING + G::STG + G::EXCH + G::TAIL)) = sl.v;
        if constexpr (PROF == 2) {
            const bool own_row = row >= r0 && row < r1;                    // (uniform)
            if constexpr (XOL) {
                typedef unsigned rw_u32x2 __attribute__((ext_vector_type(2)));
                const unsigned off = own_row ? (unsigned)(row * (p.W * CIN * 2) + (xoff[j] >> 1) + 8 * hq) : 0xffffffffu;
                const rw_u32x2 v = {pack2_lp(sl.f.x, sl.f.y), pack2_lp(sl.f.z, sl.f.w)};
                __builtin_amdgcn_raw_buffer_store_b64(v, xo_rsrc, off, 0, 0);
            } else {
                const unsigned off = own_row ? (unsigned)(row * (p.W * CIN * 4) + xoff[j] + 16 * hq) : 0xffffffffu;
                const rw_u32x4 v = {__float_as_uint(sl.f.x), __float_as_uint(sl.f.y), __float_as_uint(sl.f.z), __float_as_uint(sl.f.w)};
                __builtin_amdgcn_raw_buffer_store_b128(v, xo_rsrc, off, 0, 0);
            }
        }
    };
    auto row_store = [&](int row, const Row& R) __attribute__((always_inline)) {
        float4 res[PROF == 2 ? 2 * G::NL : 1];
        res[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PROF == 2) {
#pragma unroll
            for (int s_ = 0; s_ < 2 * G::NL; ++s_) res[s_] = res_load(row, s_);
        }
#pragma unroll
        for (int s_ = 0; s_ < 2 * G::NL; ++s_) row_put(row, s_, row_calc(R, s_, coef_load(s_ & 1), res[PROF == 2 ? s_ : 0]));
    };

#ifdef DEX_TIMING
    const long long tk_entry = __builtin_readcyclecounter();
#endif
    // ---- prologue: statistics + this wave's weights + the first input rows, everything in flight together
    CvGnLoads gnl{};
    if constexpr (PROF != 0) gnl = cv_gn_issue(p, b, tid, step);
    Row Ra, Rb;
    row_load(r0 - 1, Ra);
    row_load(r0, Rb);
    constexpr int KA = RW_KA;                                // K steps (of KST = 36) whose weights live in AGPRs (2 x KA x 4 registers, beside the 64 of the accumulators)
    rw_u32x4 wa[2][KA], wv[2][G::KST - KA];
    {
        // [ct][tap * Cin/16 + ks][lane] x 16 B; this wave: K steps kh2*KSH .. +KSH of every tap
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag) + ((long)(2 * cp) * 9 * (CIN / 16) + kh2 * G::KSH) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ks = 0; ks < G::KST; ++ks) {
                const uint4 w4 = Wf[(c * 9 * (CIN / 16) + (ks / G::KSH) * (CIN / 16) + ks % G::KSH) * 64];
                const rw_u32x4 wn = {w4.x, w4.y, w4.z, w4.w};
                if (ks < KA) wa[c][ks] = wn; else wv[c][ks - KA] = wn;
            }
    }
    if (tid < COUT) bs[tid] = p.bias[tid];
    if (tid < 16) gnred[tid] = 0;
    if constexpr (PROF != 0) {
        cv_gn_finish<CIN>(p, gnl, tid, reinterpret_cast<float (*)[CIN]>(coef));
        lds_barrier();
    }
    row_store(r0 - 1, Ra);
    row_store(r0, Rb);
    row_load(r0 + 1, Ra);
    row_load(r0 + 2 <= r1 ? r0 + 2 : -1, Rb);
    row_store(r0 + 1, Ra);
    lds_barrier();

    constexpr int cpg = COUT / 8, NG = 64 / cpg;                 // GroupNorm groups inside this wave's 64 channels
    float gs[NG], gq[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { gs[g] = 0.f; gq[g] = 0.f; }

#ifdef DEX_TIMING
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk0 = __builtin_readcyclecounter();
    long long tlast = tk0;
#define RSTAMP(k) do { const long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define RSTAMP(k) do {} while (0)
#endif
    constexpr int RES_AHEAD = 3, RES_RING = 4;
    float4 res_ring[PROF == 2 ? RES_RING : 1];
    if constexpr (PROF == 2) {
#pragma unroll
        for (int s_ = 0; s_ < RES_AHEAD; ++s_) res_ring[s_] = res_load(r0 + 2 <= r1 ? r0 + 2 : -1, s_);
    }
    for (int ih = r0; ih < r1; ++ih) {
        // ---- MFMA chain: acc[c][j] (channel tile c of the pair x pixel tile j ^ kh2, 32 x 32 each; j = 0 is the tile this wave
        // finishes) = this wave's K half of the 9 taps; the bias rides on the tiles this wave finishes.  The pixel operands of a batch of HS K-steps are read from the ring
        // one batch AHEAD of their MFMAs (one wave per SIMD: nobody else hides the LDS latency).  Under the first 2 NL batches:
        // input row ih + 2 (in flight since the previous tile) is transformed slice by slice into the free ring slot, then the
        // loads of row ih + 3 go out.
        f32x16 acc[2][2];
        constexpr int HS = 2, HPT = G::KSH / HS, NH = 9 * HPT;            // K steps per batch, batches per tap / per tile
        static_assert(NH > 2 * G::NL, "the row transform rides under the first batches");
        lp8 xq[2][HS * 2];
#ifdef RW_NO_XLOAD
#define RW_XREAD(dst_, ptr_) do { dst_ = __builtin_bit_cast(lp8, wv[0][0]); } while (0)
#else
#define RW_XREAD(dst_, ptr_) do { dst_ = *reinterpret_cast<const lp8*>(ptr_); } while (0)
#endif
#define RW_XLOAD(h_, buf_) do { \
            const int tap_ = (h_) / HPT, kh_ = tap_ / 3, kw_ = tap_ - kh_ * 3; \
            const unsigned char* xr_ = ring + ((ih + kh_) & 3) * G::ROWB + (i + kw_) * G::PXB + hh * 16 + kh2 * (G::KSH * 32) + ((h_) % HPT) * (HS * 32); \
            _Pragma("unroll") for (int ks_ = 0; ks_ < HS; ++ks_) \
                _Pragma("unroll") for (int pt_ = 0; pt_ < 2; ++pt_) \
                    RW_XREAD(xq[buf_][ks_ * 2 + pt_], xr_ + (pt_ ? pto1 : pto0) + ks_ * 32); \
        } while (0)
        // MFMA m (0..7) of batch h_: K step m >> 2, pixel tile (m >> 1) & 1, channel tile m & 1; the first of an accumulator has C = 0
#define RW_MF(h_, m_) do { \
            constexpr int ks_ = (m_) >> 2, pt_ = ((m_) >> 1) & 1, c_ = (m_) & 1; const int kk_ = (h_) * HS + ks_; \
            if (kk_ == 0) { if (kk_ < KA) rw_mfma_a0(acc[c_][pt_], wa[c_][kk_ < KA ? kk_ : 0], xq[(h_) & 1][ks_ * 2 + pt_]); else rw_mfma_v0(acc[c_][pt_], wv[c_][kk_ >= KA ? kk_ - KA : 0], xq[(h_) & 1][ks_ * 2 + pt_]); } \
            else if (kk_ < KA) rw_mfma_a(acc[c_][pt_], wa[c_][kk_ < KA ? kk_ : 0], xq[(h_) & 1][ks_ * 2 + pt_]); \
            else rw_mfma_v(acc[c_][pt_], wv[c_][kk_ >= KA ? kk_ - KA : 0], xq[(h_) & 1][ks_ * 2 + pt_]); \
        } while (0)
        const int pto0 = kh2 * 32 * G::PXB, pto1 = (kh2 ^ 1) * 32 * G::PXB;       // ring offsets of pixel tiles j = 0, 1
        RW_XLOAD(0, 0);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (h + 1 < NH) RW_XLOAD(h + 1, (h + 1) & 1);
            const bool sl = h < 2 * G::NL;                        // this batch carries slice h of input row ih + 2
            Coef cf{};
            if (sl) cf = coef_load(h & 1);
            const float mk = sl ? Rb.m[(h < 2 * G::NL ? h : 0) >> 1] : 0.f;
            float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PROF == 2) {
                if (sl) res = res_ring[h % RES_RING];
                // fetch ahead: this tile's slice h + RES_AHEAD, or - in the last batches - the first slices of the next tile's row
                if (h + RES_AHEAD < 2 * G::NL) res_ring[(h + RES_AHEAD) % RES_RING] = res_load(ih + 2, h + RES_AHEAD);
                else if (h >= NH - RES_AHEAD) res_ring[(h - (NH - RES_AHEAD)) % RES_RING] = res_load(ih + 3 <= r1 ? ih + 3 : -1, h - (NH - RES_AHEAD));
            }
            __builtin_amdgcn_sched_barrier(0);
            // eight MFMAs; behind MFMA 2q / 2q+1 the first / second half of value q's arithmetic (~6 plain VALU instructions each:
            // an MFMA leaves ~5 issue slots free, and the compiler left to itself clumps 30 of them into one gap)
            float y[4];
            Half hv;
#define RW_PAIR(q_) do { \
                RW_MF(h, 2 * (q_)); \
                if (sl) hv = val_a(slice_x(Rb, h < 2 * G::NL ? h : 0, q_), f4(cf.sc, q_), f4(cf.sh, q_)); \
                __builtin_amdgcn_sched_barrier(0); \
                RW_MF(h, 2 * (q_) + 1); \
                if (sl) y[q_] = val_b(hv, f4(cf.t, q_), mk, f4(res, q_)); \
                __builtin_amdgcn_sched_barrier(0); \
            } while (0)
            RW_PAIR(0); RW_PAIR(1); RW_PAIR(2); RW_PAIR(3);
#undef RW_PAIR
            if (sl) row_put(ih + 2, h, Slice{make_uint2(pack2_lp(y[0] * mk, y[1] * mk), pack2_lp(y[2] * mk, y[3] * mk)), make_float4(y[0], y[1], y[2], y[3])});
#ifndef RW_NO_ROWLOAD
            if (h == 2 * G::NL) row_load(ih + 3 <= r1 ? ih + 3 : -1, Rb);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#undef RW_XLOAD
#undef RW_MF
        rw_mfma_fence(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        // ---- the partner wave (same channel pair, other K half) finishes the OTHER pixel tile: hand it this wave's partial sums
        {
            float4* ex = reinterpret_cast<float4*>(exch) + (wave * 8) * 64 + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) ex[(c * 4 + g) * 64] = make_float4(acc[c][1][4 * g], acc[c][1][4 * g + 1], acc[c][1][4 * g + 2], acc[c][1][4 * g + 3]);
        }
        RSTAMP(0);
        RSTAMP(1);
        lds_barrier();              // partial sums visible; every wave is past the previous tile's reads of the output stage
        RSTAMP(2);
        {
            const float4* ex = reinterpret_cast<const float4*>(exch) + ((wave ^ 2) * 8) * 64 + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = ex[(c * 4 + g) * 64];
                    const float4 b4 = *reinterpret_cast<const float4*>(bs + (2 * cp + c) * 32 + 8 * g + 4 * hh);
                    acc[c][0][4 * g] = (acc[c][0][4 * g] + o.x) + b4.x; acc[c][0][4 * g + 1] = (acc[c][0][4 * g + 1] + o.y) + b4.y;
                    acc[c][0][4 * g + 2] = (acc[c][0][4 * g + 2] + o.z) + b4.z; acc[c][0][4 * g + 3] = (acc[c][0][4 * g + 3] + o.w) + b4.w;
                }
        }
        // ---- statistics of the raw output (lane = pixel: columns past the image edge do not count), then the output stage
        {
            const bool live = iw0 + kh2 * 32 + i < p.W;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = live ? acc[c][0][r] : 0.f;
                    const int g = (c * 32 + 8 * (r >> 2)) / cpg;
                    gs[g] += v; gq[g] = fmaf(v, v, gq[g]);
                }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* d = stage + (kh2 * 32 + i) * G::SPB + ((2 * cp + c) * 32 + 8 * g + 4 * hh) * (YB ? 2 : 4);
                if constexpr (YB) *reinterpret_cast<uint2*>(d) = make_uint2(pack2_lp(acc[c][0][4 * g], acc[c][0][4 * g + 1]), pack2_lp(acc[c][0][4 * g + 2], acc[c][0][4 * g + 3]));
                else *reinterpret_cast<float4*>(d) = make_float4(acc[c][0][4 * g], acc[c][0][4 * g + 1], acc[c][0][4 * g + 2], acc[c][0][4 * g + 3]);
            }
        RSTAMP(3);
        lds_barrier();
        RSTAMP(4);
        // ---- output stage -> HBM, 16 B per lane, consecutive lanes = consecutive bytes of the row segment
        {
            uint4 ov[G::NS];
#pragma unroll
            for (int j = 0; j < G::NS; ++j) {
                const int q = tid + G::NTHR * j;
                ov[j] = *reinterpret_cast<const uint4*>(stage + (q / G::OCH) * G::SPB + (q % G::OCH) * 16);
            }
            const long rowbase = ((long)b * p.H + ih) * p.W * COUT;          // wave-uniform
            // (whole strips store unpredicated: behind a per-lane branch the compiler sinks each LDS read above into its branch and
            // the four read -> store pairs run one LDS latency after the other)
#define RW_OUT(pred_) _Pragma("unroll") for (int j = 0; j < G::NS; ++j) { \
                const int q = tid + G::NTHR * j; \
                const int wo = iw0 + q / G::OCH; \
                const int e = wo * COUT + (q % G::OCH) * (YB ? 8 : 4); \
                if (pred_) { \
                    if constexpr (YB) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.Y) + rowbase + e) = ov[j]; \
                    else *reinterpret_cast<uint4*>(p.Y + rowbase + e) = ov[j]; \
                } \
            }
            if (full_strip) { RW_OUT(true) } else { RW_OUT(wo < p.W) }
#undef RW_OUT
        }
        RSTAMP(5);
    }
#ifdef DEX_TIMING
    if (p.dbg && lane == 0) {
        long long* d = p.dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int k = 0; k < 6; ++k) d[k] = tk[k];
        d[6] = tk0 - tk_entry; d[7] = __builtin_readcyclecounter() - tk0;
    }
#endif
    if (p.gn_stats) {
        const double inv_n = 1.0 / ((double)p.H * p.W * cpg);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float a_ = gs[g], q_ = gq[g];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { a_ += __shfl_xor(a_, o); q_ += __shfl_xor(q_, o); }
            if (lane == 0) { gn_add(&gnred[((cp * 64) / cpg + g) * 2], gn_fix(a_, inv_n)); gn_add(&gnred[((cp * 64) / cpg + g) * 2 + 1], gn_fix(q_, inv_n)); }
        }
        __syncthreads();
        if (tid < 16) {
            const long long v = gnred[tid];
            if (v != 0) gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + blockIdx.x % GN_SLOTS) * 2 + (tid & 1), v);
        }
    }
}

// ---- 64 -> 128 channels with the fused 1x1 shortcut (the first conv of the half-resolution stage: conv3x3(x * mask) and
// res_conv(x * mask) of the Downsample output, diffusion.py:66-71).  2.9 KB of HBM traffic per output pixel (16-bit h1, fp32
// shortcut) against 83 kFLOP: HBM-bound (the patch kernel moved it at 2.4 TB/s, 120 us at B = 32).  Same strip walker, the plain
// way round: wave ct owns output-channel tile ct of both outputs with its full-K weights in registers (36 + 4 K-steps,
// 160 VGPRs, compiler-visible MFMAs), both 32-pixel tiles of the 64-column strip; no K split, no exchange; the two outputs leave
// through two LDS stages as contiguous runs.
template <bool XB>
__global__ __launch_bounds__(256) void conv3x3_rw_res128_kernel(const Conv3P p, const int nseg, const int rows_per_wg) {
    constexpr int CIN = 64, COUT = 128;
    constexpr int MPX = 64, PC = MPX + 2, PXB = CIN * 2 + 16, ROWB = PC * PXB, RING = 4 * ROWB;
    constexpr int SPB = COUT * 2 + 16, STG = MPX * SPB, SPBR = COUT * 4 + 16, STGR = MPX * SPBR;
    constexpr int KSH = CIN / 16, KST = 9 * KSH, CH = CIN / 8, NL = (PC * CH + 255) / 256;
    constexpr int OCH = COUT * 2 / 16, NS = MPX * OCH / 256, OCHR = COUT * 4 / 16, NSR = MPX * OCHR / 256;
    using Row = RwRow<NL, XB, false>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rw[];
    unsigned char* ring = smem_rw;
    unsigned char* stage = smem_rw + RING;
    unsigned char* stager = stage + STG;
    float* bs = reinterpret_cast<float*>(stager + STGR);                        // [COUT] bias, [COUT] shortcut bias
    long long* gnred = reinterpret_cast<long long*>(bs + 2 * COUT);             // [8][2]
    const int DUMMY = RING + STG + STGR + 2 * COUT * 4 + 16 * 8;                // ring writes of lanes without a column
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int seg = blockIdx.x % nseg, chunk = blockIdx.x / nseg, b = blockIdx.y;
    const int iw0 = seg * MPX;
    const int r0 = chunk * rows_per_wg, r1 = min(p.H, r0 + rows_per_wg);
    const bool full_strip = iw0 + MPX <= p.W;
    const float* X = p.X + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const u16* Xh = reinterpret_cast<const u16*>(p.X) + (long)b * p.H * p.W * p.ldx + p.x_coff;
    const float* mrow = p.mask + (long)b * p.mask_bstride;
    const int c8 = (tid % CH) * 8, px0 = tid / CH;
    int coff[NL], roff[NL]; float cmask[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int px = px0 + j * (256 / CH), wi = iw0 - 1 + px;
        const bool ok = px < PC && (unsigned)wi < (unsigned)p.W;
        const int wc = ok ? wi : 0;
        coff[j] = wc * p.ldx + c8;
        cmask[j] = ok ? mrow[wc * p.mask_ws] : 0.f;
        roff[j] = px < PC ? px * PXB + c8 * 2 : -1;
    }
    auto row_load = [&](int row, Row& R) __attribute__((always_inline)) {
        const bool rok = (unsigned)row < (unsigned)p.H;
        const int rc = __builtin_amdgcn_readfirstlane(rok ? row : 0);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if constexpr (XB) R.a[j] = *reinterpret_cast<const uint4*>(Xh + (long)rc * p.W * p.ldx + coff[j]);
            else { const float* xf = X + (long)rc * p.W * p.ldx + coff[j]; R.a[j] = *reinterpret_cast<const uint4*>(xf); R.b[j] = *reinterpret_cast<const uint4*>(xf + 4); }
            R.m[j] = rok ? cmask[j] : 0.f;
        }
    };
    auto row_store = [&](int row, const Row& R) __attribute__((always_inline)) {      // x * mask -> ring slot (row + 1) & 3, branch-free
        const int slot = ((row + 1) & 3) * ROWB;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const float m = R.m[j];
            uint4 v;
            if constexpr (XB) {
                v.x = pack2_lp(lp_lo(R.a[j].x) * m, lp_hi(R.a[j].x) * m); v.y = pack2_lp(lp_lo(R.a[j].y) * m, lp_hi(R.a[j].y) * m);
                v.z = pack2_lp(lp_lo(R.a[j].z) * m, lp_hi(R.a[j].z) * m); v.w = pack2_lp(lp_lo(R.a[j].w) * m, lp_hi(R.a[j].w) * m);
            } else {
                v.x = pack2_lp(__uint_as_float(R.a[j].x) * m, __uint_as_float(R.a[j].y) * m); v.y = pack2_lp(__uint_as_float(R.a[j].z) * m, __uint_as_float(R.a[j].w) * m);
                v.z = pack2_lp(__uint_as_float(R.b[j].x) * m, __uint_as_float(R.b[j].y) * m); v.w = pack2_lp(__uint_as_float(R.b[j].z) * m, __uint_as_float(R.b[j].w) * m);
            }
            *reinterpret_cast<uint4*>(smem_rw + (roff[j] >= 0 ? slot + roff[j] : DUMMY)) = v;
        }
    };

    // ---- prologue
    Row Ra, Rb;
    row_load(r0 - 1, Ra);
    row_load(r0, Rb);
    uint4 wr[KST], wq[KSH];
    {
        const uint4* Wf = reinterpret_cast<const uint4*>(p.Wfrag) + (long)ct * KST * 64 + lane;      // [ct][tap * 4 + ks][lane] x 16 B
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) wr[ks] = Wf[ks * 64];
        const uint4* Rf = reinterpret_cast<const uint4*>(p.res_wfrag) + (long)ct * KSH * 64 + lane;  // [ct][ks][lane]
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) wq[ks] = Rf[ks * 64];
    }
    if (tid < COUT) { bs[tid] = p.bias[tid]; bs[COUT + tid] = p.res_b[tid]; }
    if (tid < 16) gnred[tid] = 0;
    row_store(r0 - 1, Ra);
    row_store(r0, Rb);
    row_load(r0 + 1, Ra);
    row_load(r0 + 2 <= r1 ? r0 + 2 : -1, Rb);
    row_store(r0 + 1, Ra);
    lds_barrier();

    constexpr int cpg = COUT / 8, NG = 32 / cpg;
    float gs[NG], gq[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { gs[g] = 0.f; gq[g] = 0.f; }

    for (int ih = r0; ih < r1; ++ih) {
        // input row ih + 2 (in flight since the previous tile) -> the free ring slot; then the loads of row ih + 3
        row_store(ih + 2, Rb);
        row_load(ih + 3 <= r1 ? ih + 3 : -1, Rb);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2], accr[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[pt][r] = 0.f; accr[pt][r] = 0.f; }
        lp8 xq[2][KSH * 2];
#define RWC_XLOAD(tap_, buf_) do { \
            const int kh_ = (tap_) / 3, kw_ = (tap_) - kh_ * 3; \
            const unsigned char* xr_ = ring + ((ih + kh_) & 3) * ROWB + (i + kw_) * PXB + hh * 16; \
            _Pragma("unroll") for (int ks_ = 0; ks_ < KSH; ++ks_) \
                _Pragma("unroll") for (int pt_ = 0; pt_ < 2; ++pt_) xq[buf_][ks_ * 2 + pt_] = *reinterpret_cast<const lp8*>(xr_ + pt_ * 32 * PXB + ks_ * 32); \
        } while (0)
        RWC_XLOAD(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) RWC_XLOAD(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    acc[pt] = DEX_MFMA_LP(__builtin_bit_cast(lp8, wr[tap * KSH + ks]), xq[tap & 1][ks * 2 + pt], acc[pt], 0, 0, 0);
                    if (tap == 4) accr[pt] = DEX_MFMA_LP(__builtin_bit_cast(lp8, wq[ks]), xq[tap & 1][ks * 2 + pt], accr[pt], 0, 0, 0);   // the 1x1 shortcut: centre tap
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef RWC_XLOAD
        lds_barrier();              // every wave is past the previous tile's reads of the output stages
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const bool live = iw0 + pt * 32 + i < p.W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bs + ct * 32 + 8 * g + 4 * hh), r4 = *reinterpret_cast<const float4*>(bs + COUT + ct * 32 + 8 * g + 4 * hh);
                const float v0 = acc[pt][4 * g] + b4.x, v1 = acc[pt][4 * g + 1] + b4.y, v2 = acc[pt][4 * g + 2] + b4.z, v3 = acc[pt][4 * g + 3] + b4.w;
                const int gi = (8 * g) / cpg;                // statistics of the raw conv output (lane = pixel: columns past the image edge do not count)
                if (live) { gs[gi] += (v0 + v1) + (v2 + v3); gq[gi] = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, gq[gi])))); }
                *reinterpret_cast<uint2*>(stage + (pt * 32 + i) * SPB + (ct * 32 + 8 * g + 4 * hh) * 2) = make_uint2(pack2_lp(v0, v1), pack2_lp(v2, v3));
                *reinterpret_cast<float4*>(stager + (pt * 32 + i) * SPBR + (ct * 32 + 8 * g + 4 * hh) * 4) =
                    make_float4(accr[pt][4 * g] + r4.x, accr[pt][4 * g + 1] + r4.y, accr[pt][4 * g + 2] + r4.z, accr[pt][4 * g + 3] + r4.w);
            }
        }
        lds_barrier();
        {   // output stages -> HBM, 16 B per lane
            const long rowpix = ((long)b * p.H + ih) * p.W;          // wave-uniform
#define RWC_OUT(pred_) do { \
                _Pragma("unroll") for (int h0 = 0; h0 < NS; h0 += 4) { \
                    uint4 ov[4]; \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int q = tid + 256 * (h0 + j); ov[j] = *reinterpret_cast<const uint4*>(stage + (q / OCH) * SPB + (q % OCH) * 16); } \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int q = tid + 256 * (h0 + j), wo = iw0 + q / OCH, e = wo * COUT + (q % OCH) * 8; \
                        if (pred_) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.Y) + rowpix * COUT + e) = ov[j]; } \
                } \
                _Pragma("unroll") for (int h0 = 0; h0 < NSR; h0 += 4) { \
                    uint4 ov[4]; \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int q = tid + 256 * (h0 + j); ov[j] = *reinterpret_cast<const uint4*>(stager + (q / OCHR) * SPBR + (q % OCHR) * 16); } \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const int q = tid + 256 * (h0 + j), wo = iw0 + q / OCHR, e = wo * COUT + (q % OCHR) * 4; \
                        if (pred_) *reinterpret_cast<uint4*>(p.res_y + rowpix * COUT + e) = ov[j]; } \
                } \
            } while (0)
            if (full_strip) RWC_OUT(true); else RWC_OUT(wo < p.W);
#undef RWC_OUT
        }
    }
    if (p.gn_stats) {
        const double inv_n = 1.0 / ((double)p.H * p.W * cpg);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float a_ = gs[g], q_ = gq[g];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { a_ += __shfl_xor(a_, o); q_ += __shfl_xor(q_, o); }
            if (lane == 0) { gn_add(&gnred[((ct * 32) / cpg + g) * 2], gn_fix(a_, inv_n)); gn_add(&gnred[((ct * 32) / cpg + g) * 2 + 1], gn_fix(q_, inv_n)); }
        }
        __syncthreads();
        if (tid < 16) {
            const long long v = gnred[tid];
            if (v != 0) gn_add(p.gn_stats + (((long)b * 8 + (tid >> 1)) * GN_SLOTS + blockIdx.x % GN_SLOTS) * 2 + (tid & 1), v);
        }
    }
}

// the batch regime of the 128-channel layers: enough strips x rows for one full round of workgroups with long strips
bool conv3x3_regw_form(const Conv3P& p) {
#if defined(DEX_LP_WSPLIT)
    return false;            // no split-weight form (lp_config.h)
#endif
   
    const bool off = knob_off("DEX_CONV_REGW");
    if (off || !p.Wfrag || p.res2_w) return false;
    const long min_tiles_r = knob_or("DEX_REGW_MIN_TILES", 1024);
    if (p.res_w) {      // 64 -> 128 + 1x1 shortcut on the plain Downsample output (conv3x3_rw_res128_kernel)
        const bool res_off = knob_off("DEX_CONV_REGW_RES");
        return !res_off && p.res_wfrag && p.Cin == 64 && p.Cout == 128 && (p.ldx % 8) == 0 && (p.x_coff % 8) == 0 && !p.pro_stats && !p.pro_res && p.y_bf16 &&
               (long)p.H * ((p.W + 63) / 64) * p.B >= min_tiles_r;
    }
    if (!(p.Cin == 128 && p.Cout == 128 && p.ldx == 128 && p.x_coff == 0)) return false;
    if (!p.y_bf16) return false;
    if (p.pro_stats ? !p.x_bf16 : true) return false;         // instantiated: the two GroupNorm-prologue forms on 16-bit h
    const long min_tiles = knob_or("DEX_REGW_MIN_TILES", 1024);
    return (long)p.H * ((p.W + 63) / 64) * p.B >= min_tiles;
}

template <int CIN, int COUT, int PROF, bool XB, bool YB, bool XOL = false>
static void rw_launch(const Conv3P& p, hipStream_t st) {
    using G = RwGeom<CIN, COUT, YB>;
    const int nseg = (p.W + G::MPX - 1) / G::MPX;
    const int target = knob_or("DEX_REGW_WGS", 256);
    int nchunk = (target + nseg * p.B - 1) / (nseg * p.B);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > p.H) nchunk = p.H;
    const int R = (p.H + nchunk - 1) / nchunk;
    nchunk = (p.H + R - 1) / R;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rw_kernel<CIN, COUT, PROF, XB, YB, XOL>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS); attr = true; }
    hipLaunchKernelGGL((conv3x3_rw_kernel<CIN, COUT, PROF, XB, YB, XOL>), dim3(nseg * nchunk, p.B), dim3(G::NTHR), G::LDS, st, p, nseg, R);
}

template <bool XB>
static void rw_launch_res128(const Conv3P& p, hipStream_t st) {
    constexpr int LDS = 4 * 66 * 144 + 64 * (128 * 2 + 16) + 64 * (128 * 4 + 16) + 2 * 128 * 4 + 16 * 8 + 16;
    const int nseg = (p.W + 63) / 64;
    int nchunk = (256 + nseg * p.B - 1) / (nseg * p.B);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > p.H) nchunk = p.H;
    const int R = (p.H + nchunk - 1) / nchunk;
    nchunk = (p.H + R - 1) / R;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rw_res128_kernel<XB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    hipLaunchKernelGGL((conv3x3_rw_res128_kernel<XB>), dim3(nseg * nchunk, p.B), dim3(256), LDS, st, p, nseg, R);
}

void launch_conv3x3_regw(const Conv3P& p, hipStream_t st) {
    if (p.res_w) { g_last_symbol = "conv3x3_rw_res128_kernel"; p.x_bf16 ? rw_launch_res128<true>(p, st) : rw_launch_res128<false>(p, st); return; }
    g_last_symbol = "conv3x3_rw_kernel";
    if (p.pro_res && p.xout_lp) rw_launch<128, 128, 2, true, true, true>(p, st);
    else if (p.pro_res) rw_launch<128, 128, 2, true, true>(p, st);
    else rw_launch<128, 128, 1, true, true>(p, st);
}

}  // namespace DEX_LP_NS
}  // namespace dex
