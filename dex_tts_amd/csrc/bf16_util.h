// bf16_util.h — fp32 -> bf16 packing on the hardware converter.
#pragma once
#include <hip/hip_runtime.h>

namespace dex {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of 2 x bf16 (lo in bits 0..15), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950
// (the bit-twiddled software rounding it replaces cost ~10 VALU ops per pair on every staged element).
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}
// same instruction through inline asm: the attention kernels' register-heavy bodies make hipcc spill the
// staging arrays to scratch with the vector-convert form; the opaque asm form keeps them in VGPRs there
// (it schedules worse in the convolution/GEMM staging loops, which keep the form above).
__device__ __forceinline__ unsigned pack2_bf16_asm(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// (lo*m, hi*m) -> packed bf16 as ONE volatile asm block.  For register rings of raw fp32 loads: any plain C++ op on
// a loaded value (a mask multiply, a select, the vector convert) is pure, so the compiler hoists it to right behind
// the load to shorten live ranges — and with it the s_waitcnt, which turns a 3-deep prefetch ring back into
// load -> wait -> compute.  A volatile asm stays where it is written (below the barrier that precedes the LDS store).
__device__ __forceinline__ unsigned pack2_mul_bf16_pinned(float lo, float hi, float m) {
    unsigned r; float a, b;
    asm volatile("v_mul_f32 %1, %3, %5\n\tv_mul_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
                 : "=v"(r), "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi), "v"(m));
    return r;
}
__device__ __forceinline__ float mul_pinned(float x, float y) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Packed fp32 pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per VALU issue).  The GroupNorm +
// Mish prologues of the convolutions are VALU-bound at batch size (phase counters: the conversion of a row chunk costs
// more cycles than its 72 MFMAs), so they run on pairs; only v_exp_f32 / v_rcp_f32 / the clamp stay per value.
#ifdef DEX_NO_PK
struct f32x2 { float x, y; };      // same layout, single-value arithmetic
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, float b) { return f32x2{a.x * b, a.y * b}; }
__device__ __forceinline__ f32x2 operator+(f32x2 a, float b) { return f32x2{a.x + b, a.y + b}; }
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
#endif
// DEX_NO_PK (per translation unit): the same arithmetic on single values.  A packed fp32 VALU instruction beside a wave's
// MFMAs is not free the way a plain one is: tools/mfmabench measures 32.1 cycles per v_mfma_f32_32x32x16_bf16 alone, 32.8 with
// two v_fma_f32 per MFMA, 43.3 with ONE v_pk_fma_f32 per MFMA - kernels whose prologue arithmetic runs on the SIMDs that are
// issuing MFMAs (the ping-pong strip convolution, the weights-in-registers convolution) take the single-value form.
#ifdef DEX_NO_PK
__device__ __forceinline__ f32x2 pk_fma_pinned(f32x2 x, f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(r.x), "=&v"(r.y) : "v"(x.x), "v"(x.y), "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    return r;
}
__device__ __forceinline__ float mish1_add(float t, float add) {
    const float e = __builtin_amdgcn_exp2f(fminf(t * 1.44269504088896f, 28.8539008f));
    const float n = e * (e + 2.f);
    return fmaf(t, n * __builtin_amdgcn_rcpf(n + 2.f), add);
}
__device__ __forceinline__ f32x2 mish2_add(f32x2 t, f32x2 add) { f32x2 r; r.x = mish1_add(t.x, add.x); r.y = mish1_add(t.y, add.y); return r; }
#else
__device__ __forceinline__ f32x2 pk_fma_pinned(f32x2 x, f32x2 a, f32x2 b) {     // opaque: stays where it is written (see fma_pinned)
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    return r;
}
// Mish(t) + add = t * n / (n + 2) + add with n = e (e + 2), e = exp(min(t, 20))  (tanh(softplus(t)) == 1 to fp32 beyond 20)
__device__ __forceinline__ f32x2 mish2_add(f32x2 t, f32x2 add) {
    f32x2 l = t * 1.44269504088896f;
    l.x = fminf(l.x, 28.8539008f); l.y = fminf(l.y, 28.8539008f);
    f32x2 e; e.x = __builtin_amdgcn_exp2f(l.x); e.y = __builtin_amdgcn_exp2f(l.y);
    const f32x2 n = e * (e + 2.f);
    const f32x2 d = n + 2.f;
    f32x2 r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return t * (n * r) + add;
}
#endif

__device__ __forceinline__ unsigned mov_pinned(unsigned x) {
    unsigned r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// one dword of 2 x bf16 -> the two fp32 values (exact)
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned short bf16_bits(float x) { return (unsigned short)(pack2_bf16(x, 0.f) & 0xffffu); }

// fp16 twins (the fp16 mode's stores / loads in the element-wise kernels that are compiled once and take the mode as a
// runtime field: 1 = bf16, 2 = fp16)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_f16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));      // v_cvt_pk_f16_f32, round to nearest even
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2_t, u)[0]; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2_t, u)[1]; }
__device__ __forceinline__ unsigned pack2_kind(float lo, float hi, int kind) { return kind == 2 ? pack2_f16(lo, hi) : pack2_bf16(lo, hi); }
__device__ __forceinline__ float lo_kind(unsigned u, int kind) { return kind == 2 ? f16_lo(u) : bf16_lo(u); }
__device__ __forceinline__ float hi_kind(unsigned u, int kind) { return kind == 2 ? f16_hi(u) : bf16_hi(u); }

// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: one VALU move, no LDS crossbar
__device__ __forceinline__ float lane_xor1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}

// ---- order-independent normalisation statistics -------------------------------------------------------------------
// A partial sum (fp32, produced in a fixed order by one wave / workgroup) enters the shared accumulators as
// round(partial / n * 2^36) in a 64-bit integer, n = elements per statistic.  Integer atomics commute, so the totals — and
// everything downstream — are bitwise identical from call to call.  Resolution 1.5e-11 on the mean / mean of squares
// (finer than one fp32 ulp of any variance that matters next to eps = 1e-5), range +-1.3e8 (an RMS of 11 600 before the
// norm); conversions saturate instead of wrapping.
constexpr double GN_FIX_ONE = 68719476736.0;           // 2^36
__device__ __forceinline__ long long gn_fix(float partial, double inv_n) {
    double d = (double)partial * inv_n * GN_FIX_ONE;
    d = fmin(fmax(d, -9.0e18), 9.0e18);
    return __double2ll_rn(d);
}
__device__ __forceinline__ void gn_add(long long* dst, long long v) {       // global: global_atomic_add_x2, LDS: ds_add_u64
    atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)v);
}
// every lane of a GN_SLOTS-lane group holds one slot's (s1, s2): after the xor reduction all of them hold the totals
template <int SLOTS>
__device__ __forceinline__ void gn_slots_reduce(long long& s1, long long& s2) {
#pragma unroll
    for (int o = 1; o < SLOTS; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
}
// totals -> mean, 1/sqrt(biased var + eps)
__device__ __forceinline__ void gn_moments(long long s1, long long s2, double eps, float& mean_out, float& rstd_out) {
    const double mean = (double)s1 * (1.0 / GN_FIX_ONE);
    double var = (double)s2 * (1.0 / GN_FIX_ONE) - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mean_out = (float)mean;
    rstd_out = (float)(1.0 / sqrt(var + eps));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full workgroup-scope fence, which the
// compiler implements as s_waitcnt vmcnt(0): every global load still in flight (a prefetched weight tile, the next
// K tile of a register ring) is waited for at EVERY barrier, which serialises software pipelines.  Use this one
// when the data exchanged across the barrier lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace dex
