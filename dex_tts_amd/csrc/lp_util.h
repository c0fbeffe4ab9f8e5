// lp_util.h — operand-type helpers of the low-precision translation units (see lp_config.h): packed converts, unpack,
// the MFMA of the mode.  Lives in namespace dex::DEX_LP_NS so the bf16 and fp16 builds of one source never share a symbol.
#pragma once
#include "bf16_util.h"
#include "lp_config.h"

namespace dex {
namespace DEX_LP_NS {

#ifdef DEX_LP_F16
typedef _Float16 lp_t;
#define DEX_MFMA_LP __builtin_amdgcn_mfma_f32_32x32x16_f16
constexpr bool LP_IS_F16 = true;
#else
typedef __bf16 lp_t;
#define DEX_MFMA_LP __builtin_amdgcn_mfma_f32_32x32x16_bf16
constexpr bool LP_IS_F16 = false;
#endif
typedef lp_t lp8 __attribute__((ext_vector_type(8)));
typedef lp_t lp2_t __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of 2 low-precision values (lo in bits 0..15), round-to-nearest-even: ONE v_cvt_pk_{bf16,f16}_f32
__device__ __forceinline__ unsigned pack2_lp(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const lp2_t r = __builtin_convertvector(v, lp2_t);
    return __builtin_bit_cast(unsigned, r);
}
// same instruction through inline asm (see pack2_bf16_asm in bf16_util.h for why both forms exist)
__device__ __forceinline__ unsigned pack2_lp_asm(float lo, float hi) {
    unsigned r;
#ifdef DEX_LP_F16
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#else
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
#endif
    return r;
}
// (lo*m, hi*m) -> packed, as ONE volatile asm block (register rings: stays below the barrier it is written under)
__device__ __forceinline__ unsigned pack2_mul_lp_pinned(float lo, float hi, float m) {
    unsigned r; float a, b;
#ifdef DEX_LP_F16
    asm volatile("v_mul_f32 %1, %3, %5\n\tv_mul_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %0, %1, %2"
                 : "=v"(r), "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi), "v"(m));
#else
    asm volatile("v_mul_f32 %1, %3, %5\n\tv_mul_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %0, %1, %2"
                 : "=v"(r), "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi), "v"(m));
#endif
    return r;
}
// one dword of 2 low-precision values -> the two fp32 values (exact)
#ifdef DEX_LP_F16
__device__ __forceinline__ float lp_lo(unsigned u) { return (float)__builtin_bit_cast(lp2_t, u)[0]; }
__device__ __forceinline__ float lp_hi(unsigned u) { return (float)__builtin_bit_cast(lp2_t, u)[1]; }
#else
__device__ __forceinline__ float lp_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float lp_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif
__device__ __forceinline__ unsigned short lp_bits(float x) { return (unsigned short)(pack2_lp(x, 0.f) & 0xffffu); }

}  // namespace DEX_LP_NS
}  // namespace dex
