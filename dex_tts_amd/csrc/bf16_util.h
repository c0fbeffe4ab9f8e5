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

}  // namespace dex
