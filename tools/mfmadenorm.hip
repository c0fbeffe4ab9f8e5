// does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the low halves of split weights live there)?  hipcc --offload-arch=gfx950 -O2 tools/mfmadenorm.hip -o tools/mfmadenorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // lane l holds row (l & 31), k = 8 * (l >> 5) .. + 7: put one non-zero per row/col at k = 0
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f16v c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float as[] = {1.0f, 6.103515625e-05f, 3.0517578125e-05f, 5.9604644775390625e-08f, 1.2e-6f, 2.5e-5f};
    for (float a : as) {
        k<<<1, 64>>>(a, 1.0f, d); float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        k<<<1, 64>>>(1.0f, a, d); float h2; hipMemcpy(&h2, d, 4, hipMemcpyDeviceToHost);
        k<<<1, 64>>>(a, 1024.0f, d); float h3; hipMemcpy(&h3, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.6e (fp16 %s): a*1 = %.6e, 1*a = %.6e, a*1024 = %.6e  (expected %.6e)\n", a, fabsf(a) < 6.1e-5f ? "subnormal" : "normal", h, h2, h3, (float)(_Float16)a);
    }
    return 0;
}
