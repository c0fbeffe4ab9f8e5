// tools/mfmabench.hip — issue rate of v_mfma_f32_32x32x16_bf16 on one wave per SIMD under the ingredients of the convolution
// kernels: operand files (AGPR / VGPR weights, accumulators in either file), ds_read_b128 per MFMA, VALU fillers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF_A(acc, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(x))
#define MF_V(acc, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x))
#define MF_VV(acc, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x))
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int q = tid; q < 8192; q += 256) reinterpret_cast<unsigned*>(smem)[q] = 0;
    __syncthreads();
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; a2[r] = 0; a3[r] = 0; }
    u32x4 w[8];
    for (int j = 0; j < 8; ++j) w[j] = u32x4{(unsigned)j, 0u, 0u, 0u};
    u32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
    float f0 = tid, f1 = 1.f, f2 = 0.5f;
    typedef float f2t __attribute__((ext_vector_type(2)));
    f2t p0 = {1.f, 2.f}, p1 = {0.5f, 0.25f};
    const unsigned char* lp = smem + (lane & 31) * 272 + (lane >> 5) * 16;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            if (MODE == 3 || MODE == 6) { x0 = *reinterpret_cast<const u32x4*>(lp + j * 32); x1 = *reinterpret_cast<const u32x4*>(lp + j * 32 + 8704); }
            if (MODE == 0 || MODE >= 3) { MF_A(a0, w[j], x0); MF_A(a1, w[j + 1], x0); }
            if (MODE == 1) { MF_V(a0, w[j], x0); MF_V(a1, w[j + 1], x0); }
            if (MODE == 2) { MF_VV(a0, w[j], x0); MF_VV(a1, w[j + 1], x0); }
            if (MODE == 4) { asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); }
            if (MODE == 5 || MODE == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1)); }
            if (MODE == 0 || MODE >= 3) { MF_A(a2, w[j], x1); MF_A(a3, w[j + 1], x1); }
            if (MODE == 1) { MF_V(a2, w[j], x1); MF_V(a3, w[j + 1], x1); }
            if (MODE == 2) { MF_VV(a2, w[j], x1); MF_VV(a3, w[j + 1], x1); }
            if (MODE == 4) { asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); }
            if (MODE == 5 || MODE == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1)); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3));
    const long long t1 = __builtin_readcyclecounter();
    float s = f0 + p0.x + p0.y;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static void run(const char* nm, float* out, long long* cyc) {
    const int iters = 2000, nb = 256;
    hipLaunchKernelGGL((k<MODE>), dim3(nb), dim3(256), 32768, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double a = 0; for (int q = 0; q < nb; ++q) a += h[q];
    printf("%-64s %.1f cycles per MFMA\n", nm, a / nb / (iters * 16.0));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("acc AGPR, weights AGPR, pixels VGPR", out, cyc);
    run<1>("acc AGPR, weights VGPR", out, cyc);
    run<2>("acc VGPR, weights VGPR", out, cyc);
    run<3>("acc AGPR, weights AGPR + 1 ds_read_b128 per 2 MFMAs", out, cyc);
    run<4>("acc AGPR, weights AGPR + 2 v_fma_f32 per MFMA", out, cyc);
    run<5>("acc AGPR, weights AGPR + 1 v_pk_fma_f32 per MFMA", out, cyc);
    run<6>("acc AGPR, weights AGPR + ds_read + v_pk_fma", out, cyc);
    return 0;
}
