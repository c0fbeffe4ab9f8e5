// clock_probe: effective shader clock while (a) a single long MFMA-chain kernel runs, (b) many short kernels run.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void busy(float* out, long* cyc, long* wall, int iters) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
int main() {
    float* out; long *cyc, *wall; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8); hipMalloc(&wall, 8);
    long hc, hw;
    for (int iters : {200, 2000, 20000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, 0, out, cyc, wall, iters);
            hipDeviceSynchronize();
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
            printf("iters=%d: shader cycles=%ld wall ticks(100MHz)=%ld -> %.3f GHz, %.1f cyc/mfma\n", iters, hc, hw, hc / (hw * 10.0), (double)hc / iters);
        }
    }
    // many short kernels back to back
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, 0, out, cyc, wall, 64);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
        printf("2000 short kernels: %.3f us each; last kernel clock %.3f GHz (%ld cyc)\n", ms * 1e3 / 2000, hc / (hw * 10.0), hc);
    }
    return 0;
}
