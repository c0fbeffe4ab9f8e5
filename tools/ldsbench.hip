// tools/ldsbench.hip — ds_read_b128 throughput per CU for the operand-fetch patterns of the convolution kernels:
// lane-contiguous (fragment order) against a pixel-strided row layout (lane i -> pixel i, 272 B apart; upper half-wave +16 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int BATCH>
__global__ __launch_bounds__(1024) void lds_read_kernel(unsigned* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < 16384; k += blockDim.x) reinterpret_cast<unsigned*>(smem)[k] = k;
    __syncthreads();
    const int i = lane & 31, hh = lane >> 5;
    unsigned base;
    if (MODE == 0) base = lane * 16;                               // contiguous 1 KB per wave
    else if (MODE == 1) base = i * 272 + hh * 16;                  // pixel rows, 272 B apart
    else if (MODE == 2) base = i * 144 + hh * 16;                  // 144 B apart (64-channel rows)
    else base = (i * 272 + hh * 16) ^ ((i >> 2 & 3) << 4);          // 272 B + a chunk swizzle
    base += wave * 32;
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) v[j] = *reinterpret_cast<const u32x4*>(smem + ((base + j * 32 + (it & 7) * 2048) & 0xffff));
#pragma unroll
        for (int j = 0; j < BATCH; ++j) acc ^= v[j];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int BATCH>
static void run(const char* nm, unsigned* out, long long* cyc, int nthreads) {
    const int iters = 2000, nb = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_read_kernel<MODE, BATCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((lds_read_kernel<MODE, BATCH>), dim3(nb), dim3(nthreads), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double a = 0; for (int k = 0; k < nb; ++k) a += h[k];
    a /= nb;
    const double reads = (double)iters * BATCH * (nthreads / 64);
    printf("%-28s batch %2d, %d waves/CU: %.1f cycles per ds_read_b128 per wave, %.0f B/clk/CU\n", nm, BATCH, nthreads / 64, a / (iters * BATCH), reads * 1024 / a);
}
int main() {
    unsigned* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int nt : {64, 256, 512, 1024}) {
        run<0, 8>("lane-contiguous", out, cyc, nt); run<0, 16>("lane-contiguous", out, cyc, nt);
        run<1, 8>("pixel rows 272 B", out, cyc, nt); run<1, 16>("pixel rows 272 B", out, cyc, nt);
        run<2, 8>("pixel rows 144 B", out, cyc, nt); run<2, 16>("pixel rows 144 B", out, cyc, nt);
        run<3, 8>("272 B + swizzle", out, cyc, nt);
    }
    return 0;
}
