// tools/mfmaceil.hip — the practical dense-MFMA ceiling of this chip: v_mfma_f32_32x32x16_{bf16,f16}
// with RANDOM operands on ALL CUs for >= 1 ms per measurement, nothing else in the loop.  Operands are refreshed from a small
// register pool every MFMA (16 distinct A / B fragments per wave, rotated), accumulators are 4 independent chains per wave, so
// the matrix pipe sees neither zero operands (which draw less power) nor a dependent-issue stall.  Reports sustained TFLOP/s
// from HIP events, the shader clock inside the kernel (clock64 / wall_clock64) and cycles per MFMA, for 1 / 2 / 3 waves per SIMD
// and for all-zero operands as the contrast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ __launch_bounds__(256) void ceil_kernel(const u32x4* __restrict__ pool, float* __restrict__ out, long long* __restrict__ clk, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = pool[((blockIdx.x * 4 + wave) % 61 * 16 + j) * 64 + lane];
        b[j] = pool[((blockIdx.x * 4 + wave) % 59 * 16 + 8 + j) * 64 + lane];
    }
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            if constexpr (F16) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[j]), __builtin_bit_cast(h8, b[j]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[j + 1]), __builtin_bit_cast(h8, b[j]), c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[j]), __builtin_bit_cast(h8, b[j + 1]), c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[j + 1]), __builtin_bit_cast(h8, b[j + 1]), c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[j]), __builtin_bit_cast(bf8, b[j]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[j + 1]), __builtin_bit_cast(bf8, b[j]), c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[j]), __builtin_bit_cast(bf8, b[j + 1]), c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[j + 1]), __builtin_bit_cast(bf8, b[j + 1]), c3, 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[(long)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static unsigned short f16_of(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

template <bool F16> static void run(const char* nm, int waves_per_simd, bool zeros, int iters) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, nb = cus * waves_per_simd;
    const size_t pool_n = 61 * 16 * 64;
    std::vector<unsigned> h(pool_n * 4);
    srand(1234);
    for (auto& w : h) {
        if (zeros) { w = 0; continue; }
        const float x = (float)rand() / RAND_MAX * 2.f - 1.f, y = (float)rand() / RAND_MAX * 2.f - 1.f;      // uniform(-1, 1): dense mantissas, mixed signs
        w = F16 ? ((unsigned)f16_of(x) | ((unsigned)f16_of(y) << 16)) : ((unsigned)bf16_of(x) | ((unsigned)bf16_of(y) << 16));
    }
    u32x4* pool; float* out; long long* clk;
    hipMalloc(&pool, h.size() * 4); hipMalloc(&out, (size_t)nb * 256 * 4); hipMalloc(&clk, (size_t)nb * 16);
    hipMemcpy(pool, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ceil_kernel<F16>), dim3(nb), dim3(256), 0, 0, pool, out, clk, iters / 8);     // warm-up, lets the clocks settle
    hipDeviceSynchronize();
    float best = 1e30f, ms;
    double ghz = 0, cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ceil_kernel<F16>), dim3(nb), dim3(256), 0, 0, pool, out, clk, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            std::vector<long long> hc((size_t)nb * 2); hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
            double a = 0, w = 0; for (int q = 0; q < nb; ++q) { a += hc[2 * q]; w += hc[2 * q + 1]; }
            ghz = a / w * 0.1;                          // wall_clock64 ticks at 100 MHz
            cyc = a / nb / ((double)iters * 16) / waves_per_simd;      // shader cycles of SIMD time per MFMA
        }
    }
    const double flop = (double)nb * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("%-46s %d wave/SIMD  %8.3f ms  %8.1f TFLOP/s (%.3f of 2.5 PF)  shader clock %.2f GHz  %.1f cycles of SIMD time per MFMA\n",
           nm, waves_per_simd, best, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 2.5e15, ghz, cyc);
    hipFree(pool); hipFree(out); hipFree(clk);
}

int main() {
    const int iters = 40000;       // 640k MFMAs per wave: >= 8 ms at the nominal rate
    for (int w = 1; w <= 3; ++w) run<false>("bf16 32x32x16, random operands", w, false, iters / w);
    for (int w = 1; w <= 2; ++w) run<true>("f16  32x32x16, random operands", w, false, iters / w);
    run<false>("bf16 32x32x16, ALL-ZERO operands (contrast)", 1, true, iters);
    run<false>("bf16 32x32x16, ALL-ZERO operands (contrast)", 2, true, iters / 2);
    return 0;
}
