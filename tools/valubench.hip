// tools/valubench.hip — what fits between two v_mfma_f32_32x32x16_bf16 of a wave for free: k independent fillers of one kind per MFMA
// (plain / packed fp32 FMA, transcendentals, packs, permlane swaps, LDS reads / writes), one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valubench.hip -o tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f2t __attribute__((ext_vector_type(2)));
#define MF(acc, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x))

template <int KIND, int K>
__device__ __forceinline__ void fill(float (&f)[8], f2t (&p)[8], u32x4& l0, u32x4& l1, unsigned lds_addr) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
        if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(f[(q + 1) & 7]));
        if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q & 7]) : "v"(p[(q + 1) & 7]));
        if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
        if constexpr (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[q & 7]));
        if constexpr (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[q & 7]) : "v"(f[(q + 1) & 7]), "v"(f[(q + 2) & 7]));
        if constexpr (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(f[(2 * q) & 7]), "+v"(f[(2 * q + 1) & 7]));
        if constexpr (KIND == 7) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(q & 1 ? l1 : l0) : "v"(lds_addr));
        if constexpr (KIND == 8) asm volatile("ds_write_b128 %0, %1 offset:0" : : "v"(lds_addr), "v"(l0));
        if constexpr (KIND == 9) asm volatile("ds_write_b64 %0, %1 offset:0" : : "v"(lds_addr), "v"(p[q & 7]));
        if constexpr (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[q & 7]) : "v"(p[(q + 1) & 7]));
        if constexpr (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[q & 7]) : "v"(p[(q + 1) & 7]));
        if constexpr (KIND == 12) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(f[q & 7]));
        if constexpr (KIND == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[q & 7]) : "v"(f[(q + 1) & 7]));
        if constexpr (KIND == 14) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[0]) : "v"(f[(q + 1) & 7]), "v"(f[(q + 2) & 7]));      // dependent chain on f[0], as a row sum
        if constexpr (KIND == 15) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[q & 7]) : "v"(f[(q + 1) & 7]), "v"(f[(q + 2) & 7]));
        if constexpr (KIND == 16) asm volatile("s_add_i32 %0, %0, 3" : "+s"(l0[0]) : : "scc");
        if constexpr (KIND == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[(q + 1) & 7]));                                   // dependent chain on f[0]
        if constexpr (KIND == 18) { asm volatile("v_exp_f32 %0, %0" : "+v"(f[1 + (q & 3)])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[1 + (q & 3)])); }   // exp + dependent add
        if constexpr (KIND == 19) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[0]) : "v"(f[(q + 1) & 7]), "v"(f[(q + 2) & 7]));
        if constexpr (KIND == 20) asm volatile("s_nop 0");
    }
}

template <int KIND, int K>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int q = tid; q < 16384; q += blockDim.x) reinterpret_cast<unsigned*>(smem)[q] = 0;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    u32x4 w = {1u, 0u, 0u, 0u}, x = {0, 0, 0, 0}, l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
    float f[8]; f2t p[8];
    for (int q = 0; q < 8; ++q) { f[q] = 1.f + 0.001f * q; p[q] = f2t{1.f, 0.5f}; }
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + (tid >> 6) * 4352 + (lane & 31) * 272 + (lane >> 5) * 16;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            MF(a0, w, x); fill<KIND, K>(f, p, l0, l1, lds_addr);
            MF(a1, w, x); fill<KIND, K>(f, p, l0, l1, lds_addr);
        }
        if (KIND == 7 || KIND == 8 || KIND == 9) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1));
    const long long t1 = __builtin_readcyclecounter();
    float s = l0[0] + l1[1];
    for (int q = 0; q < 8; ++q) s += f[q] + p[q].x + p[q].y;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0 && tid < 256) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}
template <int KIND, int K> static void run1(int threads, float* out, long long* cyc, double* res) {
    const int iters = 1000, nb = 256;
    hipLaunchKernelGGL((k<KIND, K>), dim3(nb), dim3(threads), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[1024]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double a = 0; for (int q = 0; q < nb * 4; ++q) a += h[q];
    *res = a / (nb * 4) / (iters * 16.0);
}
template <int KIND> static void run(const char* nm, float* out, long long* cyc) {
    double r[2][6];
    for (int t = 0; t < 2; ++t) {
        const int th = t ? 512 : 256;
        run1<KIND, 1>(th, out, cyc, &r[t][0]); run1<KIND, 2>(th, out, cyc, &r[t][1]); run1<KIND, 3>(th, out, cyc, &r[t][2]);
        run1<KIND, 4>(th, out, cyc, &r[t][3]); run1<KIND, 6>(th, out, cyc, &r[t][4]); run1<KIND, 8>(th, out, cyc, &r[t][5]);
    }
    printf("%-22s 1 wave/SIMD:", nm); for (int q = 0; q < 6; ++q) printf(" %6.1f", r[0][q]);
    printf("   2 waves/SIMD (cycles per MFMA of ONE wave):"); for (int q = 0; q < 6; ++q) printf(" %6.1f", r[1][q]);
    printf("\n");
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
    printf("cycles per MFMA with k = 1 2 3 4 6 8 independent fillers after every MFMA (bare: 32 at one wave per SIMD, 64 at two)\n");
    run<1>("v_fma_f32", out, cyc); run<13>("v_mul_f32", out, cyc); run<2>("v_pk_fma_f32", out, cyc); run<10>("v_pk_mul_f32", out, cyc); run<11>("v_pk_add_f32", out, cyc);
    run<3>("v_exp_f32", out, cyc); run<4>("v_rcp_f32", out, cyc); run<5>("v_cvt_pk_bf16_f32", out, cyc); run<6>("v_permlane32_swap", out, cyc);
    run<12>("v_accvgpr_read", out, cyc);
    run<14>("v_dot2c_f32_bf16 chain", out, cyc); run<19>("v_dot2_f32_bf16 chain", out, cyc); run<15>("v_max3_f32", out, cyc); run<16>("s_add_i32", out, cyc); run<17>("v_add_f32 chain", out, cyc);
    run<18>("v_exp+v_add dep", out, cyc); run<20>("s_nop 0", out, cyc);
    run<7>("ds_read_b128", out, cyc); run<8>("ds_write_b128", out, cyc); run<9>("ds_write_b64", out, cyc);
    return 0;
}
