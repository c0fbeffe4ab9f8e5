// tools/rc64check.hip — dit_rowchain64_kernel against dit_rowchain_kernel<false> on the same inputs (qkv-only launch): bitwise.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dex_tts_amd/csrc tools/rc64check.hip -o tools/rc64check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include "../dex_tts_amd/csrc/dit_rowchain.hip"
namespace dex { thread_local const char* g_last_symbol = ""; }
using namespace dex;
using namespace dex::bf16;
static void* dfill(size_t bytes, bool half, unsigned seed) {
    void* p; hipMalloc(&p, bytes);
    std::vector<unsigned char> h(bytes);
    if (half) { unsigned short* u = (unsigned short*)h.data(); for (size_t i = 0; i < bytes / 2; ++i) u[i] = 0x3c00 + (unsigned short)(((i + seed) * 2654435761u) >> 23) % 0x180 + ((i & 1) ? 0x8000 : 0); }
    else { float* f = (float*)h.data(); for (size_t i = 0; i < bytes / 4; ++i) f[i] = 0.5f * (float)(((i + seed) * 2654435761u) % 1000) / 1000.f - 0.25f; }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
    return p;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 80, B = argc > 2 ? atoi(argv[2]) : 2;
    const int Npad = (N + 31) / 32 * 32 + 32;
    float* X = (float*)dfill((size_t)B * N * 256 * 4, false, 1);
    void* Wq = dfill(256 * 768 * 2, true, 2);
    float* bias = (float*)dfill(768 * 4, false, 3); float* ada = (float*)dfill(6 * 256 * 4, false, 4);
    const size_t qb = (size_t)B * 2 * Npad * 128 * 2;
    void* out[2][3];
    for (int s = 0; s < 2; ++s) for (int k = 0; k < 3; ++k) { hipMalloc(&out[s][k], qb); hipMemset(out[s][k], 0, qb); }
    for (int s = 0; s < 2; ++s) {
        DitChainP c{}; c.heads = 2; c.rows_per_batch = N; c.X = X; c.Wq = Wq; c.bq = bias; c.ada = ada;
        c.next_shift = ada; c.next_scale = ada + 256; c.next_step_stride = 0; c.Npad = Npad; c.qscale = 0.127f; c.M = B * N; c.B = B;
        c.qkv_only = 1; c.Qh = out[s][0]; c.Kh = out[s][1]; c.Vt = out[s][2];
        setenv("DEX_ROWCHAIN64", s ? "2" : "0", 1);
        if (s == 0) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC_LDS);
            hipLaunchKernelGGL(dit_rowchain_kernel<false>, dim3(B * ((N + 31) / 32)), dim3(512), RC_LDS, 0, c);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&dit_rowchain64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC64_LDS);
            hipLaunchKernelGGL(dit_rowchain64_kernel, dim3(B * ((N + 63) / 64)), dim3(512), RC64_LDS, 0, c);
        }
        hipError_t e = hipDeviceSynchronize();
        printf("launch %d: %s\n", s, hipGetErrorString(e));
    }
    const char* nm[3] = {"q", "k", "v^T"};
    for (int k = 0; k < 3; ++k) {
        std::vector<unsigned short> a(qb / 2), b2(qb / 2);
        hipMemcpy(a.data(), out[0][k], qb, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), out[1][k], qb, hipMemcpyDeviceToHost);
        long diff = 0, first = -1;
        const long tiles = (N + 31) / 32;         // compare the tiles the 32-row kernel writes
        for (int bh = 0; bh < B * 2; ++bh)
            for (long e = 0; e < tiles * 4096; ++e) { const long idx = (long)bh * Npad * 128 + e; if (a[idx] != b2[idx]) { ++diff; if (first < 0) first = idx; } }
        printf("%s: %ld differing elements of %ld (first at %ld: bh %ld tile %ld e %ld)\n", nm[k], diff, (long)B * 2 * tiles * 4096, first,
               first < 0 ? -1 : first / ((long)Npad * 128), first < 0 ? -1 : (first % ((long)Npad * 128)) / 4096, first < 0 ? -1 : first % 4096);
    }
    return 0;
}
