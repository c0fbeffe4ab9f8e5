// tools/rcbench.hip — the fused DiT block launch (dit_rowchain_kernel<true>: attention core + projection + MLP + next qkv)
// alone at B=1 / N=650, with the phase stamps of a -DDEX_TIMING build.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEX_TIMING -I dex_tts_amd/csrc tools/rcbench.hip -o tools/rcbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>
#include "../dex_tts_amd/csrc/dit_rowchain.hip"
namespace dex { thread_local const char* g_last_symbol = ""; }
using namespace dex;
using namespace dex::bf16;

static void* dfill(size_t bytes, bool half) {
    void* p; hipMalloc(&p, bytes);
    std::vector<unsigned char> h(bytes);
    if (half) { unsigned short* u = (unsigned short*)h.data(); for (size_t i = 0; i < bytes / 2; ++i) u[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 23) % 0x180; }   // bf16 ~ [0.008, 0.03]
    else { float* f = (float*)h.data(); for (size_t i = 0; i < bytes / 4; ++i) f[i] = 0.1f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.05f; }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 650, B = argc > 2 ? atoi(argv[2]) : 1;
    const int Npad = (N + 31) / 32 * 32 + 32;
    float* X = (float*)dfill((size_t)B * N * 256 * 4, false);
    void *Wp = dfill(256 * 256 * 2, true), *W1 = dfill(256 * 512 * 2, true), *W2 = dfill(512 * 256 * 2, true), *Wq = dfill(256 * 768 * 2, true);
    void* qkv[2][3];
    for (int s = 0; s < 2; ++s) for (int k = 0; k < 3; ++k) qkv[s][k] = dfill((size_t)B * 2 * Npad * 128 * 2, true);
    float* bias = (float*)dfill(768 * 4, false); float* ada = (float*)dfill(6 * 256 * 4, false);
    DitChainP c{}; c.heads = 2; c.rows_per_batch = N; c.X = X; c.Wp = Wp; c.W1 = W1; c.W2 = W2; c.Wq = Wq; c.bp = bias; c.b1 = bias; c.b2 = bias; c.bq = bias;
    c.ada = ada; c.next_shift = ada; c.next_scale = ada + 256; c.next_step_stride = 0; c.Npad = Npad; c.qscale = 0.088f * 1.4427f; c.M = B * N; c.B = B;
    c.attn_inline = 1; c.Qin = qkv[0][0]; c.Kin = qkv[0][1]; c.Vin = qkv[0][2]; c.Qh = qkv[1][0]; c.Kh = qkv[1][1]; c.Vt = qkv[1][2];
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch_dit_rowchain(c, 0);
    hipDeviceSynchronize();
    const int iters = 200;
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch_dit_rowchain(c, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("dit_rowchain<attn> N=%d B=%d: %.2f us per launch (back to back)\n", N, B, ms * 1e3 / iters);
#ifdef DEX_TIMING
    long long* dbg; hipMalloc(&dbg, 4096 * 8); hipMemset(dbg, 0, 4096 * 8);
    c.dbg = dbg; launch_dit_rowchain(c, 0); hipDeviceSynchronize(); c.dbg = nullptr;
    std::vector<long long> h(4096); hipMemcpy(h.data(), dbg, 4096 * 8, hipMemcpyDeviceToHost);
    const int nb = B * ((N + 31) / 32);
    long long t0 = h[0], t1 = 0;
    for (int bl = 0; bl < nb; ++bl) { t0 = std::min(t0, h[bl * 8]); t1 = std::max(t1, h[bl * 8 + 7]); }
    printf("  first start -> last end: %lld (10 ns)\n", t1 - t0);
    for (int bl : {0, nb / 2, nb - 1}) { long long* d = &h[bl * 8]; long long at = h[1024 + bl * 4]; long long* e = &h[1024 + bl * 4];
        printf("           partial write+loads issued=%lld barrier=%lld merge=%lld stage As=%lld\n", e[1] - e[0], e[2] - e[1], e[3] - e[2], d[1] - e[3]);
        printf("  blk %2d: start+%lld attn=%lld merge+stage=%lld proj=%lld LN1=%lld fc1=%lld fc2=%lld LN2=%lld qkv=%lld total=%lld (10 ns)\n", bl, d[0] - t0,
               at - d[0], d[1] - at, d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5], d[7] - d[6], d[7] - d[0]); }
#endif
    return 0;
}
