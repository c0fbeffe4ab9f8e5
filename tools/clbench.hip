// tools/clbench.hip — the cluster form of the fused DiT block (dit_rowchain_cluster_kernel) alone at B=1 / N=650, against the
// one-workgroup form, with the phase stamps of a -DDEX_TIMING build (10-ns units, per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEX_TIMING -I dex_tts_amd/csrc tools/clbench.hip -o tools/clbench
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
    if (half) { unsigned short* u = (unsigned short*)h.data(); for (size_t i = 0; i < bytes / 2; ++i) u[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 23) % 0x180; }
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
    const int tiles_live = B * ((N + 31) / 32);
    const int tiles = (tiles_live + 7) / 8 * 8;           // (room for the XCD-local grid)
    float* slab; unsigned* flag; int* err;
    hipMalloc(&slab, (size_t)tiles * DIT_CLUSTER_SLAB_FLOATS * 4); hipMalloc(&flag, (size_t)tiles * DIT_CLUSTER_FLAG_WORDS * 4); hipMalloc(&err, 4);
    hipMemset(flag, 0, (size_t)tiles * DIT_CLUSTER_FLAG_WORDS * 4); hipMemset(err, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 200;
    unsigned epoch = 0;
    for (int form = 0; form < 3; ++form) {
        c.xslab = form ? slab : nullptr; c.xflag = flag; c.xerr = err; c.xlocal = form == 2;
        for (int i = 0; i < 5; ++i) { c.epoch = ++epoch; launch_dit_rowchain(c, 0); }
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i) { c.epoch = ++epoch; launch_dit_rowchain(c, 0); }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s N=%d B=%d: %.2f us per launch (back to back)\n", form == 2 ? "cluster form, members on one XCD        " : form ? "cluster form (4 workgroups per row tile)" : "one workgroup per row tile              ", N, B, ms * 1e3 / iters);
    }
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("hand-off time-outs: %d\n", herr);
#ifdef DEX_TIMING
    long long* dbg; hipMalloc(&dbg, 65536 * 8); hipMemset(dbg, 0, 65536 * 8);
  for (int loc = 0; loc < 2; ++loc) {
    c.xlocal = loc; hipMemset(dbg, 0, 65536 * 8);
    printf(loc ? "XCD-local clusters:\n" : "clusters across XCDs:\n");
    c.dbg = dbg; c.epoch = ++epoch; launch_dit_rowchain(c, 0); hipDeviceSynchronize(); c.dbg = nullptr;
    const int nb = (loc ? tiles : tiles_live) * DIT_CLUSTER;
    std::vector<long long> h((size_t)nb * 16); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0], t1 = 0;
    int live = 0;
    for (int bl = 0; bl < nb; ++bl) if (h[bl * 16]) { t0 = t0 ? std::min(t0, h[bl * 16]) : h[bl * 16]; t1 = std::max(t1, h[bl * 16 + 14]); ++live; }
    printf("  first start -> last end: %lld (10 ns)\n", t1 - t0);
    const char* nm[14] = {"attn", "merge", "proj", "publish0", "wait0", "reduce0", "LN1", "fc1", "fc2", "publish1", "wait1", "reduce1+X", "LN2(part)", "qkv"};
    double avg[14] = {0};
    for (int bl = 0; bl < nb; ++bl) if (h[bl * 16]) for (int q = 0; q < 14; ++q) avg[q] += (double)(h[bl * 16 + q + 1] - h[bl * 16 + q]) / live;
    printf("  mean over %d workgroups (10 ns):", live);
    for (int q = 0; q < 14; ++q) printf(" %s=%.0f", nm[q], avg[q]);
    printf("\n");
    for (int bl : {0, 1, 2, 3, nb / 2, nb - 1}) {
        long long* d = &h[bl * 16];
        if (!d[0]) continue;
        printf("  wg %3d (member %d): start+%lld", bl, loc ? (bl >> 3) % DIT_CLUSTER : bl % DIT_CLUSTER, d[0] - t0);
        for (int q = 0; q < 14; ++q) printf(" %s=%lld", nm[q], d[q + 1] - d[q]);
        printf(" total=%lld\n", d[14] - d[0]);
    }
  }
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("hand-off errors after the stamped launches: %d\n", herr);
#endif
    return 0;
}
