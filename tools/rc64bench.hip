// tools/rc64bench.hip — the 64-row DiT row chain (dit_rowchain64_kernel: projection + MLP + next qkv of 64 token rows per workgroup)
// alone at a batch shape (default DEX B = 32, N = 1300: 672 workgroups), with the phase stamps of a -DDEX_TIMING build.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DDEX_TIMING] [-DRC64_V2] -I dex_tts_amd/csrc -I include tools/rc64bench.hip -o tools/rc64bench
//   tools/rc64bench [N] [B] [o_lp]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>
#include "../dex_tts_amd/csrc/dit_rowchain.hip"
namespace dex { thread_local const char* g_last_symbol = ""; thread_local bool g_lp_wsplit = false;
int knob(const char* name) { const char* e = getenv(name); return e ? atoi(e) : KNOB_UNSET; } }
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
    const int N = argc > 1 ? atoi(argv[1]) : 1300, B = argc > 2 ? atoi(argv[2]) : 32, o_lp = argc > 3 ? atoi(argv[3]) : 1;
    const int Npad = (N + 31) / 32 * 32;        // the product's padding (dex_api.hip); the buffers carry a guard tile behind the last head
    const size_t xb = (size_t)B * N * 256 * 4;
    float* X0 = (float*)dfill(xb, false, 1);
    float* X; hipMalloc(&X, xb);
    void* O = dfill((size_t)B * N * 256 * (o_lp ? 2 : 4), o_lp != 0, 7);
    void *Wp = dfill(256 * 256 * 2, true, 2), *W1 = dfill(256 * 512 * 2, true, 3), *W2 = dfill(512 * 256 * 2, true, 4), *Wq = dfill(256 * 768 * 2, true, 5);
    void* qkv[3];
    const size_t qbytes = (size_t)B * 2 * Npad * 128 * 2, qguard = 64 * 128 * 2;
    for (int k = 0; k < 3; ++k) { hipMalloc(&qkv[k], qbytes + qguard); hipMemset(qkv[k], 0, qbytes + qguard); }
    float* bias = (float*)dfill(768 * 4, false, 6); float* ada = (float*)dfill(6 * 256 * 4, false, 8);
    DitChainP c{}; c.heads = 2; c.rows_per_batch = N; c.X = X; c.Wp = Wp; c.W1 = W1; c.W2 = W2; c.Wq = Wq; c.bp = bias; c.b1 = bias; c.b2 = bias; c.bq = bias;
    c.ada = ada; c.next_shift = ada; c.next_scale = ada + 256; c.next_step_stride = 0; c.Npad = Npad; c.qscale = 0.088f * 1.4427f; c.M = B * N; c.B = B;
    c.O = (const float*)O; c.ksplit = 1; c.o_sstride = 0; c.o_lp = o_lp; c.Qh = qkv[0]; c.Kh = qkv[1]; c.Vt = qkv[2];
    setenv("DEX_ROWCHAIN64", "2", 1);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> keepX; std::vector<unsigned short> keepQ[3];
    for (int form = 0; form < 2; ++form) {
    setenv("DEX_ROWCHAIN64P", form ? "1" : "0", 1);
    for (int i = 0; i < 3; ++i) launch_dit_rowchain(c, 0);
    hipDeviceSynchronize();
    const int iters = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch_dit_rowchain(c, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * B * N * (256.0 * 256 + 2 * 256.0 * 512 + 256.0 * 768);
    printf("%s N=%d B=%d o_lp=%d: %.2f us per launch (back to back), %.1f TFLOP/s\n", g_last_symbol, N, B, o_lp, ms * 1e3 / iters, fl / (ms * 1e-3 / iters) / 1e12);
    // checksum of one launch from a fixed input (A/B of kernel versions: must be bitwise equal)
    hipMemcpy(X, X0, xb, hipMemcpyDeviceToDevice);
    launch_dit_rowchain(c, 0); hipDeviceSynchronize();
    {
        std::vector<unsigned> hx(xb / 4); hipMemcpy(hx.data(), X, xb, hipMemcpyDeviceToHost);
        unsigned long long s = 0; for (size_t i = 0; i < hx.size(); ++i) s = s * 1000003ull + hx[i];
        unsigned long long sq[3];
        for (int k = 0; k < 3; ++k) { std::vector<unsigned short> hq((size_t)B * 2 * Npad * 128); hipMemcpy(hq.data(), qkv[k], hq.size() * 2, hipMemcpyDeviceToHost);
            unsigned long long t = 0; const long tiles = (N + 31) / 32;
            for (int bh = 0; bh < B * 2; ++bh) for (long e = 0; e < tiles * 4096; ++e) t = t * 1000003ull + hq[(size_t)bh * Npad * 128 + e];
            { std::vector<unsigned short> g(qguard / 2); hipMemcpy(g.data(), (char*)qkv[k] + qbytes, qguard, hipMemcpyDeviceToHost); for (auto u : g) if (u) { printf("  !! write behind the operand buffer %d\n", k); break; } }
            sq[k] = t; }
        float* fx = (float*)hx.data(); double m = 0; for (size_t i = 0; i < hx.size(); ++i) m = std::max(m, (double)fabsf(fx[i]));
        printf("  checksum X %016llx q %016llx k %016llx vT %016llx  |X|max %.4f\n", s, sq[0], sq[1], sq[2], m);
        // the two forms against each other (not bitwise: the LayerNorm statistics are summed in another order)
        std::vector<unsigned short> hq[3];
        for (int k = 0; k < 3; ++k) { hq[k].resize((size_t)B * 2 * Npad * 128); hipMemcpy(hq[k].data(), qkv[k], hq[k].size() * 2, hipMemcpyDeviceToHost); }
        if (form == 0) { keepX.assign(fx, fx + hx.size()); for (int k = 0; k < 3; ++k) keepQ[k] = hq[k]; }
        else {
            double dx = 0; size_t nbad = 0;
            for (size_t i = 0; i < hx.size(); ++i) { const double d = fabs((double)fx[i] - keepX[i]); if (!(d <= 1e30)) ++nbad; else dx = std::max(dx, d); }
            printf("  p form vs round-3 form: X max|d| %.3e (%zu non-finite)", dx, nbad);
            auto bf = [](unsigned short u) { unsigned v = (unsigned)u << 16; float f; memcpy(&f, &v, 4); return (double)f; };
            const long tiles = (N + 31) / 32;
            for (int k = 0; k < 3; ++k) { double dq = 0, mq = 0; size_t nd = 0, tot = 0;
                for (int bh = 0; bh < B * 2; ++bh) for (long e = 0; e < tiles * 4096; ++e) { const size_t ix = (size_t)bh * Npad * 128 + e; ++tot;
                    const double u = bf(hq[k][ix]), v = bf(keepQ[k][ix]); if (hq[k][ix] != keepQ[k][ix]) ++nd; dq = std::max(dq, fabs(u - v)); mq = std::max(mq, fabs(v)); }
                printf("  %s: %zu of %zu differ, max|d| %.3e (|v|max %.3f)", k == 0 ? "q" : k == 1 ? "k" : "vT", nd, tot, dq, mq); }
            printf("\n");
        }
    }
#ifdef DEX_TIMING
    long long* dbg; const int nb = B * ((N + 63) / 64); hipMalloc(&dbg, (size_t)nb * 16 * 8); hipMemset(dbg, 0, (size_t)nb * 16 * 8);
    c.dbg = dbg; launch_dit_rowchain(c, 0); hipDeviceSynchronize(); c.dbg = nullptr;
    std::vector<long long> h((size_t)nb * 16); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0], t1 = 0; double ph[9] = {0};
    for (int bl = 0; bl < nb; ++bl) { t0 = std::min(t0, h[bl * 16]); t1 = std::max(t1, h[bl * 16 + 9]); for (int k = 0; k < 9; ++k) ph[k] += (double)(h[bl * 16 + k + 1] - h[bl * 16 + k]); }
    printf("  first start -> last end: %.2f us; mean per workgroup (us): ", (t1 - t0) * 0.01);
    const char* nm1[9] = {"stage O", "proj", "LN1", "fc1a+GELU", "fc2a", "fc1b+GELU", "fc2b+x2", "LN2", "qkv"};
    const char* nm2[9] = {"stage O", "proj+stats", "LN1", "fc1+GELU", "fc2+x2", "stats", "LN2", "qkv", "-"};
    const char** nm = form ? nm2 : nm1;
    double tot = 0; for (int k = 0; k < 9; ++k) { printf("%s=%.2f ", nm[k], ph[k] / nb * 0.01); tot += ph[k] / nb * 0.01; }
    printf("total=%.2f\n", tot);
#endif
    }
    return 0;
}
