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
#ifdef RCA_TIMING
    long long* dbgbuf; hipMalloc(&dbgbuf, (size_t)B * ((N + 63) / 64) * 4 * 32 * 8); hipMemset(dbgbuf, 0, (size_t)B * ((N + 63) / 64) * 4 * 32 * 8);
    c.dbg = dbgbuf;
#endif
    const char* mode_name[3] = {"block + next qkv", "last block (no qkv)", "qkv only (first launch)"};
    for (int mode = 0; mode < 3; ++mode) {
    DitChainP m = c;
    if (mode == 1) { m.next_shift = m.next_scale = nullptr; m.Wq = nullptr; m.bq = nullptr; }
    if (mode == 2) m.qkv_only = 1;
    std::vector<float> keepX; std::vector<unsigned short> keepQ[3];
    printf("-- %s\n", mode_name[mode]);
    for (int form = 0; form < 2; ++form) {
    setenv("DEX_ROWCHAIN64A", form ? "1" : "0", 1);
    for (int i = 0; i < 3; ++i) launch_dit_rowchain(m, 0);
    hipDeviceSynchronize();
    const int iters = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) launch_dit_rowchain(m, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * B * N * ((mode == 2 ? 0.0 : 256.0 * 256 + 2 * 256.0 * 512) + (mode == 1 ? 0.0 : 256.0 * 768));
    printf("%s N=%d B=%d o_lp=%d: %.2f us per launch (back to back), %.1f TFLOP/s\n", g_last_symbol, N, B, o_lp, ms * 1e3 / iters, fl / (ms * 1e-3 / iters) / 1e12);
    // one launch from a fixed input; the two forms against each other (not bitwise: the LayerNorm statistics are summed in another order)
    hipMemcpy(X, X0, xb, hipMemcpyDeviceToDevice);
    for (int k = 0; k < 3; ++k) hipMemset(qkv[k], 0, qbytes + qguard);
    launch_dit_rowchain(m, 0); hipDeviceSynchronize();
    {
        std::vector<float> hx(xb / 4); hipMemcpy(hx.data(), X, xb, hipMemcpyDeviceToHost);
        std::vector<unsigned short> hq[3];
        for (int k = 0; k < 3; ++k) { hq[k].resize((qbytes + qguard) / 2); hipMemcpy(hq[k].data(), qkv[k], qbytes + qguard, hipMemcpyDeviceToHost);
            for (size_t e = qbytes / 2; e < hq[k].size(); ++e) if (hq[k][e]) { printf("  !! write behind the operand buffer %d\n", k); break; } }
        double mx = 0; for (float v : hx) mx = std::max(mx, (double)fabsf(v));
        if (form == 0) { keepX = hx; for (int k = 0; k < 3; ++k) keepQ[k] = hq[k]; printf("  |X|max %.4f\n", mx); }
        else {
            double dx = 0; size_t nbad = 0, nd = 0;
            for (size_t i = 0; i < hx.size(); ++i) { const double d = fabs((double)hx[i] - keepX[i]); if (!(d <= 1e30)) ++nbad; else { dx = std::max(dx, d); if (d > 1e-3) ++nd; } }
            printf("  a form vs round-3 form: X max|d| %.3e (%zu non-finite, %zu above 1e-3)", dx, nbad, nd);
            auto bf = [](unsigned short u) { unsigned v = (unsigned)u << 16; float f; memcpy(&f, &v, 4); return (double)f; };
            for (int k = 0; k < 3; ++k) { double dq = 0, mq = 0; size_t ndq = 0, big = 0;
                for (size_t ix = 0; ix < qbytes / 2; ++ix) { const double u = bf(hq[k][ix]), v = bf(keepQ[k][ix]); if (hq[k][ix] != keepQ[k][ix]) ++ndq;
                    const double d = fabs(u - v); if (!(d <= 1e30)) { ++big; continue; } dq = std::max(dq, d); mq = std::max(mq, fabs(v)); if (d > 0.02 * std::max(1.0, fabs(v))) ++big; }
                printf("  %s: %zu of %zu differ (%zu by more than 2 %%), max|d| %.3e (|v|max %.3f)", k == 0 ? "q" : k == 1 ? "k" : "vT", ndq, qbytes / 2, big, dq, mq); }
            printf("\n");
        }
    }
#ifdef RCA_TIMING
    if (form == 1 && mode == 0) {
        m.dbg = dbgbuf; launch_dit_rowchain(m, 0); hipDeviceSynchronize();
        const int nb = B * ((N + 63) / 64);
        std::vector<long long> h((size_t)nb * 4 * 32); hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
        const char* nm[] = {RCA_ASM_FULL_STAMPS};
        const int ns = (int)(sizeof(nm) / sizeof(nm[0]));
        std::vector<double> ph(ns, 0.0); long long t0 = h[0], t1 = 0;
        for (int r = 0; r < nb * 4; ++r) { t0 = std::min(t0, h[(size_t)r * 32]); t1 = std::max(t1, h[(size_t)r * 32 + ns - 1]); for (int k = 1; k < ns; ++k) ph[k] += (double)(h[(size_t)r * 32 + k] - h[(size_t)r * 32 + k - 1]); }
        { double r1 = 0, r2 = 0; int n1 = 0, n2 = 0; for (int r = 0; r < nb * 4; ++r) { const double d = (double)(h[(size_t)r * 32 + ns - 1] - h[(size_t)r * 32]); if (r < 256 * 4) { r1 += d; ++n1; } else { r2 += d; ++n2; } }
          printf("  start -> end per wave: workgroups 0..255 %.0f, the later ones %.0f cycles\n", r1 / std::max(n1, 1), r2 / std::max(n2, 1)); }
        { double pr = 0; for (int r = 0; r < nb * 4; ++r) pr += (double)(h[(size_t)r * 32] - h[(size_t)r * 32 + 31]); printf("  kernel entry -> first stamp of the streams (parameter staging in C++): %.0f cycles\n", pr / (nb * 4)); }
        double tot = 0; printf("  s_memtime per wave (mean over %d waves, cycles of its clock):\n   ", nb * 4);
        for (int k = 1; k < ns; ++k) { printf("  %s=%.0f", nm[k], ph[k] / (nb * 4)); tot += ph[k] / (nb * 4); }
        printf("\n    total=%.0f  (first start -> last end %lld)\n", tot, t1 - t0);
    }
#endif
    }
    }
    return 0;
}
#if 0
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
#endif
