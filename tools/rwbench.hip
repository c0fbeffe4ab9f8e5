// tools/rwbench.hip — the weights-in-registers 128 -> 128 convolution (conv3x3_regw.hip) alone at batch size: timing and, with
// -DDEX_TIMING, its per-wave phase cycle counters.  Build + run: tools/rwbench.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../dex_tts_amd/csrc/kernels.h"
#include "../dex_tts_amd/csrc/kernels_lp.h"
using namespace dex;
using namespace dex::bf16;
namespace dex { thread_local const char* g_last_symbol = nullptr; }
static float* dalloc(size_t n, int fill = 0) { float* p; hipMalloc(&p, n * 4); hipMemset(p, fill, n * 4); return p; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = 40, W = argc > 2 ? atoi(argv[2]) : 256, C = 128;
    const long npix = (long)H * W;
    float* x = dalloc(B * npix * C / 2); float* y = dalloc(B * npix * C / 2); float* res = dalloc(B * npix * C); float* xout = dalloc(B * npix * C);
    unsigned short* wf; hipMalloc(&wf, 9L * C * C * 2); hipMemset(wf, 0, 9L * C * C * 2);
    float* bias = dalloc(C); float* mask = dalloc((size_t)B * W, 0x3f); gnfix_t* st = (gnfix_t*)dalloc(8 * 64 * 2 * 2 * B); gnfix_t* st2 = (gnfix_t*)dalloc(8 * 64 * 2 * 2 * B);
    float* gam = dalloc(C); float* bet = dalloc(C);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        Conv3P p{}; p.X = x; p.ldx = C; p.H = H; p.W = W; p.Cin = C; p.Cout = C; p.Wbf = wf; p.Wfrag = wf; p.bias = bias; p.Y = y;
        p.mask = mask; p.mask_ws = 1; p.mask_bstride = W; p.gn_stats = st; p.B = B;
        p.pro_stats = st2; p.pro_gamma = gam; p.pro_beta = bet; p.x_bf16 = 1; p.y_bf16 = 1;
        const char* nm = "PRO  bf16->bf16";
        if (variant == 1) { p.pro_res = res; p.pro_xout = xout; nm = "PRO2 bf16->bf16"; }
        if (!conv3x3_regw_form(p)) { printf("%s: grid too small for the strip form\n", nm); continue; }
        for (int it = 0; it < 3; ++it) launch_conv3x3_regw(p, 0);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it) launch_conv3x3_regw(p, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / 20, gf = 2.0 * 9 * C * C * npix * B * 1e-9;
        printf("%-16s B=%d %dx%d: %8.2f us  %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", nm, B, H, W, us, gf / us * 1e-3 * 1e3, gf / us * 1e-3 * 1e3 / 25.0);
#ifdef DEX_TIMING
        {
            const int nb = 4096 * 4;
            long long* dbg; hipMalloc(&dbg, (size_t)nb * 64); hipMemset(dbg, 0, (size_t)nb * 64);
            p.dbg = dbg; launch_conv3x3_regw(p, 0); hipDeviceSynchronize(); p.dbg = nullptr;
            std::vector<long long> h((size_t)nb * 8); hipMemcpy(h.data(), dbg, (size_t)nb * 64, hipMemcpyDeviceToHost);
            double a[8] = {0}; int n = 0;
            for (int w = 0; w < nb; ++w) if (h[(size_t)w * 8 + 7]) { ++n; for (int k = 0; k < 8; ++k) a[k] += h[(size_t)w * 8 + k]; }
            if (n) printf("   avg cycles per wave (%d waves): mfma chain + row transform %.0f | statistics %.0f | barrier A %.0f | stage write %.0f | barrier B %.0f | copy-out %.0f | prologue %.0f | row loop %.0f\n",
                          n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n, a[7] / n);
            hipFree(dbg);
        }
#endif
    }
    return 0;
}
