// tools/convbench.hip — the 64->64 convolution kernels alone at batch size: timing + (with -DDEX_TIMING) the phase
// cycle counters of the strip-streaming kernel.  Build: see tools/convbench.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../dex_tts_amd/csrc/kernels.h"
#include "../dex_tts_amd/csrc/kernels_lp.h"      // reduced-precision launchers, bf16 build (namespace dex::bf16)
using namespace dex;
using namespace dex::bf16;
namespace dex {
thread_local const char* g_last_symbol = nullptr;                         // defined in lp_dispatch.hip in the library build
int knob(const char* name) { const char* e = getenv(name); return e ? atoi(e) : KNOB_UNSET; }   // (dex_api.hip in the library build)
}
static float* dalloc(size_t n, int fill = 0) { float* p; hipMalloc(&p, n * 4); hipMemset(p, fill, n * 4); return p; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = 80, W = 512, C = 64;
    const long npix = (long)H * W;
    float* x = dalloc(B * npix * C); float* y = dalloc(B * npix * C); float* res = dalloc(B * npix * C); float* xout = dalloc(B * npix * C);
    unsigned short* wb; hipMalloc(&wb, 9L * C * C * 2); hipMemset(wb, 0, 9L * C * C * 2);
    float* bias = dalloc(C); float* mask = dalloc((size_t)B * W, 0x3f); gnfix_t* st = (gnfix_t*)dalloc(8 * 64 * 2 * 2 * B); gnfix_t* st2 = (gnfix_t*)dalloc(8 * 64 * 2 * 2 * B);
    float* gam = dalloc(C); float* bet = dalloc(C);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 6; ++variant) {
        Conv3P p{}; p.X = x; p.ldx = C; p.H = H; p.W = W; p.Cin = C; p.Cout = C; p.Wbf = wb; p.bias = bias; p.Y = y;
        p.mask = mask; p.mask_ws = 1; p.mask_bstride = W; p.gn_stats = st; p.B = B;
        const char* nm = "plain fp32->fp32";
        if (variant >= 1) { p.pro_stats = st2; p.pro_gamma = gam; p.pro_beta = bet; nm = "PRO fp32->fp32"; }
        if (variant == 2) { p.x_bf16 = 1; p.y_bf16 = 1; nm = "PRO bf16->bf16"; }
        if (variant == 3) { p.pro_res = res; p.pro_xout = xout; nm = "PRO2 fp32->fp32"; }
        if (variant == 4) { p.pro_res = res; p.pro_xout = xout; p.x_bf16 = 1; p.y_bf16 = 1; nm = "PRO2 bf16->bf16"; }
        if (variant == 5) { p.y_bf16 = 1; nm = "plain fp32->bf16"; }
        for (int mode = 0; mode < 3; ++mode) {
            setenv("DEX_CONV_STREAM", mode ? "1" : "0", 1);
            setenv("DEX_CONV_PP", mode == 2 ? "1" : "0", 1);
            for (int it = 0; it < 3; ++it) launch_conv3x3_lp(p, 0);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 20; ++it) launch_conv3x3_lp(p, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-18s %s: %8.2f us\n", nm, mode == 2 ? "pingpong" : mode ? "stream  " : "tile    ", ms * 1000 / 20);
        }
#ifdef DEX_TIMING
        {   // ping-pong form: per group (two per workgroup) cycles by role
            setenv("DEX_CONV_PP", "1", 1); setenv("DEX_CONV_STREAM", "1", 1);
            const int nb = 2 * 1024;
            long long* dbg; hipMalloc(&dbg, (size_t)nb * 64); hipMemset(dbg, 0, (size_t)nb * 64);
            p.dbg = dbg; launch_conv3x3_lp(p, 0); hipDeviceSynchronize(); p.dbg = nullptr;
            std::vector<long long> h((size_t)nb * 8); hipMemcpy(h.data(), dbg, (size_t)nb * 64, hipMemcpyDeviceToHost);
            double a[8] = {0}; int n = 0; for (int bl = 0; bl < nb; ++bl) if (h[(size_t)bl * 8 + 7]) { ++n; for (int k = 0; k < 8; ++k) a[k] += h[(size_t)bl * 8 + k]; }
            if (n) printf("   ping-pong, avg cycles per group (%d groups): mfma role %.0f | emit %.0f | convert %.0f | barrier wait %.0f | loop total %.0f\n",
                   n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[7] / n);
            hipFree(dbg);
        }
#endif
        setenv("DEX_CONV_PP", "0", 1);
#ifdef DEX_TIMING
        if (!conv3x3_stream_tiles(p)) {                    // small grid: the tile kernel's phase counters
            const int nb = (W / 32) * (H / 4) * B;
            long long* dbg; hipMalloc(&dbg, (size_t)nb * 64); hipMemset(dbg, 0, (size_t)nb * 64);
            p.dbg = dbg; launch_conv3x3_lp(p, 0); hipDeviceSynchronize(); p.dbg = nullptr;
            std::vector<long long> h((size_t)nb * 8); hipMemcpy(h.data(), dbg, (size_t)nb * 64, hipMemcpyDeviceToHost);
            double a[8] = {0}; for (int bl = 0; bl < nb; ++bl) for (int k = 0; k < 8; ++k) a[k] += h[(size_t)bl * 8 + k];
            printf("   tile kernel, avg cycles/wg (%d wgs): issue loads %.0f | GN coeffs + barrier %.0f | convert + LDS %.0f | nine taps %.0f | epilogue %.0f | total %.0f\n",
                   nb, a[0] / nb, a[1] / nb, a[2] / nb, a[3] / nb, a[4] / nb, a[7] / nb);
            hipFree(dbg);
        } else {
            const int tpw = conv3x3_stream_tiles(p);
            const int nb = (W / 32) * ((H / 8 + tpw - 1) / tpw) * B;
            long long* dbg; hipMalloc(&dbg, (size_t)nb * 64); hipMemset(dbg, 0, (size_t)nb * 64);
            p.dbg = dbg; launch_conv3x3_lp(p, 0); hipDeviceSynchronize(); p.dbg = nullptr;
            std::vector<long long> h((size_t)nb * 8); hipMemcpy(h.data(), dbg, (size_t)nb * 64, hipMemcpyDeviceToHost);
            double a[8] = {0}; for (int bl = 0; bl < nb; ++bl) for (int k = 0; k < 8; ++k) a[k] += h[(size_t)bl * 8 + k];
            printf("   avg cycles/wg (tpw=%d, %d wgs): setup %.0f | per wg total: load-issue %.0f mfma %.0f epilogue %.0f barrier1 %.0f convert+lds %.0f barrier2 %.0f | total %.0f\n",
                   tpw, nb, a[0] / nb, a[1] / nb, a[2] / nb, a[3] / nb, a[4] / nb, a[5] / nb, a[6] / nb, a[7] / nb);
            hipFree(dbg);
        }
#endif
    }
    return 0;
}
