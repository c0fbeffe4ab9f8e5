// tools/tvchainbench.hip — the one-launch TV adaptor (attention_bf16.hip tv_chain_kernel) alone at configs[2]'s shape: time per launch
// and, built with -DTVC_SKIP=<mask>, the same launch with phases left out (1 attention tiles, 2 result I/O, 4 x loads).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTVC_SKIP=1] -DDEX_LP_NS_OVERRIDE=tvb -I include -I dex_tts_amd/csrc tools/tvchainbench.hip dex_tts_amd/csrc/attention_bf16.hip -o tools/tvchainbench_x
#include <hip/hip_runtime.h>
#ifndef TVC_SKIP
#define TVC_SKIP 0
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../dex_tts_amd/csrc/kernels.h"
#include "../dex_tts_amd/csrc/kernels_lp.h"
namespace dex {
thread_local const char* g_last_symbol = nullptr;
int knob(const char* name) { const char* e = getenv(name); return e ? atoi(e) : KNOB_UNSET; }
}
using namespace dex;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, Wm = argc > 2 ? atoi(argv[2]) : 128, Hm = 40, C = 128, Ts = 348;
    const int fold = argc > 3 ? atoi(argv[3]) : 0;       // 1: the folded form (tv_chain_kernel<true>: no projections in the launch)
    const long npix = (long)Hm * Wm;
    const int Nk = Ts + 1, NkPad = (Nk + 63) / 64 * 64;
    float *X, *out, *mask, *beff, *K, *V; void *Weff, *Wl, *Kp, *VTp; gnfix_t* stats; int* lens;
    hipMalloc(&X, B * npix * C * 4); hipMalloc(&out, B * npix * C * 4); hipMalloc(&mask, (size_t)B * Wm * 2 * 4); hipMalloc(&beff, B * C * 4);
    hipMalloc(&K, (size_t)B * Nk * C * 4); hipMalloc(&V, (size_t)B * Nk * C * 4);
    hipMalloc(&Weff, (size_t)B * C * C * 2); hipMalloc(&Wl, (size_t)C * C * 2); hipMalloc(&Kp, (size_t)B * NkPad * C * 2); hipMalloc(&VTp, (size_t)B * NkPad * C * 2);
    hipMalloc(&stats, (size_t)B * C * GN_SLOTS * 2 * 8); hipMalloc(&lens, B * 4);
    hipMemset(X, 0x3c, B * npix * C * 4); hipMemset(mask, 0x3f, (size_t)B * Wm * 2 * 4); hipMemset(beff, 0, B * C * 4);
    hipMemset(K, 0x3c, (size_t)B * Nk * C * 4); hipMemset(V, 0x3c, (size_t)B * Nk * C * 4);
    hipMemset(Weff, 0x2c, (size_t)B * C * C * 2); hipMemset(Wl, 0x2c, (size_t)C * C * 2); hipMemset(stats, 0, (size_t)B * C * GN_SLOTS * 2 * 8);
    { int* h = (int*)malloc(B * 4); for (int i = 0; i < B; ++i) h[i] = Ts - 3 * i; hipMemcpy(lens, h, B * 4, hipMemcpyHostToDevice); free(h); }
    TvKvPrepP kp{K, V, (long)Nk * C, Nk, NkPad, Kp, VTp, B};
    TvChainP tc{X, C, 0, npix * C, (int)npix, Wm, mask, 2, (long)Wm * 2, Weff, 0, beff, Wl, 0, Kp, VTp, NkPad, Nk, lens, 1, 0.0883883f, out, stats, B};
    if (fold) { float* xm; hipMalloc(&xm, B * C * 4); hipMemset(xm, 0, B * C * 4); tc.xmean = xm; tc.scale = 1.f; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) { tvb::launch_tv_kv_prep(kp, 0); tvb::launch_tv_chain(tc, 0); }
    hipEventRecord(e0, 0);
    const int n = 20;
    for (int it = 0; it < n; ++it) tvb::launch_tv_chain(tc, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / n;
    const double gf = (4.0 * B * npix * C * C + 4.0 * B * npix * (double)Nk * C) * 1e-9, mb = 8.0 * B * npix * C * 1e-6;
    printf("tv_chain fold=%d TVC_SKIP=%d B=%d %dx%d: %8.2f us per launch  %.0f TFLOP/s  %.2f TB/s of the %.0f MB that have to move (%s)\n", fold, TVC_SKIP, B, Hm, Wm, us, gf / us * 1e-3,
           mb / us * 1e-6 * 1e6 * 1e-6, mb, hipGetErrorString(hipGetLastError()));
    (void)fold;
    return 0;
}
