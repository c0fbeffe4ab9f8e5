// kbench: isolated timing of libdexamd kernels (steady state, back-to-back launches of the same kernel).
//   hipcc --offload-arch=gfx950 -O3 -I dex_tts_amd/csrc tools/kbench.hip -L dex_tts_amd/lib -ldexamd -Wl,-rpath,$PWD/dex_tts_amd/lib -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#include <algorithm>
#include "kernels.h"
#include "kernels_lp.h"          // the reduced-precision launchers, bf16 build (namespace dex::bf16)
using namespace dex;
using namespace dex::bf16;

static float* dalloc(size_t n, float val = 0.01f) {
    float* p; hipMalloc(&p, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = val * (float)((i * 2654435761u) % 1000) / 1000.f - val * 0.5f;
    hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}
static void timeit(const char* name, int iters, double flops, double bytes, std::function<void()> f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    printf("%-44s %8.2f us  %8.2f TF/s %8.1f GB/s\n", name, us, flops / us * 1e-6, bytes / us * 1e-3);
}

// ---- pure-register MFMA chains: the practical matrix-core ceiling on this part (clock under load included)
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void mfma_f32_probe(float* out, int iters) {
    f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-6f, y = 1.0f + threadIdx.x * 1e-7f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;
}
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void mfma_bf16_probe(float* out, int iters) {
    f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8_t x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(threadIdx.x * 1e-3f); y[j] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;
}

// ---- per-CU L2 streaming ceiling: NB workgroups of 512 threads each read the same `bytes` (hot in L2), 16 B per lane,
// UNR independent loads in flight per wave
template <int UNR>
__global__ __launch_bounds__(512) void l2_stream_probe(const uint4* src, long n16, unsigned* out) {
    const int tid = threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long i = tid; i < n16; i += 512 * UNR) {
        uint4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = src[min(i + 512L * u, n16 - 1)];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

int main() {
    {
        const long bytes = 1441792;   // the row chain's weights per workgroup
        uint4* src; hipMalloc(&src, bytes); hipMemset(src, 1, bytes); unsigned* o; hipMalloc(&o, 4);
        for (int nb : {1, 21, 84, 256}) {
            char nm[80];
            snprintf(nm, 80, "L2 stream 1.4 MB per wg, %d wgs, 8 loads/wave", nb); timeit(nm, 20, 0, (double)bytes * nb, [&] { hipLaunchKernelGGL(l2_stream_probe<8>, dim3(nb), dim3(512), 0, 0, src, bytes / 16, o); });
            snprintf(nm, 80, "L2 stream 1.4 MB per wg, %d wgs, 32 loads/wave", nb); timeit(nm, 20, 0, (double)bytes * nb, [&] { hipLaunchKernelGGL(l2_stream_probe<32>, dim3(nb), dim3(512), 0, 0, src, bytes / 16, o); });
        }
    }
    {
        float* o; hipMalloc(&o, 4);
        const int it = 4096;
        const double wf32 = 4.0 * it * 2.0 * 32 * 32 * 2, wbf = 4.0 * it * 2.0 * 32 * 32 * 16;
        timeit("MFMA probe fp32 32x32x2  1 wave/SIMD", 5, wf32 * 1024 * 4, 0, [&] { hipLaunchKernelGGL(mfma_f32_probe<1>, dim3(1024), dim3(256), 0, 0, o, it); });
        timeit("MFMA probe fp32 32x32x2  2 waves/SIMD", 5, wf32 * 1024 * 8, 0, [&] { hipLaunchKernelGGL(mfma_f32_probe<2>, dim3(1024), dim3(512), 0, 0, o, it); });
        timeit("MFMA probe bf16 32x32x16 1 wave/SIMD", 5, wbf * 1024 * 4, 0, [&] { hipLaunchKernelGGL(mfma_bf16_probe<1>, dim3(1024), dim3(256), 0, 0, o, it); });
        timeit("MFMA probe bf16 32x32x16 2 waves/SIMD", 5, wbf * 1024 * 8, 0, [&] { hipLaunchKernelGGL(mfma_bf16_probe<2>, dim3(1024), dim3(512), 0, 0, o, it); });
    }
    const int B = 1, T = 512;
    // ---- attention at scale: B=32 N=1300 (DEX-VCTK T=256) and B=1 N=5010 (long form)
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int Bb = cfg == 0 ? 32 : 1, N = cfg == 0 ? 1300 : 5010;
        float* qkv = dalloc((size_t)Bb * N * 768, 1.0f);
        float* o = dalloc((size_t)Bb * N * 256);
        AttnP a{}; a.Q = qkv; a.ldq = 768; a.qb = (long)N * 768; a.K = qkv + 256; a.ldk = 768; a.kb = a.qb; a.V = qkv + 512; a.ldv = 768; a.vb = a.qb;
        a.O = o; a.ldo = 256; a.ob = (long)N * 256; a.Nq = N; a.Nk = N; a.heads = 2; a.scale = 0.088f; a.B = Bb;
        char nm[64];
        snprintf(nm, 64, "attention fp32 B=%d N=%d", Bb, N); timeit(nm, 10, 4.0 * Bb * N * (double)N * 256, 16.0 * Bb * N * 256, [&] { launch_attention(a, 0, 0); });
#ifdef DEX_TIMING
        {
            const int nb = ((N + 31) / 32) * 2 * Bb;
            long long* dbg; hipMalloc(&dbg, nb * 64); hipMemset(dbg, 0, nb * 64);
            a.dbg = dbg; launch_attention(a, 0, 0); hipDeviceSynchronize(); a.dbg = nullptr;
            std::vector<long long> h(nb * 8); hipMemcpy(h.data(), dbg, nb * 64, hipMemcpyDeviceToHost);
            for (int bl : {0, nb / 2, nb - 1}) { long long* d = &h[bl * 8];
                printf("   wg %4d wave0: K stage=%lld S^T=%lld softmax=%lld PV=%lld loop total=%lld cycles, wall=%lld x10ns -> %.2f GHz\n", bl, d[0], d[1], d[2], d[3], d[4], d[5], d[4] / (d[5] * 10.0)); }
        }
#endif
        snprintf(nm, 64, "attention bf16 B=%d N=%d", Bb, N); timeit(nm, 10, 4.0 * Bb * N * (double)N * 256, 16.0 * Bb * N * 256, [&] { launch_attention(a, 1, 0); });
        hipFree(qkv); hipFree(o);
    }
    // ---- attention N=650 (GeDEX) and N=2580 (DEX)
    for (int N : {650, 2580}) {
        float* qkv = dalloc((size_t)B * N * 768, 1.0f);
        float* o = dalloc((size_t)B * N * 256);
        AttnP a{}; a.Q = qkv; a.ldq = 768; a.qb = (long)N * 768; a.K = qkv + 256; a.ldk = 768; a.kb = a.qb; a.V = qkv + 512; a.ldv = 768; a.vb = a.qb;
        a.O = o; a.ldo = 256; a.ob = (long)N * 256; a.Nq = N; a.Nk = N; a.heads = 2; a.scale = 0.088f; a.B = B;
        char nm[64];
        snprintf(nm, 64, "attention fp32 N=%d", N); timeit(nm, 50, 4.0 * N * N * 256, 16.0 * N * 256, [&] { launch_attention(a, 0, 0); });
        snprintf(nm, 64, "attention bf16 N=%d", N); timeit(nm, 50, 4.0 * N * N * 256, 16.0 * N * 256, [&] { launch_attention(a, 1, 0); });
        float* o4 = dalloc((size_t)B * N * 256 * 4); float* ml = dalloc((size_t)B * N * 16);
        a.O = o4; a.ksplit = N == 650 ? 3 : 4; a.o_sstride = (long)B * N * 256; a.ml = ml;
        snprintf(nm, 64, "attention bf16 N=%d ksplit=%d", N, a.ksplit); timeit(nm, 50, 4.0 * N * N * 256, 16.0 * N * 256, [&] { launch_attention(a, 1, 0); });
#ifdef DEX_TIMING
        {
            const int nb = ((N + 31) / 32) * 2 * a.ksplit;
            long long* dbg; hipMalloc(&dbg, nb * 16 * 8); hipMemset(dbg, 0, nb * 16 * 8);
            a.dbg = dbg; launch_attention(a, 1, 0); hipDeviceSynchronize(); a.dbg = nullptr;
            std::vector<long long> h(nb * 16); hipMemcpy(h.data(), dbg, nb * 16 * 8, hipMemcpyDeviceToHost);
            long long t0 = 1LL << 62, te = 0;
            for (int bl = 0; bl < nb * 2; ++bl) if (h[bl * 8]) { t0 = std::min(t0, h[bl * 8]); te = std::max(te, h[bl * 8 + 6]); }
            printf("   blocks=%d span=%lld (10ns)\n", nb, te - t0);
            for (int bl : {0, 1, nb, nb + 1, 2 * nb - 2, 2 * nb - 1}) { long long* d = &h[bl * 8];
                printf("   blk %4d w%d: start+%lld  loads+q=%lld stage=%lld tile(s)=%lld barrier=%lld merge+store=%lld\n", bl / 2, bl & 1, d[0] - t0,
                       d[1] - d[0], d[3] - d[1], d[4] - d[3], d[5] - d[4], d[6] - d[5]); }
        }
#endif
        a.ksplit = 0; a.O = o;
    }
    // ---- conv3x3 64->64 @80x512 and 128->128 @40x256 (bf16 patch kernel and fp32 igemm)
    struct CS { int H, W, Cin, Cout; };
    for (CS c : {CS{80, 512, 64, 64}, CS{40, 256, 128, 128}, CS{40, 256, 256, 64}}) {
        const long npix = (long)c.H * c.W;
        float* x = dalloc(B * npix * c.Cin, 1.0f); float* y = dalloc(B * npix * c.Cout);
        float* w = dalloc(9L * c.Cin * c.Cout, 0.1f); float* bias = dalloc(c.Cout);
        unsigned short* wb; hipMalloc(&wb, 9L * c.Cin * c.Cout * 2); hipMemset(wb, 0, 9L * c.Cin * c.Cout * 2);
        float* mask = dalloc(B * T, 0.f); hipMemset(mask, 0, 4); // values irrelevant
        gnfix_t* st; hipMalloc(&st, 8 * 64 * 2 * 8 * B); hipMemset(st, 0, 8 * 64 * 2 * 8 * B);
        Conv3P p{}; p.X = x; p.ldx = c.Cin; p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.Cout = c.Cout; p.Wbf = wb; p.bias = bias; p.Y = y;
        p.mask = mask; p.mask_ws = 512 / c.W; p.mask_bstride = T; p.gn_stats = st; p.B = B;
        char nm[80];
        const double fl = 2.0 * npix * c.Cout * 9 * c.Cin, by = 4.0 * npix * (c.Cin + c.Cout);
        snprintf(nm, 80, "conv3x3 bf16 patch %d->%d @%dx%d", c.Cin, c.Cout, c.H, c.W); timeit(nm, 50, fl, by, [&] { launch_conv3x3_lp(p, 0); });
        p.gn_stats = nullptr;
        snprintf(nm, 80, "  .. same, no GN atomics"); timeit(nm, 50, fl, by, [&] { launch_conv3x3_lp(p, 0); });
        p.gn_stats = st;
        IGemmP g{}; g.A = x; g.lda = c.Cin; g.a_bstride = npix * c.Cin; g.Hi = c.H; g.Wi = c.W; g.Cin = c.Cin; g.KH = 3; g.KW = 3; g.sh = g.sw = 1; g.off_h = g.off_w = -1;
        g.step_h = g.step_w = 1; g.Ho = c.H; g.Wo = c.W; g.W = w; g.Wbf = wb; g.N = c.Cout; g.K = 9 * c.Cin; g.ksplit = 1; g.groups = 1; g.bias = bias;
        g.C = y; g.ldc = c.Cout; g.c_bstride = npix * c.Cout; g.OHf = c.H; g.OWf = c.W; g.osh = g.osw = 1; g.gate_nstride = 1; g.B = B; g.mask_bstride = T;
        g.gn_stats = st; g.gn_groups = 8; g.gn_cpg = c.Cout / 8;
        snprintf(nm, 80, "conv3x3 bf16 igemm %d->%d @%dx%d", c.Cin, c.Cout, c.H, c.W); timeit(nm, 50, fl, by, [&] { launch_igemm(g, 1, 0); });
        snprintf(nm, 80, "conv3x3 fp32 igemm %d->%d @%dx%d", c.Cin, c.Cout, c.H, c.W); timeit(nm, 50, fl, by, [&] { launch_igemm(g, 0, 0); });
    }
    // ---- DiT linear M=650 K=256 N=768 / K=512 N=256
    struct LS { int M, K, N; };
    for (LS l : {LS{650, 256, 768}, LS{650, 512, 256}, LS{650, 256, 512}, LS{40960, 64, 384}}) {
        float* x = dalloc((size_t)l.M * l.K, 1.0f); float* y = dalloc((size_t)l.M * l.N); float* w = dalloc((size_t)l.K * l.N, 0.1f);
        unsigned short* wb; hipMalloc(&wb, (size_t)l.K * l.N * 2); hipMemset(wb, 0, (size_t)l.K * l.N * 2);
        IGemmP g{}; g.A = x; g.lda = l.K; g.a_bstride = (long)l.M * l.K; g.Hi = 1; g.Wi = l.M; g.Cin = l.K; g.KH = g.KW = 1; g.sh = g.sw = 1; g.step_h = g.step_w = 1;
        g.Ho = 1; g.Wo = l.M; g.W = w; g.Wbf = wb; g.N = l.N; g.K = l.K; g.ksplit = 1; g.groups = 1; g.C = y; g.ldc = l.N; g.c_bstride = (long)l.M * l.N;
        g.OHf = 1; g.OWf = l.M; g.osh = g.osw = 1; g.gate_nstride = 1; g.B = 1;
        char nm[80];
        const double fl = 2.0 * l.M * l.K * l.N, by = 4.0 * (l.M * l.K + l.M * l.N + l.K * l.N);
        snprintf(nm, 80, "linear bf16 M=%d K=%d N=%d", l.M, l.K, l.N); timeit(nm, 50, fl, by, [&] { launch_igemm(g, 1, 0); });
        snprintf(nm, 80, "linear fp32 M=%d K=%d N=%d", l.M, l.K, l.N); timeit(nm, 50, fl, by, [&] { launch_igemm(g, 0, 0); });
    }
    // ---- DiT attention on fragment-tiled bf16 operands: batch regime (shared-KV kernel) and key-split regime
    {
        struct AC { int B, N, ks; };
        for (AC c : {AC{32, 1300, 1}, AC{32, 650, 1}, AC{1, 650, 3}, AC{1, 2580, 4}}) {
            const int Npad = (c.N + 31) / 32 * 32;
            const size_t el = (size_t)c.B * 2 * Npad * 128;
            unsigned short *q, *k, *v; hipMalloc(&q, el * 2); hipMalloc(&k, el * 2); hipMalloc(&v, el * 2);
            std::vector<unsigned short> hbuf(el);
            for (size_t i = 0; i < el; ++i) hbuf[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 20 & 0x1ff) - ((i & 1) ? 0 : 0x8000));   // ~ +-0.01..0.03
            hipMemcpy(q, hbuf.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(k, hbuf.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(v, hbuf.data(), el * 2, hipMemcpyHostToDevice);
            float* O = dalloc((size_t)c.B * c.N * 256 * c.ks); float* ml = dalloc((size_t)c.B * c.N * 4 * c.ks + 16);
            AttnDirectP a{q, k, v, c.N, Npad, c.B, O, (long)c.B * c.N * 256, c.ks > 1 ? ml : nullptr, c.ks, nullptr};
            char nm[80]; snprintf(nm, 80, "attention direct bf16 B=%d N=%d ksplit=%d", c.B, c.N, c.ks);
            timeit(nm, 20, 4.0 * c.B * c.N * (double)c.N * 256, 2.0 * 3 * el + 4.0 * c.B * c.N * 256 * c.ks, [&] { launch_attention_direct(a, 0); });
#ifdef DEX_TIMING
            if (c.ks == 1) {
                const int nb = ((c.N + 31) / 32 + 3) / 4 * 2 * c.B;
                long long* dbg; hipMalloc(&dbg, nb * 64); hipMemset(dbg, 0, nb * 64);
                a.dbg = dbg; launch_attention_direct(a, 0); hipDeviceSynchronize(); a.dbg = nullptr;
                std::vector<long long> h(nb * 8); hipMemcpy(h.data(), dbg, nb * 64, hipMemcpyDeviceToHost);
                for (int bl : {0, nb / 2, nb - 1}) { long long* d = &h[bl * 8];
                    printf("   wg %4d: S^T=%lld softmax=%lld PV=%lld copy+barrier=%lld total=%lld cycles, wall=%lld x10ns -> %.2f GHz\n", bl, d[0], d[1], d[2], d[3], d[4], d[5], d[4] / (d[5] * 10.0)); }
            }
#endif
        }
    }
    // ---- linear-attention context pass + merge at 80x512 (C=64): pixels per workgroup trade-off
    {
        const int npix = 40960, C = 64;
        float* x = dalloc((size_t)npix * C, 1.0f);
        unsigned short* wkv; hipMalloc(&wkv, 256 * C * 2); hipMemset(wkv, 0, 256 * C * 2);
        float* pm = dalloc(4 * 320 * 32); float* ps = dalloc(4 * 320 * 32, 1.0f); float* pc = dalloc(4 * 320 * 1024);
        float* wout = dalloc(C * 128, 0.1f); float* g = dalloc(1, 1.0f); void* w2; hipMalloc(&w2, C * 128 * 2);
        for (int nsub : {1, 2, 4}) {
            const int nblk = (npix + 128 * nsub - 1) / (128 * nsub);
            LinKvCtxP k{x, C, 0, (long)npix * C, npix, C, wkv, nsub, nblk, pm, ps, pc, 1};
            LinMergeP mg{pm, ps, pc, nblk, wout, g, C, w2, 1};
            char nm[80];
            snprintf(nm, 80, "linattn kvctx 80x512 nsub=%d (%d wgs)", nsub, nblk); timeit(nm, 50, 2.0 * npix * (256.0 * C + 128 * 32), 4.0 * npix * C, [&] { launch_linattn_kvctx(k, 0); });
            snprintf(nm, 80, "linattn merge nblk=%d", nblk); timeit(nm, 50, 0, 4.0 * nblk * 4 * 1088, [&] { launch_linattn_merge(mg, 0); });
        }
    }
    // ---- DiT row chain (M=650): weights hot in L2 (back-to-back) vs evicted by a 64 MB sweep between launches
    {
        const int M = 650;
        float* O = dalloc((size_t)M * 256 * 4, 1.0f); float* X = dalloc((size_t)M * 256, 1.0f); float* ml = dalloc((size_t)M * 16, 1.0f);
        unsigned short *Wp, *W1, *W2, *Wq, *qh, *kh, *vt;
        hipMalloc(&Wp, 256 * 256 * 2); hipMalloc(&W1, 256 * 512 * 2); hipMalloc(&W2, 512 * 256 * 2); hipMalloc(&Wq, 256 * 768 * 2);
        hipMemset(Wp, 0, 256 * 256 * 2); hipMemset(W1, 0, 256 * 512 * 2); hipMemset(W2, 0, 512 * 256 * 2); hipMemset(Wq, 0, 256 * 768 * 2);
        hipMalloc(&qh, 672 * 256 * 2); hipMalloc(&kh, 672 * 256 * 2); hipMalloc(&vt, 672 * 256 * 2);
        float* bias = dalloc(768, 0.1f); float* ada = dalloc(6 * 256, 0.1f);
        DitChainP c{}; c.O = O; c.ksplit = 3; c.o_sstride = (long)M * 256; c.ml = ml; c.heads = 2; c.rows_per_batch = M; c.X = X;
        c.Wp = Wp; c.W1 = W1; c.W2 = W2; c.Wq = Wq; c.bp = bias; c.b1 = bias; c.b2 = bias; c.bq = bias; c.ada = ada;
        c.next_shift = ada; c.next_scale = ada + 256; c.next_step_stride = 0; c.Qh = qh; c.Kh = kh; c.Vt = vt; c.Npad = 672; c.qscale = 0.088f; c.M = M; c.B = 1;
        const double wel = 256.0 * 256 + 2.0 * 256 * 512 + 3.0 * 256 * 256;
        timeit("dit_rowchain M=650 (weights hot)", 50, 2.0 * M * wel, 2.0 * wel, [&] { launch_dit_rowchain(c, 0); });
#ifdef DEX_TIMING
        {
            long long* dbg; hipMalloc(&dbg, 512 * 8); hipMemset(dbg, 0, 512 * 8);
            c.dbg = dbg; launch_dit_rowchain(c, 0); hipDeviceSynchronize(); c.dbg = nullptr;
            std::vector<long long> h(512); hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost);
            for (int bl : {0, 10, 20}) { long long* d = &h[bl * 8];
                printf("   blk %2d: stageO=%lld S1=%lld LN1=%lld S2=%lld S3=%lld LN2=%lld S4=%lld total=%lld (10ns)\n", bl,
                       d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5], d[7] - d[6], d[7] - d[0]);
                printf("            first qkv tile: weights wait=%lld  A reads+16 MFMA=%lld  store_qkv_tile=%lld (10ns)\n", h[256 + bl * 4], h[256 + bl * 4 + 1], h[256 + bl * 4 + 2]); }
        }
#endif
        float* big = dalloc(16 << 20, 1.0f); float* big2 = dalloc(16 << 20, 1.0f);
        timeit("  64 MB d2d copy alone", 20, 0, 128e6, [&] { hipMemcpyAsync(big2, big, 64 << 20, hipMemcpyDeviceToDevice, 0); });
        timeit("  copy + dit_rowchain (weights cold)", 20, 2.0 * M * wel, 2.0 * wel, [&] { hipMemcpyAsync(big2, big, 64 << 20, hipMemcpyDeviceToDevice, 0); launch_dit_rowchain(c, 0); });
        c.qkv_only = 1;
        timeit("dit_rowchain qkv-only (hot)", 50, 2.0 * M * 3 * 256 * 256, 0, [&] { launch_dit_rowchain(c, 0); });
    }
    // ---- trivial kernel floor
    int* stp; hipMalloc(&stp, 4);
    timeit("step_inc (launch floor)", 200, 0, 0, [&] { launch_step_inc(stp, 0); });
    return 0;
}
