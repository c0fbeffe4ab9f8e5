// tools/attnq64.hip — bench + check of the 64-queries-per-wave attention (dex_tts_amd/csrc/attention_q64.hip) against the shipped
// shared-ring kernel (library) on the batch / long-form shapes, with key splits merged on the host the way the row chain merges
// them, and (-DQ64_STAMP) the per-unit phase anatomy from s_memtime stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DQ64_STAMP] -I dex_tts_amd/csrc -I include tools/attnq64.hip \
//         -L dex_tts_amd/lib -ldexamd -Wl,-rpath,$PWD/dex_tts_amd/lib -o tools/attnq64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <functional>
#include <algorithm>
#define DEX_LP_NS_OVERRIDE q64tool
#include "../dex_tts_amd/csrc/attention_q64.hip"

using namespace dex;

static double timeit(const char* name, int iters, double flops, std::function<void()> f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    printf("%-44s %9.2f us  %8.1f TF/s  (%.3f of 2.5 PF)\n", name, us, flops / us * 1e-6, flops / us * 1e-6 / 2500.0);
    return us;
}
static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    struct AC { int B, N, bench; };
    std::vector<AC> cases = {{2, 1, 0}, {3, 33, 0}, {1, 64, 0}, {4, 70, 0}, {5, 129, 0}, {2, 257, 0}, {3, 520, 0}, {8, 300, 0},
                             {32, 650, 1}, {32, 1300, 1}, {8, 2580, 1}, {1, 5010, 1}, {32, 2580, 1}};
    const int only_case = argc > 1 ? atoi(argv[1]) : -1;
    const int spike = argc > 2 ? atoi(argv[2]) : 1;
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        if (only_case >= 0 && (int)ci != only_case) continue;
        const AC c = cases[ci];
        const int Npad = (c.N + 31) / 32 * 32;
        const size_t el = (size_t)c.B * 2 * Npad * 128;
        unsigned short *q, *k, *v; hipMalloc(&q, el * 2); hipMalloc(&k, el * 2); hipMalloc(&v, el * 2);
        std::vector<unsigned short> hq(el), hk(el), hv(el);
        unsigned long long st = 88172645463325252ull + ci;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
        for (size_t j = 0; j < el; ++j) { hq[j] = f2bf(rnd() * 0.35f); hk[j] = f2bf(rnd() * 0.9f); hv[j] = f2bf(rnd()); }
        if (spike) {
            // force the reference-maximum move: a few K fragments far larger than the rest (whole 16-byte lane chunks of some late tiles)
            for (int s = 0; s < 6; ++s) {
                const size_t t32 = (size_t)((Npad / 32) * (0.3 + 0.1 * s));
                const size_t base = ((size_t)(s % (c.B * 2)) * Npad / 32 + t32) * 32 * 128 + (size_t)(s * 7 % 64) * 8;
                for (int e = 0; e < 8 && base + e < el; ++e) hk[base + e] = f2bf(12.f + s);
            }
        }
        hipMemcpy(q, hq.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), el * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), el * 2, hipMemcpyHostToDevice);
        const size_t on = (size_t)c.B * c.N * 256;
        const int KSMAX = 8;
        float *O, *O2, *ml; hipMalloc(&O, on * 4); hipMalloc(&O2, on * 4 * KSMAX); hipMalloc(&ml, (size_t)KSMAX * c.B * 2 * c.N * 2 * 4);
        long long* dbg = nullptr;
        AttnDirectP a{q, k, v, c.N, Npad, c.B, O, (long)on, nullptr, 1, nullptr};
        a.o_lp = 0; a.xcd_map = 0;
        const double fl = 4.0 * c.B * c.N * (double)c.N * 256;
        printf("---- B=%d N=%d (%.2f GFLOP)\n", c.B, c.N, fl * 1e-9);
        hipMemset(O, 0, on * 4);
        if (c.bench) timeit("shipped attn_direct (library)", 20, fl, [&] { dex::launch_attention_direct(a, 1 /*bf16*/, 0); });
        else dex::launch_attention_direct(a, 1, 0);
        hipDeviceSynchronize();
        std::vector<float> r(on); hipMemcpy(r.data(), O, on * 4, hipMemcpyDeviceToHost);
        const int nt32 = (c.N + 31) / 32, nT = (nt32 + 1) / 2;
        const int auto_ks = dex::q64tool::attention_q64_ksplit(c.N, c.B, KSMAX);
        std::vector<int> splits = {1};
        if (nT >= 2) splits.push_back(2);
        if (nT >= 3) splits.push_back(3);
        if (auto_ks > 3) splits.push_back(auto_ks);
        // (key split, half plan): half_g = -1: whole units only; -2: the launcher's own plan; >= 0: that many whole groups, the rest as half units
        struct Var { int ks, hg; };
        std::vector<Var> vars;
        for (int ks : splits) vars.push_back({ks, -1});
        {
            int phg = 0, phn = 0;
            if (dex::q64tool::attention_q64_half_plan(c.N, c.B, &phg, &phn)) vars.push_back({1, -2});
            vars.push_back({1, 0});                                      // half units only
            if (nt32 > 8) vars.push_back({1, 1});
            if (nt32 > 16) vars.push_back({1, nt32 / 8 - 1});
        }
        for (const Var& var : vars) {
            const int ks = var.ks;
            AttnDirectP a2 = a; a2.O = O2; a2.ksplit = ks; a2.ml = ks > 1 ? ml : nullptr; a2.o_sstride = (long)on;
            const bool olp = getenv("Q64_OLP") && ks == 1;       // 16-bit output rows, as the product path at batch size
            if (olp) a2.o_lp = 1;
            if (var.hg == -1) a2.half_n = -1;
            else if (var.hg >= 0) { a2.half_g = var.hg; a2.half_n = (nt32 - 8 * var.hg + 3) / 4; if (a2.half_n <= 0) continue; }
#ifdef Q64_STAMP
            const int ng = (nt32 + 7) / 8, nunits = 2 * c.B * (ng * ks + 2 * ng + 8);        // (an upper bound of every plan's unit count)
            hipMalloc(&dbg, (size_t)nunits * 4 * 8 * 8); hipMemset(dbg, 0, (size_t)nunits * 4 * 8 * 8);
            a2.dbg = dbg;
#endif
            hipMemset(O2, 0, on * 4 * ks);
            char label[96];
            if (var.hg == -1) snprintf(label, sizeof label, "q64 ksplit=%d%s", ks, ks == auto_ks ? " (auto)" : "");
            else if (var.hg == -2) snprintf(label, sizeof label, "q64 half plan (launcher)");
            else snprintf(label, sizeof label, "q64 %d whole + %d half units", var.hg, a2.half_n);
            if (c.bench) timeit(label, 20, fl, [&] { dex::q64tool::launch_attention_q64(a2, 0); });
            else dex::q64tool::launch_attention_q64(a2, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("      %s: LAUNCH FAILED: %s\n", label, hipGetErrorString(hipGetLastError())); return 1; }
            std::vector<float> g(on * ks), hml((size_t)ks * c.B * 2 * c.N * 2);
            hipMemcpy(g.data(), O2, on * 4 * ks, hipMemcpyDeviceToHost);
            if (olp) {
                std::vector<unsigned short> h16(on); memcpy(h16.data(), g.data(), on * 2);
                for (size_t j = 0; j < on; ++j) { const unsigned u = (unsigned)h16[j] << 16; memcpy(&g[j], &u, 4); }
            }
            if (ks > 1) hipMemcpy(hml.data(), ml, hml.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, ref = 0; size_t bad = 0, nbad = 0; int ndiag = 0; std::vector<size_t> badrow((c.N + 31) / 32 + 1, 0), badd(4, 0);
            for (int b = 0; b < c.B; ++b)
                for (int n = 0; n < c.N; ++n)
                    for (int h = 0; h < 2; ++h) {
                        double M = -1e300, W = 0;
                        if (ks > 1) {
                            for (int s = 0; s < ks; ++s) M = std::max(M, (double)hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]);
                            for (int s = 0; s < ks; ++s) { const float* e = &hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]; W += e[1] * exp2((double)e[0] - M); }
                        }
                        for (int d = 0; d < 128; ++d) {
                            const size_t idx = ((size_t)b * c.N + n) * 256 + h * 128 + d;
                            double val;
                            if (ks == 1) val = g[idx];
                            else {
                                val = 0;
                                for (int s = 0; s < ks; ++s) { const float* e = &hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]; val += g[(size_t)s * on + idx] * e[1] * exp2((double)e[0] - M); }
                                val /= W;
                            }
                            const double dlt = fabs(val - r[idx]);
                            if (!(dlt <= 1e30)) ++bad;
                            if (getenv("Q64_DIAG") && !(dlt <= 2e-2) && ndiag < 24) { ++ndiag; printf("        diag b=%d n=%d h=%d d=%d got %.5g want %.5g\n", b, n, h, d, val, (double)r[idx]); }
                            if (!(dlt <= 2e-2)) { ++nbad; badrow[n >> 5]++; badd[d >> 5]++; }
                            mx = std::max(mx, dlt); ref = std::max(ref, (double)fabsf(r[idx]));
                        }
                    }
            if (nbad && getenv("Q64_DIAG")) {
                printf("        %zu bad elements of %zu; by d-tile: %zu %zu %zu %zu; by 32-row tile:", nbad, on, badd[0], badd[1], badd[2], badd[3]);
                for (size_t t = 0; t < badrow.size() && t < 48; ++t) printf(" %zu", badrow[t]);
                printf("\n");
            }
            printf("      %s vs shipped: max|d| = %.3e (|O|max %.3f)%s\n", label, mx, ref, bad ? "  NON-FINITE VALUES" : (mx > 2e-2 * std::max(ref, 1e-3) ? "  MISMATCH" : ""));
#ifdef Q64_STAMP
            if (c.bench) {
                std::vector<long long> hd((size_t)nunits * 4 * 8); hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
                for (int cls = 0; cls < 3; ++cls) {                       // waves on the whole / half / passive stream
                    double ph[5] = {0, 0, 0, 0, 0}; double tiles = 0, nw = 0; long long t0 = 1LL << 62, t1 = 0;
                    for (int u = 0; u < nunits * 4; ++u) {
                        const long long* d = &hd[(size_t)u * 8];
                        if (d[5] == 0 || (int)(d[6] >> 16) != cls) continue;
                        for (int k2 = 0; k2 < 5; ++k2) ph[k2] += (double)(d[k2 + 1] - d[k2]);
                        tiles += d[6] & 0xffff; nw += 1; t0 = std::min(t0, d[0]); t1 = std::max(t1, d[5]);
                    }
                    if (nw == 0) continue;
                    printf("      stamps %s (shader cycles, mean per wave-unit, %.0f wave-units): core %.0f (%.1f per tile, %.1f tiles) | seam + output %.0f ; first start to last end %lld\n",
                           cls == 2 ? "pass." : cls ? "HALF " : "whole", nw, ph[2] / nw, ph[2] / std::max(1.0, tiles), tiles / nw, ph[4] / nw, t1 - t0);
                }
            }
            hipFree(dbg);
#endif
        }
        // ---- tail split: whole units (slot 0) for the query groups below tail_g, a tail_ks-way key split (slots 1.., ml) for the rest
        {
            const int ng = (nt32 + 7) / 8;
            int pks = 1, ptg = ng, ptk = 1;
            dex::q64tool::attention_q64_plan(c.N, c.B, KSMAX, &pks, &ptg, &ptk);
            struct TP { int tg, tk, planned; };
            std::vector<TP> tps;
            if (pks == 1 && ptk > 1) tps.push_back({ptg, ptk, 1});
            if (ng >= 2 && nT >= 3) { tps.push_back({ng - 1, 2, 0}); tps.push_back({std::max(1, ng / 2), 3, 0}); }
            for (const TP& tp : tps) {
                AttnDirectP a2 = a; a2.O = O2; a2.ksplit = 1; a2.ml = ml; a2.o_sstride = (long)on; a2.tail_g = tp.tg; a2.tail_ks = tp.tk; a2.half_n = -1;
                hipMemset(O2, 0, on * 4 * KSMAX);
                char label[96]; snprintf(label, sizeof label, "q64 tail split g>=%d x%d%s", tp.tg, tp.tk, tp.planned ? " (plan)" : "");
                if (c.bench) timeit(label, 20, fl, [&] { dex::q64tool::launch_attention_q64(a2, 0); });
                else dex::q64tool::launch_attention_q64(a2, 0);
                if (hipDeviceSynchronize() != hipSuccess) { printf("      %s: LAUNCH FAILED\n", label); return 1; }
                std::vector<float> g(on * (tp.tk + 1)), hml((size_t)tp.tk * c.B * 2 * c.N * 2);
                hipMemcpy(g.data(), O2, g.size() * 4, hipMemcpyDeviceToHost);
                hipMemcpy(hml.data(), ml, hml.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, ref = 0; size_t bad = 0;
                for (int b = 0; b < c.B; ++b)
                    for (int n = 0; n < c.N; ++n)
                        for (int h = 0; h < 2; ++h) {
                            const bool tail = n >= tp.tg * 256;
                            double M = -1e300, W = 0;
                            if (tail) {
                                for (int s = 0; s < tp.tk; ++s) M = std::max(M, (double)hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]);
                                for (int s = 0; s < tp.tk; ++s) { const float* e = &hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]; W += e[1] * exp2((double)e[0] - M); }
                            }
                            for (int d = 0; d < 128; ++d) {
                                const size_t idx = ((size_t)b * c.N + n) * 256 + h * 128 + d;
                                double val = 0;
                                if (!tail) val = g[idx];
                                else {
                                    for (int s = 0; s < tp.tk; ++s) { const float* e = &hml[((((size_t)s * c.B + b) * 2 + h) * c.N + n) * 2]; val += g[(size_t)(1 + s) * on + idx] * e[1] * exp2((double)e[0] - M); }
                                    val /= W;
                                }
                                const double dlt = fabs(val - r[idx]);
                                if (!(dlt <= 1e30)) ++bad;
                                mx = std::max(mx, dlt); ref = std::max(ref, (double)fabsf(r[idx]));
                            }
                        }
                printf("      %s vs shipped: max|d| = %.3e (|O|max %.3f)%s\n", label, mx, ref, bad ? "  NON-FINITE VALUES" : (mx > 2e-2 * std::max(ref, 1e-3) ? "  MISMATCH" : ""));
            }
        }
        hipFree(q); hipFree(k); hipFree(v); hipFree(O); hipFree(O2); hipFree(ml);
    }
    return 0;
}
