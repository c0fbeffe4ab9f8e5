// tools/gridsync_bench.hip — what a grid-wide dependency costs on gfx950 in its two forms (round-6 verdict item 7: would ONE persistent launch
// for the whole DiT at B = 1 beat the chain of kernel nodes of a captured graph?):
//   A  a hipGraph of K dependent kernel nodes, G workgroups each; every workgroup writes 1 KB and reads the 1 KB its neighbour
//      (workgroup + G/2: another XCD) wrote in the previous node;
//   B  ONE persistent kernel of G co-resident workgroups with K phases and the same exchange, separated by a grid barrier (agent-scope
//      release / acquire on one counter; the L2s of the eight XCDs are not coherent with each other, so the release writes the 1 KB back
//      and the acquire invalidates - exactly what a kernel boundary does for free);
//   C  the same with the barrier only among the workgroups of one XCD (workgroup L runs on XCD L % 8): the hand-off the B = 1 cluster
//      block already uses (exchange partner on the same XCD).
//   B2 the grid barrier made hierarchical: arrival counter per XCD, the last arriver of an XCD arrives at the grid counter, waits for the
//      other seven and releases its XCD through a word the others poll (8 pollers per address instead of G on one);
//   B3 / C3 no fences at all - the protocol of the B = 1 cluster block (dit_rowchain.hip): payload stores write through (agent-scope
//      relaxed stores: sc1) / stay in the XCD's L2 (C3: plain stores), every workgroup then stores ITS flag word, and one lane per peer
//      polls that peer's flag (B3: G peers, loads from memory; C3: the G / 8 peers on this XCD, loads that bypass only the L1);
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridsync_bench.hip -o /tmp/gridsync_bench && /tmp/gridsync_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void exchange(float* buf, int phase, int wg, int partner, int G, float& acc) {
    // write this phase's 1 KB, (dependency), read the partner's 1 KB of the previous phase
    float* mine = buf + ((long)(phase & 1) * G + wg) * 256;
    const float* theirs = buf + ((long)((phase + 1) & 1) * G + partner) * 256;
    acc = 0.25f * acc + 0.25f * theirs[threadIdx.x] + 1.f;
    mine[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void node_kernel(float* buf, float* out, int phase, int G) {
    const int wg = blockIdx.x, partner = (wg + G / 2) % G;
    float acc = out[wg * 256 + threadIdx.x];
    exchange(buf, phase, wg, partner, G, acc);
    out[wg * 256 + threadIdx.x] = acc;
}

// hierarchical grid barrier (B2): ctr[x * 64] arrivals of XCD x, ctr[8 * 64] grid arrivals, ctr[(9 + x) * 64] release word of XCD x
__global__ __launch_bounds__(256) void persistent2_kernel(float* buf, float* out, unsigned* ctr, int K, int G) {
    const int wg = blockIdx.x, x = wg & 7, members = G / 8, partner = (wg + G / 2) % G;
    unsigned *cx = ctr + x * 64, *cg = ctr + 8 * 64, *rx = ctr + (9 + x) * 64;
    float acc = out[wg * 256 + threadIdx.x];
    for (int k = 0; k < K; ++k) {
        exchange(buf, k, wg, partner, G, acc);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(cx, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(k + 1) * members - 1) {
                __hip_atomic_fetch_add(cg, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(cg, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(k + 1) * 8) __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(rx, (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(rx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(k + 1)) __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    out[wg * 256 + threadIdx.x] = acc;
}

// flag protocol (B3 grid-wide / C3 XCD-local): flags[phase parity][wg]
template <int LOCAL>
__global__ __launch_bounds__(256) void flags_kernel(float* buf, float* out, unsigned* flags, int K, int G) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int members = LOCAL ? G / 8 : G;
    const int partner = LOCAL ? (wg + 8 * (members / 2)) % G : (wg + G / 2) % G;
    float acc = out[wg * 256 + tid];
    for (int k = 0; k < K; ++k) {
        float* mine = buf + ((long)(k & 1) * G + wg) * 256;
        float* theirs = buf + ((long)((k + 1) & 1) * G + partner) * 256;
        float t;
        if (LOCAL) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(theirs + tid) : "memory");
        else t = __hip_atomic_load(theirs + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = 0.25f * acc + 0.25f * t + 1.f;
        if (LOCAL) mine[tid] = acc; else __hip_atomic_store(mine + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* fl = flags + (long)(k & 1) * 256;
        if (tid == 0) {
            if (LOCAL) __hip_atomic_store(fl + wg, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(fl + wg, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < members) {
            unsigned* f = fl + (LOCAL ? (wg & 7) + 8 * tid : tid);
            for (;;) {
                unsigned v;
                if (LOCAL) asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f), "v"(0u) : "memory");
                else v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= (unsigned)(k + 1)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    out[wg * 256 + tid] = acc;
}

template <int LOCAL>
__global__ __launch_bounds__(256) void persistent_kernel(float* buf, float* out, unsigned* ctr, int K, int G) {
    const int wg = blockIdx.x;
    // LOCAL: barrier + exchange among the workgroups of this XCD (L % 8); else grid-wide, partner on another XCD
    const int members = LOCAL ? G / 8 : G;
    const int partner = LOCAL ? (wg + 8 * (members / 2)) % G : (wg + G / 2) % G;
    unsigned* c = ctr + (LOCAL ? (wg & 7) * 64 : 0);
    float acc = out[wg * 256 + threadIdx.x];
    for (int k = 0; k < K; ++k) {
        exchange(buf, k, wg, partner, G, acc);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(k + 1) * members;
            while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    out[wg * 256 + threadIdx.x] = acc;
}

int main() {
    const int K = 200;
    float *buf, *out; unsigned* ctr;
    CK(hipMalloc(&buf, 2 * 256 * 256 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&ctr, 8192));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("microseconds per grid-wide dependency (K = %d in a row), by workgroups G\n", K);
    for (int G : {64, 128, 256}) {
        CK(hipMemset(buf, 0, 2 * 256 * 256 * 4)); CK(hipMemset(out, 0, 256 * 256 * 4));
        // A: graph of K kernel nodes
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(node_kernel, dim3(G), dim3(256), 0, st, buf, out, k, G);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float msA = 1e30f, msB = 1e30f, msC = 1e30f, msS = 1e30f, msB2 = 1e30f, msB3 = 1e30f, msC3 = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msA = ms < msA ? ms : msA;
        }
        // (stream launches of the same K kernels, no graph)
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(node_kernel, dim3(G), dim3(256), 0, st, buf, out, k, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msS = ms < msS ? ms : msS;
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_kernel<0>, dim3(G), dim3(256), 0, st, buf, out, ctr, K, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msB = ms < msB ? ms : msB;
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_kernel<1>, dim3(G), dim3(256), 0, st, buf, out, ctr, K, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msC = ms < msC ? ms : msC;
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 8192, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent2_kernel, dim3(G), dim3(256), 0, st, buf, out, ctr, K, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msB2 = ms < msB2 ? ms : msB2;
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 8192, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(flags_kernel<0>, dim3(G), dim3(256), 0, st, buf, out, ctr, K, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msB3 = ms < msB3 ? ms : msB3;
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 8192, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(flags_kernel<1>, dim3(G), dim3(256), 0, st, buf, out, ctr, K, G);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msC3 = ms < msC3 ? ms : msC3;
        }
        std::vector<float> h(256);
        CK(hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost));
        printf("G = %3d   A graph node boundary %6.2f   (stream launches %6.2f)   B grid barrier %6.2f   B2 hierarchical %6.2f   C XCD-local barrier %6.2f   B3 grid flags %6.2f   C3 XCD-local flags %6.2f      (check %.3f)\n",
               G, msA * 1e3 / K, msS * 1e3 / K, msB * 1e3 / K, msB2 * 1e3 / K, msC * 1e3 / K, msB3 * 1e3 / K, msC3 * 1e3 / K, h[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
