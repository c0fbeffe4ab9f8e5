// tools/dmaprobe.hip — does the immediate offset of buffer_load ... lds move the LDS destination too?  (one wave, two pieces)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const unsigned* g, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) unsigned s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 1 << 20, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)&s[0], 16, threadIdx.x * 16, 0, 0, 0);          // piece 0 -> s[0..255]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)&s[0], 16, threadIdx.x * 16, 0, 2048, 0);       // imm 2048: global +2048; LDS +2048 too?
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = s[i];
}
int main() {
    unsigned *g, *o; hipMalloc(&g, 1 << 20); hipMalloc(&o, 4096 * 4);
    unsigned h[262144]; for (int i = 0; i < 262144; ++i) h[i] = i;
    hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    unsigned r[4096]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("s[0]=%u s[255]=%u | s[256]=%x | s[512]=%x (LDS+2048 B: global word 512 expected if the offset applies to LDS) s[767]=%x | s[1024]=%x\n", r[0], r[255], r[256], r[512], r[767], r[1024]);
    return 0;
}
