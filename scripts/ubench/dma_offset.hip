// does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds;
    const unsigned long long g = (unsigned long long)src + threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\ts_waitcnt vmcnt(0)"
                 : : "v"(g), "s"(base) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *s, *o, h[4096];
    hipMalloc(&s, 16384); hipMalloc(&o, 16384);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(s, h, 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o);
    hipMemcpy(h, o, 16384, hipMemcpyDeviceToHost);
    // piece 0: src words 0..255 -> lds words 0..255.  piece 1 (offset 2048 B = 512 words): src words 512..767 -> lds words 512..767 (if the
    // offset applies to the LDS side too) or lds words 0..255 again (if not)
    printf("lds[0]=%u lds[255]=%u lds[256]=%x lds[512]=%x lds[767]=%x lds[768]=%x\n", h[0], h[255], h[256], h[512], h[767], h[768]);
    return 0;
}
