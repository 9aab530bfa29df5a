// lds_dma_spin.hip - aggressor that does NOTHING but LDS-DMA (global_load_lds_dword / _dwordx4) in a loop, with a small register
// footprint (so that waves of other kernels share its SIMDs).  Round 5: waves of OTHER kernels that share a SIMD with the split-fp16
// field kernels consumed registers before their own global loads had landed; this isolates the instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int X4>
__global__ void __launch_bounds__(256) dma_spin(const char* __restrict__ src, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* g = src + ((size_t)blockIdx.x * 4 + wave) * 16384 + lane * (X4 ? 16 : 4);
    for (int i = 0; i < iters; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(base + wave * 16384);
        if (X4) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                         : : "v"(g), "s"(dst) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256\n\t"
                         "global_load_lds_dword %0, off offset:512\n\tglobal_load_lds_dword %0, off offset:768"
                         : : "v"(g), "s"(dst) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (iters < 0) out[threadIdx.x] = lds[threadIdx.x];
}
// the same loop with ordinary loads (control)
__global__ void __launch_bounds__(256) load_spin(const char* __restrict__ src, int iters, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4* g = reinterpret_cast<const float4*>(src + ((size_t)blockIdx.x * 4 + wave) * 16384 + lane * 16);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < iters; ++i) {
        const volatile float4* vg = g;
        float a = vg[0].x, b = vg[64].y, c = vg[128].z, d = vg[192].w;
        acc.x += a + b + c + d;
        asm volatile("" : "+v"(acc.x));
    }
    if (acc.x == 12345.f) out[threadIdx.x] = acc.x;
}
extern "C" int launch_dma(int kind, int groups, int iters, const void* src, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 4) hipLaunchKernelGGL(dma_spin<1>, dim3(groups), dim3(256), 0, st, (const char*)src, iters, out);
    else if (kind == 1) hipLaunchKernelGGL(dma_spin<0>, dim3(groups), dim3(256), 0, st, (const char*)src, iters, out);
    else hipLaunchKernelGGL(load_spin, dim3(groups), dim3(256), 0, st, (const char*)src, iters, out);
    return hipGetLastError() != hipSuccess;
}
