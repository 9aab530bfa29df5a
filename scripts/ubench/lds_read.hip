// microbenchmark: LDS operand-read rate of a 256-thread workgroup (1 wave per SIMD), ds_read_b128, lane-linear
// addresses (the k_field16 pattern), optionally interleaved with f16 MFMAs (6 per 4 reads, like one weight block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: reads only, 1: reads + 6 MFMA per 4 reads, 2: MFMA only, 3: one read between MFMAs, 4: as 3 with 3 accumulators
__global__ void __launch_bounds__(256, 1) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char ring[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)ring)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0};
    half8 h0 = {0}, l0 = {0}, h1 = {0}, l1 = {0}, x = {0};
    for (int j = 0; j < 8; ++j) x[j] = (_Float16)(0.01f * (lane + j));
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            half8 n0, m0, n1, m1;
            if (MODE == 3 || MODE == 4) {
                const unsigned a = base + b * 4096;
                f32x16& c2 = (MODE == 4) ? a2 : a1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, x, a0, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1" : "=v"(n0) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, x, a1, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(m0) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0, x, c2, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(n1) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, x, a0, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(m1) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, x, a1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1, x, c2, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                h0 = n0; l0 = m0; h1 = n1; l1 = m1;
                continue;
            }
            if (MODE != 2) {
                const unsigned a = base + b * 4096;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                             : "=v"(n0), "=v"(m0), "=v"(n1), "=v"(m1) : "v"(a) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, x, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0, x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, x, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1, x, a1, 0, 0, 0);
            } else {
                asm volatile("" :: "v"(h0), "v"(l0), "v"(h1), "v"(l1));
            }
            if (MODE != 2) { h0 = n0; l0 = m0; h1 = n1; l1 = m1; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r];
    s += (float)h0[0] + (float)l1[3];
    if (s == 123.456f) out[0] = s;
}
template <int MODE> void run(const char* name, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double blocks = 256.0 * 8 * iters * 16;                 // weight blocks per ... (4 waves each read 4 KB per block)
    double bytes = blocks * 4 * 4096;
    double per_cu_per_block_us = ms * 1e3 / (8.0 * iters * 16);   // 8 workgroups per CU run back to back
    printf("%-22s %8.3f ms  LDS read %.1f GB/s/CU  time per block %.3f us (= %.0f cycles @2.1GHz)\n", name, ms,
           bytes / 256 / (ms * 1e-3) / 1e9, per_cu_per_block_us, per_cu_per_block_us * 2100);
}
int main() {
    float* d; hipMalloc(&d, 1024);
    run<0>("reads only", d, 400);
    run<2>("mfma only", d, 400);
    run<1>("reads + mfma", d, 400);
    run<3>("interleaved", d, 400);
    run<4>("interleaved 3 acc", d, 400);
    return 0;
}
