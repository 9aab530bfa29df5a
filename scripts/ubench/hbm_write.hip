// What WRITE bandwidth does a kernel get from HBM on this box, by access pattern?  (k_field16<train>, k_tangent16 and k_adjoint16 write
// 5.4 / 2.2 / 2.2 GB at 2 - 2.6 TB/s and the training step is bound by exactly those bytes: is that the memory system's write limit?)
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_write hbm_write.hip        run: ./hbm_write [GiB = 4]
// Patterns, all over the same buffer, 20 launches each after 3 warm-up launches:
//   0 grid-stride float4 stores (a wave-instruction = 1 KB contiguous), blocks = CUs x 8
//   1 one contiguous slab per workgroup (256 workgroups)
//   2 the field kernels' pattern: lane l of a wave owns ROW l (1 KB rows), every instruction stores 16 bytes of each of 64 rows; a wave
//     walks its 64 rows left to right (64 instructions per 64 KB), 4 waves per workgroup, one tile per workgroup
//   3 pattern 2 with 32 rows per wave and the two half-waves 16 bytes apart (what the MFMA accumulator layout gives: lane = (half, row))
//   4 pattern 2 persistent: 256 workgroups walk the tiles
//   5 read + write (copy), grid-stride: what a mixed stream gets
//   6 pattern 3 into SEVEN arrays round-robin (layer stride = the whole array), as k_tangent16 writes its layers
//   7 pattern 6 from ONE workgroup of four waves per CU (100 KB of LDS keeps a second one out), persistent over the tiles: the field
//     kernels' occupancy - one wave per SIMD has to keep the whole write stream of its CU in flight
//   8 pattern 7 with two workgroups per CU (48 KB of LDS each)
//   9 pattern 7 with ~1 us of dependent FMAs between the layers (stores spread out as between MFMA blocks)
//  10 pattern 7 with FOUR workgroups per CU (24 KB each)
//  11 pattern 7 + a "ring wait" after every 8 stores: one 4-byte load from an L2-resident line, consumed at once.  vmcnt counts loads
//     AND stores in issue order, so waiting for that load waits for every store issued before it (what the field kernels' waits for
//     their LDS-DMA weight ring do to the activation stores issued in front of them)
//  12 the same load issued BEFORE the 8 stores and consumed behind them (the wait leaves the 8 younger stores in flight)
//  13-16 a MODEL of k_field16<train>'s memory behaviour: one four-wave workgroup per CU; per "chunk" (one 32-feature output block of one
//     layer): s_waitcnt vmcnt(0) + barrier, LDS-DMA of the next 32 KB weight chunk from a 7 MB L2-resident image (8 x 1 KB per wave),
//     ~1500 cycles of dependent FMAs (the chunk's 48 MFMAs), then the block's 4 stores (32 rows x 128 B per wave).
//     13 = all three; 14 = no stores; 15 = no DMA; 16 = no FMAs; 17 = all three with the wait leaving the 4 younger stores in flight
//     (vmcnt(4)); 18 = FMAs + barriers alone; 19 / 20 = 16 / 13 with every instruction writing WHOLE 128-byte lines (8 lanes per row, 8 rows per
//     instruction); 21 / 22 = the same with 64-byte segments (lane quads).  Do stores + weight stream + compute add up or overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_stride(float4* __restrict__ p, size_t n) {
    const size_t step = (size_t)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) p[i] = v;
}
__global__ void __launch_bounds__(256) k_slab(float4* __restrict__ p, size_t n) {
    const size_t per = n / gridDim.x;
    float4* q = p + per * blockIdx.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    for (size_t i = threadIdx.x; i < per; i += 256) q[i] = v;
}
// rows of 64 float4 (1 KB); tile = 256 rows per workgroup (64 per wave)
__global__ void __launch_bounds__(256) k_rows64(float4* __restrict__ p, size_t rows, int persistent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    for (size_t t = blockIdx.x; t * 256 < rows; t += persistent ? gridDim.x : (size_t)1 << 60) {
        const size_t row = t * 256 + wave * 64 + lane;
        if (row < rows) {
            float4* q = p + row * 64;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) q[c] = v;
        }
        if (!persistent) break;
    }
}
// 32 rows per wave, lane = (half, row): the half-waves write neighbouring 16-byte pieces; tile = 128 rows per workgroup
__global__ void __launch_bounds__(256) k_rows32(float4* __restrict__ p, size_t rows, size_t layer_stride, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    const size_t row = (size_t)blockIdx.x * 128 + wave * 32 + (lane & 31);
    if (row >= rows) return;
    for (int L = 0; L < layers; ++L) {
        float4* q = p + L * layer_stride + row * 64 + half;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) q[2 * c] = v;
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) k_rows32_ringwait(float4* __restrict__ p, size_t rows, size_t layer_stride, int layers, const float* ringsrc,
                                                         float* sink) {
    __shared__ float pad[100 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    if (threadIdx.x == 0) pad[0] = 1.0f;
    __syncthreads();
    float x = pad[0] + (float)lane;
    const float* rp = ringsrc + (threadIdx.x & 255);
    for (size_t t = blockIdx.x; t * 128 < rows; t += gridDim.x) {
        const size_t row = t * 128 + wave * 32 + (lane & 31);
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = {x, 2.f, 3.f, (float)t};
        for (int L = 0; L < layers; ++L) {
            float4* q = p + L * layer_stride + (row < rows ? row : 0) * 64 + half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float r;
                if (MODE == 12) asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(rp) : "memory");
#pragma unroll
                for (int c = 0; c < 8; ++c) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(q + 2 * (8 * g + c)), "v"(v) : "memory");
                if (MODE == 11) {
                    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(rp) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                x += r * 1e-30f;
            }
        }
    }
    if (x == 123.456f) sink[0] = x;
}
template <bool STORES, bool DMA, int SPIN, bool LAG = false, int SEG = 32, bool DEEP = false, int SMOD = 0, int LMOD = 0, bool SPREAD = false>
__global__ void __launch_bounds__(256) k_train_model(float4* __restrict__ p, size_t rows, size_t layer_stride, const char* __restrict__ wimg,
                                                     float* sink) {
    __shared__ __attribute__((aligned(16))) char ring[3 * 32768];
    __shared__ float pad[4 * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    if (threadIdx.x == 0) pad[0] = 1.0f;
    __syncthreads();
    float x = pad[0] + (float)lane;
    const unsigned ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    typedef float f4 __attribute__((ext_vector_type(4)));
    int chunk = 0;
    for (size_t t = blockIdx.x; t * 128 < rows; t += gridDim.x) {
        const size_t row = t * 128 + wave * 32 + (lane & 31);
        const f4 v = {x, 2.f, 3.f, (float)t};
        for (int L = 0; L < 7; ++L) {
            float4* q = p + L * layer_stride + (row < rows ? row : 0) * 64 + half;
#pragma unroll 1
            for (int m = 0; m < 8; ++m, ++chunk) {
                // DEEP: the DMA runs TWO chunks ahead (three ring slots).  In issue order: ... DMA(c+1) stores DMA(c+2) stores | boundary c+1
                // needs DMA(c+1): 4 + 8 + 4 younger operations may stay in flight - a store then has 3 chunk periods to be acknowledged
                if (DEEP && chunk > 1) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
                else if (LAG && chunk > 0) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (DMA) {
                    const char* src = wimg + (size_t)(chunk % 218) * 32768 + wave * 8192 + 4096 + lane * 16;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_off + (DEEP ? chunk % 3 : (chunk & 1)) * 32768 + wave * 8192 + 4096);
if (LMOD == 0) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072\n\tglobal_load_lds_dwordx4 %0, off offset:-4096\n\tglobal_load_lds_dwordx4 %0, off offset:-3072\n\tglobal_load_lds_dwordx4 %0, off offset:-2048\n\tglobal_load_lds_dwordx4 %0, off offset:-1024" : : "v"(src), "s"(dst) : "memory", "m0");
                    if (LMOD == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt\n\tglobal_load_lds_dwordx4 %0, off offset:1024 nt\n\tglobal_load_lds_dwordx4 %0, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, off offset:3072 nt\n\tglobal_load_lds_dwordx4 %0, off offset:-4096 nt\n\tglobal_load_lds_dwordx4 %0, off offset:-3072 nt\n\tglobal_load_lds_dwordx4 %0, off offset:-2048 nt\n\tglobal_load_lds_dwordx4 %0, off offset:-1024 nt" : : "v"(src), "s"(dst) : "memory", "m0");
                    if (LMOD == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1\n\tglobal_load_lds_dwordx4 %0, off offset:1024 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:2048 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:3072 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-4096 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-3072 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-2048 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-1024 sc1" : : "v"(src), "s"(dst) : "memory", "m0");
                    if (LMOD == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:1024 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:2048 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:3072 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-4096 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-3072 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-2048 sc0 sc1\n\tglobal_load_lds_dwordx4 %0, off offset:-1024 sc0 sc1" : : "v"(src), "s"(dst) : "memory", "m0");
                }
                if (SPREAD) {      // one store behind each quarter of the chunk's arithmetic (with LAG: the wait leaves these four in flight)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll 1
                        for (int i = 0; i < SPIN / 4; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
                        float4* a = p + L * layer_stride + (t * 128 + wave * 32 + (lane & 31)) * 64 + 8 * m + 2 * c + half;
                        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(a), "v"(v) : "memory");
                    }
                    continue;
                }
                if (STORES) {      // (the previous block's values: issued right behind the boundary, a whole chunk period before the next wait)
                    // the wave's block = 32 rows x 128 B (8 pieces of 16 B per row), written by 4 instructions of 64 pieces.  SEG = bytes of a
                    // row one instruction writes contiguously: 32 (the kernels: lane = (half, row), 2 pieces per row), 64 (lane quads),
                    // 128 (8 lanes = one row's whole line, 8 rows per instruction)
                    constexpr int PPR = SEG / 16;                 // pieces per row per instruction
                    constexpr int RPI = 64 / PPR;                 // rows per instruction
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = SEG == 32 ? (lane & 31) : (c * RPI + lane / PPR) % 32;
                        const int piece = SEG == 32 ? (2 * c + half) : (SEG == 128 ? lane % 8 : (lane % PPR) + PPR * (c / (4 * PPR / 8)) % 8);
                        float4* a = p + L * layer_stride + (t * 128 + wave * 32 + r) * 64 + 8 * m + piece;
                        if (SMOD == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(a), "v"(v) : "memory");
                        if (SMOD == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(a), "v"(v) : "memory");
                        if (SMOD == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(a), "v"(v) : "memory");
                        if (SMOD == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(a), "v"(v) : "memory");
                        if (SMOD == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(a), "v"(v) : "memory");
                    }
                }
                if (SPIN) {
#pragma unroll 1
                    for (int i = 0; i < SPIN; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
                }
            }
        }
    }
    if (x == 123.456f) sink[0] = x + ring[lane];
}
template <int LDS_KB, int SPIN>
__global__ void __launch_bounds__(256) k_rows32_persistent(float4* __restrict__ p, size_t rows, size_t layer_stride, int layers, float* sink) {
    __shared__ float pad[LDS_KB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    if (threadIdx.x == 0) pad[0] = 1.0f;       // (keeps the allocation)
    __syncthreads();
    float x = pad[0] + (float)lane;
    for (size_t t = blockIdx.x; t * 128 < rows; t += gridDim.x) {
        const size_t row = t * 128 + wave * 32 + (lane & 31);
        const float4 v = make_float4(x, 2.f, 3.f, (float)t);
        for (int L = 0; L < layers; ++L) {
            if (row < rows) {
                float4* q = p + L * layer_stride + row * 64 + half;
#pragma unroll 8
                for (int c = 0; c < 32; ++c) q[2 * c] = v;
            }
            if (SPIN) {
#pragma unroll 1
                for (int i = 0; i < SPIN; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
            }
        }
    }
    if (x == 123.456f) sink[0] = x;
}
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) b[i] = a[i];
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / (7 * 65536) * (7 * 65536);
    const size_t n = bytes / 16, rows = bytes / 1024;
    float4 *p, *q;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 0, bytes); hipMemset(q, 0, bytes);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* sink; hipMalloc(&sink, 64);
    float* sink2; hipMalloc(&sink2, 4096); hipMemset(sink2, 0, 4096);
    char* wimg; hipMalloc(&wimg, 218 * 32768); hipMemset(wimg, 0, 218 * 32768);
    for (int pat = 0; pat < 42; ++pat) {
        auto launch = [&]() {
            switch (pat) {
                case 0: hipLaunchKernelGGL(k_stride, dim3(cus * 8), dim3(256), 0, 0, p, n); break;
                case 1: hipLaunchKernelGGL(k_slab, dim3(256), dim3(256), 0, 0, p, n); break;
                case 2: hipLaunchKernelGGL(k_rows64, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, p, rows, 0); break;
                case 3: hipLaunchKernelGGL(k_rows32, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, 0, p, rows, (size_t)0, 1); break;
                case 4: hipLaunchKernelGGL(k_rows64, dim3(256), dim3(256), 0, 0, p, rows, 1); break;
                case 5: hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, (const float4*)q, p, n); break;
                case 6: hipLaunchKernelGGL(k_rows32, dim3((unsigned)((rows / 7 + 127) / 128)), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7); break;
                case 7: hipLaunchKernelGGL((k_rows32_persistent<100, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, sink); break;
                case 8: hipLaunchKernelGGL((k_rows32_persistent<48, 0>), dim3(cus * 2), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, sink); break;
                case 9: hipLaunchKernelGGL((k_rows32_persistent<100, 500>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, sink); break;
                case 11: hipLaunchKernelGGL((k_rows32_ringwait<11>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, (const float*)sink2, sink); break;
                case 12: hipLaunchKernelGGL((k_rows32_ringwait<12>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, (const float*)sink2, sink); break;
                case 13: hipLaunchKernelGGL((k_train_model<true, true, 48>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 14: hipLaunchKernelGGL((k_train_model<false, true, 48>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 15: hipLaunchKernelGGL((k_train_model<true, false, 48>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 16: hipLaunchKernelGGL((k_train_model<true, true, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 17: hipLaunchKernelGGL((k_train_model<true, true, 48, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 18: hipLaunchKernelGGL((k_train_model<false, false, 48>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 19: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 128>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 20: hipLaunchKernelGGL((k_train_model<true, true, 48, false, 128>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 21: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 64>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 22: hipLaunchKernelGGL((k_train_model<true, true, 48, false, 64>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 23: hipLaunchKernelGGL((k_train_model<true, true, 48, false, 32, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 24: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 25: hipLaunchKernelGGL((k_train_model<false, true, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 26: hipLaunchKernelGGL((k_train_model<true, false, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 27: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 1, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 28: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 2, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 29: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 3, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 30: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 4, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 31: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 0, 1>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 32: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 0, 2>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 33: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 0, 3>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 34: hipLaunchKernelGGL((k_train_model<true, true, 0, false, 32, false, 1, 1>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 35: hipLaunchKernelGGL((k_train_model<true, true, 48, false, 32, false, 1, 0>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 36: hipLaunchKernelGGL((k_train_model<true, true, 92>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 37: hipLaunchKernelGGL((k_train_model<false, true, 92>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 38: hipLaunchKernelGGL((k_train_model<false, false, 92>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 39: hipLaunchKernelGGL((k_train_model<true, true, 92, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 40: hipLaunchKernelGGL((k_train_model<true, true, 92, true, 32, false, 0, 0, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 41: hipLaunchKernelGGL((k_train_model<true, true, 92, false, 32, false, 0, 0, true>), dim3(cus), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, (const char*)wimg, sink); break;
                case 10: hipLaunchKernelGGL((k_rows32_persistent<24, 0>), dim3(cus * 4), dim3(256), 0, 0, p, rows / 7, (rows / 7) * 64, 7, sink); break;
            }
        };
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        const double moved = (pat == 5 ? 2.0 : 1.0) * (double)bytes;
        printf("pattern %d: %.3f ms per launch, %.2f TB/s %s\n", pat, ms / 20, moved / (ms / 20 * 1e-3) / 1e12, pat == 5 ? "(read + write)" : "written");
    }
    return 0;
}
