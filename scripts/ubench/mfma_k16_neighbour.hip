// mfma_k16_neighbour.hip - stand-alone reproducer attempt for DESIGN 4.5 (round 6): does a wave that issues v_mfma_f32_32x32x16_f16 make
// co-resident waves of ANOTHER kernel on its SIMD consume their vector-memory loads early?
//
//   aggressor<KIND, FEED> : 256-thread workgroups, one per CU, ~200 registers (so that small waves fit beside them), a loop of matrix
//                           instructions on register operands:  KIND 16 = v_mfma_f32_32x32x16_f16,  8 = two v_mfma_f32_32x32x8_f16,
//                           2 = eight v_mfma_f32_32x32x2_f32 (fp32), 0 = VALU only.
//                           FEED 0 = registers only, 1 = + operands re-read from LDS every iteration (ds_read_b128),
//                           2 = + an LDS-DMA stream (global_load_lds_dwordx4) refilling that LDS, with its own vmcnt waits / barriers
//   victim                : the shape of k_normal - every thread gathers a 64-byte record from an L2-resident table at a hashed index,
//                           combines its 16 floats and stores one float; 24 registers, many waves per SIMD.
// The victim runs alone (reference), then `reps` times beside the aggressor on another stream; printed: elements that differ.
//   hipcc --offload-arch=gfx950 -O3 mfma_k16_neighbour.hip -o mfma_k16_neighbour && ./mfma_k16_neighbour [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int KIND, int FEED>
__global__ void __launch_bounds__(256, 1) aggressor(int iters, const char* __restrict__ stream_src, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 32768];
    // ~200 registers: 8 accumulator tiles (128) + operands; ask for a few more so that the allocation is the same for every KIND
    asm volatile("v_mov_b32 v199, 0" ::: "v199");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    half8 a, b;
    // RANDOM operand bits (the multiplier arrays' switching sets the power the matrix pipe draws: zeros / constants run cool)
    unsigned hsh = 0x9e3779b9u * (unsigned)(tid + 1) + 0x85ebca6bu * (unsigned)(blockIdx.x + 1);
    auto nexth = [&]() { hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5; return hsh; };
    auto rh = [&]() { return (_Float16)((float)(nexth() & 0xffffu) * (2.0f / 65536.0f) - 1.0f); };
    for (int j = 0; j < 8; ++j) { a[j] = rh(); b[j] = rh(); }
    for (int i = tid; i < 2 * 32768 / 2; i += 256) reinterpret_cast<_Float16*>(lds)[i] = rh();
    __syncthreads();
    const unsigned ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const char* g = stream_src + wave * 8192 + lane * 16;
    float x = tid * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (FEED == 2) {      // the weight ring of k_field16 in miniature: this wave moves 8 KB into the half of the ring nobody reads now
            const unsigned dst = ring_off + ((it + 1) & 1) * 32768 + wave * 8192;
            const char* src = g + (size_t)(it & 63) * 32768;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                         : : "v"(src), "s"(dst) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                         : : "v"(src + 4096), "s"(dst + 4096) : "memory", "m0");
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (FEED >= 1) {
                const char* s = lds + (it & 1) * 32768 + t * 1024 + lane * 16;
                a = *reinterpret_cast<const half8*>(s);
                b = *reinterpret_cast<const half8*>(s + 8192);
            }
            if (KIND == 16) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[t], 0, 0, 0);
            } else if (KIND == 8) {
                const half4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]}, b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(b0, a0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(b1, a1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, a1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, a0, acc[t], 0, 0, 0);
            } else if (KIND == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[k], (float)b[k], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(acc[t][r], 0.999f, (float)a[r & 7]);
            }
            // a little VALU work between the matrix instructions, like the field kernels' epilogue slices
            x = fmaf(x, 1.0001f, acc[t][it & 15] * 1e-30f);
        }
        if (FEED == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    float s = x;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

struct Rec { float v[16]; };
__global__ void __launch_bounds__(256) victim(const Rec* __restrict__ table, int n_table, const int* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = idx[i];
    const Rec r = table[k];                       // four 16-byte loads in flight
    const Rec q = table[(k * 7 + 13) % n_table];   // ... and four more, from another line
    float u = 0.0f, w = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { u = fmaf(r.v[j], q.v[15 - j], u); w += r.v[j] - q.v[j]; }
    out[i] = u * 0.5f + w;
}

template <int KIND, int FEED>
static void launch_agg(int iters, const char* src, float* out, hipStream_t st) {
    hipLaunchKernelGGL((aggressor<KIND, FEED>), dim3(256), dim3(256), 0, st, iters, src, out);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const int n_table = 13776, n = 1 << 22, rounds = 24;
    std::vector<Rec> h_table(n_table);
    std::vector<int> h_idx(n);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (auto& r : h_table) for (float& v : r.v) v = (float)(rnd() >> 8) * (1.0f / 16777216.0f) - 0.5f;
    for (int& k : h_idx) k = (int)(rnd() % n_table);
    Rec* d_table; int* d_idx; float *d_out, *d_ref, *d_dummy; char* d_src;
    CK(hipMalloc(&d_table, sizeof(Rec) * n_table)); CK(hipMalloc(&d_idx, 4 * (size_t)n)); CK(hipMalloc(&d_out, 4 * (size_t)n * rounds));
    CK(hipMalloc(&d_ref, 4 * (size_t)n)); CK(hipMalloc(&d_dummy, 64)); CK(hipMalloc(&d_src, 64 * 32768 + 65536));
    CK(hipMemcpy(d_table, h_table.data(), sizeof(Rec) * n_table, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, h_idx.data(), 4 * (size_t)n, hipMemcpyHostToDevice));
    {   // random fp16 values in [-1, 1) for the LDS-DMA stream
        std::vector<_Float16> hs((64 * 32768 + 65536) / 2);
        for (auto& v : hs) v = (_Float16)((float)(rnd() >> 16) * (2.0f / 65536.0f) - 1.0f);
        CK(hipMemcpy(d_src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    }
    hipStream_t A, B;
    CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B));
    hipLaunchKernelGGL(victim, dim3(n / 256), dim3(256), 0, B, d_table, n_table, d_idx, d_ref, n);
    CK(hipDeviceSynchronize());
    std::vector<float> ref(n), got(n);
    CK(hipMemcpy(ref.data(), d_ref, 4 * (size_t)n, hipMemcpyDeviceToHost));
    struct Case { const char* name; void (*fn)(int, const char*, float*, hipStream_t); int iters; };
    const Case cases[] = {
        {"no aggressor", nullptr, 0},
        {"VALU only, registers", launch_agg<0, 0>, 40000},
        {"fp32 MFMA 32x32x2, registers", launch_agg<2, 0>, 6000},
        {"f16 MFMA K = 8 (x2), registers", launch_agg<8, 0>, 12000},
        {"f16 MFMA K = 16, registers", launch_agg<16, 0>, 24000},
        {"f16 MFMA K = 8 (x2), operands from LDS", launch_agg<8, 1>, 12000},
        {"f16 MFMA K = 16, operands from LDS", launch_agg<16, 1>, 24000},
        {"f16 MFMA K = 8 (x2), LDS-DMA ring", launch_agg<8, 2>, 12000},
        {"f16 MFMA K = 16, LDS-DMA ring", launch_agg<16, 2>, 24000},
    };
    for (const Case& c : cases) {
        printf("%-44s differing victim elements per repetition (%d victim launches of %d threads each):", c.name, rounds, n);
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipMemsetAsync(d_out, 0, 4 * (size_t)n * rounds, B));
            CK(hipDeviceSynchronize());
            hipEvent_t a0, a1, b0, b1;
            CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
            CK(hipEventRecord(a0, A));
            if (c.fn) c.fn(c.iters, d_src, d_dummy, A);
            CK(hipEventRecord(a1, A));
            CK(hipEventRecord(b0, B));
            for (int r = 0; r < rounds; ++r)
                hipLaunchKernelGGL(victim, dim3(n / 256), dim3(256), 0, B, d_table, n_table, d_idx, d_out + (size_t)r * n, n);
            CK(hipEventRecord(b1, B));
            CK(hipDeviceSynchronize());
            long long bad = 0;
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemcpy(got.data(), d_out + (size_t)r * n, 4 * (size_t)n, hipMemcpyDeviceToHost));
                for (int i = 0; i < n; ++i) bad += (got[i] != ref[i]);
            }
            float ta, tb, tall;
            CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&tall, a0, b1));
            printf(" %lld (agg %.1f ms, victims %.1f ms, both %.1f)", bad, ta, tb, tall);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
