// coresident.hip - does a wave with a large register allocation corrupt small waves of OTHER kernels that share its SIMD?
// (round 5: frames in flight were not bit-identical; the aggressors turned out to be the 448-register field kernels)
// spin<V, A, LDS>: persistent workgroups of 256 threads that only allocate (V arch VGPRs, A AGPRs, LDS bytes) and spin on VALU work that
// WRITES every register it owns (mode 1) or none of them beyond v0..v15 (mode 0).
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int V, int A, int LDSB>
__global__ void __launch_bounds__(256, 1) spin(int iters, int mode, float* out) {
    __shared__ char lds[LDSB > 0 ? LDSB : 16];
    if (V >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    else if (V >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (A >= 192) asm volatile("v_accvgpr_write_b32 a191, 0" ::: "a191");
    else if (A >= 64) asm volatile("v_accvgpr_write_b32 a63, 0" ::: "a63");
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    if (LDSB > 0) lds[threadIdx.x] = 1;
    for (int i = 0; i < iters; ++i) {
        x = fmaf(x, 1.0001f, y); y = fmaf(y, 0.9999f, x);
        if (mode == 1) {
            if (V >= 256) asm volatile("v_mov_b32 v255, %0\n v_mov_b32 v224, %0\n v_mov_b32 v192, %0\n v_mov_b32 v160, %0" :: "v"(x) : "v255", "v224", "v192", "v160");
            if (A >= 192) asm volatile("v_accvgpr_write_b32 a191, %0\n v_accvgpr_write_b32 a160, %0\n v_accvgpr_write_b32 a128, %0\n v_accvgpr_write_b32 a150, %0\n v_accvgpr_write_b32 a176, %0\n v_accvgpr_write_b32 a184, %0"
                                       :: "v"(x) : "a191", "a160", "a128", "a150", "a176", "a184");
        }
    }
    if (x == 12345.f) out[0] = x + y + lds[0];
}
// spin2<KIND>: 448 registers allocated; the loop issues loads whose DESTINATION is (1) high AGPRs, (2) high VGPRs, (3) AGPRs from LDS
// (the field kernels' compiler-chosen forms: global_load_dwordx3 a[170:172] for the sample points, ds_read_b128 a[0:3] for operands)
template <int KIND>
__global__ void __launch_bounds__(256, 1) spin2(int iters, const float* __restrict__ src, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a191, 0" ::: "a191");
    const float* p = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
    lds[threadIdx.x] = 1.0f; lds[threadIdx.x + 256] = 2.0f;
    __syncthreads();
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + threadIdx.x * 16;
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 1) asm volatile("global_load_dwordx3 a[170:172], %0, off\n global_load_dwordx3 a[174:176], %0, off offset:16\n global_load_dwordx3 a[144:146], %0, off offset:32\n s_waitcnt vmcnt(0)"
                                    :: "v"(p) : "a170", "a171", "a172", "a174", "a175", "a176", "a144", "a145", "a146", "memory");
        if (KIND == 2) asm volatile("global_load_dwordx3 v[200:202], %0, off\n global_load_dwordx3 v[204:206], %0, off offset:16\n global_load_dwordx3 v[240:242], %0, off offset:32\n s_waitcnt vmcnt(0)"
                                    :: "v"(p) : "v200", "v201", "v202", "v204", "v205", "v206", "v240", "v241", "v242", "memory");
        if (KIND == 3) asm volatile("ds_read_b128 a[0:3], %0\n ds_read_b128 a[180:183], %0 offset:4096\n ds_read2st64_b32 a[54:55], %0 offset0:8 offset1:12\n s_waitcnt lgkmcnt(0)"
                                    :: "v"(la) : "a0", "a1", "a2", "a3", "a180", "a181", "a182", "a183", "a54", "a55", "memory");
        x = fmaf(x, 1.0001f, 1.0f);
    }
    if (x == 12345.f) out[0] = x + lds[3];
}
extern "C" int launch2(int kind, int groups, int iters, const float* src, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 1) hipLaunchKernelGGL(spin2<1>, dim3(groups), dim3(256), 0, st, iters, src, out);
    else if (kind == 2) hipLaunchKernelGGL(spin2<2>, dim3(groups), dim3(256), 0, st, iters, src, out);
    else hipLaunchKernelGGL(spin2<3>, dim3(groups), dim3(256), 0, st, iters, src, out);
    return hipGetLastError() != hipSuccess;
}
extern "C" int launch(int v, int a, int lds, int groups, int iters, int mode, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(VV, AA, LL) if (v == VV && a == AA && lds == LL) { hipLaunchKernelGGL((spin<VV, AA, LL>), dim3(groups), dim3(256), 0, st, iters, mode, out); return hipGetLastError() != hipSuccess; }
    L(256, 192, 0) L(256, 192, 131072) L(256, 0, 0) L(128, 0, 0) L(128, 64, 0) L(0, 0, 0) L(0, 0, 131072) L(256, 64, 0)
    return 2;
}
