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
extern "C" int launch(int v, int a, int lds, int groups, int iters, int mode, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(VV, AA, LL) if (v == VV && a == AA && lds == LL) { hipLaunchKernelGGL((spin<VV, AA, LL>), dim3(groups), dim3(256), 0, st, iters, mode, out); return hipGetLastError() != hipSuccess; }
    L(256, 192, 0) L(256, 192, 131072) L(256, 0, 0) L(128, 0, 0) L(128, 64, 0) L(0, 0, 0) L(0, 0, 131072) L(256, 64, 0)
    return 2;
}
