// victim_ops.hip - which VALU operations of a small co-resident wave come out different beside the split-fp16 field kernels?
// every thread computes a few op chains on inputs derived from its index only; out[op][i] is compared with a run on an idle GPU
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) victim(int n, int iters, float* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = 1.0f + (float)(i % 9973) * 1.0e-3f, y = 0.37f + (float)(i % 7919) * 2.3e-4f;
    float a_sqrt = 0.f, a_rcp = 0.f, a_div = 0.f, a_rsq = 0.f, a_fma = 0.f, a_exp = 0.f, a_sin = 0.f, a_mul = 0.f;
    float fx = x, fy = y;
    for (int k = 0; k < iters; ++k) {
        a_sqrt += sqrtf(fx);                       // correctly rounded sqrt: v_sqrt_f32 + refinement
        a_rcp += __builtin_amdgcn_rcpf(fx);        // v_rcp_f32
        a_div += fy / fx;                          // correctly rounded division: v_div_scale / v_rcp / v_fma / v_div_fmas / v_div_fixup
        a_rsq += __builtin_amdgcn_rsqf(fx);        // v_rsq_f32
        a_fma = fmaf(a_fma, 0.999f, fx * fy);      // plain VALU
        a_exp += __expf(-fy);                      // v_exp_f32
        a_sin += __sinf(fx);                       // v_sin_f32
        a_mul += fx * fy;
        fx += 1.0e-3f; fy += 7.0e-4f;
    }
    out[0 * (size_t)n + i] = a_sqrt; out[1 * (size_t)n + i] = a_rcp; out[2 * (size_t)n + i] = a_div; out[3 * (size_t)n + i] = a_rsq;
    out[4 * (size_t)n + i] = a_fma; out[5 * (size_t)n + i] = a_exp; out[6 * (size_t)n + i] = a_sin; out[7 * (size_t)n + i] = a_mul;
}
extern "C" int launch_victim(int n, int iters, float* out, void* stream) {
    hipLaunchKernelGGL(victim, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, iters, out);
    return hipGetLastError() != hipSuccess;
}
