// Does the clock the chip holds under back-to-back v_mfma_f32_32x32x16_f16 depend on the DATA?  Register-only loops (no LDS, no memory),
// one wave per SIMD, 1024 workgroups; operands per MFMA cycle through NSET different register sets whose contents are
//   KIND 0: all zero                1: one small constant          2: random fp16 in [-1, 1) (full-entropy mantissas)
//   KIND 3: random, but every MFMA uses the SAME operand set (values random, no toggling between consecutive MFMAs)
//   KIND 4: random with the low 5 mantissa bits of both operands cleared      5: random, half of the B values zero (post-ReLU activations)
//   KIND 6: A random in [-1, 1), B = small residuals (random x 2^-11: what the lo halves of a hi / lo split look like)
// Prints ms, shader clock (s_memtime / s_memrealtime) and the fraction of the nominal 2.4 GHz rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    half8 A[8], B[4];
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 8; ++q)
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (KIND == 1) v = 0.01f;
            if (KIND >= 2) v = (float)(int)(rng(s) >> 8) * (1.0f / 8388608.0f) - 1.0f;
            _Float16 hv = (_Float16)v;
            if (KIND == 4) { unsigned short b = __builtin_bit_cast(unsigned short, hv) & 0xffe0u; hv = __builtin_bit_cast(_Float16, b); }
            A[q][j] = hv;
        }
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (KIND == 1) v = 0.02f;
            if (KIND >= 2) v = (float)(int)(rng(s) >> 8) * (1.0f / 8388608.0f) - 1.0f;
            if (KIND == 5 && (rng(s) & 0x10000u)) v = 0.f;
            if (KIND == 6) v *= (1.0f / 2048.0f);
            _Float16 hv = (_Float16)v;
            if (KIND == 4) { unsigned short b = __builtin_bit_cast(unsigned short, hv) & 0xffe0u; hv = __builtin_bit_cast(_Float16, b); }
            B[q][j] = hv;
        }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int qa = KIND == 3 ? 0 : u, qb = KIND == 3 ? 0 : (u & 3);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[qa], B[qb], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(qa + 1) & 7], B[(qb + 1) & 3], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(qa + 2) & 7], B[(qb + 2) & 3], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(qa + 3) & 7], B[(qb + 3) & 3], a3, 0, 0, 0);
        }
        if (KIND >= 2 && (it & 63) == 63) {      // keep the accumulators finite: their magnitude is part of the data
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] *= 1e-3f; a1[r] *= 1e-3f; a2[r] *= 1e-3f; a3[r] *= 1e-3f; }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float t = 0; for (int r = 0; r < 16; ++r) t += a0[r] + a1[r] + a2[r] + a3[r];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int KIND> void run(const char* name, float* d, unsigned long long* clk, int iters, double* ms_out, double* ghz_out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, d, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    *ms_out = ms; *ghz_out = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0;
}
int main() {
    float* d; (void)hipMalloc(&d, 1024);
    unsigned long long* clk; (void)hipMalloc(&clk, 64);
    const char* name[7] = {"operands all zero", "one small constant", "random fp16, 8 x 4 operand sets in rotation", "random fp16, the same operands every MFMA",
                           "random, low 5 mantissa bits cleared", "random, half of B zero (post-ReLU)", "A random, B random x 2^-11 (lo halves)"};
    const int iters = 25000;      // 32 MFMAs per iteration, 4 workgroups per CU back to back: ~100 ms per run
    double ms[7][5], g[7][5];
    double dm, dg;
    run<0>("", d, clk, 4000, &dm, &dg);
    for (int r = 0; r < 5; ++r) {
        run<0>(name[0], d, clk, iters, &ms[0][r], &g[0][r]); run<1>(name[1], d, clk, iters, &ms[1][r], &g[1][r]);
        run<2>(name[2], d, clk, iters, &ms[2][r], &g[2][r]); run<3>(name[3], d, clk, iters, &ms[3][r], &g[3][r]);
        run<4>(name[4], d, clk, iters, &ms[4][r], &g[4][r]); run<5>(name[5], d, clk, iters, &ms[5][r], &g[5][r]);
        run<6>(name[6], d, clk, iters, &ms[6][r], &g[6][r]);
    }
    for (int v = 0; v < 7; ++v) {
        double m = 0, c = 0; for (int r = 1; r < 5; ++r) { m += ms[v][r]; c += g[v][r]; } m /= 4; c /= 4;
        // 1024 workgroups x 4 waves x iters x 32 MFMAs x 32768 flop
        const double tf = 1024.0 * 4 * iters * 32 * 32768.0 / (m * 1e-3) / 1e12;
        printf("%-46s %8.2f ms  shader clock %.2f GHz  %.0f TFLOP/s = %.2f of the 2500 nominal\n", name[v], m, c, tf, tf / 2500.0);
    }
    return 0;
}
