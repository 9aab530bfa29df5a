// dsn_image.hip - the step AFTER the hot path, on the device:
//   post_process (utils/render_utils.py:466-472): rows of the compacted per-ray outputs go, in order, to the pixels
//   where mask_at_box is set; every other pixel is zero; optional clamp of the colour image (test.py:62-63);
//   mse / psnr with and without the mask (metrics.py:8-21, test.py:70-71) accumulated in float64 like the reference
//   (its ground-truth image is float64).
// Integer / byte work is exact: the rank of a masked pixel is its exclusive prefix count over the mask.
#include "dsn_common.h"
#include "dsn_kernels.h"

#define IMG_THREADS 256

// per-block popcount of the mask
__global__ void __launch_bounds__(IMG_THREADS) k_img_count(const uint8_t* __restrict__ mask, int n, int32_t* __restrict__ counts) {
    const int i = blockIdx.x * IMG_THREADS + threadIdx.x;
    const bool m = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(m);
    __shared__ int s[IMG_THREADS / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of the block counts (single workgroup, chunks of 1024 with a carry)
__global__ void __launch_bounds__(1024) k_img_scan(int32_t* __restrict__ counts, int nblocks) {
    __shared__ int s[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? counts[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) counts[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(IMG_THREADS) k_img_scatter(const float* __restrict__ rgb, const float* __restrict__ disp,
                                                              const float* __restrict__ acc, const float* __restrict__ depth,
                                                              int R, const uint8_t* __restrict__ mask, int n,
                                                              const int32_t* __restrict__ offs, int clamp_rgb,
                                                              float* __restrict__ img_rgb, float* __restrict__ img_disp,
                                                              float* __restrict__ img_acc, float* __restrict__ img_depth) {
    const int i = blockIdx.x * IMG_THREADS + threadIdx.x;
    const bool m = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int s[IMG_THREADS / 64];
    if (lane == 0) s[wave] = __popcll(b);
    __syncthreads();
    int before = offs[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += s[w];
    const int rank = before + __popcll(b & ((1ull << lane) - 1ull));
    if (i >= n) return;
    const bool take = m && rank < R;
    float c[3] = {0.f, 0.f, 0.f};
    if (take) {
        for (int k = 0; k < 3; ++k) {
            float v = rgb[3 * rank + k];
            if (clamp_rgb) v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);   // torch.clamp keeps NaN; so does this
            c[k] = v;
        }
    }
    for (int k = 0; k < 3; ++k) img_rgb[3 * i + k] = c[k];
    if (img_disp) img_disp[i] = take ? disp[rank] : 0.0f;
    if (img_acc) img_acc[i] = take ? acc[rank] : 0.0f;
    if (img_depth) img_depth[i] = take ? depth[rank] : 0.0f;
}

// sums of squared differences: out[0] over all pixels, out[1] over masked pixels, out[2] number of masked pixels
__global__ void __launch_bounds__(IMG_THREADS) k_img_sqerr(const float* __restrict__ img, const double* __restrict__ gt64,
                                                            const float* __restrict__ gt32, const uint8_t* __restrict__ mask,
                                                            int n, double* __restrict__ sums) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = blockIdx.x * IMG_THREADS + threadIdx.x; i < n; i += gridDim.x * IMG_THREADS) {
        double e = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double g = gt64 ? gt64[3 * i + k] : (double)gt32[3 * i + k];
            const double d = (double)img[3 * i + k] - g;
            e += d * d;
        }
        a += e;
        if (mask && mask[i]) { b += e; c += 1.0; }
    }
    __shared__ double s[3][IMG_THREADS];
    s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
    __syncthreads();
    for (int off = IMG_THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off)
            for (int k = 0; k < 3; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) atomicAdd(sums + k, s[k][0]);
}

// out4 = {mse_all, mse_masked, psnr_all, psnr_masked}   (metrics.py:8-21)
__global__ void k_img_psnr(const double* __restrict__ sums, int n, double* __restrict__ out4) {
    const double mse_all = sums[0] / (3.0 * (double)n);
    const double mse_m = sums[1] / (3.0 * sums[2]);
    out4[0] = mse_all;
    out4[1] = mse_m;
    out4[2] = -10.0 * log10(mse_all);
    out4[3] = -10.0 * log10(mse_m);
}

size_t dsn_image_workspace_size(int H, int W) {
    const size_t nblocks = ((size_t)H * W + IMG_THREADS - 1) / IMG_THREADS;
    return dsn_align256(sizeof(int32_t) * nblocks) + 256;
}

void dsn_launch_image_scatter(const float* rgb, const float* disp, const float* acc, const float* depth, int R,
                              const uint8_t* mask, int H, int W, int clamp_rgb, float* img_rgb, float* img_disp,
                              float* img_acc, float* img_depth, void* workspace, hipStream_t st) {
    const int n = H * W;
    const int nblocks = (n + IMG_THREADS - 1) / IMG_THREADS;
    int32_t* counts = (int32_t*)workspace;
    hipLaunchKernelGGL(k_img_count, dim3(nblocks), dim3(IMG_THREADS), 0, st, mask, n, counts);
    hipLaunchKernelGGL(k_img_scan, dim3(1), dim3(1024), 0, st, counts, nblocks);
    hipLaunchKernelGGL(k_img_scatter, dim3(nblocks), dim3(IMG_THREADS), 0, st, rgb, disp, acc, depth, R, mask, n, counts,
                       clamp_rgb, img_rgb, img_disp, img_acc, img_depth);
}

void dsn_launch_image_psnr(const float* img_rgb, const double* gt64, const float* gt32, const uint8_t* mask, int H, int W,
                           double* out4, void* workspace, hipStream_t st) {
    const int n = H * W;
    double* sums = (double*)((char*)workspace + dsn_image_workspace_size(H, W) - 256);
    (void)hipMemsetAsync(sums, 0, 3 * sizeof(double), st);
    int blocks = (n + IMG_THREADS - 1) / IMG_THREADS;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_img_sqerr, dim3(blocks), dim3(IMG_THREADS), 0, st, img_rgb, gt64, gt32, mask, n, sums);
    hipLaunchKernelGGL(k_img_psnr, dim3(1), dim3(1), 0, st, sums, n, out4);
}
