// dsn_nn.hip - device-side construction of the exact nearest-centroid lists (see dsn_nn.h).
// HBM/L2-bound integer+float work: one wavefront per grid cell sweeps the centroid table (coalesced
// float4 loads, table is L2-resident), wave-level min / ballot compaction, single-block scan.
#include "dsn_common.h"
#include "dsn_kernels.h"

__global__ void __launch_bounds__(256) k_grid_params(const float4* __restrict__ cent, int F, float pad,
                                                      int target_cells, int maxcell, int cap, DsnGrid* __restrict__ g) {
    __shared__ float s_lo[3][256], s_hi[3][256];
    const int t = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int f = t; f < F; f += 256) {
        const float4 c = cent[f];
        lo[0] = fminf(lo[0], c.x); lo[1] = fminf(lo[1], c.y); lo[2] = fminf(lo[2], c.z);
        hi[0] = fmaxf(hi[0], c.x); hi[1] = fmaxf(hi[1], c.y); hi[2] = fmaxf(hi[2], c.z);
    }
    for (int k = 0; k < 3; ++k) { s_lo[k][t] = lo[k]; s_hi[k][t] = hi[k]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s)
            for (int k = 0; k < 3; ++k) {
                s_lo[k][t] = fminf(s_lo[k][t], s_lo[k][t + s]);
                s_hi[k][t] = fmaxf(s_hi[k][t], s_hi[k][t + s]);
            }
        __syncthreads();
    }
    if (t == 0) {
        float e[3];
        for (int k = 0; k < 3; ++k) { g->lo[k] = s_lo[k][0] - pad; e[k] = (s_hi[k][0] + pad) - g->lo[k]; }
        float cell = cbrtf(e[0] * e[1] * e[2] / (float)target_cells);
        int nx, ny, nz;
        for (int it = 0; it < 64; ++it) {
            nx = (int)ceilf(e[0] / cell) + 1; ny = (int)ceilf(e[1] / cell) + 1; nz = (int)ceilf(e[2] / cell) + 1;
            if ((long long)nx * ny * nz <= (long long)maxcell) break;
            cell *= 1.08f;
        }
        g->cell = cell; g->inv_cell = 1.0f / cell;
        g->nx = nx; g->ny = ny; g->nz = nz; g->ncell = nx * ny * nz;
        g->ok = 0; g->total = 0; g->cap = cap; g->maxcell = maxcell;
    }
}

__device__ __forceinline__ void dsn_cell_box(const DsnGrid& g, int cell, float* blo, float* bhi) {
    const int iz = cell % g.nz, iy = (cell / g.nz) % g.ny, ix = cell / (g.nz * g.ny);
    blo[0] = g.lo[0] + ix * g.cell - DSN_GRID_GUARD; bhi[0] = g.lo[0] + (ix + 1) * g.cell + DSN_GRID_GUARD;
    blo[1] = g.lo[1] + iy * g.cell - DSN_GRID_GUARD; bhi[1] = g.lo[1] + (iy + 1) * g.cell + DSN_GRID_GUARD;
    blo[2] = g.lo[2] + iz * g.cell - DSN_GRID_GUARD; bhi[2] = g.lo[2] + (iz + 1) * g.cell + DSN_GRID_GUARD;
}
__device__ __forceinline__ float dsn_box_dmin2(const float4 c, const float* blo, const float* bhi) {
    const float dx = fmaxf(fmaxf(blo[0] - c.x, c.x - bhi[0]), 0.f);
    const float dy = fmaxf(fmaxf(blo[1] - c.y, c.y - bhi[1]), 0.f);
    const float dz = fmaxf(fmaxf(blo[2] - c.z, c.z - bhi[2]), 0.f);
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ float dsn_box_dmax2(const float4 c, const float* blo, const float* bhi) {
    const float dx = fmaxf(fabsf(c.x - blo[0]), fabsf(c.x - bhi[0]));
    const float dy = fmaxf(fabsf(c.y - blo[1]), fabsf(c.y - bhi[1]));
    const float dz = fmaxf(fabsf(c.z - blo[2]), fabsf(c.z - bhi[2]));
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ bool dsn_in_list(float dmin2, float u2) { return dmin2 <= u2 * (1.0f + 1e-5f) + 1e-12f; }

// pass 1+2: U(B)^2 and the list length of every cell (one wavefront per cell)
__global__ void __launch_bounds__(256) k_grid_count(const float4* __restrict__ cent, int F, const DsnGrid* __restrict__ gp,
                                                     float* __restrict__ u2, int32_t* __restrict__ offsets) {
    const DsnGrid g = *gp;
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= g.ncell) return;
    float blo[3], bhi[3];
    dsn_cell_box(g, cell, blo, bhi);
    float m = INFINITY;
    for (int f = lane; f < F; f += 64) m = fminf(m, dsn_box_dmax2(cent[f], blo, bhi));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    int cnt = 0;
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int f = f0 + lane;
        const bool in = f < F && dsn_in_list(dsn_box_dmin2(cent[f < F ? f : 0], blo, bhi), m);
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) { u2[cell] = m; offsets[cell + 1] = cnt; }
}

// exclusive scan of the counts (single block), capacity check
__global__ void __launch_bounds__(1024) k_grid_scan(DsnGrid* __restrict__ g, int32_t* __restrict__ offsets) {
    __shared__ int s_part[1024];
    const int n = g->ncell;
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int a = min(n, t * per), b = min(n, a + per);
    int sum = 0;
    for (int i = a; i < b; ++i) sum += offsets[i + 1];
    s_part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
        int v = (t >= o) ? s_part[t - o] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;   // exclusive prefix of this thread's chunk
    for (int i = a; i < b; ++i) { int c = offsets[i + 1]; run += c; offsets[i + 1] = run; }
    if (t == 0) offsets[0] = 0;
    if (t == 1023) { g->total = s_part[1023]; g->ok = (s_part[1023] <= g->cap) ? 1 : 0; }
}

// pass 3: write the lists in ascending face order (ballot compaction keeps the order)
template <bool INLINE>
__global__ void __launch_bounds__(256) k_grid_fill(const float4* __restrict__ cent, int F, const DsnGrid* __restrict__ gp,
                                                    const float* __restrict__ u2, const int32_t* __restrict__ offsets,
                                                    void* __restrict__ list) {
    const DsnGrid g = *gp;
    if (!g.ok) return;
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= g.ncell) return;
    float blo[3], bhi[3];
    dsn_cell_box(g, cell, blo, bhi);
    const float m = u2[cell];
    int base = offsets[cell];
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int f = f0 + lane;
        const float4 c = cent[f < F ? f : 0];   // .w already holds the face index bits
        const bool in = f < F && dsn_in_list(dsn_box_dmin2(c, blo, bhi), m);
        const unsigned long long mask = __ballot(in);
        if (in) {
            const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (INLINE) reinterpret_cast<float4*>(list)[at] = c;
            else reinterpret_cast<int32_t*>(list)[at] = f;
        }
        base += __popcll(mask);
    }
}

static void dsn_build_level(const float4* cent, int F, const DsnGridView& v, float pad, int target, int maxcell, int cap,
                            bool inline_entries, hipStream_t st) {
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(256), 0, st, cent, F, pad, target, maxcell, cap, v.g);
    hipLaunchKernelGGL(k_grid_count, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets);
    hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, st, v.g, v.offsets);
    if (inline_entries)
        hipLaunchKernelGGL(k_grid_fill<true>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list);
    else
        hipLaunchKernelGGL(k_grid_fill<false>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list);
}

static int dsn_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

void dsn_launch_build_nn(const float4* cent, int F, const DsnNNView& nn, float pad_fine, float pad_coarse, hipStream_t st) {
    const int t_fine = dsn_clampi(3 * F, 512, 44000);
    const int t_coarse = dsn_clampi(F / 3, 64, 5000);
    dsn_build_level(cent, F, nn.fine, pad_fine, t_fine, DSN_NN_FINE_MAXCELL, dsn_nn_fine_cap(F), true, st);
    dsn_build_level(cent, F, nn.coarse, pad_coarse, t_coarse, DSN_NN_COARSE_MAXCELL, dsn_nn_coarse_cap(F), false, st);
}
