// dsn_nn.hip - device-side construction of the exact nearest-centroid lists (see dsn_nn.h).
// HBM/L2-bound integer+float work: one wavefront per grid cell sweeps the centroid table (coalesced
// float4 loads, table is L2-resident), wave-level min / ballot compaction, single-block scan.
#include "dsn_common.h"
#include "dsn_kernels.h"
#include <cstdlib>

__global__ void __launch_bounds__(256) k_grid_params(const float4* __restrict__ cent, int F, float pad,
                                                      int target_cells, int maxcell, int cap, DsnGrid* __restrict__ g, int lazy) {
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
        g->lazy = lazy;      // (1: this is all dsn_set_frame_ex does for the level - the frame that uses it builds the lists of the cells it visits)
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

// Build acceleration.  For a super-cell SB (4 x 4 x 4 cells, box = union of their guarded boxes) and any cell B in it:
// dmax(B, c) <= dmax(SB, c) and dmin(SB, c) <= dmin(B, c) for every centroid c (same fp32 expressions, rounding is
// monotonic), hence U(B) <= U(SB), the minimiser of U(B) and every member of L(B) lie in
//   S(SB) = { f : dmin(SB, c_f)^2 <= U(SB)^2 (1 + 1e-5) + 1e-12 },
// and sweeping S(SB) instead of all F centroids yields the SAME U(B) and the SAME list, entry for entry.
// One workgroup per super-cell writes S(SB) in ascending face order (<= DSN_SUPER_CAP entries, else the super-cell is
// marked unusable and its cells sweep the whole table as before).
__device__ __forceinline__ int dsn_super_dims(const DsnGrid& g, int& sx, int& sy, int& sz) {
    sx = (g.nx + DSN_SUPER - 1) / DSN_SUPER; sy = (g.ny + DSN_SUPER - 1) / DSN_SUPER; sz = (g.nz + DSN_SUPER - 1) / DSN_SUPER;
    return sx * sy * sz;
}
__device__ __forceinline__ int dsn_super_of(const DsnGrid& g, int cell) {
    int sx, sy, sz;
    dsn_super_dims(g, sx, sy, sz);
    const int iz = cell % g.nz, iy = (cell / g.nz) % g.ny, ix = cell / (g.nz * g.ny);
    return ((ix / DSN_SUPER) * sy + iy / DSN_SUPER) * sz + iz / DSN_SUPER;
}

__global__ void __launch_bounds__(256) k_grid_super(const float4* __restrict__ cent, int F, const DsnGrid* __restrict__ gp,
                                                     int maxsuper, int32_t* __restrict__ super_cnt,
                                                     float4* __restrict__ super_list, const int32_t* __restrict__ visited, int lazy_build) {
    // visited (optional; the lazy build of a frame, DsnGrid::lazy): per-cell sample counts of the frame - only super-cells with a visited
    // cell are swept, and only for a level that is waiting for its lists
    const DsnGrid g = *gp;
    if (lazy_build && !g.lazy) return;
    int sx, sy, sz;
    const int nsuper = dsn_super_dims(g, sx, sy, sz);
    const int sb = blockIdx.x;
    if (sb >= nsuper || nsuper > maxsuper) return;
    const int kz = sb % sz, ky = (sb / sz) % sy, kx = sb / (sz * sy);
    // union of the guarded cell boxes: lo of the first cell, hi of the last cell (the cells' own expressions)
    const int x0 = kx * DSN_SUPER, y0 = ky * DSN_SUPER, z0 = kz * DSN_SUPER;
    const int x1 = min(x0 + DSN_SUPER, g.nx) - 1, y1 = min(y0 + DSN_SUPER, g.ny) - 1, z1 = min(z0 + DSN_SUPER, g.nz) - 1;
    if (visited) {
        int any = 0;
        if (threadIdx.x < DSN_SUPER * DSN_SUPER * DSN_SUPER) {
            const int cx = x0 + (int)threadIdx.x / (DSN_SUPER * DSN_SUPER), cy = y0 + ((int)threadIdx.x / DSN_SUPER) % DSN_SUPER,
                      cz = z0 + (int)threadIdx.x % DSN_SUPER;
            if (cx <= x1 && cy <= y1 && cz <= z1) any = visited[(cx * g.ny + cy) * g.nz + cz] > 0;
        }
        if (!__syncthreads_or(any)) return;
    }
    float blo[3], bhi[3], t0[3], t1[3];
    dsn_cell_box(g, (x0 * g.ny + y0) * g.nz + z0, blo, t1);
    dsn_cell_box(g, (x1 * g.ny + y1) * g.nz + z1, t0, bhi);
    __shared__ float s_m[256];
    __shared__ int s_cnt[4][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int SUP_U = 4;                          // rounds of 256 centroids requested together (latency-bound sweeps: see k_grid_count)
    float m = INFINITY;
    for (int f0 = 0; f0 < F; f0 += 256 * SUP_U) {
        float4 c[SUP_U];
#pragma unroll
        for (int u = 0; u < SUP_U; ++u) { const int f = f0 + 256 * u + t; c[u] = cent[f < F ? f : 0]; }
#pragma unroll
        for (int u = 0; u < SUP_U; ++u) if (f0 + 256 * u + t < F) m = fminf(m, dsn_box_dmax2(c[u], blo, bhi));
    }
    s_m[t] = m;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (t < o) s_m[t] = fminf(s_m[t], s_m[t + o]);
        __syncthreads();
    }
    m = s_m[0];
    int base = 0;
    float4* out = super_list + (size_t)sb * DSN_SUPER_CAP;
    for (int f0 = 0; f0 < F; f0 += 256 * SUP_U) {    // (one barrier pair per SUP_U rounds; the entries land in ascending face order as before)
        float4 c[SUP_U];
        bool in[SUP_U];
        unsigned long long mask[SUP_U];
#pragma unroll
        for (int u = 0; u < SUP_U; ++u) { const int f = f0 + 256 * u + t; c[u] = cent[f < F ? f : 0]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SUP_U; ++u) {
            in[u] = f0 + 256 * u + t < F && dsn_in_list(dsn_box_dmin2(c[u], blo, bhi), m);
            mask[u] = __ballot(in[u]);
            if (lane == 0) s_cnt[u][wave] = __popcll(mask[u]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SUP_U; ++u) {
            int at = base + __popcll(mask[u] & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) at += s_cnt[u][w];
            if (in[u] && at < DSN_SUPER_CAP) out[at] = c[u];
            base += s_cnt[u][0] + s_cnt[u][1] + s_cnt[u][2] + s_cnt[u][3];
        }
    }
    if (t == 0) super_cnt[sb] = base;
}

// the table a cell sweeps: its super-cell's superset when there is a usable one, else all centroids
__device__ __forceinline__ const float4* dsn_cell_source(const DsnGrid& g, int cell, const float4* cent, int F, int maxsuper,
                                                         const int32_t* super_cnt, const float4* super_list, int& n) {
    n = F;
    if (!super_cnt) return cent;
    int sx, sy, sz;
    if (dsn_super_dims(g, sx, sy, sz) > maxsuper) return cent;
    const int sb = dsn_super_of(g, cell);
    const int c = super_cnt[sb];
    if (c > DSN_SUPER_CAP) return cent;
    n = c;
    return super_list + (size_t)sb * DSN_SUPER_CAP;
}

#ifndef DSN_GRID_U
#define DSN_GRID_U 4
#endif
// pass 1+2: U(B)^2 and the list length of every cell (one wavefront per cell)
__global__ void __launch_bounds__(256) k_grid_count(const float4* __restrict__ cent_all, int F_all, const DsnGrid* __restrict__ gp,
                                                     float* __restrict__ u2, int32_t* __restrict__ offsets, int maxsuper,
                                                     const int32_t* __restrict__ super_cnt, const float4* __restrict__ super_list,
                                                     const int32_t* __restrict__ visited, int lazy_build,
                                                     unsigned long long* __restrict__ member) {
    // member (optional): the membership bits of the second sweep, 64 entries of the superset per word - k_grid_fill places the entries
    // from them instead of sweeping the superset a third time (cells that sweep the whole table have no words: they are swept again)
    const DsnGrid g = *gp;
    if (lazy_build && !g.lazy) return;
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= g.ncell) return;
    if (visited && visited[cell] <= 0) {      // (lazy build: a cell no sample of the frame lies in gets an empty list)
        if (lane == 0) offsets[cell + 1] = 0;
        return;
    }
    float blo[3], bhi[3];
    dsn_cell_box(g, cell, blo, bhi);
    int F;
    const float4* __restrict__ cent = dsn_cell_source(g, cell, cent_all, F_all, maxsuper, super_cnt, super_list, F);
    // The sweeps are bound by the latency of their loads (one wave per cell, ~31 rounds of 64 entries from L2), not by their arithmetic:
    // GRID_U rounds' entries are requested together (the last session of round 6: k_grid_count 0.138 -> 0.120 ms per frame; rounds 1-5
    // waited for every round of 64).
    constexpr int GRID_U = DSN_GRID_U;
    float m = INFINITY;
    for (int f0 = 0; f0 < F; f0 += 64 * GRID_U) {
        float4 c[GRID_U];
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) { const int f = f0 + 64 * u + lane; c[u] = cent[f < F ? f : 0]; }
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) if (f0 + 64 * u + lane < F) m = fminf(m, dsn_box_dmax2(c[u], blo, bhi));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fminf(m, __shfl_xor(m, o));
    int cnt = 0;
    unsigned long long* __restrict__ mw = (member && cent != cent_all) ? member + (size_t)cell * (DSN_SUPER_CAP / 64) : nullptr;
    for (int f0 = 0; f0 < F; f0 += 64 * GRID_U) {
        float4 c[GRID_U];
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) { const int f = f0 + 64 * u + lane; c[u] = cent[f < F ? f : 0]; }
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) {
            const int fb = f0 + 64 * u;                  // (wave-uniform)
            const bool in = fb + lane < F && dsn_in_list(dsn_box_dmin2(c[u], blo, bhi), m);
            const unsigned long long mk = __ballot(in);
            cnt += __popcll(mk);
            if (mw && lane == 0 && fb < F) mw[fb >> 6] = mk;
        }
    }
    if (lane == 0) { u2[cell] = m; offsets[cell + 1] = cnt; }
}

// exclusive scan of the counts (single block), capacity check
// exclusive prefix of one value per thread over a 1024-thread block (wave shuffles + one LDS hop); total in *tot
#define SCAN_PER 3                        // cells per thread and tile: 3 072-cell tiles = 12 KB of LDS (with 11 the two scan kernels
                                          // held 45 KB and could not share a compute unit with a persistent field workgroup: 104-137 KB of 160)
__device__ __forceinline__ int dsn_block_exscan(int v, int* s_w, int& tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
    __syncthreads();                      // (s_w may still be read from the previous call)
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int wp = 0, t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int x = s_w[k]; if (k < wave) wp += x; t += x; }
    tot = t;
    return wp + inc - v;
}

// Exclusive scan of the per-cell counts (offsets[i + 1] holds count(i) on entry; inclusive sums land in place, offsets[0] = 0).
// Rounds 1-5: ONE workgroup walked the cells in tiles of 3 072 (0.030 ms for 55 k cells - a fixed cost of every frame, every share of
// a partitioned frame and every training step).  Now DSN_SCAN_BLOCKS workgroups: k_grid_scan_local scans its tile of
// DSN_SCAN_TILE cells in place and leaves the tile's total in the 192 spare bytes behind the level's 64-byte header,
// k_grid_scan_add adds the totals of the tiles in front (integer sums: the same offsets bit for bit) and sets the header's flags.
#define DSN_SCAN_TILE 2048
#define DSN_SCAN_BLOCKS(maxcell) (((maxcell) + DSN_SCAN_TILE - 1) / DSN_SCAN_TILE)
static_assert(sizeof(DsnGrid) + 4 * DSN_SCAN_BLOCKS(DSN_NN_FINE_MAXCELL) <= 256, "the tile totals live in the header's 256-byte slot");
__device__ __forceinline__ int32_t* dsn_grid_tile_totals(DsnGrid* g) { return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(g) + sizeof(DsnGrid)); }
__global__ void __launch_bounds__(1024) k_grid_scan_local(DsnGrid* __restrict__ g, int32_t* __restrict__ offsets, int lazy_build) {
    __shared__ int s_w[16];
    if (lazy_build && !g->lazy) return;      // (block-uniform: the level holds the lists of every cell already)
    const int n = g->ncell;
    const int t = threadIdx.x, i0 = blockIdx.x * DSN_SCAN_TILE + 2 * t;
    const int v0 = i0 < n ? offsets[i0 + 1] : 0, v1 = i0 + 1 < n ? offsets[i0 + 2] : 0;
    int tot;
    const int ex = dsn_block_exscan(v0 + v1, s_w, tot);
    if (i0 < n) offsets[i0 + 1] = ex + v0;
    if (i0 + 1 < n) offsets[i0 + 2] = ex + v0 + v1;
    if (t == 0) dsn_grid_tile_totals(g)[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_grid_scan_add(DsnGrid* __restrict__ g, int32_t* __restrict__ offsets, int lazy_build) {
    if (lazy_build && !g->lazy) return;      // (block 0 changes the flag below while others may still read it - between non-zero values only)
    const int n = g->ncell;
    const int t = threadIdx.x, b = blockIdx.x, i0 = b * DSN_SCAN_TILE + 2 * t;
    const int32_t* __restrict__ tt = dsn_grid_tile_totals(g);
    int front = 0, all = 0;
    for (int j = 0; j < (int)gridDim.x; ++j) { const int x = tt[j]; all += x; if (j < b) front += x; }      // (uniform: scalar loads)
    if (b > 0 && front) {
        if (i0 < n) offsets[i0 + 1] += front;
        if (i0 + 1 < n) offsets[i0 + 2] += front;
    }
    if (b == 0 && t == 0) {
        offsets[0] = 0; g->total = all;
        // lazy build: `ok` stays 0 - the lists cover the visited cells only, good for the frame's own fused search and nothing else
        // (lazy_build = 2, the COMPLETING build of a lazily set level - every cell, dsn_launch_build_nn_complete: 3 = "fits, being
        //  filled"; k_grid_complete turns that into lazy = 0 / ok = 1 behind the fill)
        if (lazy_build) g->lazy = (all <= g->cap) ? (lazy_build == 2 ? 3 : 2) : 1;
        else g->ok = (all <= g->cap) ? 1 : 0;
    }
}
static void dsn_launch_grid_scan(const DsnGridView& v, int maxcell, int lazy_build, hipStream_t st) {
    hipLaunchKernelGGL(k_grid_scan_local, dim3(DSN_SCAN_BLOCKS(maxcell)), dim3(1024), 0, st, v.g, v.offsets, lazy_build);
    hipLaunchKernelGGL(k_grid_scan_add, dim3(DSN_SCAN_BLOCKS(maxcell)), dim3(1024), 0, st, v.g, v.offsets, lazy_build);
}

// pass 3: write the lists in ascending face order (ballot compaction keeps the order)
template <bool INLINE>
__global__ void __launch_bounds__(256) k_grid_fill(const float4* __restrict__ cent_all, int F_all, const DsnGrid* __restrict__ gp,
                                                    const float* __restrict__ u2, const int32_t* __restrict__ offsets,
                                                    void* __restrict__ list, int maxsuper, const int32_t* __restrict__ super_cnt,
                                                    const float4* __restrict__ super_list, const int32_t* __restrict__ visited, int lazy_build,
                                                    const unsigned long long* __restrict__ member) {
    const DsnGrid g = *gp;
    if (lazy_build ? g.lazy != (lazy_build == 2 ? 3 : 2) : !g.ok) return;
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= g.ncell) return;
    if (visited && visited[cell] <= 0) return;
    int F;
    const float4* __restrict__ cent = dsn_cell_source(g, cell, cent_all, F_all, maxsuper, super_cnt, super_list, F);
    int base = offsets[cell];
    if (member && cent != cent_all) {
        // the membership words k_grid_count left (round 6): lane j holds word j (F <= DSN_SUPER_CAP = 64 words) and the number of members in
        // front of it; only words with members are visited, and only to move their entries - the same entries at the same places
        static_assert(DSN_SUPER_CAP / 64 <= 64, "one membership word per lane");
        const int nw = (F + 63) >> 6;
        const unsigned long long mine = lane < nw ? member[(size_t)cell * (DSN_SUPER_CAP / 64) + lane] : 0ull;
        const int pc = __popcll(mine);
        int inc = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
        const int front = inc - pc;
        unsigned long long todo = __ballot(pc > 0);
        constexpr int FILL_U = 4;                    // words in flight (the loop is bound by the latency of its gathers)
        while (todo) {                               // (wave-uniform)
            int j[FILL_U];
            unsigned long long mk[FILL_U];
            float4 c[FILL_U];
#pragma unroll
            for (int u = 0; u < FILL_U; ++u) {
                j[u] = todo ? __ffsll((long long)todo) - 1 : -1;
                todo &= todo - 1;                    // (0 stays 0)
                mk[u] = j[u] >= 0 ? __shfl(mine, j[u]) : 0ull;
                c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((mk[u] >> lane) & 1ull) c[u] = cent[(j[u] << 6) + lane];
            }
#pragma unroll
            for (int u = 0; u < FILL_U; ++u) {
                if (j[u] < 0) continue;              // (wave-uniform)
                const int b0 = base + __shfl(front, j[u]);
                if ((mk[u] >> lane) & 1ull) {
                    const int at = b0 + __popcll(mk[u] & ((1ull << lane) - 1ull));
                    if (INLINE) reinterpret_cast<float4*>(list)[at] = c[u];
                    else reinterpret_cast<int32_t*>(list)[at] = __float_as_int(c[u].w);
                }
            }
        }
        return;
    }
    float blo[3], bhi[3];
    dsn_cell_box(g, cell, blo, bhi);
    const float m = u2[cell];
    constexpr int GRID_U = DSN_GRID_U;                // (rounds of 64 entries requested together: see k_grid_count)
    for (int f0 = 0; f0 < F; f0 += 64 * GRID_U) {
        float4 c[GRID_U];
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) { const int f = f0 + 64 * u + lane; c[u] = cent[f < F ? f : 0]; }   // .w already holds the face index bits
#pragma unroll
        for (int u = 0; u < GRID_U; ++u) {
            const bool in = f0 + 64 * u + lane < F && dsn_in_list(dsn_box_dmin2(c[u], blo, bhi), m);
            const unsigned long long mask = __ballot(in);
            if (in) {
                const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (INLINE) reinterpret_cast<float4*>(list)[at] = c[u];
                else reinterpret_cast<int32_t*>(list)[at] = __float_as_int(c[u].w);
            }
            base += __popcll(mask);
        }
    }
}

// the membership words between k_grid_count and k_grid_fill: only where the cells sweep supersets (DSN_NN_NO_MEMBER=1: cross-check
// switch - k_grid_fill sweeps again, as rounds 1-5)
static unsigned long long* dsn_member_words(const DsnGridView& vv) {
    const bool off = getenv("DSN_NN_NO_MEMBER") != nullptr;
    return (vv.super_cnt && !off) ? vv.member : nullptr;
}
static void dsn_build_level(const float4* cent, int F, const DsnGridView& v, float pad, int target, int maxcell, int cap,
                            bool inline_entries, hipStream_t st, bool params_only = false) {
    const int maxsuper = dsn_grid_maxsuper(maxcell);
    DsnGridView vv = v;
    if (getenv("DSN_NN_NO_SUPER")) vv.super_cnt = nullptr;     // cross-check switch: build with full sweeps
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(256), 0, st, cent, F, pad, target, maxcell, cap, v.g, params_only ? 1 : 0);
    if (params_only) return;
    const int32_t* none = nullptr;
    if (vv.super_cnt)
        hipLaunchKernelGGL(k_grid_super, dim3(maxsuper), dim3(256), 0, st, cent, F, v.g, maxsuper, vv.super_cnt, vv.super_list, none, 0);
    hipLaunchKernelGGL(k_grid_count, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, maxsuper,
                       (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, none, 0, dsn_member_words(vv));
    dsn_launch_grid_scan(v, maxcell, 0, st);
    if (inline_entries)
        hipLaunchKernelGGL(k_grid_fill<true>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list,
                           maxsuper, (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, none, 0, (const unsigned long long*)dsn_member_words(vv));
    else
        hipLaunchKernelGGL(k_grid_fill<false>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list,
                           maxsuper, (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, none, 0, (const unsigned long long*)dsn_member_words(vv));
}

// The lists of a LAZY fine level (DsnGrid::lazy = 1 after dsn_set_frame_ex with DSN_FRAME_LAZY_LISTS), for the cells the frame's samples
// visit: visited[cell] = samples of the frame in that cell (the sampler's classification counts, dsn_nns_classify_one).  Same kernels,
// same sweeps, same lists entry for entry as the full build - for fewer cells: a 512 x 512 frame's samples visit 48 % of the posed
// mesh's fine cells, an eighth of its rays (a rank's block of a partitioned frame) a tenth.  No-ops on a level that holds every
// cell's lists (lazy = 0).
void dsn_launch_build_nn_visited(const float4* cent, int F, const DsnNNView& nn, const int32_t* visited, hipStream_t st) {
    const DsnGridView& v = nn.fine;
    const int maxcell = DSN_NN_FINE_MAXCELL, maxsuper = dsn_grid_maxsuper(maxcell);
    DsnGridView vv = v;
    if (getenv("DSN_NN_NO_SUPER")) vv.super_cnt = nullptr;
    if (getenv("DSN_LAZY_ALL_CELLS")) visited = nullptr;      // (cross-check switch: the lazy build for every cell)
    if (vv.super_cnt)
        hipLaunchKernelGGL(k_grid_super, dim3(maxsuper), dim3(256), 0, st, cent, F, v.g, maxsuper, vv.super_cnt, vv.super_list, visited, 1);
    hipLaunchKernelGGL(k_grid_count, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, maxsuper,
                       (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, visited, 1, dsn_member_words(vv));
    dsn_launch_grid_scan(v, maxcell, 1, st);
    hipLaunchKernelGGL(k_grid_fill<true>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list,
                       maxsuper, (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, visited, 1, (const unsigned long long*)dsn_member_words(vv));
}

// A lazily set level completed for EVERY cell (a lazily set frame rendered outside the fused cell-major path: small ray batches,
// DSN_NN_UNFUSED, the exhaustive cross-check).  The device header decides: the sweeps run while it says lazy != 0 and leave a complete
// level (lazy = 0, ok = 1) behind, so the second chunk of a chunked frame finds nothing to do (round 5 re-ran a full
// dsn_launch_build_nn - grid parameters, coarse level switched off again - on every such call).  The coarse level is not touched.
__global__ void k_grid_complete(DsnGrid* __restrict__ g) {
    if (g->lazy == 3) { g->lazy = 0; g->ok = 1; }
}
void dsn_launch_build_nn_complete(const float4* cent, int F, const DsnNNView& nn, hipStream_t st) {
    const DsnGridView& v = nn.fine;
    const int maxcell = DSN_NN_FINE_MAXCELL, maxsuper = dsn_grid_maxsuper(maxcell);
    DsnGridView vv = v;
    if (getenv("DSN_NN_NO_SUPER")) vv.super_cnt = nullptr;
    const int32_t* none = nullptr;
    if (vv.super_cnt)
        hipLaunchKernelGGL(k_grid_super, dim3(maxsuper), dim3(256), 0, st, cent, F, v.g, maxsuper, vv.super_cnt, vv.super_list, none, 2);
    hipLaunchKernelGGL(k_grid_count, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, maxsuper,
                       (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, none, 2, dsn_member_words(vv));
    dsn_launch_grid_scan(v, maxcell, 2, st);
    hipLaunchKernelGGL(k_grid_fill<true>, dim3((maxcell + 3) / 4), dim3(256), 0, st, cent, F, v.g, v.u2, v.offsets, v.list,
                       maxsuper, (const int32_t*)vv.super_cnt, (const float4*)vv.super_list, none, 2, (const unsigned long long*)dsn_member_words(vv));
    hipLaunchKernelGGL(k_grid_complete, dim3(1), dim3(1), 0, st, v.g);
}

static int dsn_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// a level that is not built: dsn_grid_cell() answers -1 for every point (ok = 0), queries go on to the next level / the sweep
__global__ void k_grid_disable(DsnGrid* __restrict__ g) {
    g->ok = 0; g->ncell = 0; g->total = 0; g->nx = g->ny = g->nz = 0; g->lazy = 0;
    g->cap = 0; g->maxcell = 0;      // (the whole header: the host mirror reads `total` against `cap` - a scene blob starts as uninitialised memory)
}

void dsn_launch_build_nn(const float4* cent, int F, const DsnNNView& nn, float pad_fine, float pad_coarse, hipStream_t st,
                         bool fine_only, bool dense_fine, bool lazy) {
    // lazy (implies fine_only): only the fine grid's geometry - the frame that uses the level builds the lists of the cells its
    // samples visit (dsn_launch_build_nn_visited, behind the sampler's classification)
    // dense_fine (the canonical mesh: built once, queried by a per-lane list scan in k_normal): as many fine cells as the level
    // holds - shorter lists per query; the posed mesh's lists are rebuilt per frame and stay at 3 F cells
    int t_fine = dense_fine ? dsn_clampi(5 * F, 512, 62000) : dsn_clampi(3 * F, 512, 44000);
    if (!dense_fine) {      // DSN_NN_FINE_TARGET (tuning): cells of the posed mesh's fine level (lists stay exact for any cell size)
        static const long long tgt = [] { const char* e = getenv("DSN_NN_FINE_TARGET"); return e ? atoll(e) : 0ll; }();
        if (tgt > 0) t_fine = dsn_clampi((int)tgt, 512, 62000);
    }
    const int t_coarse = dsn_clampi(F / 3, 64, 5000);
    // DSN_NN_FINE_CAP (tests): a smaller LOGICAL capacity of the fine level - provokes the overflow path (level unusable, queries fall
    // through to the coarse level / the sweep, the host mirror warns) on a mesh that fits the real one
    const char* ce = getenv("DSN_NN_FINE_CAP");
    const int fine_cap = ce && atoll(ce) > 0 && atoll(ce) < dsn_nn_fine_cap(F) ? (int)atoll(ce) : dsn_nn_fine_cap(F);
    dsn_build_level(cent, F, nn.fine, pad_fine, t_fine, DSN_NN_FINE_MAXCELL, fine_cap, true, st, lazy);
    if (fine_only || lazy) hipLaunchKernelGGL(k_grid_disable, dim3(1), dim3(1), 0, st, nn.coarse.g);
    else dsn_build_level(cent, F, nn.coarse, pad_coarse, t_coarse, DSN_NN_COARSE_MAXCELL, dsn_nn_coarse_cap(F), false, st);
}

// ---------------------------------------------------------------------------------------------
// Cell-major query of the fine lists (fused path).  k_warp's per-lane list scan is bound by the vector L1
// return path: every lane pulls 16 B per candidate (~290 candidates per sample) even when neighbouring lanes
// read the same entry.  Here the samples are counting-sorted by fine cell first; one wave then owns 128 samples (two per lane)
// of ONE cell, the candidate list is read through the scalar cache (uniform address -> s_load, operands in
// SGPRs) and only the distance arithmetic runs on the VALU.  Same list, same order, same fma chain, same strict
// '<' as dsn_nearest_lists, hence the same index bit for bit.  Samples outside the fine grid keep nn = -1 and
// are searched by k_warp as before.
// ---------------------------------------------------------------------------------------------
#define NNS_THREADS 256
#ifndef NNS_SURVIVORS
#define NNS_SURVIVORS 320     // candidates a wave of k_nns_search keeps in LDS after pruning (5 KB per wave, 20 KB per workgroup)
#endif
#define NNS_PER 128           // samples of one cell a wave of k_nns_search takes (two per lane)

__device__ __forceinline__ void nns_point(const float* __restrict__ pts, const float* __restrict__ ray_o,
                                          const float* __restrict__ ray_d, const float* __restrict__ z_vals, int64_t i, int S,
                                          float* p) {
    if (pts) { p[0] = pts[3 * i]; p[1] = pts[3 * i + 1]; p[2] = pts[3 * i + 2]; return; }
    const int64_t ray = i / S;
    const float z = z_vals[i];      // the same expression as k_warp: bit-identical points
    p[0] = ray_o[3 * ray + 0] + ray_d[3 * ray + 0] * z;
    p[1] = ray_o[3 * ray + 1] + ray_d[3 * ray + 1] * z;
    p[2] = ray_o[3 * ray + 2] + ray_d[3 * ray + 2] * z;
}

__global__ void __launch_bounds__(NNS_THREADS) k_nns_classify(const DsnGrid* __restrict__ gf, const float* __restrict__ pts,
                                                              const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                              const float* __restrict__ z_vals, int64_t N, int S,
                                                              int32_t* __restrict__ cell_of, int32_t* __restrict__ nn,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ outside) {
    const int64_t i = (int64_t)blockIdx.x * NNS_THREADS + threadIdx.x;
    if (gf->lazy) {
        // a lazily set level (DSN_FRAME_LAZY_LISTS) that nobody completed for this call - the two-kernel form is never the one the
        // visited cells' lists were built for: every sample goes to the pass behind it (k_warp: next level / exhaustive sweep, exact)
        if (i < N) { cell_of[i] = -1; if (nn) nn[i] = -1; }
        if (outside) {
            const unsigned long long om = __ballot(i < N);
            if (om && (threadIdx.x & 63) == 0) atomicAdd(outside, __popcll(om));
        }
        return;
    }
    float p[3] = {0.f, 0.f, 0.f};
    if (i < N) nns_point(pts, ray_o, ray_d, z_vals, i, S, p);
    const int c = dsn_nns_classify_one(gf, i, i < N, p[0], p[1], p[2], nullptr, cell_of, counts, outside);
    if (i < N && c < 0 && nn) nn[i] = -1;
}

// exclusive scans over the cells: sample offsets and wave offsets (ceil(count / NNS_PER) waves per cell); counts are
// cleared for their second life as scatter cursors.  Single workgroup, LDS-staged tiles (see k_grid_scan).
// lazy_call: THIS render call ran dsn_launch_build_nn_visited on the level (DSN_LAZY_LISTS).  Only then do lists in state lazy = 2
// belong to the samples being searched: a level left in that state by an EARLIER call holds the lists of the cells that call's rays
// visited - a later call without the flag must not walk them (cells only the new rays visit have empty lists, their
// samples came back with a wrong face and no error); it gets no waves here and k_nns_scatter* hands its samples to the exhaustive pass.
__device__ __forceinline__ bool nns_lists_usable(const DsnGrid* __restrict__ gf, int lazy_call) { return gf->ok || (lazy_call && gf->lazy == 2); }
__global__ void __launch_bounds__(1024) k_nns_scan(const DsnGrid* __restrict__ gf, int32_t* __restrict__ counts,
                                                    int32_t* __restrict__ offs, int32_t* __restrict__ wave_offs,
                                                    int32_t* __restrict__ totals, int keep_counts, int lazy_call,
                                                    const int32_t* __restrict__ list_off = nullptr, int seg = 0) {
    // list_off / seg (optional, the far search of the training forward): a cell's waves are multiplied by ceil(list length / seg) -
    // every wave then walks one SEGMENT of the cell's candidate list (k_nns_search_far)
    __shared__ int s_n[1024 * SCAN_PER];
    __shared__ int s_w[16];
    const int ncell = nns_lists_usable(gf, lazy_call) ? gf->ncell : 0;      // (lazy = 2: lists of the visited cells, built for this very search)
    auto waves_of = [&](int cell, int cnt) {
        int wv = (cnt + NNS_PER - 1) / NNS_PER;
        if (list_off && wv > 0 && cell < ncell) { const int len = list_off[cell + 1] - list_off[cell]; wv *= len > seg ? (len + seg - 1) / seg : 1; }
        return wv;
    };
    const int t = threadIdx.x;
    int carry_a = 0, carry_b = 0;
    for (int base = 0; base < ncell; base += 1024 * SCAN_PER) {
        for (int i = t; i < 1024 * SCAN_PER; i += 1024) s_n[i] = base + i < ncell ? counts[base + i] : 0;
        __syncthreads();
        int v[SCAN_PER], a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { v[k] = s_n[t * SCAN_PER + k]; a += v[k]; b += waves_of(base + t * SCAN_PER + k, v[k]); }
        int tot_a, tot_b;
        int ra = carry_a + dsn_block_exscan(a, s_w, tot_a);
        int rb = carry_b + dsn_block_exscan(b, s_w, tot_b);
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { s_n[t * SCAN_PER + k] = ra; ra += v[k]; }
        __syncthreads();
        for (int i = t; i < 1024 * SCAN_PER; i += 1024) if (base + i < ncell) offs[base + i] = s_n[i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { s_n[t * SCAN_PER + k] = rb; rb += waves_of(base + t * SCAN_PER + k, v[k]); }
        __syncthreads();
        for (int i = t; i < 1024 * SCAN_PER; i += 1024)
            if (base + i < ncell) { wave_offs[base + i] = s_n[i]; if (!keep_counts) counts[base + i] = 0; }
        carry_a += tot_a; carry_b += tot_b;
        __syncthreads();
    }
    if (t == 0) { totals[0] = carry_b; totals[1] = carry_a; }
}

// The same two scans by DSN_NN_SCAN_BLOCKS workgroups (keep_counts = 1 only: nothing is written where another block reads).  Every
// block first sums what lies in front of its tile of 1024 cells (coalesced reads of at most 256 KB from L2, the same integer sums in
// another order), then scans the tile; the block that holds the last cell writes the totals.  One launch, no scratch, no flags:
// 0.061 -> ~0.01 ms per frame / share / training step.
__global__ void __launch_bounds__(1024) k_nns_scan_mb(const DsnGrid* __restrict__ gf, const int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ offs, int32_t* __restrict__ wave_offs,
                                                       int32_t* __restrict__ totals, int lazy_call,
                                                       const int32_t* __restrict__ list_off = nullptr, int seg = 0) {
    __shared__ int s_w[16];
    const int ncell = nns_lists_usable(gf, lazy_call) ? gf->ncell : 0;
    const int t = threadIdx.x, base = blockIdx.x * 1024;
    if (base >= ncell && blockIdx.x != 0) return;                    // (block-uniform; block 0 writes the totals of an empty level)
    auto waves_of = [&](int cell, int cnt) {
        int wv = (cnt + NNS_PER - 1) / NNS_PER;
        if (list_off && wv > 0) { const int len = list_off[cell + 1] - list_off[cell]; wv *= len > seg ? (len + seg - 1) / seg : 1; }
        return wv;
    };
    int a = 0, b = 0;
    if (!list_off) {
        const int4* __restrict__ c4 = reinterpret_cast<const int4*>(counts);      // (base is a multiple of 1024: whole int4s)
        for (int i = t; i < base / 4; i += 1024) {
            const int4 c = c4[i];
            a += c.x + c.y + c.z + c.w;
            b += (c.x + NNS_PER - 1) / NNS_PER + (c.y + NNS_PER - 1) / NNS_PER + (c.z + NNS_PER - 1) / NNS_PER + (c.w + NNS_PER - 1) / NNS_PER;
        }
    } else {
        for (int i = t; i < base; i += 1024) { const int c = counts[i]; a += c; b += waves_of(i, c); }
    }
    int front_a, front_b, tot_a, tot_b;
    (void)dsn_block_exscan(a, s_w, front_a);
    (void)dsn_block_exscan(b, s_w, front_b);
    const int i = base + t;
    const int c = i < ncell ? counts[i] : 0, wv = i < ncell ? waves_of(i, c) : 0;
    const int ea = front_a + dsn_block_exscan(c, s_w, tot_a);
    const int eb = front_b + dsn_block_exscan(wv, s_w, tot_b);
    if (i < ncell) { offs[i] = ea; wave_offs[i] = eb; }
    if (t == 0 && base + 1024 >= ncell) { totals[0] = front_b + tot_b; totals[1] = front_a + tot_a; }
}
#define DSN_NN_SCAN_BLOCKS(maxcell) (((maxcell) + 1023) / 1024)

// wave w -> its cell (cells with many samples own several consecutive waves)
__global__ void __launch_bounds__(NNS_THREADS) k_nns_expand(const DsnGrid* __restrict__ gf, const int32_t* __restrict__ wave_offs,
                                                            const int32_t* __restrict__ totals, int32_t* __restrict__ wave_cell, int lazy_call) {
    const int c = blockIdx.x * NNS_THREADS + threadIdx.x;
    const int ncell = nns_lists_usable(gf, lazy_call) ? gf->ncell : 0;
    if (c >= ncell) return;
    const int w0 = wave_offs[c], w1 = (c + 1 < ncell) ? wave_offs[c + 1] : totals[0];
    for (int w = w0; w < w1; ++w) wave_cell[w] = c;
}

// sorted[pos] = (point, sample id): the search kernel then reads its samples with one coalesced 16-byte load
__global__ void __launch_bounds__(NNS_THREADS) k_nns_scatter(const int32_t* __restrict__ cell_of, const float* __restrict__ pts,
                                                             const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                             const float* __restrict__ z_vals, int64_t N, int S,
                                                             const int32_t* __restrict__ offs, int32_t* __restrict__ cursor,
                                                             float4* __restrict__ sorted, const DsnGrid* __restrict__ gf = nullptr,
                                                             int32_t* __restrict__ cell_rw = nullptr, int32_t* __restrict__ outside = nullptr,
                                                             int lazy_call = 0) {
    const int64_t i = (int64_t)blockIdx.x * NNS_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (gf && gf->lazy && !nns_lists_usable(gf, lazy_call)) {      // (a lazy level without usable lists: see k_nns_scatter_ranked)
        const bool mine = i < N && cell_rw[i] >= 0;
        if (mine) cell_rw[i] = -1;
        const unsigned long long m = __ballot(mine);
        if (m && lane == 0) atomicAdd(outside, __popcll(m));
        return;
    }
    const int c = i < N ? cell_of[i] : -1;
    const NnsRun r = nns_run(c, lane);
    int base = 0;
    if (r.head && c >= 0) base = offs[c] + atomicAdd(cursor + c, r.len);
    base = __shfl(base, r.head_lane);
    if (c >= 0) {
        float p[3];
        nns_point(pts, ray_o, ray_d, z_vals, i, S, p);
        sorted[base + r.rank] = make_float4(p[0], p[1], p[2], __int_as_float((int)i));
    }
}

// the same scatter from the ranks the classification kept (dsn_nns_classify_one, rank_of): position = cell offset + rank, no atomics
// (k_nns_scatter spent its 0.29 ms per 16.8 M samples on them), the per-cell counts stay what the classification counted
__global__ void __launch_bounds__(NNS_THREADS) k_nns_scatter_ranked(int32_t* __restrict__ cell_of, const int32_t* __restrict__ rank_of,
                                                                    const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                                    const float* __restrict__ z_vals, int64_t N, int S,
                                                                    const int32_t* __restrict__ offs, float4* __restrict__ sorted,
                                                                    const DsnGrid* __restrict__ gf, int32_t* __restrict__ outside, int lazy_call) {
    const int64_t i = (int64_t)blockIdx.x * NNS_THREADS + threadIdx.x;
    if (gf->lazy && !nns_lists_usable(gf, lazy_call)) {
        // a lazy level whose visited cells' lists did not fit the capacity (k_grid_scan left lazy = 1; the host mirror sees
        // total > cap in the header and warns), or whose lists are another call's (lazy = 2 without DSN_LAZY_LISTS in this call): the
        // cell-major search has nothing to walk - every sample is handed to the k_warp
        // pass behind it as "outside the fine grid", which sweeps all centroids: the same index, slowly
        const bool mine = i < N && cell_of[i] >= 0;
        if (mine) cell_of[i] = -1;
        const unsigned long long m = __ballot(mine);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(outside, __popcll(m));
        return;
    }
    if (i >= N) return;
    const int c = cell_of[i];
    if (c < 0) return;
    float p[3];
    nns_point(nullptr, ray_o, ray_d, z_vals, i, S, p);
    sorted[offs[c] + rank_of[i]] = make_float4(p[0], p[1], p[2], __int_as_float((int)i));
}

// WARP = true (round 3): the rest of the warp stage (can_render.py:333-379: projection onto the nearest posed face, transparency,
// canonical re-embedding, active list) runs right here, on the point and the face index the search has in registers - no nn[] round
// trip, no second pass over the 16.8 M samples, and the face records of a wave's samples (one cell: a handful of faces) come from L1.
// Same helpers and expressions as k_warp: bit-identical transparent / x_c.  The active list comes out cell-major instead of
// ray-major (its order never reaches a value; the sigma > 0 list inherits it, which puts the lanes of k_normal's waves into the
// same canonical cells).
struct NnsWarp {
    const DsnFaceRec* face_world; const DsnFaceRec* face_canon; uint8_t* transparent; float* x_c; int32_t* active_list;
    int32_t* active_count; int lazy_canon;
};
template <bool WARP>
__global__ void __launch_bounds__(NNS_THREADS) k_nns_search(const int32_t* __restrict__ off_f, const float4* __restrict__ list_f,
                                                            const int32_t* __restrict__ wave_cell, const int32_t* __restrict__ wave_offs,
                                                            const int32_t* __restrict__ totals, const int32_t* __restrict__ offs,
                                                            const int32_t* __restrict__ counts, const float4* __restrict__ sorted,
                                                            int32_t* __restrict__ nn, NnsWarp wp, const float4* __restrict__ cent) {
    // cent (optional, WARP = false): the lists hold face INDICES (the coarse level: 4-byte entries) and the centroids come from
    // this array - two dependent scalar loads per candidate instead of one, hidden by the eight waves a SIMD holds of this kernel
    const int lane = threadIdx.x & 63;
    // XCD-aware block -> wave map.  Consecutive waves work on the same cell and read the same candidate list; workgroup b
    // runs on XCD b % 8 (MI355X_MICROARCH.md), each with its own L2, so the plain map b -> waves 4b .. 4b+3 sends every
    // list through all eight L2s (PMC round 1: 1.2 GB fetched per launch for 0.13 GB algorithmic, L2 hit 35 %).  Here XCD x
    // takes runs of 16 consecutive virtual blocks (64 waves, ~11 cells), dealt round-robin: a cell's list is (mostly) fetched
    // into ONE L2: 0.65 GB per launch, L2 hit 60 %, 0.87 ms (0.89 before).  (One contiguous eighth of the waves per XCD gets
    // 0.58 GB / 67 % - but the kernel is VALU-bound and the work per cell varies over the body: 1.09 ms from the imbalance.)
    const unsigned bq = blockIdx.x >> 3, bx = blockIdx.x & 7u;
    const unsigned vb = ((bq >> 4) * 8u + bx) * 16u + (bq & 15u);
    const int w = __builtin_amdgcn_readfirstlane(vb * (NNS_THREADS / 64) + (threadIdx.x >> 6));
    const int nwaves = totals[0];
    if (WARP ? (int)(vb * (NNS_THREADS / 64)) >= nwaves : w >= nwaves) return;      // (WARP: block-uniform - the append below has barriers)
    const bool wave_on = w < nwaves;
    const int c = __builtin_amdgcn_readfirstlane(wave_cell[wave_on ? w : 0]);
    // TWO samples per lane (slots lane and lane + 64 of the wave's 128): the distance arithmetic runs on packed fp32
    // (v_pk_add / v_pk_mul / v_pk_fma_f32 with the candidate broadcast from SGPRs: 6 instructions per candidate and PAIR of samples
    // instead of 12 - the same IEEE operations per component, so the same distances bit for bit); compare + select stay per sample
    const int slot = (w - __builtin_amdgcn_readfirstlane(wave_offs[c])) * NNS_PER + lane;
    const int cnt_c = __builtin_amdgcn_readfirstlane(counts[c]);
    const bool valid[2] = {wave_on && slot < cnt_c, wave_on && slot + 64 < cnt_c};
    float4 q[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int so = __builtin_amdgcn_readfirstlane(offs[c]);
    if (valid[0]) q[0] = sorted[so + slot];
    if (valid[1]) q[1] = sorted[so + slot + 64];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 px = {q[0].x, q[1].x}, py = {q[0].y, q[1].y}, pz = {q[0].z, q[1].z};
    const int o = __builtin_amdgcn_readfirstlane(off_f[c]);
    const int n = wave_on ? __builtin_amdgcn_readfirstlane(off_f[c + 1]) - o : 0;
    const float4* __restrict__ e = list_f + o;
    float best[2] = {INFINITY, INFINITY};
    int bi[2] = {0, 0};
    // squared distance exactly as dsn_d2: dx * dx, then fma(dy, dy, .), then fma(dz, dz, .) - per component of the pair
    auto step = [&](const float4 a) {
        const f32x2 cx = {a.x, a.x}, cy = {a.y, a.y}, cz = {a.z, a.z};
        const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
        f32x2 d = dx * dx;
        d = __builtin_elementwise_fma(dy, dy, d);
        d = __builtin_elementwise_fma(dz, dz, d);
        const int id = __float_as_int(a.w);
        if (d.x < best[0]) { best[0] = d.x; bi[0] = id; }
        if (d.y < best[1]) { best[1] = d.y; bi[1] = id; }
    };
    int k = 0;
    if (!WARP && cent) {
        const int32_t* __restrict__ ids = reinterpret_cast<const int32_t*>(list_f) + o;
        auto at = [&](int f) { float4 a = cent[f]; a.w = __int_as_float(f); return a; };
        // (far points are few per coarse cell - a wave with at most 64 of them leaves the second slot of every lane empty: one
        //  sample per lane, the plain fma chain of dsn_d2)
        const bool single = cnt_c - (w - __builtin_amdgcn_readfirstlane(wave_offs[c])) * NNS_PER <= 64;      // wave-uniform
        if (single) {
            float b0 = INFINITY;
            int i0 = 0;
            const float qx = q[0].x, qy = q[0].y, qz = q[0].z;
            auto step1 = [&](const float4 a) {
                const float d = dsn_d2(qx, qy, qz, a);
                if (d < b0) { b0 = d; i0 = __float_as_int(a.w); }
            };
            constexpr int NB1 = 4;
            const int nb1 = n / NB1;
            int g1[NB1], g2[NB1];
            float4 c0[NB1], c1[NB1];
            if (nb1 > 0) {
#pragma unroll
                for (int j = 0; j < NB1; ++j) g1[j] = ids[j];
#pragma unroll
                for (int j = 0; j < NB1; ++j) c0[j] = at(g1[j]);
#pragma unroll
                for (int j = 0; j < NB1; ++j) g1[j] = ids[NB1 * (1 < nb1 ? 1 : 0) + j];
                for (int b = 0; b < nb1; ++b) {
                    if (b + 1 < nb1) {
#pragma unroll
                        for (int j = 0; j < NB1; ++j) c1[j] = at(g1[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NB1; ++j) g2[j] = ids[NB1 * (b + 2 < nb1 ? b + 2 : 0) + j];
#pragma unroll
                    for (int j = 0; j < NB1; ++j) step1(c0[j]);
#pragma unroll
                    for (int j = 0; j < NB1; ++j) { c0[j] = c1[j]; g1[j] = g2[j]; }
                }
            }
            for (int kk = NB1 * nb1; kk < n; ++kk) step1(at(ids[kk]));
            if (valid[0]) nn[__float_as_int(q[0].w)] = i0;
            return;
        }
        // Two dependent scalar loads per candidate (index, then centroid), and scalar loads return out of order - the only wait is
        // "all of them".  So the loop runs one batch ahead on each: while batch b is compared, the centroids of batch b + 1 (whose
        // indices arrived a batch ago) and the indices of batch b + 2 are in flight; one wait per batch, behind the arithmetic.
        constexpr int NB = 4;                       // (batches of 8 need more scalar registers than there are: spills into VGPR lanes)
        const int nb = n / NB;
        int f1[NB], f2[NB];
        float4 a0[NB], a1[NB];
        auto load_ids = [&](int b, int (&f)[NB]) {
#pragma unroll
            for (int j = 0; j < NB; ++j) f[j] = ids[NB * (b < nb ? b : 0) + j];      // (beyond the list: batch 0 again, never used)
        };
        auto load_cent = [&](const int (&f)[NB], float4 (&a)[NB]) {
#pragma unroll
            for (int j = 0; j < NB; ++j) a[j] = at(f[j]);
        };
        if (nb > 0) {
            load_ids(0, f1);
            load_cent(f1, a0);                      // a0 = batch 0
            load_ids(1, f1);                        // f1 = indices of batch 1
            for (int b = 0; b < nb; ++b) {
                if (b + 1 < nb) load_cent(f1, a1);  // centroids of batch b + 1
                load_ids(b + 2, f2);                // indices of batch b + 2
#pragma unroll
                for (int j = 0; j < NB; ++j) step(a0[j]);
#pragma unroll
                for (int j = 0; j < NB; ++j) { a0[j] = a1[j]; f1[j] = f2[j]; }
            }
        }
        for (k = NB * nb; k < n; ++k) step(at(ids[k]));
    } else {
    // Per-wave pruning of the cell's candidate list (round 6).  The list holds every face that can be nearest to SOME point of the cell;
    // the wave's 128 samples fill a part of it.  With B = the bounding box of
    // the wave's samples and T = min over the candidates of the squared distance from the candidate to the FARTHEST corner of B (every
    // sample has a candidate within T), a candidate whose squared distance to the NEAREST point of B exceeds T (1 + 1e-4) is farther
    // than the winner from every sample of the wave, whatever the float32 rounding of the two distances (relative 4e-7 each): it can
    // neither win nor tie.  Both bounds are evaluated one candidate per LANE (a sixty-fourth of the per-sample loop's cost per
    // candidate); the survivors go to LDS in list order, so the per-sample loop below sees the same candidates in the same order minus
    // those that cannot matter: the same nearest face, bit for bit.  DSN_NN_NO_PRUNE (experiment builds): off.
    // Measured on the bench frame (profiles/r06_nns_prune.txt): lists of 292 candidates on average, 78 % of them survive (a cell's
    // consecutive samples are spread over most of the cell - the sampler counts them row segment by row segment, depth by depth):
    // 0.885 -> 0.83 ms per frame.  With the samples sub-sorted inside their cell (6-bit Morton key through LDS, one workgroup per cell)
    // 54 % survive and the search takes 0.716 ms - and the sort 0.146: not kept.  More survivors than the wave's LDS holds: drained in rounds.
    bool pruned = false;
    int ns = 0;
#if !defined(DSN_NN_NO_PRUNE)
    __shared__ __attribute__((aligned(16))) float4 s_surv[NNS_THREADS / 64][NNS_SURVIVORS];
    float4* const surv = s_surv[threadIdx.x >> 6];
    if (n >= 96) {                                   // wave-uniform
        const float INF = INFINITY;
        float lo[3] = {fminf(valid[0] ? q[0].x : INF, valid[1] ? q[1].x : INF), fminf(valid[0] ? q[0].y : INF, valid[1] ? q[1].y : INF),
                       fminf(valid[0] ? q[0].z : INF, valid[1] ? q[1].z : INF)};
        float hi[3] = {fmaxf(valid[0] ? q[0].x : -INF, valid[1] ? q[1].x : -INF), fmaxf(valid[0] ? q[0].y : -INF, valid[1] ? q[1].y : -INF),
                       fmaxf(valid[0] ? q[0].z : -INF, valid[1] ? q[1].z : -INF)};
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], m));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m));
            }
        float t = INF;
        for (int k0 = 0; k0 < n; k0 += 64) {
            if (k0 + lane < n) {
                const float4 a = e[k0 + lane];
                const float fx = fmaxf(fabsf(a.x - lo[0]), fabsf(a.x - hi[0])), fy = fmaxf(fabsf(a.y - lo[1]), fabsf(a.y - hi[1])),
                            fz = fmaxf(fabsf(a.z - lo[2]), fabsf(a.z - hi[2]));
                t = fminf(t, fx * fx + fy * fy + fz * fz);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t = fminf(t, __shfl_xor(t, m));
        t = t * 1.0001f;
        if (t < INF) {                               // (wave-uniform; an unbounded box or a non-finite bound: the whole list)
            pruned = true;
            // the survivors collected so far, in list order (a list that leaves more than the wave's LDS holds is drained in rounds)
            auto drain = [&]() {
                __builtin_amdgcn_wave_barrier();
                int j = 0;
                for (; j + 8 <= ns; j += 8) {
                    float4 a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = surv[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) step(a[u]);
                }
                for (; j < ns; ++j) step(surv[j]);
                __builtin_amdgcn_wave_barrier();
                ns = 0;
            };
            for (int k0 = 0; k0 < n; k0 += 64) {
                bool keep = false;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + lane < n) {
                    a = e[k0 + lane];
                    const float gx = fmaxf(fmaxf(lo[0] - a.x, a.x - hi[0]), 0.0f), gy = fmaxf(fmaxf(lo[1] - a.y, a.y - hi[1]), 0.0f),
                                gz = fmaxf(fmaxf(lo[2] - a.z, a.z - hi[2]), 0.0f);
                    keep = !(gx * gx + gy * gy + gz * gz > t);      // (a NaN bound keeps the candidate)
                }
                const unsigned long long mk = __ballot(keep);
                const int add = __popcll(mk);
                if (ns + add > NNS_SURVIVORS) drain();               // (wave-uniform)
                if (keep) surv[ns + __popcll(mk & ((1ull << lane) - 1ull))] = a;
                ns += add;
            }
            drain();
        }
    }
    if (!pruned) {
#endif
    for (; k + 8 <= n; k += 8) {                     // wave-uniform addresses: 128 B of candidates per scalar-load batch
        float4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = e[k + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) step(a[j]);
    }
    for (; k < n; ++k) step(e[k]);
#if !defined(DSN_NN_NO_PRUNE)
    }
#endif
    }
    if (!WARP) {
        if (valid[0]) nn[__float_as_int(q[0].w)] = bi[0];
        if (valid[1]) nn[__float_as_int(q[1].w)] = bi[1];
        return;
    }
    bool active[2] = {false, false};
    int64_t idx[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        idx[h2] = (int64_t)__float_as_int(q[h2].w);
        if (valid[h2]) {
            const float p[3] = {q[h2].x, q[h2].y, q[h2].z};
            const DsnFaceRec fw = dsn_load_face(wp.face_world, bi[h2]);
            float u, v, h, xc[3] = {0.f, 0.f, 0.f};
            dsn_project(p, fw, u, v, h);
            const bool tr = (u > 5.f) || (u < -4.f) || (v > 5.f) || (v < -4.f) || (fabsf(h) > 0.1f);
            if (!(tr && wp.lazy_canon)) {
                const DsnFaceRec fc = dsn_load_face(wp.face_canon, bi[h2]);
                dsn_map2face(u, v, h, fc, xc);
            }
            const int64_t i = idx[h2];
            wp.transparent[i] = tr ? 1 : 0;
            wp.x_c[3 * i] = xc[0]; wp.x_c[3 * i + 1] = xc[1]; wp.x_c[3 * i + 2] = xc[2];
            active[h2] = !tr;
        }
    }
    if (wp.active_list) {      // workgroup-aggregated append, as in k_warp: one atomic per 512 samples
        __shared__ int s_cnt[NNS_THREADS / 64][2];
        __shared__ int s_base;
        const unsigned long long m0 = __ballot(active[0]), m1 = __ballot(active[1]);
        const int wave = threadIdx.x >> 6;
        if (lane == 0) { s_cnt[wave][0] = __popcll(m0); s_cnt[wave][1] = __popcll(m1); }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int kk = 0; kk < NNS_THREADS / 64; ++kk) tot += s_cnt[kk][0] + s_cnt[kk][1];
            s_base = tot ? atomicAdd(wp.active_count, tot) : 0;
        }
        __syncthreads();
        int off = s_base;
        for (int kk = 0; kk < wave; ++kk) off += s_cnt[kk][0] + s_cnt[kk][1];
        if (active[0]) wp.active_list[off + __popcll(m0 & ((1ull << lane) - 1ull))] = (int32_t)idx[0];
        if (active[1]) wp.active_list[off + s_cnt[wave][0] + __popcll(m1 & ((1ull << lane) - 1ull))] = (int32_t)idx[1];
    }
}

// the fused form: nearest face + the rest of the warp stage for every sample inside the fine grid; *outside (device int) counts the
// samples it leaves alone (cell_of < 0).  cell_of: N ints; sorted: N float4 of scratch
void dsn_launch_nn_cellmajor_warp(const DsnNNView& v, const float* ray_o, const float* ray_d, const float* z_vals, int64_t N, int S,
                                  int32_t* cell_of, void* sorted, void* small, const DsnFaceRec* face_world, const DsnFaceRec* face_canon,
                                  uint8_t* transparent, float* x_c, int32_t* active_list, int32_t* active_count, bool lazy_canon,
                                  int32_t** outside, hipStream_t st, bool classified, bool lazy_call) {
    // classified: the sampler has filled cell_of / counts / the outside counter already (dsn_nn_cellmajor_begin + dsn_launch_sample_gg)
    char* q = (char*)small;
    int32_t* counts = (int32_t*)q;     q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* offs = (int32_t*)q;       q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* wave_offs = (int32_t*)q;  q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* totals = (int32_t*)q;     q += 256;
    int32_t* wave_cell = (int32_t*)q;
    *outside = totals + 2;
    const dim3 gN((unsigned)((N + NNS_THREADS - 1) / NNS_THREADS)), b(NNS_THREADS);
    if (!classified) {
        (void)hipMemsetAsync(counts, 0, 4 * (size_t)(DSN_NN_FINE_MAXCELL + 1), st);
        (void)hipMemsetAsync(totals, 0, 256, st);
        hipLaunchKernelGGL(k_nns_classify, gN, b, 0, st, v.fine.g, (const float*)nullptr, ray_o, ray_d, z_vals, N, S, cell_of, (int32_t*)nullptr,
                           counts, totals + 2);
    }
    // classified by the sampler: it also kept every sample's rank inside its cell (cell_of + N) - the scatter places by rank
    // (DSN_NN_ATOMIC_SCATTER: A/B switch, round 3's scatter with atomic cursors; a lazily built level always takes the ranked form - it
    //  is the one that hands the samples over when the visited cells' lists did not fit)
    const bool ranked = classified && (lazy_call || !getenv("DSN_NN_ATOMIC_SCATTER"));
    const int lc = lazy_call ? 1 : 0;
    if (ranked)
        hipLaunchKernelGGL(k_nns_scan_mb, dim3(DSN_NN_SCAN_BLOCKS(DSN_NN_FINE_MAXCELL)), dim3(1024), 0, st, v.fine.g, (const int32_t*)counts, offs, wave_offs,
                           totals, lc, (const int32_t*)nullptr, 0);
    else
        hipLaunchKernelGGL(k_nns_scan, dim3(1), dim3(1024), 0, st, v.fine.g, counts, offs, wave_offs, totals, 0, lc);
    hipLaunchKernelGGL(k_nns_expand, dim3(DSN_NN_FINE_MAXCELL / NNS_THREADS), b, 0, st, v.fine.g, wave_offs, totals, wave_cell, lc);
    if (ranked)
        hipLaunchKernelGGL(k_nns_scatter_ranked, gN, b, 0, st, cell_of, (const int32_t*)(cell_of + N), ray_o, ray_d, z_vals, N, S, offs,
                           (float4*)sorted, (const DsnGrid*)v.fine.g, totals + 2, lc);
    else
        hipLaunchKernelGGL(k_nns_scatter, gN, b, 0, st, cell_of, (const float*)nullptr, ray_o, ray_d, z_vals, N, S, offs, counts, (float4*)sorted,
                           (const DsnGrid*)v.fine.g, cell_of, totals + 2, lc);
    const int64_t max_waves = N / NNS_PER + DSN_NN_FINE_MAXCELL + 1;
    const NnsWarp wp = {face_world, face_canon, transparent, x_c, active_list, active_count, lazy_canon ? 1 : 0};
    hipLaunchKernelGGL(k_nns_search<true>, dim3((unsigned)((max_waves + 3) / 4)), b, 0, st, v.fine.offsets, (const float4*)v.fine.list,
                       wave_cell, wave_offs, totals, offs, counts, (const float4*)sorted, (int32_t*)nullptr, wp, (const float4*)nullptr);
}

void dsn_nn_cellmajor_begin(void* small, int32_t** counts, int32_t** outside, hipStream_t st) {
    char* q = (char*)small;
    *counts = (int32_t*)q;
    int32_t* totals = (int32_t*)(q + 3 * dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1)));
    *outside = totals + 2;
    (void)hipMemsetAsync(*counts, 0, 4 * (size_t)(DSN_NN_FINE_MAXCELL + 1), st);
    (void)hipMemsetAsync(totals, 0, 256, st);
}

size_t dsn_nn_sort_scratch_size(int64_t N) {
    // counts, offs, wave_offs: one int per fine cell (+1); totals; wave_cell: one int per wave
    return 3 * dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1)) + 256 + dsn_align256(4 * (size_t)(N / 64 + DSN_NN_FINE_MAXCELL + 1));
}

// nn [N] <- exact nearest-centroid index for every sample inside the fine grid, -1 elsewhere.
// cell_of: N ints of scratch; sorted: N float4 of scratch; small: dsn_nn_sort_scratch_size(N) bytes
void dsn_launch_nn_cellmajor(const DsnNNView& v, const float* pts, const float* ray_o, const float* ray_d, const float* z_vals,
                             int64_t N, int S, int32_t* cell_of, void* sorted, int32_t* nn, void* small, hipStream_t st) {
    char* q = (char*)small;
    int32_t* counts = (int32_t*)q;     q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* offs = (int32_t*)q;       q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* wave_offs = (int32_t*)q;  q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* totals = (int32_t*)q;     q += 256;
    int32_t* wave_cell = (int32_t*)q;
    (void)hipMemsetAsync(counts, 0, 4 * (size_t)(DSN_NN_FINE_MAXCELL + 1), st);
    const dim3 gN((unsigned)((N + NNS_THREADS - 1) / NNS_THREADS)), b(NNS_THREADS);
    hipLaunchKernelGGL(k_nns_classify, gN, b, 0, st, v.fine.g, pts, ray_o, ray_d, z_vals, N, S, cell_of, nn, counts, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_nns_scan, dim3(1), dim3(1024), 0, st, v.fine.g, counts, offs, wave_offs, totals, 0, 0);
    hipLaunchKernelGGL(k_nns_expand, dim3(DSN_NN_FINE_MAXCELL / NNS_THREADS), b, 0, st, v.fine.g, wave_offs, totals, wave_cell, 0);
    hipLaunchKernelGGL(k_nns_scatter, gN, b, 0, st, cell_of, pts, ray_o, ray_d, z_vals, N, S, offs, counts, (float4*)sorted);
    const int64_t max_waves = N / NNS_PER + DSN_NN_FINE_MAXCELL + 1;
    hipLaunchKernelGGL(k_nns_search<false>, dim3((unsigned)((max_waves + 3) / 4)), b, 0, st, v.fine.offsets, (const float4*)v.fine.list,
                       wave_cell, wave_offs, totals, offs, counts, (const float4*)sorted, nn, NnsWarp{}, (const float4*)nullptr);
}

// The same cell-major search at the COARSE level, for the points that lie outside the fine grid (and inside the coarse one): a
// training batch evaluates transparent samples whose noise is positive, and their canonical points sit far from the body - 44 % of
// the rows k_normal sees, each lane gathering a ~1000-entry coarse list of its own (0.6 of the kernel's 0.9 ms).  Sorted by coarse
// cell, a wave shares one list through the scalar cache.  Same lists, same order, same strict '<' as dsn_nearest_lists.
//   live (optional): only points with live[i] != 0 take part.  nn[i] = -1 for every point that is not searched here.
__global__ void __launch_bounds__(NNS_THREADS) k_nns_classify_coarse(const DsnGrid* __restrict__ gf, const DsnGrid* __restrict__ gc,
                                                                     const float* __restrict__ pts, const uint8_t* __restrict__ live,
                                                                     int64_t N, int32_t* __restrict__ cell_of, int32_t* __restrict__ nn,
                                                                     int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * NNS_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1;
    if (i < N && (!live || live[i])) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        if (dsn_grid_cell(*gf, px, py, pz) < 0) c = dsn_grid_cell(*gc, px, py, pz);
    }
    if (i < N) { cell_of[i] = c; if (c < 0) nn[i] = -1; }
    const NnsRun r = nns_run(c, lane);
    if (r.head && c >= 0) atomicAdd(counts + c, r.len);
}
// The far search with the candidate lists cut into SEGMENTS (round 6).  Far from the body a coarse cell's list approaches all F
// centroids; one wave walking it alone (two dependent scalar loads per candidate) took ~0.2 ms, and the kernel lasted as long as
// its longest such wave whatever the chip had free (0.39 - 0.42 ms per 8192 x 64 training step, the largest non-matrix kernel).
// Here wave (chunk of <= 128 samples, segment of <= NNS_FAR_SEG candidates) finds the segment's nearest candidate per sample and the
// segments meet in a 64-bit atomicMin on (distance bits, face index): distances are non-negative floats, so the integer order IS
// the (distance, index) lexicographic order - the smallest distance and, among equal ones, the smallest index: what the serial
// ascending sweep with its strict '<' returns, from the same dsn_d2 values.
#define NNS_FAR_SEG 768
__global__ void __launch_bounds__(NNS_THREADS) k_nns_search_far(const int32_t* __restrict__ off_c, const int32_t* __restrict__ list_c,
                                                                const int32_t* __restrict__ wave_cell, const int32_t* __restrict__ wave_offs,
                                                                const int32_t* __restrict__ totals, const int32_t* __restrict__ offs,
                                                                const int32_t* __restrict__ counts, const float4* __restrict__ sorted,
                                                                const float4* __restrict__ cent, unsigned long long* __restrict__ keys, int seg) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (NNS_THREADS / 64) + (threadIdx.x >> 6));
    if (w >= totals[0]) return;
    const int c = __builtin_amdgcn_readfirstlane(wave_cell[w]);
    const int cnt_c = __builtin_amdgcn_readfirstlane(counts[c]);
    const int chunks = (cnt_c + NNS_PER - 1) / NNS_PER;
    const int lw = w - __builtin_amdgcn_readfirstlane(wave_offs[c]);
    const int sg = lw / chunks, ch = lw - sg * chunks;            // (segment-major: the chunks of one segment share its candidates in cache)
    const int slot = ch * NNS_PER + lane;
    const bool valid[2] = {slot < cnt_c, slot + 64 < cnt_c};
    float4 q[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int so = __builtin_amdgcn_readfirstlane(offs[c]);
    if (valid[0]) q[0] = sorted[so + slot];
    if (valid[1]) q[1] = sorted[so + slot + 64];
    const int o = __builtin_amdgcn_readfirstlane(off_c[c]);
    const int len = __builtin_amdgcn_readfirstlane(off_c[c + 1]) - o;
    const int k0 = sg * seg;
    const int n = (len > seg ? (k0 + seg < len ? seg : len - k0) : len);
    const int32_t* __restrict__ ids = list_c + o + (len > seg ? k0 : 0);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 px = {q[0].x, q[1].x}, py = {q[0].y, q[1].y}, pz = {q[0].z, q[1].z};
    float best[2] = {INFINITY, INFINITY};
    int bi[2] = {0x7fffffff, 0x7fffffff};
    auto step = [&](const float4 a) {      // exactly dsn_d2 per component of the pair
        const f32x2 cx = {a.x, a.x}, cy = {a.y, a.y}, cz = {a.z, a.z};
        const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
        f32x2 d = dx * dx;
        d = __builtin_elementwise_fma(dy, dy, d);
        d = __builtin_elementwise_fma(dz, dz, d);
        const int id = __float_as_int(a.w);
        if (d.x < best[0]) { best[0] = d.x; bi[0] = id; }
        if (d.y < best[1]) { best[1] = d.y; bi[1] = id; }
    };
    auto at = [&](int f) { float4 a = cent[f]; a.w = __int_as_float(f); return a; };
    constexpr int NB = 4;      // one batch ahead on the indices, one on the centroids (see k_nns_search)
    const int nb = n / NB;
    int f1[NB], f2[NB];
    float4 a0[NB], a1[NB];
    if (nb > 0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) f1[j] = ids[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) a0[j] = at(f1[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) f1[j] = ids[NB * (1 < nb ? 1 : 0) + j];
        for (int b = 0; b < nb; ++b) {
            if (b + 1 < nb) {
#pragma unroll
                for (int j = 0; j < NB; ++j) a1[j] = at(f1[j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) f2[j] = ids[NB * (b + 2 < nb ? b + 2 : 0) + j];
#pragma unroll
            for (int j = 0; j < NB; ++j) step(a0[j]);
#pragma unroll
            for (int j = 0; j < NB; ++j) { a0[j] = a1[j]; f1[j] = f2[j]; }
        }
    }
    for (int k = NB * nb; k < n; ++k) step(at(ids[k]));
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
        if (valid[h2] && best[h2] < INFINITY)
            atomicMin(keys + __float_as_int(q[h2].w), ((unsigned long long)__float_as_uint(best[h2]) << 32) | (unsigned)bi[h2]);
}
__global__ void __launch_bounds__(NNS_THREADS) k_nns_far_finish(const int32_t* __restrict__ cell_of, const unsigned long long* __restrict__ keys,
                                                                int64_t N, int32_t* __restrict__ nn) {
    const int64_t i = (int64_t)blockIdx.x * NNS_THREADS + threadIdx.x;
    if (i < N && cell_of[i] >= 0) nn[i] = (int32_t)(keys[i] & 0xffffffffull);
}
void dsn_launch_nn_cellmajor_coarse(const DsnNNView& v, const float4* cent, const float* pts, const uint8_t* live, int64_t N,
                                    int32_t* cell_of, void* sorted, int32_t* nn, void* small, hipStream_t st, void* keys8N, int F,
                                    int32_t* wave_scratch, int64_t wave_scratch_ints) {
    // keys8N (optional, 8 N bytes of scratch) + wave_scratch (ints: the wave -> cell map of the segmented launch + scatter cursors): the
    // segmented search (k_nns_search_far).  NULL, too little scratch or DSN_FAR_SEGMENTS=0: one wave per list, as rounds 3-5
    char* q = (char*)small;
    int32_t* counts = (int32_t*)q;     q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* offs = (int32_t*)q;       q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* wave_offs = (int32_t*)q;  q += dsn_align256(4 * (size_t)(DSN_NN_FINE_MAXCELL + 1));
    int32_t* totals = (int32_t*)q;     q += 256;
    int32_t* wave_cell = (int32_t*)q;
    static_assert(DSN_NN_COARSE_MAXCELL <= DSN_NN_FINE_MAXCELL, "the sort scratch is sized for the fine level");
    (void)hipMemsetAsync(counts, 0, 4 * (size_t)(DSN_NN_COARSE_MAXCELL + 1), st);
    const dim3 gN((unsigned)((N + NNS_THREADS - 1) / NNS_THREADS)), b(NNS_THREADS);
    hipLaunchKernelGGL(k_nns_classify_coarse, gN, b, 0, st, v.fine.g, v.coarse.g, pts, live, N, cell_of, nn, counts);
    static const bool seg_off = [] { const char* e = getenv("DSN_FAR_SEGMENTS"); return e && e[0] == '0'; }();
    static const int seg = [] { const char* e = getenv("DSN_FAR_SEG"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 16384 ? v & ~3 : NNS_FAR_SEG; }();
    const int64_t nseg_max = ((int64_t)F + seg - 1) / seg;
    const int64_t mw = (N / NNS_PER + DSN_NN_COARSE_MAXCELL + 1) * (nseg_max > 0 ? nseg_max : 1);      // most waves the scan can ask for
    if (keys8N && wave_scratch && !seg_off && mw + DSN_NN_COARSE_MAXCELL + 1 <= wave_scratch_ints) {
        wave_cell = wave_scratch;
        (void)hipMemsetAsync(keys8N, 0xff, 8 * (size_t)N, st);
        hipLaunchKernelGGL(k_nns_scan_mb, dim3(DSN_NN_SCAN_BLOCKS(DSN_NN_COARSE_MAXCELL)), dim3(1024), 0, st, v.coarse.g, (const int32_t*)counts, offs,
                           wave_offs, totals, 0, (const int32_t*)v.coarse.offsets, seg);
        hipLaunchKernelGGL(k_nns_expand, dim3(DSN_NN_COARSE_MAXCELL / NNS_THREADS), b, 0, st, v.coarse.g, wave_offs, totals, wave_cell, 0);
        // (counts kept by the scan: the atomic scatter needs cursors of its own - the wave offsets' neighbour array is free: ranks via a
        //  cleared copy would cost a launch; k_nns_scatter runs on a zeroed cursor array placed behind wave_cell)
        int32_t* cursor = wave_cell + mw;
        (void)hipMemsetAsync(cursor, 0, 4 * (size_t)(DSN_NN_COARSE_MAXCELL + 1), st);
        hipLaunchKernelGGL(k_nns_scatter, gN, b, 0, st, cell_of, pts, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, N, 1,
                           offs, cursor, (float4*)sorted);
        hipLaunchKernelGGL(k_nns_search_far, dim3((unsigned)((mw + 3) / 4)), b, 0, st, v.coarse.offsets, (const int32_t*)v.coarse.list, wave_cell,
                           wave_offs, totals, offs, counts, (const float4*)sorted, cent, (unsigned long long*)keys8N, seg);
        hipLaunchKernelGGL(k_nns_far_finish, gN, b, 0, st, cell_of, (const unsigned long long*)keys8N, N, nn);
        return;
    }
    hipLaunchKernelGGL(k_nns_scan, dim3(1), dim3(1024), 0, st, v.coarse.g, counts, offs, wave_offs, totals, 0, 0);
    hipLaunchKernelGGL(k_nns_expand, dim3(DSN_NN_COARSE_MAXCELL / NNS_THREADS), b, 0, st, v.coarse.g, wave_offs, totals, wave_cell, 0);
    hipLaunchKernelGGL(k_nns_scatter, gN, b, 0, st, cell_of, pts, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, N, 1,
                       offs, counts, (float4*)sorted);
    const int64_t max_waves = N / NNS_PER + DSN_NN_COARSE_MAXCELL + 1;
    hipLaunchKernelGGL(k_nns_search<false>, dim3((unsigned)((max_waves + 3) / 4)), b, 0, st, v.coarse.offsets, (const float4*)v.coarse.list,
                       wave_cell, wave_offs, totals, offs, counts, (const float4*)sorted, nn, NnsWarp{}, cent);
}
