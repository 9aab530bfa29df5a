// dsn_nn.h - exact nearest-centroid search structure ("cell candidate lists").
//
// The reference finds, for every sample point, the face whose centroid is nearest
// (utils/render_utils.py:84-99 -> pytorch3d knn_points K=1): 13 776 distance evaluations per point,
// twice per point.  This structure returns the SAME index (bit-identical argmin, first index on ties)
// from a short list:
//
//   For a box B (a grid cell, grown by a guard band), every point p in B satisfies
//       dist(p, c_nn(p)) <= dist(p, c_g) <= dmax(B, c_g)        for all faces g,
//   hence  dmin(B, c_nn(p)) <= U(B) := min_g dmax(B, c_g).
//   So the list  L(B) = { f : dmin(B, c_f)^2 <= U(B)^2 (1 + 1e-5) }  contains the nearest centroid of
//   every point of B (the 1e-5 slack dominates all float32 rounding in dmin/dmax/d2; the guard band
//   dominates the rounding of the point->cell assignment).  L(B) is stored in ascending face order, and
//   the query scans it with the same fma chain and strict '<' as the brute-force search, so ties resolve
//   to the lowest index exactly as pytorch3d does.
//
// Two levels per mesh: a fine grid (~3F cells) over the centroid AABB + pad1 and a coarse grid over
// AABB + pad2 for far-away points; points outside both fall back to an exhaustive scan.  Lists are
// rebuilt on the device for every posed frame (dsn_set_frame) and once for the canonical mesh.
#pragma once
// (included from dsn_common.h after dsn_align256 is defined)

struct DsnGrid {            // device-resident descriptor (64 B)
    float lo[3];
    float cell;
    float inv_cell;
    int nx, ny, nz;
    int ncell;
    int ok;                 // 1 when the lists fit the capacity and the level is usable
    int total;              // number of list entries
    int cap;                // capacity of the list array (entries)
    int maxcell;
    int lazy;               // 0: lists of EVERY cell (ok says whether they fit); 1: grid geometry only, lists still to be built by the frame that
                            // uses them (dsn_set_frame_ex DSN_FRAME_LAZY_LISTS); 2: lists built for the cells that frame's samples visit -
                            // valid for that frame's fused search only (ok stays 0: every other query takes the next level / the sweep)
    int pad_[2];
};

#define DSN_GRID_GUARD 1e-4f     // metres added on every side of a cell before bounding
#define DSN_NN_FINE_MAXCELL 65536
#define DSN_NN_COARSE_MAXCELL 16384

// List capacities (entries per level).  Measured needs at F = 13 776 (dsn_debug_nn_stats; tests/test_gpu_round4.py): the uniform
// lattice body 970 F (fine, posed) / 820 F (coarse); the SMPL-like body of synth.make_body(nonuniform=True) - half of the vertices in
// dense caps at head / hands / feet - 1260 F / 1060 F: round 3's 1600 F / 1000 F left the latter's COARSE level switched off (every
// far query on the exhaustive sweep, silently).  2000 F / 2000 F now; a level that still does not fit is reported by the host mirror
// (Scene.nn_overflow, a warning) instead of failing silently.
__host__ __device__ inline int dsn_nn_fine_cap(int F) { long long c = 2000LL * F; return (int)(c < (1 << 20) ? (1 << 20) : (c > 0x7fffff00LL ? 0x7fffff00LL : c)); }
__host__ __device__ inline int dsn_nn_coarse_cap(int F) { long long c = 2000LL * F; return (int)(c < (1 << 20) ? (1 << 20) : (c > 0x7fffff00LL ? 0x7fffff00LL : c)); }

#define DSN_SUPER 4                 // build acceleration: super-cells of 4 x 4 x 4 cells ...
#define DSN_SUPER_CAP 4096          // ... each with a candidate superset of at most this many faces
struct DsnGridView {        // one level
    DsnGrid* g;
    int32_t* offsets;       // [maxcell + 1]
    float* u2;              // [maxcell] scratch: U(B)^2
    void* list;             // [cap] entries: float4 {x,y,z,index bits} (fine level) or int32 index (coarse level)
    int32_t* super_cnt;     // [maxcell] (fine level only, else NULL): size of each super-cell's superset (> CAP: unusable)
    float4* super_list;     // [maxsuper][DSN_SUPER_CAP] supersets, ascending face order
    unsigned long long* member;   // [maxcell][DSN_SUPER_CAP / 64] (fine level only): which entries of its super-cell's superset a cell's list
                                  // holds, one bit per entry - k_grid_count leaves them for k_grid_fill (no third sweep of the superset)
};
__host__ __device__ inline int dsn_grid_maxsuper(int maxcell) { return maxcell / 32; }   // grids with more super-cells (very thin ones) build unaccelerated
struct DsnNNView { DsnGridView fine, coarse; };

__host__ __device__ inline size_t dsn_grid_bytes(int maxcell, int cap, int entry_bytes) {
    size_t b = dsn_align256(sizeof(DsnGrid)) + dsn_align256(sizeof(int32_t) * ((size_t)maxcell + 1)) +
               dsn_align256(sizeof(float) * (size_t)maxcell) + dsn_align256((size_t)entry_bytes * (size_t)cap);
    if (entry_bytes == 16)
        b += dsn_align256(sizeof(int32_t) * (size_t)maxcell) +
             dsn_align256(sizeof(float) * 4 * (size_t)dsn_grid_maxsuper(maxcell) * DSN_SUPER_CAP) +
             dsn_align256(8 * (size_t)(DSN_SUPER_CAP / 64) * (size_t)maxcell);
    return b;
}
__host__ __device__ inline DsnGridView dsn_grid_view(char*& p, int maxcell, int cap, int entry_bytes) {
    DsnGridView v;
    v.g = (DsnGrid*)p;          p += dsn_align256(sizeof(DsnGrid));
    v.offsets = (int32_t*)p;    p += dsn_align256(sizeof(int32_t) * ((size_t)maxcell + 1));
    v.u2 = (float*)p;           p += dsn_align256(sizeof(float) * (size_t)maxcell);
    v.list = (void*)p;          p += dsn_align256((size_t)entry_bytes * (size_t)cap);
    v.super_cnt = nullptr;
    v.super_list = nullptr;
    v.member = nullptr;
    if (entry_bytes == 16) {
        v.super_cnt = (int32_t*)p;  p += dsn_align256(sizeof(int32_t) * (size_t)maxcell);
        v.super_list = (float4*)p;  p += dsn_align256(sizeof(float) * 4 * (size_t)dsn_grid_maxsuper(maxcell) * DSN_SUPER_CAP);
        v.member = (unsigned long long*)p;  p += dsn_align256(8 * (size_t)(DSN_SUPER_CAP / 64) * (size_t)maxcell);
    }
    return v;
}
__host__ __device__ inline size_t dsn_nn_bytes(int F) {
    return dsn_grid_bytes(DSN_NN_FINE_MAXCELL, dsn_nn_fine_cap(F), 16) +
           dsn_grid_bytes(DSN_NN_COARSE_MAXCELL, dsn_nn_coarse_cap(F), 4);
}
__host__ __device__ inline DsnNNView dsn_nn_view(char*& p, int F) {
    DsnNNView v;
    v.fine = dsn_grid_view(p, DSN_NN_FINE_MAXCELL, dsn_nn_fine_cap(F), 16);
    v.coarse = dsn_grid_view(p, DSN_NN_COARSE_MAXCELL, dsn_nn_coarse_cap(F), 4);
    return v;
}

#ifdef __HIPCC__
// squared distance exactly as the exhaustive search computes it
__device__ __forceinline__ float dsn_d2(float px, float py, float pz, const float4 c) {
    float dx = px - c.x, dy = py - c.y, dz = pz - c.z;
    float d = dx * dx;
    d = fmaf(dy, dy, d);
    d = fmaf(dz, dz, d);
    return d;
}

// cell index of p in grid g by its geometry alone, or -1 when p is outside the grid
__device__ __forceinline__ int dsn_grid_cell_geom(const DsnGrid& g, float px, float py, float pz) {
    float fx = (px - g.lo[0]) * g.inv_cell, fy = (py - g.lo[1]) * g.inv_cell, fz = (pz - g.lo[2]) * g.inv_cell;
    if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f)) return -1;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    if (ix >= g.nx || iy >= g.ny || iz >= g.nz) return -1;
    return (ix * g.ny + iy) * g.nz + iz;
}
// cell index of p in grid g, or -1 when p is outside the grid / the level is unusable
__device__ __forceinline__ int dsn_grid_cell(const DsnGrid& g, float px, float py, float pz) {
    if (!g.ok) return -1;
    float fx = (px - g.lo[0]) * g.inv_cell, fy = (py - g.lo[1]) * g.inv_cell, fz = (pz - g.lo[2]) * g.inv_cell;
    if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f)) return -1;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    if (ix >= g.nx || iy >= g.ny || iz >= g.nz) return -1;
    return (ix * g.ny + iy) * g.nz + iz;
}

// exact nearest centroid through the two-level lists; exhaustive scan (global memory) beyond them.
// Fine lists carry the centroid inline (one 16-byte load per candidate, sequential addresses, lanes in the
// same cell share them) and are scanned four at a time in list order, which keeps first-index-wins.
// nearest centroid through the fine / coarse candidate lists; -1 when the point lies outside both grids
__device__ __forceinline__ int dsn_nearest_lists_try(const DsnGrid* __restrict__ gf, const int32_t* __restrict__ off_f,
                                                 const float4* __restrict__ list_f, const DsnGrid* __restrict__ gc,
                                                 const int32_t* __restrict__ off_c, const int32_t* __restrict__ list_c,
                                                 const float4* __restrict__ cent, int F, float px, float py, float pz) {
    float best = INFINITY;
    int bi = 0;
    int c = dsn_grid_cell(*gf, px, py, pz);
    if (c >= 0) {
        const int o = off_f[c], n = off_f[c + 1] - o;
        const float4* e = list_f + o;
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const float4 a = e[i], b = e[i + 1], cc = e[i + 2], d = e[i + 3];
            const float da = dsn_d2(px, py, pz, a), db = dsn_d2(px, py, pz, b);
            const float dc = dsn_d2(px, py, pz, cc), dd = dsn_d2(px, py, pz, d);
            if (da < best) { best = da; bi = __float_as_int(a.w); }
            if (db < best) { best = db; bi = __float_as_int(b.w); }
            if (dc < best) { best = dc; bi = __float_as_int(cc.w); }
            if (dd < best) { best = dd; bi = __float_as_int(d.w); }
        }
        for (; i < n; ++i) {
            const float4 a = e[i];
            const float da = dsn_d2(px, py, pz, a);
            if (da < best) { best = da; bi = __float_as_int(a.w); }
        }
        return bi;
    }
    c = dsn_grid_cell(*gc, px, py, pz);
    if (c >= 0) {
        const int o = off_c[c], n = off_c[c + 1] - o;
        const int32_t* lst = list_c + o;
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const int f0 = lst[i], f1 = lst[i + 1], f2 = lst[i + 2], f3 = lst[i + 3];
            const float d0 = dsn_d2(px, py, pz, cent[f0]), d1 = dsn_d2(px, py, pz, cent[f1]);
            const float d2 = dsn_d2(px, py, pz, cent[f2]), d3 = dsn_d2(px, py, pz, cent[f3]);
            if (d0 < best) { best = d0; bi = f0; }
            if (d1 < best) { best = d1; bi = f1; }
            if (d2 < best) { best = d2; bi = f2; }
            if (d3 < best) { best = d3; bi = f3; }
        }
        for (; i < n; ++i) {
            const int f = lst[i];
            const float d = dsn_d2(px, py, pz, cent[f]);
            if (d < best) { best = d; bi = f; }
        }
        return bi;
    }
    return -1;      // outside both grids
}

// ... and the full function: points outside both grids sweep all F centroids (ascending, strict '<': first index wins)
__device__ __forceinline__ int dsn_nearest_lists(const DsnGrid* __restrict__ gf, const int32_t* __restrict__ off_f,
                                                 const float4* __restrict__ list_f, const DsnGrid* __restrict__ gc,
                                                 const int32_t* __restrict__ off_c, const int32_t* __restrict__ list_c,
                                                 const float4* __restrict__ cent, int F, float px, float py, float pz) {
    const int r = dsn_nearest_lists_try(gf, off_f, list_f, gc, off_c, list_c, cent, F, px, py, pz);
    if (r >= 0) return r;
    float best = INFINITY;
    int bi = 0;
    for (int f = 0; f < F; ++f) {
        const float d = dsn_d2(px, py, pz, cent[f]);
        if (d < best) { best = d; bi = f; }
    }
    return bi;
}

// Runs of equal cell inside a wave: the 64 lanes of a wave are consecutive samples, i.e. (pieces of) rays, and a
// straight line visits a convex cell in ONE contiguous run - so "distinct cells of the wave" are found by comparing
// with the previous lane, and each run head issues one atomic for the whole run (no per-cell loop).  A cell that does
// come back in a later run of the same wave (ray boundary inside the wave) simply gets a second atomic.
struct NnsRun { bool head; int head_lane; int len; int rank; };
__device__ __forceinline__ NnsRun nns_run(int c, int lane) {
    const int prev = __shfl_up(c, 1);
    const bool head = lane == 0 || c != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);         // heads at or below this lane (lane 63: all)
    NnsRun r;
    r.head = head;
    r.head_lane = 63 - __clzll((long long)(lane == 63 ? heads : below));
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;      // next head after this lane
    r.len = next - r.head_lane;       // same for every lane of the run
    r.rank = lane - r.head_lane;
    return r;
}

// one sample's step of the cell-major classification, called by every lane of a wave (wave-wide ballots inside): its fine cell
// -> cell_of[i], the cell's counter bumped once per run, samples outside the fine grid counted in *outside (optional).  Shared by
// k_nns_classify (dsn_nn.hip) and the sampler's emit loop (dsn_geom.hip: the fused path classifies while it writes z).
// rank_of (optional, round 4): the atomic's return value is the run's offset inside its cell - kept per sample, it is the sample's place
// in the cell-major order, and the counting sort's scatter needs no second round of atomics (k_nns_scatter_ranked).
__device__ __forceinline__ int dsn_nns_classify_one(const DsnGrid* __restrict__ gf, int64_t i, bool valid, float px, float py, float pz,
                                                    int32_t* __restrict__ rank_of, int32_t* __restrict__ cell_of,
                                                    int32_t* __restrict__ counts, int32_t* __restrict__ outside) {
    const int lane = threadIdx.x & 63;
    int c = -1;
    if (valid) {
        // (a lazy level - lists still to be built for the cells THIS frame visits - classifies by its geometry)
        c = gf->lazy ? dsn_grid_cell_geom(*gf, px, py, pz) : dsn_grid_cell(*gf, px, py, pz);
        cell_of[i] = c;
    }
    const NnsRun r = nns_run(c, lane);
    int base = 0;
    if (r.head && c >= 0) base = atomicAdd(counts + c, r.len);
    if (rank_of) {
        base = __shfl(base, r.head_lane);
        if (valid && c >= 0) rank_of[i] = base + r.rank;
    }
    // samples outside the fine grid (none for rays clipped to the body's bounds): counted, the fused search + warp leaves them to
    // a second pass (k_warp on the samples with cell_of < 0)
    if (outside) {
        const unsigned long long om = __ballot(valid && c < 0);
        if (om && lane == 0) atomicAdd(outside, __popcll(om));
    }
    return c;
}

// fine level only: -1 when the point is outside the fine grid
__device__ __forceinline__ int dsn_nearest_fine_try(const DsnGrid* __restrict__ gf, const int32_t* __restrict__ off_f,
                                                    const float4* __restrict__ list_f, float px, float py, float pz) {
    const int c = dsn_grid_cell(*gf, px, py, pz);
    if (c < 0) return -1;
    float best = INFINITY;
    int bi = 0;
    const int o = off_f[c], n = off_f[c + 1] - o;
    const float4* e = list_f + o;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float4 a = e[i], b = e[i + 1], cc = e[i + 2], d = e[i + 3];
        const float da = dsn_d2(px, py, pz, a), db = dsn_d2(px, py, pz, b);
        const float dc = dsn_d2(px, py, pz, cc), dd = dsn_d2(px, py, pz, d);
        if (da < best) { best = da; bi = __float_as_int(a.w); }
        if (db < best) { best = db; bi = __float_as_int(b.w); }
        if (dc < best) { best = dc; bi = __float_as_int(cc.w); }
        if (dd < best) { best = dd; bi = __float_as_int(d.w); }
    }
    for (; i < n; ++i) {
        const float4 a = e[i];
        const float da = dsn_d2(px, py, pz, a);
        if (da < best) { best = da; bi = __float_as_int(a.w); }
    }
    return bi;
}

// wave-level (distance, index) minimum with the serial sweeps' tie rule: equal distances keep the smaller index
__device__ __forceinline__ int dsn_wave_argmin(float best, int bi) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float od = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    return bi == 0x7fffffff ? 0 : bi;
}

// one coarse-level candidate list (ascending face indices) scanned by the whole wave for ONE point (wave-uniform arguments)
__device__ __forceinline__ int dsn_nearest_idlist_wave(const int32_t* __restrict__ lst, int n, const float4* __restrict__ cent,
                                                       float px, float py, float pz) {
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) {
        const int f = lst[i];
        const float d = dsn_d2(px, py, pz, cent[f]);
        if (d < best) { best = d; bi = f; }
    }
    return dsn_wave_argmin(best, bi);
}

// The same sweep done by the whole wave for ONE point (wave-uniform px, py, pz): lane l looks at f = l, l + 64, ... in
// ascending order, then the wave keeps the smallest distance and, among equal distances, the smallest index - the
// index the serial sweep returns, from the same dsn_d2 values.  Must be called by all 64 lanes.
__device__ __forceinline__ int dsn_nearest_sweep_wave(const float4* __restrict__ cent, int F, float px, float py, float pz) {
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int f = (int)(threadIdx.x & 63); f < F; f += 64) {
        const float d = dsn_d2(px, py, pz, cent[f]);
        if (d < best) { best = d; bi = f; }
    }
    return dsn_wave_argmin(best, bi);
}
#endif
