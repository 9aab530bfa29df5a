// dsn_geom.hip - geometry kernels of the hot path for gfx950: body/frame setup, geometry-guided
// sampler, nearest-face warp, compositing.  HBM/VALU-bound integer+float work: coalesced loads,
// LDS-staged broadcast tiles, wave-level scans; no MFMA here (the networks live in dsn_field.hip).
#include "dsn_common.h"
#include "dsn_kernels.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// setup: per-face records + centroids  (utils/render_utils.py:94, utils/geo_utils.py:181-200)
// ---------------------------------------------------------------------------------------------
__global__ void k_face_setup(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F,
                             DsnFaceRec* __restrict__ recs, float4* __restrict__ cent) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    float v0[3], v1[3], v2[3];
    for (int c = 0; c < 3; ++c) { v0[c] = verts[3 * i0 + c]; v1[c] = verts[3 * i1 + c]; v2[c] = verts[3 * i2 + c]; }
    DsnFaceRec r;
    dsn_make_face(v0, v1, v2, r);
    recs[f] = r;
    // meshes.mean(dim=-2) on CPU torch: (v0 + v1 + v2) / 3
    float4 c4;
    c4.x = dsn_div(dsn_sum3(v0[0], v1[0], v2[0]), 3.0f);
    c4.y = dsn_div(dsn_sum3(v0[1], v1[1], v2[1]), 3.0f);
    c4.z = dsn_div(dsn_sum3(v0[2], v1[2], v2[2]), 3.0f);
    c4.w = __int_as_float(f);
    cent[f] = c4;
}

void dsn_launch_face_setup(const float* verts, const int32_t* faces, int F, DsnFaceRec* recs, float4* cent,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_face_setup, dim3((F + 255) / 256), dim3(256), 0, st, verts, faces, F, recs, cent);
}

// model/spacenet.py:314-331 batch_rod2quat + :199-205 pose_mlp + :125-129 embedding row, and the
// fold of the 24 per-frame-constant input columns of stage1.0 into its bias.
__global__ void __launch_bounds__(256) k_pose_setup(const float* __restrict__ packed, const float* __restrict__ poses,
                                                     int frame_idx, int zero_code, const float* __restrict__ light_shift,
                                                     const float* __restrict__ rot, const float* __restrict__ rot_center,
                                                     DsnFrameState* __restrict__ fs, const float* __restrict__ pose_feat16) {
    // pose_feat16 (optional): the 16 pose features given explicitly (SpaceNet.forward's pose_feats argument,
    // model/spacenet.py:93-131) instead of batch_rod2quat + pose_mlp on `poses` (DualSpaceNeRF.forward :223-236)
    __shared__ float q[92];
    __shared__ float h1[64], h2[64], feat[16], code[8];
    int t = threadIdx.x;
    if (t < 23 && !pose_feat16) {
        const float* r = poses + 3 * (t + 1);
        float a[3] = {r[0] + 1e-16f, r[1] + 1e-16f, r[2] + 1e-16f};
        float angle = dsn_norm3(a);
        float half = dsn_div(angle, 2.0f);
        float s = sinf(half), c = cosf(half);
        q[4 * t + 0] = dsn_div(r[0], angle) * s;
        q[4 * t + 1] = dsn_div(r[1], angle) * s;
        q[4 * t + 2] = dsn_div(r[2], angle) * s;
        q[4 * t + 3] = c - 1.0f;
    }
    if (t < 8) code[t] = zero_code ? 0.0f : packed[OFF_RAW_EMB + frame_idx * 8 + t];
    __syncthreads();
    if (t < 64 && !pose_feat16) {
        float acc = packed[OFF_RAW_PM0B + t];
        const float* w = packed + OFF_RAW_PM0W + t * 92;
        for (int k = 0; k < 92; ++k) acc += w[k] * q[k];
        h1[t] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    if (t < 64 && !pose_feat16) {
        float acc = packed[OFF_RAW_PM2B + t];
        const float* w = packed + OFF_RAW_PM2W + t * 64;
        for (int k = 0; k < 64; ++k) acc += w[k] * h1[k];
        h2[t] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    if (t < 16) {
        float acc;
        if (pose_feat16) acc = pose_feat16[t];
        else {
            acc = packed[OFF_RAW_PM4B + t];
            const float* w = packed + OFF_RAW_PM4W + t * 64;
            for (int k = 0; k < 64; ++k) acc += w[k] * h2[k];
        }
        feat[t] = acc;
        fs->pose_feat[t] = acc;
    }
    if (t < 8) fs->code[t] = code[t];
    __syncthreads();
    {   // bias0[o] = b[o] + sum_k W0[o][k] code[k] + sum_k W0[o][71+k] pose[k]
        float acc = packed[OFF_RAW_B0 + t];
        const float* w = packed + OFF_RAW_W0 + t * 87;
        for (int k = 0; k < 8; ++k) acc = fmaf(w[k], code[k], acc);
        for (int k = 0; k < 16; ++k) acc = fmaf(w[71 + k], feat[k], acc);
        fs->bias0[t] = acc;
    }
    if (t == 0) {
        fs->has_light = light_shift ? 1.0f : 0.0f;
        for (int c = 0; c < 3; ++c) fs->light_shift[c] = light_shift ? light_shift[c] : 0.0f;
        fs->has_rot = (rot && rot_center) ? 1.0f : 0.0f;
        for (int c = 0; c < 4; ++c) fs->rot[c] = (rot && rot_center) ? rot[c] : 0.0f;
        for (int c = 0; c < 2; ++c) fs->rot_center[c] = (rot && rot_center) ? rot_center[c] : 0.0f;
    }
}

void dsn_launch_pose_setup(const float* packed, const float* poses, int frame_idx, int zero_code,
                           const float* light_shift, const float* rot, const float* rot_center, DsnFrameState* fs,
                           hipStream_t st, const float* pose_feat16) {
    hipLaunchKernelGGL(k_pose_setup, dim3(1), dim3(256), 0, st, packed, poses, frame_idx, zero_code, light_shift, rot,
                       rot_center, fs, pose_feat16);
}

// ---------------------------------------------------------------------------------------------
// sampler: utils/pts_utils.py:18-58 + :3-16.  One thread per ray for the union-of-spheres
// interval (vertices broadcast from an LDS tile; (v - o0) and |v - o0|^2 are per-vertex constants
// because the reference uses the FIRST ray's origin for every ray), then the block writes
// z_vals / pts with lanes running over samples (coalesced).
// ---------------------------------------------------------------------------------------------
#define GG_TILE 512      // 8 KB of vertices + 4 KB of per-wave survivor lists: with 2048 the kernel held 51 KB of LDS and could not share a
                         // compute unit with k_field16<forward> (137 KB of 160) when two frames are in flight
#define GG_THREADS 256
// order-preserving int key of a float (for atomicMin / atomicMax on intervals found by different vertex slices)
__device__ __forceinline__ int gg_key(float f) {
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float gg_unkey(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// the vertex sweep of utils/pts_utils.py:27-47 over [v_begin, v_end): interval of ray r (thread) in units of |d| = 1
__device__ __forceinline__ void gg_sweep(const float* __restrict__ xyz, int v_begin, int v_end, const float* __restrict__ ray_o,
                                         const float* __restrict__ ray_d, int R, float& zmin, float& zmax, bool& any,
                                         float& nrm) {
    __shared__ float4 sv2[2][GG_TILE];      // two tiles: the next one is fetched while this one is swept (one barrier per tile)
    const float gamma2 = (float)(0.05 * 0.05);
    const int tid = threadIdx.x;
    const int r = blockIdx.x * GG_THREADS + tid;
    const float o0x = ray_o[0], o0y = ray_o[1], o0z = ray_o[2];
    float du[3] = {0.f, 0.f, 1.f};
    nrm = 1.f;
    if (r < R) {
        float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
        nrm = dsn_norm3(d);
        du[0] = dsn_div(d[0], nrm); du[1] = dsn_div(d[1], nrm); du[2] = dsn_div(d[2], nrm);
    }
    zmin = 99999.f; zmax = -99999.f;
    any = false;
    // Per-wave conservative cull: the 64 rays of a wave (consecutive pixels) form a narrow bundle around the axis a with
    // half-angle theta = max angle(a, du_r).  A vertex w (relative to the common origin) at angle phi from the axis is at
    // distance >= |w| sin(phi' - theta), phi' = min(phi, pi - phi), from every line of the bundle; when that exceeds
    // gamma (+10 % and 0.1 mm of slack, far above the rounding of this test) it cannot satisfy rho^2 < gamma^2 for any
    // ray of the wave and is left out.  The survivors go through the reference's exact test unchanged, and min / max
    // do not depend on the order, so near / far are bit-identical to the full sweep.
    __shared__ unsigned short s_keep[GG_THREADS / 64][GG_TILE];
    const int lane = tid & 63, wave = tid >> 6;
    float ax, ay, az, cos_t, sin_t;
    {
        const bool ok = r < R;
        const int src = __ffsll((long long)__ballot(ok)) - 1;       // first valid lane (or -1)
        float ux = du[0], uy = du[1], uz = du[2];
        if (!ok && src >= 0) { ux = __shfl(du[0], src); uy = __shfl(du[1], src); uz = __shfl(du[2], src); }
        else if (!ok) { ux = 0.f; uy = 0.f; uz = 1.f; }
        float sx = ux, sy = uy, sz = uz;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
        const float inv = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
        ax = sx * inv; ay = sy * inv; az = sz * inv;
        float c = ax * ux + ay * uy + az * uz;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c = fminf(c, __shfl_xor(c, o));
        cos_t = fminf(c, 1.0f);
        sin_t = sqrtf(fmaxf(0.f, 1.0f - cos_t * cos_t));
    }
    const float keep_r = 0.05f * 1.10f + 1e-4f;
    static_assert(GG_TILE == 2 * GG_THREADS, "two vertices per thread and tile");
    auto tile_fetch = [&](int base, float (&c)[2][3]) {           // (requests only: the values are used after the current tile's sweep)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = base + tid + u * GG_THREADS;
            const int jj = j < v_end ? j : v_end - 1;
            c[u][0] = xyz[3 * jj]; c[u][1] = xyz[3 * jj + 1]; c[u][2] = xyz[3 * jj + 2];
        }
    };
    auto tile_store = [&](float4* dst, const float (&c)[2][3]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float dx = c[u][0] - o0x, dy = c[u][1] - o0y, dz = c[u][2] - o0z;
            dst[tid + u * GG_THREADS] = make_float4(dx, dy, dz, dsn_sum3(dx * dx, dy * dy, dz * dz));
        }
    };
    if (v_begin < v_end) {
        float c0[2][3];
        tile_fetch(v_begin, c0);
        tile_store(sv2[0], c0);
    }
    __syncthreads();
    int buf = 0;
    for (int base = v_begin; base < v_end; base += GG_TILE, buf ^= 1) {
        int n = min(GG_TILE, v_end - base);
        const float4* __restrict__ sv = sv2[buf];
        const bool more = base + GG_TILE < v_end;                 // (block-uniform)
        float cn[2][3];
        if (more) tile_fetch(base + GG_TILE, cn);
        int kept = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            bool keep = false;
            if (j < n) {
                const float4 v = sv[j];
                const float wn = sqrtf(v.w);
                const float ca = (ax * v.x + ay * v.y + az * v.z) / fmaxf(wn, 1e-20f);
                const float cc = fminf(fabsf(ca), 1.0f);
                const float ss = sqrtf(fmaxf(0.f, 1.0f - cc * cc));
                const float lb = wn * fmaxf(0.f, ss * cos_t - cc * sin_t);
                keep = !(lb > keep_r);
            }
            const unsigned long long m = __ballot(keep);
            if (keep) s_keep[wave][kept + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
            kept += __popcll(m);
        }
        // (same-wave LDS writes are visible to the wave's later reads; no barrier needed for a wave-private list)
#pragma unroll 4
        for (int k = 0; k < kept; ++k) {
            float4 v = sv[s_keep[wave][k]];
            float z0 = dsn_sum3(v.x * du[0], v.y * du[1], v.z * du[2]);
            float tmp = v.w - z0 * z0;
            if (tmp < gamma2) {
                float dz = sqrtf(gamma2 - tmp);
                zmin = fminf(zmin, z0 - dz);
                zmax = fmaxf(zmax, z0 + dz);
                any = true;
            }
        }
        if (more) tile_store(sv2[buf ^ 1], cn);                   // (the other buffer was last read before the previous barrier)
        __syncthreads();
    }
}

// z_vals (and pts) of the block's rays from their [near, far] in s_near / s_far (utils/pts_utils.py:3-16, 55-58)
// cls (fused path): the first step of the cell-major nearest-face search rides along - the sample's fine cell goes to cls.cell_of and
// the cell's counter is bumped once per run of equal cells in the wave (dsn_nn.hip, k_nns_classify: the same code on the z just
// computed instead of a second pass that reads it back)
struct GgClassify { const DsnGrid* gf; int32_t* cell_of; int32_t* counts; int32_t* outside; int32_t* rank_of; };
__device__ __forceinline__ void gg_emit(const float* s_near, const float* s_far, const float* __restrict__ ray_o,
                                        const float* __restrict__ ray_d, int R, int S, const float* __restrict__ t_vals,
                                        const float* __restrict__ jitter, float* __restrict__ z_vals, float* __restrict__ pts,
                                        const GgClassify cls = GgClassify{nullptr, nullptr, nullptr, nullptr, nullptr},
                                        int part = 0, int nparts = 1) {
    // part / nparts (k_sample_gg_emit): this workgroup takes every nparts-th stripe of GG_THREADS samples of the block's rays
    const int tid = threadIdx.x;
    const int rays_here = min(GG_THREADS, R - blockIdx.x * GG_THREADS);
    const int total = rays_here * S;
    const int total_up = cls.gf ? (total + 63) & ~63 : total;      // (classification uses wave-wide ballots: whole waves iterate)
    for (int e = part * GG_THREADS + tid; e < total_up; e += GG_THREADS * nparts) {
        // ONE convergent call of the classification per iteration: its run detection uses wave-wide shuffles / ballots, so
        // the tail lanes of a partial wave (e >= total) go through the same call site with valid = false instead of a call of their own
        // under a partial exec mask
        const bool valid = e < total;
        const int ec = valid ? e : total - 1;
        int lr = ec / S, i = ec - lr * S;
        float nn = s_near[lr], ff = s_far[lr];
        float ti = t_vals[i];
        float z = nn * (1.0f - ti) + ff * ti;
        int64_t g = (int64_t)(blockIdx.x * GG_THREADS + lr) * S + i;
        if (jitter) {
            float lower = z, upper = z;
            if (i > 0) { float tp = t_vals[i - 1]; lower = 0.5f * (z + (nn * (1.0f - tp) + ff * tp)); }
            if (i < S - 1) { float tn = t_vals[i + 1]; upper = 0.5f * ((nn * (1.0f - tn) + ff * tn) + z); }
            z = lower + (upper - lower) * jitter[g];
        }
        if (valid) z_vals[g] = z;
        if (pts || cls.gf) {
            int rr = blockIdx.x * GG_THREADS + lr;
            const float px = ray_o[3 * rr + 0] + ray_d[3 * rr + 0] * z;      // (the expression of k_warp / nns_point: same point, bit for bit)
            const float py = ray_o[3 * rr + 1] + ray_d[3 * rr + 1] * z;
            const float pz = ray_o[3 * rr + 2] + ray_d[3 * rr + 2] * z;
            if (pts && valid) { pts[3 * g + 0] = px; pts[3 * g + 1] = py; pts[3 * g + 2] = pz; }
            if (cls.gf) dsn_nns_classify_one(cls.gf, valid ? g : (int64_t)-1, valid, px, py, pz, cls.rank_of, cls.cell_of, cls.counts, cls.outside);
        }
    }
}

__global__ void __launch_bounds__(GG_THREADS) k_sample_gg(const float* __restrict__ xyz, int V,
                                                           const float* __restrict__ ray_o,
                                                           const float* __restrict__ ray_d, float* __restrict__ near,
                                                           float* __restrict__ far, int R, int S,
                                                           const float* __restrict__ t_vals,
                                                           const float* __restrict__ jitter, float* __restrict__ z_vals,
                                                           float* __restrict__ pts, GgClassify cls) {
    __shared__ float s_near[GG_THREADS], s_far[GG_THREADS];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * GG_THREADS + tid;
    float zmin, zmax, nrm;
    bool any;
    gg_sweep(xyz, 0, V, ray_o, ray_d, R, zmin, zmax, any, nrm);
    float n_ = 0.f, f_ = 0.f;
    if (r < R) {
        zmin = dsn_div(zmin, nrm);
        zmax = dsn_div(zmax, nrm);
        n_ = near[r]; f_ = far[r];
        if (any && zmin < zmax) { n_ = zmin; f_ = zmax; near[r] = n_; far[r] = f_; }
    }
    s_near[tid] = n_; s_far[tid] = f_;
    __syncthreads();
    gg_emit(s_near, s_far, ray_o, ray_d, R, S, t_vals, jitter, z_vals, pts, cls);
}

// Few rays (a training batch, a 3072-ray chunk): the sweep of one block of rays is split over blockIdx.y vertex slices so the
// launch fills the chip; the slices meet in two int keys per ray kept in the ray's first two z_vals slots (min / max do
// not depend on the order: bit-identical to the single sweep), k_sample_gg_bounds turns them into near / far and k_sample_gg_emit
// those into z_vals.
__global__ void __launch_bounds__(GG_THREADS) k_sample_gg_init(int R, int S, float* __restrict__ z_vals) {
    const int r = blockIdx.x * GG_THREADS + threadIdx.x;
    if (r < R) {
        reinterpret_cast<int*>(z_vals)[(int64_t)r * S] = 0x7fffffff;
        reinterpret_cast<int*>(z_vals)[(int64_t)r * S + 1] = (int)0x80000000;
    }
}

__global__ void __launch_bounds__(GG_THREADS) k_sample_gg_slice(const float* __restrict__ xyz, int V, int slice,
                                                                 const float* __restrict__ ray_o,
                                                                 const float* __restrict__ ray_d, int R, int S,
                                                                 float* __restrict__ z_vals) {
    const int r = blockIdx.x * GG_THREADS + threadIdx.x;
    const int v0 = blockIdx.y * slice, v1 = min(V, v0 + slice);
    float zmin, zmax, nrm;
    bool any;
    gg_sweep(xyz, v0, v1, ray_o, ray_d, R, zmin, zmax, any, nrm);
    if (r < R && any) {
        atomicMin(reinterpret_cast<int*>(z_vals) + (int64_t)r * S, gg_key(zmin));
        atomicMax(reinterpret_cast<int*>(z_vals) + (int64_t)r * S + 1, gg_key(zmax));
    }
}

// (rounds 1-5: ONE kernel, k_sample_gg_finish, one workgroup per 256 rays for the bounds AND the emission of their 256 S samples - 32
//  workgroups for a training batch, 128 for an eighth of a frame: 0.117 ms of a mostly idle chip, each workgroup waiting for 64 rounds
//  of classification atomics.  Now the bounds in a kernel of their own - they replace the keys in near / far, the keys' slots are
//  z_vals about to be written - and the emission by blockIdx.y parts per block of rays: the same z from the same near / far.)
__global__ void __launch_bounds__(GG_THREADS) k_sample_gg_bounds(const float* __restrict__ ray_d, float* __restrict__ near,
                                                                  float* __restrict__ far, int R, int S, const float* __restrict__ z_vals) {
    const int r = blockIdx.x * GG_THREADS + threadIdx.x;
    if (r >= R) return;
    const int kmin = reinterpret_cast<const int*>(z_vals)[(int64_t)r * S];
    const int kmax = reinterpret_cast<const int*>(z_vals)[(int64_t)r * S + 1];
    const bool any = kmin != 0x7fffffff;
    const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float nrm = dsn_norm3(d);
    const float zmin = dsn_div(any ? gg_unkey(kmin) : 99999.f, nrm);
    const float zmax = dsn_div(any ? gg_unkey(kmax) : -99999.f, nrm);
    if (any && zmin < zmax) { near[r] = zmin; far[r] = zmax; }
}
__global__ void __launch_bounds__(GG_THREADS) k_sample_gg_emit(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                                const float* __restrict__ near, const float* __restrict__ far, int R, int S,
                                                                const float* __restrict__ t_vals,
                                                                const float* __restrict__ jitter, float* __restrict__ z_vals,
                                                                float* __restrict__ pts, GgClassify cls) {
    __shared__ float s_near[GG_THREADS], s_far[GG_THREADS];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * GG_THREADS + tid;
    s_near[tid] = r < R ? near[r] : 0.f;
    s_far[tid] = r < R ? far[r] : 0.f;
    __syncthreads();
    gg_emit(s_near, s_far, ray_o, ray_d, R, S, t_vals, jitter, z_vals, pts, cls, (int)blockIdx.y, (int)gridDim.y);
}

void dsn_launch_sample_gg(const float* xyz, int V, const float* ray_o, const float* ray_d, float* near, float* far,
                          int R, int S, const float* t_vals, const float* jitter, float* z_vals, float* pts,
                          hipStream_t st, const DsnGrid* cls_grid, int32_t* cls_cell_of, int32_t* cls_counts, int32_t* cls_outside,
                          int32_t* cls_rank_of) {
    const GgClassify cls = {cls_grid, cls_cell_of, cls_counts, cls_outside, cls_rank_of};
    const int blocks = (R + GG_THREADS - 1) / GG_THREADS;
    int slices = blocks > 0 ? 1024 / blocks : 1;           // aim at ~4 workgroups per CU
    if (slices > V / 256) slices = V / 256;                // at least 256 vertices per slice
    if (slices >= 2 && S >= 2) {
        const int slice = (((V + slices - 1) / slices) + 63) & ~63;
        slices = (V + slice - 1) / slice;
        hipLaunchKernelGGL(k_sample_gg_init, dim3(blocks), dim3(GG_THREADS), 0, st, R, S, z_vals);
        hipLaunchKernelGGL(k_sample_gg_slice, dim3(blocks, slices), dim3(GG_THREADS), 0, st, xyz, V, slice, ray_o, ray_d, R, S,
                           z_vals);
        hipLaunchKernelGGL(k_sample_gg_bounds, dim3(blocks), dim3(GG_THREADS), 0, st, ray_d, near, far, R, S, (const float*)z_vals);
        int parts = 2048 / blocks;                             // ~8 workgroups per CU for the emission
        if (parts > S) parts = S;                              // (a block of rays has S stripes of GG_THREADS samples)
        if (parts < 1) parts = 1;
        hipLaunchKernelGGL(k_sample_gg_emit, dim3(blocks, parts), dim3(GG_THREADS), 0, st, ray_o, ray_d, (const float*)near, (const float*)far, R, S,
                           t_vals, jitter, z_vals, pts, cls);
        return;
    }
    hipLaunchKernelGGL(k_sample_gg, dim3(blocks), dim3(GG_THREADS), 0, st, xyz, V, ray_o,
                       ray_d, near, far, R, S, t_vals, jitter, z_vals, pts, cls);
}

// ---------------------------------------------------------------------------------------------
// exact nearest centroid, brute force: centroids broadcast from LDS tiles, squared distance as
// the fma chain of pytorch3d's kernel, strict '<' in ascending index order (first index wins).
// ---------------------------------------------------------------------------------------------
#define NN_TILE 2048
__device__ __forceinline__ int dsn_nearest_bruteforce(const float4* __restrict__ cent, int F, float px, float py,
                                                      float pz, float4* s_tile) {
    float best = INFINITY;
    int bi = 0;
    for (int base = 0; base < F; base += NN_TILE) {
        int n = min(NN_TILE, F - base);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) s_tile[j] = cent[base + j];
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            float4 c = s_tile[j];
            float dx = px - c.x, dy = py - c.y, dz = pz - c.z;
            float d = dx * dx;
            d = fmaf(dy, dy, d);
            d = fmaf(dz, dz, d);
            if (d < best) { best = d; bi = base + j; }
        }
    }
    return bi;
}

// can_render.py:333-379 w2l_without_lbs.  One thread per sample point.
#define WARP_THREADS 256
struct DsnNNArgs {   // device pointers of one mesh's two-level lists
    const DsnGrid* gf; const int32_t* off_f; const float4* list_f;
    const DsnGrid* gc; const int32_t* off_c; const int32_t* list_c;
};
static DsnNNArgs dsn_nn_args(const DsnNNView& v) {
    DsnNNArgs a = {v.fine.g, v.fine.offsets, (const float4*)v.fine.list, v.coarse.g, v.coarse.offsets, (const int32_t*)v.coarse.list};
    return a;
}

template <bool EXHAUSTIVE>
__global__ void __launch_bounds__(WARP_THREADS) k_warp(DsnNNArgs nn, const float4* __restrict__ cent_world,
                                                        const DsnFaceRec* __restrict__ face_world,
                                                        const DsnFaceRec* __restrict__ face_canon, int F,
                                                        const float* __restrict__ pts, const float* __restrict__ ray_o,
                                                        const float* __restrict__ ray_d,
                                                        const float* __restrict__ z_vals, int64_t N, int S,
                                                        int32_t* __restrict__ face_idx, float* __restrict__ uv_out,
                                                        float* __restrict__ h_out, uint8_t* __restrict__ transparent,
                                                        float* __restrict__ x_c, float* __restrict__ ray_d_can,
                                                        int32_t* __restrict__ active_list,
                                                        int32_t* __restrict__ active_count,
                                                        const int32_t* __restrict__ nn_pre, int lazy_canon,
                                                        const int32_t* __restrict__ only_cell_of, const int32_t* __restrict__ only_count) {
    __shared__ float4 s_tile[EXHAUSTIVE ? NN_TILE : 1];
    const int64_t i = (int64_t)blockIdx.x * WARP_THREADS + threadIdx.x;
    // only_cell_of / only_count (fused path, behind dsn_launch_nn_cellmajor_warp): this launch takes the samples the cell-major pass
    // left alone - those outside the fine grid (cell_of < 0), *only_count of them; normally none: the launch is 65 k workgroups that look
    if (only_count && *only_count == 0) return;      // block-uniform
    const bool valid = i < N && (!only_cell_of || only_cell_of[i] < 0);
    float p[3] = {0.f, 0.f, 0.f};
    int64_t ray = 0;
    if (valid) {
        ray = i / S;
        if (pts) { p[0] = pts[3 * i]; p[1] = pts[3 * i + 1]; p[2] = pts[3 * i + 2]; }
        else {
            float z = z_vals[i];
            p[0] = ray_o[3 * ray + 0] + ray_d[3 * ray + 0] * z;
            p[1] = ray_o[3 * ray + 1] + ray_d[3 * ray + 1] * z;
            p[2] = ray_o[3 * ray + 2] + ray_d[3 * ray + 2] * z;
        }
    }
    int fi = 0;
    if (EXHAUSTIVE) fi = dsn_nearest_bruteforce(cent_world, F, p[0], p[1], p[2], s_tile);
    else if (valid) {
        const int pre = nn_pre ? nn_pre[i] : -1;     // cell-major search result (dsn_launch_nn_cellmajor), -1 = not covered
        fi = pre >= 0 ? pre : dsn_nearest_lists(nn.gf, nn.off_f, nn.list_f, nn.gc, nn.off_c, nn.list_c, cent_world, F, p[0], p[1], p[2]);
    }
    bool active = false;
    if (valid) {
        DsnFaceRec fw = dsn_load_face(face_world, fi);
        float u, v, h, xc[3] = {0.f, 0.f, 0.f};
        dsn_project(p, fw, u, v, h);
        bool tr = (u > 5.f) || (u < -4.f) || (v > 5.f) || (v < -4.f) || (fabsf(h) > 0.1f);
        // fused eval path (lazy_canon): the canonical point of a transparent sample is never read, so the gather of its
        // canonical face record is skipped (61 % of the samples of the benchmark frame); x_c is written as zeros
        const bool need_c = !(tr && lazy_canon);
        DsnFaceRec fc;
        if (need_c) {
            fc = dsn_load_face(face_canon, fi);
            dsn_map2face(u, v, h, fc, xc);
        }
        if (face_idx) face_idx[i] = fi;
        if (uv_out) { uv_out[2 * i] = u; uv_out[2 * i + 1] = v; }
        if (h_out) h_out[i] = h;
        if (transparent) transparent[i] = tr ? 1 : 0;
        if (x_c) { x_c[3 * i] = xc[0]; x_c[3 * i + 1] = xc[1]; x_c[3 * i + 2] = xc[2]; }
        if (ray_d_can && ray_d) {
            float p2[3] = {p[0] + ray_d[3 * ray + 0], p[1] + ray_d[3 * ray + 1], p[2] + ray_d[3 * ray + 2]};
            float u2, v2, h2, xe[3], df[3], o[3];
            dsn_project(p2, fw, u2, v2, h2);
            dsn_map2face(u2, v2, h2, fc, xe);
            for (int c = 0; c < 3; ++c) df[c] = xe[c] - xc[c];
            dsn_normalize3(df, o);
            ray_d_can[3 * i] = o[0]; ray_d_can[3 * i + 1] = o[1]; ray_d_can[3 * i + 2] = o[2];
        }
        active = !tr;
    }
    if (active_list) {
        // workgroup-aggregated append: ONE atomic per 256 samples (every wave of the grid hits the same counter, and
        // same-address atomics serialise in L2 - one per wave was the floor of this kernel)
        __shared__ int s_cnt[WARP_THREADS / 64];
        __shared__ int s_base;
        const unsigned long long m = __ballot(active);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int k = 0; k < WARP_THREADS / 64; ++k) tot += s_cnt[k];
            s_base = tot ? atomicAdd(active_count, tot) : 0;
        }
        __syncthreads();
        if (active) {
            int off = s_base + __popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < wave; ++k) off += s_cnt[k];
            active_list[off] = (int32_t)i;
        }
    }
}

void dsn_launch_warp(const DsnSceneView& s, const float* pts, const float* ray_o, const float* ray_d,
                     const float* z_vals, int64_t N, int S, int32_t* face_idx, float* uv, float* h,
                     uint8_t* transparent, float* x_c, float* ray_d_can, int32_t* active_list, int32_t* active_count,
                     bool exhaustive, hipStream_t st, const int32_t* nn_pre, bool lazy_canon, const int32_t* only_cell_of,
                     const int32_t* only_count) {
    int64_t blocks = (N + WARP_THREADS - 1) / WARP_THREADS;
    DsnNNArgs nn = dsn_nn_args(s.nn_world);
    if (exhaustive)
        hipLaunchKernelGGL(k_warp<true>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_world, s.face_world,
                           s.face_canon, s.F, pts, ray_o, ray_d, z_vals, N, S, face_idx, uv, h, transparent, x_c,
                           ray_d_can, active_list, active_count, nullptr, (int)(lazy_canon && !ray_d_can), (const int32_t*)nullptr,
                           (const int32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_warp<false>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_world, s.face_world,
                           s.face_canon, s.F, pts, ray_o, ray_d, z_vals, N, S, face_idx, uv, h, transparent, x_c,
                           ray_d_can, active_list, active_count, nn_pre, (int)(lazy_canon && !ray_d_can), only_cell_of, only_count);
}

// ---------------------------------------------------------------------------------------------
// Dormant alternate of the warp (no reference entry point calls it any more): nearest posed face ->
// blend weights of the point (utils/render_utils.py:112-164: "rigid_center" = mean of the three vertex rows,
// "rigid_interp" = softmax over the three vertex distances, as written) -> inverse linear-blend skinning
// (utils/blend_utils.py:72-81 ppts_to_pts: blend the 24 joint transforms, subtract the translation, apply the inverse
// rotation).  One thread per point; the nearest face comes from the same exact lists as k_warp.
// ---------------------------------------------------------------------------------------------
template <bool EXHAUSTIVE>
__global__ void __launch_bounds__(WARP_THREADS) k_lbs_warp(DsnNNArgs nn, const float4* __restrict__ cent_world,
                                                            const DsnFaceRec* __restrict__ face_world,
                                                            const int32_t* __restrict__ faces, const float* __restrict__ xyz, int F,
                                                            const float* __restrict__ pts, int64_t N,
                                                            const float* __restrict__ smpl_w, const float* __restrict__ A,
                                                            int bw_type, int32_t* __restrict__ face_idx,
                                                            float* __restrict__ weights, uint8_t* __restrict__ transparent,
                                                            float* __restrict__ pts_zero) {
    __shared__ float4 s_tile[EXHAUSTIVE ? NN_TILE : 1];
    __shared__ float s_A[24 * 12];
    for (int e = threadIdx.x; e < 24 * 12; e += WARP_THREADS) s_A[e] = A[16 * (e / 12) + (e % 12)];   // rows 0..2 of every 4x4
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * WARP_THREADS + threadIdx.x;
    const bool valid = i < N;
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) { p[0] = pts[3 * i]; p[1] = pts[3 * i + 1]; p[2] = pts[3 * i + 2]; }
    int fi = 0;
    if (EXHAUSTIVE) fi = dsn_nearest_bruteforce(cent_world, F, p[0], p[1], p[2], s_tile);
    else if (valid) fi = dsn_nearest_lists(nn.gf, nn.off_f, nn.list_f, nn.gc, nn.off_c, nn.list_c, cent_world, F, p[0], p[1], p[2]);
    if (!valid) return;
    const DsnFaceRec fw = dsn_load_face(face_world, fi);
    float u, v, h;
    dsn_project(p, fw, u, v, h);
    if (face_idx) face_idx[i] = fi;
    if (transparent) transparent[i] = ((u > 5.f) || (u < -4.f) || (v > 5.f) || (v < -4.f) || (fabsf(h) > 0.1f)) ? 1 : 0;
    const int v0 = faces[3 * fi], v1 = faces[3 * fi + 1], v2 = faces[3 * fi + 2];
    float w3[3] = {0.f, 0.f, 0.f};
    if (bw_type == 1) {
        float d[3];
        const int vid[3] = {v0, v1, v2};
        for (int k = 0; k < 3; ++k) {
            const float e[3] = {xyz[3 * vid[k]] - p[0], xyz[3 * vid[k] + 1] - p[1], xyz[3 * vid[k] + 2] - p[2]};
            d[k] = dsn_norm3(e);
        }
        const float m = fmaxf(fmaxf(d[0], d[1]), d[2]);
        float s = 0.f;
        for (int k = 0; k < 3; ++k) { w3[k] = expf(d[k] - m); s += w3[k]; }
        for (int k = 0; k < 3; ++k) w3[k] = dsn_div(w3[k], s);
    }
    float M[12];
    for (int e = 0; e < 12; ++e) M[e] = 0.f;
    for (int j = 0; j < 24; ++j) {
        const float a = smpl_w[24 * v0 + j], b = smpl_w[24 * v1 + j], c = smpl_w[24 * v2 + j];
        const float bw = bw_type == 1 ? (w3[0] * a + w3[1] * b) + w3[2] * c : dsn_div((a + b) + c, 3.0f);
        if (weights) weights[24 * i + j] = bw;
        for (int e = 0; e < 12; ++e) M[e] = fmaf(bw, s_A[12 * j + e], M[e]);
    }
    if (pts_zero) {
        const float a = M[0], b = M[1], c = M[2], d = M[4], e2 = M[5], f = M[6], g = M[8], hh = M[9], k2 = M[10];
        const float c00 = e2 * k2 - f * hh, c01 = d * k2 - f * g, c02 = d * hh - e2 * g;
        const float det = a * c00 - b * c01 + c * c02;
        const float q[3] = {p[0] - M[3], p[1] - M[7], p[2] - M[11]};
        const float inv[9] = {c00, c * hh - b * k2, b * f - c * e2, -c01, a * k2 - c * g, c * d - a * f, c02, b * g - a * hh, a * e2 - b * d};
        for (int r = 0; r < 3; ++r)
            pts_zero[3 * i + r] = dsn_div(inv[3 * r] * q[0] + inv[3 * r + 1] * q[1] + inv[3 * r + 2] * q[2], det);
    }
}

void dsn_launch_lbs_warp(const DsnSceneView& s, const float* pts, int64_t N, const float* smpl_w, const float* A, int bw_type,
                         int32_t* face_idx, float* weights, uint8_t* transparent, float* pts_zero, bool exhaustive, hipStream_t st) {
    int64_t blocks = (N + WARP_THREADS - 1) / WARP_THREADS;
    DsnNNArgs nn = dsn_nn_args(s.nn_world);
    if (exhaustive)
        hipLaunchKernelGGL(k_lbs_warp<true>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_world, s.face_world, s.faces,
                           s.xyz, s.F, pts, N, smpl_w, A, bw_type, face_idx, weights, transparent, pts_zero);
    else
        hipLaunchKernelGGL(k_lbs_warp<false>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_world, s.face_world, s.faces,
                           s.xyz, s.F, pts, N, smpl_w, A, bw_type, face_idx, weights, transparent, pts_zero);
}

// ---------------------------------------------------------------------------------------------
// normals: model/spacenet.py:278-298 normal_local2world.  One thread per (listed) point.
// ---------------------------------------------------------------------------------------------
template <bool EXHAUSTIVE>
__global__ void __launch_bounds__(WARP_THREADS) k_normal(DsnNNArgs nn, const float4* __restrict__ cent_canon,
                                                          const DsnFaceRec* __restrict__ face_world,
                                                          const DsnFaceRec* __restrict__ face_canon, int F,
                                                          const float* __restrict__ x_c, const float* grad,
                                                          int64_t N, const int32_t* __restrict__ active_list,
                                                          const int32_t* __restrict__ active_count,
                                                          int32_t* __restrict__ face_idx_canon,
                                                          float* n_w, const int32_t* __restrict__ nn_far) {
    // (grad and n_w may be the SAME array - the fused path's workspace keeps the normal where the gradient was: every thread reads its
    //  sample's gradient before it writes its sample's normal; hence no __restrict__ on the two)
    // nn_far (optional): the answer for points outside the fine grid, where the coarse-level cell-major search has found one
    // (dsn_launch_nn_cellmajor_coarse: training batches); -1 elsewhere
    __shared__ float4 s_tile[EXHAUSTIVE ? NN_TILE : 1];
    int64_t count = active_list ? (int64_t)(*active_count) : N;
    int64_t slot0 = (int64_t)blockIdx.x * WARP_THREADS;
    if (slot0 >= count) return;   // uniform per block
    int64_t slot = slot0 + threadIdx.x;
    bool valid = slot < count;
    int64_t i = valid ? (active_list ? (int64_t)active_list[slot] : slot) : 0;
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) { p[0] = x_c[3 * i]; p[1] = x_c[3 * i + 1]; p[2] = x_c[3 * i + 2]; }
    int fi = 0;
    if (EXHAUSTIVE) fi = dsn_nearest_bruteforce(cent_canon, F, p[0], p[1], p[2], s_tile);
    else {
        if (valid) {
            const int far = nn_far ? nn_far[i] : -1;
            fi = far >= 0 ? far : dsn_nearest_fine_try(nn.gf, nn.off_f, nn.list_f, p[0], p[1], p[2]);
        }
        // Canonical points outside the fine grid (dense training batches: transparent samples away from the body) scan a coarse
        // list (hundreds to thousands of gathered centroids) or all F centroids.  When only a few lanes of the wave are in that
        // position, the wave takes them one at a time and scans each list together (64 candidates per step, then an argmin with
        // the serial tie rule) instead of a few lanes looping alone while the rest wait; when most lanes are, every lane scans
        // its own list as before.  Same candidates, same dsn_d2 values, same index either way.
        const bool pending = valid && fi < 0;
        const unsigned long long pend = __ballot(pending);
        if (pend) {
            int my_n = 0;
            if (pending) {
                const int c = dsn_grid_cell(*nn.gc, p[0], p[1], p[2]);
                my_n = c >= 0 ? nn.off_c[c + 1] - nn.off_c[c] : F;
            }
            int coop = pending ? my_n / 64 + 25 : 0, alone = my_n;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { coop += __shfl_xor(coop, o); alone = max(alone, __shfl_xor(alone, o)); }
            if (coop < alone) {
                // scalar copies of the mask drive the loop (wave-uniform control flow around the cross-lane operations)
                const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pend);
                const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pend >> 32));
                for (int src = 0; src < 64; ++src) {
                    const unsigned bit = src < 32 ? (plo >> src) & 1u : (phi >> (src - 32)) & 1u;
                    if (!bit) continue;
                    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0]), src));
                    const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[1]), src));
                    const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[2]), src));
                    const int cc = dsn_grid_cell(*nn.gc, qx, qy, qz);
                    int r;
                    if (cc >= 0) {
                        const int o = nn.off_c[cc];
                        r = dsn_nearest_idlist_wave(nn.list_c + o, nn.off_c[cc + 1] - o, cent_canon, qx, qy, qz);
                    } else {
                        r = dsn_nearest_sweep_wave(cent_canon, F, qx, qy, qz);
                    }
                    if ((int)(threadIdx.x & 63) == src) fi = r;
                }
            } else if (pending) {
                fi = dsn_nearest_lists(nn.gf, nn.off_f, nn.list_f, nn.gc, nn.off_c, nn.list_c, cent_canon, F, p[0], p[1], p[2]);
            }
        }
    }
    if (!valid) return;
    DsnFaceRec fc = dsn_load_face(face_canon, fi);
    DsnFaceRec fw = dsn_load_face(face_world, fi);
    float u, v, h, s[3], e[3], pe[3], df[3], o[3];
    dsn_project(p, fc, u, v, h);
    dsn_map2face(u, v, h, fw, s);
    for (int c = 0; c < 3; ++c) pe[c] = p[c] + grad[3 * i + c];
    dsn_project(pe, fc, u, v, h);
    dsn_map2face(u, v, h, fw, e);
    for (int c = 0; c < 3; ++c) df[c] = e[c] - s[c];
    dsn_normalize3(df, o);
    if (face_idx_canon) face_idx_canon[i] = fi;
    n_w[3 * i] = o[0]; n_w[3 * i + 1] = o[1]; n_w[3 * i + 2] = o[2];
}

void dsn_launch_normal(const DsnSceneView& s, const float* x_c, const float* grad, int64_t N,
                       const int32_t* active_list, const int32_t* active_count, int32_t* face_idx_canon, float* n_w,
                       bool exhaustive, hipStream_t st, const int32_t* nn_far) {
    int64_t blocks = (N + WARP_THREADS - 1) / WARP_THREADS;
    DsnNNArgs nn = dsn_nn_args(s.nn_canon);
    if (exhaustive)
        hipLaunchKernelGGL(k_normal<true>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_canon, s.face_world,
                           s.face_canon, s.F, x_c, grad, N, active_list, active_count, face_idx_canon, n_w, (const int32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_normal<false>, dim3((unsigned)blocks), dim3(WARP_THREADS), 0, st, nn, s.cent_canon, s.face_world,
                           s.face_canon, s.F, x_c, grad, N, active_list, active_count, face_idx_canon, n_w, nn_far);
}

// ---------------------------------------------------------------------------------------------
// compositing: utils/nerf_net_utils.py:5-56 raw2outputs (+ can_render.py:115-120).
// One wave per ray, lanes run over samples (S = 64 = one wavefront in every shipped config; longer
// rays go in chunks of 64 with a carried transmittance).  The exclusive transmittance product
// T_i = prod_{j<i}(1 - alpha_j + 1e-10) is a wavefront-level inclusive multiplicative scan
// (6 shuffle steps) shifted by one lane; the five per-ray sums are wavefront butterfly reductions.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dsn_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// lazy_colour (eval mode with the transparent skip): a colour exists only where the density is positive (elsewhere the weight is
// exactly 0 and the reference's colour never reaches the pixel), so it is read only there - the colour array needs no clearing.
// colour == NULL: weights from the densities alone (the shading list of DSN_EARLY_STOP); rgb_map == NULL: no per-ray outputs.
// WEIGHTS_ONLY: the early-stop shading list's call - no colour, no per-ray outputs: the same scan, none of the five reductions
template <bool WEIGHTS_ONLY>
__global__ void __launch_bounds__(256) k_composite(const float* __restrict__ colour, const float* __restrict__ sigma,
                                                    const uint8_t* __restrict__ transparent,
                                                    const float* __restrict__ z_vals, const float* __restrict__ ray_d,
                                                    const float* __restrict__ noise, int R, int S,
                                                    float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                                    float* __restrict__ acc_map, float* __restrict__ weights,
                                                    float* __restrict__ depth_map, int lazy_colour,
                                                    int32_t* __restrict__ colour_max) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;   // wave-uniform
    float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float dn = dsn_norm3(d);
    float carry = 1.0f;   // transmittance entering this chunk
    float sr = 0.f, sg = 0.f, sb = 0.f, sdep = 0.f, sacc = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        const bool in = i < S;
        const int64_t g = (int64_t)r * S + i;
        float z = in ? z_vals[g] : 0.f;
        float zn = __shfl_down(z, 1);
        if (lane == 63 && i + 1 < S) zn = z_vals[g + 1];
        float dist = (i + 1 < S) ? (zn - z) : 1e10f;
        dist = dist * dn;
        float s = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        if (in) {
            s = sigma[g];
            if (transparent && transparent[g]) s = 0.f;
            if (noise) s = s + noise[g];
            s = s > 0.f ? s : 0.f;
            if (!WEIGHTS_ONLY && colour && (!lazy_colour || s > 0.f)) { cr = colour[3 * g]; cg = colour[3 * g + 1]; cb = colour[3 * g + 2]; }
        }
        const float alpha = in ? (1.0f - expf(-s * dist)) : 0.f;
        const float fac = in ? ((1.0f - alpha) + 1e-10f) : 1.0f;
        float incl = fac;   // inclusive product scan over the wavefront
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            float t = __shfl_up(incl, off);
            if (lane >= off) incl = incl * t;
        }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63);
        const float w = alpha * T;
        if (in && weights) weights[g] = w;
        if (WEIGHTS_ONLY) continue;
        sr += dsn_wave_sum(w * cr);
        sg += dsn_wave_sum(w * cg);
        sb += dsn_wave_sum(w * cb);
        sdep += dsn_wave_sum(w * z);
        sacc += dsn_wave_sum(w);
        if (colour_max) {      // largest |colour| the frame weighs (float bits of a value >= 0 order like ints): what DSN_EARLY_STOP's bound scales with
            float m = fmaxf(fabsf(cr), fmaxf(fabsf(cg), fabsf(cb)));
            m = m == m ? m : INFINITY;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (lane == 0 && __float_as_int(m) > __hip_atomic_load(colour_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(colour_max, __float_as_int(m));   // (a fresh copy: see k_composite16)
        }
    }
    if (lane == 0 && rgb_map) {
        rgb_map[3 * r] = sr; rgb_map[3 * r + 1] = sg; rgb_map[3 * r + 2] = sb;
        depth_map[r] = sdep;
        acc_map[r] = sacc;
        float q = dsn_div(sdep, sacc);          // NaN when acc == 0, like the reference
        float m = (1e-10f > q) ? 1e-10f : q;    // torch.max propagates NaN
        if (q != q) m = q;
        disp_map[r] = dsn_div(1.0f, m);
    }
}

// Round 5: SIXTEEN lanes per ray, CH = S / 16 consecutive samples per lane (S = 64: 4, S = 128: 8 - every configuration the reference
// ships).  The one-wave-per-ray form above spends its time in cross-lane steps - a 6-step product scan, five 6-step sums and a 6-step
// maximum per 64 samples, most of them LDS-crossbar permutes - for 9 bytes of input per sample (1 TB/s).  Here a lane multiplies its
// own CH factors in registers, the scan and the reductions run over the 16 lanes of a DPP row (4 row_shr steps each, plain VALU
// operand shifts), and the inputs are 16-byte loads: four rays per wave for 28 cross-lane steps.  Same per-sample alpha (same expf,
// same operands); the ORDER of the products of T and of the five sums differs from the wave form (both differ from the reference's
// cumprod / sum order by the same few ulps).
__device__ __forceinline__ float dsn_row_shr(float old, float v, int n) {      // value of lane (l - n) of the same 16-lane row, `old` where there is none
    switch (n) {
        case 1: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false));
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x112, 0xf, 0xf, false));
        case 4: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x114, 0xf, 0xf, false));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x118, 0xf, 0xf, false));
    }
}
__device__ __forceinline__ float dsn_row_sum_to_last(float v) {      // lane 15 of every row ends up with the row's sum
    v += dsn_row_shr(0.f, v, 1);
    v += dsn_row_shr(0.f, v, 2);
    v += dsn_row_shr(0.f, v, 4);
    v += dsn_row_shr(0.f, v, 8);
    return v;
}
template <bool WEIGHTS_ONLY, int CH>
__global__ void __launch_bounds__(256) k_composite16(const float* __restrict__ colour, const float* __restrict__ sigma,
                                                      const uint8_t* __restrict__ transparent,
                                                      const float* __restrict__ z_vals, const float* __restrict__ ray_d,
                                                      const float* __restrict__ noise, int R, float* __restrict__ rgb_map,
                                                      float* __restrict__ disp_map, float* __restrict__ acc_map,
                                                      float* __restrict__ weights, float* __restrict__ depth_map, int lazy_colour,
                                                      int32_t* __restrict__ colour_max) {
    constexpr int S = 16 * CH;
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int r_of_row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    if (r_of_row - (lane >> 4) >= R) return;              // wave-uniform
    // a row past the last ray stays in step (the wave-wide maximum below reads every row) on a copy of the last ray and stores nothing
    const bool ok = r_of_row < R;
    const int r = ok ? r_of_row : R - 1;
    const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float dn = dsn_norm3(d);
    const int64_t g0 = (int64_t)r * S + sub * CH;
    float z[CH + 1], sg[CH];
    uint8_t tr[CH];
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
        const float4 z4 = *reinterpret_cast<const float4*>(z_vals + g0 + 4 * q);
        const float4 s4 = *reinterpret_cast<const float4*>(sigma + g0 + 4 * q);
        z[4 * q] = z4.x; z[4 * q + 1] = z4.y; z[4 * q + 2] = z4.z; z[4 * q + 3] = z4.w;
        sg[4 * q] = s4.x; sg[4 * q + 1] = s4.y; sg[4 * q + 2] = s4.z; sg[4 * q + 3] = s4.w;
        uchar4 t4 = make_uchar4(0, 0, 0, 0);
        if (transparent) t4 = *reinterpret_cast<const uchar4*>(transparent + g0 + 4 * q);
        tr[4 * q] = t4.x; tr[4 * q + 1] = t4.y; tr[4 * q + 2] = t4.z; tr[4 * q + 3] = t4.w;
        if (noise) {
            const float4 n4 = *reinterpret_cast<const float4*>(noise + g0 + 4 * q);
            // (the order of the reference: transparent samples have their density forced to 0 first, the noise is added afterwards)
            sg[4 * q] = (tr[4 * q] ? 0.f : sg[4 * q]) + n4.x; sg[4 * q + 1] = (tr[4 * q + 1] ? 0.f : sg[4 * q + 1]) + n4.y;
            sg[4 * q + 2] = (tr[4 * q + 2] ? 0.f : sg[4 * q + 2]) + n4.z; sg[4 * q + 3] = (tr[4 * q + 3] ? 0.f : sg[4 * q + 3]) + n4.w;
        }
    }
    // the first depth of the next lane (row_shl:1: lane l reads lane l + 1 of its row; the ray's last sample has no successor)
    z[CH] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z[0]), 0x101, 0xf, 0xf, false));
    float alpha[CH], pre[CH];                // pre[j]: product of this lane's factors 0 .. j
    float run = 1.0f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        float sv = sg[j];
        if (!noise && tr[j]) sv = 0.f;
        sv = sv > 0.f ? sv : 0.f;
        sg[j] = sv;
        const float dist = ((sub * CH + j + 1 < S) ? (z[j + 1] - z[j]) : 1e10f) * dn;
        alpha[j] = 1.0f - expf(-sv * dist);
        run = run * ((1.0f - alpha[j]) + 1e-10f);
        pre[j] = run;
    }
    // exclusive product scan of the lanes' totals over the row
    float incl = run;
    incl = incl * dsn_row_shr(1.0f, incl, 1);
    incl = incl * dsn_row_shr(1.0f, incl, 2);
    incl = incl * dsn_row_shr(1.0f, incl, 4);
    incl = incl * dsn_row_shr(1.0f, incl, 8);
    const float before = dsn_row_shr(1.0f, incl, 1);      // transmittance entering this lane's first sample
    float w[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) w[j] = alpha[j] * (j == 0 ? before : before * pre[j - 1]);
    if (weights && ok) {
#pragma unroll
        for (int q = 0; q < CH / 4; ++q)
            *reinterpret_cast<float4*>(weights + g0 + 4 * q) = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
    if (WEIGHTS_ONLY) return;
    float sr = 0.f, sgn = 0.f, sb = 0.f, sdep = 0.f, sacc = 0.f, cm = 0.f;
    float col[3 * CH];
    if (colour && !lazy_colour) {            // every colour is wanted: four samples' colours are 48 contiguous bytes
#pragma unroll
        for (int q = 0; q < 3 * CH / 4; ++q) {
            const float4 c4 = *reinterpret_cast<const float4*>(colour + 3 * g0 + 4 * q);
            col[4 * q] = c4.x; col[4 * q + 1] = c4.y; col[4 * q + 2] = c4.z; col[4 * q + 3] = c4.w;
        }
    } else {
        // a lazily filled colour array holds rubbish where the density is not positive (about one sample in eight is): never let it
        // into a product, and do not pull its lines through the cache either
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            col[3 * j] = col[3 * j + 1] = col[3 * j + 2] = 0.f;
            if (colour && sg[j] > 0.f) { col[3 * j] = colour[3 * (g0 + j)]; col[3 * j + 1] = colour[3 * (g0 + j) + 1]; col[3 * j + 2] = colour[3 * (g0 + j) + 2]; }
        }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const float cr = col[3 * j], cg = col[3 * j + 1], cb = col[3 * j + 2];
        sr += w[j] * cr; sgn += w[j] * cg; sb += w[j] * cb;
        sdep += w[j] * z[j];
        sacc += w[j];
        float m = fmaxf(fabsf(cr), fmaxf(fabsf(cg), fabsf(cb)));
        m = m == m ? m : INFINITY;
        cm = fmaxf(cm, m);
    }
    sr = dsn_row_sum_to_last(sr); sgn = dsn_row_sum_to_last(sgn); sb = dsn_row_sum_to_last(sb);
    sdep = dsn_row_sum_to_last(sdep); sacc = dsn_row_sum_to_last(sacc);
    if (colour_max) {
        cm = fmaxf(cm, dsn_row_shr(0.f, cm, 1)); cm = fmaxf(cm, dsn_row_shr(0.f, cm, 2));
        cm = fmaxf(cm, dsn_row_shr(0.f, cm, 4)); cm = fmaxf(cm, dsn_row_shr(0.f, cm, 8));
        // one candidate per wave, checked against a FRESH copy of the running maximum (a plain load may be served by this CU's L1 for
        // the whole kernel, and then every row of the frame queues an atomic on the one address: that was most of this kernel's time)
        cm = fmaxf(cm, __shfl_xor(cm, 16));
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        if (lane == 63 && __float_as_int(cm) > __hip_atomic_load(colour_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(colour_max, __float_as_int(cm));
    }
    if (sub == 15 && rgb_map && ok) {
        rgb_map[3 * r] = sr; rgb_map[3 * r + 1] = sgn; rgb_map[3 * r + 2] = sb;
        depth_map[r] = sdep;
        acc_map[r] = sacc;
        float q = dsn_div(sdep, sacc);          // NaN when acc == 0, like the reference
        float m = (1e-10f > q) ? 1e-10f : q;    // torch.max propagates NaN
        if (q != q) m = q;
        disp_map[r] = dsn_div(1.0f, m);
    }
}

void dsn_launch_composite(const float* colour, const float* sigma, const uint8_t* transparent, const float* z_vals,
                          const float* ray_d, const float* noise, int R, int S, float* rgb_map, float* disp_map,
                          float* acc_map, float* weights, float* depth_map, hipStream_t st, bool lazy_colour, int32_t* colour_max) {
    // 16 lanes per ray where the ray length allows 16-byte loads per lane (S = 64 / 128) and the arrays are 16-byte aligned;
    // DSN_COMPOSITE=wave (A/B, tests) keeps the one-wave-per-ray form
    static const bool wave_form = [] { const char* e = getenv("DSN_COMPOSITE"); return e && e[0] == 'w'; }();
    const bool aligned = (((uintptr_t)z_vals | (uintptr_t)sigma | (uintptr_t)weights | (uintptr_t)noise | (uintptr_t)colour) & 15) == 0 &&
                         (((uintptr_t)transparent) & 3) == 0;
    if (!wave_form && aligned && (S == 64 || S == 128)) {
        const dim3 grid((unsigned)((R + 15) / 16)), block(256);
        const bool wo = !colour && !rgb_map;
        int32_t* cmx = colour ? colour_max : nullptr;
#define DSN_C16(W, C) hipLaunchKernelGGL((k_composite16<W, C>), grid, block, 0, st, colour, sigma, transparent, z_vals, ray_d, noise, R, rgb_map, \
                                         disp_map, acc_map, weights, depth_map, lazy_colour ? 1 : 0, cmx)
        if (S == 64) { if (wo) DSN_C16(true, 4); else DSN_C16(false, 4); }
        else { if (wo) DSN_C16(true, 8); else DSN_C16(false, 8); }
#undef DSN_C16
        return;
    }
    if (!colour && !rgb_map)
        hipLaunchKernelGGL(k_composite<true>, dim3((R + 3) / 4), dim3(256), 0, st, colour, sigma, transparent, z_vals, ray_d,
                           noise, R, S, rgb_map, disp_map, acc_map, weights, depth_map, 0, (int32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_composite<false>, dim3((R + 3) / 4), dim3(256), 0, st, colour, sigma, transparent, z_vals, ray_d,
                           noise, R, S, rgb_map, disp_map, acc_map, weights, depth_map, lazy_colour ? 1 : 0, colour ? colour_max : nullptr);
}

// ---------------------------------------------------------------------------------------------
// Front-to-back evaluation with exact ray termination (DSN_EARLY_STOP, eval mode).
// utils/nerf_net_utils.py:24-39: weight_i = alpha_i * T_i with T_i = prod_{j<i}(1 - alpha_j + 1e-10) non-increasing along the ray, so
// once T < eps every later sample has weight < eps and all of them together add less than eps to acc_map (eps * colour to the
// pixel, eps * far to the depth).  The samples of a frame are therefore evaluated in slices of `L` samples along the rays; after
// each slice T of every ray is advanced with the densities just computed (the compositor's own formula), and the next slice
// leaves out the rays that are finished.  A density the split-fp16 pass flagged (NaN until the fp32 fallback has run) counts
// as 0 here: T is then an upper bound and no ray ends early because of it.
// ---------------------------------------------------------------------------------------------
// List building is workgroup-aggregated: a workgroup takes DSN_AGG_ITEMS consecutive entries, places them with LDS atomics and
// reserves its share of every output list with ONE global atomic per list (a global atomic per wavefront on a handful of hot
// counters costs ~15 ns each: 9 ms for the 8-way split of a 6.6 M-entry list).
#define DSN_AGG_PER_THREAD 8
#define DSN_AGG_ITEMS (256 * DSN_AGG_PER_THREAD)
// The slices of a frame: slice k = samples [b[k], b[k + 1]) of every ray, K <= 32 slices.  Uniform (b[k] = k L) by default; a caller that
// has the statistics of a probe frame passes its own lengths (dsn_render_rays_ex: longer slices where few rays end, see dsnerf.h).
struct DsnSliceBounds { int K; int b[DSN_STOP_MAX_SLICES + 1]; };
__device__ __forceinline__ int dsn_slice_of(const DsnSliceBounds& sb, int i) {
    int k = 0;
#pragma unroll 4
    for (int q = 1; q < sb.K; ++q) k += i >= sb.b[q] ? 1 : 0;
    return k;
}
// the active list split by slice -> lists + R * b[k] (slice k holds at most R * (b[k + 1] - b[k]) entries), counts[k]
__global__ void __launch_bounds__(256) k_slice_bucket(const int32_t* __restrict__ active, const int32_t* __restrict__ active_count, int S,
                                                       DsnSliceBounds sb, int64_t R, int32_t* __restrict__ lists,
                                                       int32_t* __restrict__ counts) {
    __shared__ int s_cnt[32], s_base[32];
    const int n = *active_count;
    const int K = sb.K;
    for (int64_t base = (int64_t)blockIdx.x * DSN_AGG_ITEMS; base < n; base += (int64_t)gridDim.x * DSN_AGG_ITEMS) {
        if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        int32_t idx[DSN_AGG_PER_THREAD];
        int off[DSN_AGG_PER_THREAD], kk[DSN_AGG_PER_THREAD];
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            const int64_t i = base + j * 256 + threadIdx.x;
            kk[j] = -1;
            if (i < n) {
                idx[j] = active[i];
                kk[j] = dsn_slice_of(sb, idx[j] % S);
                off[j] = atomicAdd(&s_cnt[kk[j]], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < K && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(counts + threadIdx.x, s_cnt[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j)
            if (kk[j] >= 0) lists[R * (int64_t)sb.b[kk[j]] + s_base[kk[j]] + off[j]] = idx[j];
        __syncthreads();
    }
}
// The samples of slice k (k >= 1) whose ray is still alive (T >= eps); stopped[0] += the others.  The transmittance is advanced HERE
// (round 4: k_advance_T's 15 launches per frame are gone): Tk[r] = (T of ray r, number of slices it covers).  An entry whose ray's
// pair covers fewer than k slices multiplies the missing slices' factors in - k_composite's alpha: transparent samples and densities
// <= 0 give alpha = 0, a flagged density (NaN until the fp32 fallback has run) counts as 0, so T is then an upper bound - in sample
// order, and publishes (T', k) with ONE 8-byte store.  Entries of the same ray in other waves may read the old or the new pair: both
// lead to the same T' (same factors, same order), so the lists do not depend on the race.  A ray without samples in a stretch of
// slices simply catches up at its next entry.  eps comes from the packed scalars (dsn_stop_eps_scaled: S and the colour scale).
__device__ __forceinline__ float dsn_slice_factor(const float* __restrict__ sigma, const uint8_t* __restrict__ transparent,
                                                  const float* __restrict__ z_vals, float dn, int64_t g0, int S, int s0, int s1) {
    float P = 1.0f;
    float z = z_vals[g0 + s0];
    for (int i = s0; i < s1; ++i) {
        const float zn = (i + 1 < S) ? z_vals[g0 + i + 1] : 0.f;
        const float dist = ((i + 1 < S) ? (zn - z) : 1e10f) * dn;
        float sg = sigma[g0 + i];
        if (transparent[g0 + i]) sg = 0.f;
        sg = sg > 0.f ? sg : 0.f;
        const float alpha = 1.0f - expf(-sg * dist);
        P *= (1.0f - alpha) + 1e-10f;
        z = zn;
    }
    return P;
}
// LC > 0: the slice length at compile time - the factor of slice k - 1 (the one nearly every entry needs) is straight-line code, so the
// loads of a thread's eight entries are issued together instead of one dependent chain per entry inside a branch (the first fused
// version took 53 us per slice against 25 us for the two kernels it replaced; the catch-up over older slices stays a rare branch)
template <int LC>
__global__ void __launch_bounds__(256) k_slice_alive(const int32_t* __restrict__ list, const int32_t* __restrict__ count, int S, int L, int k,
                                                      unsigned long long* __restrict__ Tk, const float* __restrict__ sigma,
                                                      const uint8_t* __restrict__ transparent, const float* __restrict__ z_vals,
                                                      const float* __restrict__ ray_d, const float* __restrict__ scal,
                                                      int32_t* __restrict__ out, int32_t* __restrict__ out_count,
                                                      int32_t* __restrict__ stopped) {
    __shared__ int s_cnt[2], s_base;
    const int n = *count;
    const float eps = dsn_stop_eps_scaled(S, scal[6]);
    if (LC > 0) L = LC;
    const int p0 = (k - 1) * L, p1 = k * L < S ? k * L : S;          // slice k - 1
    for (int64_t base = (int64_t)blockIdx.x * DSN_AGG_ITEMS; base < n; base += (int64_t)gridDim.x * DSN_AGG_ITEMS) {
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        int32_t idx[DSN_AGG_PER_THREAD];
        int off[DSN_AGG_PER_THREAD];
        float Tn[DSN_AGG_PER_THREAD], P[DSN_AGG_PER_THREAD], dn[DSN_AGG_PER_THREAD];
        int kd[DSN_AGG_PER_THREAD];
        bool ok[DSN_AGG_PER_THREAD];
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            const int64_t i = base + j * 256 + threadIdx.x;
            ok[j] = i < n;
            idx[j] = list[ok[j] ? i : (int64_t)n - 1];
        }
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            const int r = idx[j] / S;
            // (single-copy-atomic 8-byte access at WORKGROUP scope = a plain global_load_dwordx2: a stale pair is harmless - the entry
            //  recomputes the same value - and the system scope an unqualified atomic gets bypasses the caches: 2x the kernel time)
            const unsigned long long pr = __hip_atomic_load(Tk + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            Tn[j] = __uint_as_float((uint32_t)pr);
            kd[j] = (int)(pr >> 32);
            const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
            dn[j] = dsn_norm3(d);
            if (LC > 0) {
                // slice k - 1 of this ray, unrolled: k_composite's alpha (see dsn_slice_factor: the same factors in the same order)
                const int64_t g0 = (int64_t)r * S;
                constexpr int LA = LC > 0 ? LC : 1;      // (LC = 0 never gets here)
                float z[LA + 1], sg[LA];
                uint8_t tr[LA];
#pragma unroll
                for (int q = 0; q <= LC; ++q) z[q] = p0 + q < S ? z_vals[g0 + p0 + q] : 0.f;
#pragma unroll
                for (int q = 0; q < LC; ++q) { const bool in = p0 + q < p1; sg[q] = in ? sigma[g0 + p0 + q] : 0.f; tr[q] = in ? transparent[g0 + p0 + q] : (uint8_t)1; }
                float pp = 1.0f;
#pragma unroll
                for (int q = 0; q < LC; ++q) {
                    if (p0 + q < p1) {
                        const float dist = ((p0 + q + 1 < S) ? (z[q + 1] - z[q]) : 1e10f) * dn[j];
                        float sv = tr[q] ? 0.f : sg[q];
                        sv = sv > 0.f ? sv : 0.f;
                        const float alpha = 1.0f - expf(-sv * dist);
                        pp *= (1.0f - alpha) + 1e-10f;
                    }
                }
                P[j] = pp;
            }
        }
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            off[j] = -1;
            if (ok[j]) {
                const int r = idx[j] / S;
                float T = Tn[j];
                int kq = kd[j];
                if (kq < k) {
                    for (; kq < (LC > 0 ? k - 1 : k); ++kq) {       // (LC > 0: only the slices before k - 1 - a ray that had no entry for a while)
                        const int s0 = kq * L, s1 = (kq + 1) * L < S ? (kq + 1) * L : S;
                        T = T * dsn_slice_factor(sigma, transparent, z_vals, dn[j], (int64_t)r * S, S, s0, s1);
                    }
                    if (LC > 0) T = T * P[j];
                    __hip_atomic_store(Tk + r, ((unsigned long long)(uint32_t)k << 32) | (unsigned long long)__float_as_uint(T), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (!(T < eps)) off[j] = atomicAdd(&s_cnt[0], 1);
                else atomicAdd(&s_cnt[1], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = s_cnt[0] ? atomicAdd(out_count, s_cnt[0]) : 0;
            if (s_cnt[1]) atomicAdd(stopped, s_cnt[1]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j)
            if (off[j] >= 0) out[s_base + off[j]] = idx[j];
        __syncthreads();
    }
}
// The per-ray alternative (DSN_STOP_ADVANCE=ray): one lane per (ray, sample of slice k - 1), groups of G lanes share a ray (consecutive
// samples: coalesced reads), the factors multiplied in sample order by the group's first lane - the same product, bit for bit, as an
// entry of k_slice_alive computes - and (T', k) published for EVERY ray; the list filter behind it then finds every pair up to date.
template <int G>
__global__ void __launch_bounds__(256) k_advance_T(const float* __restrict__ sigma, const uint8_t* __restrict__ transparent,
                                                    const float* __restrict__ z_vals, const float* __restrict__ ray_d, int R, int S,
                                                    int s0, int s1, int k, unsigned long long* __restrict__ Tk) {
    const int64_t gl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gl / G;
    const int j = (int)(gl % G);
    if (r >= R) return;
    const unsigned long long pr = Tk[r];
    const int kd = (int)(pr >> 32);
    float T = __uint_as_float((uint32_t)pr);
    const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float dn = dsn_norm3(d);
    // (every ray is advanced every slice: a ray is exactly one slice behind here)
    const int i = s0 + j;
    float fac = 1.0f;
    if (i < s1) {
        const int64_t g = r * S + i;
        const float z = z_vals[g];
        const float dist = ((i + 1 < S) ? (z_vals[g + 1] - z) : 1e10f) * dn;
        float sv = sigma[g];
        if (transparent[g]) sv = 0.f;
        sv = sv > 0.f ? sv : 0.f;
        fac = (1.0f - (1.0f - expf(-sv * dist))) + 1e-10f;
    }
    // sample order: P = ((f0 f1) f2) f3 ..., as dsn_slice_factor multiplies (P starts at 1)
    float P = 1.0f;
#pragma unroll
    for (int q = 0; q < G; ++q) P *= __shfl(fac, (int)((threadIdx.x & 63) - j + q));
    if (j == 0 && kd < k) Tk[r] = ((unsigned long long)(uint32_t)k << 32) | (unsigned long long)__float_as_uint(T * P);
}
// T of every ray over slice k - 1 = samples [s0, s1) (at most 64)
void dsn_launch_advance_T(const float* sigma, const uint8_t* transparent, const float* z_vals, const float* ray_d, int R, int S, int s0,
                          int s1, int k, void* Tk, hipStream_t st) {
    const int n = s1 - s0;
#define DSN_ADV(G) hipLaunchKernelGGL(k_advance_T<G>, dim3((unsigned)(((int64_t)R * G + 255) / 256)), dim3(256), 0, st, sigma, transparent, \
                                      z_vals, ray_d, R, S, s0, s1, k, (unsigned long long*)Tk)
    if (n <= 1) DSN_ADV(1);
    else if (n <= 2) DSN_ADV(2);
    else if (n <= 4) DSN_ADV(4);
    else if (n <= 8) DSN_ADV(8);
    else if (n <= 16) DSN_ADV(16);
    else if (n <= 32) DSN_ADV(32);
    else DSN_ADV(64);
#undef DSN_ADV
}
__global__ void __launch_bounds__(256) k_fill_u64(unsigned long long* __restrict__ p, int64_t n, unsigned long long v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) k_fill_f32(float* __restrict__ p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
// Shading list: the samples of the sigma > 0 list whose compositing weight (k_composite's, from the densities alone) is not below
// eps - the others add less than eps * colour each to their pixel, so d sigma/dx, the normal and the lighting MLP are skipped for
// them (their colour stays 0).  sel = their SLOTS on the sigma > 0 list (the reverse pass reads the relu records by slot; only slots
// below rec_cap, the rest takes the single-launch overflow pass), lit = their sample indices (normals, lighting).  A NaN weight
// keeps the sample.
__global__ void __launch_bounds__(256) k_cull_lit(const int32_t* __restrict__ pos, const int32_t* __restrict__ pos_count, int64_t rec_cap,
                                                   const float* __restrict__ weight, const float* __restrict__ sigma, int S,
                                                   const float* __restrict__ scal, int32_t* __restrict__ sel, int32_t* __restrict__ sel_count, int32_t* __restrict__ lit,
                                                   int32_t* __restrict__ lit_count, int32_t* __restrict__ culled, float* __restrict__ colour) {
    __shared__ int s_cnt[3], s_base[2];
    const int n = *pos_count;
    const float eps = dsn_stop_eps_scaled(S, scal[6]);
    for (int64_t base = (int64_t)blockIdx.x * DSN_AGG_ITEMS; base < n; base += (int64_t)gridDim.x * DSN_AGG_ITEMS) {
        if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        int32_t idx[DSN_AGG_PER_THREAD];
        int osel[DSN_AGG_PER_THREAD], olit[DSN_AGG_PER_THREAD];
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            const int64_t i = base + j * 256 + threadIdx.x;
            osel[j] = olit[j] = -1;
            if (i < n) {
                idx[j] = pos[i];
                // (a flagged density is NaN until the fp32 fallback has run; the weights took it as 0, so the sample itself must stay)
                const bool keep = i >= rec_cap || !(weight[idx[j]] < eps) || sigma[idx[j]] != sigma[idx[j]];
                if (keep) {
                    olit[j] = atomicAdd(&s_cnt[1], 1);
                    if (i < rec_cap) osel[j] = atomicAdd(&s_cnt[0], 1);
                } else {
                    atomicAdd(&s_cnt[2], 1);
                    // (its density is positive, so the compositor reads its colour: the colour of a sample that is not shaded is 0)
                    colour[3 * (int64_t)idx[j]] = 0.f; colour[3 * (int64_t)idx[j] + 1] = 0.f; colour[3 * (int64_t)idx[j] + 2] = 0.f;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base[0] = s_cnt[0] ? atomicAdd(sel_count, s_cnt[0]) : 0;
            s_base[1] = s_cnt[1] ? atomicAdd(lit_count, s_cnt[1]) : 0;
            if (s_cnt[2]) atomicAdd(culled, s_cnt[2]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DSN_AGG_PER_THREAD; ++j) {
            const int64_t i = base + j * 256 + threadIdx.x;
            if (osel[j] >= 0) sel[s_base[0] + osel[j]] = (int32_t)i;
            if (olit[j] >= 0) lit[s_base[1] + olit[j]] = idx[j];
        }
        __syncthreads();
    }
}
// statistics for the host's decision (DSN_STOP_STATS, a frame rendered WITHOUT early stop): out[0] += the non-transparent samples
// that lie in a slice whose ray had T < eps when the slice began - what DSN_EARLY_STOP would have left out with uniform slices of L.
// hist (optional, [K][K] ints, K = ceil(S / L) <= 32, zeroed by the caller): hist[g][k] += the non-transparent samples of slice k on rays
// whose first slice with T < eps at its start is g (g = K: never) - from it the host prices ANY grouping of the slices (a group that
// starts at slice a evaluates slice k >= a on the rays with g > a) and picks the schedule of dsn_render_rays_ex.  Row K is stored as row
// g = K - 1 + 1 -> the array has K + 1 rows.
__global__ void __launch_bounds__(256) k_stop_stats(const float* __restrict__ sigma, const uint8_t* __restrict__ transparent,
                                                     const float* __restrict__ z_vals, const float* __restrict__ ray_d, int R, int S,
                                                     int L, const float* __restrict__ scal, int32_t* __restrict__ out,
                                                     int32_t* __restrict__ hist, const int32_t* __restrict__ colour_max, int Lu) {
    // L: samples per slice of the HISTOGRAM; Lu: samples per uniform slice of DSN_EARLY_STOP - what *out counts (the samples the uniform
    // slicing would leave out) follows Lu, as ever; the histogram may be finer (dsn_stop_stats_slice_len)
    __shared__ int s_h[(DSN_STOP_MAX_SLICES + 1) * DSN_STOP_MAX_SLICES];
    const int K = (S + L - 1) / L;
    if (hist) {
        for (int i = threadIdx.x; i < (K + 1) * K; i += 256) s_h[i] = 0;
        __syncthreads();
    }
    const int r = blockIdx.x * 256 + threadIdx.x;
    // the threshold the sliced frames of these parameters will run with: colour scale = max(the scale in `packed`, DSN_STOP_COLOUR_HEADROOM
    // x the largest colour THIS frame's compositor weighed) - what the caller sets from this very frame (the statistics
    // used the unscaled threshold, the frames decided by them a smaller one)
    float cs = scal[6];
    if (colour_max) { const float c = DSN_STOP_COLOUR_HEADROOM * __int_as_float(*colour_max); if (c > cs) cs = c; }
    const float eps = dsn_stop_eps_scaled(S, cs);
    int skipped = 0;
    if (r < R) {
        float t = 1.0f;
        float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
        const float dn = dsn_norm3(d);
        const int64_t g0 = (int64_t)r * S;
        float z = z_vals[g0];
        bool dead = false, dead_u = false;
        int gstar = K;                     // first slice that finds the ray finished at its start
        unsigned long long nt_lo = 0, nt_hi = 0;      // non-transparent samples per slice, 4 bits each would overflow at L > 15: counted below
        for (int i = 0; i < S; ++i) {
            if (i % L == 0) { dead = t < eps; if (dead && gstar == K) gstar = i / L; }
            if (i % Lu == 0) dead_u = t < eps;
            const bool tr = transparent && transparent[g0 + i];
            if (dead_u && !tr) ++skipped;
            const float zn = (i + 1 < S) ? z_vals[g0 + i + 1] : 0.f;
            const float dist = ((i + 1 < S) ? (zn - z) : 1e10f) * dn;
            float s = tr ? 0.f : sigma[g0 + i];
            s = s > 0.f ? s : 0.f;
            t = t * ((1.0f - (1.0f - expf(-s * dist))) + 1e-10f);
            z = zn;
        }
        (void)nt_lo; (void)nt_hi;
        if (hist) {
            for (int k = 0; k < K; ++k) {
                int c = 0;
                for (int i = k * L; i < (k + 1) * L && i < S; ++i) c += (transparent && transparent[g0 + i]) ? 0 : 1;
                if (c) atomicAdd(&s_h[gstar * K + k], c);
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) skipped += __shfl_xor(skipped, off);
    if ((threadIdx.x & 63) == 0 && skipped) atomicAdd(out, skipped);
    if (hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < (K + 1) * K; i += 256) if (s_h[i]) atomicAdd(hist + i, s_h[i]);
    }
}

void dsn_launch_slice_bucket(const int32_t* active, const int32_t* active_count, int64_t N, int S, int R, const int* bounds, int K,
                             int32_t* lists, int32_t* counts, hipStream_t st) {
    DsnSliceBounds sb;
    sb.K = K;
    for (int k = 0; k <= DSN_STOP_MAX_SLICES; ++k) sb.b[k] = k <= K ? bounds[k] : S;
    const int64_t blocks = std::min<int64_t>((N + DSN_AGG_ITEMS - 1) / DSN_AGG_ITEMS, 2048);
    hipLaunchKernelGGL(k_slice_bucket, dim3((unsigned)blocks), dim3(256), 0, st, active, active_count, S, sb, (int64_t)R, lists, counts);
}
void dsn_launch_slice_alive(const int32_t* list, const int32_t* count, int64_t N, int S, int L, int k, void* Tk, const float* sigma,
                            const uint8_t* transparent, const float* z_vals, const float* ray_d, const float* packed_scal, int32_t* out,
                            int32_t* out_count, int32_t* stopped, hipStream_t st, bool pairs_current) {
    const int64_t blocks = std::min<int64_t>((N + DSN_AGG_ITEMS - 1) / DSN_AGG_ITEMS, 2048);
#define DSN_ALIVE(LC) hipLaunchKernelGGL(k_slice_alive<LC>, dim3((unsigned)blocks), dim3(256), 0, st, list, count, S, L, k, (unsigned long long*)Tk, \
                                         sigma, transparent, z_vals, ray_d, packed_scal, out, out_count, stopped)
    if (pairs_current) DSN_ALIVE(0);      // (every pair was advanced by k_advance_T: nothing to compute, the generic form only filters)
    else if (L == 4) DSN_ALIVE(4);
    else if (L == 8) DSN_ALIVE(8);
    else DSN_ALIVE(0);
#undef DSN_ALIVE
}
// Tk[r] = (T = 1, covers 0 slices)
void dsn_launch_slice_T_init(void* Tk, int R, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, (unsigned long long*)Tk, (int64_t)R,
                       (unsigned long long)0x3f800000ull);
}
void dsn_launch_fill_f32(float* p, int64_t n, float v, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}
void dsn_launch_cull_lit(const int32_t* pos, const int32_t* pos_count, int64_t N, int64_t rec_cap, const float* weight,
                         const float* sigma, int S, const float* packed_scal, int32_t* sel, int32_t* sel_count, int32_t* lit, int32_t* lit_count,
                         int32_t* culled, float* colour, hipStream_t st) {
    const int64_t blocks = std::min<int64_t>((N + DSN_AGG_ITEMS - 1) / DSN_AGG_ITEMS, 2048);
    hipLaunchKernelGGL(k_cull_lit, dim3((unsigned)blocks), dim3(256), 0, st, pos, pos_count, rec_cap, weight, sigma, S, packed_scal, sel, sel_count, lit,
                       lit_count, culled, colour);
}
void dsn_launch_stop_stats(const float* sigma, const uint8_t* transparent, const float* z_vals, const float* ray_d, int R, int S, int L,
                           const float* packed_scal, int32_t* out, hipStream_t st, int32_t* hist, const int32_t* colour_max, int Lu) {
    hipLaunchKernelGGL(k_stop_stats, dim3((R + 255) / 256), dim3(256), 0, st, sigma, transparent, z_vals, ray_d, R, S, L, packed_scal, out, hist, colour_max, Lu > 0 ? Lu : L);
}

// ---------------------------------------------------------------------------------------------
// "next" row f-2: camera rays + AABB near/far on the device (whole-image path of
// utils/rays_utils.py:16-30 get_rays, :63-97 get_near_far as used by my_sample_ray(nrays<=0), :176-184).
// The reference does this in float64 numpy on the CPU and casts to float32: rays are produced in double,
// ROUNDED to float32 (:177-178), and the slab intersections are evaluated in double FROM THE ROUNDED rays (:179);
// the same order is kept here so that results are the reference's to the last float32 bit (up to BLAS summation order
// in the 1e-16 range).  One thread per pixel; K^-1 by the adjugate.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_camera_rays(const double* __restrict__ K, const double* __restrict__ Rm,
                                                      const double* __restrict__ T, const double* __restrict__ bounds,
                                                      int H, int W, float* __restrict__ ray_o, float* __restrict__ ray_d,
                                                      float* __restrict__ near, float* __restrict__ far,
                                                      uint8_t* __restrict__ mask_at_box) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const double k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5], k20 = K[6], k21 = K[7], k22 = K[8];
    const double det = k00 * (k11 * k22 - k12 * k21) - k01 * (k10 * k22 - k12 * k20) + k02 * (k10 * k21 - k11 * k20);
    const double id = 1.0 / det;
    const double Ki[9] = {(k11 * k22 - k12 * k21) * id, (k02 * k21 - k01 * k22) * id, (k01 * k12 - k02 * k11) * id,
                          (k12 * k20 - k10 * k22) * id, (k00 * k22 - k02 * k20) * id, (k02 * k10 - k00 * k12) * id,
                          (k10 * k21 - k11 * k20) * id, (k01 * k20 - k00 * k21) * id, (k00 * k11 - k01 * k10) * id};
    // rays_o = -R^T T
    double o[3];
    for (int c = 0; c < 3; ++c) o[c] = -(Rm[0 * 3 + c] * T[0] + Rm[1 * 3 + c] * T[1] + Rm[2 * 3 + c] * T[2]);
    const double i = (double)(float)(p % W), j = (double)(float)(p / W);
    double pc[3], pw[3];
    for (int c = 0; c < 3; ++c) pc[c] = i * Ki[c * 3 + 0] + j * Ki[c * 3 + 1] + Ki[c * 3 + 2];   // xy1 . Kinv^T
    for (int c = 0; c < 3; ++c)
        pw[c] = (pc[0] - T[0]) * Rm[0 * 3 + c] + (pc[1] - T[1]) * Rm[1 * 3 + c] + (pc[2] - T[2]) * Rm[2 * 3 + c];   // (pc - T) . R
    float of[3], df[3];
    for (int c = 0; c < 3; ++c) { of[c] = (float)o[c]; df[c] = (float)(pw[c] - o[c]); }
    for (int c = 0; c < 3; ++c) { ray_o[3 * p + c] = of[c]; ray_d[3 * p + c] = df[c]; }
    // get_near_far on the float32-rounded rays, in double
    const double ro[3] = {(double)of[0], (double)of[1], (double)of[2]}, rd[3] = {(double)df[0], (double)df[1], (double)df[2]};
    double b[2][3];
    for (int c = 0; c < 3; ++c) { b[0][c] = bounds[c] + (-0.01); b[1][c] = bounds[3 + c] + 0.01; }
    const double eps = 1e-6;
    int hits = 0;
    double dsel[2] = {0.0, 0.0};
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c) {   // plane order of the reference's reshape(-1, 6): min xyz, then max xyz
            const double dint = (b[s][c] - ro[c]) / rd[c];
            const double px = dint * rd[0] + ro[0], py = dint * rd[1] + ro[1], pz = dint * rd[2] + ro[2];
            const bool in = (px >= b[0][0] - eps) && (px <= b[1][0] + eps) && (py >= b[0][1] - eps) && (py <= b[1][1] + eps) &&
                            (pz >= b[0][2] - eps) && (pz <= b[1][2] + eps);
            if (in) {
                if (hits < 2) {
                    const double ex = px - ro[0], ey = py - ro[1], ez = pz - ro[2];
                    dsel[hits] = sqrt(ex * ex + ey * ey + ez * ez);
                }
                ++hits;
            }
        }
    const bool m = hits == 2;   // "intersect exactly twice" (:86)
    // np.linalg.norm on the float32 rays stays in float32 (:92): sqrt((x*x + y*y) + z*z), unfused
    const double nr = (double)sqrtf((df[0] * df[0] + df[1] * df[1]) + df[2] * df[2]);
    const double d0 = dsel[0] / nr, d1 = dsel[1] / nr;
    mask_at_box[p] = m ? 1 : 0;
    near[p] = m ? (float)fmin(d0, d1) : 0.0f;
    far[p] = m ? (float)fmax(d0, d1) : 0.0f;
}

// Human3.6M convention (utils/h36m_utils.py:14-28 get_rays, :61-76 get_near_far, composed by get_rays_within_bounds
// :162-176 / the test split of sample_ray_h36m :147-157): the direction is NORMALISED in float64 before the cast to
// float32 (:26), and the box test is the float32 slab test on the unit direction with the reference's +-1e-5 clamp of
// near-zero components (:64-66), against the FIRST ray's origin (ray_o[:1], :67-68) and the UNPADDED float32 bounds;
// near / far are divided by the float32 norm of the (already unit) direction (:74-75).  All of get_near_far is float32
// numpy: one rounding per operation, kept here (-ffp-contract=off).
__global__ void __launch_bounds__(256) k_camera_rays_h36m(const double* __restrict__ K, const double* __restrict__ Rm,
                                                           const double* __restrict__ T, const double* __restrict__ bounds,
                                                           int H, int W, float* __restrict__ ray_o, float* __restrict__ ray_d,
                                                           float* __restrict__ near, float* __restrict__ far,
                                                           uint8_t* __restrict__ mask_at_box) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const double k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5], k20 = K[6], k21 = K[7], k22 = K[8];
    const double det = k00 * (k11 * k22 - k12 * k21) - k01 * (k10 * k22 - k12 * k20) + k02 * (k10 * k21 - k11 * k20);
    const double id = 1.0 / det;
    const double Ki[9] = {(k11 * k22 - k12 * k21) * id, (k02 * k21 - k01 * k22) * id, (k01 * k12 - k02 * k11) * id,
                          (k12 * k20 - k10 * k22) * id, (k00 * k22 - k02 * k20) * id, (k02 * k10 - k00 * k12) * id,
                          (k10 * k21 - k11 * k20) * id, (k01 * k20 - k00 * k21) * id, (k00 * k11 - k01 * k10) * id};
    double o[3];
    for (int c = 0; c < 3; ++c) o[c] = -(Rm[0 * 3 + c] * T[0] + Rm[1 * 3 + c] * T[1] + Rm[2 * 3 + c] * T[2]);
    const double i = (double)(float)(p % W), j = (double)(float)(p / W);
    double pc[3], pw[3], dd[3];
    for (int c = 0; c < 3; ++c) pc[c] = i * Ki[c * 3 + 0] + j * Ki[c * 3 + 1] + Ki[c * 3 + 2];
    for (int c = 0; c < 3; ++c)
        pw[c] = (pc[0] - T[0]) * Rm[0 * 3 + c] + (pc[1] - T[1]) * Rm[1 * 3 + c] + (pc[2] - T[2]) * Rm[2 * 3 + c];
    for (int c = 0; c < 3; ++c) dd[c] = pw[c] - o[c];
    const double nd = sqrt((dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]);       // np.linalg.norm(axis=2) in float64 (:26)
    float of[3], df[3];
    for (int c = 0; c < 3; ++c) { of[c] = (float)o[c]; df[c] = (float)(dd[c] / nd); }
    for (int c = 0; c < 3; ++c) { ray_o[3 * p + c] = of[c]; ray_d[3 * p + c] = df[c]; }
    // get_near_far (:61-76), float32 throughout
    const float nrm = sqrtf((df[0] * df[0] + df[1] * df[1]) + df[2] * df[2]);
    float tn = -INFINITY, tf = INFINITY;
    for (int c = 0; c < 3; ++c) {
        float v = dsn_div(df[c], nrm);
        if (v < 1e-5f && v > -1e-10f) v = 1e-5f;          // the two clamps in the reference's order (:65-66)
        if (v > -1e-5f && v < 1e-10f) v = -1e-5f;
        const float bmin = (float)bounds[c], bmax = (float)bounds[3 + c];
        const float a = dsn_div(bmin - of[c], v), b = dsn_div(bmax - of[c], v);   // ray_o[:1]: every ray shares the camera origin
        tn = fmaxf(tn, fminf(a, b));
        tf = fminf(tf, fmaxf(a, b));
    }
    const bool m = tn < tf;
    mask_at_box[p] = m ? 1 : 0;
    near[p] = m ? dsn_div(tn, nrm) : 0.0f;
    far[p] = m ? dsn_div(tf, nrm) : 0.0f;
}

void dsn_launch_camera_rays(const double* K, const double* R, const double* T, const double* bounds, int H, int W,
                            float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask, hipStream_t st, int h36m) {
    if (h36m)
        hipLaunchKernelGGL(k_camera_rays_h36m, dim3((H * W + 255) / 256), dim3(256), 0, st, K, R, T, bounds, H, W, ray_o, ray_d,
                           near, far, mask);
    else
        hipLaunchKernelGGL(k_camera_rays, dim3((H * W + 255) / 256), dim3(256), 0, st, K, R, T, bounds, H, W, ray_o, ray_d,
                           near, far, mask);
}
