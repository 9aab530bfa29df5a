/*
 * dsn_oracle.c - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32, scalar semantics) of the volume-rendering hot path of
 * zyhbili/Dual-Space-NeRF.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / timed CPU baseline.
 * The product (dual-space-nerf_amd/) never links or calls it.
 *
 * Parity status: PINNED against the reference itself - tests/test_oracle_golden.py checks
 * every function below against tests/golden/(*).npz, which tests/golden/make_golden.py
 * produced by importing and running /root/reference in the build container.  The one
 * unpinned call is pytorch3d.ops.knn_points (un-vendored dependency, pytorch3d==0.4.0,
 * /root/reference/requirements.txt:56): restated from its published contract, see
 * orc_nearest_face.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off: no implicit fma; fmaf is explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* --------------------------------------------------------------------------------------
 * parameter table: DualSpaceNeRF.state_dict() order (model/spacenet.py:18-81,152-172,191-205)
 * ------------------------------------------------------------------------------------ */
enum {
    P_EMB = 0,
    P_S1_0W, P_S1_0B, P_S1_2W, P_S1_2B, P_S1_4W, P_S1_4B, P_S1_6W, P_S1_6B,
    P_S2_0W, P_S2_0B, P_S2_2W, P_S2_2B, P_S2_4W, P_S2_4B,
    P_DEN_W, P_DEN_B, P_RGB1_W, P_RGB1_B, P_RGB3_W, P_RGB3_B,
    P_L0_W, P_L0_B, P_L2_W, P_L2_B, P_L4_W, P_L4_B,
    P_PM0_W, P_PM0_B, P_PM2_W, P_PM2_B, P_PM4_W, P_PM4_B,
    P_COUNT
};

/* --------------------------------------------------------------------------------------
 * small helpers (all float32, one rounding per operation)
 * ------------------------------------------------------------------------------------ */
static inline float sum3(float a, float b, float c) { return (a + b) + c; }
static inline float dot3(const float* a, const float* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
/* torch.norm(x, dim=-1) on 3 floats: the ATen reduction is acc = fma(x_i, x_i, acc) (verified bit-exact
 * on 200k random vectors against torch 2.10 CPU), then sqrt */
static inline float norm3(const float* a) { return sqrtf(fmaf(a[2], a[2], fmaf(a[1], a[1], a[0] * a[0]))); }
/* torch.cross: each component is fma(a_i, b_j, -(a_j * b_i)) (verified bit-exact the same way) */
static inline void cross3(const float* a, const float* b, float* o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

/* torch.linspace(0,1,S) float32: step=(end-start)/(S-1); i<S/2: start+step*i else end-step*(S-1-i)
 * (ATen RangeFactories, scalar path); used by utils/pts_utils.py:4.  The vectorised ATen path differs
 * from this in the last bit depending on the host's SIMD width, so callers that need bit parity with a
 * particular torch build pass torch.linspace's own output as t_vals (orc_sample_gg). */
/* thread count of the OpenMP loops below (bench.py's cpu_baseline: the host cores the cgroup really grants, not the hardware
 * thread count); returns the count in effect */
ORC_API int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

ORC_API void orc_linspace01(int S, float* t) {
    if (S == 1) { t[0] = 0.f; return; }
    float step = 1.0f / (float)(S - 1);
    int half = S / 2;
    for (int i = 0; i < S; ++i) t[i] = (i < half) ? (0.f + step * (float)i) : (1.0f - step * (float)(S - 1 - i));
}

/* --------------------------------------------------------------------------------------
 * utils/pts_utils.py:18-58  geometry_guided_ray_marching  +  :3-16 uniform_sampling
 * near/far are updated IN PLACE like the reference (:52-53).  The first ray's origin is
 * used for every ray (:31,:33 ray_o[:,0:1]).  jitter = the torch.rand draw (:12) or NULL.
 * ------------------------------------------------------------------------------------ */
ORC_API void orc_sample_gg(const float* ray_o, const float* ray_d, float* near, float* far, int R,
                           const float* xyz, int V, int S, const float* t_vals, const float* jitter, float* z_vals,
                           float* pts) {
    const float gamma2 = (float)(0.05 * 0.05); /* python double 0.05**2 cast to float32 */
    float* t = (float*)malloc(sizeof(float) * S);
    if (t_vals) memcpy(t, t_vals, sizeof(float) * S);
    else orc_linspace01(S, t);
    const float* o0 = ray_o;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        const float* d = ray_d + 3 * r;
        float nrm = norm3(d);
        float du[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};
        float zmin = 99999.f, zmax = -99999.f;
        int any = 0;
        for (int v = 0; v < V; ++v) {
            float df[3] = {xyz[3 * v] - o0[0], xyz[3 * v + 1] - o0[1], xyz[3 * v + 2] - o0[2]};
            float z0 = sum3(df[0] * du[0], df[1] * du[1], df[2] * du[2]);
            float tmp = sum3(df[0] * df[0], df[1] * df[1], df[2] * df[2]) - z0 * z0;
            if (tmp < gamma2) {
                float dz = sqrtf(gamma2 - tmp);
                float a = z0 - dz, b = z0 + dz;
                if (a < zmin) zmin = a;
                if (b > zmax) zmax = b;
                any = 1;
            }
        }
        zmin = zmin / nrm;
        zmax = zmax / nrm;
        if (any && zmin < zmax) { near[r] = zmin; far[r] = zmax; }
        float n = near[r], f = far[r];
        float* z = z_vals + (size_t)r * S;
        for (int i = 0; i < S; ++i) z[i] = n * (1.0f - t[i]) + f * t[i];
        if (jitter) {
            /* :6-13 stratified jitter: mids of the un-jittered z, z = lower + (upper-lower)*U */
            float zu[S];
            for (int i = 0; i < S; ++i) zu[i] = z[i];
            for (int i = 0; i < S; ++i) {
                float lower = (i == 0) ? zu[0] : 0.5f * (zu[i] + zu[i - 1]);
                float upper = (i == S - 1) ? zu[S - 1] : 0.5f * (zu[i + 1] + zu[i]);
                z[i] = lower + (upper - lower) * jitter[(size_t)r * S + i];
            }
        }
        if (pts) {
            const float* o = ray_o + 3 * r;
            for (int i = 0; i < S; ++i)
                for (int c = 0; c < 3; ++c) pts[((size_t)r * S + i) * 3 + c] = o[c] + d[c] * z[i];
        }
    }
    free(t);
}

/* utils/render_utils.py:94  mesh_centroid = meshes.mean(dim=-2): (v0+v1+v2)/3 on CPU torch */
ORC_API void orc_centroids(const float* verts, const int32_t* faces, int F, float* cent) {
    for (int f = 0; f < F; ++f)
        for (int c = 0; c < 3; ++c)
            cent[3 * f + c] = sum3(verts[3 * faces[3 * f] + c], verts[3 * faces[3 * f + 1] + c], verts[3 * faces[3 * f + 2] + c]) / 3.0f;
}

/* utils/render_utils.py:95 -> pytorch3d.ops.knn_points(K=1) (pytorch3d 0.4.0, not vendored).
 * Published contract: squared L2, accumulated per coordinate (dist += diff*diff, which the
 * CUDA build contracts to an fma chain), smallest first, first index wins ties. */
ORC_API void orc_nearest_face(const float* pts, int64_t N, const float* cent, int F, int32_t* idx) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        float best = INFINITY;
        int bi = 0;
        for (int f = 0; f < F; ++f) {
            float dx = px - cent[3 * f], dy = py - cent[3 * f + 1], dz = pz - cent[3 * f + 2];
            float d = dx * dx;
            d = fmaf(dy, dy, d);
            d = fmaf(dz, dz, d);
            if (d < best) { best = d; bi = f; }
        }
        idx[i] = bi;
    }
}

/* utils/geo_utils.py:181-200 project_point2mesh + :96-113 get_barycentric_coordinates.
 * tri = 3 vertices (v0,v1,v2), row-major 9 floats. */
static void project_pt(const float* p, const float* tri, float* uv, float* h) {
    const float *m0 = tri, *m1 = tri + 3, *m2 = tri + 6;
    float v10[3], v20[3], n[3], tmp[3], q[3];
    for (int c = 0; c < 3; ++c) { v10[c] = m1[c] - m0[c]; v20[c] = m2[c] - m0[c]; }
    cross3(v10, v20, n);
    float nn = norm3(n);
    for (int c = 0; c < 3; ++c) n[c] = n[c] / nn;
    for (int c = 0; c < 3; ++c) tmp[c] = p[c] - m0[c];
    float sd = dot3(tmp, n);
    for (int c = 0; c < 3; ++c) q[c] = p[c] - n[c] * sd;
    /* barycentric: v0 = m2-m0, v1 = m1-m0, v2 = q-m0 */
    float w[3];
    for (int c = 0; c < 3; ++c) w[c] = q[c] - m0[c];
    float dot00 = dot3(v20, v20), dot01 = dot3(v20, v10), dot02 = dot3(v20, w);
    float dot11 = dot3(v10, v10), dot12 = dot3(v10, w);
    float inv = 1.0f / (dot00 * dot11 - dot01 * dot01);
    uv[0] = (dot11 * dot02 - dot01 * dot12) * inv;
    uv[1] = (dot00 * dot12 - dot01 * dot02) * inv;
    *h = sd;
}

/* utils/geo_utils.py:138-156 barycentric_map2can */
static void map2face(const float* uv, float h, const float* tri, float* out) {
    const float *m0 = tri, *m1 = tri + 3, *m2 = tri + 6;
    float v2[3], v1[3], n[3];
    for (int c = 0; c < 3; ++c) { v2[c] = m2[c] - m0[c]; v1[c] = m1[c] - m0[c]; }
    cross3(v1, v2, n);
    float nn = norm3(n);
    for (int c = 0; c < 3; ++c) n[c] = n[c] / nn;
    for (int c = 0; c < 3; ++c) {
        float proj = (m0[c] + uv[0] * v2[c]) + uv[1] * v1[c];
        out[c] = proj + h * n[c];
    }
}

static void gather_tri(const float* verts, const int32_t* faces, int f, float* tri) {
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) tri[3 * k + c] = verts[3 * faces[3 * f + k] + c];
}

/* F.normalize(x, dim=-1): x / max(||x||, 1e-12) */
static void normalize3(const float* a, float* o) {
    float n = norm3(a);
    if (n < 1e-12f) n = 1e-12f;
    for (int c = 0; c < 3; ++c) o[c] = a[c] / n;
}

/* can_render.py:333-379 w2l_without_lbs (+ utils/render_utils.py:103-109 get_transparent_mask).
 * dirs: per-point world ray direction [N,3] (or NULL -> ray_d_can not produced).
 * Outputs (any may be NULL): idx[N], uv[N,2], h[N], transparent[N] (u8), x_c[N,3], ray_d_can[N,3]. */
ORC_API void orc_warp(const float* pts, const float* dirs, int64_t N, const float* xyz, const float* canon,
                      const int32_t* faces, int F, int32_t* idx_out, float* uv_out, float* h_out,
                      uint8_t* transparent, float* x_c, float* ray_d_can) {
    float* cent = (float*)malloc(sizeof(float) * 3 * F);
    int32_t* idx = idx_out ? idx_out : (int32_t*)malloc(sizeof(int32_t) * N);
    orc_centroids(xyz, faces, F, cent);
    orc_nearest_face(pts, N, cent, F, idx);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float tw[9], tc[9], uv[2], h, xc[3];
        gather_tri(xyz, faces, idx[i], tw);
        gather_tri(canon, faces, idx[i], tc);
        project_pt(pts + 3 * i, tw, uv, &h);
        if (uv_out) { uv_out[2 * i] = uv[0]; uv_out[2 * i + 1] = uv[1]; }
        if (h_out) h_out[i] = h;
        if (transparent)
            transparent[i] = (uv[0] > 5.f) || (uv[0] < -4.f) || (uv[1] > 5.f) || (uv[1] < -4.f) || (fabsf(h) > 0.1f);
        map2face(uv, h, tc, xc);
        if (x_c) for (int c = 0; c < 3; ++c) x_c[3 * i + c] = xc[c];
        if (dirs && ray_d_can) {
            float p2[3], uv2[2], h2, xe[3], df[3];
            for (int c = 0; c < 3; ++c) p2[c] = pts[3 * i + c] + dirs[3 * i + c];
            project_pt(p2, tw, uv2, &h2);
            map2face(uv2, h2, tc, xe);
            for (int c = 0; c < 3; ++c) df[c] = xe[c] - xc[c];
            normalize3(df, ray_d_can + 3 * i);
        }
    }
    free(cent);
    if (!idx_out) free(idx);
}

/* dormant alternate of the warp: utils/render_utils.py:352-403 compute_nn_mesh with
 * :112-164 get_base_blending_weights (bw_type 0 = "rigid_center": mean of the nearest face's three vertex weights;
 * 1 = "rigid_interp": softmax over the three vertex DISTANCES as written there) and utils/blend_utils.py:72-81
 * ppts_to_pts (blend the 24 joint transforms, subtract the translation, apply the inverse rotation).
 * smpl_w [V,24], A [24,16] row-major 4x4.  Outputs (any may be NULL): idx [N], weights [N,24], transparent [N], pts_zero [N,3]. */
ORC_API void orc_lbs_warp(const float* pts, int64_t N, const float* xyz, const int32_t* faces, int F, const float* smpl_w,
                          const float* A, int bw_type, int32_t* idx_out, float* weights, uint8_t* transparent,
                          float* pts_zero) {
    float* cent = (float*)malloc(sizeof(float) * 3 * F);
    int32_t* idx = idx_out ? idx_out : (int32_t*)malloc(sizeof(int32_t) * N);
    orc_centroids(xyz, faces, F, cent);
    orc_nearest_face(pts, N, cent, F, idx);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float tw[9], uv[2], h, bw[24], wk[3];
        const float* p = pts + 3 * i;
        gather_tri(xyz, faces, idx[i], tw);
        project_pt(p, tw, uv, &h);
        if (transparent)
            transparent[i] = (uv[0] > 5.f) || (uv[0] < -4.f) || (uv[1] > 5.f) || (uv[1] < -4.f) || (fabsf(h) > 0.1f);
        const int32_t* vid = faces + 3 * idx[i];
        if (bw_type == 1) {
            float d[3], m, s = 0.f;
            for (int k = 0; k < 3; ++k) {
                float e[3] = {tw[3 * k] - p[0], tw[3 * k + 1] - p[1], tw[3 * k + 2] - p[2]};
                d[k] = norm3(e);
            }
            m = d[0] > d[1] ? d[0] : d[1];
            m = m > d[2] ? m : d[2];
            for (int k = 0; k < 3; ++k) { wk[k] = expf(d[k] - m); s += wk[k]; }
            for (int k = 0; k < 3; ++k) wk[k] = wk[k] / s;
            for (int j = 0; j < 24; ++j)
                bw[j] = (wk[0] * smpl_w[24 * vid[0] + j] + wk[1] * smpl_w[24 * vid[1] + j]) + wk[2] * smpl_w[24 * vid[2] + j];
        } else {
            for (int j = 0; j < 24; ++j)
                bw[j] = ((smpl_w[24 * vid[0] + j] + smpl_w[24 * vid[1] + j]) + smpl_w[24 * vid[2] + j]) / 3.0f;
        }
        if (weights) for (int j = 0; j < 24; ++j) weights[24 * i + j] = bw[j];
        if (pts_zero) {
            double M[12];      /* blended [R | t], rows 0..2 of the 4x4; double accumulation, then the closed-form inverse */
            for (int e = 0; e < 12; ++e) {
                double acc = 0.0;
                for (int j = 0; j < 24; ++j) acc += (double)bw[j] * (double)A[16 * j + e];
                M[e] = (double)(float)acc;
            }
            const double a = M[0], b = M[1], c = M[2], d = M[4], e2 = M[5], f = M[6], g = M[8], hh = M[9], k2 = M[10];
            const double det = a * (e2 * k2 - f * hh) - b * (d * k2 - f * g) + c * (d * hh - e2 * g);
            const double q[3] = {(double)p[0] - M[3], (double)p[1] - M[7], (double)p[2] - M[11]};
            const double inv[9] = {(e2 * k2 - f * hh) / det, (c * hh - b * k2) / det, (b * f - c * e2) / det,
                                   (f * g - d * k2) / det, (a * k2 - c * g) / det, (c * d - a * f) / det,
                                   (d * hh - e2 * g) / det, (b * g - a * hh) / det, (a * e2 - b * d) / det};
            for (int r = 0; r < 3; ++r)
                pts_zero[3 * i + r] = (float)(inv[3 * r] * q[0] + inv[3 * r + 1] * q[1] + inv[3 * r + 2] * q[2]);
        }
    }
    free(cent);
    if (!idx_out) free(idx);
}

/* --------------------------------------------------------------------------------------
 * network
 * ------------------------------------------------------------------------------------ */
/* y[o] = b[o] + sum_k W[o][k] x[k], k ascending (torch Linear layout [out,in]) */
static void linear_row(const float* W, const float* b, const float* x, int in, int out, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = b ? b[o] : 0.f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) acc += w[k] * x[k];
        y[o] = acc;
    }
}

/* model/spacenet.py:314-331 batch_rod2quat on joints 1..23 (:223) + pose_mlp (:199-205,:236) */
ORC_API void orc_pose_feat(const float* poses24x3, const float* const* P, float* quat92, float* feat16) {
    float q[92];
    for (int j = 0; j < 23; ++j) {
        const float* r = poses24x3 + 3 * (j + 1);
        float a[3] = {r[0] + 1e-16f, r[1] + 1e-16f, r[2] + 1e-16f};
        float angle = norm3(a);
        float half = angle / 2.0f;
        float s = sinf(half), c = cosf(half);
        q[4 * j + 0] = (r[0] / angle) * s;
        q[4 * j + 1] = (r[1] / angle) * s;
        q[4 * j + 2] = (r[2] / angle) * s;
        q[4 * j + 3] = c - 1.0f;
    }
    if (quat92) memcpy(quat92, q, sizeof(q));
    float h1[64], h2[64];
    linear_row(P[P_PM0_W], P[P_PM0_B], q, 92, 64, h1);
    for (int i = 0; i < 64; ++i) h1[i] = h1[i] > 0.f ? h1[i] : 0.f;
    linear_row(P[P_PM2_W], P[P_PM2_B], h1, 64, 64, h2);
    for (int i = 0; i < 64; ++i) h2[i] = h2[i] > 0.f ? h2[i] : 0.f;
    linear_row(P[P_PM4_W], P[P_PM4_B], h2, 64, 16, feat16);
}

/* model/dimension_kernel.py:5-35: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^9 x), cos(2^9 x)] (63) */
static void pos_enc(const float* x, float* pe, float* s_out, float* c_out) {
    pe[0] = x[0]; pe[1] = x[1]; pe[2] = x[2];
    for (int j = 0; j < 10; ++j) {
        float fr = (float)(1 << j);
        for (int a = 0; a < 3; ++a) {
            float t = x[a] * fr;
            float s = sinf(t), c = cosf(t);
            pe[3 + 6 * j + a] = s;
            pe[3 + 6 * j + 3 + a] = c;
            if (s_out) { s_out[3 * j + a] = s; c_out[3 * j + a] = c; }
        }
    }
}

#define PB 32 /* points per block in the batched MLP */

/* out[p][o] = b[o] + sum_k in[p][k] * Wt[k][o]   (Wt = W transposed, [in][out]); k ascending */
static void gemm_fwd(const float* in, int ldin, const float* Wt, const float* b, int K, int O, float* out, int ldout, int np) {
    for (int p = 0; p < np; ++p) {
        float* y = out + (size_t)p * ldout;
        for (int o = 0; o < O; ++o) y[o] = b ? b[o] : 0.f;
        const float* x = in + (size_t)p * ldin;
        for (int k = 0; k < K; ++k) {
            const float xk = x[k];
            const float* w = Wt + (size_t)k * O;
            for (int o = 0; o < O; ++o) y[o] = fmaf(xk, w[o], y[o]);
        }
    }
}
/* gin[p][k] (+)= sum_o g[p][o] * W[o][k]  (W in torch layout [out][in], row stride ldw, column offset applied by caller) */
static void gemm_bwd(const float* g, int ldg, const float* W, int ldw, int O, int K, float* gin, int ldgin, int np, int accumulate) {
    for (int p = 0; p < np; ++p) {
        float* y = gin + (size_t)p * ldgin;
        if (!accumulate) for (int k = 0; k < K; ++k) y[k] = 0.f;
        const float* gp = g + (size_t)p * ldg;
        for (int o = 0; o < O; ++o) {
            const float go = gp[o];
            if (go == 0.f) continue;
            const float* w = W + (size_t)o * ldw;
            for (int k = 0; k < K; ++k) y[k] = fmaf(go, w[k], y[k]);
        }
    }
}

static float* transpose(const float* W, int out, int in) {
    float* t = (float*)malloc(sizeof(float) * (size_t)out * in);
    for (int o = 0; o < out; ++o)
        for (int k = 0; k < in; ++k) t[(size_t)k * out + o] = W[(size_t)o * in + k];
    return t;
}

/* model/spacenet.py:93-148 SpaceNet.forward (use_dir=False) + :301-311 gradient (d sigma / d x_c,
 * restated analytically: reverse-mode through density_net, stage2, stage1 and the encoding).
 * code8: embedding row (already x0 if net.w is set, :126-129); pose16: pose_mlp output.
 * Outputs: sigma[N], essence[N,3], grad[N,3] (grad may be NULL -> density/colour only). */
ORC_API void orc_field(const float* x_c, int64_t N, const float* const* P, const float* code8, const float* pose16,
                       float* sigma, float* essence, float* grad) {
    const int in0 = 87, in4 = 319;
    float* W0t = transpose(P[P_S1_0W], 256, in0);
    float* W1t = transpose(P[P_S1_2W], 256, 256);
    float* W2t = transpose(P[P_S1_4W], 256, 256);
    float* W3t = transpose(P[P_S1_6W], 256, 256);
    float* W4t = transpose(P[P_S2_0W], 256, in4);
    float* W5t = transpose(P[P_S2_2W], 256, 256);
    float* W6t = transpose(P[P_S2_4W], 256, 256);
    float* Wr1t = transpose(P[P_RGB1_W], 128, 256);
    const float* Wd = P[P_DEN_W];
#pragma omp parallel
    {
        float* in_a = (float*)malloc(sizeof(float) * PB * 320);
        float* h[7];
        for (int l = 0; l < 7; ++l) h[l] = (float*)malloc(sizeof(float) * PB * 256);
        float* in4b = (float*)malloc(sizeof(float) * PB * 320);
        float* r1 = (float*)malloc(sizeof(float) * PB * 128);
        float* ga = (float*)malloc(sizeof(float) * PB * 320);
        float* gb = (float*)malloc(sizeof(float) * PB * 320);
        float* sn = (float*)malloc(sizeof(float) * PB * 30);
        float* cs = (float*)malloc(sizeof(float) * PB * 30);
        float* dpe = (float*)malloc(sizeof(float) * PB * 63);
#pragma omp for schedule(dynamic, 1)
        for (int64_t base = 0; base < N; base += PB) {
            int np = (int)((N - base) < PB ? (N - base) : PB);
            for (int p = 0; p < np; ++p) {
                float* a = in_a + (size_t)p * 320;
                for (int k = 0; k < 8; ++k) a[k] = code8[k];
                pos_enc(x_c + 3 * (base + p), a + 8, sn + 30 * p, cs + 30 * p);
                for (int k = 0; k < 16; ++k) a[71 + k] = pose16[k];
            }
            gemm_fwd(in_a, 320, W0t, P[P_S1_0B], in0, 256, h[0], 256, np);
            for (int i = 0; i < np * 256; ++i) h[0][i] = h[0][i] > 0.f ? h[0][i] : 0.f;
            const float* Wt[3] = {W1t, W2t, W3t};
            const float* Bs[3] = {P[P_S1_2B], P[P_S1_4B], P[P_S1_6B]};
            for (int l = 0; l < 3; ++l) {
                gemm_fwd(h[l], 256, Wt[l], Bs[l], 256, 256, h[l + 1], 256, np);
                for (int i = 0; i < np * 256; ++i) h[l + 1][i] = h[l + 1][i] > 0.f ? h[l + 1][i] : 0.f;
            }
            for (int p = 0; p < np; ++p) {
                memcpy(in4b + (size_t)p * 320, h[3] + (size_t)p * 256, sizeof(float) * 256);
                memcpy(in4b + (size_t)p * 320 + 256, in_a + (size_t)p * 320 + 8, sizeof(float) * 63);
            }
            gemm_fwd(in4b, 320, W4t, P[P_S2_0B], in4, 256, h[4], 256, np);
            for (int i = 0; i < np * 256; ++i) h[4][i] = h[4][i] > 0.f ? h[4][i] : 0.f;
            gemm_fwd(h[4], 256, W5t, P[P_S2_2B], 256, 256, h[5], 256, np);
            for (int i = 0; i < np * 256; ++i) h[5][i] = h[5][i] > 0.f ? h[5][i] : 0.f;
            gemm_fwd(h[5], 256, W6t, P[P_S2_4B], 256, 256, h[6], 256, np);
            for (int i = 0; i < np * 256; ++i) h[6][i] = h[6][i] > 0.f ? h[6][i] : 0.f;
            for (int p = 0; p < np; ++p) {
                const float* x = h[6] + (size_t)p * 256;
                float acc = P[P_DEN_B][0];
                for (int k = 0; k < 256; ++k) acc = fmaf(Wd[k], x[k], acc);
                sigma[base + p] = acc;
            }
            if (essence) {
                gemm_fwd(h[6], 256, Wr1t, P[P_RGB1_B], 256, 128, r1, 128, np);
                for (int i = 0; i < np * 128; ++i) r1[i] = r1[i] > 0.f ? r1[i] : 0.f;
                for (int p = 0; p < np; ++p)
                    linear_row(P[P_RGB3_W], P[P_RGB3_B], r1 + (size_t)p * 128, 128, 3, essence + 3 * (base + p));
            }
            if (!grad) continue;
            /* reverse mode: g wrt h6 = Wd * [h6>0] */
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) ga[(size_t)p * 320 + k] = h[6][(size_t)p * 256 + k] > 0.f ? Wd[k] : 0.f;
            gemm_bwd(ga, 320, P[P_S2_4W], 256, 256, 256, gb, 320, np, 0);
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) if (!(h[5][(size_t)p * 256 + k] > 0.f)) gb[(size_t)p * 320 + k] = 0.f;
            gemm_bwd(gb, 320, P[P_S2_2W], 256, 256, 256, ga, 320, np, 0);
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) if (!(h[4][(size_t)p * 256 + k] > 0.f)) ga[(size_t)p * 320 + k] = 0.f;
            gemm_bwd(ga, 320, P[P_S2_0W], in4, 256, in4, gb, 320, np, 0); /* gb[0:256]=d h3, gb[256:319]=d pe (skip) */
            for (int p = 0; p < np; ++p) {
                memcpy(dpe + (size_t)p * 63, gb + (size_t)p * 320 + 256, sizeof(float) * 63);
                for (int k = 0; k < 256; ++k) if (!(h[3][(size_t)p * 256 + k] > 0.f)) gb[(size_t)p * 320 + k] = 0.f;
            }
            gemm_bwd(gb, 320, P[P_S1_6W], 256, 256, 256, ga, 320, np, 0);
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) if (!(h[2][(size_t)p * 256 + k] > 0.f)) ga[(size_t)p * 320 + k] = 0.f;
            gemm_bwd(ga, 320, P[P_S1_4W], 256, 256, 256, gb, 320, np, 0);
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) if (!(h[1][(size_t)p * 256 + k] > 0.f)) gb[(size_t)p * 320 + k] = 0.f;
            gemm_bwd(gb, 320, P[P_S1_2W], 256, 256, 256, ga, 320, np, 0);
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 256; ++k) if (!(h[0][(size_t)p * 256 + k] > 0.f)) ga[(size_t)p * 320 + k] = 0.f;
            gemm_bwd(ga, 320, P[P_S1_0W] + 8, in0, 256, 63, dpe, 63, np, 1); /* columns 8..70 = encoding */
            for (int p = 0; p < np; ++p) {
                const float* d = dpe + (size_t)p * 63;
                const float* s = sn + 30 * p;
                const float* c = cs + 30 * p;
                for (int a = 0; a < 3; ++a) {
                    float acc = d[a];
                    for (int j = 0; j < 10; ++j) {
                        float fr = (float)(1 << j);
                        acc += (d[3 + 6 * j + a] * c[3 * j + a]) * fr;
                        acc -= (d[3 + 6 * j + 3 + a] * s[3 * j + a]) * fr;
                    }
                    grad[3 * (base + p) + a] = acc;
                }
            }
        }
        free(in_a); for (int l = 0; l < 7; ++l) free(h[l]);
        free(in4b); free(r1); free(ga); free(gb); free(sn); free(cs); free(dpe);
    }
    free(W0t); free(W1t); free(W2t); free(W3t); free(W4t); free(W5t); free(W6t); free(Wr1t);
}

/* model/spacenet.py:278-298 normal_local2world */
ORC_API void orc_normal_world(const float* x_c, const float* g, int64_t N, const float* canon, const float* xyz,
                              const int32_t* faces, int F, int32_t* idx_out, float* n_w) {
    float* cent = (float*)malloc(sizeof(float) * 3 * F);
    int32_t* idx = idx_out ? idx_out : (int32_t*)malloc(sizeof(int32_t) * N);
    orc_centroids(canon, faces, F, cent);
    orc_nearest_face(x_c, N, cent, F, idx);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float tc[9], tw[9], uv[2], h, s[3], e[3], xe[3], df[3];
        gather_tri(canon, faces, idx[i], tc);
        gather_tri(xyz, faces, idx[i], tw);
        project_pt(x_c + 3 * i, tc, uv, &h);
        map2face(uv, h, tw, s);
        for (int c = 0; c < 3; ++c) xe[c] = x_c[3 * i + c] + g[3 * i + c];
        project_pt(xe, tc, uv, &h);
        map2face(uv, h, tw, e);
        for (int c = 0; c < 3; ++c) df[c] = e[c] - s[c];
        normalize3(df, n_w + 3 * i);
    }
    free(cent);
    if (!idx_out) free(idx);
}

/* model/spacenet.py:254-265 (rot / light-centre edits of xyz_world) + :174-188 LightingMLP.forward.
 * rot4 (row-major 2x2) and rot_center2 may be NULL; light_shift3 (= light_center - Th) may be NULL. */
ORC_API void orc_lighting(const float* n_w, const float* x_w, const float* view_dir, const float* essence, int64_t N,
                          const float* const* P, const float* rot4, const float* rot_center2, const float* light_shift3,
                          float* colour) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float in[9], h1[128], h2[128];
        float xw[3] = {x_w[3 * i], x_w[3 * i + 1], x_w[3 * i + 2]};
        if (rot4 && rot_center2) {
            float ax = xw[0] - rot_center2[0], ay = xw[1] - rot_center2[1];
            float nx = (ax * rot4[0] + ay * rot4[2]) + rot_center2[0];
            float ny = (ax * rot4[1] + ay * rot4[3]) + rot_center2[1];
            xw[0] = nx; xw[1] = ny;
        }
        if (light_shift3) for (int c = 0; c < 3; ++c) xw[c] += light_shift3[c];
        const float* vd = view_dir + 3 * i;
        float vn = norm3(vd);
        for (int c = 0; c < 3; ++c) { in[c] = n_w[3 * i + c]; in[3 + c] = xw[c]; in[6 + c] = vd[c] / vn; }
        linear_row(P[P_L0_W], P[P_L0_B], in, 9, 128, h1);
        for (int k = 0; k < 128; ++k) h1[k] = h1[k] > 0.f ? h1[k] : 0.f;
        linear_row(P[P_L2_W], P[P_L2_B], h1, 128, 128, h2);
        for (int k = 0; k < 128; ++k) h2[k] = h2[k] > 0.f ? h2[k] : 0.f;
        float o;
        linear_row(P[P_L4_W], P[P_L4_B], h2, 128, 1, &o);
        float w = (o > 0.f ? o : expm1f(o)) + 1.0f; /* ELU(alpha=1) + 1 */
        for (int c = 0; c < 3; ++c) colour[3 * i + c] = w * essence[3 * i + c];
    }
}

/* utils/nerf_net_utils.py:5-56 raw2outputs (white_bkgd=False).  raw [R,S,4] = (rgb, sigma);
 * transparent sigma-zeroing (can_render.py:115-120) is applied by the caller into raw.
 * noise = the torch.randn draw * raw_noise_std (:30-31) or NULL. */
ORC_API void orc_composite(const float* raw, const float* z_vals, const float* rays_d, const float* noise, int R, int S,
                           float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        const float* z = z_vals + (size_t)r * S;
        float dn = norm3(rays_d + 3 * r);
        float T = 1.0f; /* cumprod of [1, 1-alpha+1e-10], exclusive */
        float rgb[3] = {0.f, 0.f, 0.f}, depth = 0.f, acc = 0.f;
        for (int i = 0; i < S; ++i) {
            float dist = (i + 1 < S) ? (z[i + 1] - z[i]) : 1e10f;
            dist = dist * dn;
            float s = raw[((size_t)r * S + i) * 4 + 3];
            if (noise) s = s + noise[(size_t)r * S + i];
            s = s > 0.f ? s : 0.f;
            float alpha = 1.0f - expf(-s * dist);
            float w = alpha * T;
            T = T * ((1.0f - alpha) + 1e-10f);
            if (weights) weights[(size_t)r * S + i] = w;
            for (int c = 0; c < 3; ++c) rgb[c] += w * raw[((size_t)r * S + i) * 4 + c];
            depth += w * z[i];
            acc += w;
        }
        for (int c = 0; c < 3; ++c) rgb_map[3 * r + c] = rgb[c];
        depth_map[r] = depth;
        acc_map[r] = acc;
        float q = depth / acc; /* NaN when acc == 0, like the reference */
        float m = (1e-10f > q) ? 1e-10f : q; /* torch.max(a,b) propagates NaN */
        if (q != q) m = q;
        disp_map[r] = 1.0f / m;
    }
}

/* can_render.py:137-168 Renderer.render (eval or train), whole path on [R] rays.
 * frame state: code8 (embedding row or zeros), light_shift / rot as in orc_lighting.
 * out_raw [R,S,4] optional. */
ORC_API void orc_render(const float* ray_o, const float* ray_d, float* near, float* far, int R, int S,
                        const float* xyz, const float* canon, const int32_t* faces, int V, int F,
                        const float* const* P, const float* poses24x3, const float* code8,
                        const float* rot4, const float* rot_center2, const float* light_shift3,
                        const float* t_vals, const float* jitter, const float* noise,
                        float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map, float* z_out,
                        float* raw_out) {
    int64_t N = (int64_t)R * S;
    float* z = z_out ? z_out : (float*)malloc(sizeof(float) * N);
    float* pts = (float*)malloc(sizeof(float) * 3 * N);
    float* dirs = (float*)malloc(sizeof(float) * 3 * N);
    float* x_c = (float*)malloc(sizeof(float) * 3 * N);
    uint8_t* tr = (uint8_t*)malloc(N);
    float* sigma = (float*)malloc(sizeof(float) * N);
    float* ess = (float*)malloc(sizeof(float) * 3 * N);
    float* g = (float*)malloc(sizeof(float) * 3 * N);
    float* nw = (float*)malloc(sizeof(float) * 3 * N);
    float* col = (float*)malloc(sizeof(float) * 3 * N);
    float* raw = raw_out ? raw_out : (float*)malloc(sizeof(float) * 4 * N);
    float pose16[16];
    orc_sample_gg(ray_o, ray_d, near, far, R, xyz, V, S, t_vals, jitter, z, pts);
    for (int64_t i = 0; i < N; ++i)
        for (int c = 0; c < 3; ++c) dirs[3 * i + c] = ray_d[3 * (i / S) + c];
    orc_warp(pts, NULL, N, xyz, canon, faces, F, NULL, NULL, NULL, tr, x_c, NULL);
    orc_pose_feat(poses24x3, P, NULL, pose16);
    orc_field(x_c, N, P, code8, pose16, sigma, ess, g);
    orc_normal_world(x_c, g, N, canon, xyz, faces, F, NULL, nw);
    orc_lighting(nw, pts, dirs, ess, N, P, rot4, rot_center2, light_shift3, col);
    for (int64_t i = 0; i < N; ++i) {
        raw[4 * i] = col[3 * i]; raw[4 * i + 1] = col[3 * i + 1]; raw[4 * i + 2] = col[3 * i + 2];
        raw[4 * i + 3] = tr[i] ? 0.f : sigma[i];
    }
    orc_composite(raw, z, ray_d, noise, R, S, rgb_map, disp_map, acc_map, weights, depth_map);
    if (!z_out) free(z);
    if (!raw_out) free(raw);
    free(pts); free(dirs); free(x_c); free(tr); free(sigma); free(ess); free(g); free(nw); free(col);
}

ORC_API int orc_param_count(void) { return P_COUNT; }
