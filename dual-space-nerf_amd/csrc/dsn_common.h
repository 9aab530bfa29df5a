// dsn_common.h - shared device helpers and the scene / packed-parameter layouts (gfx950 only).
//
// Arithmetic contract.  Every geometric quantity follows the reference's float32 operation order
// exactly (one rounding per torch op, fma only where the torch CPU kernel has one); the whole
// library is compiled with -ffp-contract=off so that `a*b+c` is never fused implicitly and fmaf()
// marks every intended fusion.  With IEEE add/mul/fma/div/sqrt this makes sampler, warp and normal
// stages bit-reproducible against the oracle (oracle/dsn_oracle.c) and the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DSN_WAVE 64
#define DSN_NUM_PARAMS_INTERNAL 33

// ---------------------------------------------------------------------------------------------
// scene blob (device memory owned by the caller, dsn_scene_bytes(V,F) bytes)
// ---------------------------------------------------------------------------------------------
struct DsnFaceRec {   // 16 floats = 64 B per face: everything utils/geo_utils.py:181-200,96-113,138-156 derive
    float m0[3];      // vertex 0
    float d00;        // dot(v20,v20)
    float v10[3];     // v1 - v0
    float d01;        // dot(v20,v10)
    float v20[3];     // v2 - v0
    float d11;        // dot(v10,v10)
    float n[3];       // normalize(cross(v10, v20))
    float inv;        // 1 / (d00*d11 - d01*d01)
};

struct DsnFrameState {   // small per-frame vectors
    float pose_feat[16];
    float code[8];
    float light_shift[3];
    float has_light;
    float rot[4];
    float rot_center[2];
    float has_rot;
    float pad[29];       // (keeps bias0 256-byte aligned inside the struct)
    float bias0[256];    // stage1.0 bias with the 24 constant input columns (code, pose) folded in
};

__host__ __device__ inline size_t dsn_align256(size_t x) { return (x + 255) & ~(size_t)255; }
#include "dsn_nn.h"

struct DsnSceneView {
    int V, F;
    float* canon;            // [V,3] copy
    int32_t* faces;          // [F,3] copy
    float* xyz;              // [V,3] copy of the posed vertices (sampler reads it)
    float4* cent_world;      // [F] xyz + bit-cast index
    float4* cent_canon;      // [F]
    DsnFaceRec* face_world;  // [F]
    DsnFaceRec* face_canon;  // [F]
    DsnFrameState* frame;
    DsnNNView nn_world;      // exact nearest-centroid lists of the posed mesh (rebuilt per frame)
    DsnNNView nn_canon;      // ... of the canonical mesh (built once)
};

__host__ __device__ inline DsnSceneView dsn_scene_view(void* base, int V, int F) {
    DsnSceneView s;
    char* p = (char*)base;
    // header: V, F stored for sanity checks
    s.V = V; s.F = F;
    p += 256;
    // the per-frame state comes FIRST: a "pose-only" blob (dsn_pose_state_bytes(): header + this struct, no body model)
    // is a valid prefix of a scene, so density-only / stand-alone network queries need no mesh (dsn_set_pose, dsn_field V=F=0)
    s.frame = (DsnFrameState*)p;    p += dsn_align256(sizeof(DsnFrameState));
    s.canon = (float*)p;            p += dsn_align256(sizeof(float) * 3 * (size_t)V);
    s.faces = (int32_t*)p;          p += dsn_align256(sizeof(int32_t) * 3 * (size_t)F);
    s.xyz = (float*)p;              p += dsn_align256(sizeof(float) * 3 * (size_t)V);
    s.cent_world = (float4*)p;      p += dsn_align256(sizeof(float4) * (size_t)F);
    s.cent_canon = (float4*)p;      p += dsn_align256(sizeof(float4) * (size_t)F);
    s.face_world = (DsnFaceRec*)p;  p += dsn_align256(sizeof(DsnFaceRec) * (size_t)F);
    s.face_canon = (DsnFaceRec*)p;  p += dsn_align256(sizeof(DsnFaceRec) * (size_t)F);
    s.nn_world = dsn_nn_view(p, F);
    s.nn_canon = dsn_nn_view(p, F);
    return s;
}
__host__ __device__ inline size_t dsn_pose_state_size() { return 256 + dsn_align256(sizeof(DsnFrameState)); }
__host__ __device__ inline size_t dsn_scene_size(int V, int F) {
    return 256 + 2 * dsn_align256(sizeof(float) * 3 * (size_t)V) + dsn_align256(sizeof(int32_t) * 3 * (size_t)F) +
           2 * dsn_align256(sizeof(float4) * (size_t)F) + 2 * dsn_align256(sizeof(DsnFaceRec) * (size_t)F) +
           dsn_align256(sizeof(DsnFrameState)) + 2 * dsn_nn_bytes(F);
}

// ---------------------------------------------------------------------------------------------
// float32 helpers with the reference's rounding sequence (see oracle/dsn_oracle.c helpers)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dsn_sum3(float a, float b, float c) { return (a + b) + c; }
__device__ __forceinline__ float dsn_dot3(const float* a, const float* b) {
    return dsn_sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]);
}
// torch.norm(dim=-1) on 3 floats: fma-accumulated sum of squares, then sqrt
__device__ __forceinline__ float dsn_norm3(const float* a) {
    return sqrtf(fmaf(a[2], a[2], fmaf(a[1], a[1], a[0] * a[0])));
}
// torch.cross: fma(a_i, b_j, -(a_j*b_i))
__device__ __forceinline__ void dsn_cross3(const float* a, const float* b, float* o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ float dsn_div(float a, float b) { return __fdiv_rn(a, b); }

// Branch-free sincos for the positional encoding (arguments x * 2^j, |x| of order 1 m, j < 10).
// Two-term Cody-Waite reduction by pi/2 with fma (error ~ |k| * 2e-15, k < 2^16 safe) + the classic
// minimax polynomials on [-pi/4, pi/4] (~1 ulp); no divergent large-argument path, so 30 inlined calls
// cost neither exec-mask SGPRs nor scratch (ocml's sincosf carries a Payne-Hanek branch per call).
__device__ __forceinline__ void dsn_sincos(float x, float& s, float& c) {
    const float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(kf, -1.57079637050628662109375f, x);
    r = fmaf(kf, 4.37113900018624283e-8f, r);
    const float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)kf;
    const float a = (q & 1) ? cr : sr;
    const float b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}



// per-face record: identical values to what the reference recomputes per point
__device__ __forceinline__ void dsn_make_face(const float* v0, const float* v1, const float* v2, DsnFaceRec& r) {
    float n[3];
    for (int c = 0; c < 3; ++c) { r.m0[c] = v0[c]; r.v10[c] = v1[c] - v0[c]; r.v20[c] = v2[c] - v0[c]; }
    dsn_cross3(r.v10, r.v20, n);
    float nn = dsn_norm3(n);
    for (int c = 0; c < 3; ++c) r.n[c] = dsn_div(n[c], nn);
    r.d00 = dsn_dot3(r.v20, r.v20);
    r.d01 = dsn_dot3(r.v20, r.v10);
    r.d11 = dsn_dot3(r.v10, r.v10);
    r.inv = dsn_div(1.0f, r.d00 * r.d11 - r.d01 * r.d01);
}

// utils/geo_utils.py:181-200 project_point2mesh + :96-113 get_barycentric_coordinates
__device__ __forceinline__ void dsn_project(const float* p, const DsnFaceRec& f, float& u, float& v, float& h) {
    float tmp[3], q[3], w[3];
    for (int c = 0; c < 3; ++c) tmp[c] = p[c] - f.m0[c];
    float sd = dsn_dot3(tmp, f.n);
    for (int c = 0; c < 3; ++c) q[c] = p[c] - f.n[c] * sd;
    for (int c = 0; c < 3; ++c) w[c] = q[c] - f.m0[c];
    float d02 = dsn_dot3(f.v20, w), d12 = dsn_dot3(f.v10, w);
    u = (f.d11 * d02 - f.d01 * d12) * f.inv;
    v = (f.d00 * d12 - f.d01 * d02) * f.inv;
    h = sd;
}
// utils/geo_utils.py:138-156 barycentric_map2can
__device__ __forceinline__ void dsn_map2face(float u, float v, float h, const DsnFaceRec& f, float* out) {
    for (int c = 0; c < 3; ++c) {
        float proj = (f.m0[c] + u * f.v20[c]) + v * f.v10[c];
        out[c] = proj + h * f.n[c];
    }
}
// F.normalize(x, dim=-1, eps=1e-12)
__device__ __forceinline__ void dsn_normalize3(const float* a, float* o) {
    float n = dsn_norm3(a);
    n = n < 1e-12f ? 1e-12f : n;
    for (int c = 0; c < 3; ++c) o[c] = dsn_div(a[c], n);
}
__device__ __forceinline__ DsnFaceRec dsn_load_face(const DsnFaceRec* __restrict__ recs, int f) {
    const float4* p = (const float4*)(recs + f);
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    DsnFaceRec r;
    r.m0[0] = a.x; r.m0[1] = a.y; r.m0[2] = a.z; r.d00 = a.w;
    r.v10[0] = b.x; r.v10[1] = b.y; r.v10[2] = b.z; r.d01 = b.w;
    r.v20[0] = c.x; r.v20[1] = c.y; r.v20[2] = c.z; r.d11 = c.w;
    r.n[0] = d.x; r.n[1] = d.y; r.n[2] = d.z; r.inv = d.w;
    return r;
}

// ---------------------------------------------------------------------------------------------
// packed network parameters (device blob, dsn_packed_param_bytes())
// MFMA-ordered images of every matrix; float offsets.  See dsn_field.hip for the lane mapping.
// ---------------------------------------------------------------------------------------------
// A-operand image of one [32 out rows] x [32 k] block: 16 k-steps x 64 lanes = 1024 floats, stored
// [r4][lane][4] so that one 16-byte load per lane fetches 4 consecutive k-steps.
#define DSN_BLK 1024

enum {
    // forward images (M = out features)
    OFF_L0   = 0,                          // stage1.0 : 8 m x 2 kb(pe)     (code/pose columns folded into bias0)
    OFF_L1   = OFF_L0 + 8 * 2 * DSN_BLK,   // stage1.2 : 8 x 8
    OFF_L2   = OFF_L1 + 8 * 8 * DSN_BLK,
    OFF_L3   = OFF_L2 + 8 * 8 * DSN_BLK,
    OFF_L4   = OFF_L3 + 8 * 8 * DSN_BLK,   // stage2.0 : 8 x 10 (8 h + 2 pe)
    OFF_L5   = OFF_L4 + 8 * 10 * DSN_BLK,
    OFF_L6   = OFF_L5 + 8 * 8 * DSN_BLK,
    OFF_RGB1 = OFF_L6 + 8 * 8 * DSN_BLK,   // rgb_net.1 : 4 x 8
    // transposed images (M = in features) for d sigma / d x
    OFF_L6T  = OFF_RGB1 + 4 * 8 * DSN_BLK, // 8 x 8
    OFF_L5T  = OFF_L6T + 8 * 8 * DSN_BLK,
    OFF_L4T  = OFF_L5T + 8 * 8 * DSN_BLK,  // 10 m (8 h + 2 pe) x 8
    OFF_L3T  = OFF_L4T + 10 * 8 * DSN_BLK,
    OFF_L2T  = OFF_L3T + 8 * 8 * DSN_BLK,
    OFF_L1T  = OFF_L2T + 8 * 8 * DSN_BLK,
    OFF_L0T  = OFF_L1T + 8 * 8 * DSN_BLK,  // 2 m (pe) x 8
    // lighting MLP
    OFF_LT0  = OFF_L0T + 2 * 8 * DSN_BLK,  // lights_encoding.0 : 4 m x 1 kb (5 k-steps used, rest zero)
    OFF_LT1  = OFF_LT0 + 4 * 1 * DSN_BLK,  // lights_encoding.2 : 4 x 4
    // vectors in accumulator (C-layout) order: [m][half][16] floats per 32-row block
    OFF_B1   = OFF_LT1 + 4 * 4 * DSN_BLK,  // biases of stage1.2/4/6, stage2.0/2/4 : 6 x 256
    OFF_BRGB1 = OFF_B1 + 6 * 256,          // 128
    OFF_WDEN = OFF_BRGB1 + 128,            // density_net weight in C-layout order : 256
    OFF_WRGB3 = OFF_WDEN + 256,            // rgb_net.3 weight, 3 x 128 in C-layout order
    OFF_BLT0 = OFF_WRGB3 + 3 * 128,        // lighting biases 128, 128
    OFF_BLT1 = OFF_BLT0 + 128,
    OFF_WLT2 = OFF_BLT1 + 128,             // lights_encoding.4 weight in C-layout order : 128
    OFF_SCAL = OFF_WLT2 + 128,             // [0]=density bias, [1..3]=rgb3 bias, [4]=lights_encoding.4 bias,
                                           // [5]=margin of the density screen (default / dsn_calibrate_screen, k_screen16)
                                           // [6]=colour scale of the early-stop threshold (1 / dsn_set_early_stop_colour_scale)
    // raw (unpacked) copies used by the per-frame setup kernel
    OFF_RAW_W0 = OFF_SCAL + 64,            // stage1.0.weight [256,87]
    OFF_RAW_B0 = OFF_RAW_W0 + 256 * 87,    // stage1.0.bias [256]
    OFF_RAW_EMB = OFF_RAW_B0 + 256,        // embedding [500,8]
    OFF_RAW_PM0W = OFF_RAW_EMB + 500 * 8,  // pose_mlp
    OFF_RAW_PM0B = OFF_RAW_PM0W + 64 * 92,
    OFF_RAW_PM2W = OFF_RAW_PM0B + 64,
    OFF_RAW_PM2B = OFF_RAW_PM2W + 64 * 64,
    OFF_RAW_PM4W = OFF_RAW_PM2B + 64,
    OFF_RAW_PM4B = OFF_RAW_PM4W + 16 * 64,
    OFF_END32 = OFF_RAW_PM4B + 16,
    // ---- split-fp16 images of the same 872-block stream (csrc/dsn_field16.hip): 4 KB per block, stored as
    // [t(2)][part(hi,lo)][lane(64)][8 halves]; block b of the fp32 stream <-> block b here.
    OFF16_BASE = (OFF_END32 + 63) & ~63,          // float offset, 256-byte aligned
    DSN_STREAM_BLOCKS = OFF_LT0 / DSN_BLK,        // 872: stage1.0 ... stage1.0^T in consumption order (k_field16)
    DSN_STREAM_BLOCKS_ALL = OFF_B1 / DSN_BLK,     // 892: + lights_encoding.0 (4) and .2 (16) for k_light16
    OFF_END = OFF16_BASE + DSN_STREAM_BLOCKS_ALL * DSN_BLK
};
// Position (in halfwords) of halfword hw (0..2047) of split-fp16 stream block gb.  The 872 trunk blocks are stored
// chunk-major: a chunk = 8 consecutive blocks = 32 KB, laid out [quarter q = hw >> 9 (hi/lo of k-step 0, hi/lo of k-step 1)]
// [block in chunk][512 halfwords], so that the 8 one-KB pieces one wave moves per chunk are contiguous - in memory AND in the
// LDS ring - and one LDS-DMA base + 8 immediate offsets cover them (k_field16).  The lighting blocks behind them stay block-major.
__host__ __device__ inline size_t dsn_stream16_index(int gb, int hw) {
    if (gb >= DSN_STREAM_BLOCKS) return (size_t)gb * 2048 + hw;
    return (size_t)(gb >> 3) * 16384 + (size_t)(hw >> 9) * 4096 + (size_t)(gb & 7) * 512 + (hw & 511);
}
// DSN_EARLY_STOP threshold: the frame moves by at most (S + 1) eps x max|colour| (S unshaded samples of weight < eps each + a
// terminated tail of total weight < eps), so eps shrinks with S AND with the colour scale c of the loaded parameters
// (packed[OFF_SCAL + 6], dsn_set_early_stop_colour_scale; 1 until measured): eps = min(2^-20, 1e-4 / (2 (S + 1) max(1, c))) - half of
// the 1e-4 parity bar, absolute, as long as the colours stay below c (colour = (ELU + 1) x essence is unbounded)
#define DSN_STOP_COLOUR_HEADROOM 2.0f      // colour scale handed to the threshold = this x the largest colour seen (dsn_early_stop_colour_headroom)
__host__ __device__ inline float dsn_stop_eps_scaled(int S, float colour_scale) {
    const float cap = 9.5367431640625e-07f;      // 2^-20
    const float c = colour_scale > 1.0f ? colour_scale : 1.0f;
    const float e = 1e-4f / (2.0f * (float)(S + 1) * c);
    return e < cap ? e : cap;
}
// DSN_OWN_SIMD(): the wave's register allocation covers the WHOLE 512-entry register file of its SIMD, so that no wave of another kernel
// can be resident beside it.
// THE RULE (round 6): a kernel that issues v_mfma_f32_32x32x16_f16 - gfx950's K = 16 form of the f16 MFMA - must own its SIMDs.
// Why.  Round 5 (docs/LAB_NOTEBOOK.md "frames in flight were not bit-identical"): with several frames in flight, waves of the small
// kernels of one frame (k_normal, k_nns_search, ...: 24-64 registers) that landed on a SIMD beside a wave of the split-fp16 field kernels
// of another frame consumed registers BEFORE their own global loads had landed - 0.2-1 % of a frame's samples came out with a slightly
// different canonical point or normal, never the same ones twice.  The aggressors themselves are unaffected and stay within their
// allocation (ISA checked).  Round 5 found the condition (co-residency) and this guard; round 6 found the trigger
// (scripts/dbg/race_train.py + race_bisect.sh, profiles/r06_coresidency_bisect.txt): builds that leave ONE kernel unguarded beside 14
// frames' shading / geometry phases.  Aggressors: k_field16 (all modes), k_light16, k_tangent16, k_adjoint16, k_t_wgrad16q, k_t_wgrad16p
// (weak: 1 repetition in 12) - every kernel that issues the K = 16 f16 MFMA.  Clean: k_t_lin, k_t_wgrad (fp32 MFMA 32x32x2), whatever
// their register and AGPR use.  Not the trigger: negative immediate offsets of the weight ring's LDS-DMA, the v_fma_mix / v_bfe inline
// asm, accumulators in AGPRs vs VGPRs (each exchanged alone: still an aggressor).  THE trigger: k_t_wgrad16q with every
// v_mfma_f32_32x32x16_f16 replaced by two v_mfma_f32_32x32x8_f16 (same registers, same LDS-DMA, same occupancy, twice the matrix-pipe
// time): 0 differing samples in 12 repetitions, against 9 of 12 with the K = 16 instruction.
// Consequences: every K = 16 kernel carries the guard (below; tests/test_guard_coverage.py enforces it in source); k_t_wgrad16p, which
// is bound by its operand stream, runs on the K = 8 instruction instead and keeps two workgroups per CU; fp32-MFMA kernels need no guard
// (k_field / k_light keep theirs: 480 registers / the fallback path).  Costs the co-residency of other frames' small kernels on the
// compute units the persistent field workgroups occupy (they keep the eighth DSN_SHARE_CUS leaves them): ~0.15 ms per frame.
// Two dummy register writes at kernel entry (v255 AND a255: 256 architectural + 256 accumulation registers whatever the kernel itself
// needs - a255 alone sits behind the kernel's own VGPR count, rounded to 4).
#define DSN_OWN_SIMD() asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255")
// The training backward's matrix kernels carry the guard through DSN_OWN_SIMD_T(bit): 1 k_tangent16, 2 k_adjoint16, 4 k_t_wgrad16d
// (256 + 256 registers as it is), 64 k_t_wgrad16c, 128 k_t_wgrad16q - the K = 16 kernels; 16 k_t_lin, 32 k_t_wgrad (fp32 MFMA) do not.
// Experiment builds (-DDSN_EXPERIMENTS -DDSN_TRAIN_UNGUARDED=bits) leave the kernels of the given bits without the guard and guard all
// the others - the bisect's tool.
#define DSN_TRAIN_AGGRESSORS 199
#if defined(DSN_EXPERIMENTS) && defined(DSN_TRAIN_UNGUARDED)
#define DSN_OWN_SIMD_T(bit) do { if (!((DSN_TRAIN_UNGUARDED) & (bit))) DSN_OWN_SIMD(); } while (0)
#else
#define DSN_OWN_SIMD_T(bit) do { if ((DSN_TRAIN_AGGRESSORS) & (bit)) DSN_OWN_SIMD(); } while (0)
#endif
#define DSN_SCREEN_MARGIN_DEFAULT 0.01f          // conservative margin of the density screen until it has been calibrated
#define DSN_LO_SCALE 4096.0f                      // lo = (x - hi) * 2^12, products accumulated apart, folded at the end
#define DSN_LO_INV (1.0f / 4096.0f)

// accumulator (C/D) layout of v_mfma_f32_32x32x2_f32: lane l, register r holds
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31
__host__ __device__ inline int dsn_crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// positional-encoding k-steps: 32 steps, (step t, half) -> index into the 63-vector of
// model/dimension_kernel.py:34-35, or -1 for the single pad slot.
__host__ __device__ inline int dsn_pe_index(int t, int half) {
    if (t < 30) { int j = t / 3, a = t % 3; return 3 + 6 * j + (half ? 3 : 0) + a; }
    if (t == 30) return half ? 1 : 0;
    return half ? -1 : 2;
}
