// dsn_api.hip - the C ABI of libdsnerf_hip.so (include/dsnerf.h): argument checks, scene/workspace
// carving and kernel launches on the caller's stream.  No allocation, no synchronisation, no CPU
// fallback: every entry point only enqueues gfx950 kernels.
#include "../../include/dsnerf.h"
#include "dsn_common.h"
#include "dsn_kernels.h"
#include <cstdlib>

#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int dsn_fail(const char* fmt, const char* a = "", long long b = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return 1;
}
static int dsn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e)); return 1; }
    return 0;
}
#define DSN_REQUIRE(cond, msg) do { if (!(cond)) return dsn_fail("%s", msg); } while (0)

thread_local int g_dsn_persistent_override = 0;
namespace {
struct DsnShareCus {      // DSN_SHARE_CUS for the duration of one dsn_render_rays call
    explicit DsnShareCus(bool on) {
        // (DSN_SHARE_EIGHTHS, experiments: how many eighths of the compute units the persistent kernels take; 7 by default)
        static const int eighths = [] { const char* e = getenv("DSN_SHARE_EIGHTHS"); const int v = e ? atoi(e) : 7; return v >= 1 && v <= 8 ? v : 7; }();
        if (on) { const int n = dsn_cu_count_raw(); g_dsn_persistent_override = std::max(8, (n * eighths / 8) / 8 * 8); }
    }
    ~DsnShareCus() { g_dsn_persistent_override = 0; }
};
}

struct DsnSceneHeader { int magic, V, F, has_body, has_frame; };
#define DSN_MAGIC 0x44534e31

extern "C" {

int dsn_abi_version(void) { return DSN_ABI_VERSION; }
const char* dsn_last_error(void) { return g_err; }

size_t dsn_packed_param_bytes(void) { return sizeof(float) * (size_t)OFF_END; }

int dsn_pack_params(const float* const* params33_host, void* packed, void* stream) {
    DSN_REQUIRE(params33_host && packed, "dsn_pack_params: null argument");
    for (int i = 0; i < DSN_NUM_PARAMS; ++i) DSN_REQUIRE(params33_host[i], "dsn_pack_params: null parameter pointer");
    dsn_launch_pack_params(params33_host, (float*)packed, (hipStream_t)stream);
    dsn_launch_set_screen_margin((float*)packed, DSN_SCREEN_MARGIN_DEFAULT, (hipStream_t)stream);   // until dsn_calibrate_screen has run
    dsn_launch_set_packed_scalar((float*)packed, 6, 1.0f, (hipStream_t)stream);      // colour scale of the early-stop threshold until measured
    return dsn_check_launch("dsn_pack_params");
}

// layout utility (no device work): the packed image of HOST parameter arrays, for tests / tooling
int dsn_pack_params_host_image(const float* const* params33_host, float* packed_host) {
    DSN_REQUIRE(params33_host && packed_host, "dsn_pack_params_host_image: null argument");
    dsn_pack_params_host(params33_host, packed_host);
    packed_host[OFF_SCAL + 5] = DSN_SCREEN_MARGIN_DEFAULT;
    packed_host[OFF_SCAL + 6] = 1.0f;
    return 0;
}

size_t dsn_scene_bytes(int V, int F) { return (V > 0 && F > 0) ? dsn_scene_size(V, F) : 0; }

int dsn_set_body(void* scene, const float* canon_vertex, const int32_t* faces, int V, int F, void* stream) {
    DSN_REQUIRE(scene && canon_vertex && faces && V > 0 && F > 0, "dsn_set_body: bad argument");
    hipStream_t st = (hipStream_t)stream;
    DsnSceneView s = dsn_scene_view(scene, V, F);
    DsnSceneHeader h = {DSN_MAGIC, V, F, 1, 0};
    // header lives in device memory too (sanity for later calls is done with the V/F the caller passes)
    if (hipMemcpyAsync(scene, &h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) return dsn_fail("%s", "dsn_set_body: header copy failed");
    if (hipMemcpyAsync(s.canon, canon_vertex, sizeof(float) * 3 * (size_t)V, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(s.faces, faces, sizeof(int32_t) * 3 * (size_t)F, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return dsn_fail("%s", "dsn_set_body: copy failed");
    dsn_launch_face_setup(s.canon, s.faces, F, s.face_canon, s.cent_canon, st);
    // canonical-space queries are x_c = face frame re-embedding with |h| <= 0.1 and uv in [-4,5]: wide pads
    // (fine level: pad 0.12 m - a non-transparent sample lies within |h| <= 0.1 m of its face, utils/render_utils.py:103-109, so its
    //  canonical point is inside; whatever is further out takes the coarse level / the sweep, same index)
    dsn_launch_build_nn(s.cent_canon, F, s.nn_canon, 0.12f, 0.7f, st, false, true);
    return dsn_check_launch("dsn_set_body");
}

int dsn_set_frame(void* scene, int V, int F, const void* packed, const float* xyz, const float* poses24x3, int frame_idx,
                  int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2,
                  void* stream) {
    return dsn_set_frame_ex(scene, V, F, packed, xyz, poses24x3, frame_idx, zero_code, light_shift3, rot2x2, rot_center2, 0, stream);
}

int dsn_set_frame_ex(void* scene, int V, int F, const void* packed, const float* xyz, const float* poses24x3, int frame_idx,
                     int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2, int flags,
                     void* stream) {
    DSN_REQUIRE(scene && packed && xyz && poses24x3, "dsn_set_frame_ex: null argument");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_set_frame_ex: bad V/F");
    DSN_REQUIRE(frame_idx >= 0 && frame_idx < 500, "dsn_set_frame_ex: frame index outside the embedding table (maxFrame=500)");
    hipStream_t st = (hipStream_t)stream;
    DsnSceneView s = dsn_scene_view(scene, V, F);
    if (hipMemcpyAsync(s.xyz, xyz, sizeof(float) * 3 * (size_t)V, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return dsn_fail("%s", "dsn_set_frame_ex: copy failed");
    dsn_launch_face_setup(s.xyz, s.faces, F, s.face_world, s.cent_world, st);
    // world-space queries lie inside the (padded) body AABB the rays were clipped to; coarse level beyond
    dsn_launch_build_nn(s.cent_world, F, s.nn_world, 0.12f, 0.7f, st, (flags & DSN_FRAME_FINE_ONLY) != 0, false,
                        (flags & DSN_FRAME_LAZY_LISTS) != 0);
    dsn_launch_pose_setup((const float*)packed, poses24x3, frame_idx, zero_code, light_shift3, rot2x2, rot_center2,
                          s.frame, st);
    return dsn_check_launch("dsn_set_frame_ex");
}

size_t dsn_pose_state_bytes(void) { return dsn_pose_state_size(); }

int dsn_set_pose(void* scene_or_pose_state, const void* packed, const float* poses24x3, const float* pose_feat16, int frame_idx,
                 int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2, void* stream) {
    DSN_REQUIRE(scene_or_pose_state && packed, "dsn_set_pose: null argument");
    DSN_REQUIRE(poses24x3 || pose_feat16, "dsn_set_pose: give poses (batch['poses'], pose_mlp is applied) or the 16 pose features");
    DSN_REQUIRE(frame_idx >= 0 && frame_idx < 500, "dsn_set_pose: frame index outside the embedding table (maxFrame=500)");
    // the per-frame state sits at the same offset in a scene blob and in a pose-only blob (dsn_common.h dsn_scene_view)
    DsnFrameState* fs = (DsnFrameState*)((char*)scene_or_pose_state + 256);
    dsn_launch_pose_setup((const float*)packed, poses24x3, frame_idx, zero_code, light_shift3, rot2x2, rot_center2, fs,
                          (hipStream_t)stream, pose_feat16);
    return dsn_check_launch("dsn_set_pose");
}

int dsn_sample_gg(const void* scene, int V, int F, const float* ray_o, const float* ray_d, float* near, float* far, int R, int S,
                  const float* t_vals, const float* jitter, float* z_vals, float* pts, void* stream) {
    DSN_REQUIRE(scene && ray_o && ray_d && near && far && t_vals && z_vals, "dsn_sample_gg: null argument");
    DSN_REQUIRE(R > 0 && S > 0, "dsn_sample_gg: empty ray batch");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_sample_gg: bad V/F");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_sample_gg(s.xyz, V, ray_o, ray_d, near, far, R, S, t_vals, jitter, z_vals, pts, (hipStream_t)stream);
    return dsn_check_launch("dsn_sample_gg");
}

// utils/pts_utils.py:3-16 uniform_sampling alone (sample_points_mode == "uniform", can_render.py:42-51)
int dsn_sample_uniform(const void* scene, int V, int F, const float* ray_o, const float* ray_d, float* near, float* far,
                       int R, int S, const float* t_vals, const float* jitter, float* z_vals, float* pts, void* stream) {
    DSN_REQUIRE(ray_o && ray_d && near && far && t_vals && z_vals, "dsn_sample_uniform: null argument");
    DSN_REQUIRE(R > 0 && S > 0, "dsn_sample_uniform: empty ray batch");
    (void)scene; (void)V; (void)F;
    dsn_launch_sample_gg(nullptr, 0, ray_o, ray_d, near, far, R, S, t_vals, jitter, z_vals, pts, (hipStream_t)stream);
    return dsn_check_launch("dsn_sample_uniform");
}

int dsn_warp(const void* scene, int V, int F, const float* pts, const float* ray_d, int64_t N, int S, int32_t* face_idx, float* uv,
             float* h, uint8_t* transparent, float* x_c, float* ray_d_can, int32_t* active_list,
             int32_t* active_count, int flags, void* stream) {
    DSN_REQUIRE(scene && pts, "dsn_warp: null argument");
    DSN_REQUIRE(N > 0 && S > 0, "dsn_warp: empty point batch");
    DSN_REQUIRE((active_list == nullptr) == (active_count == nullptr), "dsn_warp: active_list and active_count go together");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_warp: bad V/F");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_warp(s, pts, nullptr, ray_d, nullptr, N, S, face_idx, uv, h, transparent, x_c, ray_d_can, active_list,
                    active_count, (flags & DSN_NN_EXHAUSTIVE) != 0, (hipStream_t)stream);
    return dsn_check_launch("dsn_warp");
}

int dsn_lbs_warp(const void* scene, int V, int F, const float* pts, int64_t N, const float* smpl_weights,
                 const float* joint_transforms, int bw_type, int32_t* face_idx, float* weights, uint8_t* transparent,
                 float* pts_zero, int flags, void* stream) {
    DSN_REQUIRE(scene && pts && smpl_weights && joint_transforms, "dsn_lbs_warp: null argument");
    DSN_REQUIRE(N > 0 && V > 0 && F > 0, "dsn_lbs_warp: bad sizes");
    DSN_REQUIRE(bw_type == 0 || bw_type == 1, "dsn_lbs_warp: unsupport value: bw_type");   // the reference raises ValueError
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_lbs_warp(s, pts, N, smpl_weights, joint_transforms, bw_type, face_idx, weights, transparent, pts_zero,
                        (flags & DSN_NN_EXHAUSTIVE) != 0, (hipStream_t)stream);
    return dsn_check_launch("dsn_lbs_warp");
}

int dsn_field(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N, const int32_t* active_list,
              const int32_t* active_count, float* sigma, float* essence, float* grad, int flags, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && sigma, "dsn_field: null argument");
    DSN_REQUIRE(N > 0, "dsn_field: empty point batch");
    DSN_REQUIRE((active_list == nullptr) == (active_count == nullptr), "dsn_field: active_list and active_count go together");
    DSN_REQUIRE((V > 0 && F > 0) || (V == 0 && F == 0), "dsn_field: bad V/F (0/0 = pose-only state of dsn_pose_state_bytes())");
    DsnSceneView s = dsn_scene_view((void*)scene, V > 0 ? V : 1, F > 0 ? F : 1);     // only s.frame is used (fixed offset)
    // split-fp16 kernel for the full evaluation; the exact-fp32 kernel serves density-only / colour-only queries
    if (!(flags & DSN_FIELD_FP32) && essence && grad) {
        dsn_launch_field16((const float*)packed, s.frame, x_c, N, active_list, active_count, sigma, essence, grad,
                           (hipStream_t)stream);
        // range fallback: samples the split-fp16 kernel flagged (sigma = NaN) are re-evaluated in exact fp32
        dsn_launch_field_fix((const float*)packed, s.frame, x_c, N, active_list, active_count, sigma, essence, grad,
                             (hipStream_t)stream);
    } else
        dsn_launch_field((const float*)packed, s.frame, x_c, N, active_list, active_count, sigma, essence, grad,
                         (hipStream_t)stream);
    return dsn_check_launch("dsn_field");
}

size_t dsn_field_record_bytes(int64_t N) { return N > 0 ? dsn_align256(224 * (size_t)N) : 0; }

int dsn_field_forward(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N,
                      const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence, void* records,
                      int32_t* pos_list, int32_t* pos_count, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && sigma && essence && records && pos_list && pos_count, "dsn_field_forward: null argument");
    DSN_REQUIRE(N > 0, "dsn_field_forward: empty point batch");
    DSN_REQUIRE((active_list == nullptr) == (active_count == nullptr), "dsn_field_forward: active_list and active_count go together");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_field_forward: bad V/F");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_field16_fwd((const float*)packed, s.frame, x_c, N, active_list, active_count, sigma, essence, records, pos_list,
                           pos_count, (hipStream_t)stream, N);
    return dsn_check_launch("dsn_field_forward");
}

int dsn_field_reverse(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N, const int32_t* pos_list,
                      const int32_t* pos_count, const void* records, float* grad, float* sigma, float* essence, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && pos_list && pos_count && records && grad && sigma && essence, "dsn_field_reverse: null argument");
    DSN_REQUIRE(N > 0, "dsn_field_reverse: empty point batch");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_field_reverse: bad V/F");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_field16_bwd((const float*)packed, s.frame, x_c, N, pos_list, pos_count, grad, records, (hipStream_t)stream, sigma, N);
    // range fallback for what dsn_field_forward or the reverse pass flagged (sigma = NaN): exact fp32, all three outputs
    dsn_launch_field_fix((const float*)packed, s.frame, x_c, N, pos_list, pos_count, sigma, essence, grad, (hipStream_t)stream);
    return dsn_check_launch("dsn_field_reverse");
}

// model/spacenet.py:174-188 LightingMLP.forward as a pure function of its four arguments
int dsn_light(const void* packed, const float* normal, const float* xyz_world, const float* view_dir_world,
              const float* essence, int64_t N, float* colour, void* zero_pose_state, int flags, void* stream) {
    DSN_REQUIRE(packed && normal && xyz_world && view_dir_world && essence && colour && zero_pose_state, "dsn_light: null argument");
    DSN_REQUIRE(N > 0, "dsn_light: empty point batch");
    hipStream_t st = (hipStream_t)stream;
    // the lighting kernels read the light-centre / rotation edits from a frame state; those belong to DualSpaceNeRF.forward
    // (:254-263), not to this function: run on a zeroed state
    if (hipMemsetAsync(zero_pose_state, 0, dsn_pose_state_size(), st) != hipSuccess) return dsn_fail("%s", "dsn_light: memset failed");
    const DsnFrameState* fs = (const DsnFrameState*)((const char*)zero_pose_state + 256);
    if (flags & DSN_FIELD_FP32)
        dsn_launch_light((const float*)packed, fs, normal, xyz_world, nullptr, view_dir_world, nullptr, essence, N, 1, nullptr, nullptr,
                         colour, st);
    else
        dsn_launch_light16((const float*)packed, fs, normal, xyz_world, nullptr, view_dir_world, nullptr, essence, N, 1, nullptr,
                           nullptr, colour, st);
    return dsn_check_launch("dsn_light");
}

size_t dsn_calibrate_workspace_bytes(int64_t n_points) { return n_points > 0 ? dsn_calibrate_workspace_size(n_points) : 0; }

int dsn_calibrate_screen(const void* scene, int V, int F, void* packed, int64_t n_points, void* workspace, float* out4, void* stream) {
    DSN_REQUIRE(scene && packed && workspace, "dsn_calibrate_screen: null argument");
    DSN_REQUIRE(V > 0 && F > 0 && n_points > 0, "dsn_calibrate_screen: bad sizes");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_calibrate_screen(s, (float*)packed, n_points, workspace, out4, (hipStream_t)stream);
    return dsn_check_launch("dsn_calibrate_screen");
}

int dsn_set_screen_margin(void* packed, float margin, void* stream) {
    DSN_REQUIRE(packed, "dsn_set_screen_margin: null argument");
    DSN_REQUIRE(margin == margin, "dsn_set_screen_margin: NaN margin");     // (+inf = keep every sample; <= 0 is unsafe, tests only)
    dsn_launch_set_screen_margin((float*)packed, margin, (hipStream_t)stream);
    return dsn_check_launch("dsn_set_screen_margin");
}

int dsn_set_early_stop_colour_scale(void* packed, float colour_scale, void* stream) {
    DSN_REQUIRE(packed, "dsn_set_early_stop_colour_scale: null argument");
    DSN_REQUIRE(colour_scale == colour_scale, "dsn_set_early_stop_colour_scale: NaN scale");
    dsn_launch_set_packed_scalar((float*)packed, 6, colour_scale, (hipStream_t)stream);
    return dsn_check_launch("dsn_set_early_stop_colour_scale");
}

int dsn_shade(const void* scene, int V, int F, const void* packed, const float* x_c, const float* grad, const float* x_w,
              const float* ray_d, const float* essence, int64_t N, int S, const int32_t* active_list,
              const int32_t* active_count, int32_t* face_idx_canon, float* n_w, float* colour, int flags, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && grad && x_w && ray_d && essence && n_w && colour, "dsn_shade: null argument");
    DSN_REQUIRE(N > 0 && S > 0, "dsn_shade: empty point batch");
    DSN_REQUIRE((active_list == nullptr) == (active_count == nullptr), "dsn_shade: active_list and active_count go together");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_shade: bad V/F");
    hipStream_t st = (hipStream_t)stream;
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_normal(s, x_c, grad, N, active_list, active_count, face_idx_canon, n_w, (flags & DSN_NN_EXHAUSTIVE) != 0, st);
    if (flags & DSN_FIELD_FP32)
        dsn_launch_light((const float*)packed, s.frame, n_w, x_w, nullptr, ray_d, nullptr, essence, N, S, active_list,
                         active_count, colour, st);
    else
        dsn_launch_light16((const float*)packed, s.frame, n_w, x_w, nullptr, ray_d, nullptr, essence, N, S, active_list,
                           active_count, colour, st);
    return dsn_check_launch("dsn_shade");
}

int dsn_composite(const float* colour, const float* sigma, const uint8_t* transparent, const float* z_vals,
                  const float* ray_d, const float* noise, int R, int S, float* rgb_map, float* disp_map,
                  float* acc_map, float* weights, float* depth_map, void* stream) {
    DSN_REQUIRE(colour && sigma && z_vals && ray_d && rgb_map && disp_map && acc_map && depth_map, "dsn_composite: null argument");
    DSN_REQUIRE(R > 0 && S > 0, "dsn_composite: empty ray batch");
    dsn_launch_composite(colour, sigma, transparent, z_vals, ray_d, noise, R, S, rgb_map, disp_map, acc_map, weights,
                         depth_map, (hipStream_t)stream);
    return dsn_check_launch("dsn_composite");
}

size_t dsn_image_workspace_bytes(int H, int W) { return (H > 0 && W > 0) ? dsn_image_workspace_size(H, W) : 0; }

int dsn_image_scatter(const float* rgb, const float* disp, const float* acc, const float* depth, int R, const uint8_t* mask_at_box,
                      int H, int W, int clamp_rgb, float* img_rgb, float* img_disp, float* img_acc, float* img_depth,
                      void* workspace, void* stream) {
    DSN_REQUIRE((rgb || R == 0) && mask_at_box && img_rgb && workspace, "dsn_image_scatter: null argument");
    DSN_REQUIRE(H > 0 && W > 0 && R >= 0, "dsn_image_scatter: bad sizes");
    DSN_REQUIRE((int64_t)H * W < ((int64_t)1 << 31), "dsn_image_scatter: image too large");
    DSN_REQUIRE(R == 0 || ((!img_disp || disp) && (!img_acc || acc) && (!img_depth || depth)), "dsn_image_scatter: image requested without its source");
    dsn_launch_image_scatter(rgb, disp, acc, depth, R, mask_at_box, H, W, clamp_rgb, img_rgb, img_disp, img_acc, img_depth,
                             workspace, (hipStream_t)stream);
    return dsn_check_launch("dsn_image_scatter");
}

int dsn_image_psnr(const float* img_rgb, const double* gt_f64, const float* gt_f32, const uint8_t* mask_at_box, int H, int W,
                   double* out4, void* workspace, void* stream) {
    DSN_REQUIRE(img_rgb && out4 && workspace, "dsn_image_psnr: null argument");
    DSN_REQUIRE((gt_f64 != nullptr) != (gt_f32 != nullptr), "dsn_image_psnr: exactly one ground-truth pointer");
    DSN_REQUIRE(H > 0 && W > 0, "dsn_image_psnr: bad sizes");
    dsn_launch_image_psnr(img_rgb, gt_f64, gt_f32, mask_at_box, H, W, out4, workspace, (hipStream_t)stream);
    return dsn_check_launch("dsn_image_psnr");
}

int dsn_field_screen(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N, const int32_t* active_list,
                     const int32_t* active_count, float* sigma, int32_t* keep_list, int32_t* keep_count, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && sigma && keep_list && keep_count, "dsn_field_screen: null argument");
    DSN_REQUIRE(N > 0 && V > 0 && F > 0, "dsn_field_screen: bad sizes");
    DSN_REQUIRE((active_list == nullptr) == (active_count == nullptr), "dsn_field_screen: active_list and active_count go together");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    dsn_launch_screen16((const float*)packed, s.frame, x_c, N, active_list, active_count, sigma, keep_list, keep_count, nullptr,
                        nullptr, (hipStream_t)stream);
    return dsn_check_launch("dsn_field_screen");
}

int dsn_debug_screen(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N, float* sigma_screen,
                     float* s1, int32_t* scratch_list, int32_t* scratch_count, void* stream) {
    DSN_REQUIRE(scene && packed && x_c && sigma_screen && s1 && scratch_list && scratch_count, "dsn_debug_screen: null argument");
    DSN_REQUIRE(N > 0 && V > 0 && F > 0, "dsn_debug_screen: bad sizes");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    if (hipMemsetAsync(scratch_count, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return dsn_fail("%s", "dsn_debug_screen: memset failed");
    dsn_launch_screen16((const float*)packed, s.frame, x_c, N, nullptr, nullptr, sigma_screen, scratch_list, scratch_count,
                        sigma_screen, s1, (hipStream_t)stream);
    return dsn_check_launch("dsn_debug_screen");
}

size_t dsn_grad_workspace_bytes(int R, int S) { return (R > 0 && S > 0) ? dsn_train_workspace_size((int64_t)R * S) : 0; }

int dsn_render_rays_grad(const void* scene, int V, int F, const void* packed, const float* const* params33_host, const float* poses24x3,
                         int frame_idx, int zero_code, const float* ray_o, const float* ray_d, const float* z_vals,
                         const float* noise, int R, int S, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_depth, const float* d_weights, float* const* grads33_host, void* workspace,
                         int flags, void* stream) {
    return dsn_render_rays_grad_ex(scene, V, F, packed, params33_host, poses24x3, frame_idx, zero_code, ray_o, ray_d, z_vals, noise, R, S, d_rgb,
                                   d_disp, d_acc, d_depth, d_weights, grads33_host, workspace, flags, stream, nullptr, nullptr, nullptr);
}

int dsn_render_rays_grad_ex(const void* scene, int V, int F, const void* packed, const float* const* params33_host, const float* poses24x3,
                            int frame_idx, int zero_code, const float* ray_o, const float* ray_d, const float* z_vals,
                            const float* noise, int R, int S, const float* d_rgb, const float* d_disp, const float* d_acc,
                            const float* d_depth, const float* d_weights, float* const* grads33_host, void* workspace,
                            int flags, void* stream, void* aux_stream, void* ev_fork, void* ev_join) {
    DSN_REQUIRE(scene && packed && params33_host && poses24x3 && ray_o && ray_d && z_vals && d_rgb && grads33_host && workspace,
                "dsn_render_rays_grad: null argument");
    DSN_REQUIRE(R > 0 && S > 0 && V > 0 && F > 0, "dsn_render_rays_grad: bad sizes");
    DSN_REQUIRE(frame_idx >= 0 && frame_idx < 500, "dsn_render_rays_grad: frame index outside the embedding table");
    DSN_REQUIRE((aux_stream != nullptr) == (ev_fork != nullptr) && (aux_stream != nullptr) == (ev_join != nullptr),
                "dsn_render_rays_grad_ex: the auxiliary stream and its two events go together");
    DSN_REQUIRE(!aux_stream || aux_stream != stream, "dsn_render_rays_grad_ex: the auxiliary stream must not be the call's own stream");
    for (int i = 0; i < DSN_NUM_PARAMS; ++i)
        DSN_REQUIRE(params33_host[i] && grads33_host[i], "dsn_render_rays_grad: null parameter / gradient pointer");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    const DsnTrainAux aux = {(hipStream_t)aux_stream, (hipEvent_t)ev_fork, (hipEvent_t)ev_join};
    const char* err = dsn_train_run(s, (const float*)packed, params33_host, poses24x3, frame_idx, zero_code, ray_o, ray_d, z_vals, noise, R, S, d_rgb,
                                    d_disp, d_acc, d_depth, d_weights, grads33_host, workspace, (hipStream_t)stream,
                                    (flags & DSN_GRAD_CACHED) != 0, nullptr, nullptr, nullptr, aux_stream ? &aux : nullptr);
    if (err) return dsn_fail("dsn_render_rays_grad: %s failed", err);
    return dsn_check_launch("dsn_render_rays_grad");
}

// the caller's auxiliary stream + fork / join events of dsn_render_rays_grad_ex, for callers without a HIP binding of their own
// (created and destroyed by the caller through these two; the library keeps no record of them)
int dsn_aux_create(void** aux_stream, void** ev_fork, void** ev_join) {
    DSN_REQUIRE(aux_stream && ev_fork && ev_join, "dsn_aux_create: null argument");
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        if (st) (void)hipStreamDestroy(st);
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        return dsn_fail("%s", "dsn_aux_create: hipStreamCreate / hipEventCreate failed");
    }
    *aux_stream = (void*)st; *ev_fork = (void*)a; *ev_join = (void*)b;
    return 0;
}
int dsn_aux_destroy(void* aux_stream, void* ev_fork, void* ev_join) {
    int bad = 0;
    if (aux_stream) bad |= hipStreamDestroy((hipStream_t)aux_stream) != hipSuccess;
    if (ev_fork) bad |= hipEventDestroy((hipEvent_t)ev_fork) != hipSuccess;
    if (ev_join) bad |= hipEventDestroy((hipEvent_t)ev_join) != hipSuccess;
    return bad ? dsn_fail("%s", "dsn_aux_destroy: destroy failed") : 0;
}

// backward of DualSpaceNeRF.forward on explicit points (model/spacenet.py:210-266): see dsnerf.h
int dsn_module_grad(const void* scene, int V, int F, const void* packed, const float* const* params33_host, const float* poses24x3,
                    int frame_idx, int zero_code, const float* x_world, const float* x_canon, const float* view_dir,
                    const float* zeros_n, int64_t N, const float* d_colour, const float* d_sigma, float* const* grads33_host,
                    void* workspace, void* stream) {
    DSN_REQUIRE(scene && packed && params33_host && poses24x3 && x_world && x_canon && view_dir && zeros_n && d_colour && d_sigma &&
                grads33_host && workspace, "dsn_module_grad: null argument");
    DSN_REQUIRE(N > 0 && N < ((int64_t)1 << 30) && V > 0 && F > 0, "dsn_module_grad: bad sizes");
    DSN_REQUIRE(frame_idx >= 0 && frame_idx < 500, "dsn_module_grad: frame index outside the embedding table");
    for (int i = 0; i < DSN_NUM_PARAMS; ++i)
        DSN_REQUIRE(params33_host[i] && grads33_host[i], "dsn_module_grad: null parameter / gradient pointer");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    // N "rays" of ONE sample each: origin = the world point, direction = its view direction, z = 0
    const char* err = dsn_train_run(s, (const float*)packed, params33_host, poses24x3, frame_idx, zero_code, x_world, view_dir, zeros_n,
                                    nullptr, (int)N, 1, d_colour, nullptr, nullptr, nullptr, nullptr, grads33_host, workspace,
                                    (hipStream_t)stream, false, x_canon, d_colour, d_sigma);
    if (err) return dsn_fail("dsn_module_grad: %s failed", err);
    return dsn_check_launch("dsn_module_grad");
}

// diagnostics (synchronises the stream): {ncell, ok, total, cap} of world-fine, world-coarse, canon-fine,
// canon-coarse nearest-face levels -> out16_host
int dsn_debug_nn_stats(const void* scene, int V, int F, int32_t* out16_host, void* stream) {
    DSN_REQUIRE(scene && out16_host && V > 0 && F > 0, "dsn_debug_nn_stats: bad argument");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    const DsnGrid* gs[4] = {s.nn_world.fine.g, s.nn_world.coarse.g, s.nn_canon.fine.g, s.nn_canon.coarse.g};
    for (int i = 0; i < 4; ++i) {
        DsnGrid h;
        if (hipMemcpyAsync(&h, gs[i], sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
            return dsn_fail("%s", "dsn_debug_nn_stats: copy failed");
        out16_host[4 * i + 0] = h.ncell; out16_host[4 * i + 1] = h.ok; out16_host[4 * i + 2] = h.total; out16_host[4 * i + 3] = h.cap;
    }
    return 0;
}

// where the four 64-byte level headers live inside a scene blob (host function): lets a caller copy them out asynchronously and
// watch for a level that did not fit its list capacity without the synchronisation dsn_debug_nn_stats implies
int dsn_nn_header_offsets(int V, int F, size_t* out4_host) {
    DSN_REQUIRE(out4_host && V > 0 && F > 0, "dsn_nn_header_offsets: bad argument");
    DsnSceneView s = dsn_scene_view(nullptr, V, F);
    const DsnGrid* gs[4] = {s.nn_world.fine.g, s.nn_world.coarse.g, s.nn_canon.fine.g, s.nn_canon.coarse.g};
    for (int i = 0; i < 4; ++i) out4_host[i] = (size_t)((const char*)gs[i] - (const char*)nullptr);
    return 0;
}

// utils/rays_utils.py:16-30 get_rays + :63-97 get_near_far, whole-image path (:176-184)
int dsn_camera_rays(const double* K3x3, const double* R3x3, const double* T3, const double* bounds2x3, int H, int W,
                    int convention, float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask_at_box, void* stream) {
    DSN_REQUIRE(K3x3 && R3x3 && T3 && bounds2x3 && ray_o && ray_d && near && far && mask_at_box, "dsn_camera_rays: null argument");
    DSN_REQUIRE(H > 0 && W > 0, "dsn_camera_rays: empty image");
    DSN_REQUIRE(convention == DSN_RAYS_ZJU || convention == DSN_RAYS_H36M, "dsn_camera_rays: unknown convention");
    dsn_launch_camera_rays(K3x3, R3x3, T3, bounds2x3, H, W, ray_o, ray_d, near, far, mask_at_box, (hipStream_t)stream,
                           convention == DSN_RAYS_H36M);
    return dsn_check_launch("dsn_camera_rays");
}

// workspace carve for the fused path
#define DSN_CELLMAJOR_MIN (1 << 20)   // below ~1 M samples the five extra launches cost more than they save (measured:
                                      // -0.12 ms at 128x128x32, +0.42 ms at 256x256x64)
#define DSN_TRAIN_FAR_SEARCH_MIN (1 << 15)   // train mode: batches from here on search their far canonical points cell-major
#define DSN_TRAIN_CELLMAJOR_MIN (1 << 18)    // train mode: batches from here on take the fused sampler + cell-major search + warp (round 6)
struct DsnWorkspace {
    int32_t* count;       // [128] (first word = number of active samples)
    int32_t* active;      // [N]
    uint8_t* transparent; // [N]
    float* z;             // [N] (used when the caller does not want z_vals)
    float* x_c;           // [N,3]
    float* sigma;         // [N]
    // Two 12 N-byte regions, adjacent, each used by two arrays IN PLACE (round 4; round 3 kept four arrays = 48 B per sample):
    //   G: d sigma / dx_c, overwritten sample by sample with the world normal made from it (k_normal reads its sample's gradient, then
    //      writes its normal); before the field phase the screen's keep list, after the slices the shading weights
    //   E: essence, overwritten sample by sample with the colour made from it (k_light16 / k_light write a tile's colours after they
    //      have read its essences; DSN_EARLY_STOP zeroes the "colour" of a sample it does not shade - its essence is dead by then)
    // and, during the geometry phase, together the 24 N bytes of the cell-major sort: cells (4 N) | ranks (4 N) | sorted records (16 N).
    float* essence;       // [N,3]  = E
    float* grad;          // [N,3]  = G
    float* n_w;           // [N,3]  = G
    float* colour;        // [N,3]  = E
    void* sort_scratch;   // 16 N bytes at G + 8 N (the sorted (point, id) records of the nearest-face search)
    int32_t* pos;         // [N]   samples with sigma > 0 (eval-mode split of the field kernel)
    void* masks;          // [rec_cap] x 224 B relu-mask records, indexed by the slot on the sigma > 0 list
    int64_t rec_cap;
    void* nn_small;       // per-cell scratch of the cell-major nearest-face search
    int32_t* keep;        // [N]   samples the density screen could not rule out
    int32_t* audit;       // [audit_cap] samples declared empty that DSN_SCREEN_AUDIT sends through the accurate pass anyway
    int audit_cap;
    void* T;              // [R] x 8 B  DSN_EARLY_STOP: (transmittance, slices it covers) of every ray (k_slice_alive)
    int32_t* slices;      // [R (S + 64)] the active list split by slice (slice k at k * R * L); later the shading list
    int32_t* alive;       // [N]   the current slice's samples on rays that are not finished
    size_t bytes;
};
// words of DsnWorkspace::count (device, int32): diagnostics the host mirror reads after a frame
#define DSN_CNT_ACTIVE 0      // non-transparent samples
#define DSN_CNT_POS 16        // samples with sigma > 0 (reverse pass, normals, lighting)
#define DSN_CNT_KEEP 32       // samples the density screen sent to the accurate pass
#define DSN_CNT_AUDIT 40      // DSN_SCREEN_AUDIT: audit candidates (runs past the capacity), [44] audited samples with accurate sigma > 0,
                              //                   [45] their max sigma (float bits), [46] samples audited (<= capacity)
#define DSN_CNT_RANGE 48      // dsn_render_rays_train: samples whose activations / adjoints left the fp16 range
#define DSN_CNT_FLAGGED 20    // eval mode: samples the split-fp16 passes flagged for the exact-fp32 fallback (k_field<fix> looks here first)
#define DSN_CNT_ALIVE_K 96    // DSN_EARLY_STOP: [96..127] live samples of slice k, [128..159] samples of slice k the screen kept (a counter
#define DSN_CNT_KEEP_K 128    //                 of its own per slice: no clearing launches inside the slice loop)
#define DSN_CNT_BYTES 8192     // 256 count words + (from word 256) the slice histogram of DSN_STOP_STATS: (K + 1) x K ints, K <= 32
#define DSN_CNT_HIST 256
#define DSN_CNT_SLICE 64      // DSN_EARLY_STOP: [64..95] active samples per slice (at most 32 slices), [12] alive in the current slice,
#define DSN_CNT_ALIVE 12      //                 [13] reverse-pass slots, [14] shaded samples
#define DSN_CNT_SEL 13
#define DSN_CNT_LIT 14
#define DSN_CNT_STOP 56       // [56] samples left out by ray termination, [57] samples not shaded (weight < eps), [58] DSN_STOP_STATS,
                              // [59] largest |colour| the compositor weighed (float bits; eval frames) - the scale of the early-stop bound
// (the DSN_EARLY_STOP threshold: dsn_stop_eps_scaled, dsn_common.h - it follows S and the colour scale kept in `packed`)
// samples per slice: 4 on big frames (finer termination: the converged set's 512 x 512 x 64 frame 10.32 -> 9.99 ms, w3 12.52 -> 12.22 with
// two frames in flight; profiles/r03_stop_slice_sweep.txt), 8 where the three launches per slice weigh more than the samples they save
static inline int dsn_slice_len(int R, int S) {
    const char* e = getenv("DSN_STOP_SLICE");      // experiments: samples per slice
    int L = e ? atoi(e) : ((int64_t)R * S >= ((int64_t)1 << 22) ? 4 : 8);
    if (L < 1) L = 8;
    if (L > 64) L = 64;
    return (S + L - 1) / L <= DSN_STOP_MAX_SLICES ? L : (S + DSN_STOP_MAX_SLICES - 1) / DSN_STOP_MAX_SLICES;
}
// samples per slice of the DSN_STOP_STATS histogram: HALF the uniform slice where that keeps K <= 32 - the schedule a caller cuts from the
// histogram (dsn_render_rays_ex) may put its borders there (round 6's last session: the bench frame's schedule from 2-sample statistics
// evaluates 1.3 % fewer samples than from 4-sample ones, -0.05 ms); the uniform slicing itself stays as it is (2-sample uniform slices
// measured worse: 32 launches)
static inline int dsn_stats_slice_len(int R, int S) {
    const int L = dsn_slice_len(R, S), half = L >= 2 ? L / 2 : L;
    return (S + half - 1) / half <= DSN_STOP_MAX_SLICES ? half : L;
}
// entries of the per-slice lists: K slices of R * L each (K L < S + L) for the slice length in use - round 3 reserved R (S + 64) for any
// length (8 bytes per sample at S = 64; now 4).  DSN_STOP_SLICE (experiments) changes L: set it before the workspace is sized.
static inline size_t dsn_slice_entries(size_t R, int S) {
    const int L = dsn_slice_len((int)std::min<size_t>(R, 0x7fffffff), S), K = (S + L - 1) / L;
    return R * (size_t)K * (size_t)L;
}
// Capacity of the relu-record array of a frame.  The records (224 B per sample) are what the reverse pass needs of the forward
// pass, only for samples with sigma > 0, so they are indexed by the slot on that list and sized for a FRACTION of the samples of a
// big frame.  The capacity is a property of the WORKSPACE (round 5; rounds 3-4 kept a process-wide fraction): the records are the
// LAST array of the carve, everything else has a place that depends on (R, S) alone, and the capacity is whatever the caller's
// workspace_bytes leave behind the fixed part - dsn_render_workspace_bytes_for(R, S, fraction) sizes it, dsn_render_rays_ex reads
// it back from the size it is handed.  The bench frame puts 11.6 % (hash-random parameters) / 14 % (converged parameters,
// front-to-back slices) of its samples there, a briefly trained solid 39 %: the host mirror's probe frame sizes each of its
// workspaces per checkpoint; samples beyond the capacity take the single-launch forward + reverse pass instead
// (dsn_launch_field16_from): same values, no records, ~10 % more time for them.
// DSN_RECORD_CAP (tests) lowers the capacity IN USE below what the workspace holds.
#define DSN_RECORD_FRACTION_DEFAULT 0.125f
static int64_t dsn_record_cap_for(int64_t N, float fraction) {
    const int64_t floor_ = (int64_t)1 << 21;
    if (N <= floor_) return N;
    if (!(fraction > 0.0f)) fraction = DSN_RECORD_FRACTION_DEFAULT;
    if (fraction > 1.0f) fraction = 1.0f;
    int64_t c = (int64_t)((double)N * (double)fraction);
    c = (c + 255) & ~(int64_t)255;
    return c < floor_ ? floor_ : (c > N ? N : c);
}
// rec_cap < 0: the fixed part only (w.bytes = where the records begin)
static DsnWorkspace dsn_carve(void* base, int R, int S, int64_t rec_cap) {
    DsnWorkspace w;
    size_t N = (size_t)R * S;
    char* p = (char*)base;
    w.count = (int32_t*)p;        p += DSN_CNT_BYTES;
    w.active = (int32_t*)p;       p += dsn_align256(4 * N);
    w.transparent = (uint8_t*)p;  p += dsn_align256(N);
    w.z = (float*)p;              p += dsn_align256(4 * N);
    w.x_c = (float*)p;            p += dsn_align256(12 * N);
    w.sigma = (float*)p;          p += dsn_align256(4 * N);
    w.grad = w.n_w = (float*)p;          p += 12 * N;                       // G (no padding between G and E: the sort spans both)
    w.essence = w.colour = (float*)p;    p += dsn_align256(12 * N + 256);   // E
    w.sort_scratch = (void*)((char*)w.grad + ((8 * N + 15) & ~(size_t)15));      // (16-byte aligned records; E carries 256 bytes of slack)
    w.pos = (int32_t*)p;          p += dsn_align256(4 * N);
    w.nn_small = (void*)p;        p += dsn_nn_sort_scratch_size((int64_t)N);
    // the density screen's keep list lives from the screen to the forward launch of a slice / frame: inside the field phase, where the
    // normal buffer is free (the geometry phase's sort is done with it, the early-stop shading weights and the normals come later)
    w.keep = (int32_t*)w.n_w;
    w.audit_cap = (int)(N / 32 + 1024);                                      // 1/128 of the empty samples are audited
    w.audit = (int32_t*)p;        p += dsn_align256(4 * (size_t)w.audit_cap);
    w.T = (void*)p;               p += dsn_align256(8 * (size_t)R);
    w.slices = (int32_t*)p;       p += dsn_align256(4 * dsn_slice_entries((size_t)R, S));
    w.alive = (int32_t*)p;        p += dsn_align256(4 * N);
    // the relu records LAST: nothing else moves with their capacity
    w.rec_cap = rec_cap < 0 ? 0 : rec_cap;
    w.masks = (void*)p;           p += dsn_align256(224 * (size_t)w.rec_cap);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}
static size_t dsn_workspace_fixed_bytes(int R, int S) { return dsn_carve(nullptr, R, S, -1).bytes; }
// records a workspace of `bytes` bytes holds for an R x S frame (-1: not even the fixed part fits)
static int64_t dsn_record_cap_of(int R, int S, size_t bytes) {
    const size_t fixed = dsn_workspace_fixed_bytes(R, S);
    if (bytes < fixed) return -1;
    const int64_t N = (int64_t)R * S;
    int64_t c = (int64_t)((bytes - fixed) / 224);
    if (c > N) c = N;
    const char* e = getenv("DSN_RECORD_CAP");      // tests: a smaller capacity IN USE (provokes the overflow pass)
    if (e) { const long long v = atoll(e); c = v < 1 ? (c < 1 ? c : 1) : (v < c ? v : c); }
    return c;
}

size_t dsn_render_workspace_bytes_for(int R, int S, float record_fraction) {
    if (R <= 0 || S <= 0) return 0;
    return dsn_carve(nullptr, R, S, dsn_record_cap_for((int64_t)R * S, record_fraction)).bytes;
}
size_t dsn_render_workspace_bytes(int R, int S) { return dsn_render_workspace_bytes_for(R, S, DSN_RECORD_FRACTION_DEFAULT); }
int64_t dsn_render_workspace_record_capacity(int R, int S, size_t workspace_bytes) {
    if (R <= 0 || S <= 0) return -1;
    return dsn_record_cap_of(R, S, workspace_bytes ? workspace_bytes : dsn_render_workspace_bytes(R, S));
}

// samples per uniform slice of DSN_EARLY_STOP for an R x S frame (what dsn_render_rays_ex cuts and the DSN_STOP_STATS histogram counts)
int dsn_stop_slice_len(int R, int S) { return (R > 0 && S > 0) ? dsn_slice_len(R, S) : 0; }
// samples per slice of the DSN_STOP_STATS histogram (<= dsn_stop_slice_len: see dsn_stats_slice_len)
int dsn_stop_stats_slice_len(int R, int S) { return (R > 0 && S > 0) ? dsn_stats_slice_len(R, S) : 0; }

float dsn_early_stop_colour_headroom(void) { return DSN_STOP_COLOUR_HEADROOM; }
float dsn_early_stop_eps(int S) { return dsn_stop_eps_scaled(S > 0 ? S : 1, 1.0f); }
float dsn_early_stop_eps_scaled(int S, float colour_scale) { return dsn_stop_eps_scaled(S > 0 ? S : 1, colour_scale); }

// the same calibration on the points of a FRAME: see dsnerf.h
int dsn_calibrate_screen_frame(const void* scene, int V, int F, void* packed, const void* render_workspace, int R, int S,
                               int64_t n_points, void* workspace, float* out4, void* stream) {
    DSN_REQUIRE(scene && packed && workspace && render_workspace, "dsn_calibrate_screen_frame: null argument");
    DSN_REQUIRE(V > 0 && F > 0 && n_points > 0 && R > 0 && S > 0, "dsn_calibrate_screen_frame: bad sizes");
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    const DsnWorkspace w = dsn_carve((void*)render_workspace, R, S, -1);      // (the fixed part: x_c, the active list, the count words)
    dsn_launch_calibrate_screen(s, (float*)packed, n_points, workspace, out4, (hipStream_t)stream, w.x_c, w.active, w.count + DSN_CNT_ACTIVE);
    return dsn_check_launch("dsn_calibrate_screen_frame");
}


int dsn_render_rays(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                    float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise,
                    int flags, float* out_rgb, float* out_disp, float* out_acc, float* out_depth,
                    float* out_weights, float* out_z, void* workspace, void* stream) {
    return dsn_render_rays_ex(scene, V, F, packed, ray_o, ray_d, near, far, R, S, t_vals, jitter, noise, flags, out_rgb, out_disp, out_acc,
                              out_depth, out_weights, out_z, workspace, 0, nullptr, 0, stream);
}

int dsn_render_rays_ex(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                       float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise,
                       int flags, float* out_rgb, float* out_disp, float* out_acc, float* out_depth,
                       float* out_weights, float* out_z, void* workspace, size_t workspace_bytes, const int32_t* slice_lengths_host,
                       int n_slices, void* stream) {
    DSN_REQUIRE(R > 0 && S > 0, "dsn_render_rays: empty ray batch");      // (first: empty tensors come with null pointers)
    // front-to-back schedule (DSN_EARLY_STOP): uniform slices by default, the caller's lengths otherwise
    int bounds[DSN_STOP_MAX_SLICES + 1];
    int K;
    const int L = dsn_slice_len(R, S);
    if (slice_lengths_host && n_slices > 0) {
        DSN_REQUIRE(n_slices <= DSN_STOP_MAX_SLICES, "dsn_render_rays_ex: more than 32 slices");
        int at = 0;
        for (int k = 0; k < n_slices; ++k) {
            DSN_REQUIRE(slice_lengths_host[k] >= 1 && slice_lengths_host[k] <= 64, "dsn_render_rays_ex: a slice holds 1 to 64 samples");
            bounds[k] = at;
            at += slice_lengths_host[k];
        }
        DSN_REQUIRE(at == S, "dsn_render_rays_ex: the slice lengths must add up to S");
        K = n_slices;
        bounds[K] = S;
    } else {
        K = (S + L - 1) / L;
        for (int k = 0; k <= K; ++k) bounds[k] = k * L < S ? k * L : S;
    }
    const bool custom_schedule = slice_lengths_host && n_slices > 0;
    DSN_REQUIRE(scene && packed && ray_o && ray_d && near && far && t_vals && workspace, "dsn_render_rays: null argument");
    DSN_REQUIRE(out_rgb && out_disp && out_acc && out_depth, "dsn_render_rays: null output");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_render_rays: bad V/F");
    const bool skip = (flags & DSN_SKIP_TRANSPARENT) != 0;
    DSN_REQUIRE(!(skip && noise), "dsn_render_rays: DSN_SKIP_TRANSPARENT is only exact without noise (eval mode)");
    // rays of more than 32 x 64 samples: a uniform slice would hold more than the 64 samples k_advance_T covers per launch (ADVICE r04:
    // the transmittance was then only partly advanced - safe, but termination silently did nothing).  Such frames render in one pass.
    if (!custom_schedule && L > 64) flags &= ~DSN_EARLY_STOP;
    hipStream_t st = (hipStream_t)stream;
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    // the relu-record capacity is what the caller's workspace holds behind the fixed part (0 = sized by dsn_render_workspace_bytes)
    const int64_t rec_cap = dsn_record_cap_of(R, S, workspace_bytes ? workspace_bytes : dsn_render_workspace_bytes(R, S));
    DSN_REQUIRE(rec_cap >= 1, "dsn_render_rays_ex: workspace_bytes is smaller than the fixed part of the workspace + one record "
                              "(size it with dsn_render_workspace_bytes_for)");
    DsnWorkspace w = dsn_carve(workspace, R, S, rec_cap);
    const int64_t N = (int64_t)R * S;
    float* z = out_z ? out_z : w.z;
    const DsnShareCus share((flags & DSN_SHARE_CUS) != 0);
    // Phases (DSN_PHASE_*): none of the three bits = the whole frame on `stream`.  With bits set only those parts are enqueued, so
    // that a caller with several frames in flight can put the geometry / shading kernels of one frame on a stream of their own
    // BESIDE the matrix-bound field kernels of another (events between the calls are the caller's; the state that travels
    // between the phases lives in `workspace`).  Same arguments in all three calls.
    const int ph = flags & (DSN_PHASE_GEOMETRY | DSN_PHASE_FIELD | DSN_PHASE_SHADE);
    const bool do_geom = !ph || (ph & DSN_PHASE_GEOMETRY), do_field = !ph || (ph & DSN_PHASE_FIELD), do_shade = !ph || (ph & DSN_PHASE_SHADE);
    int32_t* list = skip ? w.active : nullptr;
    int32_t* cnt = skip ? w.count : nullptr;
    const bool exh = (flags & DSN_NN_EXHAUSTIVE) != 0;
    if (do_geom) {
    const char* cm_env = getenv("DSN_CELLMAJOR_MIN");      // test / tuning override
    const long long cellmajor_min = cm_env ? atoll(cm_env) : (long long)DSN_CELLMAJOR_MIN;
    const bool cellmajor = !exh && N >= (int64_t)cellmajor_min;
    const bool fused_nn = cellmajor && !getenv("DSN_NN_UNFUSED");
    const bool lazy = (flags & DSN_LAZY_LISTS) != 0;
    // a lazily set frame (DSN_FRAME_LAZY_LISTS) outside the fused cell-major path: every cell's lists, here, before anything reads them
    // (small ray batches, DSN_NN_UNFUSED, the exhaustive cross-check) - ONCE per frame: the device header says whether the level is
    // still lazy, the chunks after the first find it complete (round 5 rebuilt everything per call)
    if (lazy && !fused_nn) dsn_launch_build_nn_complete(s.cent_world, F, s.nn_world, st);
    if (fused_nn) {
        // the sampler classifies the samples by fine cell while it writes their z (the first step of the cell-major search)
        int32_t *counts = nullptr, *outside0 = nullptr;
        dsn_nn_cellmajor_begin(w.nn_small, &counts, &outside0, st);
        dsn_launch_sample_gg(s.xyz, (flags & DSN_SAMPLE_UNIFORM) ? 0 : V, ray_o, ray_d, near, far, R, S, t_vals, jitter, z, nullptr, st,
                             s.nn_world.fine.g, (int32_t*)w.grad, counts, outside0, (int32_t*)w.grad + N);      // (+ N: every sample's rank in its cell)
        // ... and a lazily set frame gets the lists of the cells that classification found samples in (counts > 0)
        if (lazy) dsn_launch_build_nn_visited(s.cent_world, F, s.nn_world, counts, st);
    } else
        dsn_launch_sample_gg(s.xyz, (flags & DSN_SAMPLE_UNIFORM) ? 0 : V, ray_o, ray_d, near, far, R, S, t_vals, jitter, z, nullptr, st);
    if (skip) {
        if (hipMemsetAsync(w.count, 0, DSN_CNT_BYTES, st) != hipSuccess) return dsn_fail("%s", "dsn_render_rays: memset failed");
    }
    if (cellmajor) {
        // cell-major search: samples counting-sorted by fine cell, lists through the scalar cache, and the rest of the warp stage
        // fused behind the search (dsn_nn.hip, k_nns_search<WARP>).  Scratch: buffers that are not written before the field's
        // forward pass - G | E, 24 N bytes back to back (dsn_carve): cell ids (4 N) and ranks (4 N) at the start of G, the sorted
        // (point, id) records (16 N) behind them, into E (an essence / colour is read only where the density is positive: no clearing).
        // Samples outside the fine grid (none for rays clipped to the body's bounds) are left to a k_warp pass of their own.
        int32_t* g3 = (int32_t*)w.grad;
        if (!fused_nn) {      // (DSN_NN_UNFUSED, cross-check / A-B switch: round 2's form - search writes nn[], k_warp reads it)
            dsn_launch_nn_cellmajor(s.nn_world, nullptr, ray_o, ray_d, z, N, S, g3, w.sort_scratch, g3 + N, w.nn_small, st);
            dsn_launch_warp(s, nullptr, ray_o, ray_d, z, N, S, nullptr, nullptr, nullptr, w.transparent, w.x_c, nullptr, list, cnt, exh, st,
                            g3 + N, skip);
        } else {
            int32_t* outside = nullptr;
            dsn_launch_nn_cellmajor_warp(s.nn_world, ray_o, ray_d, z, N, S, g3, w.sort_scratch, w.nn_small, s.face_world, s.face_canon,
                                         w.transparent, w.x_c, list, cnt, skip, &outside, st, true, lazy);
            dsn_launch_warp(s, nullptr, ray_o, ray_d, z, N, S, nullptr, nullptr, nullptr, w.transparent, w.x_c, nullptr, list, cnt, exh, st,
                            nullptr, skip, g3, outside);
        }
    } else
        dsn_launch_warp(s, nullptr, ray_o, ray_d, z, N, S, nullptr, nullptr, nullptr, w.transparent, w.x_c, nullptr, list, cnt, exh, st,
                        nullptr, skip);
    if (skip) {
        // untouched (skipped) samples must hold density 0 for the compositor (cleared here, after the nearest-face search has
        // finished with its scratch); a colour is read only where the density is positive (k_composite, lazy_colour)
        if (hipMemsetAsync(w.sigma, 0, sizeof(float) * N, st) != hipSuccess)
            return dsn_fail("%s", "dsn_render_rays: memset failed");
    }
    }       // geometry phase
    if (do_field) {
    // (eval mode: the count words were cleared in the geometry phase; the split-fp16 passes count what they flag for k_field<fix>)
    int32_t* const fcnt = skip ? w.count + DSN_CNT_FLAGGED : nullptr;
    if (flags & DSN_FIELD_FP32)
        dsn_launch_field((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.essence, w.grad, st);
    else if (skip && (flags & DSN_EARLY_STOP)) {
        // eval mode, front to back: slices of L samples along the rays; a ray whose transmittance has fallen below eps is finished
        const bool screen = (flags & DSN_DENSITY_SCREEN) != 0;
        const bool audit = screen && (flags & DSN_SCREEN_AUDIT);
        int32_t* pcnt = w.count + DSN_CNT_POS;
        dsn_launch_slice_bucket(w.active, w.count + DSN_CNT_ACTIVE, N, S, R, bounds, K, w.slices, w.count + DSN_CNT_SLICE, st);
        dsn_launch_slice_T_init(w.T, R, st);
        // How the rays' transmittance follows the slices.  "ray" (default): one coalesced per-ray pass over slice k - 1 (k_advance_T,
        // 14 us) + the list filter (8 us).  "list" (DSN_STOP_ADVANCE=list): the filter advances the rays of its own entries (no
        // k_advance_T launch - what VERDICT r03 #4 asked for; measured: 45 us per slice, the per-entry gathers cost more than the
        // launch they save: frame 9.84 against 9.71 ms with three frames in flight, profiles/r04_stop_advance_ab.txt).  Same pairs,
        // same products, same lists either way.
        static const bool adv_env_ray = [] { const char* e = getenv("DSN_STOP_ADVANCE"); return !(e && e[0] == 'l'); }();
        const bool adv_per_ray = adv_env_ray || custom_schedule;      // (the list-fused form knows uniform slices only)
        const float* scal = (const float*)packed + OFF_SCAL;
        for (int k = 0; k < K; ++k) {
            const int s0 = bounds[k], s1 = bounds[k + 1];
            const int64_t Nk = (int64_t)R * (s1 - s0);
            const int32_t* sl = w.slices + (int64_t)R * s0;
            const int32_t* sc = w.count + DSN_CNT_SLICE + k;
            if (k > 0) {
                int32_t* acnt = w.count + DSN_CNT_ALIVE_K + k;
                // (advances the rays' transmittance over slice k - 1 on the way: no k_advance_T launch between the slices)
                if (adv_per_ray) dsn_launch_advance_T(w.sigma, w.transparent, z, ray_d, R, S, bounds[k - 1], s0, k, w.T, st);
                dsn_launch_slice_alive(sl, sc, Nk, S, L, k, w.T, w.sigma, w.transparent, z, ray_d, scal, w.alive, acnt, w.count + DSN_CNT_STOP, st,
                                       adv_per_ray);
                sl = w.alive;
                sc = acnt;
            }
            if (screen) {
                int32_t* kcnt = w.count + DSN_CNT_KEEP_K + k;
                dsn_launch_screen16((const float*)packed, s.frame, w.x_c, Nk, sl, sc, w.sigma, w.keep, kcnt, nullptr, nullptr, st,
                                    audit ? w.audit : nullptr, audit ? w.count + DSN_CNT_AUDIT : nullptr, w.audit_cap);
                sl = w.keep;
                sc = kcnt;
            }
            // (the sigma > 0 list and its relu records keep growing from slice to slice)
            dsn_launch_field16_fwd((const float*)packed, s.frame, w.x_c, Nk, sl, sc, w.sigma, w.essence, w.masks, w.pos, pcnt, st,
                                   w.rec_cap, fcnt);
            (void)s0;
        }
        // shading list: weights from the densities alone (the compositor without a colour and without per-ray outputs; flagged
        // densities are still NaN and keep their rays' samples); scratch = the normal buffer
        float* wq = w.n_w;
        dsn_launch_composite(nullptr, w.sigma, w.transparent, z, ray_d, nullptr, R, S, nullptr, nullptr, nullptr, wq, nullptr, st);
        int32_t* sel = w.alive;             // (dead by now; the list of non-transparent samples stays intact: dsn_calibrate_screen_frame)
        int32_t* lit = w.slices;
        dsn_launch_cull_lit(w.pos, pcnt, N, w.rec_cap, wq, w.sigma, S, scal, sel, w.count + DSN_CNT_SEL, lit, w.count + DSN_CNT_LIT,
                            w.count + DSN_CNT_STOP + 1, w.colour, st);
        dsn_launch_field16_bwd((const float*)packed, s.frame, w.x_c, N, w.pos, pcnt, w.grad, w.masks, st, w.sigma, w.rec_cap, sel,
                               w.count + DSN_CNT_SEL, fcnt);
        if (w.rec_cap < N)
            dsn_launch_field16_from((const float*)packed, s.frame, w.x_c, N, w.pos, pcnt, w.rec_cap, w.sigma, w.essence, w.grad, st, fcnt);
        dsn_launch_field_fix((const float*)packed, s.frame, w.x_c, N, w.pos, pcnt, w.sigma, w.essence, w.grad, st, fcnt);
        if (audit) dsn_launch_screen_audit(w.audit, w.count + DSN_CNT_AUDIT, w.audit_cap, w.sigma, w.count + DSN_CNT_AUDIT + 4, st);
        list = lit;
        cnt = w.count + DSN_CNT_LIT;
    } else if (skip) {
        // eval mode: forward for every non-transparent sample, then d sigma/dx, normals and lighting only where sigma > 0
        // (elsewhere alpha = 0 exactly and the colour is never used); count[16] = number of such samples
        int32_t* pcnt = w.count + DSN_CNT_POS;
        const bool screen = (flags & DSN_DENSITY_SCREEN) != 0;
        const bool audit = screen && (flags & DSN_SCREEN_AUDIT);
        if (screen) {
            // plain-fp16 screen: samples whose fp16 density is negative by the safety margin keep that (negative) density and leave the
            // list; count[32] = samples that go through the accurate pass
            int32_t* kcnt = w.count + DSN_CNT_KEEP;
            dsn_launch_screen16((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.keep, kcnt, nullptr, nullptr, st,
                                audit ? w.audit : nullptr, audit ? w.count + DSN_CNT_AUDIT : nullptr, w.audit_cap);
            list = w.keep;
            cnt = kcnt;
        }
        dsn_launch_field16_fwd((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.essence, w.masks, w.pos, pcnt, st,
                               w.rec_cap, fcnt);
        dsn_launch_field16_bwd((const float*)packed, s.frame, w.x_c, N, w.pos, pcnt, w.grad, w.masks, st, w.sigma, w.rec_cap, nullptr, nullptr,
                               fcnt);
        list = w.pos;
        cnt = pcnt;
        // samples of the sigma > 0 list beyond the record capacity (none on ordinary frames: the launch is N - cap empty
        // workgroups' worth of looking): forward + reverse in one launch, bit-identical values
        if (w.rec_cap < N)
            dsn_launch_field16_from((const float*)packed, s.frame, w.x_c, N, list, cnt, w.rec_cap, w.sigma, w.essence, w.grad, st, fcnt);
        // range fallback: whatever either pass flagged (sigma = NaN; such samples are on the sigma > 0 list) in exact fp32
        dsn_launch_field_fix((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.essence, w.grad, st, fcnt);
        if (audit) dsn_launch_screen_audit(w.audit, w.count + DSN_CNT_AUDIT, w.audit_cap, w.sigma, w.count + DSN_CNT_AUDIT + 4, st);
    } else {
        dsn_launch_field16((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.essence, w.grad, st);
        dsn_launch_field_fix((const float*)packed, s.frame, w.x_c, N, list, cnt, w.sigma, w.essence, w.grad, st);
    }
    }       // field phase
    // the list the shading kernels walk (what the field phase leaves; recomputed here so that a DSN_PHASE_SHADE call finds it)
    if (!(flags & DSN_FIELD_FP32) && skip) {
        if (flags & DSN_EARLY_STOP) { list = w.slices; cnt = w.count + DSN_CNT_LIT; }
        else { list = w.pos; cnt = w.count + DSN_CNT_POS; }
    }
    if (do_shade) {
    dsn_launch_normal(s, w.x_c, w.grad, N, list, cnt, nullptr, w.n_w, exh, st);
    if (flags & DSN_FIELD_FP32)
        dsn_launch_light((const float*)packed, s.frame, w.n_w, nullptr, ray_o, ray_d, z, w.essence, N, S, list, cnt, w.colour, st);
    else
        dsn_launch_light16((const float*)packed, s.frame, w.n_w, nullptr, ray_o, ray_d, z, w.essence, N, S, list, cnt, w.colour, st);
    dsn_launch_composite(w.colour, w.sigma, w.transparent, z, ray_d, noise, R, S, out_rgb, out_disp, out_acc,
                         out_weights, out_depth, st, skip, skip ? w.count + DSN_CNT_STOP + 3 : nullptr);
    if ((flags & DSN_STOP_STATS) && skip)
        dsn_launch_stop_stats(w.sigma, w.transparent, z, ray_d, R, S, dsn_stats_slice_len(R, S), (const float*)packed + OFF_SCAL,
                              w.count + DSN_CNT_STOP + 2, st, w.count + DSN_CNT_HIST, w.count + DSN_CNT_STOP + 3, dsn_slice_len(R, S));
    }       // shading phase
    return dsn_check_launch("dsn_render_rays");
}

int dsn_render_rays_train_ex(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                          float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise, int flags,
                          float* out_rgb, float* out_disp, float* out_acc, float* out_depth, float* out_weights, float* out_z,
                          void* workspace, void* grad_workspace, void* stream, void* aux_stream, void* ev_fork, void* ev_join) {
    DSN_REQUIRE(R > 0 && S > 0, "dsn_render_rays_train: empty ray batch");
    DSN_REQUIRE(scene && packed && ray_o && ray_d && near && far && t_vals && workspace && grad_workspace,
                "dsn_render_rays_train: null argument");
    DSN_REQUIRE(out_rgb && out_disp && out_acc && out_depth, "dsn_render_rays_train: null output");
    DSN_REQUIRE(V > 0 && F > 0, "dsn_render_rays_train: bad V/F");
    DSN_REQUIRE(!(flags & (DSN_SKIP_TRANSPARENT | DSN_FIELD_FP32)), "dsn_render_rays_train: dense split-fp16 evaluation only");
    DSN_REQUIRE((aux_stream != nullptr) == (ev_fork != nullptr) && (aux_stream != nullptr) == (ev_join != nullptr),
                "dsn_render_rays_train_ex: the auxiliary stream and its two events go together");
    DSN_REQUIRE(!aux_stream || aux_stream != stream, "dsn_render_rays_train_ex: the auxiliary stream must not be the call's own stream");
    hipStream_t st = (hipStream_t)stream;
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    DsnWorkspace w = dsn_carve(workspace, R, S, -1);      // (train mode keeps its records in grad_workspace: the fixed part only)
    const int64_t N = (int64_t)R * S;
    const DsnTrainCache c = dsn_train_cache(grad_workspace, N);
    float* z = out_z ? out_z : w.z;
    const bool exh = (flags & DSN_NN_EXHAUSTIVE) != 0;
    // Geometry.  Round 6: batches of DSN_TRAIN_CELLMAJOR_MIN samples or more take the eval frames' fused path - the sampler classifies
    // the samples by fine cell, the cell-major search + warp run behind it (same index, same x_c, bit for bit: dsn_nn.hip) - and with
    // DSN_LAZY_LISTS on a lazily set frame (DSN_FRAME_LAZY_LISTS) the posed mesh's lists are built for the cells the batch visits
    // only: the per-step list build of every cell (0.48 ms of an 11.6 ms step at 8192 x 64) and the per-lane list walk of k_warp
    // (0.18 ms) were the two largest non-matrix items of the step.  Smaller batches: k_warp on every cell's lists (a lazily set
    // frame is completed first).  Every sample is warped (transparent samples with positive noise are evaluated: no lazy_canon).
    static const long long train_cm_min = [] { const char* e = getenv("DSN_TRAIN_CELLMAJOR_MIN"); return e ? atoll(e) : (long long)DSN_TRAIN_CELLMAJOR_MIN; }();
    const bool lazy = (flags & DSN_LAZY_LISTS) != 0;
    const bool fused = !exh && N >= (int64_t)train_cm_min && !getenv("DSN_NN_UNFUSED");
    if (lazy && !fused) dsn_launch_build_nn_complete(s.cent_world, F, s.nn_world, st);
    if (fused) {
        int32_t *counts = nullptr, *outside0 = nullptr, *outside = nullptr;
        int32_t* g3 = (int32_t*)w.grad;      // scratch: cells (4 N) | ranks (4 N) | sorted records (16 N) across G and E, as in dsn_render_rays_ex
        dsn_nn_cellmajor_begin(w.nn_small, &counts, &outside0, st);
        dsn_launch_sample_gg(s.xyz, (flags & DSN_SAMPLE_UNIFORM) ? 0 : V, ray_o, ray_d, near, far, R, S, t_vals, jitter, z, nullptr, st,
                             s.nn_world.fine.g, g3, counts, outside0, g3 + N);
        if (lazy) dsn_launch_build_nn_visited(s.cent_world, F, s.nn_world, counts, st);
        dsn_launch_nn_cellmajor_warp(s.nn_world, ray_o, ray_d, z, N, S, g3, w.sort_scratch, w.nn_small, s.face_world, s.face_canon,
                                     c.transparent, c.x_c, nullptr, nullptr, false, &outside, st, true, lazy);
        dsn_launch_warp(s, nullptr, ray_o, ray_d, z, N, S, nullptr, nullptr, nullptr, c.transparent, c.x_c, nullptr, nullptr, nullptr, exh, st,
                        nullptr, false, g3, outside);
    } else {
        dsn_launch_sample_gg(s.xyz, (flags & DSN_SAMPLE_UNIFORM) ? 0 : V, ray_o, ray_d, near, far, R, S, t_vals, jitter, z, nullptr, st);
        dsn_launch_warp(s, nullptr, ray_o, ray_d, z, N, S, nullptr, nullptr, nullptr, c.transparent, c.x_c, nullptr, nullptr, nullptr, exh, st);
    }
    // Rows the step evaluates: a transparent sample has its density forced to 0 (can_render.py:115-120), so with noise <= 0 its
    // alpha is exactly 0 (utils/nerf_net_utils.py:30-36: relu(0 + noise) = 0): neither its colour nor its density reaches an
    // output or receives a gradient.  Everything else - transparent samples with positive noise included, their colour is
    // weighted - goes through the networks.  (~30 % of a batch is skipped with raw_noise_std > 0, every transparent sample without.)
    dsn_train_forward_rows(c.transparent, noise, N, c.live, c.bcnt, c.list1, c.rowcnt, st);
    // (train mode has no exact-fp32 twin of the stored activations: samples outside the fp16 range are counted in count[48],
    //  which the host mirror checks - Renderer.range_overflow_count())
    if (hipMemsetAsync(w.count, 0, DSN_CNT_BYTES, st) != hipSuccess) return dsn_fail("%s", "dsn_render_rays_train: memset failed");
    // The canonical points of transparent samples (evaluated when their noise is positive) lie far from the body, outside the fine
    // grid: the coarse-level cell-major search finds their nearest face first (same lists, same index as k_normal's own walk;
    // scratch: buffers of the render workspace that train mode does not use) and k_normal takes it from there.  It reads the warp
    // stage's points and the row flags, nothing of the networks: with the caller's auxiliary stream (dsn_render_rays_train_ex) it is
    // enqueued BESIDE the field kernel - whose last round of row blocks leaves most compute units idle - and joined in front of the
    // normals; without one it follows the field kernel on `stream` as before.  (The matrix kernel owns its SIMDs, DESIGN 4.5: the
    // search's waves only ever run on compute units that hold none of its workgroups.)
    const int32_t* nn_far = nullptr;
    const char* far_env = getenv("DSN_TRAIN_FAR_SEARCH_MIN");      // test / tuning override (a huge value switches it off)
    const bool far_search = !exh && N >= (far_env ? atoll(far_env) : (long long)DSN_TRAIN_FAR_SEARCH_MIN);
    hipStream_t sf = st;
    const bool forked = far_search && aux_stream != nullptr;
    if (forked) {
        sf = (hipStream_t)aux_stream;
        if (hipEventRecord((hipEvent_t)ev_fork, st) != hipSuccess || hipStreamWaitEvent(sf, (hipEvent_t)ev_fork, 0) != hipSuccess)
            return dsn_fail("%s", "dsn_render_rays_train_ex: fork of the auxiliary stream failed");
    }
    auto search_far = [&]() {
        // (scratch of the segmented search, round 6: buffers the training forward does not use - the per-slice lists + the live list,
        //  8 N contiguous bytes, for the (distance, index) keys; the active list for the wave -> cell map and the scatter cursors)
        dsn_launch_nn_cellmajor_coarse(s.nn_canon, s.cent_canon, c.x_c, c.live, N, (int32_t*)w.grad, w.sort_scratch, w.pos, w.nn_small, sf,
                                       (void*)w.slices, F, w.active, N);
        nn_far = w.pos;
    };
    if (forked) search_far();
    // (skipped rows keep whatever their density slot held: the compositor masks transparent samples itself)
    dsn_launch_field16_train((const float*)packed, s.frame, c.x_c, N, c.sigma, c.essence, c.grad, c.h0, c.a0, c.rr, c.masks, st,
                             w.count + DSN_CNT_RANGE, c.list1, c.rowcnt);
    if (forked) {
        if (hipEventRecord((hipEvent_t)ev_join, sf) != hipSuccess || hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0) != hipSuccess)
            return dsn_fail("%s", "dsn_render_rays_train_ex: join of the auxiliary stream failed");
    } else if (far_search) search_far();
    dsn_launch_normal(s, c.x_c, c.grad, N, c.list1, c.rowcnt, c.idx_c, c.n_w, exh, st, nn_far);
    dsn_launch_light16((const float*)packed, s.frame, c.n_w, nullptr, ray_o, ray_d, z, c.essence, N, S, c.list1, c.rowcnt, w.colour, st,
                       c.hl1, c.hl2, c.pre);
    // (a colour is read only where relu(density + noise) > 0: skipped rows never are)
    dsn_launch_composite(w.colour, c.sigma, c.transparent, z, ray_d, noise, R, S, out_rgb, out_disp, out_acc, out_weights,
                         out_depth, st, true);
    return dsn_check_launch("dsn_render_rays_train");
}

int dsn_render_rays_train(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                          float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise, int flags,
                          float* out_rgb, float* out_disp, float* out_acc, float* out_depth, float* out_weights, float* out_z,
                          void* workspace, void* grad_workspace, void* stream) {
    return dsn_render_rays_train_ex(scene, V, F, packed, ray_o, ray_d, near, far, R, S, t_vals, jitter, noise, flags, out_rgb, out_disp, out_acc,
                                    out_depth, out_weights, out_z, workspace, grad_workspace, stream, nullptr, nullptr, nullptr);
}

}  // extern "C"
