/*
 * dsnerf.h - C ABI of libdsnerf_hip.so: the MI355X (gfx950) implementation of the
 * Dual-Space-NeRF volume-rendering hot path.
 *
 * The reference (zyhbili/Dual-Space-NeRF) has no FFI layer: its boundary is the Python call
 * surface can_render.Renderer / model.spacenet.DualSpaceNeRF.  Each entry point below replaces
 * the device work behind one reference function (cited per function, paths relative to the
 * reference repo) and is what a ctypes binding inside that function would call; see
 * INTEGRATION.md for the stubs.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; float32 row-major,
 *     indices int32, masks uint8;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - `scene` is an opaque device blob (dsn_scene_bytes(V,F)); V and F are passed with it on every call
 *     (the library keeps no host-side state);
 *   - functions never allocate, never synchronise, never touch the host copy of any tensor:
 *     the caller owns every buffer (sizes from the *_bytes() helpers) and the ordering is the
 *     stream's;
 *   - return 0 on success; non-zero = error, message from dsn_last_error() (thread-local).
 *     There is no CPU fallback: if no gfx950 device / code object is available the call fails.
 */
#ifndef DSNERF_H
#define DSNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define DSN_EXPORT __attribute__((visibility("default")))
extern "C" {
#endif

#define DSN_ABI_VERSION 8
#define DSN_NUM_PARAMS 33 /* DualSpaceNeRF.state_dict(), model/spacenet.py:18-81,152-172,191-205 */

DSN_EXPORT int dsn_abi_version(void);
DSN_EXPORT const char* dsn_last_error(void);

/* ---- network parameters ------------------------------------------------------------------
 * params33_host: HOST array of 33 DEVICE pointers in state_dict order (nerf.embedding.weight,
 * nerf.stage1.{0,2,4,6}.{weight,bias}, nerf.stage2.{0,2,4}.{weight,bias}, nerf.density_net.0.*,
 * nerf.rgb_net.{1,3}.*, lighting_mlp.lights_encoding.{0,2,4}.*, pose_mlp.{0,2,4}.*), torch
 * Linear layout [out,in].  Replaces nn.Module parameter storage consumed by
 * model/spacenet.py:93-148,174-188,223-236; re-run after every optimizer step / load_state_dict. */
DSN_EXPORT size_t dsn_packed_param_bytes(void);
DSN_EXPORT int dsn_pack_params(const float* const* params33_host, void* packed, void* stream);

/* layout utility, no device work: writes the packed image of 33 HOST arrays into a HOST buffer of
 * dsn_packed_param_bytes() bytes (lets tools / tests inspect the MFMA operand layout without a GPU) */
DSN_EXPORT int dsn_pack_params_host_image(const float* const* params33_host, float* packed_host);

/* ---- body model + per-frame state ----------------------------------------------------------
 * dsn_set_body : can_render.py:382-406 Renderer.load_body_model (canonical_model["meshes"]).
 * dsn_set_frame: per-batch state: posed mesh batch["xyz"] (can_render.py:352-355), centroids
 *   (utils/render_utils.py:94), pose code batch_rod2quat+pose_mlp (model/spacenet.py:223-236,
 *   314-331), embedding row (model/spacenet.py:125-129; zero_code = `net.nerf.w is not None`),
 *   light-centre shift / rotation (model/spacenet.py:254-263; NULL = not set).
 *   light_shift3 = light_center - Th. */
DSN_EXPORT size_t dsn_scene_bytes(int V, int F);
DSN_EXPORT int dsn_set_body(void* scene, const float* canon_vertex, const int32_t* faces, int V, int F, void* stream);
DSN_EXPORT int dsn_set_frame(void* scene, int V, int F, const void* packed, const float* xyz, const float* poses24x3, int frame_idx,
                  int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2,
                  void* stream);
/* dsn_set_frame with flags (ABI 3).  DSN_FRAME_FINE_ONLY: of the posed mesh's two nearest-face levels only the fine one is
 * built (it covers the posed centroids' bounding box + 0.12 m: every sample of a ray clipped to the body's bounds, which is
 * all Renderer.render / render_view ever produce, utils/rays_utils.py:63-97 + utils/pts_utils.py:18-58); points beyond it
 * take the exhaustive sweep instead of the coarse lists - the same index, slower for such points, 0.3 ms less per frame. */
#define DSN_FRAME_FINE_ONLY 1
/* DSN_FRAME_LAZY_LISTS (ABI 6; implies DSN_FRAME_FINE_ONLY): dsn_set_frame_ex only lays out the fine level's grid; the candidate
 * lists are built by the frame that uses them - dsn_render_rays[_ex] with DSN_LAZY_LISTS - and only for the cells its own samples
 * lie in (the sampler classifies the samples by cell while it writes them; the build sits between it and the search).  Same kernels,
 * same sweeps, same lists entry for entry as the full build, for fewer cells: the samples of a 512 x 512 frame visit 48 % of the
 * posed mesh's fine cells, a rank's contiguous eighth of a partitioned frame a tenth (0.44 ms of list build per frame -> 0.1).
 * Such a level answers NO other query (its header keeps ok = 0): dsn_warp / dsn_lbs_warp / dsn_render_rays_grad without
 * DSN_GRAD_CACHED on a lazily set frame take the exhaustive sweep - the same index, slowly; set the frame without the flag for
 * those.  dsn_render_rays_train takes DSN_LAZY_LISTS like dsn_render_rays[_ex] (round 6). */
#define DSN_FRAME_LAZY_LISTS 2
DSN_EXPORT int dsn_set_frame_ex(void* scene, int V, int F, const void* packed, const float* xyz, const float* poses24x3, int frame_idx,
                     int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2, int flags,
                     void* stream);

/* Pose-only state: what a density-only query needs of a frame (DualSpaceNeRF.forward(density_only=True),
 * model/spacenet.py:223-241, used by Renderer.query_volume can_render.py:280-296 with a batch_info that holds only
 * 'poses') and what the stand-alone SpaceNet.forward(pos, rays, idx, density_only, pose_feats) consumes
 * (model/spacenet.py:93-131): embedding row `frame_idx` (zeroed when zero_code), the 16 pose features - either computed
 * from poses24x3 by batch_rod2quat + pose_mlp, or given explicitly as pose_feat16 (the pose_feats argument; one of the
 * two must be non-NULL, pose_feat16 wins) - and the light / rotation edits.  The target is a scene blob (its mesh is left
 * untouched) or a blob of dsn_pose_state_bytes() bytes, which dsn_field accepts with V = F = 0. */
DSN_EXPORT size_t dsn_pose_state_bytes(void);
DSN_EXPORT int dsn_set_pose(void* scene_or_pose_state, const void* packed, const float* poses24x3, const float* pose_feat16,
                 int frame_idx, int zero_code, const float* light_shift3, const float* rot2x2, const float* rot_center2,
                 void* stream);

/* ---- stage kernels -----------------------------------------------------------------------*/
/* utils/pts_utils.py:18-58 geometry_guided_ray_marching + :3-16 uniform_sampling.
 * near/far [R] are updated in place (reference :52-53).  t_vals [S] = torch.linspace(0,1,S);
 * jitter [R,S] = the torch.rand draw of :12 (NULL in eval).  pts [R,S,3] may be NULL. */
DSN_EXPORT int dsn_sample_gg(const void* scene, int V, int F, const float* ray_o, const float* ray_d, float* near, float* far, int R, int S,
                  const float* t_vals, const float* jitter, float* z_vals, float* pts, void* stream);

/* utils/pts_utils.py:3-16 uniform_sampling alone (cfg.MODEL.sample_points_mode == "uniform",
 * can_render.py:42-51): same arguments, near/far are not modified. */
DSN_EXPORT int dsn_sample_uniform(const void* scene, int V, int F, const float* ray_o, const float* ray_d, float* near, float* far,
                       int R, int S, const float* t_vals, const float* jitter, float* z_vals, float* pts, void* stream);

/* can_render.py:333-379 Renderer.w2l_without_lbs (+ utils/render_utils.py:84-109,
 * utils/geo_utils.py:96-113,138-156,181-200).  pts [N,3]; ray_d [N/S,3] per-ray directions
 * (NULL -> no ray_d_can).  Any output may be NULL.  active_list/active_count (optional): indices of
 * non-transparent points appended in unspecified order (count must be zeroed by the caller). */
DSN_EXPORT int dsn_warp(const void* scene, int V, int F, const float* pts, const float* ray_d, int64_t N, int S, int32_t* face_idx, float* uv,
             float* h, uint8_t* transparent, float* x_c, float* ray_d_can, int32_t* active_list,
             int32_t* active_count, int flags, void* stream);

/* Dormant alternate of the warp (nothing in the reference calls it any more, pinned at function level):
 * utils/render_utils.py:352-403 compute_nn_mesh - blend weights of each point from its nearest posed face (:112-164,
 * bw_type 0 = "rigid_center", 1 = "rigid_interp") and the transparency mask - followed by utils/blend_utils.py:72-81
 * ppts_to_pts, the inverse linear-blend skinning with those weights.  smpl_weights [V,24], joint_transforms [24,4,4]
 * row-major (device).  Outputs (any may be NULL): face_idx [N], weights [N,24], transparent [N], pts_zero [N,3].
 * flags: DSN_NN_EXHAUSTIVE. */
DSN_EXPORT int dsn_lbs_warp(const void* scene, int V, int F, const float* pts, int64_t N, const float* smpl_weights,
                 const float* joint_transforms, int bw_type, int32_t* face_idx, float* weights, uint8_t* transparent,
                 float* pts_zero, int flags, void* stream);

/* model/spacenet.py:93-148 SpaceNet.forward + :301-311 gradient(): sigma [N], essence [N,3],
 * grad = d sigma / d x_c [N,3].  If active_list != NULL only the listed points (count read from
 * active_count on the device) are evaluated and written; the rest are left untouched. */
DSN_EXPORT int dsn_field(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N, const int32_t* active_list,
              const int32_t* active_count, float* sigma, float* essence, float* grad, int flags, void* stream);

/* The same evaluation in two launches, for eval-mode rendering (what dsn_render_rays does under
 * DSN_SKIP_TRANSPARENT): dsn_field_forward writes sigma/essence for the listed points, appends the points with sigma > 0
 * to pos_list (pos_count zeroed by the caller) and keeps THEIR ReLU patterns in `records` (dsn_field_record_bytes(N) bytes;
 * record k belongs to pos_list[k]); dsn_field_reverse then writes grad = d sigma / d x_c for the points of
 * pos_list only - the others have alpha = 0 exactly (utils/nerf_net_utils.py:24-27: relu(sigma)), so their
 * normal and colour never reach a pixel.  Results are bit-identical to dsn_field's on the points both write. */
DSN_EXPORT size_t dsn_field_record_bytes(int64_t N);
DSN_EXPORT int dsn_field_forward(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N,
                      const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                      void* records, int32_t* pos_list, int32_t* pos_count, void* stream);
DSN_EXPORT int dsn_field_reverse(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N,
                      const int32_t* pos_list, const int32_t* pos_count, const void* records, float* grad, float* sigma,
                      float* essence, void* stream);
/* Range of the split-fp16 kernels.  Activations (|h| < 65000) and adjoints (|d sigma/dh| < 4e6) must fit fp16; every
 * split-fp16 kernel watches the values it splits, and a sample that leaves the range is FLAGGED: dsn_field_forward leaves
 * sigma = NaN for it and puts it on pos_list, dsn_field_reverse (which is why it takes sigma / essence, the arrays
 * dsn_field_forward wrote) and dsn_field / dsn_render_rays re-evaluate flagged samples with the exact-fp32 kernel before
 * they return.  Results for such samples are the exact-fp32 kernel's (DSN_FIELD_FP32), bit for bit. */

/* model/spacenet.py:174-188 LightingMLP.forward(normal, xyz_world, view_dir_world, essence_feature) as a pure function:
 * colour [N,3] = (ELU(lights_encoding([normal, xyz_world, view / |view|])) + 1) * essence.  All inputs [N,3] per point.
 * zero_pose_state: scratch of dsn_pose_state_bytes() bytes (cleared here).  flags: DSN_FIELD_FP32. */
DSN_EXPORT int dsn_light(const void* packed, const float* normal, const float* xyz_world, const float* view_dir_world,
              const float* essence, int64_t N, float* colour, void* zero_pose_state, int flags, void* stream);

/* model/spacenet.py:278-298 normal_local2world + :254-265 + :174-188 LightingMLP.forward.
 * x_w = world sample points [N,3], ray_d [N/S,3]; outputs face_idx_canon [N], n_w [N,3],
 * colour [N,3] (any may be NULL except colour). */
DSN_EXPORT int dsn_shade(const void* scene, int V, int F, const void* packed, const float* x_c, const float* grad, const float* x_w,
              const float* ray_d, const float* essence, int64_t N, int S, const int32_t* active_list,
              const int32_t* active_count, int32_t* face_idx_canon, float* n_w, float* colour, int flags, void* stream);

/* can_render.py:115-120 (transparent sigma-zeroing) + utils/nerf_net_utils.py:5-56 raw2outputs.
 * colour [R,S,3], sigma [R,S], transparent [R,S] (NULL = none), noise [R,S] = randn*raw_noise_std
 * (NULL in eval).  weights [R,S] may be NULL. */
DSN_EXPORT int dsn_composite(const float* colour, const float* sigma, const uint8_t* transparent, const float* z_vals,
                  const float* ray_d, const float* noise, int R, int S, float* rgb_map, float* disp_map,
                  float* acc_map, float* weights, float* depth_map, void* stream);

/* ---- "next" row: the step in front of the path -------------------------------
 * utils/rays_utils.py:16-30 get_rays + :63-97 get_near_far, as composed by my_sample_ray(nrays<=0) (:176-184):
 * K, R [3,3], T [3], bounds [2,3] (min xyz; max xyz) are float64 DEVICE arrays (the reference's numpy dtype).
 * Outputs for all H*W pixels (row-major): ray_o, ray_d [H*W,3], near, far [H*W] (0 where the ray does not cross the
 * box exactly twice), mask_at_box [H*W].  The caller compacts with the mask like the reference does (:181-183). */
#define DSN_RAYS_ZJU 0   /* utils/rays_utils.py:16-30,63-97: un-normalised directions, padded box, "exactly two faces" test in float64 */
#define DSN_RAYS_H36M 1  /* utils/h36m_utils.py:14-28,61-76 (get_rays_within_bounds :162-176): unit directions (normalised in
                          * float64), float32 slab test on the unpadded bounds with the +-1e-5 direction clamp */
DSN_EXPORT int dsn_camera_rays(const double* K3x3, const double* R3x3, const double* T3, const double* bounds2x3, int H, int W,
                    int convention, float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask_at_box, void* stream);

/* ---- image epilogue on the device ----------------------------------------------------------
 * utils/render_utils.py:466-472 post_process: row k of the compacted per-ray outputs (rgb [R,3], disp/acc/depth [R],
 * the rays of the pixels where mask_at_box is set, in pixel order) goes to the k-th masked pixel of the [H,W] images;
 * all other pixels are zero.  clamp_rgb != 0 applies test.py:62-63's clamp to [0,1] to the colour image.
 * disp/acc/depth and their images may be NULL.  workspace: dsn_image_workspace_bytes(H,W). */
DSN_EXPORT size_t dsn_image_workspace_bytes(int H, int W);
DSN_EXPORT int dsn_image_scatter(const float* rgb, const float* disp, const float* acc, const float* depth, int R,
                      const uint8_t* mask_at_box, int H, int W, int clamp_rgb, float* img_rgb, float* img_disp,
                      float* img_acc, float* img_depth, void* workspace, void* stream);
/* metrics.py:8-21 mse / psnr of an [H,W,3] float32 image against the ground truth (float64 as in the reference's
 * batch["img"], or float32; exactly one of gt_f64 / gt_f32 non-NULL), over all pixels and over mask_at_box
 * (test.py:70-71).  out4 (device, float64) = {mse_all, mse_masked, psnr_all, psnr_masked}; mask may be NULL. */
DSN_EXPORT int dsn_image_psnr(const float* img_rgb, const double* gt_f64, const float* gt_f32, const uint8_t* mask_at_box,
                   int H, int W, double* out4, void* workspace, void* stream);

/* The density screen as a stage (what dsn_render_rays runs first in eval mode): for the listed points (or all N) the
 * plain-fp16 trunk; points whose fp16 density is negative by the safety margin (calibrated for the parameters: dsn_calibrate_screen) get that negative value in sigma [N] and are dropped, the
 * others are appended to keep_list (keep_count zeroed by the caller) for dsn_field_forward. */
DSN_EXPORT int dsn_field_screen(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N,
                     const int32_t* active_list, const int32_t* active_count, float* sigma, int32_t* keep_list,
                     int32_t* keep_count, void* stream);

/* The screen's margin is a property of the PARAMETERS and travels in `packed`: dsn_pack_params writes the conservative default
 * (0.01); dsn_calibrate_screen measures it for the packed parameters and the scene's current frame state - n_points points
 * around the canonical surface, screen density sigma~ and magnitude S1 against the exact-fp32 density sigma.  With
 * dev = |sigma~ - sigma| / (S1 + 1) and rel = |sigma| / (S1 + 1) a sample is dropped wrongly iff sigma > 0 and dev > margin + rel, so
 * the margin leaves every calibration point a factor 10 of headroom in deviation:  margin = max(10 max(dev - rel / 10), 0.002), or
 * +inf (nothing is ever declared empty) when that exceeds 0.15 - and writes it into `packed`, all on the stream.
 * out (device, optional, 8 floats) = {largest deviation max(dev), margin, fraction of points the screen overflowed on, n_points,
 * fraction of the points the screen drops with that margin (the caller's cue whether the screen pays: it costs ~0.3 of an accurate
 * forward pass per sample), the statistic max(dev - rel / 10), 0, [7]}; out[7] is an INPUT: the statistic of earlier calls on other
 * frame states of the same parameters (0 = none), folded into the margin and the dropped share of this call.
 * dsn_set_screen_margin sets the margin by hand.  workspace: dsn_calibrate_workspace_bytes(n_points). */
DSN_EXPORT size_t dsn_calibrate_workspace_bytes(int64_t n_points);
/* The same calibration on the points that are being RENDERED (round 3): `render_workspace` holds a frame of R rays x S samples
 * behind the geometry phase of dsn_render_rays (DSN_PHASE_GEOMETRY with DSN_SKIP_TRANSPARENT - or a whole frame); the n_points
 * calibration points are the canonical points of its non-transparent samples, evenly spread over the list - as they are for the first
 * half of the set, moved by up to +-2 cm per axis for the second half.  The cube around the canonical centroids that
 * dsn_calibrate_screen draws from (+-0.15 m) reaches far outside the |h| <= 0.1 m shell a non-transparent sample can lie in
 * (utils/render_utils.py:103-109) and - for a trained field - outside everything the training ever saw; what is never rendered
 * should not set the margin, what is rendered must.  Same `out`, same workspace size.  A frame with fewer than 65 536 non-transparent
 * samples (a small ray batch, the first chunk of a chunked frame: a million points drawn from a handful of samples say nothing -
 * ADVICE r03) falls back to the cube; callers fold the cube's statistic in through out[7] in any case (the host mirror does).  The
 * frame's list of non-transparent samples (workspace words / `active`) survives a whole frame, DSN_EARLY_STOP included.  Declared
 * behind dsn_render_rays' flags. */
DSN_EXPORT int dsn_calibrate_screen(const void* scene, int V, int F, void* packed, int64_t n_points, void* workspace, float* out4,
                         void* stream);
DSN_EXPORT int dsn_set_screen_margin(void* packed, float margin, void* stream);

/* diagnostics of the density screen: the plain-fp16 density sigma~ [N] and the magnitude S1 [N] of its terms for
 * every point (no lists, nothing skipped) - lets tests measure the margin against dsn_field's sigma. */
DSN_EXPORT int dsn_debug_screen(const void* scene, int V, int F, const void* packed, const float* x_c, int64_t N,
                     float* sigma_screen, float* s1, int32_t* scratch_list, int32_t* scratch_count, void* stream);

/* ---- training backward: what loss.backward() computes in trainer.py:70-81 -----------------
 * Gradients of L w.r.t. the 33 parameters, given the cotangents of Renderer.render's outputs
 * (utils/loss.py:17-27 uses color and acc_map): d_rgb [R,3] (required), d_disp / d_acc / d_depth [R] and
 * d_weights [R,S] (optional, NULL = zero).  Inputs are what the forward call produced / consumed: z_vals [R,S] (the
 * sampler's output, jitter included), noise [R,S] or NULL, the frame set by dsn_set_frame with the same parameters
 * (frame_idx / zero_code / poses repeated here for the embedding and pose_mlp gradients).  params33_host / grads33_host
 * are HOST arrays of 33 device pointers in state_dict order with torch Linear layouts; every gradient is overwritten.
 * Includes the second-order path through d sigma/dx -> normal -> lighting (model/spacenet.py:251-265) as a forward
 * tangent pass.  `packed` = dsn_pack_params image of the same parameters.  Activations stay resident in `workspace`
 * (dsn_grad_workspace_bytes(R,S), 34 KB per sample). */
DSN_EXPORT size_t dsn_grad_workspace_bytes(int R, int S);
/* Renderer.render in train mode (can_render.py:137-168 with net.training): dsn_render_rays (dense evaluation, jitter / noise
 * as given) that also leaves what its backward needs in `grad_workspace` (dsn_grad_workspace_bytes(R,S)): canonical points,
 * transparency, per-layer activations, sigma-adjoints, relu records, normals.  A following dsn_render_rays_grad on the same
 * inputs with DSN_GRAD_CACHED skips the recomputation of all of it. */
#define DSN_GRAD_CACHED 1
DSN_EXPORT int dsn_render_rays_train(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                    float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise, int flags,
                    float* out_rgb, float* out_disp, float* out_acc, float* out_depth, float* out_weights, float* out_z,
                    void* workspace, void* grad_workspace, void* stream);
/* The same with the caller's auxiliary stream + fork / join events (ABI 8; the triple of dsn_render_rays_grad_ex below, all three or
 * none).  The nearest-face search of the batch's FAR canonical points (transparent samples with positive noise: six small kernels,
 * 0.25 ms at 8192 x 64) needs the warp stage's points only, not the networks: it is enqueued on aux_stream beside the field kernel,
 * whose last round of row blocks leaves most of the chip idle (2 844 blocks on 256 CUs at that batch), and joined in front of the
 * normals.  Same kernels, same arguments, same values; NULL triple = dsn_render_rays_train. */
DSN_EXPORT int dsn_render_rays_train_ex(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                    float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise, int flags,
                    float* out_rgb, float* out_disp, float* out_acc, float* out_depth, float* out_weights, float* out_z,
                    void* workspace, void* grad_workspace, void* stream, void* aux_stream, void* ev_fork, void* ev_join);
DSN_EXPORT int dsn_render_rays_grad(const void* scene, int V, int F, const void* packed, const float* const* params33_host, const float* poses24x3,
                         int frame_idx, int zero_code, const float* ray_o, const float* ray_d, const float* z_vals,
                         const float* noise, int R, int S, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_depth, const float* d_weights, float* const* grads33_host, void* workspace,
                         int flags, void* stream);
/* The same with a SECOND stream of the caller's (round 6).  Behind the adjoint of compositing the backward splits into two chains
 * that do not depend on each other - the lighting MLP's backward + the tangent pass, and the colour head's backward + the adjoint
 * pass (trainer.py:70-81 loss.backward() through model/spacenet.py:251-265 on one side, :136-147 on the other) - most of whose
 * kernels are too small to fill the chip.  With aux_stream / ev_fork / ev_join (a hipStream_t and two hipEvent_t, all three or none)
 * the second chain is enqueued on aux_stream between a fork and a join: everything the call enqueues is still ordered after what
 * `stream` held before the call and before what it gets afterwards, and nothing is synchronised on the host.  Same kernels, same
 * values.  dsn_aux_create / dsn_aux_destroy make and free such a triple for callers without a HIP binding of their own; they are
 * the caller's, the library keeps no record of them. */
DSN_EXPORT int dsn_render_rays_grad_ex(const void* scene, int V, int F, const void* packed, const float* const* params33_host, const float* poses24x3,
                         int frame_idx, int zero_code, const float* ray_o, const float* ray_d, const float* z_vals,
                         const float* noise, int R, int S, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_depth, const float* d_weights, float* const* grads33_host, void* workspace,
                         int flags, void* stream, void* aux_stream, void* ev_fork, void* ev_join);
DSN_EXPORT int dsn_aux_create(void** aux_stream, void** ev_fork, void** ev_join);
DSN_EXPORT int dsn_aux_destroy(void* aux_stream, void* ev_fork, void* ev_join);

/* Backward of DualSpaceNeRF.forward(pos[N,6], rays[N,6], frame_idx, batch_info) (model/spacenet.py:210-266) on explicit
 * points, what autograd computes when a caller differentiates (colour, density) w.r.t. the parameters (can_render.py:97-134
 * render_rays / batchify_pts in train mode): x_world = pos[:, :3], x_canon = pos[:, 3:], view_dir = rays[:, :3],
 * d_colour [N,3] / d_sigma [N] the cotangents of the two outputs, zeros_n = N zero floats (device).  The scene holds the
 * frame (dsn_set_frame with the same parameters).  Every gradient is overwritten; inputs receive no gradient (they are
 * data in the reference too).  workspace: dsn_grad_workspace_bytes(N, 1). */
DSN_EXPORT int dsn_module_grad(const void* scene, int V, int F, const void* packed, const float* const* params33_host,
                    const float* poses24x3, int frame_idx, int zero_code, const float* x_world, const float* x_canon,
                    const float* view_dir, const float* zeros_n, int64_t N, const float* d_colour, const float* d_sigma,
                    float* const* grads33_host, void* workspace, void* stream);

/* ---- fused path: can_render.py:137-168 Renderer.render on R rays --------------------------
 * flags: DSN_SKIP_TRANSPARENT evaluates the networks only on non-transparent samples (exact in
 * eval mode: their sigma is forced to 0 and their colour is multiplied by weight 0; must not be
 * set when noise != NULL).  out_weights / out_z may be NULL. */
#define DSN_SKIP_TRANSPARENT 1
/* nearest-face search by exhaustive scan instead of the exact cell-candidate lists (same result by
 * construction; kept as the on-device cross-check of the lists).  Valid for dsn_warp, dsn_shade, dsn_render_rays. */
#define DSN_NN_EXHAUSTIVE 2
/* evaluate the field with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fp32 fma chain) instead of the
 * default split-fp16 scheme (3 x v_mfma_f32_32x32x16_f16 on hi/lo operand halves: fp32-equivalent accuracy at
 * 5.3x fewer matrix cycles).  Valid for dsn_field and dsn_render_rays. */
#define DSN_FIELD_FP32 4
/* cfg.MODEL.sample_points_mode == "uniform" (can_render.py:42-51): plain uniform_sampling between the given near/far
 * instead of the geometry-guided interval.  Valid for dsn_render_rays. */
#define DSN_SAMPLE_UNIFORM 8
/* eval-mode density screen, OPT-IN since ABI 5 (rounds 1-3: on unless DSN_NO_SCREEN = 16 was given - the bit now means the opposite):
 * with DSN_SKIP_TRANSPARENT and the split-fp16 field, dsn_render_rays first runs a plain-fp16 pass of the trunk and sends only the
 * samples whose fp16 density is not negative by a safety margin - sigma~ >= -margin (magnitude of its terms + 1) - through the
 * accurate pass; the others contribute exactly zero either way.  The margin is MEASURED for the parameters (dsn_calibrate_screen*:
 * a factor 10 of headroom on a million points; 0.01 until then) and can be audited per frame (DSN_SCREEN_AUDIT): frames come out
 * bit-identical with the screen on or off on everything tested, but that is a statistical statement about the calibration set, not
 * a bound - hence not the default under a "results identical" contract. */
#define DSN_DENSITY_SCREEN 16
/* audit of the density screen: a pseudo-random 1/128 of the samples it declares empty go through the accurate pass anyway
 * (the frame stays exact); afterwards int32 word 40 of `workspace` holds how many were audited, word 44 how many of those have
 * an accurate density > 0 (must be 0: such a sample would have been dropped wrongly) and word 45 the largest such density
 * (float bits). */
#define DSN_SCREEN_AUDIT 32
/* Front-to-back evaluation with ray termination (eval mode: needs DSN_SKIP_TRANSPARENT, split-fp16 field).  raw2outputs
 * (utils/nerf_net_utils.py:24-39) weighs sample i with alpha_i * T_i, T_i = prod_{j<i}(1 - alpha_j + 1e-10) non-increasing
 * along the ray: once T < eps (<= 2^-20) every later sample weighs less than eps and all of them together add less than eps to
 * acc_map (eps * colour to the pixel, eps * far to depth_map).  The frame is evaluated in slices of 8 samples along the rays
 * (S / 32 beyond 256 samples); after each slice T of every ray is advanced with the densities just computed, the next slice
 * leaves out the finished rays (their remaining samples keep density 0), and d sigma/dx, normals and lighting are computed only
 * for samples whose own weight is >= eps.  Worst case against the dense evaluation: the S samples not shaded add up to S eps and
 * the terminated tail to eps, i.e. (S + 1) eps x the largest colour component c, so eps follows S AND c:
 * eps(S, c) = min(2^-20, 1e-4 / (2 (S + 1) max(1, c))) keeps the worst case at half of the 1e-4 bar - ABSOLUTE - at any S for colours
 * up to c (1e-6 typical).  colour = (ELU + 1) x essence is unbounded (model/spacenet.py:174-188): c is a property of the loaded
 * parameters, kept in `packed` (1 after dsn_pack_params; dsn_set_early_stop_colour_scale sets it - the host mirror measures it: every
 * eval frame leaves the largest |colour| its compositor weighed in word 59 of `workspace`, float bits).  When no ray saturates,
 * weights / acc / depth keep their bits and only the colour moves (< eps x colour per unshaded sample).
 * int32 words 56 / 57 of `workspace`: samples left out by termination / samples not shaded. */
#define DSN_EARLY_STOP 64
/* the termination / shading threshold DSN_EARLY_STOP and DSN_STOP_STATS use for rays of S samples (host functions, no device work):
 * dsn_early_stop_eps(S) = dsn_early_stop_eps_scaled(S, 1) */
DSN_EXPORT float dsn_early_stop_eps(int S);
/* samples per uniform slice of an R x S frame: 4 on frames of >= 4 M samples, 8 below, S / 32 (rounded up) beyond 32 slices; rays of
 * more than 2048 samples (a slice would exceed 64 samples) render in one pass whatever the flag says */
DSN_EXPORT int dsn_stop_slice_len(int R, int S);
/* samples per slice of the DSN_STOP_STATS histogram (ABI 8: half the uniform slice where that keeps at most 32 slices - finer borders for
 * the schedule a caller cuts from it; the histogram is (K + 1) x K ints with K = ceil(S / this)) */
DSN_EXPORT int dsn_stop_stats_slice_len(int R, int S);
DSN_EXPORT float dsn_early_stop_eps_scaled(int S, float colour_scale);
/* the factor between the largest colour a frame weighed (word 59 of its workspace) and the colour scale a caller should set from it:
 * 2.  DSN_STOP_STATS applies it itself: its counts and histogram use the threshold for max(the scale in `packed`, 2 x the frame's own
 * largest colour) - the threshold the sliced frames decided by them will run with. */
DSN_EXPORT float dsn_early_stop_colour_headroom(void);
/* colour scale c of the early-stop threshold for THESE parameters (stream-ordered write into `packed`; values < 1 count as 1) */
DSN_EXPORT int dsn_set_early_stop_colour_scale(void* packed, float colour_scale, void* stream);
/* statistics for the caller's decision whether DSN_EARLY_STOP pays (a frame rendered WITHOUT it): word 58 of `workspace` =
 * non-transparent samples that lie in a slice whose ray had T < eps when the slice began (what DSN_EARLY_STOP would leave out;
 * compare with word 0, the non-transparent samples).  Slicing costs a few launches per slice, ~0.5 ms on a 512 x 512 x 64 frame. */
#define DSN_STOP_STATS 128
/* Phases of the fused path.  None of the three bits: the whole frame (all of the above) on `stream`.  With bits set
 * dsn_render_rays enqueues only those parts - DSN_PHASE_GEOMETRY: sampler, cell-major nearest-face search, warp, clears
 * (utils/pts_utils.py:18-58, can_render.py:333-379); DSN_PHASE_FIELD: density screen, canonical field forward / reverse, range
 * fallback, front-to-back slices (model/spacenet.py:93-148,301-311); DSN_PHASE_SHADE: normals, lighting MLP, compositing
 * (model/spacenet.py:278-298,174-188, utils/nerf_net_utils.py:5-56).  A caller with several frames in flight puts geometry and
 * shading of one frame on a stream of their own beside the matrix-bound field kernels of another (the field kernels occupy every
 * compute unit with one persistent workgroup; the geometry kernels are sized to fit into the LDS and registers they leave), with its
 * own events between the three calls; everything that travels between the phases lives in `workspace`.  Same arguments in all
 * three calls; dsn_set_frame belongs in front of the geometry phase, on its stream. */
#define DSN_PHASE_GEOMETRY 256
#define DSN_PHASE_FIELD 512
#define DSN_PHASE_SHADE 1024
/* Frames in flight.  The field kernels are persistent: one workgroup per compute unit walks the tiles of a launch.  With several
 * frames in flight on streams of their own, DSN_SHARE_CUS makes them take 7/8 of the compute units (a multiple of 8: one per XCD less
 * ... 28 of 32 per XCD on the MI355X) and leaves the rest to the neighbours' small kernels: -1.4 % per frame with three frames in
 * flight, +3 to +6 % for a frame that runs alone (profiles/r03_frames_in_flight.txt) - set it only when frames overlap.  Same values. */
#define DSN_SHARE_CUS 2048
/* the scene's frame was set with DSN_FRAME_LAZY_LISTS: this call's geometry phase completes the posed mesh's fine lists - for the
 * cells its samples visit (frames of >= 1 M samples: the cell-major search; those lists belong to THIS call: the next call with the
 * flag builds its own, the next call without it does not walk them), for every cell otherwise (once: the level is an ordinary,
 * fully built one afterwards and later calls find nothing to do).  Without the flag a lazily set frame still renders exactly (every
 * sample takes the exhaustive sweep - also after an earlier call with the flag); with the flag on a fully built level nothing is
 * rebuilt. */
#define DSN_LAZY_LISTS 4096
/* (The test override DSN_STOP_SLICE changes the workspace layout and is read at every call: set it before the workspace is sized
 *  and leave it alone while it is in use.) */
DSN_EXPORT size_t dsn_render_workspace_bytes(int R, int S);
/* The relu records the eval-mode reverse pass reads (224 B per sample with sigma > 0) are the largest item of the workspace.  Frames
 * of more than 2 M samples reserve them for a FRACTION of the samples (a frame of hash-random parameters puts 12 % of its samples
 * there, the converged checkpoint 14 % with front-to-back slices, a briefly trained solid 39 %); samples beyond the capacity are
 * evaluated by a single-launch forward + reverse pass: same values, but their forward pass runs twice.  The capacity is a property of
 * the WORKSPACE, not of the process (ABI 6; ABI 5 kept a process-wide fraction): the records are the last array of the workspace and
 * everything in front of them depends on (R, S) alone, so
 *   dsn_render_workspace_bytes_for(R, S, f)          bytes of a workspace with records for a fraction f of the samples (f <= 0 or
 *                                                    NaN: the default 1/8 = dsn_render_workspace_bytes(R, S); frames of <= 2 M
 *                                                    samples always get records for all of them);
 *   dsn_render_rays_ex(..., workspace, workspace_bytes, ...)
 *                                                    takes the capacity from the size it is handed (0 = "sized by
 *                                                    dsn_render_workspace_bytes(R, S)"); a size below the fixed part fails loudly;
 *   dsn_render_workspace_record_capacity(R, S, bytes) says what a workspace of that size holds (-1: too small).
 * A caller who has seen word 16 of a frame's workspace (samples with sigma > 0) come near the capacity gives ITS workspace more -
 * between frames; two workspaces of one process never size each other.  Host functions, no device work, no state. */
DSN_EXPORT size_t dsn_render_workspace_bytes_for(int R, int S, float record_fraction);
DSN_EXPORT int64_t dsn_render_workspace_record_capacity(int R, int S, size_t workspace_bytes);
DSN_EXPORT int dsn_calibrate_screen_frame(const void* scene, int V, int F, void* packed, const void* render_workspace, int R, int S,
                               int64_t n_points, void* workspace, float* out8, void* stream);
DSN_EXPORT int dsn_render_rays(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                    float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise,
                    int flags, float* out_rgb, float* out_disp, float* out_acc, float* out_depth,
                    float* out_weights, float* out_z, void* workspace, void* stream);
/* The same with a SCHEDULE for DSN_EARLY_STOP: slice_lengths_host[0 .. n_slices) (host array, 1 .. 64 samples each, at most 32 slices,
 * adding up to S) instead of uniform slices of 4 / 8 samples.  A slice costs a launch of every field kernel whatever it holds (a
 * 512 x 512 frame's late slices hold 40-90 k samples, one to three rounds of the chip's 32 k-sample capacity), so where few rays end a
 * longer slice is cheaper than two short ones; termination is then checked less often there, i.e. MORE samples are evaluated: the
 * error bound of DSN_EARLY_STOP is unchanged.  The statistics to choose from: a DSN_STOP_STATS frame leaves, from int32 word 256 of
 * `workspace`, hist[g][k] (g = 0 .. K, k = 0 .. K - 1, K = ceil(S / L) uniform slices) = non-transparent samples of slice k on rays whose
 * first slice with T < eps at its start is g (g = K: never) - a group of slices that starts at slice a evaluates slice k >= a on the
 * rays with g > a.  NULL / 0 = uniform slices = dsn_render_rays.  (The host mirror picks the schedule by dynamic programming over
 * that histogram: _lib.choose_stop_schedule.)  workspace_bytes: see dsn_render_workspace_bytes_for. */
DSN_EXPORT int dsn_render_rays_ex(const void* scene, int V, int F, const void* packed, const float* ray_o, const float* ray_d, float* near,
                       float* far, int R, int S, const float* t_vals, const float* jitter, const float* noise,
                       int flags, float* out_rgb, float* out_disp, float* out_acc, float* out_depth,
                       float* out_weights, float* out_z, void* workspace, size_t workspace_bytes, const int32_t* slice_lengths_host,
                       int n_slices, void* stream);

/* diagnostics, NOT for the hot path (synchronises `stream`): {ncell, ok, total entries, capacity} of the four
 * nearest-face list levels (world fine/coarse, canonical fine/coarse) into a HOST array of 16 int32. */
DSN_EXPORT int dsn_debug_nn_stats(const void* scene, int V, int F, int32_t* out16_host, void* stream);

/* byte offsets of the four 64-byte level headers inside a scene blob, in the order of dsn_debug_nn_stats (host function, no device
 * work).  Header layout (int32 words): [8] cells, [9] ok (1 = the level is in use; 0 = not built, or its lists did not fit the
 * capacity - queries then fall through to the next level / the exhaustive sweep: SAME index, 10-50x slower), [10] list entries
 * the level needs, [11] capacity.  The host mirror copies the posed mesh's headers out asynchronously every few frames and warns
 * when entries > capacity (a mesh whose tessellation the fixed capacities - 2000 F entries per level - do not cover). */
DSN_EXPORT int dsn_nn_header_offsets(int V, int F, size_t* out4_host);

/* diagnostics: after a DSN_SKIP_TRANSPARENT render, int32 word 0 of `workspace` holds the number of non-transparent
 * samples, word 32 the number the density screen sent to the accurate pass (field forward evaluated) and word 16 the number
 * of those with sigma > 0 (d sigma/dx, normal and lighting evaluated); after dsn_render_rays_train word 48 holds the number of
 * samples whose activations left the fp16 range (train mode has no exact fallback: must be 0) - device memory. */
#ifdef __cplusplus
}
#endif
#endif /* DSNERF_H */
