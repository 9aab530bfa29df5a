// dsn_kernels.h - host-side launcher prototypes shared by the .hip translation units and dsn_api.hip
#pragma once
#include "dsn_common.h"

// persistent-workgroup kernels launch one workgroup per compute unit (DSN_PERSISTENT_GROUPS overrides the count: experiments)
#include <algorithm>
#include <cstdlib>
// (DSN_SHARE_CUS: dsn_render_rays sets the number of persistent workgroups for the launches of ITS call on ITS thread)
extern thread_local int g_dsn_persistent_override;
inline int dsn_cu_count_raw();
inline int dsn_cu_count() { return g_dsn_persistent_override > 0 ? g_dsn_persistent_override : dsn_cu_count_raw(); }
inline int dsn_cu_count_raw() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        const char* e = getenv("DSN_PERSISTENT_GROUPS");
        if (e && atoi(e) > 0) v = atoi(e);
        return v;
    }();
    return n;
}

void dsn_launch_face_setup(const float* verts, const int32_t* faces, int F, DsnFaceRec* recs, float4* cent,
                           hipStream_t st);
void dsn_launch_pose_setup(const float* packed, const float* poses, int frame_idx, int zero_code,
                           const float* light_shift, const float* rot, const float* rot_center, DsnFrameState* fs,
                           hipStream_t st, const float* pose_feat16 = nullptr);
void dsn_launch_sample_gg(const float* xyz, int V, const float* ray_o, const float* ray_d, float* near, float* far,
                          int R, int S, const float* t_vals, const float* jitter, float* z_vals, float* pts,
                          hipStream_t st, const DsnGrid* cls_grid = nullptr, int32_t* cls_cell_of = nullptr,
                          int32_t* cls_counts = nullptr, int32_t* cls_outside = nullptr, int32_t* cls_rank_of = nullptr);
// scratch of the cell-major search that the sampler fills when it classifies on the way (cleared here): per-cell counters and
// the counter of samples outside the fine grid
void dsn_nn_cellmajor_begin(void* small, int32_t** counts, int32_t** outside, hipStream_t st);
void dsn_launch_warp(const DsnSceneView& s, const float* pts, const float* ray_o, const float* ray_d,
                     const float* z_vals, int64_t N, int S, int32_t* face_idx, float* uv, float* h,
                     uint8_t* transparent, float* x_c, float* ray_d_can, int32_t* active_list, int32_t* active_count,
                     bool exhaustive, hipStream_t st, const int32_t* nn_pre = nullptr, bool lazy_canon = false,
                     const int32_t* only_cell_of = nullptr, const int32_t* only_count = nullptr);
// dsn_nn.hip: cell-major exact search of the fine lists (nn [N]: index, or -1 where the fine grid does not cover)
size_t dsn_nn_sort_scratch_size(int64_t N);
// nn [N] <- exact nearest centroid for the (live) points outside the fine grid and inside the coarse one, -1 for all others
void dsn_launch_nn_cellmajor_coarse(const DsnNNView& v, const float4* cent, const float* pts, const uint8_t* live, int64_t N,
                                    int32_t* cell_of, void* sorted, int32_t* nn, void* small, hipStream_t st, void* keys8N = nullptr, int F = 0,
                                    int32_t* wave_scratch = nullptr, int64_t wave_scratch_ints = 0);
void dsn_launch_nn_cellmajor(const DsnNNView& v, const float* pts, const float* ray_o, const float* ray_d, const float* z_vals,
                             int64_t N, int S, int32_t* cell_of, void* sorted, int32_t* nn, void* small, hipStream_t st);
// the same search with the rest of the warp stage fused behind it (transparent, x_c, active list written by the search kernel);
// *outside <- device counter of the samples left alone (cell_of < 0): dsn_launch_warp(..., only_cell_of, only_count) takes those
void dsn_launch_nn_cellmajor_warp(const DsnNNView& v, const float* ray_o, const float* ray_d, const float* z_vals, int64_t N, int S,
                                  int32_t* cell_of, void* sorted, void* small, const DsnFaceRec* face_world, const DsnFaceRec* face_canon,
                                  uint8_t* transparent, float* x_c, int32_t* active_list, int32_t* active_count, bool lazy_canon,
                                  int32_t** outside, hipStream_t st, bool classified = false, bool lazy_call = false);
void dsn_launch_lbs_warp(const DsnSceneView& s, const float* pts, int64_t N, const float* smpl_w, const float* A, int bw_type,
                         int32_t* face_idx, float* weights, uint8_t* transparent, float* pts_zero, bool exhaustive, hipStream_t st);
void dsn_launch_normal(const DsnSceneView& s, const float* x_c, const float* grad, int64_t N,
                       const int32_t* active_list, const int32_t* active_count, int32_t* face_idx_canon, float* n_w,
                       bool exhaustive, hipStream_t st, const int32_t* nn_far = nullptr);
// dsn_nn.hip
void dsn_launch_build_nn(const float4* cent, int F, const DsnNNView& nn, float pad_fine, float pad_coarse, hipStream_t st,
                         bool fine_only = false, bool dense_fine = false, bool lazy = false);
// the lists of a lazy fine level for the cells with visited[cell] > 0 (no-op on a level that holds every cell's lists)
void dsn_launch_build_nn_visited(const float4* cent, int F, const DsnNNView& nn, const int32_t* visited, hipStream_t st);
// a lazily set fine level completed for every cell - decided on the device: no-op sweeps once the level is complete
void dsn_launch_build_nn_complete(const float4* cent, int F, const DsnNNView& nn, hipStream_t st);
void dsn_launch_composite(const float* colour, const float* sigma, const uint8_t* transparent, const float* z_vals,
                          const float* ray_d, const float* noise, int R, int S, float* rgb_map, float* disp_map,
                          float* acc_map, float* weights, float* depth_map, hipStream_t st, bool lazy_colour = false,
                          int32_t* colour_max = nullptr);
void dsn_launch_field16_fwd(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                            const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                            void* masks, int32_t* pos_list, int32_t* pos_count, hipStream_t st, int64_t rec_cap,
                            int32_t* flag_count = nullptr);
void dsn_launch_field16_from(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, const int32_t* list,
                             const int32_t* count, int64_t slot_base, float* sigma, float* essence, float* grad, hipStream_t st,
                             int32_t* flag_count = nullptr);
void dsn_launch_field16_bwd(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                            const int32_t* pos_list, const int32_t* pos_count, float* grad, const void* masks,
                            hipStream_t st, float* sigma, int64_t rec_cap, const int32_t* sel = nullptr,
                            const int32_t* sel_count = nullptr, int32_t* flag_count = nullptr);
void dsn_launch_screen16(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, const int32_t* active_list,
                         const int32_t* active_count, float* sigma, int32_t* keep_list, int32_t* keep_count, float* dbg_sigma,
                         float* dbg_s1, hipStream_t st, int32_t* audit_list = nullptr, int32_t* audit_count = nullptr,
                         int audit_cap = 0);
void dsn_launch_screen_audit(const int32_t* audit_list, const int32_t* audit_count, int audit_cap, const float* sigma, int32_t* out,
                             hipStream_t st);
size_t dsn_calibrate_workspace_size(int64_t n);
void dsn_launch_calibrate_screen(const DsnSceneView& s, float* packed, int64_t n, void* workspace, float* out4, hipStream_t st,
                                 const float* frame_x_c = nullptr, const int32_t* frame_list = nullptr,
                                 const int32_t* frame_count = nullptr);
void dsn_launch_set_screen_margin(float* packed, float margin, hipStream_t st);
void dsn_launch_set_packed_scalar(float* packed, int word, float v, hipStream_t st);      // packed[OFF_SCAL + word] = v
void dsn_launch_light16(const float* packed, const DsnFrameState* fs, const float* n_w, const float* x_w,
                        const float* ray_o, const float* ray_d, const float* z_vals, const float* essence, int64_t N,
                        int S, const int32_t* active_list, const int32_t* active_count, float* colour, hipStream_t st,
                        float* tr_hl1 = nullptr, float* tr_hl2 = nullptr, float* tr_pre = nullptr);
void dsn_launch_camera_rays(const double* K, const double* R, const double* T, const double* bounds, int H, int W,
                            float* ray_o, float* ray_d, float* near, float* far, uint8_t* mask, hipStream_t st, int h36m = 0);
// dsn_field.hip
void dsn_pack_params_host(const float* const* params33_host, float* packed_host);
void dsn_launch_pack_params(const float* const* params33_dev_array, float* packed, hipStream_t st);
void dsn_launch_field(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                      const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                      float* grad, hipStream_t st);
// exact-fp32 re-evaluation of the listed samples whose sigma is NaN (range fallback of the split-fp16 kernels); flag_count
// (optional): what the split-fp16 launches counted while flagging - zero: the kernel leaves without looking
void dsn_launch_field_fix(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                          const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                          float* grad, hipStream_t st, const int32_t* flag_count = nullptr);
void dsn_launch_field16(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                        const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                        float* grad, hipStream_t st, int32_t* flag_count = nullptr);
void dsn_launch_light(const float* packed, const DsnFrameState* fs, const float* n_w, const float* x_w,
                      const float* ray_o, const float* ray_d, const float* z_vals, const float* essence, int64_t N,
                      int S, const int32_t* active_list, const int32_t* active_count, float* colour, hipStream_t st);

// dsn_train.hip: parameter gradients of Renderer.render / DualSpaceNeRF.forward (fused split-fp16 passes + in-tree fp32 / split-fp16
// weight-gradient and small-GEMM kernels; no library calls)
size_t dsn_train_workspace_size(int64_t N);
// the part of that workspace a training FORWARD fills for its backward (dsn_render_rays_train -> dsn_render_rays_grad)
struct DsnTrainCache {
    uint8_t* transparent; int32_t* idx_c; float *x_c, *sigma, *essence, *grad, *n_w, *h0, *a0, *rr; void* masks;
    float *hl1, *hl2, *pre;        // lighting MLP: hidden layers after ReLU [N,128] and the output pre-activation [N]
    uint8_t* live; int32_t *list1, *bcnt, *rowcnt;      // rows the forward evaluates: flags, ascending list, build scratch, [0] = their number
};
// rows the training forward has to evaluate: all but transparent samples whose noise is <= 0 (alpha = 0 exactly) -> list, *count
void dsn_train_forward_rows(const uint8_t* transparent, const float* noise, int64_t N, uint8_t* flag, int32_t* bcnt, int32_t* list,
                            int32_t* count, hipStream_t st);
DsnTrainCache dsn_train_cache(void* workspace, int64_t N);
void dsn_launch_field16_train(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, float* sigma,
                              float* essence, float* grad, float* tr_h, float* tr_a, float* tr_rr, void* masks, hipStream_t st,
                              int32_t* range_count = nullptr, const int32_t* row_list = nullptr, const int32_t* row_count = nullptr);
void dsn_launch_adjoint16(const float* packed, int64_t N, const void* masks, const float* a_in, float* tr_a, float* gmax,
                          hipStream_t st, int32_t* range_count = nullptr, const int32_t* row_list = nullptr,
                          const int32_t* row_count = nullptr);
void dsn_launch_tangent16(const float* packed, const float* x_c, const float* u, int64_t N, const void* masks, float* tr_t,
                          float* gmax, hipStream_t st, int32_t* range_count = nullptr, const int32_t* row_list = nullptr,
                          const int32_t* row_count = nullptr, float* colsum6 = nullptr);
// a second stream of the caller's for the training backward's two independent chains (dsn_render_rays_grad_ex): fork / join events
struct DsnTrainAux { hipStream_t stream; hipEvent_t fork, join; };
const char* dsn_train_run(const DsnSceneView& s, const float* packed, const float* const* params33, const float* poses, int frame_idx,
                          int zero_code,
                          const float* ray_o, const float* ray_d, const float* z_vals, const float* noise, int R, int S,
                          const float* d_rgb, const float* d_disp, const float* d_acc, const float* d_depth,
                          const float* d_weights, float* const* grads33, void* workspace, hipStream_t st, bool cached = false,
                          const float* ext_x_c = nullptr, const float* ext_d_col = nullptr, const float* ext_d_sig = nullptr,
                          const DsnTrainAux* aux = nullptr);

// dsn_image.hip: image epilogue on the device (post_process scatter, clamp, mse / psnr)
size_t dsn_image_workspace_size(int H, int W);
void dsn_launch_image_scatter(const float* rgb, const float* disp, const float* acc, const float* depth, int R,
                              const uint8_t* mask, int H, int W, int clamp_rgb, float* img_rgb, float* img_disp,
                              float* img_acc, float* img_depth, void* workspace, hipStream_t st);
void dsn_launch_image_psnr(const float* img_rgb, const double* gt64, const float* gt32, const uint8_t* mask, int H, int W,
                           double* out4, void* workspace, hipStream_t st);
// front-to-back slices with exact ray termination (dsn_geom.hip; DSN_EARLY_STOP in dsn_render_rays)
#define DSN_STOP_MAX_SLICES 32
// bounds[0 .. K]: slice k = samples [bounds[k], bounds[k + 1]) of every ray; its list starts at lists + R * bounds[k]
void dsn_launch_slice_bucket(const int32_t* active, const int32_t* active_count, int64_t N, int S, int R, const int* bounds, int K,
                             int32_t* lists, int32_t* counts, hipStream_t st);
// slice k >= 1: advances the rays' transmittance over the slices their pair does not cover yet and keeps the samples of live rays
// (Tk: [R] x 8 bytes, dsn_launch_slice_T_init; packed_scal = packed + OFF_SCAL: the threshold's colour scale lives there)
void dsn_launch_slice_alive(const int32_t* list, const int32_t* count, int64_t N, int S, int L, int k, void* Tk, const float* sigma,
                            const uint8_t* transparent, const float* z_vals, const float* ray_d, const float* packed_scal, int32_t* out,
                            int32_t* out_count, int32_t* stopped, hipStream_t st, bool pairs_current = false);
// the per-ray form of the advance (one coalesced pass over slice k - 1 of every ray; dsn_launch_slice_alive(..., pairs_current = true) behind it)
void dsn_launch_advance_T(const float* sigma, const uint8_t* transparent, const float* z_vals, const float* ray_d, int R, int S, int s0,
                          int s1, int k, void* Tk, hipStream_t st);
void dsn_launch_slice_T_init(void* Tk, int R, hipStream_t st);
void dsn_launch_fill_f32(float* p, int64_t n, float v, hipStream_t st);
void dsn_launch_cull_lit(const int32_t* pos, const int32_t* pos_count, int64_t N, int64_t rec_cap, const float* weight,
                         const float* sigma, int S, const float* packed_scal, int32_t* sel, int32_t* sel_count, int32_t* lit, int32_t* lit_count,
                         int32_t* culled, float* colour, hipStream_t st);
void dsn_launch_stop_stats(const float* sigma, const uint8_t* transparent, const float* z_vals, const float* ray_d, int R, int S, int L,
                           const float* packed_scal, int32_t* out, hipStream_t st, int32_t* hist = nullptr, const int32_t* colour_max = nullptr,
                           int Lu = 0);      // (Lu: the uniform slice length *out counts by; 0 = L)
