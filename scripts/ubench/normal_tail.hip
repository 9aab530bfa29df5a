// normal_tail.hip - the arithmetic tail of k_normal (dsn_geom.hip) with the face index GIVEN and every intermediate stored: which of
// them comes out different when the kernel shares SIMDs with the split-fp16 field kernels?
#include "../../dual-space-nerf_amd/csrc/dsn_common.h"
__global__ void __launch_bounds__(256) k_tail(const DsnFaceRec* __restrict__ face_world, const DsnFaceRec* __restrict__ face_canon,
                                              const float* __restrict__ x_c, const float* __restrict__ grad, const int32_t* __restrict__ list,
                                              const int32_t* __restrict__ count, const int32_t* __restrict__ face_idx, float* __restrict__ out, int spin) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= *count) return;
    const int64_t i = list[slot];
    float p[3] = {x_c[3 * i], x_c[3 * i + 1], x_c[3 * i + 2]};
    const int fi = face_idx[i];
    // (optional) a divergent per-lane loop in front, like the list scan of k_normal
    float acc = 0.f;
    if (spin) { const int n = 50 + (int)(i % 300); for (int k = 0; k < n; ++k) acc = fmaf(acc, 0.5f, p[k % 3]); }
    DsnFaceRec fc = dsn_load_face(face_canon, fi);
    DsnFaceRec fw = dsn_load_face(face_world, fi);
    float u, v, h, s[3], e[3], pe[3], df[3], o[3], u2, v2, h2;
    dsn_project(p, fc, u, v, h);
    dsn_map2face(u, v, h, fw, s);
    for (int c = 0; c < 3; ++c) pe[c] = p[c] + grad[3 * i + c];
    dsn_project(pe, fc, u2, v2, h2);
    dsn_map2face(u2, v2, h2, fw, e);
    for (int c = 0; c < 3; ++c) df[c] = e[c] - s[c];
    dsn_normalize3(df, o);
    float* q = out + 24 * slot;
    q[0] = u; q[1] = v; q[2] = h; q[3] = s[0]; q[4] = s[1]; q[5] = s[2]; q[6] = pe[0]; q[7] = pe[1]; q[8] = pe[2]; q[9] = u2; q[10] = v2; q[11] = h2;
    q[12] = e[0]; q[13] = e[1]; q[14] = e[2]; q[15] = df[0]; q[16] = df[1]; q[17] = df[2]; q[18] = o[0]; q[19] = o[1]; q[20] = o[2];
    q[21] = fc.inv; q[22] = fw.inv; q[23] = acc;
}
extern "C" int launch_tail(const void* scene, int V, int F, const float* x_c, const float* grad, const int32_t* list, const int32_t* count,
                           const int32_t* face_idx, float* out, int n, int spin, void* stream) {
    DsnSceneView s = dsn_scene_view((void*)scene, V, F);
    hipLaunchKernelGGL(k_tail, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s.face_world, s.face_canon, x_c, grad, list, count, face_idx, out, spin);
    return hipGetLastError() != hipSuccess;
}
