"""TEST INFRASTRUCTURE: ctypes front-end of oracle/liboracle.so (the CPU restatement of the
reference hot path, oracle/dsn_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PARAM_ORDER = [
    "nerf.embedding.weight",
    "nerf.stage1.0.weight", "nerf.stage1.0.bias", "nerf.stage1.2.weight", "nerf.stage1.2.bias",
    "nerf.stage1.4.weight", "nerf.stage1.4.bias", "nerf.stage1.6.weight", "nerf.stage1.6.bias",
    "nerf.stage2.0.weight", "nerf.stage2.0.bias", "nerf.stage2.2.weight", "nerf.stage2.2.bias",
    "nerf.stage2.4.weight", "nerf.stage2.4.bias",
    "nerf.density_net.0.weight", "nerf.density_net.0.bias",
    "nerf.rgb_net.1.weight", "nerf.rgb_net.1.bias", "nerf.rgb_net.3.weight", "nerf.rgb_net.3.bias",
    "lighting_mlp.lights_encoding.0.weight", "lighting_mlp.lights_encoding.0.bias",
    "lighting_mlp.lights_encoding.2.weight", "lighting_mlp.lights_encoding.2.bias",
    "lighting_mlp.lights_encoding.4.weight", "lighting_mlp.lights_encoding.4.bias",
    "pose_mlp.0.weight", "pose_mlp.0.bias", "pose_mlp.2.weight", "pose_mlp.2.bias",
    "pose_mlp.4.weight", "pose_mlp.4.bias",
]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dsn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def set_threads(n):
    """OpenMP threads of the oracle's loops (0 = leave); returns the count in effect"""
    return int(lib().orc_set_threads(C.c_int(int(n))))


def _f(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Params:
    """33 float32 arrays -> const float* const* in state_dict order."""

    def __init__(self, state: dict):
        self.arrs = [np.ascontiguousarray(np.asarray(state[k], np.float32)) for k in PARAM_ORDER]
        self.ptrs = (C.c_void_p * len(self.arrs))(*[a.ctypes.data for a in self.arrs])
        self.state = state


def linspace01(S):
    t = np.empty(S, np.float32)
    lib().orc_linspace01(C.c_int(S), _p(t))
    return t


def sample_gg(ray_o, ray_d, near, far, xyz, S, jitter=None, t_vals=None):
    ray_o, ray_d, xyz = _f(ray_o), _f(ray_d), _f(xyz)
    near, far = _f(near).copy(), _f(far).copy()
    R = ray_o.shape[0]
    z = np.empty((R, S), np.float32)
    pts = np.empty((R, S, 3), np.float32)
    j, tv = _f(jitter), _f(t_vals)
    lib().orc_sample_gg(_p(ray_o), _p(ray_d), _p(near), _p(far), C.c_int(R), _p(xyz), C.c_int(xyz.shape[0]),
                        C.c_int(S), _p(tv), _p(j), _p(z), _p(pts))
    return dict(near=near, far=far, z_vals=z, pts=pts)


def centroids(verts, faces):
    verts = _f(verts)
    faces = np.ascontiguousarray(faces, np.int32)
    c = np.empty((faces.shape[0], 3), np.float32)
    lib().orc_centroids(_p(verts), _p(faces), C.c_int(faces.shape[0]), _p(c))
    return c


def nearest_face(pts, cent):
    pts, cent = _f(pts).reshape(-1, 3), _f(cent)
    idx = np.empty(pts.shape[0], np.int32)
    lib().orc_nearest_face(_p(pts), C.c_int64(pts.shape[0]), _p(cent), C.c_int(cent.shape[0]), _p(idx))
    return idx


def warp(pts, dirs, xyz, canon, faces):
    pts = _f(pts).reshape(-1, 3)
    N = pts.shape[0]
    d = None if dirs is None else _f(dirs).reshape(-1, 3)
    xyz, canon = _f(xyz), _f(canon)
    faces = np.ascontiguousarray(faces, np.int32)
    idx = np.empty(N, np.int32)
    uv = np.empty((N, 2), np.float32)
    h = np.empty(N, np.float32)
    tr = np.empty(N, np.uint8)
    xc = np.empty((N, 3), np.float32)
    rdc = np.empty((N, 3), np.float32) if d is not None else None
    lib().orc_warp(_p(pts), _p(d), C.c_int64(N), _p(xyz), _p(canon), _p(faces), C.c_int(faces.shape[0]),
                   _p(idx), _p(uv), _p(h), _p(tr), _p(xc), _p(rdc))
    return dict(idx=idx, uv=uv, h=h, transparent=tr.astype(bool), x_c=xc, ray_d_can=rdc)


def lbs_warp(pts, xyz, faces, smpl_w, A, bw_type=0):
    """nearest-face blend weights + inverse LBS (utils/render_utils.py:352-403, utils/blend_utils.py:72-81)."""
    pts, xyz = _f(pts).reshape(-1, 3), _f(xyz)
    faces = np.ascontiguousarray(faces, np.int32)
    smpl_w, A = _f(smpl_w), _f(A).reshape(24, 16)
    N = pts.shape[0]
    idx = np.empty(N, np.int32)
    w = np.empty((N, 24), np.float32)
    tr = np.empty(N, np.uint8)
    z = np.empty((N, 3), np.float32)
    lib().orc_lbs_warp(_p(pts), C.c_int64(N), _p(xyz), _p(faces), C.c_int(faces.shape[0]), _p(smpl_w), _p(A), C.c_int(bw_type),
                       _p(idx), _p(w), _p(tr), _p(z))
    return dict(idx=idx, weights=w, transparent=tr.astype(bool), pts_zero=z)


def pose_feat(poses, params: Params):
    poses = _f(poses).reshape(24, 3)
    q = np.empty(92, np.float32)
    f = np.empty(16, np.float32)
    lib().orc_pose_feat(_p(poses), params.ptrs, _p(q), _p(f))
    return q, f


def field(x_c, params: Params, code8, pose16, want_grad=True, want_essence=True):
    x_c = _f(x_c).reshape(-1, 3)
    N = x_c.shape[0]
    code8, pose16 = _f(code8), _f(pose16)
    sig = np.empty(N, np.float32)
    ess = np.empty((N, 3), np.float32) if want_essence else None
    g = np.empty((N, 3), np.float32) if want_grad else None
    lib().orc_field(_p(x_c), C.c_int64(N), params.ptrs, _p(code8), _p(pose16), _p(sig), _p(ess), _p(g))
    return sig, ess, g


def normal_world(x_c, g, canon, xyz, faces):
    x_c, g = _f(x_c).reshape(-1, 3), _f(g).reshape(-1, 3)
    canon, xyz = _f(canon), _f(xyz)
    faces = np.ascontiguousarray(faces, np.int32)
    N = x_c.shape[0]
    idx = np.empty(N, np.int32)
    nw = np.empty((N, 3), np.float32)
    lib().orc_normal_world(_p(x_c), _p(g), C.c_int64(N), _p(canon), _p(xyz), _p(faces), C.c_int(faces.shape[0]),
                           _p(idx), _p(nw))
    return idx, nw


def lighting(n_w, x_w, view_dir, essence, params: Params, rot=None, rot_center=None, light_shift=None):
    n_w, x_w, view_dir, essence = (_f(a).reshape(-1, 3) for a in (n_w, x_w, view_dir, essence))
    N = n_w.shape[0]
    col = np.empty((N, 3), np.float32)
    r, rc, ls = _f(rot), _f(rot_center), _f(light_shift)
    lib().orc_lighting(_p(n_w), _p(x_w), _p(view_dir), _p(essence), C.c_int64(N), params.ptrs, _p(r), _p(rc), _p(ls),
                       _p(col))
    return col


def composite(raw, z_vals, rays_d, noise=None):
    raw, z_vals, rays_d = _f(raw), _f(z_vals), _f(rays_d)
    R, S = z_vals.shape
    rgb = np.empty((R, 3), np.float32)
    disp = np.empty(R, np.float32)
    acc = np.empty(R, np.float32)
    w = np.empty((R, S), np.float32)
    dep = np.empty(R, np.float32)
    n = _f(noise)
    lib().orc_composite(_p(raw), _p(z_vals), _p(rays_d), _p(n), C.c_int(R), C.c_int(S), _p(rgb), _p(disp), _p(acc),
                        _p(w), _p(dep))
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, weights=w, depth_map=dep)


def render(ray_o, ray_d, near, far, S, xyz, canon, faces, params: Params, poses, code8, rot=None, rot_center=None,
           light_shift=None, jitter=None, noise=None, t_vals=None):
    ray_o, ray_d, xyz, canon = _f(ray_o), _f(ray_d), _f(xyz), _f(canon)
    near, far = _f(near).copy(), _f(far).copy()
    faces = np.ascontiguousarray(faces, np.int32)
    R = ray_o.shape[0]
    rgb = np.empty((R, 3), np.float32)
    disp = np.empty(R, np.float32)
    acc = np.empty(R, np.float32)
    w = np.empty((R, S), np.float32)
    dep = np.empty(R, np.float32)
    z = np.empty((R, S), np.float32)
    raw = np.empty((R, S, 4), np.float32)
    poses, code8 = _f(poses).reshape(24, 3), _f(code8)
    r, rc, ls, j, n, tv = _f(rot), _f(rot_center), _f(light_shift), _f(jitter), _f(noise), _f(t_vals)
    lib().orc_render(_p(ray_o), _p(ray_d), _p(near), _p(far), C.c_int(R), C.c_int(S), _p(xyz), _p(canon), _p(faces),
                     C.c_int(xyz.shape[0]), C.c_int(faces.shape[0]), params.ptrs, _p(poses), _p(code8), _p(r), _p(rc),
                     _p(ls), _p(tv), _p(j), _p(n), _p(rgb), _p(disp), _p(acc), _p(w), _p(dep), _p(z), _p(raw))
    return dict(color=rgb, disp_map=disp, acc_map=acc, weights=w, depth_map=dep, z_vals=z, raw=raw, near=near, far=far)


def camera_rays_np(K, R, T, bounds, H, W):
    """numpy (float64) restatement of the reference's whole-image ray set-up: utils/rays_utils.py:16-30 get_rays, the
    float32 cast of my_sample_ray (:177-178) and :63-97 get_near_far evaluated on the rounded rays.  Pinned by
    tests/golden/camera_rays.npz (made from the reference's own functions).  Returns UNcompacted arrays + mask."""
    K, R, T = np.asarray(K, np.float64), np.asarray(R, np.float64), np.asarray(T, np.float64).reshape(3)
    o = -(R.T @ T)
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    xy1 = np.stack([ii, jj, np.ones_like(ii)], -1).reshape(-1, 3).astype(np.float64)
    pw = (xy1 @ np.linalg.inv(K).T - T) @ R
    ray_d = (pw - o).astype(np.float32)
    ray_o = np.broadcast_to(o, pw.shape).astype(np.float32)
    b = np.asarray(bounds, np.float64) + np.array([-0.01, 0.01])[:, None]
    ro, rd = ray_o.astype(np.float64), ray_d.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        dint = ((b[None] - ro[:, None]) / rd[:, None]).reshape(-1, 6)
        pts = dint[..., None] * rd[:, None] + ro[:, None]
    eps = 1e-6
    inside = np.ones(pts.shape[:2], bool)
    for c in range(3):
        inside &= (pts[..., c] >= b[0, c] - eps) & (pts[..., c] <= b[1, c] + eps)
    mask = inside.sum(-1) == 2
    near = np.zeros(len(ro), np.float32)
    far = np.zeros(len(ro), np.float32)
    sel = pts[mask][inside[mask]].reshape(-1, 2, 3)
    nr = np.linalg.norm(ray_d[mask], axis=1)      # float32 norm: the reference passes float32 rays (:92), numpy keeps the dtype
    d0 = np.linalg.norm(sel[:, 0] - ro[mask], axis=1) / nr
    d1 = np.linalg.norm(sel[:, 1] - ro[mask], axis=1) / nr
    near[mask] = np.minimum(d0, d1).astype(np.float32)
    far[mask] = np.maximum(d0, d1).astype(np.float32)
    return ray_o, ray_d, near, far, mask


def camera_rays_h36m_np(K, R, T, bounds, H, W):
    """numpy restatement of the Human3.6M ray set-up: utils/h36m_utils.py:14-28 get_rays (direction normalised in float64, :26),
    the float32 cast and :61-76 get_near_far - a float32 slab test on the unit direction with the +-1e-5 clamp of near-zero
    components (:64-66), against the FIRST ray's origin (:67-68).  Pinned by tests/golden/camera_rays_h36m.npz (made from the
    reference's own functions).  Returns UNcompacted arrays + mask."""
    K, R, T = np.asarray(K, np.float64), np.asarray(R, np.float64), np.asarray(T, np.float64).reshape(3)
    o = -(R.T @ T)
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    xy1 = np.stack([ii, jj, np.ones_like(ii)], -1).reshape(-1, 3).astype(np.float64)
    pw = (xy1 @ np.linalg.inv(K).T - T) @ R
    d = pw - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    ray_d = d.astype(np.float32)
    ray_o = np.broadcast_to(o, pw.shape).astype(np.float32)
    b = np.asarray(bounds, np.float32)
    norm_d = np.linalg.norm(ray_d, axis=-1, keepdims=True)
    v = ray_d / norm_d
    v[(v < 1e-5) & (v > -1e-10)] = 1e-5
    v[(v > -1e-5) & (v < 1e-10)] = -1e-5
    tmin = (b[:1] - ray_o[:1]) / v
    tmax = (b[1:2] - ray_o[:1]) / v
    t1, t2 = np.minimum(tmin, tmax), np.maximum(tmin, tmax)
    nr, fr = np.max(t1, axis=-1), np.min(t2, axis=-1)
    mask = nr < fr
    near = np.where(mask, nr / norm_d[:, 0], 0).astype(np.float32)
    far = np.where(mask, fr / norm_d[:, 0], 0).astype(np.float32)
    return ray_o, ray_d, near, far, mask
