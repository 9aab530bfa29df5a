"""Deterministic synthetic inputs for the volume-rendering hot path.

The licensed inputs of the reference (SMPL_NEUTRAL.pkl, ZJU-Mocap / Human3.6M frames,
trained checkpoints) cannot ship, so every test / bench input is generated here from
integer hashes only (no torch RNG, no files): the same bytes on every machine.

What is generated mirrors the reference's batch dict
(/root/reference/dataloader/zju_mocap_dataset.py:160-187) and body model
(/root/reference/can_render.py:382-406):

* ``make_body(V)``      closed genus-0 triangle mesh, V vertices, F = 2V-4 faces
                        (V=6890 -> F=13776, the SMPL counts), X-pose-like extents.
* ``pose_body``         smooth non-rigid deformation + translation -> posed ``xyz``.
* ``make_rays``         pinhole camera rays + AABB near/far (ZJU convention: ray_d is
                        NOT normalised, /root/reference/utils/rays_utils.py:27).
* ``make_state_dict``   the 33 tensors of DualSpaceNeRF.state_dict() (names / shapes as
                        dumped from /root/reference/model/spacenet.py:191-208).
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n floats in [0,1): splitmix64 of (index, seed); top 24 bits -> float32 exactly."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed + 1) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def hash_normal(n: int, seed: int) -> np.ndarray:
    """Box-Muller on two hash streams (float64 math, float32 result)."""
    u1 = hash_uniform(n, seed).astype(np.float64)
    u2 = hash_uniform(n, seed + 7919).astype(np.float64)
    u1 = np.maximum(u1, 2.0 ** -24)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


# --------------------------------------------------------------------------------------
# body
# --------------------------------------------------------------------------------------
def _fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)


def _cap_lattice(n: int, axis, radius: float) -> np.ndarray:
    """n Fibonacci-lattice points on the spherical cap of angular radius `radius` (rad) about `axis` (area-uniform on the cap)"""
    i = np.arange(n, dtype=np.float64) + 0.5
    cos_t = 1.0 - (1.0 - np.cos(radius)) * i / n
    sin_t = np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t))
    th = np.pi * (1.0 + 5.0 ** 0.5) * i
    local = np.stack([np.cos(th) * sin_t, np.sin(th) * sin_t, cos_t], -1)
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    h = np.array([1.0, 0.0, 0.0]) if abs(a[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    e1 = np.cross(a, h); e1 /= np.linalg.norm(e1)
    e2 = np.cross(a, e1)
    return local[:, 0:1] * e1 + local[:, 1:2] * e2 + local[:, 2:3] * a


# share of the vertices and angular radius of the dense caps of make_body(nonuniform=True): head, two hands, two feet.  SMPL's X-pose
# fixture of the reference (tool/X_smpl_vertices.npy, measured, not shipped) has 18.8 % of its 6890 vertices in the head, 22.7 % in the
# two hands and 7.8 % in the feet; its nearest-neighbour spacing runs from 1.4 mm (1st percentile) to 30 mm (99th), up to 372 vertices
# lie within 5 cm of one (uniform lattice body: 117)
# (with these radii the synthetic body has 1.2 mm / 5.1 mm / 52 mm spacing at the 1st / 50th / 99th percentile, up to 358 vertices within
#  5 cm of one, and triangle areas spanning 560:1 between the 1st and the 99th percentile)
_DENSE_CAPS = [((0.0, 1.0, 0.0), 0.188, 0.40), ((1.0, 0.35, 0.0), 0.1135, 0.16), ((-1.0, 0.35, 0.0), 0.1135, 0.16),
               ((0.30, -1.0, 0.0), 0.039, 0.14), ((-0.30, -1.0, 0.0), 0.039, 0.14)]


def make_body(V: int = 6890, nonuniform: bool = False):
    """Return (canonical_vertex [V,3] f32, faces [2V-4,3] int64).

    Topology: convex hull of V points on the unit sphere (every point is a hull vertex, so the triangulation is a closed genus-0
    mesh with exactly 2V-4 faces).  nonuniform=False: a Fibonacci lattice - uniform triangle density.  nonuniform=True (VERDICT r03
    #3): SMPL-like tessellation - 49 % of the vertices sit in dense caps at the head, the hands and the feet (shares measured on the
    reference's X-pose vertex fixture, _DENSE_CAPS), the rest is the uniform lattice: triangle areas span more than 30:1, a
    hand's ~780 vertices lie within a few centimetres of each other - what the nearest-face candidate lists have to cope with.
    Geometry: the points are pushed out radially into a five-lobed star ("gingerbread"
    figure: head, two arms, two legs) and flattened in z, giving extents close to the
    reference's X-pose fixture (tool/X_smpl_vertices.npy: x +-0.87, y -1.0..0.56, z +-0.15).
    """
    from scipy.spatial import ConvexHull

    if nonuniform:
        parts, used = [], 0
        for axis, share, radius in _DENSE_CAPS:
            n = int(round(share * V))
            parts.append(_cap_lattice(n, axis, radius))
            used += n
        parts.append(_fibonacci_sphere(V - used))
        d = np.concatenate(parts, 0)
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    else:
        d = _fibonacci_sphere(V)
    hull = ConvexHull(d)
    f = hull.simplices.astype(np.int64)
    assert f.shape[0] == 2 * V - 4, f.shape
    # orient outward
    a, b, c = d[f[:, 0]], d[f[:, 1]], d[f[:, 2]]
    flip = (np.cross(b - a, c - a) * (a + b + c)).sum(-1) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    # deterministic face order (qhull's order is implementation-defined)
    key = np.lexsort((f[:, 2], f[:, 1], f[:, 0]))
    f = f[key]

    def unit(v):
        v = np.asarray(v, np.float64)
        return v / np.linalg.norm(v)

    # the star is built in the xy plane; z is the thin axis
    planar = d.copy()
    planar[:, 2] *= 0.0
    nrm = np.linalg.norm(planar, axis=-1, keepdims=True)
    pdir = planar / np.maximum(nrm, 1e-9)
    lobes = [  # axis (xy), reach, sharpness
        (unit([0.0, 1.0, 0.0]), 0.36, 9.0),       # head
        (unit([1.0, 0.35, 0.0]), 0.62, 12.0),     # left arm
        (unit([-1.0, 0.35, 0.0]), 0.62, 12.0),    # right arm
        (unit([0.30, -1.0, 0.0]), 0.72, 11.0),    # left leg
        (unit([-0.30, -1.0, 0.0]), 0.72, 11.0),   # right leg
    ]
    r = np.full(V, 0.22)
    for ax, reach, k in lobes:
        r = r + reach * np.exp(k * ((pdir * ax).sum(-1) - 1.0))
    xy = d[:, :2] * r[:, None]
    z = d[:, 2] * (0.11 + 0.03 * np.cos(3.0 * d[:, 1]))
    canon = np.stack([xy[:, 0], xy[:, 1] - 0.22, z + 0.02], -1)
    return canon.astype(np.float32), f


def make_small_body(subdiv: int = 2):
    """Icosphere-like small closed mesh (V=162, F=320 at subdiv=2) stretched to a capsule."""
    V = 10 * 4 ** subdiv + 2
    from scipy.spatial import ConvexHull

    d = _fibonacci_sphere(V)
    f = ConvexHull(d).simplices.astype(np.int64)
    a, b, c = d[f[:, 0]], d[f[:, 1]], d[f[:, 2]]
    flip = (np.cross(b - a, c - a) * (a + b + c)).sum(-1) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    canon = d * np.array([0.35, 0.80, 0.22]) + np.array([0.0, -0.1, 0.02])
    return canon.astype(np.float32), f


def pose_body(canon: np.ndarray, seed: int = 3, trans=(0.2, -0.1, 1.0)) -> np.ndarray:
    """Smooth non-rigid 'pose': twist about y growing with height, a forward bend and a mild
    anisotropic scale, then the global translation Th.  Returns xyz [V,3] f32."""
    p = canon.astype(np.float64)
    u = hash_uniform(4, seed).astype(np.float64)
    tw = (0.5 + 0.4 * u[0]) * p[:, 1]                 # twist angle (rad) ~ height
    c, s = np.cos(tw), np.sin(tw)
    x = c * p[:, 0] + s * p[:, 2]
    z = -s * p[:, 0] + c * p[:, 2]
    y = p[:, 1]
    bend = (0.25 + 0.2 * u[1]) * x                     # arms swing in z
    z = z + bend * np.abs(x)
    y = y * (1.0 + 0.05 * u[2]) + 0.04 * np.sin(3.0 * x)
    out = np.stack([x * (1.0 - 0.04 * u[3]), y, z], -1) + np.asarray(trans, np.float64)
    return out.astype(np.float32)


# --------------------------------------------------------------------------------------
# camera
# --------------------------------------------------------------------------------------
def make_rays(H: int, W: int, xyz: np.ndarray, cam_dist: float = 2.6, focal_frac: float = 1.05,
              pad: float = 0.05, unit_dirs: bool = False, fit_box: bool = False, yaw: float = 0.35, pitch: float = -0.12):
    """Pinhole rays for an H x W image looking at the body's centre from -z at `cam_dist` (turned by `yaw` about y and
    `pitch` about x: other values give the other cameras of a multi-view rig, scripts/train_w4.py).

    Returns dict(ray_o [R,3], ray_d [R,3], near [R], far [R]) float32, R = H*W.
    ray_d = K^-1 [u,v,1] rotated to world (|d| in 1..~1.1, not normalised) - ZJU convention,
    /root/reference/utils/rays_utils.py:16-30.  near/far: slab intersection with the padded
    body AABB (/root/reference/utils/rays_utils.py:63-97); rays missing the box get the
    box's depth range so that all R rays are renderable (mask_at_box == all True).
    """
    ctr = 0.5 * (xyz.min(0) + xyz.max(0)).astype(np.float64)
    lo = xyz.min(0).astype(np.float64) - pad
    hi = xyz.max(0).astype(np.float64) + pad
    # (defaults: a small fixed yaw/pitch so rays are not axis-aligned)
    Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
    Rc2w = Ry @ Rx
    o = ctr - Rc2w @ np.array([0.0, 0.0, cam_dist])
    focal = focal_frac * max(H, W)
    if fit_box:
        # The reference renders only rays that cross the body's 3-D bounds (mask_at_box,
        # dataloader/zju_mocap_dataset.py + utils/rays_utils.py:63-97).  To get H*W such rays, zoom in until
        # the padded AABB's silhouette covers the whole image: the largest focal length whose corner rays
        # still hit the box (bisection), i.e. the whole image is "mask_at_box".
        def all_hit(fc):
            cs = np.array([[-0.5 * W + 0.5, -0.5 * H + 0.5], [0.5 * W - 0.5, -0.5 * H + 0.5],
                           [-0.5 * W + 0.5, 0.5 * H - 0.5], [0.5 * W - 0.5, 0.5 * H - 0.5]])
            # border pixels (the silhouette of a convex box is convex: borders suffice)
            ts = np.linspace(-0.5, 0.5, 65)
            bx = np.concatenate([np.stack([ts * W, np.full_like(ts, -0.5 * H + 0.5)], 1), np.stack([ts * W, np.full_like(ts, 0.5 * H - 0.5)], 1),
                                 np.stack([np.full_like(ts, -0.5 * W + 0.5), ts * H], 1), np.stack([np.full_like(ts, 0.5 * W - 0.5), ts * H], 1), cs])
            dd = np.concatenate([bx / fc, np.ones((bx.shape[0], 1))], 1) @ Rc2w.T
            with np.errstate(divide="ignore", invalid="ignore"):
                a, b = (lo - o) / dd, (hi - o) / dd
            return bool((np.minimum(a, b).max(-1) < np.maximum(a, b).min(-1)).all())
        f_lo, f_hi = focal, 64.0 * focal
        if not all_hit(f_lo):
            for _ in range(40):
                mid = 0.5 * (f_lo + f_hi)
                if all_hit(mid):
                    f_hi = mid
                else:
                    f_lo = mid
            focal = f_hi * 1.001
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dc = np.stack([(jj - 0.5 * W + 0.5) / focal, (ii - 0.5 * H + 0.5) / focal, np.ones_like(jj)], -1)
    d = (dc.reshape(-1, 3) @ Rc2w.T)
    if unit_dirs:
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / d
        t1 = (hi - o) / d
    tn = np.minimum(t0, t1).max(-1)
    tf = np.maximum(t0, t1).min(-1)
    hit = tn < tf
    dn = np.linalg.norm(d, axis=-1)
    near = np.where(hit, tn, (cam_dist - 0.6) / dn)
    far = np.where(hit, tf, (cam_dist + 0.6) / dn)
    R = H * W
    return {
        "ray_o": np.broadcast_to(o, (R, 3)).astype(np.float32).copy(),
        "ray_d": d.astype(np.float32),
        "near": near.astype(np.float32),
        "far": far.astype(np.float32),
        "hit_box": hit,
    }


# --------------------------------------------------------------------------------------
# network parameters
# --------------------------------------------------------------------------------------
STATE_SHAPES = [
    ("nerf.embedding.weight", (500, 8)),
    ("nerf.stage1.0.weight", (256, 87)), ("nerf.stage1.0.bias", (256,)),
    ("nerf.stage1.2.weight", (256, 256)), ("nerf.stage1.2.bias", (256,)),
    ("nerf.stage1.4.weight", (256, 256)), ("nerf.stage1.4.bias", (256,)),
    ("nerf.stage1.6.weight", (256, 256)), ("nerf.stage1.6.bias", (256,)),
    ("nerf.stage2.0.weight", (256, 319)), ("nerf.stage2.0.bias", (256,)),
    ("nerf.stage2.2.weight", (256, 256)), ("nerf.stage2.2.bias", (256,)),
    ("nerf.stage2.4.weight", (256, 256)), ("nerf.stage2.4.bias", (256,)),
    ("nerf.density_net.0.weight", (1, 256)), ("nerf.density_net.0.bias", (1,)),
    ("nerf.rgb_net.1.weight", (128, 256)), ("nerf.rgb_net.1.bias", (128,)),
    ("nerf.rgb_net.3.weight", (3, 128)), ("nerf.rgb_net.3.bias", (3,)),
    ("lighting_mlp.lights_encoding.0.weight", (128, 9)), ("lighting_mlp.lights_encoding.0.bias", (128,)),
    ("lighting_mlp.lights_encoding.2.weight", (128, 128)), ("lighting_mlp.lights_encoding.2.bias", (128,)),
    ("lighting_mlp.lights_encoding.4.weight", (1, 128)), ("lighting_mlp.lights_encoding.4.bias", (1,)),
    ("pose_mlp.0.weight", (64, 92)), ("pose_mlp.0.bias", (64,)),
    ("pose_mlp.2.weight", (64, 64)), ("pose_mlp.2.bias", (64,)),
    ("pose_mlp.4.weight", (16, 64)), ("pose_mlp.4.bias", (16,)),
]


def make_state_dict(seed: int = 11, gain: float = 1.6) -> dict:
    """33 float32 numpy arrays keyed like DualSpaceNeRF.state_dict().

    Values: U(-b, b) with b = gain/sqrt(fan_in) (torch.nn.Linear's default is gain=1;
    gain>1 keeps activations alive through 8 ReLU layers so the field is non-trivial);
    embedding ~ N(0,1) like nn.Embedding.  The density head is rescaled (weight x40,
    bias 0.5; colour head x3, bias 0.4) so that sigma spans roughly [-3, 12] and colours
    [-0.2, 1] on the synthetic body - with default init sigma <= 0 almost everywhere and
    every image is black (SURVEY.md 8c caveat 1).
    """
    sd = {}
    for i, (name, shape) in enumerate(STATE_SHAPES):
        n = int(np.prod(shape))
        if name == "nerf.embedding.weight":
            w = hash_normal(n, seed * 1000 + i)
        else:
            fan_in = shape[1] if len(shape) == 2 else dict(STATE_SHAPES)[name.replace("bias", "weight")][1]
            b = gain / np.sqrt(fan_in)
            w = (hash_uniform(n, seed * 1000 + i) * 2.0 - 1.0) * np.float32(b)
        sd[name] = w.reshape(shape).astype(np.float32)
    sd["nerf.density_net.0.weight"] = sd["nerf.density_net.0.weight"] * np.float32(40.0)
    sd["nerf.density_net.0.bias"] = np.full((1,), 0.5, np.float32)
    sd["nerf.rgb_net.3.weight"] = sd["nerf.rgb_net.3.weight"] * np.float32(3.0)
    sd["nerf.rgb_net.3.bias"] = np.full((3,), 0.4, np.float32)
    return sd


def make_poses(seed: int = 5) -> np.ndarray:
    """SMPL axis-angle pose [24,3] ~ N(0, 0.2^2) (batch['poses'] without the leading 1)."""
    return (hash_normal(72, seed) * np.float32(0.2)).reshape(24, 3).astype(np.float32)


def make_joint_transforms(seed: int = 303) -> np.ndarray:
    """24 rigid joint transforms [24,4,4] (Rodrigues rotations up to ~0.8 rad + small translations), float32."""
    r = (hash_uniform(72, seed).reshape(24, 3) - 0.5) * 0.9
    t = (hash_uniform(72, seed + 1).reshape(24, 3) - 0.5) * 0.3
    A = np.zeros((24, 4, 4), np.float64)
    for j in range(24):
        th = np.linalg.norm(r[j]) + 1e-12
        k = r[j] / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        A[j, :3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        A[j, :3, 3] = t[j]
        A[j, 3, 3] = 1
    return A.astype(np.float32)


def make_skin_weights(V: int, seed: int = 302) -> np.ndarray:
    """[V,24] rows on the simplex with a few dominant joints each (the shape of SMPL's blend weights), float32."""
    u = hash_uniform(V * 24, seed).reshape(V, 24).astype(np.float64)
    w = np.exp(12.0 * u)
    return (w / w.sum(1, keepdims=True)).astype(np.float32)

