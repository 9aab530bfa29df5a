"""train_oracle.py - TEST INFRASTRUCTURE ONLY (never imported by the product).

A differentiable CPU restatement of Renderer.render for the training row: the geometry that
does not depend on the parameters (sample points, nearest faces, canonical points, transparency) comes from the C
oracle (oracle/dsn_oracle.c, bit-exact against the reference's golden vectors); everything the parameters flow
through is restated here with torch CPU ops in the reference's order, so that torch.autograd produces the same
parameter gradients as the reference's loss.backward() (trainer.py:70-81), including the double-backward path
d sigma/dx -> normal -> lighting (model/spacenet.py:243-265, 278-311).  Pinned against gradients captured from the
reference itself: tests/golden/*_grads.npz (tests/test_oracle_golden.py).

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np
import torch

import oracle as O


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def rod2quat(poses, dtype):
    """model/spacenet.py:314-331 on joints 1..23 (pose row 0 is dropped, :224)."""
    r = poses[1:].to(torch.float32)
    angle = torch.norm(r + 1e-16, p=2, dim=1, keepdim=True)
    half = angle / 2
    q = torch.cat([r / angle * torch.sin(half), torch.cos(half) - 1], dim=1)
    return q.reshape(1, 92)


def encode(x):
    """model/dimension_kernel.py:34-35,56-75: [x, sin(2^j x), cos(2^j x)] for j < 10."""
    outs = [x]
    for j in range(10):
        outs += [torch.sin(x * float(2 ** j)), torch.cos(x * float(2 ** j))]
    return torch.cat(outs, dim=-1)


def linear(p, prefix, x):
    return torch.nn.functional.linear(x, p[prefix + ".weight"], p[prefix + ".bias"])


def face_frames(verts, faces, dtype):
    """Per-face constants of utils/geo_utils.py:96-113,138-156,181-200 (float32 values of the mesh, cast)."""
    v = torch.from_numpy(np.ascontiguousarray(verts)).to(dtype)
    f = torch.from_numpy(np.asarray(faces, np.int64))
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    n = torch.nn.functional.normalize(torch.cross(e1, e2, dim=-1), dim=-1)
    return v0, e1, e2, n


def project(p, fr):
    """utils/geo_utils.py:181-200 + :96-113: (u along v2-v0, v along v1-v0, signed height)."""
    v0, e1, e2, n = fr
    h = ((p - v0) * n).sum(-1)
    w = p - h[:, None] * n - v0
    d00, d01, d11 = (e2 * e2).sum(-1), (e2 * e1).sum(-1), (e1 * e1).sum(-1)
    d02, d12 = (e2 * w).sum(-1), (e1 * w).sum(-1)
    inv = 1.0 / (d00 * d11 - d01 * d01)
    return (d11 * d02 - d01 * d12) * inv, (d00 * d12 - d01 * d02) * inv, h


def embed(u, v, h, fr):
    """utils/geo_utils.py:138-156."""
    v0, e1, e2, n = fr
    return v0 + u[:, None] * e2 + v[:, None] * e1 + h[:, None] * n


def render(params: dict, g: dict, jitter_z=None, noise=None, zero_code=False, dtype=torch.float32, geom=None):
    """params: name -> torch tensor (requires_grad as wanted); g: batch arrays (ray_o, ray_d, xyz, canonical_vertex,
    faces, poses, frame) and z_vals [R,S] (the sampler's output, float32).  Returns the outputs of
    can_render.py:137-168 as torch tensors attached to the graph.
    geom (optional, used by bench.py's eager-torch baseline): precomputed parameter-independent geometry as torch
    tensors {x_c [N,3], transparent [N] bool, idx_canon [N] int64} on the device the parameters live on; the
    nearest-face searches are then NOT part of what runs here."""
    z = np.ascontiguousarray(jitter_z, np.float32)
    R, S = z.shape
    o, d = np.asarray(g["ray_o"], np.float32), np.asarray(g["ray_d"], np.float32)
    pts = (o[:, None, :] + d[:, None, :] * z[:, :, None]).astype(np.float32)          # utils/pts_utils.py:14
    dev = torch.device("cpu") if geom is None else geom["x_c"].device
    _t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)   # noqa: E731
    if geom is None:
        wp = O.warp(pts.reshape(-1, 3), None, g["xyz"], g["canonical_vertex"], g["faces"])    # can_render.py:333-379
        cent = O.centroids(g["canonical_vertex"], g["faces"])
        geom = {"x_c": _t(wp["x_c"], dtype), "transparent": torch.from_numpy(wp["transparent"]),
                "idx_canon": torch.from_numpy(O.nearest_face(wp["x_c"], cent).astype(np.int64))}
    x_c = geom["x_c"].to(dtype).detach().requires_grad_(True)                           # model/spacenet.py:220
    transparent = geom["transparent"]
    N = R * S

    # model/spacenet.py:223-236 pose code, :125-129 frame code
    pose = linear(params, "pose_mlp.4", torch.relu(linear(params, "pose_mlp.2", torch.relu(
        linear(params, "pose_mlp.0", rod2quat(_t(g["poses"], torch.float32), dtype)))))).to(dtype)
    code = params["nerf.embedding.weight"][int(g["frame"])][None]
    if zero_code:
        code = code * 0
    # model/spacenet.py:93-148
    pe = encode(x_c)
    h = torch.cat([code.expand(N, -1), pe, pose.expand(N, -1)], dim=-1)
    for k in (0, 2, 4, 6):
        h = torch.relu(linear(params, f"nerf.stage1.{k}", h))
    h = torch.cat([h, pe], dim=-1)
    for k in (0, 2, 4):
        h = torch.relu(linear(params, f"nerf.stage2.{k}", h))
    sigma = linear(params, "nerf.density_net.0", h)
    essence = linear(params, "nerf.rgb_net.3", torch.relu(linear(params, "nerf.rgb_net.1", torch.relu(h))))
    # model/spacenet.py:301-311
    grad = torch.autograd.grad(sigma.sum(), x_c, create_graph=True)[0]
    # model/spacenet.py:278-298: nearest canonical face (a constant index), both points through the same face pair
    idx = geom["idx_canon"]
    fc = tuple(t.to(dev)[idx] for t in face_frames(g["canonical_vertex"], g["faces"], dtype))
    fw = tuple(t.to(dev)[idx] for t in face_frames(g["xyz"], g["faces"], dtype))
    start = embed(*project(x_c, fc), fw)
    end = embed(*project(x_c + grad, fc), fw)
    n_w = torch.nn.functional.normalize(end - start, dim=-1)
    # model/spacenet.py:174-188, :254-265 (light / rotation edits are inference-time only)
    x_w = _t(pts.reshape(-1, 3), dtype)
    dd = _t(d, dtype)
    view = (dd / torch.norm(dd, dim=-1, keepdim=True))[:, None, :].expand(R, S, 3).reshape(-1, 3)
    li = torch.cat([n_w, x_w, view], dim=-1)
    hl = torch.relu(linear(params, "lighting_mlp.lights_encoding.0", li))
    hl = torch.relu(linear(params, "lighting_mlp.lights_encoding.2", hl))
    wl = torch.nn.functional.elu(linear(params, "lighting_mlp.lights_encoding.4", hl)) + 1
    colour = wl * essence
    # can_render.py:115-120 + utils/nerf_net_utils.py:5-56
    sig = sigma.reshape(R, S)
    sig = torch.where(transparent.reshape(R, S), torch.zeros_like(sig), sig)
    zt = _t(z, dtype)
    dists = torch.cat([zt[:, 1:] - zt[:, :-1], torch.full((R, 1), 1e10, dtype=dtype, device=dev)], dim=-1)
    dists = dists * torch.norm(dd, dim=-1, keepdim=True)
    if noise is not None:
        sig = sig + _t(noise, dtype)
    alpha = 1.0 - torch.exp(-torch.relu(sig) * dists)
    T = torch.cumprod(torch.cat([torch.ones(R, 1, dtype=dtype, device=dev), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * T
    rgb = (weights[:, :, None] * colour.reshape(R, S, 3)).sum(1)
    depth = (weights * zt).sum(-1)
    acc = weights.sum(-1)
    disp = 1.0 / torch.max(torch.full_like(depth, 1e-10), depth / acc)
    return {"color": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth, "weights": weights,
            "sigma": sigma.reshape(-1), "essence": essence, "grad_sigma": grad, "n_w": n_w, "colour": colour}


def loss_and_grads(state: dict, g: dict, z_vals, noise, target, occupancy=None, dtype=torch.float32):
    """utils/loss.py:11-30 (L2 + 0.1 * L1 occupancy term with the in-place acc edit) and loss.backward()."""
    params = {k: torch.from_numpy(np.array(v)).to(dtype if not k.startswith("pose_mlp") else torch.float32)
              .requires_grad_(True) for k, v in state.items()}
    out = render(params, g, jitter_z=z_vals, noise=noise, dtype=dtype)
    loss = torch.nn.functional.mse_loss(out["color"], _t(target, dtype))
    if occupancy is not None:
        occ = _t(occupancy, dtype)
        acc = torch.where(occ == 1, torch.ones_like(occ), out["acc_map"])
        loss = loss + 0.1 * torch.nn.functional.l1_loss(acc, occ)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy() for k, p in params.items()}
    return float(loss.detach()), grads, {k: v.detach().numpy() for k, v in out.items()}
