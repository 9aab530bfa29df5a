"""TEST INFRASTRUCTURE - runs ONLY in the build container (needs /root/reference).

Imports the reference implementation (zyhbili/Dual-Space-NeRF, read-only at
/root/reference) UNMODIFIED, with three harness-side shims, and drives its own functions
to produce golden input/output vectors for the volume-rendering hot path.  Nothing in
here is shipped or imported by the product; the GPU box never sees /root/reference.

Shims:
  * ``pytorch3d.ops.knn_points`` (pytorch3d==0.4.0 is an un-vendored dependency,
    /root/reference/requirements.txt:56): restated from its published contract - squared
    L2 distance accumulated per coordinate, K smallest ascending, lowest index on ties.
    PARITY UNPINNED for this one call: the reference holds no test that pins it.
  * ``.cuda()`` -> identity (no GPU in the build container).
  * ``can_render.load_bodydata`` -> synthetic closed mesh (SMPL pkl is licensed).
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np

REF = "/root/reference"


def install_shims():
    import torch

    sys.dont_write_bytecode = True
    if "pytorch3d" not in sys.modules:
        p3d = types.ModuleType("pytorch3d")
        ops = types.ModuleType("pytorch3d.ops")

        def knn_points(p1, p2, K=1, return_nn=False, **kw):
            # p1 [B,N,3], p2 [B,M,3].  pytorch3d's kernel accumulates
            #   dist = 0; for d in x,y,z: dist += (p1_d - p2_d)^2
            # which nvcc contracts to an fma chain; emulate that chain for float32 inputs by
            # forming each fma in float64 (the product of two float32 is exact there) and
            # rounding to float32 after every step.  K=1 -> first minimal index on ties.
            assert K == 1
            f32 = p1.dtype == torch.float32
            outs_d, outs_i = [], []
            # (no autograd graph through the distances: the reference uses the INDEX only - utils/render_utils.py:95-99 gathers the
            #  face by idx, `dist` is dropped - and a graph over [N, F] chunks does not fit the container at 8192 x 64 samples)
            p1, p2 = p1.detach(), p2.detach()
            for b in range(p1.shape[0]):
                dd, ii = [], []
                for s in range(0, p1.shape[1], 2048):
                    q = p1[b, s:s + 2048]
                    diff = q[:, None, :] - p2[b][None, :, :]
                    if f32:
                        df = diff.double()
                        d2 = (df[..., 0] * df[..., 0]).float()
                        d2 = (df[..., 1] * df[..., 1] + d2.double()).float()
                        d2 = (df[..., 2] * df[..., 2] + d2.double()).float()
                    else:
                        d2 = diff[..., 0] * diff[..., 0]
                        d2 = d2 + diff[..., 1] * diff[..., 1]
                        d2 = d2 + diff[..., 2] * diff[..., 2]
                    v, i = d2.min(dim=1, keepdim=True)
                    dd.append(v)
                    ii.append(i)
                outs_d.append(torch.cat(dd))
                outs_i.append(torch.cat(ii))
            dist = torch.stack(outs_d)
            idx = torch.stack(outs_i)
            nn = torch.stack([p2[b][idx[b]] for b in range(p2.shape[0])]) if return_nn else None
            return dist, idx, nn

        def knn_gather(x, idx, lengths=None):
            return torch.stack([x[b][idx[b]] for b in range(x.shape[0])])

        ops.knn_points = knn_points
        ops.knn_gather = knn_gather
        p3d.ops = ops
        sys.modules["pytorch3d"] = p3d
        sys.modules["pytorch3d.ops"] = ops
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    if REF not in sys.path:
        sys.path.insert(0, REF)


def make_cfg(S: int, perturb: float = 1.0, raw_noise_std: float = 1.0):
    return SimpleNamespace(
        DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
        MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=perturb,
                              raw_noise_std=raw_noise_std, TYPE="nerf", FINE_RAY_SAMPLING=-1,
                              LOSS="L2", LOSSwMask=False),
    )


def build_reference(canon: np.ndarray, faces: np.ndarray, state: dict, S: int, dtype="float32"):
    """Reference Renderer + DualSpaceNeRF with the given mesh / parameters."""
    install_shims()
    import torch
    import warnings

    warnings.filterwarnings("ignore")
    import can_render
    from model.spacenet import DualSpaceNeRF

    V = canon.shape[0]
    can_render.load_bodydata = lambda *a, **k: {
        "f": faces.astype(np.uint32),
        "weights": np.zeros((V, 24), np.float32),
        "kintree_table": np.stack([np.arange(-1, 23), np.arange(24)]).astype(np.int64),
    }
    cfg = make_cfg(S)
    net = DualSpaceNeRF(cfg)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    tdt = getattr(torch, dtype)
    cv = torch.from_numpy(canon.copy()).to(tdt)
    if dtype == "float64":
        net.double()
        net.pose_mlp.float()  # spacenet.py:223 forces poses to float32
    import io
    import contextlib

    with contextlib.redirect_stdout(io.StringIO()):
        render = can_render.Renderer(net, None, cfg, cv)
    return render


def make_batch(rays: dict, xyz, poses, Th, frame: int, dtype="float32", sel=None, H=None, W=None):
    import torch

    tdt = getattr(torch, dtype)
    if sel is None:
        sel = np.arange(rays["ray_o"].shape[0])
    b = {
        "ray_o": torch.from_numpy(rays["ray_o"][sel].copy()).to(tdt)[None],
        "ray_d": torch.from_numpy(rays["ray_d"][sel].copy()).to(tdt)[None],
        "near": torch.from_numpy(rays["near"][sel].copy()).to(tdt)[None],
        "far": torch.from_numpy(rays["far"][sel].copy()).to(tdt)[None],
        "xyz": torch.from_numpy(xyz.copy()).to(tdt)[None],
        "poses": torch.from_numpy(poses.copy())[None],
        "Th": torch.from_numpy(np.asarray(Th, np.float32).reshape(1, 1, 3).copy()).to(tdt),
        "frame": torch.tensor([frame], dtype=torch.int64),
    }
    return b


def run_stages(render, batch, train: bool, seed: int = 233):
    """Drive the reference's own hot-path functions stage by stage (same calls, same order as
    Renderer.render, can_render.py:137-168) and record every intermediate."""
    import torch
    from utils.pts_utils import geometry_guided_ray_marching
    from utils.render_utils import get_closest_mesh, get_transparent_mask
    from utils.geo_utils import project_point2mesh
    from utils.nerf_net_utils import raw2outputs
    from model.spacenet import batch_rod2quat, gradient, normal_local2world

    out = {}
    net = render.net
    if train:
        render.train()
    else:
        render.eval()
    cfg = render.cfg
    S = cfg.MODEL.COARSE_RAY_SAMPLING
    ray_o, ray_d = batch["ray_o"], batch["ray_d"]
    near, far = batch["near"].clone(), batch["far"].clone()
    R = ray_o.shape[1]
    tdt = ray_o.dtype
    if train:
        torch.manual_seed(seed)
        jitter = torch.rand(1, R, S)
        noise = torch.randn(R, S)
        out["jitter"], out["noise"] = jitter.numpy(), noise.numpy()
        torch.manual_seed(seed)  # the reference now draws the same two tensors, same order
    pts, z_vals = geometry_guided_ray_marching(ray_o, ray_d, S, near, far, batch["xyz"],
                                               cfg.MODEL.perturb, net.training)
    out["near_gg"], out["far_gg"] = near[0].numpy(), far[0].numpy()
    out["z_vals"], out["pts"] = z_vals[0].numpy(), pts[0].numpy()

    # warp: pieces first (to record idx / uv / h), then the reference's own w2l
    meshes = batch["xyz"][:, render.face_idx]
    cm, idx = get_closest_mesh(pts.reshape(1, -1, 3), meshes)
    uv, h = project_point2mesh(pts.reshape(-1, 3), meshes=cm.reshape(-1, 3, 3))
    out["idx_world"] = idx.reshape(-1).numpy().astype(np.int32)
    out["uv"], out["h"] = uv.numpy(), h.numpy()
    out["centroid_world"] = meshes.mean(dim=-2)[0].numpy()
    pts6, rays6, transparent = render.w2l(pts, ray_o, ray_d, batch)
    out["transparent"] = transparent.reshape(-1).numpy()
    out["x_c"] = pts6[..., 3:].reshape(-1, 3).detach().numpy().copy()
    out["ray_d_can"] = rays6[..., 3:].reshape(-1, 3).numpy()

    batch["transparent_mask"] = transparent.reshape(-1, S)
    batch["canonical_model"] = render.canonical_model
    batch["face_idx"] = render.face_idx
    frame_idx = batch["frame"][..., None, None].repeat(1, R, S).reshape(-1, S)

    # network, piecewise (model/spacenet.py:210-266)
    pos = pts6.reshape(-1, 6).clone()
    rays = rays6.reshape(-1, 6)
    xyz_cano = pos[..., 3:]
    xyz_cano.requires_grad = True
    body_pose = batch["poses"][0][1:, :].float()
    q = batch_rod2quat(body_pose.reshape(-1, 3)).reshape(1, -1)
    out["pose_quat"] = q.numpy()
    pf1 = net.pose_mlp(q)
    out["pose_feat"] = pf1.detach().numpy()
    pose_feat = net.pose_mlp(q.repeat(xyz_cano.shape[0], 1))
    essence, density, _ = net.nerf(xyz_cano, rays, frame_idx, False, pose_feat)
    out["sigma"] = density.detach().reshape(-1).numpy()
    out["essence"] = essence.detach().numpy()
    g = gradient(xyz_cano, density)
    out["grad_sigma"] = g.detach().numpy()
    cmc, idxc = get_closest_mesh(xyz_cano[None].detach(), render.canonical_model["meshes"][None])
    out["idx_canon"] = idxc.reshape(-1).numpy().astype(np.int32)
    n_w = normal_local2world(g, xyz_cano, batch)
    out["n_w"] = n_w.detach().numpy()

    # full forward through the reference module (includes rot / light-centre variants)
    pos2 = pts6.reshape(-1, 6).clone().detach()
    color, density2, _ = net(pos2, rays, frame_idx, batch_info=batch)
    out["colour"] = color.detach().numpy()
    assert np.array_equal(density2.detach().reshape(-1).numpy(), out["sigma"])

    raw = torch.cat([color, density2], -1).reshape(R, S, -1).detach()
    t = raw[batch["transparent_mask"]]
    t[..., -1] = 0
    raw[batch["transparent_mask"]] = t
    out["raw"] = raw.numpy().copy()
    if train:
        torch.manual_seed(seed)
        _ = torch.rand(1, R, S)  # keep the stream aligned: rand then randn
    rgb_map, disp, acc, weights, depth, _ = raw2outputs(
        raw, z_vals.reshape(-1, S), rays6[:, 0, :3], cfg.MODEL.raw_noise_std if train else 0, False)
    out.update(rgb_map=rgb_map.numpy(), disp_map=disp.numpy(), acc_map=acc.numpy(),
               weights=weights.numpy(), depth_map=depth.numpy())
    return out


def run_render(render, batch, train: bool, seed: int = 233, target=None, grad_params=()):
    """End-to-end Renderer.render (the trainer's call, trainer.py:70)."""
    import torch

    if train:
        render.train()
        torch.manual_seed(seed)
    else:
        render.eval()
    b = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in batch.items()}
    ret = render.render(b)["coarse"]
    out = {k: v.detach().numpy() for k, v in ret.items()}
    if target is not None:
        loss = torch.nn.functional.mse_loss(ret["color"], torch.from_numpy(target).to(ret["color"]))
        render.net.zero_grad()
        loss.backward()
        out["loss"] = np.float64(loss.item())
        sd = dict(render.net.named_parameters())
        for name in grad_params:
            out["gradnorm:" + name] = np.float64(sd[name].grad.norm().item())
    return out
