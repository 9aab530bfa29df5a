"""Host-side mirror of the reference's Renderer (the upper drop-in boundary).

Same constructor, attributes and methods as /root/reference/can_render.py:14-406 so that
trainer.py:70 (`render.render(batch)`), test.py:60 / validate.py:50 (`render.render_view(batch)`),
utils/visualizer.py:47-66 (`w2l_without_lbs`, `query_volume`, `.canonical_model`) and
validate.py:27 (`render.net.load_state_dict`) can use it unchanged.  Orchestration is Python; every
number is produced by libdsnerf_hip.so (no torch math, no fallback).

What differs from the reference, on purpose (DESIGN.md "boundary"):
  * a frame is rendered in one pass over ALL its rays (the reference loops over 3072-ray chunks with
    an empty_cache + 6 device->host copies each, can_render.py:172-245); one device->host copy at the end;
  * in eval mode the networks are evaluated only on non-transparent samples (their sigma is zeroed and
    their colour has weight 0 in the reference, can_render.py:115-120) - outputs are identical;
  * `render()` in train mode returns tensors attached to ONE autograd node whose backward is
    dsn_render_rays_grad (analytic parameter gradients, csrc/dsn_train.hip) instead of an op-by-op graph.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def load_bodydata(model_type="smpl", gender="neutral", model_path=""):
    """utils/smpl_utils.py:3-14: unpickle the SMPL model (dict with 'f', 'weights', 'kintree_table')."""
    import os
    import pickle

    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, "{}_{}.pkl".format(model_type.upper(), gender.upper()))
    assert os.path.exists(model_path), "Path {} does not exist!".format(model_path)
    with open(model_path, "rb") as f:
        return pickle.load(f, encoding="latin1")


_OUT_KEYS = ("color", "disp_map", "acc_map", "depth_map", "weights", "z_vals")


class _RenderRays(torch.autograd.Function):
    """render_rays as one differentiable node: forward = dsn_render_rays, backward = dsn_render_rays_grad
    (what loss.backward() does in trainer.py:70-81).  Inputs that are not parameters carry no gradient, as in the
    reference (rays, near/far, xyz, poses are data)."""

    @staticmethod
    def forward(ctx, renderer, call, *params):
        o, d, near, far, S, jitter, noise, uniform, frame_args = call
        if not hasattr(renderer, "_grad_ws"):
            renderer._grad_ws = _lib.GradWorkspace(renderer.device)
        renderer._cache_gen = getattr(renderer, "_cache_gen", 0) + 1      # whose activations the cache holds
        out = _lib.render_rays(renderer.scene, renderer.net.packed(renderer.device), renderer._ws, o, d, near, far, S,
                               renderer._t_vals(S), jitter, noise, skip_transparent=False, uniform=uniform,
                               train_cache=renderer._grad_ws)
        ctx.renderer, ctx.call, ctx.params, ctx.gen = renderer, (o, d, noise, frame_args), params, renderer._cache_gen
        ctx.z_vals = out["z_vals"]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out["z_vals"])
        return tuple(out[k] for k in _OUT_KEYS)

    @staticmethod
    def backward(ctx, g_color, g_disp, g_acc, g_depth, g_weights, g_z):
        r = ctx.renderer
        o, d, noise, frame_args = ctx.call
        xyz, poses, frame, zero_code, ls, rot, rc = frame_args
        params = [p.detach() for p in ctx.params]
        sd = dict(zip(_lib.PARAM_ORDER, params))
        packed = r.net.packed(r.device)
        r.scene.set_frame(packed, xyz, poses, frame, zero_code, ls, rot, rc, reuse=True)   # no-op unless another frame was rendered since
        if g_color is None:
            g_color = torch.zeros(o.shape[0], 3, device=r.device)
        if not hasattr(r, "_grad_ws"):
            r._grad_ws = _lib.GradWorkspace(r.device)
        grads = _lib.render_rays_grad(r.scene, sd, poses, frame, zero_code, o, d, ctx.z_vals, noise, g_color, g_disp, g_acc,
                                      g_depth, g_weights, ws=r._grad_ws, packed=packed,
                                      cached=(ctx.gen == getattr(r, "_cache_gen", -1)))
        grads = [g.to(device=p.device, dtype=p.dtype).reshape(p.shape) for g, p in zip(grads, ctx.params)]
        return (None, None) + tuple(grads)


class Renderer:
    def __init__(self, net, fine_net=None, cfg=None, canonical_vertex=None, body_data=None, device=None):
        """`body_data` (optional, not in the reference): dict with 'f' [F,3] (and optionally 'weights',
        'kintree_table') used instead of unpickling cfg.DATASETS.SMPL_PATH - the SMPL file is licensed."""
        _lib.require_gpu()
        self.net = net
        self.cfg = cfg
        self.fine_net = fine_net
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.canonical_vertex = canonical_vertex
        self._body_data = body_data
        self.load_body_model(gender="neutral", body_model="smpl", model_path=getattr(cfg.DATASETS, "SMPL_PATH", ""))
        self.sample_points_mode = cfg.MODEL.sample_points_mode
        self._ws = _lib.RenderWorkspace(self.device)
        self._tvals = {}
        self.skip_transparent = True
        self.last_active_fraction = None

    # ---- mode switches (reference :26-38) ----
    def train(self):
        self.net.training = True
        self.net.train()
        if self.fine_net is not None:
            self.fine_net.training = True
            self.fine_net.train()

    def eval(self):
        self.net.training = False
        self.net.eval()
        if self.fine_net is not None:
            self.fine_net.training = False
            self.fine_net.eval()

    # ---- body model (reference :382-406) ----
    def load_body_model(self, gender, body_model, model_path):
        tmp = self._body_data if self._body_data is not None else load_bodydata(body_model, gender, model_path)
        if "kintree_table" in tmp:
            parents = torch.as_tensor(np.asarray(tmp["kintree_table"])[0].astype(np.int64)).long()
            parents[0] = -1
            self.parents = parents
        if "weights" in tmp:
            self.smpl_blend_weight = torch.as_tensor(np.asarray(tmp["weights"], np.float32))[None].to(self.device)
        self.face_idx = torch.as_tensor(np.asarray(tmp["f"]).astype(np.int64)).long().to(self.device)
        if self.canonical_vertex is not None:
            cv = torch.as_tensor(self.canonical_vertex, dtype=torch.float32).reshape(-1, 3).to(self.device)
            self.canonical_model = {"vertex": cv, "meshes": cv[self.face_idx]}
            self.scene = _lib.Scene(cv, self.face_idx, self.device)

    # ---- helpers ----
    def _t_vals(self, S):
        # torch.linspace on the host, exactly as utils/pts_utils.py:4 does, then uploaded once
        if S not in self._tvals:
            self._tvals[S] = torch.linspace(0.0, 1.0, steps=S).to(self.device)
        return self._tvals[S]

    def _dev(self, t, dtype=torch.float32):
        """Tensor on the device.  Host tensors of a batch (the reference's DataLoader hands over pageable CPU tensors and calls
        .cuda() on them, can_render.py:100-103,138-141) go through a small ring of persistent page-locked staging buffers and an
        asynchronous copy: a pageable .to(device) of a few MB costs 10-20 ms on the GPU boxes (the runtime pins and unpins the
        pages and waits for the stream), 70 ms per 512 x 512 batch (scripts/render_view_probe.py)."""
        if t.is_cuda or t.numel() * t.element_size() < (1 << 16) or t.is_pinned() or self.device.type != "cuda":
            return t.to(device=self.device, dtype=dtype).contiguous()
        n = t.numel()
        ring = getattr(self, "_stage_ring", None)
        if ring is None:
            ring = self._stage_ring = {"i": 0, "slots": [{"host": None, "done": torch.cuda.Event()} for _ in range(8)]}
        slot = ring["slots"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot["done"].synchronize()                       # the copy that last read this staging buffer has finished
        nbytes = n * dtype.itemsize
        if slot["host"] is None or slot["host"].numel() < nbytes:
            slot["host"] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8).pin_memory()
        stage = slot["host"][:nbytes].view(dtype).view(t.shape)
        stage.copy_(t)                                   # host memcpy (with the dtype conversion, if any)
        out = torch.empty(t.shape, dtype=dtype, device=self.device)
        out.copy_(stage, non_blocking=True)
        slot["done"].record()
        return out

    def _set_frame(self, batch):
        frame = int(torch.as_tensor(batch["frame"]).reshape(-1)[0])
        zero_code, ls, rot, rc = self.net.frame_args(batch)
        self.scene.set_frame(self.net.packed(self.device), self._dev(batch["xyz"][0]), batch["poses"][0], frame, zero_code, ls,
                             rot, rc)
        return frame

    def _draws(self, R, S):
        """Train-mode random draws from the CPU default generator in the reference's order:
        torch.rand([1,R,S]) (utils/pts_utils.py:12) then torch.randn([R,S]) (utils/nerf_net_utils.py:31)."""
        want_j = self.net.training and self.cfg.MODEL.perturb > 0.0
        want_n = self.net.training and self.cfg.MODEL.raw_noise_std > 0.0
        if not (want_j or want_n):
            return None, None
        if self.device.type != "cuda":
            raise RuntimeError("dsnerf_amd renders on the GPU only")
        # The draws go straight into a small ring of persistent page-locked staging buffers (out=: same generator
        # stream as a fresh tensor) and are uploaded with one asynchronous copy.  Fresh 2 MB host tensors cost 10-80 ms
        # per step on the GPU boxes (page faults of newly mapped memory + the runtime pinning them for the pageable
        # copy, which also blocks until the stream has drained): scripts/train_host_probe.py.
        n = 2 * R * S
        ring = getattr(self, "_draw_ring", None)
        if ring is None or ring["n"] < n:
            ring = self._draw_ring = {"n": n, "i": 0, "slots": [
                {"host": torch.empty(n, dtype=torch.float32).pin_memory(), "dev": torch.empty(n, device=self.device),
                 "done": torch.cuda.Event()} for _ in range(3)]}
        slot = ring["slots"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot["done"].synchronize()            # the copy that last read this staging buffer (three draws ago) has finished
        k = 0
        if want_j:
            torch.rand(1, R, S, out=slot["host"][:R * S].view(1, R, S))
            k = R * S
        m = k
        if want_n:
            hn = slot["host"][k:k + R * S].view(R, S)
            torch.randn(R, S, out=hn)
            if float(self.cfg.MODEL.raw_noise_std) != 1.0:
                hn.mul_(self.cfg.MODEL.raw_noise_std)
            m = k + R * S
        slot["dev"][:m].copy_(slot["host"][:m], non_blocking=True)
        slot["done"].record()
        # device-side clones (two 5 us copies): autograd may keep the draws alive for longer than the ring does
        jitter = slot["dev"][:R * S].view(R, S).clone() if want_j else None
        noise = slot["dev"][k:k + R * S].view(R, S).clone() if want_n else None
        return jitter, noise

    # ---- sampling (reference :40-63) ----
    def get_sampling_points(self, ray_o, ray_d, near, far, xyz, mode="GG"):
        """ray_o/ray_d [1,R,3], near/far [1,R] (updated in place in GG mode), xyz [1,V,3] ->
        pts [1,R,S,3], z_vals [1,R,S].  Needs the frame's xyz in the scene: Renderer.render does that;
        direct callers get it set here from `xyz` with the current pose state untouched."""
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        R = ray_o.shape[1]
        o, d = self._dev(ray_o[0]), self._dev(ray_d[0])
        n_dev, f_dev = self._dev(near[0]), self._dev(far[0])
        jitter = None
        if self.net.training and self.cfg.MODEL.perturb > 0.0:
            jitter = torch.rand(1, R, S).reshape(R, S).to(self.device)
        if getattr(self, "_frame_xyz_ptr", None) != xyz.data_ptr():
            self._upload_xyz(xyz)
        pts, z = _lib.sample(self.scene, o, d, n_dev, f_dev, S, self._t_vals(S), jitter, want_pts=True, gg=(mode == "GG"))
        if mode == "GG":   # in-place update like the reference (:52-53)
            near[0].copy_(n_dev.to(near.device))
            far[0].copy_(f_dev.to(far.device))
        return pts[None], z[None]

    def _upload_xyz(self, xyz):
        # posed mesh only (callers without a full batch: the pose code is zeroed until _set_frame runs)
        x = self._dev(xyz.reshape(-1, 3))
        poses = torch.zeros(24, 3, device=self.device)
        self.scene.set_frame(self.net.packed(self.device), x, poses, 0, True, None, None, None)
        self._frame_xyz_ptr = xyz.data_ptr()

    # ---- warp (reference :299-379) ----
    def w2l(self, pts_world, ray_o_W, ray_d_W, batch):
        B, ray, sp, _ = pts_world.shape
        pts_smpl_can, transparent_mask, ray_d_can = self.w2l_without_lbs(
            pts_world, batch, self.canonical_model, ray_d_W=ray_d_W.unsqueeze(2).expand([-1, -1, sp, -1]).reshape(B, -1, 3))
        d = self._dev(ray_d_W.reshape(-1, 3))
        rays = torch.cat([d[:, None, :].expand(-1, sp, -1).reshape(-1, 3), ray_d_can], dim=-1).reshape(B * ray, sp, 6)
        pw = self._dev(pts_world.reshape(B * ray, sp, 3))
        return torch.cat([pw, pts_smpl_can.reshape(-1, sp, 3)], dim=-1), rays, transparent_mask

    def w2l_without_lbs(self, pts_world, batch, canonical_model, ray_d_W=None, floor=-4, ceil=5):
        """pts_world [B,R,S,3] -> (pts_smpl_can [N,3], transparent_mask [B,N] bool[, ray_d_can [N,3]]).
        ray_d_W, when given, is the per-POINT direction tensor [B,N,3] the reference passes."""
        assert floor == -4 and ceil == 5, "the uv clamp range is compiled in (utils/render_utils.py:103)"
        B, ray, sp, _ = pts_world.shape
        xyz = batch["xyz"]
        if getattr(self, "_frame_xyz_ptr", None) != xyz.data_ptr():
            if "poses" in batch and "frame" in batch:
                self._set_frame(batch)
                self._frame_xyz_ptr = xyz.data_ptr()
            else:
                self._upload_xyz(xyz)
        pts = self._dev(pts_world.reshape(-1, 3))
        N = pts.shape[0]
        if ray_d_W is not None:
            d = self._dev(ray_d_W.reshape(-1, 3))   # per point -> S=1 addressing
            out = _lib.warp(self.scene, pts, d, 1, want_dir=True)
            return out["x_c"], out["transparent"].bool().reshape(B, -1), out["ray_d_can"]
        out = _lib.warp(self.scene, pts, None, 1, want_dir=False)
        return out["x_c"], out["transparent"].bool().reshape(B, -1)

    # ---- network + compositing on explicit points (reference :65-134) ----
    def render_rays(self, pts, rays, z_vals, frame_idx, net, transparent_mask=None, batch_info=None):
        pts, rays, z_vals = self._dev(pts), self._dev(rays), self._dev(z_vals)
        rays_d = rays[:, 0, :3].contiguous()
        B, sp = pts.shape[:2]
        noise = None
        if self.net.training and self.cfg.MODEL.raw_noise_std > 0.0:
            noise = (torch.randn(B, sp) * self.cfg.MODEL.raw_noise_std).to(self.device)
        rgbs, density, _ = net(pts.reshape(-1, 6), rays.reshape(-1, 6), frame_idx, batch_info=batch_info)
        tm = None if transparent_mask is None else transparent_mask.to(self.device).reshape(B, sp).to(torch.uint8).contiguous()
        rgb_map, disp_map, acc_map, weights, depth_map = _lib.composite(
            rgbs.reshape(B, sp, 3).contiguous(), density.reshape(B, sp).contiguous(), tm, z_vals, rays_d, noise)
        return {"color": rgb_map, "disp_map": disp_map, "acc_map": acc_map, "depth_map": depth_map,
                "weights": weights, "z_vals": z_vals}

    def batchify_pts(self, pts, rays, z_vals, frame_idx, chunk=1024 * 32, net=None, batch_info=None):
        net = self.net if net is None else net
        return self.render_rays(pts, rays, z_vals, frame_idx, net=net,
                                transparent_mask=batch_info["transparent_mask"], batch_info=batch_info)

    # ---- the trainer's call (reference :137-168) ----
    def render(self, batch):
        o, d = self._dev(batch["ray_o"][0]), self._dev(batch["ray_d"][0])
        near, far = self._dev(batch["near"][0]), self._dev(batch["far"][0])
        R = o.shape[0]
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        self._set_frame(batch)
        self._frame_xyz_ptr = batch["xyz"].data_ptr()
        jitter, noise = self._draws(R, S)
        if self.sample_points_mode not in ("GG", "uniform"):
            raise Exception("error")   # the reference fails on unknown modes too (get_sampling_points returns nothing)
        uniform = self.sample_points_mode == "uniform"
        sd = dict(self.net.named_parameters())
        if self.net.training and torch.is_grad_enabled() and any(p.requires_grad for p in sd.values()):
            frame = int(torch.as_tensor(batch["frame"]).reshape(-1)[0])
            frame_args = (batch["xyz"][0], batch["poses"][0], frame) + tuple(self.net.frame_args(batch))
            outs = _RenderRays.apply(self, (o, d, near, far, S, jitter, noise, uniform, frame_args),
                                     *[sd[k] for k in _lib.PARAM_ORDER])
            out = dict(zip(_OUT_KEYS, outs))
        else:
            out = _lib.render_rays(self.scene, self.net.packed(self.device), self._ws, o, d, near, far, S, self._t_vals(S),
                                   jitter, noise, skip_transparent=self.skip_transparent and not self.net.training,
                                   uniform=uniform)
        if batch["near"].is_cuda:   # in-place semantics of the reference when the batch already lives on the device
            batch["near"][0].copy_(near)
            batch["far"][0].copy_(far)
        batch["canonical_model"] = self.canonical_model
        batch["face_idx"] = self.face_idx
        return {"coarse": out}

    # ---- whole-image path (reference :172-278) ----
    def batchify_rays_view(self, ray_o, ray_d, near, far, batch, chunk=None):
        o, d = self._dev(ray_o[0]), self._dev(ray_d[0])
        n, f = self._dev(near[0]).clone(), self._dev(far[0]).clone()
        S = self.cfg.MODEL.COARSE_RAY_SAMPLING
        self._set_frame(batch)
        self._frame_xyz_ptr = batch["xyz"].data_ptr()
        R = o.shape[0]
        chunk = R if chunk is None else int(chunk)
        outs = []
        for i in range(0, R, chunk):
            j = min(R, i + chunk)
            jitter, noise = self._draws(j - i, S)
            outs.append(_lib.render_rays(self.scene, self.net.packed(self.device), self._ws, o[i:j].contiguous(),
                                         d[i:j].contiguous(), n[i:j].contiguous(), f[i:j].contiguous(), S,
                                         self._t_vals(S), jitter, noise,
                                         skip_transparent=self.skip_transparent and not self.net.training,
                                         uniform=(self.sample_points_mode == "uniform")))
        coarse = {k: torch.cat([x[k] for x in outs], 0) for k in outs[0]}
        return coarse, {}

    def render_view(self, batch, chunk=None, device_output=False):
        """device_output=True (not in the reference) keeps the [H,W,*] images on the GPU - for multi-frame sequences
        (novel_pose_vis.py:41-66) and on-device metrics (`image_metrics`)."""
        coarse, _ = self.batchify_rays_view(batch["ray_o"], batch["ray_d"], batch["near"], batch["far"], batch, chunk)
        _, H, W, _ = batch["img"].shape
        # utils/render_utils.py:466-472 post_process, done on the device (dsn_image_scatter)
        img = _lib.image_scatter(coarse, self._dev(batch["mask_at_box"][0], torch.uint8), H, W)
        if device_output:
            return img
        # four contiguous device images -> four device->host copies (0.1-0.3 ms each).  Packing them into one [H,W,6] copy and
        # splitting the channels on the host cost 19-25 ms per frame on the GPU boxes (strided CPU copies; scripts/d2h_probe.py)
        # (fresh host tensors like the reference's; on the GPU boxes first-touch page faults of new host memory make this step
        # vary between 1 and 20 ms per frame - `device_output=True` avoids it)
        return {k: img[k].contiguous().cpu() for k in ("coarse_color", "coarse_disp", "coarse_acc", "coarse_depth")}

    def image_metrics(self, color_img, batch, clamp=True):
        """test.py:62-71 on the device: clamp to [0,1], psnr with and without mask_at_box against batch["img"].
        Returns a dict of python floats (one 32-byte device->host copy)."""
        img = color_img.to(self.device)
        if clamp:
            img = torch.clamp(img, min=0.0, max=1.0)
        gt = batch["img"][0]
        gt = self._dev(gt, gt.dtype if gt.dtype in (torch.float32, torch.float64) else torch.float32)   # staged upload (6 MB of float64)
        m = _lib.image_psnr(img, gt, self._dev(batch["mask_at_box"][0], torch.uint8)).cpu()
        return {"mse": float(m[0]), "mse_wMask": float(m[1]), "psnr_woMask": float(m[2]), "psnr_wMask": float(m[3])}

    # ---- density query for marching cubes (reference :280-296) ----
    def query_volume(self, pts, code_idx, transparent_mask=None, batch_info={}):
        B, N = pts.shape[:2]
        density = self.net(pts, None, code_idx, batch_info, density_only=True)
        if transparent_mask is not None:
            density[transparent_mask.to(density.device).reshape([-1, 1])] = 0
        return density.reshape(B, N, 1)
