"""Host-side mirror of the reference's network classes for the hot path.

Same class names, constructor arguments, method names, return conventions and state_dict layout as
/root/reference/model/spacenet.py (SpaceNet :18-148, LightingMLP :152-188, DualSpaceNeRF :191-275),
so `render.net.load_state_dict(ckpt["model"])`, `render.net.set_light_center(...)`,
`render.net.nerf.w = 0` (test.py:193-196, vis_lighting.py:57-58) work unchanged.  The modules only
HOLD parameters; every forward is executed by the HIP library - there is no torch implementation of the math
here and no fallback.

Which calls carry gradients:
  * `Renderer.render` (train mode): one autograd node for the whole path (dsn_render_rays_grad);
  * `DualSpaceNeRF.forward` (train mode, grad enabled): one autograd node for (colour, density) w.r.t. the 33
    parameters (dsn_module_grad) - inputs receive no gradient, they are data in the reference's callers too;
  * `SpaceNet.forward` and `LightingMLP.forward` called on their own are evaluation-only: in train mode with grad
    enabled they raise instead of silently returning tensors without a graph.
`batch_info` may carry "_dsn_scene" (set by Renderer) so the per-frame state is not rebuilt.
"""
from __future__ import annotations

import weakref

import torch
from torch import nn

from .. import _lib

_MAX_ROW_GROUPS = 64


def _linear(i, o):
    return nn.Linear(i, o)


def _wants_graph(module):
    return module.training and torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


def _standalone_packed(module, prefix, cache_attr="_solo_packed"):
    """Packed image for a sub-module used without its DualSpaceNeRF owner: the other parameters are zero (the kernels
    that run for this sub-module do not read them)."""
    dev = next(module.parameters()).device
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    from ..synth import STATE_SHAPES
    own = {prefix + k: v for k, v in module.named_parameters()}
    cache = getattr(module, cache_attr, None)
    if cache is None or cache[0].device != dev:
        zeros = {k: torch.zeros(shape, device=dev) for k, shape in STATE_SHAPES if k not in own}
        cache = (_lib.PackedParams(dev), zeros)
        object.__setattr__(module, cache_attr, cache)
    return cache[0].update({**cache[1], **own})


class SpaceNet(nn.Module):
    """Canonical-space density + "essence" colour field (reference :18-148)."""

    def __init__(self, maxFrame=500, code_dim=8, essence_dim=3, cfg=None):
        super().__init__()
        assert code_dim == 8 and essence_dim == 3 and maxFrame == 500, \
            "the gfx950 kernels are specialised to the reference's shipped architecture"
        self.use_dir = False
        self.cfg = cfg
        self.code_dim = code_dim
        self.pos_dim = 63           # Trigonometric_kernel(L=10, include_input=True) on 3 inputs
        self.dir_dim = 0
        backbone_dim, head_dim = 256, 128
        self.embedding = nn.Embedding(maxFrame, code_dim)
        in_dim = self.pos_dim + code_dim + 16
        self.stage1 = nn.Sequential(_linear(in_dim, backbone_dim), nn.ReLU(inplace=True),
                                    _linear(backbone_dim, backbone_dim), nn.ReLU(inplace=True),
                                    _linear(backbone_dim, backbone_dim), nn.ReLU(inplace=True),
                                    _linear(backbone_dim, backbone_dim), nn.ReLU(inplace=True))
        self.stage2 = nn.Sequential(_linear(backbone_dim + self.pos_dim, backbone_dim), nn.ReLU(inplace=True),
                                    _linear(backbone_dim, backbone_dim), nn.ReLU(inplace=True),
                                    _linear(backbone_dim, backbone_dim), nn.ReLU(inplace=True))
        self.density_net = nn.Sequential(_linear(backbone_dim, 1))
        self.rgb_net = nn.Sequential(nn.ReLU(inplace=True), _linear(backbone_dim, head_dim), nn.ReLU(inplace=True),
                                     _linear(head_dim, essence_dim))
        self.w = None
        self._owner = None   # weakref to the DualSpaceNeRF that holds this module (its packed image covers all 33 tensors)

    def _packed(self):
        owner = self._owner() if self._owner is not None else None
        return owner.packed() if owner is not None else _standalone_packed(self, "nerf.")

    def forward(self, pos, rays, idx, density_only=False, pose_feats=None):
        """reference :93-148.  pos [N,3] or [R,S,3] canonical points, idx = frame index per point (any shape with N
        elements, or one element), pose_feats [N,16] (or [1,16]) -> (rgbs [N,3], density [N,1], 0), or density [N,1] when
        density_only.  `rays` is ignored (use_dir is hard-wired off, reference :21).  The frame code and the pose features
        are constant per call in every caller of the reference; distinct rows are supported by evaluating each distinct
        (idx, pose_feats) row group with its own folded first-layer bias (at most 64 groups per call)."""
        if _wants_graph(self):
            raise RuntimeError("SpaceNet.forward on its own builds no autograd graph in dsnerf_amd: differentiate through "
                               "DualSpaceNeRF.forward or Renderer.render, or call it under torch.no_grad() / in eval mode")
        if pose_feats is None:
            raise RuntimeError("SpaceNet.forward needs pose_feats [N,16] (the reference concatenates it unconditionally, "
                               "model/spacenet.py:131)")
        packed = self._packed()
        dev = packed.device
        x = pos.reshape(-1, 3).to(device=dev, dtype=torch.float32).contiguous()
        N = x.shape[0]
        idx = torch.as_tensor(idx).reshape(-1).to(dev)
        pf = pose_feats.reshape(-1, 16).to(device=dev, dtype=torch.float32)
        if idx.numel() == 1:
            idx = idx.expand(N)
        if pf.shape[0] == 1:
            pf = pf.expand(N, 16)
        if idx.numel() != N or pf.shape[0] != N:
            raise RuntimeError("SpaceNet.forward: idx / pose_feats must have one row per point")
        zero_code = self.w is not None
        key = torch.cat([idx.to(torch.float32)[:, None], pf], dim=1)
        sigma = torch.empty(N, dtype=torch.float32, device=dev)
        ess = None if density_only else torch.empty(N, 3, dtype=torch.float32, device=dev)
        state = _lib.PoseState(dev)
        if bool((key == key[:1]).all()):
            groups = [(key[0], None)]
        else:
            rows, inv = torch.unique(key, dim=0, return_inverse=True)
            if rows.shape[0] > _MAX_ROW_GROUPS:
                raise RuntimeError(f"SpaceNet.forward: {rows.shape[0]} distinct (frame index, pose_feats) rows in one call; "
                                   f"at most {_MAX_ROW_GROUPS} are supported")
            groups = [(rows[g], (inv == g).nonzero().reshape(-1)) for g in range(rows.shape[0])]
        for row, sel in groups:
            state.set_pose(packed, None, int(row[0]), zero_code=zero_code, pose_feat=row[1:].contiguous())
            xs = x if sel is None else x[sel].contiguous()
            # the exact-fp32 kernel serves sigma-only / sigma + essence queries (dsn_field without grad)
            sg, es, _ = _lib.field(state, packed, xs, want_essence=not density_only, want_grad=False)
            if sel is None:
                sigma, ess = sg, es
            else:
                sigma[sel] = sg
                if ess is not None:
                    ess[sel] = es
        if density_only:
            return sigma[:, None]
        return ess, sigma[:, None], 0


class LightingMLP(nn.Module):
    """Observation-space lighting network (reference :152-188)."""

    def __init__(self, essence_dim):
        super().__init__()
        self.in_channels = 9
        W = 128
        self.lights_encoding = nn.Sequential(_linear(self.in_channels, W), nn.ReLU(True), _linear(W, W), nn.ReLU(True),
                                             _linear(W, 1), nn.ELU(alpha=1.0, inplace=True))
        self._owner = None

    def forward(self, normal, xyz_world, view_dir_world, essence_feature):
        """reference :174-188: colour [N,3] = (ELU(MLP([normal, xyz_world, view / |view|])) + 1) * essence_feature
        (dsn_light; the light-centre / rotation edits belong to DualSpaceNeRF.forward and are not applied here)."""
        if _wants_graph(self):
            raise RuntimeError("LightingMLP.forward on its own builds no autograd graph in dsnerf_amd: differentiate through "
                               "DualSpaceNeRF.forward or Renderer.render, or call it under torch.no_grad() / in eval mode")
        owner = self._owner() if self._owner is not None else None
        packed = owner.packed() if owner is not None else _standalone_packed(self, "lighting_mlp.")
        return _lib.light(packed, normal, xyz_world, view_dir_world, essence_feature)


class _ModuleForward(torch.autograd.Function):
    """DualSpaceNeRF.forward as one differentiable node: forward = dsn_field + dsn_shade, backward = dsn_module_grad."""

    @staticmethod
    def forward(ctx, net, scene, call, *params):
        x_w, x_c, view, poses, frame, zero_code = call
        packed = net.packed(scene.device)
        sigma, ess, g = _lib.field(scene, packed, x_c)
        _, _, colour = _lib.shade(scene, packed, x_c, g, x_w, view, ess, 1)
        ctx.save_for_backward(*params)
        ctx.net, ctx.scene, ctx.call, ctx.frame_key = net, scene, call, scene.frame_key
        ctx.set_materialize_grads(False)
        return colour, sigma[:, None]

    @staticmethod
    def backward(ctx, g_colour, g_sigma):
        saved = ctx.saved_tensors
        x_w, x_c, view, poses, frame, zero_code = ctx.call
        scene = ctx.scene
        if scene.frame_key != ctx.frame_key:
            raise RuntimeError("DualSpaceNeRF.forward: another frame was set in the scene between this forward and its backward")
        N = x_c.shape[0]
        dev = scene.device
        g_colour = torch.zeros(N, 3, device=dev) if g_colour is None else g_colour
        g_sigma = torch.zeros(N, 1, device=dev) if g_sigma is None else g_sigma
        if not hasattr(ctx.net, "_grad_ws") or ctx.net._grad_ws is None:
            ctx.net._grad_ws = _lib.GradWorkspace(dev)
        grads = _lib.module_grad(scene, [p.detach() for p in saved], poses, frame, zero_code, x_w, x_c, view, g_colour,
                                 g_sigma, ws=ctx.net._grad_ws, packed=ctx.net.packed(dev))
        grads = [g.to(device=p.device, dtype=p.dtype).reshape(p.shape) for g, p in zip(grads, saved)]
        return (None, None, None) + tuple(grads)


class DualSpaceNeRF(nn.Module):
    """reference :191-275.  forward(pos[N,6], rays[N,6], frame_idx, batch_info, density_only)."""

    def __init__(self, cfg=None):
        super().__init__()
        essence_dim = 3
        self.nerf = SpaceNet(essence_dim=essence_dim, cfg=cfg)
        self.lighting_mlp = LightingMLP(essence_dim=essence_dim)
        self.pose_mlp = nn.Sequential(_linear(23 * 4, 64), nn.ReLU(inplace=True), _linear(64, 64),
                                      nn.ReLU(inplace=True), _linear(64, 16))
        self.light_center = None
        self.rot_center = None
        self.rot = None
        self.nerf._owner = weakref.ref(self)
        self.lighting_mlp._owner = weakref.ref(self)
        self._packed = None
        self._scene_cache = None
        self._pose_state = None
        self._grad_ws = None

    # ---- reference setters (:268-275) ----
    def set_rot_center(self, center):
        self.rot_center = center.cuda() if torch.cuda.is_available() else center

    def set_rot(self, rot):
        self.rot = rot.cuda() if torch.cuda.is_available() else rot

    def set_light_center(self, center):
        self.light_center = center.cuda() if torch.cuda.is_available() else center

    # ---- HIP plumbing ----
    def packed(self, device=None, force=False):
        """MFMA-ordered device image of the 33 parameters, re-packed when one changed (tensor version counters).
        force=True after edits the counters cannot see (`param.data[...] = ...`)."""
        device = device or next(self.parameters()).device
        if device.type != "cuda":
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
        if self._packed is None or self._packed.device != device:
            self._packed = _lib.PackedParams(device)
        sd = dict(self.named_parameters())
        self._packed.update(sd, force=force)
        return self._packed

    def frame_args(self, batch_info):
        """(zero_code, light_shift, rot, rot_center) per reference :126-129, :254-263."""
        zero_code = self.nerf.w is not None
        light_shift = None
        if self.light_center is not None:
            th = batch_info["Th"][0].to(self.light_center).reshape(-1, 3).mean(dim=0)
            light_shift = self.light_center.reshape(-1)[:3] - th
        rot = rc = None
        if self.rot_center is not None and self.rot is not None:
            rot, rc = self.rot, self.rot_center.reshape(-1)[:2]
        return zero_code, light_shift, rot, rc

    def scene_for(self, batch_info, frame_idx):
        """Scene with this frame's state; reuses Renderer's scene when present (it has set the frame)."""
        scene = batch_info.get("_dsn_scene")
        if scene is not None:
            return scene
        dev = torch.device("cuda", torch.cuda.current_device())
        cm = batch_info["canonical_model"]
        key = (cm["vertex"].data_ptr(), batch_info["face_idx"].data_ptr())
        if self._scene_cache is None or self._scene_cache[0] != key:
            self._scene_cache = (key, _lib.Scene(cm["vertex"], batch_info["face_idx"], dev))
        scene = self._scene_cache[1]
        fi = int(torch.as_tensor(frame_idx).reshape(-1)[0])
        zero_code, ls, rot, rc = self.frame_args(batch_info)
        scene.set_frame(self.packed(dev), batch_info["xyz"][0], batch_info["poses"][0], fi, zero_code, ls, rot, rc)
        return scene

    def forward(self, pos, rays, frame_idx=0, batch_info={}, density_only=False):
        fi = int(torch.as_tensor(frame_idx).reshape(-1)[0])
        if density_only:
            # reference :223-241: needs only batch_info['poses']; pos is [..., 3] canonical points here (query_volume passes
            # pts directly) or [..., 6]
            dev = torch.device("cuda", torch.cuda.current_device())
            packed = self.packed(dev)
            if self._pose_state is None or self._pose_state.device != dev:
                self._pose_state = _lib.PoseState(dev)
            zero_code, ls, rot, rc = self.frame_args(batch_info) if "Th" in batch_info else (self.nerf.w is not None, None, None, None)
            self._pose_state.set_pose(packed, batch_info["poses"][0], fi, zero_code, ls, rot, rc)
            xyz_cano = pos[..., 3:] if pos.shape[-1] == 6 else pos
            x_c = xyz_cano.reshape(-1, 3).to(device=dev, dtype=torch.float32).contiguous()
            sigma, _, _ = _lib.field(self._pose_state, packed, x_c, want_essence=False, want_grad=False)
            return sigma[:, None]
        scene = self.scene_for(batch_info, fi)
        dev = scene.device
        pos = pos.to(device=dev, dtype=torch.float32)
        rays = rays.to(device=dev, dtype=torch.float32)
        x_w = pos[:, :3].contiguous()
        x_c = pos[:, 3:].contiguous()
        view = rays[:, :3].contiguous()
        if _wants_graph(self):
            sd = dict(self.named_parameters())
            poses = batch_info["poses"][0].to(device=dev, dtype=torch.float32).contiguous()
            colour, density = _ModuleForward.apply(self, scene, (x_w, x_c, view, poses, fi, self.nerf.w is not None),
                                                   *[sd[k] for k in _lib.PARAM_ORDER])
            return colour, density, None
        packed = self.packed(dev)
        sigma, ess, g = _lib.field(scene, packed, x_c)
        # dsn_shade takes per-ray directions [N/S,3]; with S=1 every point carries its own direction
        _, _, colour = _lib.shade(scene, packed, x_c, g, x_w, view, ess, 1)
        return colour, sigma[:, None], None
