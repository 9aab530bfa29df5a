"""Host-side mirror of the reference's network classes for the hot path.

Same class names, constructor arguments, method names, return conventions and state_dict layout as
/root/reference/model/spacenet.py (SpaceNet :18-148, LightingMLP :152-188, DualSpaceNeRF :191-275),
so `render.net.load_state_dict(ckpt["model"])`, `render.net.set_light_center(...)`,
`render.net.nerf.w = 0` (test.py:193-196, vis_lighting.py:57-58) work unchanged.  The modules only
HOLD parameters; every forward is executed by the HIP library (csrc/dsn_field.hip) - there is no
torch implementation of the math here and no fallback.

Differences, all at the boundary and documented in DESIGN.md:
  * forward passes are inference-only in this round (training backward is SURVEY.md 8f-1);
  * `batch_info` may carry "_dsn_scene" (set by Renderer) so the per-frame state is not rebuilt.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib


def _linear(i, o):
    return nn.Linear(i, o)


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
        self._owner = None   # set by DualSpaceNeRF: the kernels need the sibling modules' parameters too

    def forward(self, pos, rays, idx, density_only=False, pose_feats=None):
        """pos [N,3] or [R,S,3] canonical points -> (rgbs [N,3], density [N,1], 0) or density [N,1].

        `pose_feats` is accepted for signature parity; the pose code is derived per frame inside
        dsn_set_frame from the owner's pose_mlp, exactly as DualSpaceNeRF.forward does (:223-236).
        """
        if self._owner is None:
            raise RuntimeError("SpaceNet must be used through DualSpaceNeRF (needs pose_mlp / scene state)")
        return self._owner()._nerf_forward(pos, idx, density_only)


class LightingMLP(nn.Module):
    """Observation-space lighting network (reference :152-188); parameters only."""

    def __init__(self, essence_dim):
        super().__init__()
        self.in_channels = 9
        W = 128
        self.lights_encoding = nn.Sequential(_linear(self.in_channels, W), nn.ReLU(True), _linear(W, W), nn.ReLU(True),
                                             _linear(W, 1), nn.ELU(alpha=1.0, inplace=True))

    def forward(self, normal, xyz_world, view_dir_world, essence_feature):
        raise RuntimeError("LightingMLP is evaluated inside DualSpaceNeRF.forward by dsn_shade (fused with the "
                           "normal transform); call DualSpaceNeRF.forward")


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
        import weakref
        self.nerf._owner = weakref.ref(self)
        self._packed = None
        self._scene_cache = None
        self._cur = None   # (scene, S) of the forward in flight, for SpaceNet.forward

    # ---- reference setters (:268-275) ----
    def set_rot_center(self, center):
        self.rot_center = center.cuda() if torch.cuda.is_available() else center

    def set_rot(self, rot):
        self.rot = rot.cuda() if torch.cuda.is_available() else rot

    def set_light_center(self, center):
        self.light_center = center.cuda() if torch.cuda.is_available() else center

    # ---- HIP plumbing ----
    def packed(self, device=None):
        device = device or next(self.parameters()).device
        if device.type != "cuda":
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
        if self._packed is None or self._packed.device != device:
            self._packed = _lib.PackedParams(device)
        sd = dict(self.named_parameters())
        self._packed.update(sd)
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
        """Scene with this frame's state; reuses Renderer's scene when present."""
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

    def _nerf_forward(self, pos, idx, density_only):
        if self._cur is None:
            raise RuntimeError("SpaceNet.forward needs a frame: call it through DualSpaceNeRF.forward / Renderer")
        scene = self._cur
        x_c = pos.reshape(-1, 3).to(device=scene.device, dtype=torch.float32).contiguous()
        sigma, ess, _ = _lib.field(scene, self.packed(scene.device), x_c, want_essence=not density_only,
                                   want_grad=False)
        if density_only:
            return sigma[:, None]
        return ess, sigma[:, None], 0

    def forward(self, pos, rays, frame_idx=0, batch_info={}, density_only=False):
        scene = self.scene_for(batch_info, frame_idx)
        self._cur = scene
        dev = scene.device
        packed = self.packed(dev)
        if density_only:
            # reference :238-241: pos is [..., 3] canonical points here (query_volume passes pts directly)
            xyz_cano = pos[..., 3:] if pos.shape[-1] == 6 else pos
            x_c = xyz_cano.reshape(-1, 3).to(device=dev, dtype=torch.float32).contiguous()
            sigma, _, _ = _lib.field(scene, packed, x_c, want_essence=False, want_grad=False)
            return sigma[:, None]
        pos = pos.to(device=dev, dtype=torch.float32)
        rays = rays.to(device=dev, dtype=torch.float32)
        x_w = pos[:, :3].contiguous()
        x_c = pos[:, 3:].contiguous()
        view = rays[:, :3].contiguous()
        sigma, ess, g = _lib.field(scene, packed, x_c)
        # dsn_shade takes per-ray directions [N/S,3]; with S=1 every point carries its own direction
        _, _, colour = _lib.shade(scene, packed, x_c, g, x_w, view, ess, 1)
        return colour, sigma[:, None], None
