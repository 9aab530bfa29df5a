"""dsnerf_amd - MI355X (gfx950) implementation of the Dual-Space-NeRF volume-rendering hot path.

Python surface mirrors the reference (zyhbili/Dual-Space-NeRF): can_render.Renderer and
model.spacenet.{DualSpaceNeRF, SpaceNet, LightingMLP}; the math runs in libdsnerf_hip.so
(include/dsnerf.h).  Import name: `dsnerf_amd` (the directory is `dual-space-nerf_amd/`, loaded by
the repo-root shim dsnerf_amd.py because a hyphenated directory is not an importable name).
"""
from . import synth  # noqa: F401
from . import _lib  # noqa: F401
from .can_render import Renderer  # noqa: F401
from .model.spacenet import DualSpaceNeRF, LightingMLP, SpaceNet  # noqa: F401
from .parallel import RayParallel  # noqa: F401

__all__ = ["Renderer", "DualSpaceNeRF", "SpaceNet", "LightingMLP", "RayParallel", "synth"]
