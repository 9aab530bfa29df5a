from .spacenet import DualSpaceNeRF, LightingMLP, SpaceNet  # noqa: F401
