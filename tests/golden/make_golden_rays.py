"""Golden vectors for the camera-ray / box near-far step, from the reference's own
utils/rays_utils.get_rays and get_near_far (cv2, which that module imports for unrelated helpers, is stubbed).
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rays.py"""
import os, sys, types
import numpy as np
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from utils import rays_utils  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
H, W = 20, 24
ang = 0.4
R = np.array([[np.cos(ang), 0, np.sin(ang)], [0.05, 1, 0.02], [-np.sin(ang), 0, np.cos(ang)]])
R, _ = np.linalg.qr(R)                       # a proper rotation-like matrix, float64 like the datasets' extrinsics
T = np.array([[0.1], [0.2], [3.1]])
K = np.array([[30.0, 0.3, 11.5], [0.0, 31.0, 9.5], [0.0, 0.0, 1.0]])
bounds = np.array([[-0.7, -0.9, -0.35], [0.6, 0.8, 0.3]])
ray_o, ray_d = rays_utils.get_rays(H, W, K, R, T)
ro32 = ray_o.reshape(-1, 3).astype(np.float32)
rd32 = ray_d.reshape(-1, 3).astype(np.float32)
near, far, mask = rays_utils.get_near_far(bounds, ro32, rd32)     # my_sample_ray(nrays<=0), rays_utils.py:176-184
np.savez_compressed(os.path.join(HERE, "camera_rays.npz"), K=K, R=R, T=T, bounds=bounds, H=H, W=W, ray_o=ro32, ray_d=rd32,
                    near=near.astype(np.float32), far=far.astype(np.float32), mask_at_box=mask)
print("rays", ro32.shape, "in box", int(mask.sum()))
