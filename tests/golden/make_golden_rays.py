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

# ---- Human3.6M convention (BASELINE configs[3]): utils/h36m_utils.py get_rays (:14-28, unit directions) + get_near_far
# (:61-76, float32 slab test) as composed by get_rays_within_bounds (:162-176) / the test split of sample_ray_h36m (:147-157).
# Two cameras: the small one above and a 1024 x 1024 one (the frame size of configs[3]; every 37th pixel is stored).
from utils import h36m_utils  # noqa: E402

bounds32 = h36m_utils.get_bounds(np.array([[-0.65, -0.85, -0.30], [0.55, 0.75, 0.25]]))      # float32, +-0.05 (:372-379)
arrs = {"K": K, "R": R, "T": T, "bounds": bounds32, "H": H, "W": W}
ro, rd, near, far, mask = h36m_utils.get_rays_within_bounds(H, W, K, R, T, bounds32)
full_o, full_d = h36m_utils.get_rays(H, W, K, R, T)
arrs.update(ray_o=full_o.reshape(-1, 3).astype(np.float32), ray_d=full_d.reshape(-1, 3).astype(np.float32), near=near, far=far,
            mask_at_box=mask.reshape(-1))
K2 = np.array([[1145.0, 0.0, 512.5], [0.0, 1143.8, 515.4], [0.0, 0.0, 1.0]])
T2 = np.array([[0.05], [0.1], [4.2]])
H2 = W2 = 1024
ro2, rd2, near2, far2, mask2 = h36m_utils.get_rays_within_bounds(H2, W2, K2, R, T2, bounds32)
fo2, fd2 = h36m_utils.get_rays(H2, W2, K2, R, T2)
pick = np.arange(0, H2 * W2, 37)
m2 = mask2.reshape(-1)
nf = np.zeros((H2 * W2, 2), np.float32)
nf[m2, 0], nf[m2, 1] = near2, far2
arrs.update(K2=K2, T2=T2, H2=H2, W2=W2, pick2=pick, ray_o2=fo2.reshape(-1, 3).astype(np.float32)[pick],
            ray_d2=fd2.reshape(-1, 3).astype(np.float32)[pick], near2=nf[pick, 0], far2=nf[pick, 1], mask2=m2[pick],
            mask2_count=np.int64(m2.sum()))
np.savez_compressed(os.path.join(HERE, "camera_rays_h36m.npz"), **arrs)
print("h36m rays", arrs["ray_o"].shape, "in box", int(mask.sum()), "| 1024^2 in box", int(m2.sum()))
