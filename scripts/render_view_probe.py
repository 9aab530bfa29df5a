"""End-to-end Renderer.render_view on host batches (the reference's call: CPU tensors in, CPU images out) vs device-resident."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import dsnerf_amd
from dsnerf_amd import synth
dev = torch.device("cuda:0")
H = W = 512; S = 64
canon, faces = synth.make_body(); sd = synth.make_state_dict(); xyz = synth.pose_body(canon)
rays = synth.make_rays(H, W, xyz, fit_box=True)
cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"), MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf", FINE_RAY_SAMPLING=-1))
net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.eval()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
def batch(to=None):
    b = {"ray_o": T(rays["ray_o"])[None], "ray_d": T(rays["ray_d"])[None], "near": T(rays["near"].copy())[None], "far": T(rays["far"].copy())[None],
         "xyz": T(xyz)[None], "poses": T(synth.make_poses())[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
         "img": torch.zeros(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}
    return b if to is None else {k: (v.to(to) if k != "frame" else v) for k, v in b.items()}
def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
with torch.no_grad():
    hb = batch(); db = batch(dev)
    print("render_view, host batch -> host images : %.2f ms per 512x512 frame" % timed(lambda: r.render_view({**hb, 'near': hb['near'].clone(), 'far': hb['far'].clone()})))
    print("render_view, device batch -> host images: %.2f ms" % timed(lambda: r.render_view({**db, 'near': db['near'].clone(), 'far': db['far'].clone()})))
    print("render_view, device batch -> device images: %.2f ms" % timed(lambda: r.render_view({**db, 'near': db['near'].clone(), 'far': db['far'].clone()}, device_output=True)))
