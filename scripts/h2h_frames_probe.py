import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, dsnerf_amd
from dsnerf_amd import synth
from types import SimpleNamespace
canon, faces = synth.make_body(); sd = synth.make_state_dict(); poses = synth.make_poses(seed=5); xyz = synth.pose_body(canon, seed=3)
rays = synth.make_rays(512, 512, xyz, fit_box=True)
dev = torch.device("cuda:0"); H=W=512; S=64
for rep in range(2):
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"), MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.eval()
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    batch = {"ray_o": C(rays["ray_o"])[None], "ray_d": C(rays["ray_d"])[None], "near": C(rays["near"])[None], "far": C(rays["far"])[None],
             "xyz": C(xyz)[None], "poses": C(poses)[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
             "img": torch.zeros(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}
    ms=[]
    for i in range(10):
        b = dict(batch); b["near"], b["far"] = torch.from_numpy(batch["near"].numpy().copy()), torch.from_numpy(batch["far"].numpy().copy())
        torch.cuda.synchronize(); t=time.perf_counter(); out = r.render_view(b); ms.append(round(1e3*(time.perf_counter()-t),1))
    print("per-frame ms:", ms, flush=True)
