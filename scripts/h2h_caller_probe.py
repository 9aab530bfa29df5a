import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, dsnerf_amd
from dsnerf_amd import synth
from types import SimpleNamespace
canon, faces = synth.make_body(); sd = synth.make_state_dict(); poses = synth.make_poses(seed=5); xyz = synth.pose_body(canon, seed=3)
rays = synth.make_rays(512, 512, xyz, fit_box=True)
dev = torch.device("cuda:0")
H=W=512; S=64
cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"), MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf", FINE_RAY_SAMPLING=-1))
net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.eval()
C = lambda a: torch.from_numpy(np.ascontiguousarray(a))
batch = {"ray_o": C(rays["ray_o"])[None], "ray_d": C(rays["ray_d"])[None], "near": C(rays["near"])[None], "far": C(rays["far"])[None],
         "xyz": C(xyz)[None], "poses": C(poses)[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
         "img": torch.zeros(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}
def run(tag, mk):
    ms=[]
    for i in range(8):
        b = dict(batch); b["near"], b["far"] = mk(batch["near"]), mk(batch["far"])
        torch.cuda.synchronize(); t=time.perf_counter(); out = r.render_view(b)
        if i>1: ms.append(1e3*(time.perf_counter()-t))
    print(f"{tag:50s} {np.mean(ms):6.2f} ms  (threads {torch.get_num_threads()})", flush=True)
run("caller copies with torch .clone()", lambda t: t.clone())
run("caller copies with numpy", lambda t: torch.from_numpy(t.numpy().copy()))
run("caller copies with torch .clone() again", lambda t: t.clone())
run("caller copies with numpy again", lambda t: torch.from_numpy(t.numpy().copy()))
