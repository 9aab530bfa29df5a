"""Where does the host spend a training step?  No syncs inside the loop; per-step host enqueue time, per-step GPU time
(events) and a cProfile of the un-synchronised loop."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import dsnerf_amd
from dsnerf_amd import synth
dev = torch.device("cuda:0")
S, R = 64, 8192
canon, faces = synth.make_body(); sd = synth.make_state_dict(); xyz = synth.pose_body(canon)
rays = synth.make_rays(512, 512, xyz, fit_box=True)
sel = np.linspace(0, 512*512-1, R).astype(np.int64)
cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"), MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf", FINE_RAY_SAMPLING=-1))
net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.train()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
batch = {"ray_o": T(rays["ray_o"][sel])[None], "ray_d": T(rays["ray_d"][sel])[None], "near": T(rays["near"][sel])[None], "far": T(rays["far"][sel])[None], "xyz": T(xyz)[None], "poses": T(synth.make_poses())[None], "Th": torch.zeros(1,1,3,device=dev), "frame": torch.tensor([5])}
target = T(synth.hash_uniform(R*3, 77).reshape(R,3).astype(np.float32))
opt = torch.optim.Adam(net.parameters(), lr=5e-4)
marks = []
def step():
    t0 = time.perf_counter()
    opt.zero_grad()
    t1 = time.perf_counter()
    out = r.render(batch)["coarse"]
    loss = torch.nn.functional.mse_loss(out["color"], target)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(5): step()
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
m = np.array(marks) * 1e3
print("host per step ms: zero %.2f fwd %.2f bwd %.2f opt %.2f | enqueue total %.2f ms/step, wall %.2f ms/step" % (*m.mean(0), 1e3*(t1-t0)/20, 1e3*(t2-t0)/20))
print("fwd host per step:", np.round(m[:, 1], 1)); print("bwd host per step:", np.round(m[:, 2], 1))
t0 = time.perf_counter(); 
buf = torch.empty(2*R*S)
for _ in range(20): a = torch.rand(1, R, S, out=buf[:R*S].view(1,R,S)); b = torch.randn(R, S, out=buf[R*S:].view(R,S))
print("draws alone: %.2f ms" % (1e3 * (time.perf_counter() - t0) / 20), "threads", torch.get_num_threads())
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
