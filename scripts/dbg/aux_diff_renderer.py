"""one-stream vs two-stream backward through Renderer.render (cached forward), 8192 x 64"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
import dsnerf_amd
from dsnerf_amd import _lib, synth
from cases import make_cfg
dev = torch.device("cuda:0")
R, S, HW = 8192, 64, 512
canon, faces = synth.make_body(); sd = synth.make_state_dict(); xyz = synth.pose_body(canon)
rays = synth.make_rays(HW, HW, xyz, fit_box=True)
sel = np.linspace(0, HW * HW - 1, R).astype(np.int64)
cfg = make_cfg(S)
net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.train()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
b0 = {"ray_o": T(rays["ray_o"][sel])[None], "ray_d": T(rays["ray_d"][sel])[None], "xyz": T(xyz)[None], "poses": T(synth.make_poses())[None],
      "Th": torch.zeros(1, 1, 3, device=dev), "frame": torch.tensor([5])}
target = T(synth.hash_uniform(R * 3, 77).reshape(R, 3).astype(np.float32))
def run():
    torch.manual_seed(3)
    b = dict(b0); b["near"], b["far"] = T(rays["near"][sel])[None], T(rays["far"][sel])[None]
    net.zero_grad()
    out = r.render(b)["coarse"]
    loss = torch.nn.functional.mse_loss(out["color"], target)
    loss.backward(); torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().double().cpu().numpy() for k, p in net.named_parameters()}
os.environ["DSN_TRAIN_AUX"] = "0"; la, a = run(); _, a2 = run()
os.environ["DSN_TRAIN_AUX"] = "1"; lb, b = run(); _, b2 = run()
print("loss", la, lb)
for k in a:
    n = max(np.linalg.norm(a[k]), 1e-30)
    print("%-40s one twice %.1e   two vs one %.1e   two twice %.1e" % (k, np.linalg.norm(a[k] - a2[k]) / n, np.linalg.norm(a[k] - b[k]) / n, np.linalg.norm(b[k] - b2[k]) / n))
