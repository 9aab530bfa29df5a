"""Soak of the training backward (two streams): the same 8192 x 64 batch N times without a synchronisation in between - the trunk's
gradients (fixed-order reductions: bit-reproducible by construction) must equal the first run's bit for bit, the atomically accumulated
tensors within 1e-5.  Catches rare races between the backward's two chains / between steps.   python scripts/soak_train_determinism.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import dsnerf_amd
from dsnerf_amd import synth
from cases import make_cfg
from helpers import state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
R, S, HW = 8192, 64, 512
canon, faces = synth.make_body(); xyz = synth.pose_body(canon)
rays = synth.make_rays(HW, HW, xyz, fit_box=True)
sel = np.linspace(0, HW * HW - 1, R).astype(np.int64)
for name in ("x", "x_w4"):
    sd = state(name) if name != "x" else state()
    cfg = make_cfg(S)
    net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.train()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    b = {"ray_o": T(rays["ray_o"][sel])[None], "ray_d": T(rays["ray_d"][sel])[None], "near": T(rays["near"][sel])[None],
         "far": T(rays["far"][sel])[None], "xyz": T(xyz)[None], "poses": T(synth.make_poses())[None], "Th": torch.zeros(1, 1, 3, device=dev),
         "frame": torch.tensor([5])}
    target = T(synth.hash_uniform(R * 3, 77).reshape(R, 3).astype(np.float32))
    runs = []
    for i in range(n + 1):
        torch.manual_seed(11)
        net.zero_grad()
        out = r.render(b)["coarse"]
        (torch.nn.functional.mse_loss(out["color"], target) + 0.1 * out["acc_map"].mean()).backward()
        runs.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})       # (no synchronisation: clones are stream-ordered)
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for g in runs[1:]:
        for k in g:
            if "stage" in k:
                bad += int(not torch.equal(g[k], runs[0][k]))
            else:
                worst = max(worst, float((g[k] - runs[0][k]).norm() / runs[0][k].norm().clamp_min(1e-30)))
    print(f"{name}: {n} steps, two streams {'on' if r._grad_ws._aux is not None else 'off'}: {bad} trunk tensors differ from the first run; atomically summed tensors within {worst:.1e}")
    assert bad == 0 and worst < 1e-5
