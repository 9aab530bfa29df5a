"""What fraction of the trunk's units is alive (relu pattern = 1) on the rows the training step evaluates?  Decides what packing the
operand arrays h_l / a_l / hdot_l / ahat_l behind the layer's relu mask could save.  Reads h_l from the training
workspace after one step of bench.py's batch (layout: dsn_train.hip carve()); both weight sets."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../..",):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
from types import SimpleNamespace
import dsnerf_amd
from dsnerf_amd import _lib, synth
from benchlib.common import load_weights
dev = torch.device("cuda:0")
R, S, HW = 8192, 64, 512
n = R * S
for weights in ("default", "w4"):
    canon, faces = synth.make_body(); sd = load_weights(synth, weights); xyz = synth.pose_body(canon, seed=3)
    rays = synth.make_rays(HW, HW, xyz, fit_box=True)
    sel = np.linspace(0, HW * HW - 1, R).astype(np.int64)
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.train()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    batch = {"ray_o": T(rays["ray_o"][sel])[None], "ray_d": T(rays["ray_d"][sel])[None], "near": T(rays["near"][sel])[None],
             "far": T(rays["far"][sel])[None], "xyz": T(xyz)[None], "poses": T(synth.make_poses(seed=5))[None],
             "Th": torch.zeros(1, 1, 3, device=dev), "frame": torch.tensor([5])}
    target = T(synth.hash_uniform(R * 3, 77).reshape(R, 3).astype(np.float32))
    torch.manual_seed(233)
    out = r.render(batch)["coarse"]
    torch.nn.functional.mse_loss(out["color"], target).backward()
    torch.cuda.synchronize()
    buf = r._grad_ws.buf
    f_rows, b_rows = _lib.grad_row_counts(r._grad_ws, R, S)
    off = n + 4 * n + 12 * n + 4 * 64 * n                      # transparent, idx_c, x_c, pe
    h_off = [off + l * 1024 * n for l in range(7)]
    tail = off + 28 * 1024 * n + 224 * n + 512 * n + 12 * n + 4 * n + 12 * n + 1024 * n + 4 * 64 * n + 12 * n + 36 * n + 512 * n + 512 * n \
        + 4 * n + 4 * n + 12 * n + 4 * n + 12 * n + 12 * n + 4 * n + 512 * n + 512 * n + 36 * n + 512 * n + 12 * n + 4 * n + n
    list1 = buf[tail:tail + 4 * n].view(torch.int32)[:f_rows].long()
    list2 = buf[tail + 4 * n:tail + 8 * n].view(torch.int32)[:b_rows].long()
    assert int(list1.max()) < n and bool((list1[1:] > list1[:-1]).all()), "layout drifted"
    print(weights, "forward rows", f_rows, "backward rows", b_rows)
    for l in range(7):
        h = buf[h_off[l]:h_off[l] + 1024 * n].view(torch.float32).view(n, 256)
        a1 = (h[list1] > 0).float(); a2 = (h[list2] > 0).float()
        per_row = a2.sum(1)
        print("  layer %d  live units: forward rows %.3f  backward rows %.3f   per row p50 %d p90 %d p99 %d max %d" %
              (l, float(a1.mean()), float(a2.mean()), int(per_row.quantile(0.5)), int(per_row.quantile(0.9)), int(per_row.quantile(0.99)), int(per_row.max())))
