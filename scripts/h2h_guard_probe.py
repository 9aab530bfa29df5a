"""Host path under the reference's caller pattern (test.py:61-76): render_view between torch CPU ops on the main thread.

    python scripts/h2h_guard_probe.py            (on the GPU box; prints one JSON object)

For every caller pattern x Renderer.host_pool_limit setting: mean ms of render_view alone and of the whole loop iteration
(caller ops included), plus what the cgroup throttled in between (cpu.stat) - the mechanism behind the 19 -> 36 ms of BENCH_r02.
"""
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import dsnerf_amd
from dsnerf_amd import synth


def cpu_stat():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            for line in open(p):
                k, v = line.split()
                out[k] = int(v)
            break
    return out


def read(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None


def main():
    info = {"nproc_affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "cpu.max": read("/sys/fs/cgroup/cpu.max"), "cfs_quota_us": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
            "OMP_WAIT_POLICY": os.environ.get("OMP_WAIT_POLICY"), "GOMP_SPINCOUNT": os.environ.get("GOMP_SPINCOUNT")}
    canon, faces = synth.make_body(); sd = synth.make_state_dict(); poses = synth.make_poses(seed=5); xyz = synth.pose_body(canon, seed=3)
    H = W = 512; S = 64
    rays = synth.make_rays(H, W, xyz, fit_box=True)
    dev = torch.device("cuda:0")
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0, TYPE="nerf",
                                                FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev); r.eval()
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    batch = {"ray_o": C(rays["ray_o"])[None], "ray_d": C(rays["ray_d"])[None], "near": C(rays["near"])[None], "far": C(rays["far"])[None],
             "xyz": C(xyz)[None], "poses": C(poses)[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
             "img": torch.rand(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}

    def caller_none(out):
        return None

    def caller_clone(out):                     # bench.py's key of round 2: a 1 MB .clone() right before the frame
        return batch["near"].clone()

    def caller_test_py(out):                   # test.py:61-76 on the previous frame's host images
        if out is None:
            return None
        c = torch.clamp(out["coarse_color"], min=0.0, max=1.0)
        gt = batch["img"][0]
        m = batch["mask_at_box"][0].bool().reshape(H, W)
        v = (c - gt) ** 2
        a = -10 * torch.log10(torch.mean(v[m]))
        b = -10 * torch.log10(torch.mean(v))
        _ = c.cpu().numpy(), gt.cpu().numpy()
        pred = (2 * c - 1).permute(2, 0, 1)[None].float().flip(1)
        return float(a) + float(b) + float(pred[0, 0, 0, 0])

    res = {"info": info, "runs": []}
    for limit in (None, 8, None, 8):
        r.host_pool_limit = limit
        for name, caller in (("numpy_only", caller_none), ("clone_1MB", caller_clone), ("test_py_metrics", caller_test_py)):
            view, loop = [], []
            out = None
            s0 = cpu_stat()
            for i in range(12):
                t0 = time.perf_counter()
                caller(out)
                b = dict(batch)
                b["near"], b["far"] = torch.from_numpy(batch["near"].numpy().copy()), torch.from_numpy(batch["far"].numpy().copy())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out = r.render_view(b)
                t2 = time.perf_counter()
                if i >= 4:
                    view.append(1e3 * (t2 - t1)); loop.append(1e3 * (t2 - t0))
            s1 = cpu_stat()
            rec = {"host_pool_limit": limit, "caller": name, "render_view_ms": float(np.mean(view)), "loop_ms": float(np.mean(loop)),
                   "render_view_ms_all": [round(v, 1) for v in view],
                   "throttled_ms": (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3,
                   "nr_throttled": s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), "threads_after": torch.get_num_threads()}
            res["runs"].append(rec)
            print(json.dumps(rec), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
