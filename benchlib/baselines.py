"""The baselines bench.py reports beside the HIP path: the C oracle and the torch restatement on the host cores, the eager-PyTorch
restatement on this GPU, and the host-batch -> host-image time of Renderer.render_view.  The only place (with tests/ and smoke())
that touches oracle/ - as the thing timed BESIDE the product path, never inside it."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import ROOT


def host_to_host(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S, caller_torch_op=False, chunk=None):
    """The reference's render_view contract (can_render.py:248-278): a batch of HOST tensors in (what its DataLoader hands
    over), four HOST images out, one frame at a time through the Renderer mirror.  PCIe-inclusive: never `value`.
    The fresh per-frame near / far tensors (render_view updates them in place) are made with numpy, like the product of a DataLoader
    worker process; caller_torch_op=True runs the reference caller's own torch CPU ops on the main thread between the frames instead
    (test.py:61-76: clamp, psnr, permute / flip on the previous frame's 512 x 512 host images, + 1 MB .clone()s).  In round 2 that
    doubled the frame time (36 vs 19 ms): torch's intra-op pool, sized from the 128-256 hardware threads, burnt the cgroup's 16-core
    CPU quota in busy-waits and the kernel froze the process for the rest of the 100 ms period (profiles/r03a_h2h_guard.json); Renderer
    now fits the pool to the quota (_lib.fit_host_pool) and retires surplus threads while a frame is in flight (_HostPoolGuard)."""
    from types import SimpleNamespace
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0,
                                                TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev)
    r.eval()
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    batch = {"ray_o": C(rays["ray_o"])[None], "ray_d": C(rays["ray_d"])[None], "near": C(rays["near"])[None], "far": C(rays["far"])[None],
             "xyz": C(xyz)[None], "poses": C(poses)[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
             "img": torch.zeros(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}
    ms = []
    out = None
    gt = torch.rand(H, W, 3, dtype=torch.float64)
    for i in range(16):      # (the first frames of a new Renderer carry its one-off work: screen calibration, early-stop probe, staging buffers)
        b = dict(batch)
        if caller_torch_op:
            # what test.py:61-76 does on the main thread between two render_view calls, on the previous frame's host images:
            # torch.clamp, two psnr's (utils/metrics.py: mean of a squared difference, log10), the lpips-style permute / flip
            if out is not None:
                c = torch.clamp(out["coarse_color"], min=0.0, max=1.0)
                v = (c - gt) ** 2
                _ = float(-10 * torch.log10(torch.mean(v))) + float(-10 * torch.log10(torch.mean(v[batch["mask_at_box"][0].reshape(H, W)])))
                _ = (2 * c - 1).permute(2, 0, 1)[None].float().flip(1).sum()
            b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        else:
            b["near"], b["far"] = torch.from_numpy(batch["near"].numpy().copy()), torch.from_numpy(batch["far"].numpy().copy())
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = r.render_view(b, chunk=chunk)      # (chunk: rays per dsn_render_rays call; the workspace is sized for one chunk)
        assert not out["coarse_color"].is_cuda
        if i >= 4:
            ms.append(1e3 * (time.perf_counter() - t))
    return float(np.mean(ms))


def host_to_host_pipelined(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S, frames=12, in_flight=3):
    """The loop the reference's callers run (novel_pose_vis.py:41-66, test.py:55-64: `for batch in loader: render.render_view(batch)`)
    as ONE Renderer.render_views call over a loader of `frames` HOST batches with `in_flight` frames in flight: host batch -> four HOST
    images per frame, staging uploads, per-frame set-up, early-stop hand-over checks and downloads included.  Per frame, after one
    untimed pass over the same loader (a new Renderer's one-off work: probe frame, staging buffers, slots)."""
    from types import SimpleNamespace
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0,
                                                TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev)
    r.eval()
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    base = {"ray_o": C(rays["ray_o"])[None], "ray_d": C(rays["ray_d"])[None], "near": C(rays["near"])[None], "far": C(rays["far"])[None],
            "xyz": C(xyz)[None], "poses": C(poses)[None], "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([5]),
            "img": torch.zeros(1, H, W, 3, dtype=torch.float64), "mask_at_box": torch.ones(1, H * W, dtype=torch.bool)}

    def loader(n):
        for _ in range(n):      # (fresh near / far per batch, made with numpy like the product of a DataLoader worker process)
            b = dict(base)
            b["near"], b["far"] = torch.from_numpy(base["near"].numpy().copy()), torch.from_numpy(base["far"].numpy().copy())
            yield b

    for _ in range(4):
        r.render_view(next(loader(1)))
    r.render_views(loader(in_flight + 1), frames_in_flight=in_flight, device_output=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = r.render_views(loader(frames), frames_in_flight=in_flight, device_output=False)
    dt = time.perf_counter() - t
    assert len(out) == frames and not out[0]["coarse_color"].is_cuda
    return {"ms_per_frame": 1e3 * dt / frames, "frames": frames, "frames_in_flight": in_flight, "early_stop": bool(r.last_frame_info.get("early_stop")),
            "what": "Renderer.render_views(loader of host batches, device_output=False): host batch -> host images, per frame"}


def eager_baseline(args, _lib, synth, dev, chunks=5, train=True):
    """Stand-in for "the reference on one MI355X" (it cannot travel): the differentiable torch restatement the tests
    use as their oracle, run with eager PyTorch-ROCm on this GPU.  The parameter-independent geometry (sampling, both
    nearest-face searches, warp) is taken from the HIP kernels and NOT timed, which favours the baseline: in the
    reference those are pytorch3d knn_points calls over 13 776 centroids per sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import train_oracle as TO
    H = W = args.hw
    S = args.samples
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    poses = synth.make_poses(seed=5)
    xyz = synth.pose_body(canon, seed=3)
    rays = synth.make_rays(H, W, xyz, fit_box=True)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    scene.set_frame(packed, torch.from_numpy(xyz), torch.from_numpy(poses), 5, False, None, None, None)
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base = {"xyz": xyz, "canonical_vertex": canon, "faces": faces, "poses": poses, "frame": 5}

    def prepare(sel):
        o, d = T(rays["ray_o"][sel]), T(rays["ray_d"][sel])
        near, far = T(rays["near"][sel]), T(rays["far"][sel])
        pts, z = _lib.sample(scene, o, d, near, far, S, t_vals, None, want_pts=True)
        w = _lib.warp(scene, pts, d, S, want_dir=False)
        sig, ess, gr = _lib.field(scene, packed, w["x_c"])
        idx, _nw, _col = _lib.shade(scene, packed, w["x_c"], gr, pts, d, ess, S)
        g = dict(base, ray_o=rays["ray_o"][sel], ray_d=rays["ray_d"][sel])
        geom = {"x_c": w["x_c"].reshape(-1, 3), "transparent": w["transparent"].reshape(-1).bool(), "idx_canon": idx.long()}
        return g, z.cpu().numpy(), geom

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # eval: one 3072-ray chunk of the frame (can_render.py:172-245 processes 86 of them per 512x512 frame)
    chunk = 3072
    sel = np.arange(H * W // 2, H * W // 2 + chunk)
    g, z, geom = prepare(sel)
    params = {k: T(v) for k, v in sd.items()}
    t_eval = timed(lambda: TO.render(params, g, jitter_z=z, geom=geom), chunks)
    res = {"metric": "eager-PyTorch restatement on this GPU (network, autograd d sigma/dx, normals, lighting, compositing on one "
                     "3072-ray chunk as the reference processes a frame, can_render.py:172-245; both nearest-face searches excluded "
                     "- they come precomputed from the HIP kernels, which favours this baseline)",
           "eval_rays_per_s": chunk / t_eval, "eval_ms_per_3072_ray_chunk": 1e3 * t_eval,
           "eval_ms_per_512x512_frame": 1e3 * t_eval * (512 * 512 / chunk), "samples_per_ray": S, "kind": "port",
           "torch": torch.__version__}
    if not train:
        return res
    # train: forward + backward of an MSE loss on 8192 rays (trainer.py:70-81)
    R = args.train_rays
    sel = np.linspace(0, H * W - 1, R).astype(np.int64)
    g2, z2, geom2 = prepare(sel)
    pt = {k: T(v).requires_grad_(True) for k, v in sd.items()}
    target = T(synth.hash_uniform(R * 3, 77).reshape(R, 3).astype(np.float32))

    def train_step():
        for p in pt.values():
            p.grad = None
        out = TO.render(pt, g2, jitter_z=z2, geom=geom2)
        torch.nn.functional.mse_loss(out["color"], target).backward()

    t_train = timed(train_step, 3)
    res.update({"train_rays_per_s": R / t_train, "train_ms_per_step": 1e3 * t_train, "train_rays": R})
    return res


def cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, S, args):
    """The oracle (a C port of the reference algorithm) on the host cores, bounded sample of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    R = rays["ray_o"].shape[0]
    P = O.Params(sd)
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    from dsnerf_amd import _lib
    quota = _lib.cpu_quota_cores()
    # threads = the cores this process is really granted: the boxes show 256 hardware threads under a cgroup quota of 16 cores, and
    # 256 OpenMP threads on 16 cores' worth of bandwidth only add throttling and barrier waits
    cores = O.set_threads(max(1, min(os.cpu_count() or 1, int(quota))) if quota else 0)

    def run(n):
        sel = np.linspace(0, R - 1, n).astype(np.int64)
        t0 = time.perf_counter()
        O.render(rays["ray_o"][sel], rays["ray_d"][sel], rays["near"][sel], rays["far"][sel], S, xyz, canon, faces, P,
                 poses, sd["nerf.embedding.weight"][5], t_vals=tv)
        return time.perf_counter() - t0

    t_cal = run(max(cores, 64))                      # calibration (also warms the OpenMP pool)
    n = int(np.clip(args.cpu_rays if args.cpu_rays > 0 else 15.0 * max(cores, 64) / t_cal, 128, 65536))
    dt = run(n)
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "cpu_quota_cores": quota, "kind": "port",
            "sample": f"{n} rays evenly spread over the same frame x {S} samples (dense evaluation, OpenMP with {cores} threads"
                      + (f" under a cgroup CPU quota of {quota:g} cores" if quota else "") + f"), {dt:.1f} s"}


def cpu_baseline_torch(synth, canon, faces, xyz, poses, sd, rays, S, args):
    """SURVEY 8d baseline (ii): the torch restatement of the path (oracle/train_oracle.py: the reference's op sequence with
    torch CPU ops, autograd for d sigma/dx) on the host cores, on one 3072-ray chunk of the same frame as the reference
    processes it (can_render.py:172-245); geometry (both nearest-face searches) from the C oracle, timed with it.
    torch's intra-op pool does not scale to the boxes' 256 hardware threads on tensors of this size (with 256 threads the
    chunk takes 45 s, with 32 it takes 2 s): two pool sizes are timed and the better one is reported with its thread count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import train_oracle as TO
    import oracle as O
    R = rays["ray_o"].shape[0]
    n = 3072
    sel = np.arange(R // 2, R // 2 + n)
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    params = {k: torch.from_numpy(v) for k, v in sd.items()}
    g = {"ray_o": rays["ray_o"][sel], "ray_d": rays["ray_d"][sel], "xyz": xyz, "canonical_vertex": canon, "faces": faces, "poses": poses,
         "frame": 5}

    def run():
        t0 = time.perf_counter()
        near, far = rays["near"][sel].copy(), rays["far"][sel].copy()
        z = O.sample_gg(g["ray_o"], g["ray_d"], near, far, xyz, S, None, tv)["z_vals"]
        TO.render(params, g, jitter_z=z)
        return time.perf_counter() - t0

    hw = os.cpu_count() or 1
    best = None
    from dsnerf_amd import _lib
    quota = _lib.cpu_quota_cores()
    for threads in sorted({min(hw, 32), min(hw, 128)} | ({max(1, min(hw, int(quota)))} if quota else set())):
        torch.set_num_threads(threads)
        run()                                     # warm the pools
        dt = run()
        if best is None or dt < best[1]:
            best = (threads, dt)
    threads, dt = best
    cores = threads if not quota else max(1, min(threads, int(quota)))      # (threads beyond the cgroup's quota are not cores)
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "threads": threads, "cpu_quota_cores": quota, "kind": "port",
            "sample": f"one {n}-ray chunk of the same frame x {S} samples, torch {torch.__version__} CPU ops with {threads} threads "
                      f"(networks, autograd d sigma/dx, normals, lighting, compositing) + C-oracle geometry, {dt:.1f} s"}

