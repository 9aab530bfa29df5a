#!/usr/bin/env python3
"""Train the "w4" parameter set: a CONVERGED Dual-Space-NeRF on the synthetic full-size body.

Runs on the MI355X box (the HIP trainer: Renderer.render in train mode -> dsn_render_rays_train / dsn_render_rays_grad,
whose gradients are pinned to the reference's own autograd in tests/test_gpu_train.py):

    /usr/local/graft/bin/gpurun --timeout 1500 -- 'python scripts/train_w4.py --steps 24000'

What is trained is what trainer.py:66-81 trains: Renderer.render(batch) on random rays of ONE camera per step (the
geometry-guided sampler uses the batch's first ray origin for every ray, utils/pts_utils.py:31), the reference's loss
(utils/loss.py MSELoss: L2 on colour + 0.1 x L1 occupancy term, LOSSwMask), Adam with the betas of solver/build.py and the
learning rate of configs/zju_mocap/313.yml (5e-4, warm-up, exponential decay to 0.09 x).  Data: an analytic multi-view
target - the posed body thickened by 3 cm (a ray hits when it passes within 3 cm of a posed vertex), textured by a smooth
position-dependent colour, black background - seen from N cameras on rings around the body (the bench's camera is camera 0).
Held-out cameras (between the training yaws) give the PSNR that says whether the field has converged.

Writes tests/golden/weights_w4.npz (the 33 tensors, float32) + gpurun_out/w4_train_log.json (loss / PSNR curve, density
statistics of the bench frame).  The goldens of this set are then generated in the build container by the REAL reference:
tests/golden/make_golden.py --other-weights w4.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def target_image(o, d, xyz, radius=0.03, chunk=4096):
    """analytic ground truth on the device: hit = the ray passes within `radius` of a posed vertex; colour = texture(first such
    vertex along the ray).  o, d [R,3] float32 (device), xyz [V,3].  Returns rgb [R,3], occupancy [R] (float32)."""
    dn = d / d.norm(dim=-1, keepdim=True)
    R = o.shape[0]
    rgb = torch.zeros(R, 3, device=o.device)
    occ = torch.zeros(R, device=o.device)
    v = xyz.double()
    for s in range(0, R, chunk):
        w = v[None, :, :] - o[s:s + chunk, None, :].double()
        t = (w * dn[s:s + chunk, None, :].double()).sum(-1)
        rho2 = (w * w).sum(-1) - t * t
        hit = rho2 < radius ** 2
        tt = torch.where(hit, t, torch.full_like(t, float("inf")))
        k = tt.argmin(1)
        any_hit = hit.any(1)
        p = v[k]
        tex = 0.5 + 0.5 * torch.stack([torch.sin(23.0 * p[:, 0] + 1.0), torch.sin(17.0 * p[:, 1] - 2.0), torch.sin(29.0 * p[:, 2] + 0.5)], -1)
        rgb[s:s + chunk] = torch.where(any_hit[:, None], tex, torch.zeros_like(tex)).float()
        occ[s:s + chunk] = any_hit.float()
    return rgb, occ


def grad_workspace_views(buf, N):
    """name -> float32 view of the training backward's workspace (mirror of carve() in csrc/dsn_train.hip; debugging only)"""
    al = lambda b: (b + 255) & ~255
    order = [("transparent", 1), ("idx_c", 4), ("x_c", 12), ("pe", 256)] + [(f"h{l}", 1024) for l in range(7)] + \
            [(f"ap{l}", 1024) for l in range(7)] + [(f"tn{l}", 1024) for l in range(7)] + [("masks", 224), ("rr", 512), ("ess", 12), ("sig", 4),
            ("g", 12), ("t0", 1024), ("tpe", 256), ("n_w", 12), ("xl", 36), ("hl1", 512), ("hl2", 512), ("pre", 4), ("wl", 4), ("col", 12),
            ("d_sig", 4), ("d_col", 12), ("d_ess", 12), ("d_pre", 4), ("d_hl2", 512), ("d_hl1", 512), ("d_xl", 36), ("d_rr", 512), ("u", 12),
            ("scratch_t", 4)]
    out, off = {}, 0
    for name, per in order:
        nb = per * N
        if name not in ("transparent", "idx_c", "masks"):
            out[name] = buf[off:off + nb].view(torch.float32).view(N, -1)
        off += al(nb)
    out["small"] = buf[off:off + 4096].view(torch.float32)
    return out


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return -10.0 * math.log10(max(mse, 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24000)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--cams", type=int, default=12)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--debug-nan", action="store_true", help="check every step's outputs / gradients for non-finite values and stop at the first")
    ap.add_argument("--init", default="default", help="default | path to an .npz with w:<name> arrays (continue a run)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "weights_w4.npz"))
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "w4_train_log.json"))
    args = ap.parse_args()

    import dsnerf_amd
    from dsnerf_amd import _lib, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    S = 64
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon, seed=3)
    poses = synth.make_poses(seed=5)
    if args.init == "default":
        sd = synth.make_state_dict()
    else:
        z = np.load(args.init)
        sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0,
                                                TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_xyz = T(xyz)
    r.density_screen = False          # (the periodic evaluations: no per-checkpoint calibration / probe frames while the parameters move)
    r.early_stop = False

    def camera(yaw, pitch, hw=args.hw):
        rays = synth.make_rays(hw, hw, xyz, yaw=yaw, pitch=pitch)
        sel = np.nonzero(rays["hit_box"])[0]                      # the reference trains on rays inside the body's bounds only
        c = {k: T(rays[k][sel]) for k in ("ray_o", "ray_d", "near", "far")}
        c["rgb"], c["occ"] = target_image(c["ray_o"], c["ray_d"], d_xyz)
        return c

    pitches = (-0.12, 0.30, -0.40)
    cams = [camera(0.35 + 2.0 * math.pi * k / args.cams, pitches[k % 3]) for k in range(args.cams)]
    held = [camera(0.35 + 2.0 * math.pi * (k + 0.5) / args.cams, pitches[(k + 1) % 3], hw=128) for k in (0, 3, 7)]
    print(f"{len(cams)} training cameras, {sum(c['rgb'].shape[0] for c in cams)} rays, occupancy "
          f"{np.mean([float(c['occ'].mean()) for c in cams]):.3f}; {len(held)} held-out cameras", flush=True)
    common = {"xyz": d_xyz[None], "poses": T(poses)[None], "Th": torch.zeros(1, 1, 3, device=dev), "frame": torch.tensor([5])}

    def batch_of(c, idx=None):
        g = (lambda t: t) if idx is None else (lambda t: t[idx])
        b = dict(common)
        b.update({"ray_o": g(c["ray_o"])[None].contiguous(), "ray_d": g(c["ray_d"])[None].contiguous(),
                  "near": g(c["near"]).clone()[None].contiguous(), "far": g(c["far"]).clone()[None].contiguous()})
        return b

    def evaluate(c):
        r.eval()
        with torch.no_grad():
            out = r.render(batch_of(c))["coarse"]
        r.train()
        return psnr(out["color"], c["rgb"]), float((out["acc_map"] - c["occ"]).abs().mean())

    opt = torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.999), weight_decay=0.0)   # solver/build.py:9-11
    warm, start, scale = 500, 2000, 0.09                                                          # 313.yml, compressed to this run

    def lr_at(it):
        f = min(1.0, (1.0 / 3.0) + (2.0 / 3.0) * it / warm)
        if it > start:
            f *= scale ** ((it - start) / max(1, args.steps - start))
        return args.lr * f

    torch.manual_seed(233)                                     # main.py:21-26 (the CPU generator feeds jitter and noise)
    gsel = torch.Generator(device=dev)
    gsel.manual_seed(2330)
    r.train()
    log = {"args": vars(args), "curve": []}
    t0 = time.time()
    run_loss, run_n = 0.0, 0
    for it in range(args.steps):
        c = cams[it % len(cams)]
        idx = torch.randint(0, c["rgb"].shape[0], (args.rays,), device=dev, generator=gsel)
        for gparam in opt.param_groups:
            gparam["lr"] = lr_at(it)
        opt.zero_grad(set_to_none=True)
        out = r.render(batch_of(c, idx))["coarse"]                 # trainer.py:70
        occ = c["occ"][idx]
        loss_rgb = torch.nn.functional.mse_loss(out["color"], c["rgb"][idx])
        loss_mask = 0.1 * (out["acc_map"] * (1.0 - occ)).abs().mean()      # utils/loss.py:19-24 (acc_map[occ == 1] = 1; L1 vs occupancy)
        loss = loss_rgb + loss_mask
        loss.backward()
        if args.debug_nan:
            bad = {k: int((~torch.isfinite(p.grad)).sum()) for k, p in net.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())}
            if it % 25 == 0 or bad:
                gn = {k: float(p.grad.norm()) for k, p in net.named_parameters() if k in ("nerf.stage1.0.weight", "nerf.density_net.0.weight", "lighting_mlp.lights_encoding.0.weight", "pose_mlp.0.weight")}
                print(json.dumps({"step": it, "loss_rgb": float(loss_rgb), "loss_mask": float(loss_mask), "acc_mean": float(out["acc_map"].mean()),
                                  "color_finite": bool(torch.isfinite(out["color"]).all()), "overflow": r.range_overflow_count(), "grad_norms": gn,
                                  "max_w": max(float(p.detach().abs().max()) for p in net.parameters())}), flush=True)
            if bad:
                print("NON-FINITE GRADIENTS at step", it, json.dumps(bad), flush=True)
                vw = grad_workspace_views(r._grad_ws.buf, args.rays * S)
                rep = {}
                for k, t in vw.items():
                    fin = torch.isfinite(t)
                    if not bool(fin.all()):
                        rows = (~fin).reshape(t.shape[0], -1).any(1).nonzero().reshape(-1) if t.dim() == 2 else None
                        rep[k] = {"nonfinite": int((~fin).sum()), "rows": None if rows is None else rows[:4].tolist()}
                    else:
                        rep[k] = {"max": float(t.abs().max())}
                print(json.dumps(rep), flush=True)
                bad_rows = [v["rows"][0] for v in rep.values() if v.get("rows")]
                if bad_rows:
                    n = bad_rows[0]
                    print("sample", n, {k: vw[k][n].flatten()[:9].tolist() for k in ("x_c", "g", "n_w", "u", "d_xl", "sig", "pre", "tpe", "tn0")}, flush=True)
                    print("small[300:304]", vw["small"][300:304].tolist(), flush=True)
                print("depth finite", bool(torch.isfinite(out["depth_map"]).all()), "weights finite", bool(torch.isfinite(out["weights"]).all()), flush=True)
                return
        opt.step()
        run_loss += float(loss_rgb.detach()) if (it % 50 == 0) else 0.0
        run_n += 1 if (it % 50 == 0) else 0
        if it % 1000 == 0 or it == args.steps - 1:
            ovf = r.range_overflow_count()
            ps = [evaluate(h) for h in held]
            rec = {"step": it, "loss_rgb": run_loss / max(run_n, 1), "lr": lr_at(it), "heldout_psnr": [p[0] for p in ps],
                   "heldout_acc_l1": [p[1] for p in ps], "range_overflow": ovf, "seconds": time.time() - t0}
            log["curve"].append(rec)
            print(json.dumps(rec), flush=True)
            run_loss, run_n = 0.0, 0
    torch.cuda.synchronize()
    log["train_seconds"] = time.time() - t0
    log["ms_per_step"] = 1e3 * log["train_seconds"] / args.steps

    # what the trained field looks like on the bench frame (512 x 512 x 64, every ray crosses the body's box)
    out_sd = {k: v.detach().float().cpu().numpy().copy() for k, v in net.state_dict().items()}
    arrs = {"w:" + k: v for k, v in out_sd.items()}
    arrs["heldout_psnr"] = np.asarray([c["heldout_psnr"] for c in log["curve"]], np.float64)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(args.out, **arrs)
    side = os.path.join(ROOT, "gpurun_out", os.path.basename(args.out))      # the copy that travels back from the GPU box
    np.savez_compressed(side, **arrs)
    print(f"wrote {args.out} and {side} ({os.path.getsize(side) / 1024:.0f} KiB)", flush=True)

    rays = synth.make_rays(512, 512, xyz, fit_box=True)
    packed = net.packed(dev)
    ws = _lib.RenderWorkspace(dev)
    scene = r.scene
    scene.set_frame(packed, d_xyz, T(poses), 5, False, None, None, None)
    o_, d_, n_, f_ = (T(rays[k]) for k in ("ray_o", "ray_d", "near", "far"))
    tv = torch.linspace(0.0, 1.0, steps=S).to(dev)
    o = _lib.render_rays(scene, packed, ws, o_, d_, n_, f_, S, tv, None, None, screen=False, stop_stats=True)
    torch.cuda.synchronize()
    N = 512 * 512 * S
    st = _lib.read_stop_stats(ws)
    cnt = ws.buf[:256].view(torch.int32).cpu()
    stats = {"non_transparent_fraction": int(cnt[_lib.CNT_ACTIVE]) / N, "positive_density_fraction_of_non_transparent":
             int(cnt[_lib.CNT_POS]) / max(int(cnt[_lib.CNT_ACTIVE]), 1),
             "early_stop_would_skip_fraction": st["would_skip"] / max(st["active"], 1),
             "acc_mean": float(o["acc_map"].mean()), "acc_gt_0.99": float((o["acc_map"] > 0.99).float().mean()),
             "acc_lt_0.01": float((o["acc_map"] < 0.01).float().mean())}
    tr, occ_b = target_image(o_, d_, d_xyz)
    stats["bench_frame_psnr_vs_target"] = psnr(o["color"], tr)
    stats["bench_frame_acc_l1_vs_occupancy"] = float((o["acc_map"] - occ_b).abs().mean())
    stats["max_abs_weight"] = {k: float(np.abs(v).max()) for k, v in out_sd.items()}
    log["bench_frame"] = stats
    print(json.dumps(stats), flush=True)
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    with open(args.log, "w") as f:
        json.dump(log, f, indent=1)


if __name__ == "__main__":
    main()
