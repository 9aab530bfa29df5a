#!/usr/bin/env python3
"""bench.py - rendered rays/s of the Dual-Space-NeRF hot path on MI355X.

A "step" = one pass of the whole hot path (per-frame setup, geometry-guided sampling, nearest-face warp,
canonical field + d sigma/dx, normals + lighting MLP, compositing) over one synthetic 512x512 frame at
64 samples/ray (BASELINE.json configs[1]) per GPU, inputs resident in HBM.  With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank renders its own frame of the multi-frame batch
(configs[4], rays partitioned across GPUs in contiguous blocks = frames) and the rendered pixels are
exchanged with one RCCL all-gather inside the timed region: weak scaling, value = all rays / max time.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     - k_field16<forward> (the dominant kernel), timed live with HIP events on the launch stream in a
                 separate stage-by-stage pass over the same frame; algorithmic FLOPs = evaluated samples
                 x 0.91776 MFLOP (2 x 458 880 MAC: trunk + heads); the reverse kernel (analytic d sigma/dx,
                 2 x 425 728 MAC per sigma > 0 sample) is reported beside it.
  cpu_baseline - the C oracle (oracle/dsn_oracle.c, a port of the reference algorithm) timed on the host
                 cores on a bounded sample of the same frame (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FIELD_PER_SAMPLE = 2.0 * 884608.0       # k_field: forward trunk+heads 458 880 MAC + reverse 425 728 MAC
FLOP_FIELD_FWD_PER_SAMPLE = 2.0 * 458880.0   # k_field16<forward>: trunk + density/essence heads
FLOP_FIELD_REV_PER_SAMPLE = 2.0 * 425728.0   # k_field16<reverse>: analytic d sigma/dx
FLOP_ALL_PER_SAMPLE = 2.0 * 902272.0         # + lighting MLP 17 664 MAC
PEAK_F32_MATRIX_TFLOPS = 157.3               # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MATRIX_TFLOPS = 2500.0              # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
SPLIT_PRODUCTS = 3                           # split-fp16: 3 f16 MFMA products per algorithmic product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hw", type=int, default=512, help="image side (BASELINE configs[1]: 512)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--dense", action="store_true", help="evaluate the networks on every sample (no transparent skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=0, help="rays the CPU oracle is timed on (0 = sized for ~15 s)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="exact-fp32 MFMA field kernel instead of split-fp16")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback exists for the product path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import dsnerf_amd
    from dsnerf_amd import _lib, synth

    H = W = args.hw
    S = args.samples
    R = H * W
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    poses = synth.make_poses(seed=5 + rank)
    xyz = synth.pose_body(canon, seed=3 + rank)          # every rank renders its own frame of the batch
    rays = synth.make_rays(H, W, xyz, fit_box=True)    # every ray crosses the padded body AABB (= mask_at_box rays)

    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    ws = _lib.RenderWorkspace(dev)
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    d_xyz = torch.from_numpy(xyz).to(dev)
    d_poses = torch.from_numpy(poses).to(dev)
    ray_o = torch.from_numpy(rays["ray_o"]).to(dev)
    ray_d = torch.from_numpy(rays["ray_d"]).to(dev)
    near0 = torch.from_numpy(rays["near"]).to(dev)
    far0 = torch.from_numpy(rays["far"]).to(dev)
    near, far = near0.clone(), far0.clone()
    out = None
    gathered = torch.empty(world * R, 6, dtype=torch.float32, device=dev) if world > 1 else None
    packed_px = torch.empty(R, 6, dtype=torch.float32, device=dev)

    def step():
        nonlocal out
        near.copy_(near0)
        far.copy_(far0)
        scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)
        out = _lib.render_rays(scene, packed, ws, ray_o, ray_d, near, far, S, t_vals, None, None,
                               skip_transparent=not args.dense, want_weights=False, out=out, fp32=args.fp32)
        if world > 1:
            packed_px[:, 0:3] = out["color"]
            packed_px[:, 3] = out["disp_map"]
            packed_px[:, 4] = out["acc_map"]
            packed_px[:, 5] = out["depth_map"]
            dist.all_gather_into_tensor(gathered, packed_px)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    n_active = int(ws.buf[:4].view(torch.int32)[0]) if not args.dense else R * S
    n_pos = int(ws.buf[64:68].view(torch.int32)[0]) if (not args.dense and not args.fp32) else n_active
    ms_step = 1e3 * dt / args.steps
    value = world * R * args.steps / dt

    result = {
        "metric": "rendered rays/sec (64 samples/ray), 512x512 frame",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{H}x{W} frame x {S} samples/ray per GPU (BASELINE configs[1]; N>1: one frame per GPU, configs[4]), "
                        f"synthetic closed body V=6890/F=13776, camera framed so that all rays cross the body AABB (mask_at_box), "
                        f"GG sampling, eval mode",
            "rays_per_gpu": R, "samples_per_ray": S,
            "transparent_skip": (not args.dense),
            "evaluated_sample_fraction": n_active / float(R * S),
            "shaded_sample_fraction": n_pos / float(R * S),
            "ms_per_frame": ms_step,
            # every ray is fully rendered, so the dense-equivalent rate is `value`; this is the dense
            # algorithmic work of the frame (2 x 902 272 MAC x R x S) over the frame time
            "dense_equivalent_tflops": FLOP_ALL_PER_SAMPLE * R * S / (ms_step * 1e-3) / 1e12,
            "exchange": "all_gather_into_tensor [R,6] fp32 per rank (RCCL)" if world > 1 else "none",
        },
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        result["roofline"] = roofline(_lib, scene, packed, ray_o, ray_d, near0, far0, S, t_vals, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, S, args)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline(_lib, scene, packed, ray_o, ray_d, near0, far0, S, t_vals, args):
    """Stage-by-stage pass over the same frame; the field kernels are timed with HIP events on the launch stream
    (torch's current stream IS the stream every dsn_* call is enqueued on).  The dominant kernel of the frame is
    k_field16<forward> (all non-transparent samples); k_field16<reverse> runs on the sigma > 0 subset."""
    import ctypes as C
    R = ray_o.shape[0]
    N = R * S
    dev = scene.device
    L = _lib.lib()
    near, far = near0.clone(), far0.clone()
    pts, z = _lib.sample(scene, ray_o, ray_d, near, far, S, t_vals, None, want_pts=True)
    w = _lib.warp(scene, pts, ray_d, S, want_dir=False, want_active=not args.dense)
    lst, cnt = (None, None) if args.dense else (w["active_list"], w["active_count"])
    n_eval = N if args.dense else int(w["active_count"][0])
    reps = max(3, min(10, args.steps))
    sig = torch.zeros(N, device=dev)
    ess = torch.zeros(N, 3, device=dev)
    g = torch.zeros(N, 3, device=dev)
    a0 = (_lib._ptr(scene.buf), scene.V, scene.F, _lib._ptr(packed.buf), _lib._ptr(w["x_c"]), C.c_int64(N))

    def timed(fn, pre=None):
        ms = []
        for i in range(reps + 2):                      # 2 untimed warm-up launches
            if pre is not None:
                pre()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn()
            b.record()
            assert rc == 0, L.dsn_last_error()
            torch.cuda.synchronize()
            if i >= 2:
                ms.append(a.elapsed_time(b))
        return float(np.mean(ms))

    split = not (args.fp32 or args.dense)
    if split:
        rec = torch.empty(L.dsn_field_record_bytes(C.c_int64(N)), dtype=torch.uint8, device=dev)
        pos = torch.zeros(N, dtype=torch.int32, device=dev)
        pcnt = torch.zeros(64, dtype=torch.int32, device=dev)
        ms = timed(lambda: L.dsn_field_forward(*a0, _lib._ptr(lst), _lib._ptr(cnt), _lib._ptr(sig), _lib._ptr(ess),
                                               _lib._ptr(rec), _lib._ptr(pos), _lib._ptr(pcnt), _lib._stream()),
                   pre=lambda: pcnt.zero_())
        n_pos = int(pcnt[0])
        ms_rev = timed(lambda: L.dsn_field_reverse(*a0, _lib._ptr(pos), _lib._ptr(pcnt), _lib._ptr(rec), _lib._ptr(g),
                                                   _lib._stream()))
        flop_per, kern = FLOP_FIELD_FWD_PER_SAMPLE, "k_field16<forward>"
    else:
        ms = timed(lambda: L.dsn_field(*a0, _lib._ptr(lst), _lib._ptr(cnt), _lib._ptr(sig), _lib._ptr(ess), _lib._ptr(g),
                                       _lib.FIELD_FP32 if args.fp32 else 0, _lib._stream()))
        flop_per, kern = FLOP_FIELD_PER_SAMPLE, ("k_field" if args.fp32 else "k_field16<full>")
    ach = n_eval * flop_per / (ms * 1e-3) / 1e12
    if args.fp32:
        peak, note = PEAK_F32_MATRIX_TFLOPS, "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    else:
        # the algorithmic FLOPs are executed as 3 f16 MFMA products each: the ceiling for ALGORITHMIC FLOP/s of this
        # scheme is the dense f16 MFMA peak / 3 (= 5.3x the fp32-matrix peak of 157.3)
        peak = PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS
        note = "split-fp16: 3 x v_mfma_f32_32x32x16_f16 per product, fp32-equivalent accuracy; peak = 2500/3"
    out = {"bound": "mfma", "kernel": kern, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
           "frac": ach / peak, "traffic": measured_traffic(kern, args), "kernel_ms": ms, "samples_per_launch": n_eval,
           "flop_per_sample": flop_per, "scheme": note, "x_fp32_matrix_peak": ach / PEAK_F32_MATRIX_TFLOPS}
    if split:
        ach_r = n_pos * FLOP_FIELD_REV_PER_SAMPLE / (ms_rev * 1e-3) / 1e12
        out["reverse_kernel"] = {"kernel": "k_field16<reverse>", "kernel_ms": ms_rev, "samples_per_launch": n_pos,
                                 "flop_per_sample": FLOP_FIELD_REV_PER_SAMPLE, "achieved": ach_r, "frac": ach_r / peak}
    return out


def measured_traffic(kern, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc.json, written by scripts/pmc_summary.py: (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction
    of MI355X_MICROARCH.md); null when the counters were not collected for this configuration."""
    path = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if args.fp32 or args.dense or args.hw != 512 or args.samples != 64 or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f).get(kern)
    return None if rec is None else rec["hbm_bytes_per_launch"]


def cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, S, args):
    """The oracle (a C port of the reference algorithm) on the host cores, bounded sample of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    R = rays["ray_o"].shape[0]
    P = O.Params(sd)
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    cores = os.cpu_count() or 1

    def run(n):
        sel = np.linspace(0, R - 1, n).astype(np.int64)
        t0 = time.perf_counter()
        O.render(rays["ray_o"][sel], rays["ray_d"][sel], rays["near"][sel], rays["far"][sel], S, xyz, canon, faces, P,
                 poses, sd["nerf.embedding.weight"][5], t_vals=tv)
        return time.perf_counter() - t0

    t_cal = run(max(cores, 64))                      # calibration (also warms the OpenMP pool)
    n = int(np.clip(args.cpu_rays if args.cpu_rays > 0 else 15.0 * max(cores, 64) / t_cal, 128, 65536))
    dt = run(n)
    return {"value": n / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n} rays evenly spread over the same frame x {S} samples (dense evaluation, OpenMP over all "
                      f"host cores), {dt:.1f} s"}


if __name__ == "__main__":
    main()
