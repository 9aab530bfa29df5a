"""bench.py --train: BASELINE configs[2], the 8192 x 64 training step (render + MSE + backward + Adam) and its roofline."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import (FLOP_FIELD_PER_SAMPLE, PEAK_F16_MATRIX_TFLOPS, ROOT, SPLIT_PRODUCTS, load_weights, _flush_c_stdio, _profile_file)


TRAIN_DTYPE = ("split-f16x3 (k_field16<train>, k_tangent16, k_adjoint16, k_t_wgrad16c/p: 3 x v_mfma_f32_32x32x16_f16 per product, f32 "
               "accumulate) + exact-f32 MFMA for the small lighting / colour-head products (k_t_lin, k_t_wgrad)")


def train_measure(args, dsnerf_amd, synth, dev, world, rank, use_dist, steps, warmup, weights=None, per_rank=None):
    """trainer.py:66-81 on one synthetic batch per step: zero_grad, render (train mode: jitter + noise, dense), MSE,
    backward (dsn_render_rays_grad), Adam step.  With N>1 every rank renders its own 8192-ray batch of the step and
    the 33 gradients are averaged with ONE 2 MB RCCL all-reduce (parallel.RayParallel.average_gradients) before the
    optimizer step - plain data parallelism (the reference itself trains on one GPU).  Returns (seconds for `steps` steps - max over
    ranks -, final loss)."""
    from types import SimpleNamespace
    import torch.distributed as dist
    S, R = args.samples, args.train_rays
    canon, faces = synth.make_body()
    sd = load_weights(synth, weights or args.weights)
    pose_rank = rank if args.per_rank_frames == "different" else 0      # (the draws below differ per rank either way)
    xyz = synth.pose_body(canon, seed=3 + pose_rank)
    rays = synth.make_rays(args.hw, args.hw, xyz, fit_box=True)
    sel = np.linspace(0, args.hw * args.hw - 1, R).astype(np.int64)
    cfg = SimpleNamespace(DATASETS=SimpleNamespace(SMPL_PATH="<synthetic>"),
                          MODEL=SimpleNamespace(sample_points_mode="GG", COARSE_RAY_SAMPLING=S, perturb=1.0, raw_noise_std=1.0,
                                                TYPE="nerf", FINE_RAY_SAMPLING=-1))
    net = dsnerf_amd.DualSpaceNeRF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(dev)
    r = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev)
    r.train()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    batch = {"ray_o": T(rays["ray_o"][sel])[None], "ray_d": T(rays["ray_d"][sel])[None], "near": T(rays["near"][sel])[None],
             "far": T(rays["far"][sel])[None], "xyz": T(xyz)[None], "poses": T(synth.make_poses(seed=5 + pose_rank))[None],
             "Th": torch.zeros(1, 1, 3, device=dev), "frame": torch.tensor([5])}
    target = T(synth.hash_uniform(R * 3, 77).reshape(R, 3).astype(np.float32))
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    torch.manual_seed(233 + rank)
    loss = None
    rp = dsnerf_amd.RayParallel()

    def step():
        nonlocal loss
        opt.zero_grad()
        out = r.render(batch)["coarse"]
        loss = torch.nn.functional.mse_loss(out["color"], target)
        loss.backward()
        rp.average_gradients(net.parameters())
        opt.step()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, mine)
        every = [float(x) for x in every.cpu()]
        dt = max(every)
        if per_rank is not None:
            per_rank.extend(every)
    from dsnerf_amd import _lib
    rows = _lib.grad_row_counts(r._grad_ws, R, S)
    return dt, float(loss.detach()), int(r.range_overflow_count()), {"samples": R * S, "forward_rows": rows[0], "backward_rows": rows[1]}


def train_roofline(ms, R, S, rows=None, weights="default"):
    """Whole-step roofline of the training step, two ways.  `achieved` / `frac`: the DENSE-EQUIVALENT figure - the six trunk-sized
    contractions per sample (forward, sigma-reverse, tangent, adjoint and the two weight-gradient products per layer: 6 x 884 608
    MAC) for EVERY sample of the batch over the step time.  `achieved_on_evaluated_rows` / `frac_on_evaluated_rows` (VERDICT r03
    weak #1): the same contractions counted only on the rows the step really evaluates - forward + sigma-reverse on the forward's
    rows (all but transparent samples with noise <= 0), the other four on the rows with a non-zero cotangent; the skipped rows add
    exactly nothing to any output, so this is the work done, and this is the honest fraction of the split-fp16 ceiling.
    hbm_gb_per_step: from the committed PMC passes of `bench.py --train` (profiles/rNN_train_pmc.json), not measured in this run."""
    flop = 3.0 * FLOP_FIELD_PER_SAMPLE * R * S          # 3 x (2 x 884 608 MAC) = 5.31 MFLOP per sample
    ach = flop / (ms * 1e-3) / 1e12
    peak = PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS
    out = {"bound": "mfma", "kernel": "whole training step (k_field16<train> + k_tangent16 + k_adjoint16 + weight-gradient kernels)",
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
           "achieved_is": "dense-equivalent: every sample of the batch counted, skipped rows included",
           "flop_per_sample": flop / (R * S), "samples_per_step": R * S}
    if rows:
        f_rows = FLOP_FIELD_PER_SAMPLE * rows["forward_rows"] + 2.0 * FLOP_FIELD_PER_SAMPLE * rows["backward_rows"]
        a_rows = f_rows / (ms * 1e-3) / 1e12
        out.update({"achieved_on_evaluated_rows": a_rows, "frac_on_evaluated_rows": a_rows / peak,
                    "forward_rows": rows["forward_rows"], "backward_rows": rows["backward_rows"]})
    path = _profile_file("train_pmc.json" if weights == "default" else f"train_{weights}_pmc.json")
    if path is not None and R * S == 8192 * 64:
        with open(path) as f:
            gb = json.load(f).get("_hbm_gb_per_step")
        if gb is not None:
            out["hbm_gb_per_step"] = gb
            out["traffic"] = gb * 1e9
            out["traffic_source"] = f"{os.path.relpath(path, ROOT)} (committed rocprofv3 --pmc passes of `bench.py --train`, all kernels of a step; not collected in this run)"
    return out


def train_bench(args, dsnerf_amd, synth, dev, world, rank, use_dist, rk):
    import torch.distributed as dist
    S, R = args.samples, args.train_rays
    per_rank_s = []
    dt, final_loss, ovf, rows = train_measure(args, dsnerf_amd, synth, dev, world, rank, use_dist, args.steps, args.warmup, per_rank=per_rank_s)
    ranks = rk.info(per_rank_s or [dt], args.steps)
    rk.finish()
    if rank == 0:
        ms = 1e3 * dt / args.steps
        _flush_c_stdio()
        print(json.dumps({
            "metric": "training rays/sec (64 samples/ray, forward + backward + Adam step)", "value": world * R * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": TRAIN_DTYPE, "data": "synthetic",
            "config": {"workload": f"training step on {R} rays x {S} samples per GPU (BASELINE configs[2]), dense evaluation "
                                   f"(jitter + noise), synthetic body V=6890/F=13776", "final_loss": final_loss,
                       "range_overflow_samples_last_step": ovf, "rows_last_step": rows,
                       "rows_note": "the forward skips transparent samples with noise <= 0 (alpha = 0 exactly), the backward every row "
                                    "whose cotangents are all zero; the roofline counts the DENSE algorithmic work of the batch"},
            "roofline": train_roofline(ms, R, S, rows, args.weights), "ranks": ranks}), flush=True)

