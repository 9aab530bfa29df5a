"""bench.py --dry-launch: launcher + collectives of the selected mode on the CPU over gloo, with a stand-in for the render."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import _flush_c_stdio


def dry_launch(args, rk):
    """--dry-launch: the launcher and the collectives of the selected mode, with a stand-in for the render, on the CPU over gloo.
    What it proves (tests/test_bench_launch.py): `bench.py --gpus N` starts N ranks; they join ONE process group and the same
    collectives the measured modes issue (weak: all_gather_into_tensor of one [R,6] frame per rank; --strong: round-robin tile deal,
    all-gather of equal slabs, un-dealing; --train: one flat all-reduce of the 33 gradients); every rank's pixels / gradients land
    where they belong; rank 0 prints ONE JSON line with n_gpus = N.  It measures nothing: `value` is null."""
    import dsnerf_amd
    dist = rk.dist
    world, rank = rk.world, rk.rank
    rp = dsnerf_amd.RayParallel()
    assert rp.world == world and rp.rank == rank
    R = 4096 if not args.strong else 10000                  # (strong: not a multiple of the tile, so the slabs are ragged)
    px_of = lambda rays, r_: torch.stack([rays.float() * (k + 1) + 1000.0 * r_ for k in range(6)], dim=1)      # any per-ray function
    checks = {}
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        if args.train:
            params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 500, 33)]
            for i, p_ in enumerate(params):
                p_.grad = torch.full_like(p_, float(rank + 1) * (i + 1))
            rp.average_gradients(params)
            want = sum(range(1, world + 1)) / world
            checks["gradients_are_the_mean_over_ranks"] = all(bool(torch.allclose(p_.grad, torch.full_like(p_, want * (i + 1))))
                                                              for i, p_ in enumerate(params))
        elif args.strong:
            fn = lambda o, d, n, f: {"color": px_of(o[:, 0], 0)[:, 0:3], "disp_map": px_of(o[:, 0], 0)[:, 3],
                                     "acc_map": px_of(o[:, 0], 0)[:, 4], "depth_map": px_of(o[:, 0], 0)[:, 5]}
            rays4 = (torch.arange(R)[:, None].float().expand(R, 3), torch.zeros(R, 3), torch.zeros(R), torch.zeros(R))
            if args.partition == "tiles":
                out = rp.render_tiled(fn, *rays4, tile=3072)
            else:       # cost-balanced contiguous blocks (the default): a lopsided cost, so the blocks - and the slabs - are ragged
                cost = (torch.arange(R).float() / R) ** 2 + 0.05
                out = rp.render_blocks(fn, *rays4, cost=cost)
                plan = rp.block_plan(R, rp.balanced_bounds(cost, world))
                checks["blocks_are_cost_balanced"] = max(plan["counts"]) > min(plan["counts"]) and \
                    max(float(cost[plan["bounds"][r_]:plan["bounds"][r_ + 1]].sum()) for r_ in range(world)) < 1.1 * float(cost.sum()) / world
            full = torch.cat([out["color"], out["disp_map"][:, None], out["acc_map"][:, None], out["depth_map"][:, None]], dim=1)
            checks["frame_reassembled_in_ray_order"] = bool(torch.equal(full, px_of(torch.arange(R), 0)))
        else:
            mine = px_of(torch.arange(R), rank)              # this rank's own frame of the batch
            allp = torch.empty(world * R, 6)
            if rk.on:
                dist.all_gather_into_tensor(allp, mine)
            else:
                allp.copy_(mine)
            checks["every_ranks_frame_present"] = all(bool(torch.equal(allp[r_ * R:(r_ + 1) * R], px_of(torch.arange(R), r_)))
                                                      for r_ in range(world))
    rk.barrier()
    dt, per_rank_s = rk.times(time.perf_counter() - t0)
    info = rk.info(per_rank_s, max(1, args.steps))
    ok = all(checks.values())
    if rk.on:                                                # every rank's verdict, not only rank 0's
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    rk.finish()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps({"metric": "DRY LAUNCH (launcher + collectives only, gloo on CPU, stand-in render): not a measurement",
                          "value": None, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
                          "vs_baseline": None, "dtype": None, "data": "none", "dry_launch": True,
                          "mode": "train" if args.train else ("strong" if args.strong else "weak"),
                          "checks": checks, "ok": ok, "ranks": info}), flush=True)
    if not ok:
        raise SystemExit(1)

