"""bench.py --strong: ONE frame partitioned over the ranks (strong scaling), and its one-GPU emulation (--emulate-world N).

The frame is the metric's own by default - 512 x 512 x 64 (BASELINE configs[1], what `--gpus N` of the headline line renders on one
GPU); --big-frame gives BASELINE configs[3], 1024 x 1024 x 128.  Rays are partitioned (north_star: "partition rays across the GPUs
with an RCCL all-gather of rendered pixels"), every rank renders its share with `--pipeline` frames in flight exactly as the N = 1
line does, and ONE all_gather_into_tensor of equal slabs of packed [rays, 6] pixels + ONE index_select back into ray order bring
every frame together on every rank, inside the timed region.  value = rays of the frame x frames / time.

Partitions (--partition):
  blocks (default)  contiguous ray blocks cut where the cumulative per-ray cost (evaluated + shaded samples of a probe frame) crosses
                    k / N: a rank's samples stay in a compact part of space - its cells of the posed mesh's nearest-face grid are
                    its own (the per-frame list build covers only visited cells), its cell-major search runs on full waves
  tiles             round-robin tiles (RayParallel.tile_indices): every rank gets the same mix of cheap and expensive rows without
                    any cost estimate, but an N-th of the samples spread over ALL the cells the frame visits
"""
from __future__ import annotations

import json
import time

import numpy as np
import torch

from .common import LAZY_LISTS, load_weights, _flush_c_stdio
from .frame import stop_setup

XGMI_LINK_GBS = 153.0         # MI355X_MICROARCH.md: per-link xGMI bandwidth (7 links per GPU); used ONLY to price the emulation's gather
XGMI_LATENCY_MS = 0.03


def strong_shape(args):
    """(H, W, S) of the partitioned frame: --hw / --samples as given (defaults: the metric's 512 x 512 x 64); --big-frame = configs[3]"""
    if getattr(args, "big_frame", False):
        return 1024, 1024, 128
    return args.hw, args.hw, args.samples


def auto_tile(R, world, tile):
    """round-robin tile size: `tile` rays at most, and every rank owns the same NUMBER of tiles (86 tiles of 3072 rays dealt to 8 ranks
    leave 11 with six of them and 10 with two: 10 % imbalance before any ray is rendered)"""
    if world <= 1:
        return int(tile)
    ntiles = world * max(1, -(-R // (world * int(tile))))
    return -(-R // ntiles)


class Frame:
    """the synthetic frame of the strong modes + everything that is the same for every share: parameters, posed mesh, rays on the host"""

    def __init__(self, args, _lib, synth, dev):
        self.H, self.W, self.S = strong_shape(args)
        self.R = self.H * self.W
        self.canon, self.faces = synth.make_body()
        sd = load_weights(synth, args.weights)
        self.xyz = synth.pose_body(self.canon, seed=3)
        self.rays = synth.make_rays(self.H, self.W, self.xyz, fit_box=True)
        self.dev = dev
        self.packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
        T = self.T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.d_xyz, self.d_poses = T(self.xyz), T(synth.make_poses(seed=5))
        self.t_vals = torch.linspace(0.0, 1.0, steps=self.S).to(dev)

    def rays_of(self, idx):
        r = self.rays
        return self.T(r["ray_o"][idx]), self.T(r["ray_d"][idx]), self.T(r["near"][idx]), self.T(r["far"][idx])


class Slots:
    """`depth` frames in flight: own scene blob, workspace and stream each (what Renderer.render_views keeps)"""

    def __init__(self, _lib, frame, depth):
        dev = frame.dev
        self.depth = depth
        self.scenes = [_lib.Scene(torch.from_numpy(frame.canon), torch.from_numpy(frame.faces), dev) for _ in range(depth)]
        self.wss = [_lib.RenderWorkspace(dev) for _ in range(depth)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]


class Share:
    """one rank's rays of the frame, rendered the way the N = 1 line renders its whole frame: per-frame set-up (posed-mesh lists),
    sampler, warp, field, shading, compositing, then the six pixel planes packed into one [slab, 6] tensor for the exchange"""

    def __init__(self, _lib, args, frame, slots, idx, slab=None, screen=False):
        self._lib, self.args, self.f, self.sl = _lib, args, frame, slots
        self.idx = np.asarray(idx)
        self.Rl = int(len(self.idx))
        self.o, self.d, self.near0, self.far0 = frame.rays_of(self.idx)
        dev = frame.dev
        slab = self.Rl if slab is None else int(slab)
        self.nears = [self.near0.clone() for _ in range(slots.depth)]
        self.fars = [self.far0.clone() for _ in range(slots.depth)]
        self.outs = [None] * slots.depth
        self.px = [torch.zeros(slab, 6, dtype=torch.float32, device=dev) for _ in range(slots.depth)]
        self.screen = bool(screen)
        self.guard_host = [torch.empty(256, dtype=torch.uint8).pin_memory() for _ in range(slots.depth)]
        self.stop_on, self.schedule, self.stop_info = False, None, {"enabled": False}
        for j in range(slots.depth):      # (set-up: every slot's workspace exists and has been touched before anything is timed)
            slots.wss[j].get(self.Rl, frame.S).zero_()

    def decide_stop(self, reduce_max=None, frame_decision=None):
        """front-to-back slices with ray termination, decided like Renderer does from one probe render of THIS share (slice schedule
        from its own histogram); frame_decision = (enabled, colour scale) of the whole frame overrides the share's own decision - the
        ranks of a real run agree on those two by all-reduce (reduce_max), the emulation probes the whole frame once"""
        f, sl = self.f, self.sl
        sl.scenes[0].set_frame(f.packed, f.d_xyz, f.d_poses, 5, False, None, None, None)
        on, sched, info = stop_setup(self._lib, self.args, sl.scenes[0], f.packed, sl.wss[0], self.o, self.d, self.near0, self.far0, f.S,
                                     f.t_vals, self.screen, reduce_max=reduce_max, more_ws=sl.wss[1:])
        if frame_decision is not None:
            on = bool(frame_decision[0])
            f.packed.set_early_stop_colour_scale(frame_decision[1])
            sched = sched if on else None
        self.stop_on, self.schedule, self.stop_info = on, sched, info
        for j in range(sl.depth):         # (the probe may have asked for more records: every slot's workspace at its final size)
            sl.wss[j].begin_frame()
            sl.wss[j].get(self.Rl, f.S)
        return on

    def render(self, j, share_cus=False):
        """one frame of this share in slot j, on torch's current stream; leaves the packed pixels in self.px[j][:Rl]"""
        f, sl, L = self.f, self.sl, self._lib
        self.nears[j].copy_(self.near0)
        self.fars[j].copy_(self.far0)
        sl.scenes[j].set_frame(f.packed, f.d_xyz, f.d_poses, 5, False, None, None, None, fine_only=True, lazy=LAZY_LISTS)
        out = self.outs[j] = L.render_rays(sl.scenes[j], f.packed, sl.wss[j], self.o, self.d, self.nears[j], self.fars[j], f.S, f.t_vals,
                                           None, None, want_weights=False, out=self.outs[j], screen=self.screen,
                                           early_stop=self.stop_on, stop_schedule=self.schedule, share_cus=share_cus)
        if self.stop_on:      # (Renderer's hand-over check of a sliced frame: its counter words to page-locked memory, behind the frame)
            self.guard_host[j].copy_(sl.wss[j].buf[:256], non_blocking=True)
        px = self.px[j]
        px[:self.Rl, 0:3] = out["color"]
        px[:self.Rl, 3] = out["disp_map"]
        px[:self.Rl, 4] = out["acc_map"]
        px[:self.Rl, 5] = out["depth_map"]
        return px

    def counters(self, j=0):
        return self.sl.wss[j].buf[:256].view(torch.int32).cpu()


def ray_costs(_lib, args, frame, slots, screen):
    """per-ray cost estimate of the whole frame for the balanced blocks (set-up, not a step): one one-pass probe render with
    weights, then cost = samples the sliced frame evaluates (non-transparent, ray still alive) + 0.85 x samples it shades (weight
    above the threshold: reverse pass, normal, lighting) + 3 (the ray's share of sampler / search / compositing, in units of one
    evaluated sample).  Only the RATIO between rays matters."""
    f = frame
    S, R = f.S, f.R
    o, d, near0, far0 = f.rays_of(np.arange(R))
    sc, ws = slots.scenes[0], slots.wss[0]
    sc.set_frame(f.packed, f.d_xyz, f.d_poses, 5, False, None, None, None)
    out = _lib.render_rays(sc, f.packed, ws, o, d, near0.clone(), far0.clone(), S, f.t_vals, None, None, want_weights=True, screen=screen)
    n2, f2 = near0.clone(), far0.clone()
    pts, _ = _lib.sample(sc, o, d, n2, f2, S, f.t_vals, None, want_pts=True)
    tr = _lib.warp(sc, pts, d, S, want_dir=False)["transparent"].reshape(R, S).bool()
    w = out["weights"].reshape(R, S)
    eps = _lib.early_stop_eps(S, f.packed.colour_scale)
    T = 1.0 - (torch.cumsum(w, 1) - w)                      # transmittance in front of every sample
    alive = T >= eps
    ev = (~tr) & alive
    cost = ev.sum(1).double() + 0.85 * (ev & (w >= eps)).sum(1).double() + 3.0
    del pts, tr, w, T, alive, ev
    return cost.cpu()


def make_partition(args, _lib, rp, frame, slots, world, screen, log=None, agree=None):
    """-> (kind, plan builder (device -> plan), info dict).  world = ranks of the (real or emulated) job.
    agree(bounds) -> bounds: a real multi-rank job passes rank 0's block bounds through here (one broadcast), so that the slab size of
    the gather cannot depend on a borderline comparison falling differently on two GPUs."""
    R = frame.R
    if args.partition == "tiles":
        tile = auto_tile(R, world, args.tile)
        return "tiles", (lambda dev, rank=None: _tile_plan(rp, R, tile, dev, world, rank)), {"partition": "tiles", "tile_rays": tile}
    cost = ray_costs(_lib, args, frame, slots, screen)
    bounds = rp.balanced_bounds(cost, world)
    if agree is not None:
        bounds = agree(bounds)
    info = {"partition": "blocks", "bounds": bounds, "cost_share_of_blocks": [float(cost[bounds[r]:bounds[r + 1]].sum() / cost.sum())
                                                                                for r in range(world)]}
    return "blocks", (lambda dev, rank=None: _block_plan(rp, R, bounds, dev, rank)), info


def _tile_plan(rp, R, tile, dev, world, rank):
    plan = dict(rp.tile_plan(R, tile, dev, world=world))
    if rank is not None:
        plan["mine"] = rp.tile_indices(R, tile, rank, world).to(dev)
    return plan


def _block_plan(rp, R, bounds, dev, rank):
    plan = dict(rp.block_plan(R, bounds, dev))
    if rank is not None:
        plan["mine"] = torch.arange(bounds[rank], bounds[rank + 1], device=dev)
    return plan


def strong_bench(args, dsnerf_amd, _lib, synth, dev, world, rank, use_dist, rk):
    import torch.distributed as dist
    if args.emulate_world > 1 and world == 1:
        return strong_emulated(args, dsnerf_amd, _lib, synth, dev)
    frame = Frame(args, _lib, synth, dev)
    H, W, S, R = frame.H, frame.W, frame.S, frame.R
    depth = max(1, args.pipeline)
    slots = Slots(_lib, frame, depth)
    rp = dsnerf_amd.RayParallel()
    slots.scenes[0].set_frame(frame.packed, frame.d_xyz, frame.d_poses, 5, False, None, None, None)
    info = frame.packed.calibrate_screen(slots.scenes[0]) if args.screen else {"usable": False, "note": "density screen not opted in (--screen)"}
    # (every rank computes the partition from the same synthetic frame; rank 0's block bounds are the ones used - one 8-byte-per-rank
    #  broadcast outside the timed region, so that no rank can disagree about the slab size of the gather)
    def agree(bounds):
        if not use_dist:
            return bounds
        t_ = torch.tensor(bounds, dtype=torch.int64, device=dev)
        dist.broadcast(t_, src=0)
        return [int(x) for x in t_.cpu()]
    kind, plan_of, part_info = make_partition(args, _lib, rp, frame, slots, world, info["usable"], agree=agree)
    plan = plan_of(dev)
    slab = plan["slab"]
    share = Share(_lib, args, frame, slots, plan["mine"].cpu().numpy(), slab=slab, screen=info["usable"])
    share.decide_stop(reduce_max=(lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.MAX)) if use_dist else None)
    allp = [torch.empty(world * slab, 6, dtype=torch.float32, device=dev) for _ in range(depth)]
    full = [torch.empty(R, 6, dtype=torch.float32, device=dev) for _ in range(depth)]
    share_cus = depth > 1
    k_step = 0

    def step():
        nonlocal k_step
        j = k_step % depth
        k_step += 1
        with torch.cuda.stream(slots.streams[j]):
            px = share.render(j, share_cus=share_cus)
            if use_dist:
                dist.all_gather_into_tensor(allp[j], px)
                rp.undeal(allp[j], plan, out=full[j])      # ONE index_select through the cached permutation
            else:
                rp.undeal(px, plan, out=full[j])

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(depth):          # set-up: every slot once (its one-off costs), then W warm-up steps and exactly K timed steps
        step()
    barrier()
    k_step = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt, per_rank_s = rk.times(dt)
    cnt = share.counters((k_step - 1) % depth)
    ms = 1e3 * dt / args.steps
    res = {"metric": f"rendered rays/sec ({S} samples/ray), {H}x{W} frame" + (" split over the GPUs" if world > 1 else ""),
           "value": R * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "split-f16x3 (3 x v_mfma_f32_32x32x16_f16 on hi/lo fp16 operand halves, f32 accumulate: f32-equivalent accuracy)"
                    + (" + plain-f16 density screen" if info["usable"] else ""),
           "data": "synthetic",
           "config": {"workload": f"ONE {H}x{W} frame x {S} samples/ray per step "
                                  + ("(BASELINE configs[1], the metric's frame)" if (H, S) == (512, 64) else "(BASELINE configs[3])")
                                  + f", its rays partitioned over {world} GPU(s) ({kind}), {share.Rl} rays on rank 0, {depth} frame(s) in "
                                  f"flight, synthetic closed body V=6890/F=13776, all rays cross the body AABB, GG sampling, eval mode, "
                                  f"parameters: {args.weights}",
                      "weights": args.weights, "early_stop": share.stop_info, "partition": part_info, "rays_per_rank": plan["counts"],
                      "rays_on_rank0": share.Rl, "samples_per_ray": S, "ms_per_frame": ms, "frames_in_flight": depth,
                      "non_transparent_sample_fraction_rank0": int(cnt[_lib.CNT_ACTIVE]) / float(max(1, share.Rl) * S),
                      "density_screen_calibration": info,
                      "exchange": (f"all_gather_into_tensor [{slab},6] fp32 per rank (RCCL) + index_select to ray order, in the timed region"
                                   if use_dist else "none (index_select to ray order only)")},
           "ranks": rk.info(per_rank_s, args.steps)}
    if world > 1 and not args.no_extras:
        # the weak-scaling line of the same run, as a secondary object (what `--weak` makes the headline): every rank renders the WHOLE
        # frame - one frame of a multi-frame batch per rank, BASELINE configs[4] - and one all-gather of [R,6] pixels per frame follows
        del share, allp, full
        res["weak_scaling_same_run"] = weak_secondary(args, _lib, frame, slots, world, use_dist, rk, info["usable"])
    rk.finish()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(res), flush=True)


def weak_secondary(args, _lib, frame, slots, world, use_dist, rk, screen):
    import torch.distributed as dist
    R, depth = frame.R, slots.depth
    sh = Share(_lib, args, frame, slots, np.arange(R), screen=screen)
    sh.decide_stop(reduce_max=(lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.MAX)) if use_dist else None)
    gathered = [torch.empty(world * R, 6, dtype=torch.float32, device=frame.dev) for _ in range(depth)]
    k = 0

    def run(n):
        nonlocal k
        for _ in range(n):
            j = k % depth
            k += 1
            with torch.cuda.stream(slots.streams[j]):
                px = sh.render(j, share_cus=depth > 1)
                if use_dist:
                    dist.all_gather_into_tensor(gathered[j], px)

    steps = max(3, min(args.steps, 10))
    run(depth)
    rk.barrier()
    run(max(1, args.warmup))
    rk.barrier()
    t0 = time.perf_counter()
    run(steps)
    rk.barrier()
    dt, per_rank = rk.times(time.perf_counter() - t0)
    return {"metric": "weak scaling: one whole frame per GPU (BASELINE configs[4]), one all-gather of [R,6] pixels per frame",
            "value": world * R * steps / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "scaling": "weak",
            "per_rank_ms_per_step": [1e3 * t / steps for t in per_rank]}


def gather_ms_priced(R, world):
    """all-gather of N equal slabs of 24 B x R / N, PRICED (one GPU cannot measure it): every rank receives (N - 1) / N of the frame
    over its N - 1 direct xGMI links (MI355X_MICROARCH.md: 153 GB/s per link) + a launch latency of 30 us.  A stated estimate."""
    if world <= 1:
        return 0.0
    return XGMI_LATENCY_MS + 1e3 * (24.0 * R * (world - 1) / world) / ((world - 1) * XGMI_LINK_GBS * 1e9)


def strong_emulated(args, dsnerf_amd, _lib, synth, dev):
    """The strong-scaling partition measured on ONE GPU (no multi-GPU node is available to the builder): the frame is partitioned
    for N = --emulate-world ranks exactly as strong_bench does, and every rank's share is rendered ALONE with the same code on this
    GPU - `--pipeline` frames in flight as the real ranks run them (`ms`), and one frame at a time (`ms_alone`).  Reported: the N
    share times, max / mean (the imbalance a real N-GPU run waits for), the whole frame on one GPU the same way, the index_select
    of N gathered slabs (local, timed here), and the speed-up these predict = T(1 GPU) / (max share + un-deal + all-gather); the
    all-gather is NOT measured - it is priced (gather_ms_priced)."""
    frame = Frame(args, _lib, synth, dev)
    H, W, S, R = frame.H, frame.W, frame.S, frame.R
    Nw = int(args.emulate_world)
    depth = max(1, args.pipeline)
    slots = Slots(_lib, frame, depth)
    rp = dsnerf_amd.RayParallel()
    slots.scenes[0].set_frame(frame.packed, frame.d_xyz, frame.d_poses, 5, False, None, None, None)
    # (the centroid cube: the shares are rendered with one margin, whichever rank calibrates)
    info = frame.packed.calibrate_screen(slots.scenes[0]) if args.screen else {"usable": False, "note": "density screen not opted in (--screen)"}
    screen = info["usable"]

    def time_share(idx, frame_decision):
        sh = Share(_lib, args, frame, slots, idx, screen=screen)
        sh.decide_stop(frame_decision=frame_decision)
        k = 0

        def run(n, d_, share_cus):
            nonlocal k
            for _ in range(n):
                j = k % d_
                k += 1
                with torch.cuda.stream(slots.streams[j]):
                    sh.render(j, share_cus=share_cus)

        run(depth, depth, depth > 1)                  # every slot once
        torch.cuda.synchronize()
        run(args.warmup, depth, depth > 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, depth, depth > 1)
        t_enq = time.perf_counter()                   # the host is done enqueueing: what a frame costs the host thread (Python + launches)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / max(1, args.steps)
        host_ms = 1e3 * (t_enq - t0) / max(1, args.steps)
        alone = []
        for i in range(2 + max(3, min(args.steps, 8))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(slots.streams[0]):
                sh.render(0, share_cus=False)
            torch.cuda.synchronize()
            if i >= 2:
                alone.append(1e3 * (time.perf_counter() - t0))
        cnt = sh.counters(0)
        st = _lib.read_stop_stats(slots.wss[0])
        rec = {"rays": sh.Rl, "ms": ms, "host_enqueue_ms": host_ms, "ms_alone": float(np.mean(alone)), "non_transparent": int(cnt[_lib.CNT_ACTIVE]),
               "skipped_by_termination": st["skipped"] if sh.stop_on else 0, "shaded": int(cnt[_lib.CNT_LIT]) if sh.stop_on else int(cnt[_lib.CNT_POS]),
               "slice_lengths": sh.schedule}
        return rec, sh

    # the whole frame on one GPU, the same way (= the N = 1 line's frame loop); its probe gives the frame's decision and colour scale
    whole, sh_w = time_share(np.arange(R), None)
    decision = (sh_w.stop_on, frame.packed.colour_scale)
    stop_info = sh_w.stop_info
    del sh_w
    worlds = [Nw] if not args.emulate_sweep else sorted({int(x) for x in args.emulate_sweep.split(",")})
    sweeps = {}
    for nw in worlds:
        kind, plan_of, part_info = make_partition(args, _lib, rp, frame, slots, nw, screen)
        shares = []
        for r in range(nw):
            plan = plan_of(dev, rank=r)
            rec, _ = time_share(plan["mine"].cpu().numpy(), decision)
            rec["rank"] = r
            shares.append(rec)
        if kind == "blocks" and args.rebalance > 0:
            # measured re-balancing (what a real run does from its ranks' own frame times): cuts move to where the cumulative measured
            # share time crosses k / N, the cost taken as uniform inside a block; the better of the partitions is kept
            for it in range(args.rebalance):
                b2 = rp.rebalance_bounds(part_info["bounds"], [s_["ms"] for s_ in shares])
                if b2 == part_info["bounds"]:
                    break
                sh2 = []
                for r in range(nw):
                    rec, _ = time_share(np.arange(b2[r], b2[r + 1]), decision)
                    rec["rank"] = r
                    sh2.append(rec)
                if max(s_["ms"] for s_ in sh2) < max(s_["ms"] for s_ in shares):
                    shares, part_info = sh2, dict(part_info, bounds=b2, rebalanced_from_measured_share_times=it + 1)
                else:
                    break
            plan = _block_plan(rp, R, part_info["bounds"], dev, 0)
        else:
            plan = plan_of(dev, rank=0)
        # the un-dealing index_select of N equal slabs into ray order (strong_bench's epilogue behind the all-gather)
        slab = plan["slab"]
        allp = torch.zeros(nw * slab, 6, dtype=torch.float32, device=dev)
        full = torch.empty(R, 6, dtype=torch.float32, device=dev)
        und = []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rp.undeal(allp, plan, out=full)
            torch.cuda.synchronize()
            if i >= 2:
                und.append(1e3 * (time.perf_counter() - t0))
        undeal_ms = float(np.mean(und))
        t = np.array([x["ms"] for x in shares])
        ta = np.array([x["ms_alone"] for x in shares])
        ag_ms = gather_ms_priced(R, nw)
        step_ms = float(t.max()) + undeal_ms + ag_ms
        sweeps[nw] = {"world": nw, "partition": part_info, "shares": shares, "share_ms_max": float(t.max()), "share_ms_mean": float(t.mean()),
                      "host_enqueue_ms_max": float(max(s_["host_enqueue_ms"] for s_ in shares)),
                      "share_ms_min": float(t.min()), "max_over_mean": float(t.max() / t.mean()),
                      "share_ms_alone_max": float(ta.max()), "undeal_ms": undeal_ms, "all_gather_ms_PRICED_not_measured": ag_ms,
                      "sum_of_shares_over_whole_frame": float(t.sum() / whole["ms"]),
                      "predicted_ms_per_frame": step_ms, "predicted_speedup": whole["ms"] / step_ms,
                      "predicted_strong_scaling_efficiency": whole["ms"] / (nw * step_ms),
                      "predicted_speedup_one_frame_at_a_time": whole["ms_alone"] / (float(ta.max()) + undeal_ms + ag_ms)}
    top = sweeps[max(sweeps)]
    res = {"metric": f"strong scaling EMULATED on one GPU: ONE {H}x{W} frame x {S} samples/ray partitioned for {max(sweeps)} ranks ({args.partition})",
           "value": R / (top["predicted_ms_per_frame"] * 1e-3),
           "unit": "rays/s (PREDICTED for the emulated world: max share + un-deal + priced all-gather)",
           "n_gpus": 1, "emulated_world": max(sweeps), "steps": args.steps, "warmup": args.warmup, "ms_per_step": top["predicted_ms_per_frame"],
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
           "dtype": "split-f16x3" + (" + plain-f16 density screen" if screen else ""),
           "config": {"workload": f"one {H}x{W} frame x {S} samples/ray "
                                  + ("(BASELINE configs[1], the metric's frame)" if (H, S) == (512, 64) else "(BASELINE configs[3])")
                                  + f"; each emulated rank's share rendered alone on one MI355X, {depth} frame(s) in flight",
                      "weights": args.weights, "frames_in_flight": depth, "whole_frame_one_gpu": whole, "early_stop": stop_info,
                      "worlds": sweeps, "density_screen_calibration": info}}
    res["config"].update({k: top[k] for k in ("share_ms_max", "share_ms_mean", "max_over_mean", "sum_of_shares_over_whole_frame",
                                              "predicted_speedup", "predicted_strong_scaling_efficiency")})
    _flush_c_stdio()
    print(json.dumps(res), flush=True)
