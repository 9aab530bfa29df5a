"""The headline line of bench.py (one 512 x 512 x 64 frame per GPU, frames in flight) and the weak-scaling emulation."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import (LAZY_LISTS, FLOP_ALL_PER_SAMPLE, FLOP_FIELD_FWD_PER_SAMPLE, FLOP_FIELD_REV_PER_SAMPLE, PEAK_F16_MATRIX_TFLOPS, ROOT,
                     SPLIT_PRODUCTS, load_weights, _flush_c_stdio)


def frame_bench(args, dsnerf_amd, _lib, synth, rk):
    import torch.distributed as dist
    from .baselines import cpu_baseline, cpu_baseline_torch, eager_baseline, host_to_host, host_to_host_pipelined
    from .roofline import roofline
    from .train import TRAIN_DTYPE, train_measure, train_roofline
    world, rank, dev, use_dist = rk.world, rk.rank, rk.dev, rk.on
    H = W = args.hw
    S = args.samples
    R = H * W
    canon, faces = synth.make_body()
    sd = load_weights(synth, args.weights)
    pose_rank = rank if args.per_rank_frames == "different" else 0      # (same: fixed per-GPU work, see --per-rank-frames)
    poses = synth.make_poses(seed=5 + pose_rank)
    xyz = synth.pose_body(canon, seed=3 + pose_rank)
    rays = synth.make_rays(H, W, xyz, fit_box=True)    # every ray crosses the padded body AABB (= mask_at_box rays)

    depth = max(1, args.pipeline)
    scenes = [_lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev) for _ in range(depth)]
    wss = [_lib.RenderWorkspace(dev) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    scene, ws = scenes[0], wss[0]
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    d_xyz = torch.from_numpy(xyz).to(dev)
    d_poses = torch.from_numpy(poses).to(dev)
    ray_o = torch.from_numpy(rays["ray_o"]).to(dev)
    ray_d = torch.from_numpy(rays["ray_d"]).to(dev)
    near0 = torch.from_numpy(rays["near"]).to(dev)
    far0 = torch.from_numpy(rays["far"]).to(dev)
    nears = [near0.clone() for _ in range(depth)]
    fars = [far0.clone() for _ in range(depth)]
    outs = [None] * depth
    gathered = [torch.empty(world * R, 6, dtype=torch.float32, device=dev) if use_dist else None for _ in range(depth)]
    packed_px = [torch.empty(R, 6, dtype=torch.float32, device=dev) for _ in range(depth)]
    for j in range(depth):          # allocate every slot's workspace up front (setup, not a step: W may be smaller than the depth)
        wss[j].get(R, S).zero_()    # ... and touch it: the first GPU access to fresh device memory costs ~12 ms per 3.4 GB (measured: a slot
    torch.cuda.synchronize()        #     first used inside the timed region made 3 frames in flight look 5 % SLOWER than 2 at W = 2)

    def prepare(state_dict, want_screen=None):
        """what Renderer does once per checkpoint (set-up, not a step): pack the parameters; if the density screen is wanted (opt-in:
        --screen / Renderer.density_screen = True) measure its margin for them (PackedParams.calibrate_screen on the frame's points +
        the centroid cube); decide front-to-back slicing from the statistics of one probe frame, which also measures the colour scale
        of the early-stop threshold (dsn_set_early_stop_colour_scale: 2 x the largest colour the probe frame weighed)"""
        want_screen = bool(args.screen or args.force_screen) if want_screen is None else want_screen
        pk = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in state_dict.items()})
        no_screen, screen_info = True, None
        if want_screen and not (args.dense or args.fp32):
            # (like Renderer on the first eval frame of a checkpoint: the geometry phase of the frame, then the margin measured on the
            #  canonical points of ITS non-transparent samples)
            scene.set_frame(pk, d_xyz, d_poses, 5, False, None, None, None)
            _lib.render_rays(scene, pk, ws, ray_o, ray_d, near0.clone(), far0.clone(), S, t_vals, None, None, want_weights=False,
                             phases=_lib.PHASE_GEOMETRY)
            screen_info = pk.calibrate_screen(scene, frame=(ws, R, S))
            no_screen = not (screen_info["usable"] or args.force_screen)
        stop_info = {"enabled": False}
        if not (args.dense or args.fp32) and args.early_stop != "off":
            scene.set_frame(pk, d_xyz, d_poses, 5, False, None, None, None)
            _lib.render_rays(scene, pk, ws, ray_o, ray_d, near0.clone(), far0.clone(), S, t_vals, None, None, want_weights=False,
                             screen=not no_screen, stop_stats=True)
            torch.cuda.synchronize()
            st = _lib.read_stop_stats(ws)
            frac = st["would_skip"] / max(st["active"], 1)
            cmax = st["colour_max"]
            finite = cmax == cmax and cmax != float("inf")
            will_stop = finite and (args.early_stop == "on" or frac >= _lib.EARLY_STOP_MIN_SKIPPED)
            # (like Renderer: the probe frame sizes the relu-record array for these parameters - a dense field gets more than the default
            #  quarter of the samples instead of the overflow pass on every frame; every slot's workspace grows at its next get().  The
            #  probe is one pass: sliced frames put far fewer samples on the sigma > 0 list, estimated by what termination leaves out)
            for w_ in wss:
                w_.fit_records(int(ws.buf[64:68].view(torch.int32)[0]) / float(R * S) * ((1.0 - frac) if will_stop else 1.0),
                               1.6 if will_stop else 1.25)
            scale = pk.set_early_stop_colour_scale(_lib.EARLY_STOP_COLOUR_HEADROOM * cmax) if finite else 1.0
            eps = _lib.early_stop_eps(S, scale)
            schedule = None
            if args.stop_schedule == "auto":
                hist, L_uni = _lib.read_stop_hist(ws, R, S)
                lens, ev, un = _lib.choose_stop_schedule(hist, L_uni, S)
                if len(lens) < hist.shape[1]:
                    schedule = lens
            stop_info = {"enabled": finite and (args.early_stop == "on" or frac >= _lib.EARLY_STOP_MIN_SKIPPED),
                         "slice_lengths": schedule if schedule is not None else f"uniform ({_lib.stop_slice_len(R, S)} samples)",
                         "probe_frame_would_skip_fraction_of_non_transparent": frac, "probe_frame_largest_colour": cmax,
                         "colour_scale": scale, "eps": eps, "bound_abs_for_colours_up_to_the_scale": (S + 1) * eps * scale,
                         "bound": "(S + 1) eps(S, c) x max|colour|: <= 5e-5 absolute while colours stay below the scale c = 2 x the probe "
                                  "frame's largest; the one feature of the frame that is error-bounded, not bit-identical"}
        if stop_info["enabled"] and screen_info is not None and not args.force_screen and screen_info["safe"]:
            # with termination in use the screen's dropped share counts among the samples still evaluated (PackedParams.screen_pays)
            pk.early_stop = {"skipped_fraction": stop_info.get("probe_frame_would_skip_fraction_of_non_transparent", 0.0), "usable": True}
            no_screen = not pk.screen_pays(True)
        torch.cuda.synchronize()
        return {"packed": pk, "no_screen": no_screen, "early": stop_info["enabled"], "screen_info": screen_info, "stop_info": stop_info,
                "schedule": stop_info.get("slice_lengths") if isinstance(stop_info.get("slice_lengths"), list) else None}

    cur = prepare(sd)
    headline_schedule = cur.get("schedule")
    packed, screen_info, stop_info, early = cur["packed"], cur["screen_info"], cur["stop_info"], cur["early"]
    args.no_screen = cur["no_screen"]      # (what the roofline pass below looks at)
    k_step = 0

    pipe = _lib.PhasePipeline(dev) if (args.overlap == "phase" and depth > 1) else None

    # frames in flight: the persistent field kernels take 7/8 of the compute units (DSN_SHARE_CUS, what Renderer.render_views sets);
    # off for the frames timed alone
    share_cus = [depth > 1 and os.environ.get("DSN_BENCH_SHARE_CUS", "1") != "0"]
    audit_every = dsnerf_amd.can_render.SCREEN_AUDIT_EVERY      # what Renderer does by default (screen_audit = "auto")
    audit_of = {}

    def frame_call(j, phases=0):
        # (one frame in `audit_every` carries the density screen's audit, like Renderer's default: 1/128 of the samples the screen
        #  drops take the accurate pass anyway; the frame is bit-identical, the cost is in the measured time)
        outs[j] = _lib.render_rays(scenes[j], cur["packed"], wss[j], ray_o, ray_d, nears[j], fars[j], S, t_vals, None, None,
                                   skip_transparent=not args.dense, want_weights=False, out=outs[j], fp32=args.fp32,
                                   screen=not cur["no_screen"], early_stop=cur["early"], phases=phases,
                                   audit=audit_of.get(j, False), share_cus=share_cus[0], stop_schedule=cur.get("schedule"))

    # what Renderer does behind every sliced frame (round 5): the frame's counter words to page-locked memory, for the hand-over check of
    # its early-stop bound (largest colour weighed <= the threshold's colour scale) - in the timed loop like everything else it does
    guard_host = [torch.empty(256, dtype=torch.uint8).pin_memory() for _ in range(depth)]
    guard_seen = {"frames": 0, "largest_colour": 0.0}

    def guard(j):
        if cur["early"]:
            guard_host[j].copy_(wss[j].buf[:256], non_blocking=True)
            guard_seen["frames"] += 1

    def exchange(j):
        if use_dist:
            packed_px[j][:, 0:3] = outs[j]["color"]
            packed_px[j][:, 3] = outs[j]["disp_map"]
            packed_px[j][:, 4] = outs[j]["acc_map"]
            packed_px[j][:, 5] = outs[j]["depth_map"]
            dist.all_gather_into_tensor(gathered[j], packed_px[j])

    def step():
        # one whole frame in slot j.  overlap = frame: on slot j's stream (consecutive frames on different streams).  overlap =
        # phase: set-up + geometry on the side stream, field kernels on the field stream, shading (+ the exchange) on the side
        # stream one step later - the small kernels run BESIDE the persistent field workgroups instead of between them
        nonlocal k_step
        j = k_step % depth
        audit_of[j] = (k_step % audit_every == 0) and not cur["no_screen"] and not (args.dense or args.fp32)
        k_step += 1

        def geometry():
            nears[j].copy_(near0)
            fars[j].copy_(far0)
            scenes[j].set_frame(cur["packed"], d_xyz, d_poses, 5, False, None, None, None, fine_only=True, lazy=LAZY_LISTS)   # what Renderer does per frame
            if pipe is not None:
                frame_call(j, _lib.PHASE_GEOMETRY)

        if pipe is not None:
            pipe.submit(geometry, lambda: frame_call(j, _lib.PHASE_FIELD), lambda: (frame_call(j, _lib.PHASE_SHADE), guard(j), exchange(j)))
            return
        with torch.cuda.stream(streams[j]):
            geometry()
            frame_call(j)
            guard(j)
            exchange(j)

    def barrier():
        if pipe is not None:
            pipe.flush()                 # (the shading of the last frame: every step's frame is complete inside the timed region)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a step: every slot (scene blob, workspace, output buffers, stream) renders one frame before anything is timed, as slot 0
    # has in prepare() - a slot's first frame carries its one-off costs (first GPU access to its buffers, allocations: 4-12 ms), and
    # with W < depth it would fall into the timed region (measured: 3 frames in flight looked 2-5 % slower than 2 at W = 2 and are
    # 1.5 % faster in steady state).  Then W warm-up steps and exactly K timed steps, as always.
    for _ in range(depth):
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
    # latency of one frame alone (no overlap with a neighbour), for the record
    k_step = 0
    share_cus[0] = False
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        step()
        if pipe is not None:
            pipe.flush()
        torch.cuda.synchronize()
    ms_serial = 1e3 * (time.perf_counter() - t1) / 3
    dt, per_rank_s = rk.times(dt)          # max over ranks + every rank's own time

    n_active = int(ws.buf[:4].view(torch.int32)[0]) if not args.dense else R * S
    n_pos = int(ws.buf[64:68].view(torch.int32)[0]) if (not args.dense and not args.fp32) else n_active
    n_kept = int(ws.buf[128:132].view(torch.int32)[0]) if (not args.dense and not args.fp32 and not cur["no_screen"]) else n_active
    n_fwd, n_rev, n_lit = (n_kept if not cur["no_screen"] else n_active), n_pos, n_pos      # one pass: what each network kernel ran on
    if early:       # sliced frame: word 32 holds the last slice's count only; report what the termination left out instead
        st = _lib.read_stop_stats(ws)
        stop_info["skipped_fraction_of_non_transparent"] = st["skipped"] / max(st["active"], 1)
        stop_info["unshaded_fraction_of_positive_density"] = st["unshaded"] / max(n_pos, 1)
        n_kept = None
        cw_ = ws.buf[:1024].view(torch.int32).cpu()
        K_ = len(cur["schedule"]) if cur.get("schedule") else (S + _lib.stop_slice_len(R, S) - 1) // _lib.stop_slice_len(R, S)
        base_ = 128 if not cur["no_screen"] else 96      # (DSN_CNT_KEEP_K / DSN_CNT_ALIVE_K: what the forward launch of slice k ran on)
        n_fwd = sum(int(cw_[base_ + k]) if (k > 0 or not cur["no_screen"]) else int(cw_[64]) for k in range(K_))
        n_rev, n_lit = int(cw_[_lib.CNT_SEL]), int(cw_[_lib.CNT_LIT])
    ms_step = 1e3 * dt / args.steps
    tflop_executed = (n_fwd * FLOP_FIELD_FWD_PER_SAMPLE + n_rev * FLOP_FIELD_REV_PER_SAMPLE) / 1e12
    value = world * R * args.steps / dt

    result = {
        "metric": f"rendered rays/sec ({S} samples/ray), {H}x{W} frame",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # (N = 1 is the first point of the STRONG curve `--gpus N` measures since round 5 - the same frame partitioned over N ranks;
        #  --weak: one whole frame per rank)
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak" if (args.weak or world > 1) else "strong", "vs_baseline": None,
        "dtype": ("f32 (v_mfma_f32_32x32x2_f32)" if args.fp32 else
                  "split-f16x3 (3 x v_mfma_f32_32x32x16_f16 on hi/lo fp16 operand halves, f32 accumulate: f32-equivalent accuracy - parity "
                  "bars of the tests: 1e-4 absolute on RGB / acc / weights against the reference; sigma 1e-4 absolute where |sigma| <= 100 "
                  "and 4e-6 RELATIVE to max|sigma| above (this checkpoint: |sigma| up to 1013, one float32 ulp there is 6e-5))"
                  + ("" if (args.dense or cur["no_screen"]) else " + plain-f16 density screen")),
        "data": "synthetic",
        "config": {
            "workload": f"{H}x{W} frame x {S} samples/ray per GPU (BASELINE configs[1]; N>1: one frame per GPU, configs[4]), "
                        f"synthetic closed body V=6890/F=13776, camera framed so that all rays cross the body AABB (mask_at_box), "
                        f"GG sampling, eval mode, parameters: {args.weights}"
                        + (" (converged on this body by scripts/train_w4.py: a test-split render of a trained model)" if args.weights == "w4" else ""),
            "rays_per_gpu": R, "samples_per_ray": S,
            "transparent_skip": (not args.dense),
            # which share of the R x S samples each stage ran on ("evaluated_sample_fraction" used to be the first of these)
            "non_transparent_sample_fraction": n_active / float(R * S),
            "forward_sample_fraction": n_fwd / float(R * S),          # k_field16<forward>: non-transparent, ray still alive, not screened out
            "reverse_sample_fraction": n_rev / float(R * S),          # k_field16<reverse>: sigma > 0 and weight above the threshold
            "shaded_sample_fraction": n_lit / float(R * S),           # normals + lighting
            "positive_density_sample_fraction": n_pos / float(R * S),
            "density_screen": not (args.dense or args.fp32 or cur["no_screen"]),
            "density_screen_calibration": screen_info, "density_screen_audit_every_n_frames": audit_every, "weights": args.weights,
            "accurate_pass_sample_fraction": None if n_kept is None else n_kept / float(R * S),
            "early_stop": stop_info,
            "ms_per_frame": ms_step,
            "frames_in_flight": depth, "overlap": ("none" if depth == 1 else args.overlap), "ms_per_frame_alone": ms_serial,
            "persistent_kernels_share_cus": bool(depth > 1 and os.environ.get("DSN_BENCH_SHARE_CUS", "1") != "0"),
            "setup_frames_per_slot": 1,      # (untimed, before the W warm-up steps: a slot's first frame carries its one-off costs)
            # every ray is fully rendered, so the dense-equivalent rate is `value`; this is the dense
            # algorithmic work of the frame (2 x 902 272 MAC x R x S) over the frame time
            # - work NOT done (transparent / terminated samples) divided by time: never a throughput, it exceeds the chip's peak by design
            "dense_equivalent_work_rate_NOT_throughput_tflops": FLOP_ALL_PER_SAMPLE * R * S / (ms_step * 1e-3) / 1e12,
            "exchange": "all_gather_into_tensor [R,6] fp32 per rank (RCCL)" if use_dist else "none",
            "per_rank_frames": ("every rank renders the same synthetic frame as the N = 1 line (fixed per-GPU work)" if args.per_rank_frames == "same"
                                else "rank r renders its own pose (seeds 3 + r, 5 + r): the step waits for the slowest frame"),
        },
        "ranks": rk.info(per_rank_s, args.steps),
        "early_stop": stop_info,
        # the whole frame against the matrix roofline: field FLOPs actually executed (forward + reverse samples above) over the frame time
        "whole_frame": {"tflop_executed": tflop_executed, "achieved": tflop_executed / (ms_step * 1e-3),
                        "peak": PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "unit": "TFLOP/s",
                        "frac": tflop_executed / (ms_step * 1e-3) / (PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS),
                        "forward_samples": n_fwd, "reverse_samples": n_rev, "shaded_samples": n_lit},
    }

    if rank == 0 and world == 1 and not args.no_extras and not (args.dense or args.fp32):
        # The frame time is a property of the CHECKPOINT as much as of the kernels: the same frame, same pipeline,
        # 5 timed frames each, for every parameter set the repo pins with reference-generated goldens.  w4 is the converged one.
        headline = cur
        # (the headline's own counters, before the other parameter sets reuse the workspaces)
        headline_cw = wss[(k_step - 1) % depth].buf[:256].view(torch.int32).cpu()
        headline_st = _lib.read_stop_stats(wss[(k_step - 1) % depth])
        headline_ws_gb = wss[0].bytes_for(R, S) / 1e9
        by = {}
        share_cus[0] = depth > 1 and os.environ.get("DSN_BENCH_SHARE_CUS", "1") != "0"      # (frames in flight again)

        def timed_frames(c):
            nonlocal cur, k_step
            cur = c
            k_step = 0
            for _ in range(max(2, depth)):      # (every slot once: a workspace that has just grown pays its first touch here, untimed)
                step()
            barrier()
            tb = time.perf_counter()
            for _ in range(5):
                step()
            barrier()
            return 1e3 * (time.perf_counter() - tb) / 5

        for name in ("default", "w2", "w3", "w4"):
            if name in ("w2", "w4") and not os.path.exists(os.path.join(ROOT, "tests", "golden", f"weights_{name}.npz")):
                continue
            if name == args.weights:
                by[name] = {"ms_per_frame": ms_step, "frames": args.steps}
                c = headline
            else:
                c = prepare(load_weights(synth, name), want_screen=False)
                by[name] = {"ms_per_frame": timed_frames(c), "frames": 5}
            if name == args.weights:
                cw, st = headline_cw, headline_st
            else:
                cw = wss[(k_step - 1) % depth].buf[:256].view(torch.int32).cpu()
                st = _lib.read_stop_stats(wss[(k_step - 1) % depth])
            by[name].update({
                "rays_per_s": R / (by[name]["ms_per_frame"] * 1e-3),
                "density_screen": not c["no_screen"],
                "early_stop": bool(c["early"]),
                "early_stop_would_skip_fraction": c["stop_info"].get("probe_frame_would_skip_fraction_of_non_transparent"),
                "early_stop_skipped_fraction": (st["skipped"] / max(st["active"], 1)) if c["early"] else 0.0,
                "early_stop_colour_scale": c["stop_info"].get("colour_scale"), "early_stop_eps": c["stop_info"].get("eps"),
                "non_transparent_fraction": int(cw[_lib.CNT_ACTIVE]) / float(R * S),
                "positive_density_fraction": int(cw[_lib.CNT_POS]) / float(R * S)})
            # the same frames with the density screen opted in (both pipelined numbers in one line): it runs only
            # where its calibration for the parameters says it is safe and pays
            cs = prepare(load_weights(synth, name), want_screen=True)
            si = cs["screen_info"] or {}
            by[name]["with_density_screen"] = {
                "runs": not cs["no_screen"], "calibration_safe": si.get("safe"), "margin": si.get("margin"),
                "dropped_fraction_at_calibration": si.get("dropped_fraction"),
                "ms_per_frame": timed_frames(cs) if not cs["no_screen"] else None}
        cur = headline
        k_step = 0
        result["config"]["by_weights"] = by
        # (top level: the frame time is a property of the checkpoint - one number per parameter set, same frame, same pipeline)
        result["ms_per_frame_by_weights"] = {k: v["ms_per_frame"] for k, v in by.items()}
        share_cus[0] = False
        result["config"]["by_weights_note"] = ("same frame and pipeline for every parameter set, Renderer's defaults (density screen off, early "
                                               "stop decided by the probe frame); default = hash-random init (thin fog), w2 = 400 reference-"
                                               "trainer steps (solid, unsaturated), w3 = hash init x3.5 (dense guess), w4 = converged with "
                                               "scripts/train_w4.py: the headline.  with_density_screen: the same frames with the opt-in "
                                               "plain-fp16 screen (statistically safe: calibrated margin + audit), where its calibration lets it run")
    if rank == 0 and world == 1 and not args.no_roofline:
        scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)      # (the frame state of THESE parameters: by_weights has used the scene)
        result["roofline"] = roofline(_lib, scene, packed, ray_o, ray_d, near0, far0, S, t_vals, args, early=early, schedule=headline_schedule)
    if rank == 0 and world == 1 and "roofline" in result:
        # kernel AND frame in the one block the driver parses: the dominant kernel's fraction is `frac`; the whole
        # frame's - field FLOPs executed per frame over the frame time - and the matrix pipe's busy share from the committed counters
        from .common import measured_mfma_busy
        result["roofline"]["whole_frame_frac"] = result["whole_frame"]["frac"]
        busy, busy_src = measured_mfma_busy("k_field16<forward>", args)
        result["roofline"]["mfma_busy"] = busy
        result["roofline"]["mfma_busy_source"] = None if busy is None else (
            f"{busy_src}: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs), committed rocprofv3 --pmc passes of this "
            f"command; not collected in this run")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # north_star names "the reference's CPU PyTorch path": the torch restatement is the closer stand-in and therefore THE
        # cpu_baseline of the line; the OpenMP C port of the same algorithm - the faster CPU implementation - beside it
        result["cpu_baseline"] = cpu_baseline_torch(synth, canon, faces, xyz, poses, sd, rays, S, args)
        result["cpu_baseline_c"] = cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, S, args)
    if rank == 0 and world == 1 and not args.no_extras and not (args.dense or args.fp32):
        # what else ran on this box, in the same line: the same frame without the screen and with the exact-fp32
        # kernel, the host-batch -> host-image path of the reference's render_view, and the eager-torch restatement on this GPU
        def frame_ms(reps, **kw):
            n_, f_ = near0.clone(), far0.clone()
            ms = []
            for i in range(reps + 1):
                n_.copy_(near0); f_.copy_(far0)
                torch.cuda.synchronize()
                t = time.perf_counter()
                scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True, lazy=LAZY_LISTS)
                _lib.render_rays(scene, packed, ws, ray_o, ray_d, n_, f_, S, t_vals, None, None, want_weights=False, out=outs[0], **kw)
                torch.cuda.synchronize()
                if i:
                    ms.append(1e3 * (time.perf_counter() - t))
            return float(np.mean(ms))
        ex = result["config"]
        ex["ms_per_frame_alone_one_pass"] = frame_ms(5)          # no slices, no termination (and no screen): every non-transparent sample, one launch per kernel
        ex["ms_per_frame_alone_fp32_exact"] = frame_ms(2, fp32=True)
        # host batch -> host images (see host_to_host: the second key is the same frame when the CALLER runs a small torch CPU op on
        # the main thread right before it - torch's intra-op pool, `host_threads` OpenMP threads here, then spins beside the GPU feeder)
        ex["host_to_host_ms"] = host_to_host(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S)
        ex["host_to_host_ms_after_a_caller_torch_cpu_op"] = host_to_host(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S,
                                                                         caller_torch_op=True)
        # ... and the loop the callers run: a sequence of host batches through render_views, three frames in flight
        ex["host_to_host_ms_pipelined"] = host_to_host_pipelined(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S)
        ex["host_threads"] = torch.get_num_threads()
        ex["host_cpu_quota_cores"] = _lib.cpu_quota_cores()
        # the per-sample workspace is the caller's to size: the same render_view in four ray chunks (the reference's own loop runs
        # 3072-ray chunks, can_render.py:172-245) needs a quarter of it, for this much time
        chunk = (H * W) // 4
        ex["chunked_frame"] = {"chunk_rays": chunk,
                               "workspace_gb_whole_frame_headline": headline_ws_gb,      # (at the record fraction the headline's probe frame left)
                               "record_capacity_fraction_now": wss[0].want_fraction,      # (of the bench's own workspaces: grown by the densest set of by_weights)
                               "workspace_gb_whole_frame": wss[0].bytes_for(H * W, S) / 1e9,
                               "workspace_gb_chunked": wss[0].bytes_for(chunk, S) / 1e9,
                               "host_to_host_ms": host_to_host(args, dsnerf_amd, synth, dev, canon, faces, sd, xyz, poses, rays, H, W, S,
                                                               chunk=chunk)}
        # BASELINE configs[2] in the same line: 8192 x 64 training step (render + MSE + backward + Adam)
        t_dt, t_loss, t_ovf, t_rows = train_measure(args, dsnerf_amd, synth, dev, 1, 0, False, 20, 5, weights="default")
        if os.path.exists(os.path.join(ROOT, "tests", "golden", "weights_w4.npz")):
            # the same step from the CONVERGED parameters (late in training the field is bimodal: most rows have alpha = 0 exactly and
            # drop out of the backward; from the hash-random start nearly every evaluated row carries a gradient)
            w_dt, w_loss, w_ovf, w_rows = train_measure(args, dsnerf_amd, synth, dev, 1, 0, False, 20, 5, weights="w4")
            result["train_w4"] = {"train_ms_per_step": 1e3 * w_dt / 20, "value": args.train_rays * 20 / w_dt, "unit": "rays/s",
                                  "rows_last_step": w_rows, "final_loss": w_loss, "range_overflow_samples_last_step": w_ovf,
                                  "roofline": train_roofline(1e3 * w_dt / 20, args.train_rays, S, w_rows, "w4")}
        result["train"] = {"metric": "training rays/sec (8192 rays x 64 samples: forward + backward + Adam step, BASELINE configs[2])",
                           "value": args.train_rays * 20 / t_dt, "unit": "rays/s", "train_ms_per_step": 1e3 * t_dt / 20, "steps": 20,
                           "warmup": 5, "dtype": TRAIN_DTYPE, "final_loss": t_loss, "range_overflow_samples_last_step": t_ovf, "rows_last_step": t_rows,
                           "roofline": train_roofline(1e3 * t_dt / 20, args.train_rays, S, t_rows, "default")}
        if not args.no_cpu_baseline:
            result["eager_gpu_baseline"] = eager_baseline(args, _lib, synth, dev, chunks=3, train=False)
            result["eager_gpu_baseline"]["x_faster_per_frame"] = result["eager_gpu_baseline"]["eval_ms_per_512x512_frame"] * \
                (H * W / (512.0 * 512.0)) / ms_serial
    rk.finish()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(result), flush=True)      # the LAST line of stdout (RCCL prints its banner at its first collective)



def stop_setup(_lib, args, scene, packed, ws, o, d, near0, far0, S, t_vals, screen, reduce_max=None, more_ws=()):
    """Front-to-back slicing for the secondary modes, decided like Renderer / the headline loop do: one probe render (one pass,
    DSN_STOP_STATS) of these rays says what termination would leave out, how large the colours are (-> the threshold's colour scale)
    and how the slices should be cut (choose_stop_schedule).  reduce_max(tensor): all-reduce MAX over the ranks of a multi-GPU run, so
    that every rank takes the same decision and threshold.  more_ws: the other workspaces that will render these rays (frames in flight):
    they get the record capacity the probe asks for as well.  Returns (enabled, schedule | None, info dict)."""
    if args.early_stop == "off" or args.dense or args.fp32:
        return False, None, {"enabled": False}
    R = o.shape[0]
    _lib.render_rays(scene, packed, ws, o, d, near0.clone(), far0.clone(), S, t_vals, None, None, want_weights=False, screen=screen,
                     stop_stats=True)
    torch.cuda.synchronize()
    st = _lib.read_stop_stats(ws)
    frac = st["would_skip"] / max(st["active"], 1)
    cmax = st["colour_max"]
    finite = cmax == cmax and cmax != float("inf")
    if reduce_max is not None:
        t_ = torch.tensor([frac, cmax if finite else float("inf")], dtype=torch.float64, device=o.device)
        reduce_max(t_)
        frac, cmax = float(t_[0]), float(t_[1])
        finite = cmax != float("inf")
    enabled = finite and (args.early_stop == "on" or frac >= _lib.EARLY_STOP_MIN_SKIPPED)
    # (relu records: the probe frame is one pass; sliced frames put far fewer samples on the sigma > 0 list - estimated here, and a
    #  frame that still overflows takes the exact overflow pass)
    for w_ in [ws] + list(more_ws):
        w_.fit_records(int(ws.buf[64:68].view(torch.int32)[0]) / float(R * S) * ((1.0 - frac) if enabled else 1.0), 1.6 if enabled else 1.25)
    scale = packed.set_early_stop_colour_scale(_lib.EARLY_STOP_COLOUR_HEADROOM * cmax) if finite else 1.0
    schedule = None
    if args.stop_schedule == "auto":      # (every rank cuts its own rays' slices from its own histogram: no collective needed)
        hist, L_uni = _lib.read_stop_hist(ws, R, S)
        lens, _, _ = _lib.choose_stop_schedule(hist, L_uni, S)
        if len(lens) < hist.shape[1]:
            schedule = lens
    return enabled, (schedule if enabled else None), {
        "enabled": enabled, "probe_would_skip_fraction_of_non_transparent": frac, "probe_largest_colour": cmax, "colour_scale": scale,
        "eps": _lib.early_stop_eps(S, scale), "slice_lengths": schedule if schedule is not None else f"uniform ({_lib.stop_slice_len(R, S)} samples)"}


def weak_emulated(args, dsnerf_amd, _lib, synth, dev):
    """The weak-scaling line's per-rank work with --per-rank-frames different, measured on ONE GPU: every rank renders its own frame
    of the multi-frame batch (pose / posed-mesh seeds 3 + rank, 5 + rank) and one all-gather of [R,6] pixels follows.  Each emulated
    rank's frame is rendered alone here (frames in flight as the ranks do); the spread of the N times is the load imbalance such a
    run waits for every step - a property of the poses, which is why the default weak line gives every rank the SAME frame (rank 0's
    here): `predicted_weak_scaling_efficiency_same_frames` prices that case (only the all-gather is added to rank 0's time)."""
    H = W = args.hw
    S = args.samples
    R = H * W
    Nw = int(args.emulate_world)
    canon, faces = synth.make_body()
    sd = load_weights(synth, args.weights)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    depth = max(1, args.pipeline)
    scenes = [_lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev) for _ in range(depth)]
    wss = [_lib.RenderWorkspace(dev) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    info, screen, ranks = None, True, []
    for r_ in range(Nw):
        xyz = synth.pose_body(canon, seed=3 + r_)
        rays = synth.make_rays(H, W, xyz, fit_box=True)
        d_xyz, d_poses = T(xyz), T(synth.make_poses(seed=5 + r_))
        o, d, near0, far0 = T(rays["ray_o"]), T(rays["ray_d"]), T(rays["near"]), T(rays["far"])
        if info is None:
            scenes[0].set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)
            _lib.render_rays(scenes[0], packed, wss[0], o, d, near0.clone(), far0.clone(), S, t_vals, want_weights=False,
                             phases=_lib.PHASE_GEOMETRY)
            info = packed.calibrate_screen(scenes[0], frame=(wss[0], R, S)) if args.screen else {"usable": False, "note": "density screen not opted in"}
            screen = bool(info["usable"])
        nears, fars, outs = [near0.clone() for _ in range(depth)], [far0.clone() for _ in range(depth)], [None] * depth
        # (every rank of the real run probes its own frame: early stop, colour scale and slice schedule per emulated rank)
        scenes[0].set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)
        stop_on, schedule, stop_info = stop_setup(_lib, args, scenes[0], packed, wss[0], o, d, near0, far0, S, t_vals, screen)

        def step(k):
            j = k % depth
            with torch.cuda.stream(streams[j]):
                nears[j].copy_(near0)
                fars[j].copy_(far0)
                scenes[j].set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True, lazy=LAZY_LISTS)
                outs[j] = _lib.render_rays(scenes[j], packed, wss[j], o, d, nears[j], fars[j], S, t_vals, want_weights=False, out=outs[j],
                                           screen=screen, early_stop=stop_on, stop_schedule=schedule, share_cus=depth > 1)

        for k in range(args.warmup):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        cnt = wss[(args.steps - 1) % depth].buf[:256].view(torch.int32).cpu()
        ranks.append({"rank": r_, "ms_per_frame": ms, "non_transparent": int(cnt[_lib.CNT_ACTIVE]), "accurate_pass": int(cnt[_lib.CNT_KEEP]),
                      "positive_density": int(cnt[_lib.CNT_POS]), "early_stop": stop_info})
    t = np.array([x["ms_per_frame"] for x in ranks])
    ag_ms = 0.03 + 1e3 * (24.0 * R * (Nw - 1)) / ((Nw - 1) * 153e9)      # every rank receives N - 1 slabs of 24 B x R over its N - 1 links
    res = {"metric": f"weak-scaling load balance: the {Nw} ranks' frames ({H}x{W} x {S} samples/ray) rendered one after the other on ONE GPU",
           "value": Nw * R / ((float(t.max()) + ag_ms) * 1e-3), "unit": "rays/s (PREDICTED for the emulated world: max frame + priced all-gather)",
           "n_gpus": 1, "emulated_world": Nw, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.max()) + ag_ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
           "dtype": "split-f16x3" + (" + plain-f16 density screen" if screen else ""),
           "config": {"workload": "one frame per emulated rank (BASELINE configs[4] / the weak line of bench.py --gpus N)", "weights": args.weights,
                      "ranks": ranks, "frame_ms_max": float(t.max()), "frame_ms_mean": float(t.mean()), "frame_ms_min": float(t.min()),
                      "max_over_mean": float(t.max() / t.mean()), "all_gather_ms_PRICED_not_measured": ag_ms,
                      "predicted_weak_scaling_efficiency": float(t.mean() / (t.max() + ag_ms)),
                      "predicted_weak_scaling_efficiency_same_frames": float(t[0] / (t[0] + ag_ms)),
                      "density_screen_calibration": info}}
    _flush_c_stdio()
    print(json.dumps(res), flush=True)

