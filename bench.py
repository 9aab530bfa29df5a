#!/usr/bin/env python3
"""bench.py - rendered rays/s of the Dual-Space-NeRF hot path on MI355X.

A "step" = one pass of the whole hot path (per-frame setup, geometry-guided sampling, nearest-face warp,
canonical field + d sigma/dx, normals + lighting MLP, compositing) over one synthetic 512x512 frame at
64 samples/ray (BASELINE.json configs[1]) per GPU, inputs resident in HBM.  With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank renders one frame of the multi-frame batch
(configs[4], rays partitioned across GPUs in contiguous blocks = frames) and the rendered pixels are
exchanged with one RCCL all-gather inside the timed region: weak scaling, value = all rays / max time.
The frames of the batch are copies of the N = 1 line's frame by default (fixed per-GPU work as N grows);
--per-rank-frames different gives every rank its own pose (the step then waits for the slowest frame).

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
FLOP_SCREEN_PER_SAMPLE = 2.0 * 425728.0      # k_screen16: trunk + density head, one fp16 product per algorithmic product
FLOP_ALL_PER_SAMPLE = 2.0 * 902272.0         # + lighting MLP 17 664 MAC
PEAK_F32_MATRIX_TFLOPS = 157.3               # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MATRIX_TFLOPS = 2500.0              # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
SPLIT_PRODUCTS = 3                           # split-fp16: 3 f16 MFMA products per algorithmic product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU).  Launched by torch.distributed.run (WORLD_SIZE in the environment) it must equal the "
                         "world size; WITHOUT a launcher and N > 1 bench.py re-executes itself under `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (weak, --strong and --train alike), so `python bench.py "
                         "--gpus 8` measures 8 GPUs, never one")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise ONLY the launcher and the collectives of the selected mode (weak / --strong / --train) on the CPU with "
                         "the gloo backend and a stand-in for the render (no GPU, no HIP library): proves that --gpus N starts N ranks which "
                         "join the same collectives and that rank 0 prints one JSON line with n_gpus = N.  Not a measurement")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hw", type=int, default=512, help="image side (BASELINE configs[1]: 512)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--dense", action="store_true", help="evaluate the networks on every sample (no transparent skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=0, help="rays the CPU oracle is timed on (0 = sized for ~15 s)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="exact-fp32 MFMA field kernel instead of split-fp16")
    ap.add_argument("--screen", action="store_true",
                    help="opt into the plain-fp16 density screen (Renderer.density_screen = True; off by default since round 4: its margin is "
                         "calibrated and audited, i.e. statistically safe, not proven exact).  It then runs when its calibration for the "
                         "parameters says it is safe and pays")
    ap.add_argument("--no-screen", action="store_true", help="(accepted for old command lines: the screen is off unless --screen is given)")
    ap.add_argument("--force-screen", action="store_true", help="(kernel experiments) keep the density screen on whatever its calibration says")
    ap.add_argument("--early-stop", default="auto", choices=["auto", "on", "off"],
                    help="front-to-back slices with ray termination (DSN_EARLY_STOP): auto = like Renderer, from the statistics of one "
                         "probe frame at set-up (used when it would leave out >= 4 %% of the non-transparent samples)")
    ap.add_argument("--stop-schedule", default="auto", choices=["auto", "uniform"],
                    help="slices of the front-to-back evaluation: auto = lengths chosen from the probe frame's statistics (longer slices where "
                         "few rays end: fewer launches for a few more samples, same error bound; Renderer.stop_schedule), uniform = 4 / 8 samples")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="frames in flight (own scene / workspace each); 1 = strictly serial.  How they overlap: --overlap.  Measured "
                         "(profiles/r03_frames_in_flight.txt): 3 against 2 is -1.5 %% on the default frame, -2.6 %% on the converged set, 4 is no better")
    ap.add_argument("--overlap", default="frame", choices=["phase", "frame"],
                    help="frame (default): one HIP stream per frame in flight; phase: one stream for the field kernels and one for "
                         "everything else (_lib.PhasePipeline: geometry of frame k+1 and shading of frame k-1 BESIDE the field kernels of "
                         "frame k).  Measured equal within 1.5 %% (profiles/r03c_*): the GPU is busy 99.7 %% of the frame either way - the "
                         "small kernels are work the chip has to do, not latency to hide; beside a persistent field workgroup they run "
                         "5-10x longer and slow it by 8 %%")
    ap.add_argument("--train", action="store_true",
                    help="secondary mode (not the headline metric): BASELINE configs[2] training step, 8192 rays x 64 "
                         "samples, render + MSE loss + backward + Adam step through the Renderer mirror")
    ap.add_argument("--train-rays", type=int, default=8192)
    ap.add_argument("--weights", default="w4", choices=["default", "w2", "w3", "w4"],
                    help="parameter set (default since round 4: w4, the CONVERGED checkpoint - BASELINE configs[1] is a test-split render of "
                         "a trained model): the hash-generated `default`, w2 = trained by the real reference (tests/golden/weights_w2.npz, "
                         "dense near the surface: the density screen calibrates itself off), w3 = large-magnitude hash set, w4 = "
                         "CONVERGED on the synthetic body by this repo's HIP trainer (scripts/train_w4.py, tests/golden/weights_w4.npz)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="on ONE GPU: with --strong render each of the N ranks' round-robin tile shares of the frame alone, one after "
                         "the other, and report the N times, max / mean (the load balance of the tile deal) and the strong-scaling "
                         "efficiency they predict; without --strong the N ranks' own frames of the weak line")
    ap.add_argument("--strong", action="store_true",
                    help="strong-scaling mode (BASELINE configs[3]): ONE 1024 x 1024 x 128 frame, its rays dealt to the ranks in "
                         "round-robin 3072-ray tiles (RayParallel.tile_indices), rendered pixels all-gathered inside the timed "
                         "region; N = 1 renders the whole frame on one GPU")
    ap.add_argument("--per-rank-frames", default="same", choices=["same", "different"],
                    help="weak mode / --train with N > 1: same (default) = every rank renders the SAME synthetic frame / batch as the N = 1 line "
                         "(pose seeds 3, 5): per-GPU work is fixed as N grows, which is what makes the line a weak-scaling measurement; "
                         "different = rank r renders its own pose (seeds 3 + r, 5 + r: frames of a sequence dealt to the GPUs, BASELINE "
                         "configs[4]) - the frames then differ by up to 13 %% in non-transparent samples and every step waits for the slowest "
                         "(profiles/r04_weak_emulated8.json: max / mean 1.06), a property of the data, not of the scaling")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements of the default N = 1 line (screen off, exact fp32, host-to-host "
                         "render_view, eager-torch GPU baseline, torch CPU baseline)")
    ap.add_argument("--eager-baseline", action="store_true",
                    help="secondary mode: time the eager-PyTorch restatement of the path (oracle/train_oracle.py) on this GPU "
                         "the way the reference runs it (3072-ray chunks; 8192-ray training step), nearest-face searches "
                         "excluded - SURVEY 8d baseline (i)")
    return ap.parse_args()


def load_weights(synth, name):
    if name in ("w2", "w4"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{name}.npz"))
        return {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    if name == "w3":
        return synth.make_state_dict(seed=7, gain=3.5)
    return synth.make_state_dict()


def _flush_c_stdio():
    """RCCL prints its banner through C stdio; push it (and ours) out so that the JSON line really is the last line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


class Ranks:
    """The process group of a bench run: one rank per GPU over RCCL (backend "nccl" on ROCm), or gloo on the CPU for --dry-launch.
    Everything the three modes need from it: barrier, the max over ranks of the timed region, every rank's own time, and what the
    JSON line reports about the group (`ranks`: did the collective library really see N ranks?)."""

    def __init__(self, args):
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = bool(args.dry_launch)
        if args.gpus is not None and args.gpus != self.world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {self.world} ranks")
        # DSN_BENCH_FORCE_DIST=1 (debug): take the RCCL path (process group, per-frame all-gather, barriers) with ONE rank too, so the
        # multi-GPU code can be exercised on a 1-GPU box
        self.on = self.world > 1 or os.environ.get("DSN_BENCH_FORCE_DIST") == "1"
        self.backend = None
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback exists for the product path)"
            # DSN_BENCH_ONE_GPU=1 (debug, with DSN_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device): every rank uses GPU 0, so
            # that the REAL code paths of a multi-rank run - partition, per-step collectives, barriers, the max over ranks - can be run
            # end to end on a one-GPU box.  A control-flow check: the ranks share the GPU, the times mean nothing.
            self.one_gpu = os.environ.get("DSN_BENCH_ONE_GPU") == "1"
            idx = 0 if self.one_gpu else self.local
            assert idx < torch.cuda.device_count(), (f"rank {self.rank}: local rank {self.local} has no GPU "
                                                     f"({torch.cuda.device_count()} visible)")
            self.dev = torch.device("cuda", idx)
            torch.cuda.set_device(self.dev)
        if self.on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            self.backend = "gloo" if self.dry else os.environ.get("DSN_BENCH_BACKEND", "nccl")
            if self.backend == "gloo":
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", device_id=self.dev, rank=self.rank, world_size=self.world)
            assert dist.get_world_size() == self.world and (args.gpus is None or dist.get_world_size() == args.gpus)

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.on:
            self.dist.barrier()
        self.sync()

    def times(self, dt):
        """(max over ranks, [every rank's own seconds]) of a timed region - one all-gather of one double per rank"""
        if not self.on:
            return dt, [dt]
        mine = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        every = torch.empty(self.world, dtype=torch.float64, device=self.dev)
        self.dist.all_gather_into_tensor(every, mine)
        every = [float(x) for x in every.cpu()]
        return max(every), every

    def info(self, per_rank_s=None, steps=1):
        """what the JSON line says about the group: the world the COLLECTIVE LIBRARY reports (not the flag), counted once more with an
        all-reduce of ones, the backend and its version, and every rank's own time per step"""
        seen = 1
        if self.on:
            one = torch.ones(1, dtype=torch.int32, device=self.dev)
            self.dist.all_reduce(one)
            seen = int(one.item())
        ver = None
        if self.backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
        return {"world_size": self.dist.get_world_size() if self.on else 1, "ranks_counted_by_all_reduce": seen,
                "backend": ({"nccl": "nccl (= RCCL on ROCm)", "gloo": "gloo (dry launch, CPU)" if self.dry else
                             "gloo over GPU tensors (DEBUG: control-flow check of the multi-rank paths, not a measurement)"}.get(self.backend)),
                "ranks_share_one_gpu_DEBUG": bool(getattr(self, "one_gpu", False)),
                "rccl_version": ver, "launcher": os.environ.get("DSN_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                                                                else ("none (single process)" if self.world == 1 else "external")),
                "per_rank_ms_per_step": None if per_rank_s is None else [1e3 * t / steps for t in per_rank_s]}

    def finish(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this same command under torch.distributed.run (what the
    driver's own N > 1 invocation does) and hand its exit status on.  The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    if not args.dry_launch and os.environ.get("DSN_BENCH_ONE_GPU") != "1":
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {n_dev} GPU(s) visible on this node")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, DSN_BENCH_LAUNCHER="bench.py --gpus N -> torch.distributed.run", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4" if args.dry_launch else str(max(1, (os.cpu_count() or 8) // max(1, args.gpus)))))
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


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
            tile = 3072
            out = rp.render_tiled(lambda o, d, n, f: {"color": px_of(o[:, 0], 0)[:, 0:3], "disp_map": px_of(o[:, 0], 0)[:, 3],
                                                      "acc_map": px_of(o[:, 0], 0)[:, 4], "depth_map": px_of(o[:, 0], 0)[:, 5]},
                                  torch.arange(R)[:, None].float().expand(R, 3), torch.zeros(R, 3), torch.zeros(R), torch.zeros(R), tile=tile)
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


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        launch_ranks(args)          # (does not return)
    rk = Ranks(args)
    world, rank, dev, use_dist = rk.world, rk.rank, rk.dev, rk.on
    import torch.distributed as dist
    if args.dry_launch:
        return dry_launch(args, rk)

    import dsnerf_amd
    from dsnerf_amd import _lib, synth

    if args.train:
        return train_bench(args, dsnerf_amd, synth, dev, world, rank, use_dist, rk)
    if args.eager_baseline:
        print(json.dumps(eager_baseline(args, _lib, synth, dev)))
        return
    if args.strong:
        return strong_bench(args, dsnerf_amd, _lib, synth, dev, world, rank, use_dist, rk)
    if args.emulate_world > 1 and world == 1:
        return weak_emulated(args, dsnerf_amd, _lib, synth, dev)
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
            _lib.fit_record_capacity(int(ws.buf[64:68].view(torch.int32)[0]) / float(R * S) * ((1.0 - frac) if will_stop else 1.0),
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
            scenes[j].set_frame(cur["packed"], d_xyz, d_poses, 5, False, None, None, None, fine_only=True)   # what Renderer does per frame
            if pipe is not None:
                frame_call(j, _lib.PHASE_GEOMETRY)

        if pipe is not None:
            pipe.submit(geometry, lambda: frame_call(j, _lib.PHASE_FIELD), lambda: (frame_call(j, _lib.PHASE_SHADE), exchange(j)))
            return
        with torch.cuda.stream(streams[j]):
            geometry()
            frame_call(j)
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
    if early:       # sliced frame: word 32 holds the last slice's count only; report what the termination left out instead
        st = _lib.read_stop_stats(ws)
        stop_info["skipped_fraction_of_non_transparent"] = st["skipped"] / max(st["active"], 1)
        stop_info["unshaded_fraction_of_positive_density"] = st["unshaded"] / max(n_pos, 1)
        n_kept = None
    ms_step = 1e3 * dt / args.steps
    value = world * R * args.steps / dt

    result = {
        "metric": f"rendered rays/sec ({S} samples/ray), {H}x{W} frame",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (v_mfma_f32_32x32x2_f32)" if args.fp32 else
                  "split-f16x3 (3 x v_mfma_f32_32x32x16_f16 on hi/lo fp16 operand halves, f32 accumulate: f32-equivalent accuracy)"
                  + ("" if (args.dense or cur["no_screen"]) else " + plain-f16 density screen")),
        "data": "synthetic",
        "config": {
            "workload": f"{H}x{W} frame x {S} samples/ray per GPU (BASELINE configs[1]; N>1: one frame per GPU, configs[4]), "
                        f"synthetic closed body V=6890/F=13776, camera framed so that all rays cross the body AABB (mask_at_box), "
                        f"GG sampling, eval mode, parameters: {args.weights}"
                        + (" (converged on this body by scripts/train_w4.py: a test-split render of a trained model)" if args.weights == "w4" else ""),
            "rays_per_gpu": R, "samples_per_ray": S,
            "transparent_skip": (not args.dense),
            "evaluated_sample_fraction": n_active / float(R * S),
            "shaded_sample_fraction": n_pos / float(R * S),
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
            "dense_equivalent_tflops": FLOP_ALL_PER_SAMPLE * R * S / (ms_step * 1e-3) / 1e12,
            "exchange": "all_gather_into_tensor [R,6] fp32 per rank (RCCL)" if use_dist else "none",
            "per_rank_frames": ("every rank renders the same synthetic frame as the N = 1 line (fixed per-GPU work)" if args.per_rank_frames == "same"
                                else "rank r renders its own pose (seeds 3 + r, 5 + r): the step waits for the slowest frame"),
        },
        "ranks": rk.info(per_rank_s, args.steps),
        "early_stop": stop_info,
    }

    if rank == 0 and world == 1 and not args.no_extras and not (args.dense or args.fp32):
        # The frame time is a property of the CHECKPOINT as much as of the kernels: the same frame, same pipeline,
        # 5 timed frames each, for every parameter set the repo pins with reference-generated goldens.  w4 is the converged one.
        headline = cur
        # (the headline's own counters, before the other parameter sets reuse the workspaces)
        headline_cw = wss[(k_step - 1) % depth].buf[:256].view(torch.int32).cpu()
        headline_st = _lib.read_stop_stats(wss[(k_step - 1) % depth])
        headline_ws_gb = _lib.lib().dsn_render_workspace_bytes(R, S) / 1e9
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
        share_cus[0] = False
        result["config"]["by_weights_note"] = ("same frame and pipeline for every parameter set, Renderer's defaults (density screen off, early "
                                               "stop decided by the probe frame); default = hash-random init (thin fog), w2 = 400 reference-"
                                               "trainer steps (solid, unsaturated), w3 = hash init x3.5 (dense guess), w4 = converged with "
                                               "scripts/train_w4.py: the headline.  with_density_screen: the same frames with the opt-in "
                                               "plain-fp16 screen (statistically safe: calibrated margin + audit), where its calibration lets it run")
    if rank == 0 and world == 1 and not args.no_roofline:
        scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)      # (the frame state of THESE parameters: by_weights has used the scene)
        result["roofline"] = roofline(_lib, scene, packed, ray_o, ray_d, near0, far0, S, t_vals, args, early=early, schedule=headline_schedule)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(synth, canon, faces, xyz, poses, sd, rays, S, args)
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
                scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True)
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
        ex["host_threads"] = torch.get_num_threads()
        ex["host_cpu_quota_cores"] = _lib.cpu_quota_cores()
        # the per-sample workspace is the caller's to size: the same render_view in four ray chunks (the reference's own loop runs
        # 3072-ray chunks, can_render.py:172-245) needs a quarter of it, for this much time
        chunk = (H * W) // 4
        ex["chunked_frame"] = {"chunk_rays": chunk,
                               "workspace_gb_whole_frame_headline": headline_ws_gb,      # (at the record fraction the headline's probe frame left)
                               "record_capacity_fraction_now": _lib.record_capacity_fraction(),      # (process-wide, grown by the densest set of by_weights)
                               "workspace_gb_whole_frame": _lib.lib().dsn_render_workspace_bytes(H * W, S) / 1e9,
                               "workspace_gb_chunked": _lib.lib().dsn_render_workspace_bytes(chunk, S) / 1e9,
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
            result["cpu_baseline_torch"] = cpu_baseline_torch(synth, canon, faces, xyz, poses, sd, rays, S, args)
    rk.finish()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(result), flush=True)      # the LAST line of stdout (RCCL prints its banner at its first collective)


def stop_setup(_lib, args, scene, packed, ws, o, d, near0, far0, S, t_vals, screen, reduce_max=None):
    """Front-to-back slicing for the secondary modes, decided like Renderer / the headline loop do: one probe render (one pass,
    DSN_STOP_STATS) of these rays says what termination would leave out, how large the colours are (-> the threshold's colour scale)
    and how the slices should be cut (choose_stop_schedule).  reduce_max(tensor): all-reduce MAX over the ranks of a multi-GPU run, so
    that every rank takes the same decision and threshold.  Returns (enabled, schedule | None, info dict)."""
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
    _lib.fit_record_capacity(int(ws.buf[64:68].view(torch.int32)[0]) / float(R * S) * ((1.0 - frac) if enabled else 1.0), 1.6 if enabled else 1.25)
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


def strong_bench(args, dsnerf_amd, _lib, synth, dev, world, rank, use_dist, rk):
    """BASELINE configs[3]: ONE 1024 x 1024 frame at 128 samples per ray, split over the ranks.  The rays are dealt in
    round-robin tiles of 3072 (with the transparent skip the rows through the torso cost several times the rows above the
    head; tiles even that out); every rank renders its tiles with the whole-frame kernels and ONE all_gather_into_tensor of
    equal slabs of packed [rays, 6] pixels brings the frame together on every rank, inside the timed region, followed by
    the un-dealing scatter into frame order.  value = rays of the frame / time: strong scaling."""
    import torch.distributed as dist
    if args.emulate_world > 1 and world == 1:
        return strong_emulated(args, dsnerf_amd, _lib, synth, dev)
    H = W = args.hw if args.hw != 512 else 1024
    S = args.samples if args.samples != 64 else 128
    R = H * W
    canon, faces = synth.make_body()
    sd = load_weights(synth, args.weights)
    poses = synth.make_poses(seed=5)
    xyz = synth.pose_body(canon, seed=3)
    rays = synth.make_rays(H, W, xyz, fit_box=True)
    rp = dsnerf_amd.RayParallel()
    plan = rp.tile_plan(R, 3072, dev)            # (cached per (R, tile, world, device): indices + un-dealing permutation on the device, built once)
    mine = plan["mine"].cpu().numpy()
    slab = plan["slab"]
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    ws = _lib.RenderWorkspace(dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # the geometry-guided sampler uses the batch's FIRST ray origin for every ray (utils/pts_utils.py:31): one camera, same origin
    o, d, near0, far0 = T(rays["ray_o"][mine]), T(rays["ray_d"][mine]), T(rays["near"][mine]), T(rays["far"][mine])
    d_xyz, d_poses = T(xyz), T(poses)
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    Rl = len(mine)
    ws.get(Rl, S)
    scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)
    info = packed.calibrate_screen(scene) if args.screen else {"usable": False, "note": "density screen not opted in (--screen)"}
    near, far = near0.clone(), far0.clone()
    # front-to-back slices with ray termination: decided like Renderer does, from the statistics of one probe render of this rank's
    # share, which also measures the colour scale of the threshold (set-up, not a step)
    stop_on, schedule, stop_info = stop_setup(_lib, args, scene, packed, ws, o, d, near0, far0, S, t_vals, info["usable"],
                                              reduce_max=(lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.MAX)) if use_dist else None)
    px = torch.zeros(slab, 6, dtype=torch.float32, device=dev)
    allp = torch.empty(world * slab, 6, dtype=torch.float32, device=dev)
    full = torch.empty(R, 6, dtype=torch.float32, device=dev)
    out = None

    def step():
        nonlocal out
        near.copy_(near0)
        far.copy_(far0)
        scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True)
        out = _lib.render_rays(scene, packed, ws, o, d, near, far, S, t_vals, None, None, want_weights=False, out=out,
                               screen=info["usable"], early_stop=stop_on, stop_schedule=schedule)
        px[:Rl, 0:3] = out["color"]
        px[:Rl, 3] = out["disp_map"]
        px[:Rl, 4] = out["acc_map"]
        px[:Rl, 5] = out["depth_map"]
        if use_dist:
            dist.all_gather_into_tensor(allp, px)
            rp.undeal_tiles(allp, R, 3072, out=full)      # ONE index_select through the cached permutation
        else:
            rp.undeal_tiles(px, R, 3072, out=full)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
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
    dt, per_rank_s = rk.times(dt)
    cnt = ws.buf[:256].view(torch.int32).cpu()
    ms = 1e3 * dt / args.steps
    res = {"metric": f"rendered rays/sec ({S} samples/ray), ONE {H}x{W} frame split over the GPUs", "value": R * args.steps / dt,
           "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None,
           "dtype": "split-f16x3 (3 x v_mfma_f32_32x32x16_f16 on hi/lo fp16 operand halves, f32 accumulate)"
                    + (" + plain-f16 density screen" if info["usable"] else ""),
           "data": "synthetic",
           "config": {"workload": f"one {H}x{W} frame x {S} samples/ray (BASELINE configs[3]) over {world} GPU(s): round-robin 3072-ray "
                                  f"tiles, {Rl} rays on rank 0, synthetic closed body V=6890/F=13776, all rays cross the body AABB, GG "
                                  f"sampling, eval mode, parameters: {args.weights}",
                      "weights": args.weights, "early_stop": stop_info,
                      "rays_on_rank0": Rl, "samples_per_ray": S, "ms_per_frame": ms,
                      "evaluated_sample_fraction_rank0": int(cnt[_lib.CNT_ACTIVE]) / float(Rl * S),
                      "accurate_pass_sample_fraction_rank0": int(cnt[_lib.CNT_KEEP]) / float(Rl * S),
                      "density_screen_calibration": info,
                      "dense_equivalent_tflops": FLOP_ALL_PER_SAMPLE * R * S / (ms * 1e-3) / 1e12,
                      "exchange": (f"all_gather_into_tensor [{slab},6] fp32 per rank (RCCL) + scatter to frame order, in the timed region"
                                   if use_dist else "none (scatter to frame order only)")},
           "ranks": rk.info(per_rank_s, args.steps)}
    rk.finish()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(res), flush=True)


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
                scenes[j].set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True)
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


def strong_emulated(args, dsnerf_amd, _lib, synth, dev):
    """Load balance of the strong-scaling partition, measured on ONE GPU (no multi-GPU node is available to the
    builder): the 1024 x 1024 x 128 frame of configs[3] is dealt to N = --emulate-world ranks exactly as strong_bench does
    (RayParallel.tile_indices, round-robin 3072-ray tiles) and every rank's share is rendered ALONE with the same code, one
    after the other, on this GPU.  Reported: the N share times, max / mean (the imbalance a real N-GPU run would wait for), the
    whole frame on one GPU, the un-dealing scatter of N gathered slabs (a local operation, timed here), and the strong-scaling
    efficiency these predict = T(1 GPU) / (N x (max share + un-deal + all-gather)); the all-gather is NOT measured (one GPU) - it
    is priced from the xGMI figures of MI355X_MICROARCH.md (each rank receives (N-1)/N of 24 B x R over its N-1 direct links at
    <= 153 GB/s each, plus a launch latency of 30 us): a stated estimate."""
    H = W = args.hw if args.hw != 512 else 1024
    S = args.samples if args.samples != 64 else 128
    R = H * W
    Nw = int(args.emulate_world)
    tile = 3072
    canon, faces = synth.make_body()
    sd = load_weights(synth, args.weights)
    poses = synth.make_poses(seed=5)
    xyz = synth.pose_body(canon, seed=3)
    rays = synth.make_rays(H, W, xyz, fit_box=True)
    rp = dsnerf_amd.RayParallel()
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    ws = _lib.RenderWorkspace(dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_xyz, d_poses = T(xyz), T(poses)
    t_vals = torch.linspace(0.0, 1.0, steps=S).to(dev)
    scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None)
    # (the centroid cube: the shares are rendered with one margin, whichever rank calibrates)
    info = packed.calibrate_screen(scene) if args.screen else {"usable": False, "note": "density screen not opted in (--screen)"}
    # (one decision for the frame, as the ranks of the real run agree on by all-reduce: probe on the whole frame; uniform slices)
    whole = rp.tile_indices(R, tile, 0, 1).numpy()
    stop_on, _, stop_info = stop_setup(_lib, args, scene, packed, ws, T(rays["ray_o"][whole]), T(rays["ray_d"][whole]), T(rays["near"][whole]),
                                       T(rays["far"][whole]), S, t_vals, info["usable"])
    frame_scale = packed.colour_scale

    def time_share(idx):
        mine = idx.numpy()
        o, d, near0, far0 = T(rays["ray_o"][mine]), T(rays["ray_d"][mine]), T(rays["near"][mine]), T(rays["far"][mine])
        near, far = near0.clone(), far0.clone()
        Rl = len(mine)
        px = torch.zeros(Rl, 6, dtype=torch.float32, device=dev)
        out = None
        ms = []
        # (a rank of the real run probes its own share: its own slice schedule; decision and colour scale are the frame's)
        schedule = None
        if stop_on:
            scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True)
            _, schedule, _ = stop_setup(_lib, args, scene, packed, ws, o, d, near0, far0, S, t_vals, info["usable"])
            packed.set_early_stop_colour_scale(frame_scale)
        for i in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            near.copy_(near0)
            far.copy_(far0)
            scene.set_frame(packed, d_xyz, d_poses, 5, False, None, None, None, fine_only=True)
            out = _lib.render_rays(scene, packed, ws, o, d, near, far, S, t_vals, None, None, want_weights=False, out=out,
                                   screen=info["usable"], early_stop=stop_on, stop_schedule=schedule)
            px[:, 0:3] = out["color"]
            px[:, 3] = out["disp_map"]
            px[:, 4] = out["acc_map"]
            px[:, 5] = out["depth_map"]
            torch.cuda.synchronize()
            if i >= args.warmup:
                ms.append(1e3 * (time.perf_counter() - t0))
        cnt = ws.buf[:256].view(torch.int32).cpu()
        return float(np.mean(ms)), float(np.min(ms)), Rl, int(cnt[_lib.CNT_ACTIVE]), int(cnt[_lib.CNT_KEEP]), int(cnt[_lib.CNT_POS])

    whole_ms, whole_min, _, a1, k1, p1 = time_share(rp.tile_indices(R, tile, 0, 1))
    shares = []
    for r in range(Nw):
        m, mn, Rl, a, k, p_ = time_share(rp.tile_indices(R, tile, r, Nw))
        shares.append({"rank": r, "rays": Rl, "ms": m, "ms_min": mn, "non_transparent": a, "accurate_pass": k, "positive_density": p_})
    # the un-dealing scatter of N equal slabs into frame order (strong_bench's epilogue behind the all-gather)
    slab = rp.tile_plan(R, tile, dev, world=Nw)["slab"]
    allp = torch.zeros(Nw * slab, 6, dtype=torch.float32, device=dev)
    full = torch.empty(R, 6, dtype=torch.float32, device=dev)
    und = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rp.undeal_tiles(allp, R, tile, out=full, world=Nw)      # ONE index_select through the cached permutation
        torch.cuda.synchronize()
        if i >= 2:
            und.append(1e3 * (time.perf_counter() - t0))
    undeal_ms = float(np.mean(und))
    t = np.array([x["ms"] for x in shares])
    ag_ms = 0.03 + 1e3 * (24.0 * R * (Nw - 1) / Nw) / ((Nw - 1) * 153e9) if Nw > 1 else 0.0
    step_ms = float(t.max()) + undeal_ms + ag_ms
    res = {"metric": f"strong-scaling load balance, ONE {H}x{W} frame x {S} samples/ray dealt to {Nw} emulated ranks on ONE GPU",
           "value": R / (step_ms * 1e-3), "unit": "rays/s (PREDICTED for the emulated world: max share + un-deal + priced all-gather)",
           "n_gpus": 1, "emulated_world": Nw, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
           "dtype": "split-f16x3" + (" + plain-f16 density screen" if info["usable"] else ""),
           "config": {"workload": f"one {H}x{W} frame x {S} samples/ray (BASELINE configs[3]); round-robin {tile}-ray tiles "
                                  f"(RayParallel.tile_indices); each emulated rank's share rendered alone on one MI355X",
                      "weights": args.weights, "tile": tile, "shares": shares,
                      "share_ms_max": float(t.max()), "share_ms_mean": float(t.mean()), "share_ms_min": float(t.min()),
                      "max_over_mean": float(t.max() / t.mean()),
                      "whole_frame_one_gpu_ms": whole_ms, "undeal_scatter_ms": undeal_ms,
                      "all_gather_ms_PRICED_not_measured": ag_ms,
                      "sum_of_shares_over_whole_frame": float(t.sum() / whole_ms),
                      "predicted_strong_scaling_efficiency": whole_ms / (Nw * step_ms),
                      "predicted_speedup": whole_ms / step_ms, "early_stop": stop_info,
                      "density_screen_calibration": info}}
    _flush_c_stdio()
    print(json.dumps(res), flush=True)


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


def roofline(_lib, scene, packed, ray_o, ray_d, near0, far0, S, t_vals, args, early=False, schedule=None):
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
    screen = split and not args.no_screen
    ms_screen = None
    n_all = n_eval
    if split:
        # The field kernels of a frame - screen -> accurate forward -> reverse - are enqueued back to back, exactly as
        # dsn_render_rays does, with an event between them and NO host synchronisation inside a repetition: a kernel timed
        # alone after an idle gap starts on a cool, fully clocked chip and reads ~4 % faster than it runs inside a frame.
        keep = torch.zeros(N, dtype=torch.int32, device=dev)
        kcnt = torch.zeros(64, dtype=torch.int32, device=dev)
        rec = torch.empty(L.dsn_field_record_bytes(C.c_int64(N)), dtype=torch.uint8, device=dev)
        pos = torch.zeros(N, dtype=torch.int32, device=dev)
        pcnt = torch.zeros(64, dtype=torch.int32, device=dev)
        f_lst, f_cnt = (keep, kcnt) if screen else (lst, cnt)
        t_s, t_f, t_r = [], [], []
        for i in range(reps + 2):                      # 2 untimed warm-up repetitions
            kcnt.zero_()
            pcnt.zero_()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            if screen:
                rc = L.dsn_field_screen(*a0, _lib._ptr(lst), _lib._ptr(cnt), _lib._ptr(sig), _lib._ptr(keep), _lib._ptr(kcnt),
                                        _lib._stream())
                assert rc == 0, L.dsn_last_error()
            ev[1].record()
            rc = L.dsn_field_forward(*a0, _lib._ptr(f_lst), _lib._ptr(f_cnt), _lib._ptr(sig), _lib._ptr(ess), _lib._ptr(rec),
                                     _lib._ptr(pos), _lib._ptr(pcnt), _lib._stream())
            assert rc == 0, L.dsn_last_error()
            ev[2].record()
            rc = L.dsn_field_reverse(*a0, _lib._ptr(pos), _lib._ptr(pcnt), _lib._ptr(rec), _lib._ptr(g), _lib._ptr(sig), _lib._ptr(ess),
                                     _lib._stream())
            assert rc == 0, L.dsn_last_error()
            ev[3].record()
            torch.cuda.synchronize()
            if i >= 2:
                t_s.append(ev[0].elapsed_time(ev[1])); t_f.append(ev[1].elapsed_time(ev[2])); t_r.append(ev[2].elapsed_time(ev[3]))
        if screen:
            ms_screen = float(np.mean(t_s))
            n_eval = int(kcnt[0])
        ms, ms_rev = float(np.mean(t_f)), float(np.mean(t_r))
        n_pos = int(pcnt[0])
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
    traffic, traffic_src = measured_traffic(kern, args)
    out = {"bound": "mfma", "kernel": kern, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
           "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": ms, "samples_per_launch": n_eval,
           "flop_per_sample": flop_per, "scheme": note, "x_fp32_matrix_peak": ach / PEAK_F32_MATRIX_TFLOPS}
    if kern == "k_field16<forward>":
        rp_ms, rp_src, _ = rocprof_kernel_ms("k_field16ILi1E", args)
        if rp_ms is not None:
            # (same sample count: the committed profile is of this command on the same frame)
            out["rocprof_kernel_ms"] = rp_ms
            out["rocprof_source"] = rp_src
            out["frac_at_rocprof_kernel_ms"] = n_eval * flop_per / (rp_ms * 1e-3) / 1e12 / peak
    if screen:
        ach_s = n_all * FLOP_SCREEN_PER_SAMPLE / (ms_screen * 1e-3) / 1e12
        out["screen_kernel"] = {"kernel": "k_screen16", "kernel_ms": ms_screen, "samples_per_launch": n_all,
                                "flop_per_sample": FLOP_SCREEN_PER_SAMPLE, "achieved": ach_s, "peak": PEAK_F16_MATRIX_TFLOPS,
                                "frac": ach_s / PEAK_F16_MATRIX_TFLOPS,
                                "scheme": "plain fp16 operands, fp32 accumulate: 1 MFMA product per algorithmic product"}
    if split:
        ach_r = n_pos * FLOP_FIELD_REV_PER_SAMPLE / (ms_rev * 1e-3) / 1e12
        out["reverse_kernel"] = {"kernel": "k_field16<reverse>", "kernel_ms": ms_rev, "samples_per_launch": n_pos,
                                 "flop_per_sample": FLOP_FIELD_REV_PER_SAMPLE, "achieved": ach_r, "frac": ach_r / peak}
    if split and early:
        # The frames of the timed loop run this kernel in SLICES (front to back, DSN_EARLY_STOP): one launch per slice on the samples
        # of rays that are still alive.  Their sizes are read from a real sliced frame (workspace words 64 / 96 + k), then the same
        # kernel is launched back to back on lists of exactly those sizes (prefixes of this frame's list of non-transparent samples -
        # the forward kernel gathers its points by index, which samples they are does not matter to it) between two events:
        # sum of samples x 0.918 MFLOP / sum of launch times = what the sliced forward achieves, launch tails included.
        ws2 = _lib.RenderWorkspace(dev)
        n2, f2 = near0.clone(), far0.clone()
        _lib.render_rays(scene, packed, ws2, ray_o, ray_d, n2, f2, S, t_vals, None, None, want_weights=False, screen=screen, early_stop=True,
                         stop_schedule=schedule)
        torch.cuda.synchronize()
        cw = ws2.buf[:1024].view(torch.int32).cpu()
        L_slice = _lib.stop_slice_len(R, S)
        K = len(schedule) if schedule is not None else (S + L_slice - 1) // L_slice
        base = 128 if screen else 96                       # (DSN_CNT_KEEP_K / DSN_CNT_ALIVE_K: what the forward launch of slice k ran on)
        sizes = [int(cw[base + k]) if (k > 0 or screen) else int(cw[64]) for k in range(K)]
        del ws2
        cnts = [torch.tensor([n_] + [0] * 15, dtype=torch.int32, device=dev) for n_ in sizes]
        t_sl = []
        for i in range(reps + 2):
            pcnt.zero_()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for c_ in cnts:
                rc = L.dsn_field_forward(*a0, _lib._ptr(lst), _lib._ptr(c_), _lib._ptr(sig), _lib._ptr(ess), _lib._ptr(rec), _lib._ptr(pos),
                                         _lib._ptr(pcnt), _lib._stream())
                assert rc == 0, L.dsn_last_error()
            b_.record()
            torch.cuda.synchronize()
            if i >= 2:
                t_sl.append(a_.elapsed_time(b_))
        ms_sl = float(np.mean(t_sl))
        ach_sl = sum(sizes) * FLOP_FIELD_FWD_PER_SAMPLE / (ms_sl * 1e-3) / 1e12
        # ... and THAT is the dominant kernel as the timed frames run it: the headline block is the per-launch average of the sliced
        # forward (achieved = average samples per launch x 0.918 MFLOP / average launch time); the single whole-frame launch measured
        # above moves to `single_launch`
        single = {k: out[k] for k in ("kernel", "achieved", "frac", "kernel_ms", "samples_per_launch", "x_fp32_matrix_peak")}
        rp_ms, rp_src, rp_calls = rocprof_kernel_ms("k_field16ILi1E", args, drop_largest=1)
        traffic, traffic_src = measured_traffic("k_field16<forward>", args)
        out.update({"kernel": "k_field16<forward>, one launch per front-to-back slice (DSN_EARLY_STOP): per-launch averages of a frame",
                    "achieved": ach_sl, "frac": ach_sl / peak, "kernel_ms": ms_sl / K, "samples_per_launch": sum(sizes) / K,
                    "launches_per_frame": K, "slice_lengths": schedule if schedule is not None else [L_slice] * K, "samples_per_slice": sizes, "samples_per_frame": sum(sizes), "sum_kernel_ms_per_frame": ms_sl,
                    "evaluated_fraction_of_non_transparent": sum(sizes) / max(1, n_all), "x_fp32_matrix_peak": ach_sl / PEAK_F32_MATRIX_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "single_launch_on_all_non_transparent_samples": single})
        out.pop("rocprof_kernel_ms", None); out.pop("rocprof_source", None); out.pop("frac_at_rocprof_kernel_ms", None)
        if rp_ms is not None:
            out["rocprof_kernel_ms"] = rp_ms
            out["rocprof_source"] = rp_src + f" ({rp_calls} launches; the one whole-frame launch of the set-up probe frame left out)"
            out["frac_at_rocprof_kernel_ms"] = (sum(sizes) / K) * FLOP_FIELD_FWD_PER_SAMPLE / (rp_ms * 1e-3) / 1e12 / peak
    return out


def _profile_file(stem):
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{stem}")
        if os.path.exists(path):
            return path
    return None


def _profile_tag(args):
    """which committed profile set belongs to this configuration: profiles/rNN_<tag>pmc.json / rNN_<tag>kernel_trace.txt"""
    if args.fp32 or args.dense or args.hw != 512 or args.samples != 64 or args.screen:
        return None
    return {"w4": "", "default": "default_"}.get(args.weights)


def measured_traffic(kern, args):
    """HBM bytes per launch of the dominant kernel - NOT measured in this run: read from the committed rocprofv3 PMC passes of this
    same command (profiles/rNN_pmc.json, written by scripts/pmc_summary.py from scripts/gpu.sh pmc: (2*FETCH_SIZE + WRITE_SIZE) KB,
    the gfx950 correction of MI355X_MICROARCH.md; counters need their own rocprofv3 passes, which a plain `python bench.py` is not).
    Returns (bytes | None, source string | None); None when no pass was collected for this configuration."""
    tag = _profile_tag(args)
    path = None if tag is None else _profile_file(tag + "pmc.json")
    if path is None:
        return None, None
    with open(path) as f:
        rec = json.load(f).get(kern)
    if rec is None:
        return None, None
    return rec["hbm_bytes_per_launch"], (f"{os.path.relpath(path, ROOT)} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                         f"`bench.py --steps 5 --warmup 2 --pipeline 1 --no-roofline`, average over the kernel's launches; "
                                         f"not collected in this run)")


def rocprof_kernel_ms(mangled_part, args, drop_largest=0):
    """average duration of a kernel in the committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/rNN_kernel_trace.txt) - beside the live HIP-event time, so that both fractions can be read off one line.
    drop_largest = 1: without the kernel's longest launch (total - max over calls - 1).  Returns (ms, file, launches)"""
    tag = _profile_tag(args)
    path = None if tag is None else _profile_file(tag + "kernel_trace.txt")
    if path is None:
        return None, None, None
    with open(path) as f:
        for line in f:
            if mangled_part in line.split(" ")[0]:
                cols = line.split()
                calls, total, avg, mx = int(cols[1]), float(cols[2]), float(cols[3]), float(cols[5])
                if drop_largest and calls > 1:
                    return (total - mx) / (calls - 1), os.path.relpath(path, ROOT), calls - 1
                return avg, os.path.relpath(path, ROOT), calls
    return None, None, None


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


if __name__ == "__main__":
    main()
