#!/usr/bin/env python3
"""bench.py - rendered rays/s of the Dual-Space-NeRF hot path on MI355X.

A "step" = one pass of the whole hot path (per-frame setup, geometry-guided sampling, nearest-face warp,
canonical field + d sigma/dx, normals + lighting MLP, compositing) over one synthetic 512x512 frame at
64 samples/ray (BASELINE.json configs[1]) per GPU, inputs resident in HBM.  With N>1 (launched by
torch.distributed.run, one rank per GPU) the SAME frame is partitioned over the ranks (cost-balanced contiguous
ray blocks), every rank renders its share, and the rendered pixels are exchanged with one RCCL all-gather
+ one index_select into ray order inside the timed region: strong scaling, value = rays of the frame x
frames / max time (benchlib/strong.py).  --weak gives the weak line instead (every rank renders one whole
frame of a multi-frame batch, configs[4]); the strong line carries it as a secondary object.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     - k_field16<forward> (the dominant kernel), timed live with HIP events on the launch stream in a
                 separate stage-by-stage pass over the same frame; algorithmic FLOPs = evaluated samples
                 x 0.91776 MFLOP (2 x 458 880 MAC: trunk + heads); the reverse kernel (analytic d sigma/dx,
                 2 x 425 728 MAC per sigma > 0 sample) is reported beside it.
                 + whole_frame_frac (field FLOPs executed per frame / frame time / ceiling) and mfma_busy (committed PMC passes).
  cpu_baseline - the torch restatement of the reference's op sequence (oracle/train_oracle.py; north_star's "reference CPU
                 PyTorch path", which itself cannot travel) timed on the host cores on one 3072-ray chunk of the same frame
                 (rank 0, N=1 only); cpu_baseline_c: the OpenMP C port of the same algorithm (oracle/dsn_oracle.c) beside it.
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

from benchlib.common import Ranks, launch_ranks


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
                    help="strong-scaling mode: ONE --hw x --hw x --samples frame per step (default: the metric's own 512 x 512 x 64, BASELINE "
                         "configs[1]; --big-frame: configs[3]), its rays partitioned over the ranks (--partition), --pipeline frames in "
                         "flight, rendered pixels all-gathered + put back into ray order inside the timed region.  This is what --gpus N "
                         "with N > 1 runs unless --weak is given (north_star: rays partitioned across the GPUs, all-gather of pixels)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: the weak-scaling line instead (every rank renders one whole frame of a multi-frame batch, BASELINE "
                         "configs[4]; one all-gather of [R,6] pixels per frame).  The strong line reports it as a secondary object anyway")
    ap.add_argument("--big-frame", action="store_true", help="--strong: BASELINE configs[3], ONE 1024 x 1024 x 128 frame")
    ap.add_argument("--partition", default="blocks", choices=["blocks", "tiles"],
                    help="--strong: blocks = contiguous ray blocks cut for equal cost (per-ray cost from a probe frame: evaluated + shaded "
                         "samples) - a rank's samples stay in its own cells of the posed mesh's nearest-face grid; tiles = round-robin "
                         "--tile-ray tiles (no cost estimate needed, every rank visits every cell)")
    ap.add_argument("--tile", type=int, default=3072, help="--partition tiles: rays per tile at most (every rank gets the same number of tiles)")
    ap.add_argument("--emulate-sweep", default="", help="--strong --emulate-world: comma-separated world sizes to emulate in ONE run (e.g. 2,4,8)")
    ap.add_argument("--rebalance", type=int, default=1,
                    help="--strong --emulate-world with blocks: rounds of measured re-balancing (cuts moved by the measured share times)")
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


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        launch_ranks(args)          # (does not return)
    rk = Ranks(args)
    world, rank, dev, use_dist = rk.world, rk.rank, rk.dev, rk.on
    if world > 1 and not (args.weak or args.train or args.eager_baseline):
        args.strong = True          # N > 1: the metric's frame partitioned over the ranks, unless --weak
    if args.dry_launch:
        from benchlib.dry import dry_launch
        return dry_launch(args, rk)

    import dsnerf_amd
    from dsnerf_amd import _lib, synth

    if args.train:
        from benchlib.train import train_bench
        return train_bench(args, dsnerf_amd, synth, dev, world, rank, use_dist, rk)
    if args.eager_baseline:
        from benchlib.baselines import eager_baseline
        print(json.dumps(eager_baseline(args, _lib, synth, dev)))
        return
    if args.strong:
        from benchlib.strong import strong_bench
        return strong_bench(args, dsnerf_amd, _lib, synth, dev, world, rank, use_dist, rk)
    from benchlib.frame import frame_bench, weak_emulated
    if args.emulate_world > 1 and world == 1:
        return weak_emulated(args, dsnerf_amd, _lib, synth, dev)
    return frame_bench(args, dsnerf_amd, _lib, synth, rk)


if __name__ == "__main__":
    main()
