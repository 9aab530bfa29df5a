"""bench.py internals shared by every mode: constants of the roofline accounting, parameter sets, the process group of a run
(one rank per GPU over RCCL), the self-launcher, and the readers of the committed rocprofv3 summaries under profiles/."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

FLOP_FIELD_PER_SAMPLE = 2.0 * 884608.0       # k_field: forward trunk+heads 458 880 MAC + reverse 425 728 MAC
FLOP_FIELD_FWD_PER_SAMPLE = 2.0 * 458880.0   # k_field16<forward>: trunk + density/essence heads
FLOP_FIELD_REV_PER_SAMPLE = 2.0 * 425728.0   # k_field16<reverse>: analytic d sigma/dx
FLOP_SCREEN_PER_SAMPLE = 2.0 * 425728.0      # k_screen16: trunk + density head, one fp16 product per algorithmic product
FLOP_ALL_PER_SAMPLE = 2.0 * 902272.0         # + lighting MLP 17 664 MAC
PEAK_F32_MATRIX_TFLOPS = 157.3               # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MATRIX_TFLOPS = 2500.0              # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
# per-frame set-up the way Renderer does it for eval frames: the posed mesh's nearest-face lists built by the frame's own render call for
# the cells its samples visit (DSN_FRAME_LAZY_LISTS); DSN_BENCH_LAZY_LISTS=0 = every cell's lists in dsn_set_frame (rounds 1-4), for A/B
LAZY_LISTS = os.environ.get("DSN_BENCH_LAZY_LISTS", "1") != "0"
SPLIT_PRODUCTS = 3                           # split-fp16: 3 f16 MFMA products per algorithmic product


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


def fail_line(args, rank, world, what, code=3):
    """A run that cannot measure says so in ONE JSON line on stdout (the driver parses the last line of stdout: it finds `error`, no
    value) and exits non-zero - instead of a traceback, or of N ranks waiting for each other until the driver's own clock runs out."""
    _flush_c_stdio()
    line = json.dumps({"metric": "rendered rays/sec", "value": None, "unit": "rays/s", "n_gpus": world, "steps": getattr(args, "steps", None),
                       "warmup": getattr(args, "warmup", None), "error": what, "failed_rank": rank})
    # ONE write() of "\n" + line + "\n" (atomic on a pipe below PIPE_BUF): several ranks fail at the same moment and share the launcher's
    # stdout - print() wrote the text and its newline separately and two ranks' lines could end up on one (seen once in this container)
    try:
        sys.stdout.flush()
        os.write(1, ("\n" + line + "\n").encode())
    except OSError:
        print(line, flush=True)
    _flush_c_stdio()
    os._exit(code)


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
            n_dev = torch.cuda.device_count()
            if idx >= n_dev:
                fail_line(args, self.rank, self.world, f"rank {self.rank}: LOCAL_RANK {self.local} has no GPU ({n_dev} visible on this node; "
                                                       f"HIP_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES')!r})")
            self.dev = torch.device("cuda", idx)
            torch.cuda.set_device(self.dev)
        # Preflight of a multi-rank run nobody has watched yet.  (1) every rank says which device it bound (stderr);
        # (2) rendezvous, communicator set-up and the FIRST collective run under a watchdog - a rank that never arrives (crashed, wrong
        # MASTER_PORT, a device another process holds) would leave the others inside RCCL for ever: after DSN_BENCH_INIT_TIMEOUT
        # seconds (default 180) the waiting rank prints a JSON error line and exits; (3) that first collective counts the ranks - an
        # all-reduce of ones must return N before anything is timed.  DSN_BENCH_TEST_FAIL (tests): provoke each failure.
        test_fail = os.environ.get("DSN_BENCH_TEST_FAIL", "")
        if test_fail == f"nodev_rank{self.rank}":
            fail_line(args, self.rank, self.world, f"rank {self.rank}: LOCAL_RANK {self.local} has no GPU (provoked: DSN_BENCH_TEST_FAIL)")
        if self.world > 1 or self.on:
            name = "cpu (dry launch)" if self.dry else f"cuda:{self.dev.index} {torch.cuda.get_device_name(self.dev)}"
            print(f"[bench rank {self.rank}/{self.world}] LOCAL_RANK {self.local} -> {name}; "
                  f"visible devices {0 if self.dry else torch.cuda.device_count()}; MASTER {os.environ.get('MASTER_ADDR', '127.0.0.1')}:"
                  f"{os.environ.get('MASTER_PORT', '29531')}", file=sys.stderr, flush=True)
        self.seen = 1
        if self.on:
            import datetime
            import threading
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            self.backend = "gloo" if self.dry else os.environ.get("DSN_BENCH_BACKEND", "nccl")
            limit = float(os.environ.get("DSN_BENCH_INIT_TIMEOUT", "180"))
            stage = ["rendezvous (init_process_group)"]
            watchdog = threading.Timer(limit, lambda: fail_line(
                args, self.rank, self.world, f"rank {self.rank}: {stage[0]} did not finish within {limit:g} s - a rank is missing or the "
                                              f"collective library cannot reach it (backend {self.backend}); nothing was timed", 4))
            watchdog.daemon = True
            watchdog.start()
            if test_fail == f"hang_rank{self.rank}":
                time.sleep(10 * limit + 3600)
            try:
                tmo = datetime.timedelta(seconds=max(limit, 30.0))
                if self.backend == "gloo":
                    dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=tmo)
                else:
                    dist.init_process_group("nccl", device_id=self.dev, rank=self.rank, world_size=self.world, timeout=tmo)
                stage[0] = "the first collective (all-reduce of ones: communicator set-up)"
                one = torch.ones(1, dtype=torch.int32, device=self.dev)
                if test_fail == "miscount" and self.rank == self.world - 1:
                    one.zero_()
                dist.all_reduce(one)
                self.seen = int(one.item())
            except Exception as e:      # (a rendezvous / communicator error is a failed preflight too, not a traceback)
                watchdog.cancel()
                fail_line(args, self.rank, self.world, f"rank {self.rank}: {stage[0]} failed: {type(e).__name__}: {str(e)[:300]}", 4)
            watchdog.cancel()
            if self.seen != self.world or dist.get_world_size() != self.world or (args.gpus is not None and dist.get_world_size() != args.gpus):
                fail_line(args, self.rank, self.world, f"rank {self.rank}: the collective library counted {self.seen} rank(s) (all-reduce of ones), "
                                                       f"world size {dist.get_world_size()}, expected {self.world}; nothing was timed", 5)

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
           "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    env = dict(os.environ, DSN_BENCH_LAUNCHER="bench.py --gpus N -> torch.distributed.run", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4" if args.dry_launch else str(max(1, (os.cpu_count() or 8) // max(1, args.gpus)))))
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def _profile_file(stem):
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
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


def measured_mfma_busy(kern, args):
    """share of the kernel's SIMD cycles with the matrix pipe busy, from the committed PMC passes of this command
    (SQ_VALU_MFMA_BUSY_CYCLES over per-XCD GRBM_GUI_ACTIVE x the chip's 1024 SIMDs; scripts/pmc_summary.py) - not collected in this run"""
    tag = _profile_tag(args)
    path = None if tag is None else _profile_file(tag + "pmc.json")
    if path is None:
        return None, None
    with open(path) as f:
        rec = json.load(f).get(kern)
    if rec is None or rec.get("mfma_busy_frac") is None:
        return None, None
    return rec["mfma_busy_frac"], os.path.relpath(path, ROOT)


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

