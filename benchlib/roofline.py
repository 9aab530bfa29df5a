"""roofline block of the bench line: the dominant kernel timed live with HIP events, beside the committed rocprofv3 summaries."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from .common import (FLOP_FIELD_FWD_PER_SAMPLE, FLOP_FIELD_PER_SAMPLE, FLOP_FIELD_REV_PER_SAMPLE, FLOP_SCREEN_PER_SAMPLE, PEAK_F16_MATRIX_TFLOPS,
                     PEAK_F32_MATRIX_TFLOPS, SPLIT_PRODUCTS, measured_traffic, rocprof_kernel_ms)


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
        # ... and the reverse kernel as the sliced frames run it: on the samples whose weight passes the threshold (workspace word 13),
        # a prefix of the sigma > 0 list of the single-launch pass (records of every slot exist); the one-pass figure moves to `single_launch`
        n_sel = int(cw[_lib.CNT_SEL])
        if n_sel > 0 and "reverse_kernel" in out:
            scnt = torch.tensor([min(n_sel, n_pos)] + [0] * 15, dtype=torch.int32, device=dev)
            t_rs = []
            for i in range(reps + 2):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                rc = L.dsn_field_reverse(*a0, _lib._ptr(pos), _lib._ptr(scnt), _lib._ptr(rec), _lib._ptr(g), _lib._ptr(sig), _lib._ptr(ess), _lib._stream())
                b_.record()
                assert rc == 0, L.dsn_last_error()
                torch.cuda.synchronize()
                if i >= 2:
                    t_rs.append(a_.elapsed_time(b_))
            ms_rs = float(np.mean(t_rs))
            ach_rs = min(n_sel, n_pos) * FLOP_FIELD_REV_PER_SAMPLE / (ms_rs * 1e-3) / 1e12
            single["reverse_kernel"] = out["reverse_kernel"]
            out["reverse_kernel"] = {"kernel": "k_field16<reverse> on the samples a sliced frame shades", "kernel_ms": ms_rs,
                                     "samples_per_launch": min(n_sel, n_pos), "flop_per_sample": FLOP_FIELD_REV_PER_SAMPLE, "achieved": ach_rs,
                                     "frac": ach_rs / peak}
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

