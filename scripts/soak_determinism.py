"""Soak: the same frame N times through the eval path - every frame must equal the first bit for bit (the list orders differ from
run to run, the per-sample values and the compositing order do not).  Catches rare races (weight ring across persistent tiles,
list building) that a single comparison would miss.   python scripts/soak_determinism.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "oracle"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), p))
import torch
from dsnerf_amd import _lib
from tests.test_gpu_round2 import _stop_pair
from helpers import state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, kw in (("default", {}), ("default, no screen", {"screen_off": True}), ("w3 early stop", {"early_stop": True}), ("w2", {})):
    sd = state("x_w3") if "w3" in name else (state("x_w2") if "w2" in name else state())
    run = _stop_pair(sd, hw=256, screen=not kw.pop("screen_off", False))
    ref, _, _ = run(**kw)
    bad = 0
    for i in range(n):
        got, _, _ = run(**kw)
        for k in ("color", "acc_map", "depth_map", "weights"):
            if not torch.equal(torch.nan_to_num(ref[k], nan=-1.0), torch.nan_to_num(got[k], nan=-1.0)):
                bad += 1
                print(name, "frame", i, k, "differs by", float((torch.nan_to_num(ref[k]) - torch.nan_to_num(got[k])).abs().max()))
                break
    print(f"{name}: {n} frames, {bad} differ from the first")
