"""cost of launching the field kernels' full grid (one workgroup per 128 samples of N) when the device-side list is
empty: every workgroup loads the count and exits"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsnerf_amd
from dsnerf_amd import _lib, synth
dev = torch.device("cuda:0")
canon, faces = synth.make_body()
sd = synth.make_state_dict()
packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
scene = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
scene.set_frame(packed, torch.from_numpy(synth.pose_body(canon)), torch.from_numpy(synth.make_poses()), 5)
N = 512 * 512 * 64
L = _lib.lib()
x = torch.zeros(N, 3, device=dev); sig = torch.zeros(N, device=dev); ess = torch.zeros(N, 3, device=dev); g = torch.zeros(N, 3, device=dev)
rec = torch.empty(L.dsn_field_record_bytes(C.c_int64(N)), dtype=torch.uint8, device=dev)
lst = torch.zeros(N, dtype=torch.int32, device=dev); cnt = torch.zeros(64, dtype=torch.int32, device=dev)
pos = torch.zeros(N, dtype=torch.int32, device=dev); pcnt = torch.zeros(64, dtype=torch.int32, device=dev)
a0 = (_lib._ptr(scene.buf), scene.V, scene.F, _lib._ptr(packed.buf), _lib._ptr(x), C.c_int64(N))
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("empty forward grid  %.3f ms" % timed(lambda: L.dsn_field_forward(*a0, _lib._ptr(lst), _lib._ptr(cnt), _lib._ptr(sig), _lib._ptr(ess), _lib._ptr(rec), _lib._ptr(pos), _lib._ptr(pcnt), _lib._stream())))
print("empty reverse grid  %.3f ms" % timed(lambda: L.dsn_field_reverse(*a0, _lib._ptr(pos), _lib._ptr(pcnt), _lib._ptr(rec), _lib._ptr(g), _lib._ptr(sig), _lib._ptr(ess), _lib._stream())))
print("empty screen grid   %.3f ms" % timed(lambda: L.dsn_field_screen(*a0, _lib._ptr(lst), _lib._ptr(cnt), _lib._ptr(sig), _lib._ptr(pos), _lib._ptr(pcnt), _lib._stream())))
