"""Phase cycle counters of k_field16<forward> (library variant built with -DF16_TIMING=1, scripts/variants.sh):
   DSNERF_LIB=.../variants/timing.so python scripts/timing_probe.py"""
import ctypes as C, os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsnerf_amd
from dsnerf_amd import _lib
L = _lib.lib() if callable(getattr(_lib, "lib", None)) else C.CDLL(_lib.LIB_PATH)
L.dsn_debug_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--pipeline", "1", "--no-roofline"]
L.dsn_debug_timing(None, 1)
bench.main()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert L.dsn_debug_timing(out, 0) == 0
n = out[7]
names = ["prologue (s_vec, first chunk DMA, barrier)", "positional encoding", "trunk: 448 blocks", "heads + stores"]
for i, nm in enumerate(names):
    cyc, rt = out[i] / max(n, 1), out[8 + i] / max(n, 1) * 10.0   # s_memrealtime: 100 MHz -> ns
    print(f"{nm:46s} {cyc:10.0f} shader cycles  {rt/1e3:8.2f} us   ({cyc / max(rt, 1e-9):.2f} GHz)")
print("workgroups", n, " sum per workgroup %.2f us" % (sum(out[8 + i] for i in range(4)) / max(n, 1) * 10.0 / 1e3))
