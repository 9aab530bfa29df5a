"""Build libdsnerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dual-space-nerf_amd/build.py [--force]

-ffp-contract=off is part of the numerical contract (csrc/dsn_common.h): fmaf() marks every fusion.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dsn_api.hip", "dsn_geom.hip", "dsn_nn.hip", "dsn_field.hip", "dsn_field16.hip", "dsn_train.hip", "dsn_image.hip"]
HEADERS = ["dsn_common.h", "dsn_nn.h", "dsn_kernels.h", os.path.join("..", "..", "include", "dsnerf.h")]
LIB = os.path.join(HERE, "libdsnerf_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wno-unused-result", "-Wno-inline-asm"]   # inline-asm: the declared m0 clobber of the LDS-DMA statements


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        # no library dependencies beyond the HIP runtime: every kernel is in-tree
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
