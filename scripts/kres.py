#!/usr/bin/env python3
"""Register / spill / LDS table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kres.py dual-space-nerf_amd/csrc/dsn_field16.hip [-DFOO=1 ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-inline-asm",
       "-Wno-unused-result", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    print(f"{r['name'][:44]:44s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} "
          f"{r.get('VGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
