"""How many RAYS are still alive at each sample index of the bench frame (512 x 512 x 64, converged parameters)?  The nearest-face search
runs on every sample of a ray (transparency is only known behind it), so what a geometry pass that follows the early-stop slices
could leave out is (rays finished at the split) x (samples behind it) - not the share of non-transparent samples termination leaves out."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = 64
canon, faces, batch = full_frame(hw=HW)
r = renderer_with(state("x_w4"), canon, faces, density_screen=False)
r.eval(); r.early_stop = False
out = r.render(batch)["coarse"]
w = out["weights"].double()
R = w.shape[0]
T = 1.0 - torch.cumsum(w, 1)                    # transmittance BEHIND sample s
eps = _lib.early_stop_eps(S, 2.64)
tr = None
print("rays", R, "eps", eps, "acc > 0.99:", float((out["acc_map"] > 0.99).float().mean()), "acc < 0.01:", float((out["acc_map"] < 0.01).float().mean()))
alive = [(T[:, s - 1] >= eps).float().mean().item() if s > 0 else 1.0 for s in range(S + 1)]
for s in (0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56):
    searched = s / S + alive[s] * (S - s) / S
    print(f"split at sample {s:2d}: rays alive {alive[s]:.3f}  -> samples a two-stage geometry pass searches: {searched:.3f} of all")
# finer: a geometry pass per slice of the schedule [12, 4, 4, 4, 8, 12, 20]
b = [0, 12, 16, 20, 24, 32, 44, 64]
tot = sum(alive[b[k]] * (b[k + 1] - b[k]) for k in range(7)) / S
print("geometry per slice of the schedule [12,4,4,4,8,12,20]:", round(tot, 3), "of all samples")
