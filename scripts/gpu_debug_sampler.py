import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import load, state
import dsnerf_amd
from dsnerf_amd import _lib
dev = torch.device("cuda:0")
sd = state()
packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
for name in ("small_eval", "full_eval"):
    g = load(name); S = int(g["S"])
    sc = _lib.Scene(torch.from_numpy(g["canonical_vertex"]), torch.from_numpy(g["faces"].astype(np.int64)), dev)
    sc.set_frame(packed, torch.from_numpy(g["xyz"]), torch.from_numpy(g["poses"]), int(g["frame"]))
    tv = torch.linspace(0.0, 1.0, steps=S)
    print(name, "linspace equals golden-implied?", end=" ")
    near, far = torch.from_numpy(g["near"].copy()).to(dev), torch.from_numpy(g["far"].copy()).to(dev)
    pts, z = _lib.sample(sc, torch.from_numpy(g["ray_o"]).to(dev), torch.from_numpy(g["ray_d"]).to(dev), near, far, S, tv.to(dev), None)
    n, f, z = near.cpu().numpy(), far.cpu().numpy(), z.cpu().numpy()
    print("near mism", (n != g["near_gg"]).sum(), "far mism", (f != g["far_gg"]).sum(), "z mism", (z != g["z_vals"]).sum(), "of", z.size,
          "max", np.abs(z - g["z_vals"]).max())
    # implied t from golden: rays where near/far equal
    ok = (n == g["near_gg"]) & (f == g["far_gg"])
    zz = g["z_vals"][ok]; zg = z[ok]
    print("  z mism on rays with identical near/far:", (zz != zg).sum(), "cols:", np.unique(np.nonzero(zz != zg)[1])[:20])
    tvn = tv.numpy()
    zc = (n[:, None] * (np.float32(1) - tvn[None]) ).astype(np.float32) + (f[:, None] * tvn[None]).astype(np.float32)
    print("  host recompute with this box's linspace vs GPU:", (zc != z).sum(), " vs golden:", (zc != g["z_vals"]).sum())
    import oracle as O
    print("  linspace(S) vs oracle scalar formula mism:", (O.linspace01(S) != tvn).sum(), "cpu capability", torch.backends.cpu.get_cpu_capability())
