import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import oracle as O
from helpers import load
import dsnerf_amd
from dsnerf_amd import _lib, synth
g = load("full_eval"); dev = torch.device("cuda:0")
for gain in (1.6, 2.5, 4.0):
    sd = synth.make_state_dict(gain=gain)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    sc = _lib.Scene(torch.from_numpy(g["canonical_vertex"]), torch.from_numpy(g["faces"].astype(np.int64)), dev)
    sc.set_frame(packed, torch.from_numpy(g["xyz"]), torch.from_numpy(g["poses"]), int(g["frame"]))
    x = torch.from_numpy(g["x_c"]).to(dev)
    sig, ess, gr = _lib.field(sc, packed, x)
    P = O.Params(sd)
    _, pf = O.pose_feat(g["poses"], P)
    osig, oess, ogr = O.field(g["x_c"], P, sd["nerf.embedding.weight"][int(g["frame"])], pf)
    s = sig.cpu().numpy(); e = ess.cpu().numpy()
    sg, s1 = _lib.screen_debug(sc, packed, x)
    rel = ((sg.cpu().numpy() - osig) / s1.cpu().numpy())
    print(f"gain {gain}: |sigma| max {np.abs(osig).max():.1f}; sigma err max {np.abs(s-osig).max():.2e} (rel to max {np.abs(s-osig).max()/np.abs(osig).max():.1e}); essence err {np.abs(e-oess).max():.2e}; screen dev/S1 max {np.abs(rel).max():.2e}")
