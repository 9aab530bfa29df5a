"""Numeric parity report (GPU box): per stage, per golden case: mismatch counts / max abs diff vs the reference goldens."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import CASES, load, state, light_kw, maxdiff
import dsnerf_amd
from dsnerf_amd import _lib
dev = torch.device("cuda:0")
sd = state()
packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def mm(a, b): a = a.cpu().numpy() if torch.is_tensor(a) else a; return f"{int((a != b).sum())}/{b.size} (max {maxdiff(a, b):.2e})"
for name in CASES:
    g = load(name); S = int(g["S"])
    sc = _lib.Scene(torch.from_numpy(g["canonical_vertex"]), torch.from_numpy(g["faces"].astype(np.int64)), dev)
    kw = light_kw(g); t = lambda k: (torch.from_numpy(np.ascontiguousarray(kw[k])) if k in kw else None)
    sc.set_frame(packed, torch.from_numpy(g["xyz"]), torch.from_numpy(g["poses"]), int(g["frame"]), zero_code=(name == "small_novel"),
                 light_shift=t("light_shift"), rot=t("rot"), rot_center=t("rot_center"))
    tv = torch.linspace(0.0, 1.0, steps=S).to(dev)
    jit = T(g["jitter"][0]) if "jitter" in g.files else None
    near, far = T(g["near"]), T(g["far"])
    pts, z = _lib.sample(sc, T(g["ray_o"]), T(g["ray_d"]), near, far, S, tv, jit)
    print(f"== {name}: near {mm(near, g['near_gg'])} far {mm(far, g['far_gg'])} z {mm(z, g['z_vals'])} pts {mm(pts, g['pts'])}")
    w = _lib.warp(sc, T(g["pts"]), T(g["ray_d"]), S, want_dir=True, want_uvh=True)
    print(f"   warp: idx {mm(w['face_idx'], g['idx_world'])} uv {mm(w['uv'], g['uv'])} h {mm(w['h'], g['h'])} x_c {mm(w['x_c'], g['x_c'])} rdc {mm(w['ray_d_can'], g['ray_d_can'])} tr {mm(w['transparent'].bool(), g['transparent'])}")
    sig, ess, gr = _lib.field(sc, packed, T(g["x_c"]))
    print(f"   field: sigma max {maxdiff(sig.cpu().numpy(), g['sigma']):.2e} ess {maxdiff(ess.cpu().numpy(), g['essence']):.2e} grad {maxdiff(gr.cpu().numpy(), g['grad_sigma']):.2e} (scale {np.abs(g['grad_sigma']).max():.0f})")
    idx, n_w, col = _lib.shade(sc, packed, T(g["x_c"]), T(g["grad_sigma"]), T(g["pts"]), T(g["ray_d"]), T(g["essence"]), S)
    print(f"   shade: idx_c {mm(idx, g['idx_canon'])} n_w {mm(n_w, g['n_w'])} colour max {maxdiff(col.cpu().numpy(), g['colour']):.2e}")
