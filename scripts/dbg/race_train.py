"""Round 6: the co-residency hazard of DESIGN 4.5, for the kernels round 5 did not try.

  aggressors : the training backward (k_tangent16, k_adjoint16, k_t_wgrad16d / 16p, k_t_lin, k_t_wgrad ...) as ONE stream of kernels,
               the training forward (k_field16<train>), and round 5's known aggressors as controls (field16 forward / reverse)
  victims    : (a) a frame's shading phase (k_normal + k_light16 + compositor: round 5's most sensitive victim),
               (b) a frame's geometry phase (sampler + k_nns_search<warp>),
               (c) torch kernels of a CALLER: an element-wise chain with three loads per element, an fp32 GEMM, a row gather

Every victim runs alone first (reference), then beside the aggressor on another stream, `REPS` times; what is printed is the number of
differing elements per repetition (0 = bit-identical).  `python scripts/dbg/race_train.py [reps]`."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib, synth

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
HW = 256
canon, faces, batch = full_frame(hw=HW)
sd = state("x_w4")
r = renderer_with(sd, canon, faces, density_screen=False)
r.eval()
dev = r.device
S = 64
N = HW * HW * S
o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
n0, f0 = r._dev(batch["near"][0]), r._dev(batch["far"][0])
xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
pk = r.net.packed(dev)
tv = r._t_vals(S)
mk = lambda: (_lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev), _lib.RenderWorkspace(dev))
(sa, wa), (sb, wb) = mk(), mk()
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
al = lambda n: (n + 255) // 256 * 256


def arrays(ws):
    """per-sample arrays of a render workspace (dsn_carve): transparent, x_c, sigma, n_w, colour"""
    b = ws.buf
    p = 8192 + al(4 * N)
    out = {"transparent": b[p:p + N].clone()}
    p += al(N) + al(4 * N)
    out["x_c"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone()
    p += al(12 * N)
    out["sigma"] = b[p:p + 4 * N].view(torch.float32).clone()
    p += al(4 * N)
    out["n_w"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone()
    p += 12 * N
    out["colour"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone()
    return out


PH = {"geom": _lib.PHASE_GEOMETRY, "field": _lib.PHASE_FIELD, "shade": _lib.PHASE_SHADE}


def run(scene, ws, phases, out=None, nf=None):
    nn, ff = nf if nf is not None else (n0.clone(), f0.clone())
    for ph in phases:
        if ph == "set":
            scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True)
        else:
            out = _lib.render_rays(scene, pk, ws, o, d, nn, ff, S, tv, phases=PH[ph], out=out)
    return out, (nn, ff)


run(sb, wb, ["set", "geom", "field", "shade"])
torch.cuda.synchronize()
ref = arrays(wb)
pos = ref["sigma"] > 0
live = ref["transparent"] == 0

# ---- aggressor material ---------------------------------------------------------------------------------------------------
Rt = 8192
sel = np.linspace(0, HW * HW - 1, Rt).astype(np.int64)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
params = {k: T(v) for k, v in sd.items()}
st = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
st.set_frame(pk, xyz, poses, 5)
ot, dt = o[sel].contiguous(), d[sel].contiguous()
jit = T(synth.hash_uniform(Rt * S, 21).reshape(Rt, S).astype(np.float32))
_, zt = _lib.sample(st, ot, dt, n0[sel].clone(), f0[sel].clone(), S, tv, jit, want_pts=False)
noise = T((synth.hash_uniform(Rt * S, 22).reshape(Rt, S).astype(np.float32) - 0.5) * 2.0)
d_rgb = T(synth.hash_uniform(Rt * 3, 23).reshape(Rt, 3).astype(np.float32) - 0.5)
gws = _lib.GradWorkspace(dev)
g_ref = [x.clone() for x in _lib.render_rays_grad(st, params, poses, 5, False, ot, dt, zt, noise, d_rgb, ws=gws)]
torch.cuda.synchronize()

sa.set_frame(pk, xyz, poses, 5, False, None, None, None)
pts, z = _lib.sample(sa, o, d, n0.clone(), f0.clone(), S, tv, None, want_pts=True)
w = _lib.warp(sa, pts, d, S, want_dir=False, want_active=True)
xc, act = w["x_c"], (w["active_list"], w["active_count"])
fwd = _lib.field_forward(sa, pk, xc, active=act)
torch.cuda.synchronize()


def agg_train():
    _lib.render_rays_grad(st, params, poses, 5, False, ot, dt, zt, noise, d_rgb, ws=gws)


def agg_fwd():
    _lib.field_forward(sa, pk, xc, active=act)


def agg_rev():
    _lib.field_reverse(sa, pk, xc, fwd[2], fwd[3], fwd[0], fwd[1])


def agg_none():
    pass


AGG = (("none", agg_none), ("training step (forward + backward kernels)", agg_train), ("field16 forward", agg_fwd), ("field16 reverse", agg_rev))


def diffs(a, b, mask=None):
    a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
    ne = a != b
    if ne.dim() > 1:
        ne = ne.any(-1)
    if mask is not None:
        ne = ne & mask
    return int(ne.sum())


# ---- victim (a): the shading phase, (b): the geometry phase ------------------------------------------------------------------
for name, fn in AGG:
    res_s, res_g = [], []
    for rep in range(REPS):
        with torch.cuda.stream(B):
            ob, nfb = run(sb, wb, ["set", "geom", "field"])
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            fn()
        with torch.cuda.stream(B):
            run(sb, wb, ["shade"], ob, nfb)
        torch.cuda.synchronize()
        a = arrays(wb)
        res_s.append((diffs(a["n_w"], ref["n_w"], pos), diffs(a["colour"], ref["colour"], pos)))
        with torch.cuda.stream(B):
            sb.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            fn()
        with torch.cuda.stream(B):
            run(sb, wb, ["geom"])
        torch.cuda.synchronize()
        a = arrays(wb)
        res_g.append((diffs(a["transparent"], ref["transparent"]), diffs(a["x_c"], ref["x_c"], live)))
    print(f"aggressor {name:45s} -> victim shade (n_w, colour): {res_s}   victim geometry (transparent, x_c): {res_g}", flush=True)

# ---- victim (c): a caller's torch kernels -----------------------------------------------------------------------------------
M = 1 << 24
x1, x2, x3 = (torch.rand(M, device=dev) for _ in range(3))
ga, gb = torch.rand(2048, 2048, device=dev), torch.rand(2048, 2048, device=dev)
tab = torch.rand(1 << 20, 16, device=dev)
idx = torch.randint(0, 1 << 20, (1 << 21,), device=dev)


def torch_victims():
    e = x1 * x2 + x3
    e = e * x1 - x2
    g = ga @ gb
    s = tab.index_select(0, idx)
    return e, g, s


t_ref = torch_victims()
torch.cuda.synchronize()
for name, fn in AGG:
    res = []
    for rep in range(REPS):
        with torch.cuda.stream(A):
            fn()
        with torch.cuda.stream(B):
            got = torch_victims()
        torch.cuda.synchronize()
        res.append(tuple(diffs(a, b) for a, b in zip(got, t_ref)))
    print(f"aggressor {name:45s} -> victim torch (elementwise, fp32 gemm, gather): {res}", flush=True)

# ---- the training step as VICTIM of the field kernels (its small kernels beside k_field16 of a frame) ----------------------------
for name, fn in AGG:
    if fn is agg_train:
        continue
    res = []
    for rep in range(REPS):
        with torch.cuda.stream(A):
            fn()
            fn()
        with torch.cuda.stream(B):
            g = _lib.render_rays_grad(st, params, poses, 5, False, ot, dt, zt, noise, d_rgb, ws=gws)
        torch.cuda.synchronize()
        # (the lighting / colour-head gradients are sums with float atomics: compare with a tolerance, the trunk bit for bit)
        res.append(sum(int(not torch.equal(a, b)) for a, b in zip(g, g_ref)))
    print(f"aggressor {name:45s} -> victim training step: tensors (of 33) not bit-identical {res}", flush=True)
