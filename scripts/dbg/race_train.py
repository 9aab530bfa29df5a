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
NF = int(os.environ.get("RACE_FRAMES", "14"))          # victim frames: their phases run back to back on stream B and span the aggressor
mk = lambda: (_lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev), _lib.RenderWorkspace(dev))
sa, wa = mk()
VF = [mk() for _ in range(NF)]
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


run(*VF[0], ["set", "geom", "field", "shade"])
torch.cuda.synchronize()
ref = arrays(VF[0][1])
pos = ref["sigma"] > 0
live = ref["transparent"] == 0

# ---- aggressor material ---------------------------------------------------------------------------------------------------
from cases import make_cfg
import dsnerf_amd
Rt = 8192
sel = np.linspace(0, HW * HW - 1, Rt).astype(np.int64)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cfg = make_cfg(S)
net = dsnerf_amd.DualSpaceNeRF(cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.to(dev)
rt = dsnerf_amd.Renderer(net, None, cfg, torch.from_numpy(canon), body_data={"f": faces}, device=dev)
rt.train()
tb = {"ray_o": o[sel][None].contiguous(), "ray_d": d[sel][None].contiguous(), "near": n0[sel][None].clone(), "far": f0[sel][None].clone(),
      "xyz": xyz[None], "poses": poses[None], "Th": torch.zeros(1, 1, 3, device=dev), "frame": torch.tensor([5])}
target = T(synth.hash_uniform(Rt * 3, 77).reshape(Rt, 3).astype(np.float32))


def train_forward():
    torch.manual_seed(3)
    bb = dict(tb)
    bb["near"], bb["far"] = tb["near"].clone(), tb["far"].clone()
    net.zero_grad()
    out = rt.render(bb)["coarse"]
    return torch.nn.functional.mse_loss(out["color"], target)


sa.set_frame(pk, xyz, poses, 5, False, None, None, None)
pts, z = _lib.sample(sa, o, d, n0.clone(), f0.clone(), S, tv, None, want_pts=True)
w = _lib.warp(sa, pts, d, S, want_dir=False, want_active=True)
xc, act = w["x_c"], (w["active_list"], w["active_count"])
fwd = _lib.field_forward(sa, pk, xc, active=act)
torch.cuda.synchronize()

state_ = {}


def prep_bwd():
    state_["loss"] = train_forward()


def agg_bwd():
    state_["loss"].backward()


def agg_train_fwd():
    train_forward()           # (inside `with torch.cuda.stream(A)`: the Renderer enqueues on the current stream)


def agg_fwd():
    for _ in range(2):
        _lib.field_forward(sa, pk, xc, active=act)


def agg_rev():
    for _ in range(3):
        _lib.field_reverse(sa, pk, xc, fwd[2], fwd[3], fwd[0], fwd[1])


def nothing():
    pass


# (name, preparation run alone before the overlapped region, aggressor)
AGG = (("none", nothing, nothing),
       ("training BACKWARD (k_tangent16, k_adjoint16, k_t_wgrad16d/p, k_t_lin, k_t_wgrad, ...)", prep_bwd, agg_bwd),
       ("training FORWARD (k_field16<train>, k_light16 with stores, far search)", nothing, agg_train_fwd),
       ("field16 forward x2 (guarded: control)", nothing, agg_fwd), ("field16 reverse x3 (guarded: control)", nothing, agg_rev))


if os.environ.get("RACE_AGG"):          # e.g. RACE_AGG=none,BACKWARD : only the aggressors whose name contains one of these
    AGG = tuple(a for a in AGG if any(t in a[0] for t in os.environ["RACE_AGG"].split(",")))
QUICK = os.environ.get("RACE_QUICK") == "1"      # victims (a) and (b) only


def diffs(a, b, mask=None):
    a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
    ne = a != b
    if ne.dim() > 1:
        ne = ne.any(-1)
    if mask is not None:
        ne = ne & mask
    return int(ne.sum())


def span(fn_a, fn_b):
    """fn_a on stream A and fn_b on stream B at once; returns (ms of A alone region, ms of B region) by events - says whether they overlapped"""
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    with torch.cuda.stream(A):
        ea0.record()
        fn_a()
        ea1.record()
    with torch.cuda.stream(B):
        eb0.record()
        fn_b()
        eb1.record()
    torch.cuda.synchronize()
    return ea0.elapsed_time(ea1), eb0.elapsed_time(eb1), ea0.elapsed_time(eb1)


# ---- victim (a): the shading phase, (b): the geometry phase, of NF frames back to back -------------------------------------------
for name, prep, fn in AGG:
    res_s, res_g, spans = [], [], []
    for rep in range(REPS):
        held = []
        for sc, ws in VF:
            held.append(run(sc, ws, ["set", "geom", "field"]))
        with torch.cuda.stream(A):      # (autograd runs a node's backward on the stream its forward ran on)
            prep()
        torch.cuda.synchronize()
        spans.append(span(fn, lambda: [run(sc, ws, ["shade"], ob, nfb) for (sc, ws), (ob, nfb) in zip(VF, held)]))
        tot = [0, 0]
        for sc, ws in VF:
            a = arrays(ws)
            tot[0] += diffs(a["n_w"], ref["n_w"], pos)
            tot[1] += diffs(a["colour"], ref["colour"], pos)
        res_s.append(tuple(tot))
        for sc, ws in VF:
            sc.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True)
        with torch.cuda.stream(A):      # (autograd runs a node's backward on the stream its forward ran on)
            prep()
        torch.cuda.synchronize()
        spans.append(span(fn, lambda: [run(sc, ws, ["geom"]) for sc, ws in VF]))
        tot = [0, 0]
        for sc, ws in VF:
            a = arrays(ws)
            tot[0] += diffs(a["transparent"], ref["transparent"])
            tot[1] += diffs(a["x_c"], ref["x_c"], live)
        res_g.append(tuple(tot))
    print(f"aggressor {name}\n    -> victim shade x{NF} (n_w, colour) diffs: {res_s}   victim geometry x{NF} (transparent, x_c): {res_g}\n"
          f"       (aggressor ms, victims ms, both ms) per overlapped region: {[tuple(round(x, 2) for x in s_) for s_ in spans]}", flush=True)

if QUICK:
    sys.exit(0)
# ---- victim (c): a caller's torch kernels -----------------------------------------------------------------------------------
M = 1 << 24
x1, x2, x3 = (torch.rand(M, device=dev) for _ in range(3))
ga, gb = torch.rand(2048, 2048, device=dev), torch.rand(2048, 2048, device=dev)
tab = torch.rand(1 << 20, 16, device=dev)
idx = torch.randint(0, 1 << 20, (1 << 21,), device=dev)


def torch_victims(n=6):
    outs = []
    for _ in range(n):
        e = x1 * x2 + x3
        e = e * x1 - x2
        outs.append((e, ga @ gb, tab.index_select(0, idx)))
    return outs


t_ref = torch_victims(1)[0]
torch.cuda.synchronize()
for name, prep, fn in AGG:
    res, spans = [], []
    for rep in range(REPS):
        with torch.cuda.stream(A):      # (autograd runs a node's backward on the stream its forward ran on)
            prep()
        torch.cuda.synchronize()
        got = []
        spans.append(span(fn, lambda: got.extend(torch_victims())))
        res.append(tuple(sum(diffs(g_[k], t_ref[k]) for g_ in got) for k in range(3)))
    print(f"aggressor {name}\n    -> victim torch x6 (elementwise, fp32 gemm, gather) diffs: {res}   spans {[tuple(round(x, 2) for x in s_) for s_ in spans]}", flush=True)

# ---- the training step as VICTIM of the (guarded) field kernels; only the tensors that are bit-stable from run to run count ------
def grads_now():
    train_forward().backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters()}


g0, g1 = grads_now(), grads_now()
stable = [k for k in g0 if torch.equal(g0[k], g1[k])]
print(f"training step alone, twice: {len(stable)} of {len(g0)} gradient tensors bit-identical (the others are sums with float atomics)")
for name, prep, fn in [a for a in AGG if "control" in a[0]]:
    res = []
    for rep in range(REPS):
        got = {}

        def victim():
            train_forward().backward()
        span(lambda: [fn() for _ in range(3)], victim)
        got = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        res.append(sum(int(not torch.equal(got[k], g0[k])) for k in stable))
    print(f"aggressor {name}\n    -> victim training step: bit-stable tensors that changed: {res}", flush=True)
