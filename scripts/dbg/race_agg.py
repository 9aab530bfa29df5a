"""victim = shading phase of a frame; aggressor = one kind of kernel on another stream"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib
HW = 256
canon, faces, batch = full_frame(hw=HW)
r = renderer_with(state("x_w4"), canon, faces, density_screen=False)
r.eval()
dev = r.device
S = 64; N = HW * HW * S
o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
n0, f0 = r._dev(batch["near"][0]), r._dev(batch["far"][0])
xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
pk = r.net.packed(dev); tv = r._t_vals(S)
mk = lambda: (_lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev), _lib.RenderWorkspace(dev))
(sa, wa), (sb, wb) = mk(), mk()
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
al = lambda n: (n + 255) // 256 * 256
def arrays(ws):
    b = ws.buf; p = 8192 + al(4 * N) + al(N) + al(4 * N) + al(12 * N)
    out = {"sigma": b[p:p + 4 * N].view(torch.float32).clone()}; p += al(4 * N)
    out["n_w"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += 12 * N
    out["colour"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone()
    return out
PH = {"geom": _lib.PHASE_GEOMETRY, "field": _lib.PHASE_FIELD, "shade": _lib.PHASE_SHADE}
def run(scene, ws, phases, out=None, nf=None):
    nn, ff = nf if nf is not None else (n0.clone(), f0.clone())
    for ph in phases:
        if ph == "set": scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True)
        else: out = _lib.render_rays(scene, pk, ws, o, d, nn, ff, S, tv, phases=PH[ph], out=out)
    return out, (nn, ff)
run(sb, wb, ["set", "geom", "field", "shade"]); torch.cuda.synchronize()
ref = arrays(wb)
pos = ref["sigma"] > 0
# aggressor material: a point set for the stage functions
sa.set_frame(pk, xyz, poses, 5, False, None, None, None)
pts, z = _lib.sample(sa, o, d, n0.clone(), f0.clone(), S, tv, None, want_pts=True)
w = _lib.warp(sa, pts, d, S, want_dir=False, want_active=True)
xc, act = w["x_c"], (w["active_list"], w["active_count"])
fwd = _lib.field_forward(sa, pk, xc, active=act)
torch.cuda.synchronize()
m1, m2 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16), torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev)
def agg_fwd(): _lib.field_forward(sa, pk, xc, active=act)
def agg_rev(): _lib.field_reverse(sa, pk, xc, fwd[2], fwd[3], fwd[0], fwd[1])
def agg_gemm():
    for _ in range(4): torch.mm(m1, m2)
def agg_copy():
    for _ in range(8): big.add_(1.0)
def agg_fp32(): _lib.field(sa, pk, xc, active=act, fp32=True)
import ctypes as C
def variant(path):
    L = C.CDLL(path)
    sig, ess = torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    rec = torch.empty(_lib.lib().dsn_field_record_bytes(C.c_int64(N)), dtype=torch.uint8, device=dev)
    posl, pc = torch.zeros(N, dtype=torch.int32, device=dev), torch.zeros(16, dtype=torch.int32, device=dev)
    def f():
        pc.zero_()
        rc = L.dsn_field_forward(_lib._ptr(sa.buf), sa.V, sa.F, _lib._ptr(pk.buf), _lib._ptr(xc), C.c_int64(N), _lib._ptr(act[0]), _lib._ptr(act[1]),
                                 _lib._ptr(sig), _lib._ptr(ess), _lib._ptr(rec), _lib._ptr(posl), _lib._ptr(pc), _lib._stream())
        assert rc == 0
    return f
def agg_light():
    for _ in range(6): run(sa, wa, ["shade"], oa, nfa)
def agg_none(): pass
CO = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "ubench", "coresident.so"))
dummy = torch.zeros(16, device=dev)
def spin(v, a, lds, mode, groups=256, iters=400000):
    def f():
        rc = CO.launch(v, a, lds, groups, iters, mode, _lib._ptr(dummy), _lib._stream())
        assert rc == 0, rc
    return f
srcbuf = torch.rand(256 * 256 * 4 + 64, device=dev)
def spin2(kind, iters=60000):
    def f():
        assert CO.launch2(kind, 256, iters, _lib._ptr(srcbuf), _lib._ptr(dummy), _lib._stream()) == 0
    return f
oa, nfa = run(sa, wa, ["set", "geom", "field"]); torch.cuda.synchronize()
SP = tuple((f"spin v{v} a{a} lds{l} mode{m}", spin(v, a, l, m)) for (v, a, l) in ((256, 192, 0), (256, 192, 131072), (256, 0, 0), (128, 0, 0), (128, 64, 0), (256, 64, 0), (0, 0, 0), (0, 0, 131072)) for m in (0, 1))
SP = (("spin2: global loads into HIGH AGPRs", spin2(1)), ("spin2: global loads into high VGPRs", spin2(2)), ("spin2: ds_read into AGPRs", spin2(3)))
for name, fn in SP + (("none", agg_none), ("field16 forward", agg_fwd), ("field16 reverse", agg_rev), ("torch bf16 gemm", agg_gemm), ("elementwise", agg_copy),
                 ("exact-fp32 field", agg_fp32), ("shade phase", agg_light)) + tuple(
                 (os.path.basename(v), variant(v)) for v in sorted(__import__("glob").glob(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "*.so")))):
    res = []
    for rep in range(4):
        with torch.cuda.stream(B): ob, nfb = run(sb, wb, ["set", "geom", "field"])
        torch.cuda.synchronize()
        try:
            with torch.cuda.stream(A): fn()
        except Exception as e:
            res.append("ERR " + str(e)[:60]); break
        with torch.cuda.stream(B): run(sb, wb, ["shade"], ob, nfb)
        torch.cuda.synchronize()
        a = arrays(wb)
        res.append((int((torch.nan_to_num(a["n_w"][pos], nan=-7.) != torch.nan_to_num(ref["n_w"][pos], nan=-7.)).any(-1).sum()),
                    int((torch.nan_to_num(a["colour"][pos], nan=-7.) != torch.nan_to_num(ref["colour"][pos], nan=-7.)).any(-1).sum())))
    print("aggressor", name, "-> victim shade (n_w, colour) diffs:", res)
