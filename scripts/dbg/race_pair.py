"""aggressor / victim: which phase of a frame on stream A corrupts which phase of a frame on stream B?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib
HW = int(os.environ.get("DBG_HW", "256"))
canon, faces, batch = full_frame(hw=HW)
sd = state(os.environ.get("DBG_W", "x_w4"))
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
    b = ws.buf; p = 8192 + al(4 * N)
    out = {}
    out["transparent"] = b[p:p + N].clone(); p += al(N)
    out["z"] = b[p:p + 4 * N].view(torch.float32).clone(); p += al(4 * N)
    out["x_c"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += al(12 * N)
    out["sigma"] = b[p:p + 4 * N].view(torch.float32).clone(); p += al(4 * N)
    out["n_w"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += 12 * N
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
# reference, alone
run(sb, wb, ["set", "geom", "field", "shade"]); torch.cuda.synchronize()
ref = arrays(wb)
nt = ref["transparent"] == 0
pos = nt & (ref["sigma"] > 0)
def diff(a):
    return {"transparent": int((a["transparent"] != ref["transparent"]).sum()), "x_c": int(((a["x_c"] != ref["x_c"]).any(-1) & nt).sum()),
            "sigma": int(((torch.nan_to_num(a["sigma"], nan=-7.) != torch.nan_to_num(ref["sigma"], nan=-7.)) & nt).sum()),
            "n_w": int((torch.nan_to_num(a["n_w"][pos], nan=-7.) != torch.nan_to_num(ref["n_w"][pos], nan=-7.)).any(-1).sum()),
            "colour": int((torch.nan_to_num(a["colour"][pos], nan=-7.) != torch.nan_to_num(ref["colour"][pos], nan=-7.)).any(-1).sum())}
run(sa, wa, ["set", "geom", "field", "shade"]); torch.cuda.synchronize()
print("A alone", diff(arrays(wa)))
for agg, vic in [(["field"], ["geom"]), (["geom"], ["geom"]), (["shade"], ["geom"]), (["field"], ["shade"]), (["geom"], ["shade"]), (["shade"], ["shade"]),
                 (["field"], ["field"]), (["geom"], ["field"]), (["set"], ["geom"]), (["set", "geom"], ["set", "geom"])]:
    res = []
    for rep in range(4):
        # both frames brought to the state in front of the phases under test, alone
        pre = lambda ph: {"set": [], "geom": ["set"], "field": ["set", "geom"], "shade": ["set", "geom", "field"]}[ph[0]]
        with torch.cuda.stream(A): oa, nfa = run(sa, wa, pre(agg))
        torch.cuda.synchronize()
        with torch.cuda.stream(B): ob, nfb = run(sb, wb, pre(vic))
        torch.cuda.synchronize()
        with torch.cuda.stream(A): run(sa, wa, agg * (3 if agg != ["field"] else 1), oa, nfa) if "set" not in agg and "geom" not in agg else run(sa, wa, agg, oa, nfa)
        with torch.cuda.stream(B): run(sb, wb, vic, ob, nfb)
        torch.cuda.synchronize()
        # finish the victim alone and compare everything
        rest = {"set": ["geom", "field", "shade"], "geom": ["field", "shade"], "field": ["shade"], "shade": []}[vic[-1]]
        with torch.cuda.stream(B): run(sb, wb, rest, ob, nfb)
        torch.cuda.synchronize()
        dd = diff(arrays(wb))
        res.append(sum(dd.values()))
        last = dd
    print("aggressor", agg, "victim", vic, "total diffs per rep", res, "last", last)
