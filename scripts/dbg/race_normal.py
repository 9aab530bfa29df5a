"""victim = dsn_shade (k_normal + k_light16, NOT in place) on stage buffers; aggressor = k_field16<forward> on another stream"""
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
sa = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
sb = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
for s_ in (sa, sb): s_.set_frame(pk, xyz, poses, 5, False, None, None, None)
pts, z = _lib.sample(sa, o, d, n0.clone(), f0.clone(), S, tv, None, want_pts=True)
w = _lib.warp(sa, pts, d, S, want_dir=False, want_active=True)
xc, act = w["x_c"], (w["active_list"], w["active_count"])
sig, ess, rec, posl = _lib.field_forward(sa, pk, xc, active=act)
grad = _lib.field_reverse(sb, pk, xc, rec, posl, sig, ess)
grad = grad[0] if isinstance(grad, tuple) else grad
torch.cuda.synchronize()
npos = int(posl[1][0]); pos_idx = posl[0][:npos].long()
print("pos samples", npos)
ref = _lib.shade(sb, pk, xc, grad, pts, d, ess, S, active=posl)
torch.cuda.synchronize()
import ctypes as C
T = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "ubench", "normal_tail.so"))
fidx = ref[0]
names = ["u", "v", "h", "s0", "s1", "s2", "pe0", "pe1", "pe2", "u2", "v2", "h2", "e0", "e1", "e2", "df0", "df1", "df2", "n0", "n1", "n2", "fc.inv", "fw.inv", "acc"]
def tail(spin):
    out = torch.zeros(npos, 24, device=dev)
    assert T.launch_tail(_lib._ptr(sb.buf), sb.V, sb.F, _lib._ptr(xc), _lib._ptr(grad), _lib._ptr(posl[0]), _lib._ptr(posl[1]), _lib._ptr(fidx), _lib._ptr(out),
                         npos, spin, _lib._stream()) == 0
    return out
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for spin in (0, 1):
    rt = tail(spin); torch.cuda.synchronize()
    print("tail alone == k_normal's n_w:", bool(torch.equal(rt[:, 18:21], ref[1][pos_idx])))
    for rep in range(3):
        with torch.cuda.stream(A): _lib.field_forward(sa, pk, xc, active=act)
        with torch.cuda.stream(B): gt = tail(spin)
        torch.cuda.synchronize()
        bad = (torch.nan_to_num(gt, nan=-7.) != torch.nan_to_num(rt, nan=-7.))
        print("spin", spin, "rep", rep, "samples with any difference", int(bad.any(1).sum()), {names[j]: int(bad[:, j].sum()) for j in range(24) if bad[:, j].any()})
        if bad.any():
            k = int(bad.any(1).nonzero().flatten()[0])
            print("   slot", k, "got", [float(x) for x in gt[k]]); print("   ref   ", [float(x) for x in rt[k]])
D = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "ubench", "lds_dma_spin.so"))
srcbuf = torch.rand(256 * 4 * 16384 // 4 + 4096, device=dev)
dummy = torch.zeros(256, device=dev)
rt = tail(1); torch.cuda.synchronize()
for kind, name in ((0, "ordinary loads"), (1, "global_load_lds_dword"), (4, "global_load_lds_dwordx4")):
    res = []
    for rep in range(3):
        with torch.cuda.stream(A): assert D.launch_dma(kind, 256, 20000, _lib._ptr(srcbuf), _lib._ptr(dummy), _lib._stream()) == 0
        with torch.cuda.stream(B): gt = tail(1)
        torch.cuda.synchronize()
        res.append(int((torch.nan_to_num(gt, nan=-7.) != torch.nan_to_num(rt, nan=-7.)).any(1).sum()))
    print("aggressor: a loop of", name, "-> samples of the k_normal tail that differ:", res)
# the shading phase's own kernels as aggressors (after the field kernels were given their SIMDs)
nrm, colr = ref[1], ref[2]
vd = d[:, None, :].expand(-1, S, -1).reshape(-1, 3).contiguous()
def agg_light():
    for _ in range(8): _lib.light(pk, nrm, pts.reshape(-1, 3), vd, ess)
def agg_light32():
    for _ in range(2): _lib.light(pk, nrm, pts.reshape(-1, 3), vd, ess, fp32=True)
def agg_comp():
    zz = z.reshape(-1, S)
    for _ in range(8): _lib.composite(colr.reshape(-1, S, 3), sig.reshape(-1, S), None, zz, d)
def agg_tail():
    for _ in range(8): tail(1)
for name, fn in (("k_light16", agg_light), ("k_light (fp32)", agg_light32), ("k_composite", agg_comp), ("k_normal tail", agg_tail)):
    res = []
    for rep in range(4):
        with torch.cuda.stream(A): fn()
        with torch.cuda.stream(B): gt = tail(1)
        torch.cuda.synchronize()
        res.append(int((torch.nan_to_num(gt, nan=-7.) != torch.nan_to_num(rt, nan=-7.)).any(1).sum()))
    print("aggressor:", name, "-> samples of the k_normal tail that differ:", res)
