import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib
HW = 256
canon, faces, batch = full_frame(hw=HW)
r = renderer_with(state("x_w4"), canon, faces, density_screen=False); r.eval()
dev = r.device; S = 64
o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
n0, f0 = r._dev(batch["near"][0]), r._dev(batch["far"][0])
xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
pk = r.net.packed(dev); tv = r._t_vals(S)
sa = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
sa.set_frame(pk, xyz, poses, 5, False, None, None, None)
pts, z = _lib.sample(sa, o, d, n0.clone(), f0.clone(), S, tv, None, want_pts=True)
w = _lib.warp(sa, pts, d, S, want_dir=False, want_active=True)
xc, act = w["x_c"], (w["active_list"], w["active_count"])
V = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "ubench", "victim_ops.so"))
n, iters = 1 << 21, 200
names = ["sqrtf", "v_rcp", "div", "v_rsq", "fma", "v_exp", "v_sin", "mul"]
def victim():
    out = torch.zeros(8, n, device=dev)
    assert V.launch_victim(n, iters, _lib._ptr(out), _lib._stream()) == 0
    return out
ref = victim(); torch.cuda.synchronize()
again = victim(); torch.cuda.synchronize()
print("alone vs alone:", {k: int((again[j] != ref[j]).sum()) for j, k in enumerate(names)})
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for rep in range(3):
    with torch.cuda.stream(A): _lib.field_forward(sa, pk, xc, active=act)
    with torch.cuda.stream(B): got = victim()
    torch.cuda.synchronize()
    print("beside k_field16<forward>:", {k: int((got[j] != ref[j]).sum()) for j, k in enumerate(names)})
    bad = (got != ref).any(0).nonzero().flatten()
    if bad.numel():
        print("   first bad threads", bad[:10].tolist(), "blocks", torch.unique(bad // 256)[:10].tolist(), "waves affected", torch.unique(bad // 64).numel())
        j = int(bad[0]); print("   got", got[:, j].tolist()); print("   ref", ref[:, j].tolist())
