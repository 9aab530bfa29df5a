"""which stage of a frame goes wrong when frames overlap: per-sample arrays of every slot's workspace against a frame rendered alone"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
HW = int(os.environ.get("DBG_HW", "256"))
canon, faces, batch = full_frame(hw=HW)
sd = state(os.environ.get("DBG_W", "x_w4"))
mk = lambda: renderer_with(sd, canon, faces, density_screen=False)
r1, r2 = mk(), mk()
for r in (r1, r2):
    r.eval(); r.early_stop = False
    if hasattr(r, "lazy_lists"): r.lazy_lists = not os.environ.get("DBG_NOLAZY")
if os.environ.get("DBG_KEEP"):
    import dsnerf_amd
    keep = []
    orig = dsnerf_amd.Renderer._dev
    def _dev(self, t, dtype=torch.float32):
        o = orig(self, t, dtype); keep.append(o); return o
    dsnerf_amd.Renderer._dev = _dev
if os.environ.get("DBG_PRESTAGE"):
    # batches already on the device: no H2D copies inside the frames
    batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ("frame", "img") else v) for k, v in batch.items()}
def fresh():
    b = dict(batch); b["near"], b["far"] = batch["near"].clone(), batch["far"].clone(); return b
N = HW * HW * 64
al = lambda n: (n + 255) // 256 * 256
def arrays(ws):
    b = ws.buf; p = 8192
    out = {}
    out["active_cnt"] = int(b[:4].view(torch.int32)[0]); out["pos_cnt"] = int(b[64:68].view(torch.int32)[0])
    active = b[p:p + 4 * N].view(torch.int32); p += al(4 * N)
    out["transparent"] = b[p:p + N].clone(); p += al(N)
    out["z"] = b[p:p + 4 * N].view(torch.float32).clone(); p += al(4 * N)
    out["x_c"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += al(12 * N)
    out["sigma"] = b[p:p + 4 * N].view(torch.float32).clone(); p += al(4 * N)
    out["n_w"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += 12 * N
    out["colour"] = b[p:p + 12 * N].view(torch.float32).reshape(N, 3).clone(); p += al(12 * N + 256)
    return out
ref_img = r2.render_view(fresh(), device_output=True)
torch.cuda.synchronize()
ref = arrays(r2._ws)
nt = ref["transparent"] == 0
pos = nt & (ref["sigma"] > 0)
print("ref: active", ref["active_cnt"], "pos", ref["pos_cnt"], "non-transparent", int(nt.sum()), "sigma>0", int(pos.sum()))
n = int(os.environ.get("DBG_N", "3"))
for rep in range(int(os.environ.get("DBG_REPS", "3"))):
  imgs = r1.render_views([fresh() for _ in range(n)], frames_in_flight=n, device_output=True)
  torch.cuda.synchronize()
  for j, sl in enumerate(r1._slots[:n]):
    a = arrays(sl.ws)
    d = {}
    d["transparent"] = int((a["transparent"] != ref["transparent"]).sum())
    d["z"] = int((a["z"] != ref["z"]).sum())
    d["x_c(nt)"] = int((a["x_c"][nt] != ref["x_c"][nt]).any(-1).sum())
    sg = torch.nan_to_num(a["sigma"], nan=-7.0) != torch.nan_to_num(ref["sigma"], nan=-7.0)
    d["sigma"] = int(sg.sum())
    d["n_w(pos)"] = int((torch.nan_to_num(a["n_w"][pos], nan=-7.0) != torch.nan_to_num(ref["n_w"][pos], nan=-7.0)).any(-1).sum())
    d["colour(pos)"] = int((torch.nan_to_num(a["colour"][pos], nan=-7.0) != torch.nan_to_num(ref["colour"][pos], nan=-7.0)).any(-1).sum())
    px = int((torch.nan_to_num(imgs[j]["coarse_color"], nan=-1) != torch.nan_to_num(ref_img["coarse_color"], nan=-1)).any(-1).sum())
    print("rep", rep, "slot", j, d, "pixels", px)
# where in the image do the wrong samples of the last repetition lie?
for j, sl in enumerate(r1._slots[:n]):
    a = arrays(sl.ws)
    bad = ((a["x_c"] != ref["x_c"]).any(-1) & nt) | (a["transparent"] != ref["transparent"])
    if not bad.any():
        continue
    idx = bad.nonzero().flatten()
    ray = idx // 64
    rows, cols = ray // HW, ray % HW
    print("slot", j, "bad samples", idx.numel(), "rays", torch.unique(ray).numel(), "rows", int(rows.min()), "-", int(rows.max()), "cols", int(cols.min()), "-", int(cols.max()))
    hist = torch.bincount(rows // 8, minlength=HW // 8)
    print("   per 8-row band:", hist.tolist())
    # per-sample displacement of x_c
    dx = (a["x_c"][idx] - ref["x_c"][idx]).norm(dim=-1)
    print("   |dx_c| min/median/max", float(dx.min()), float(dx.median()), float(dx.max()))
    # runs of consecutive bad sample ids
    dif = idx[1:] - idx[:-1]
    print("   consecutive-id pairs", int((dif == 1).sum()), "of", idx.numel())
