"""do the library's kernels write past the end (or before the start) of a caller-owned buffer?  every _scratch allocation gets guard bands"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
from dsnerf_amd import _lib
G = 64 << 20
reg = []
def guarded(nbytes, device):
    t = torch.empty(int(nbytes) + 2 * G, dtype=torch.uint8, device=device)
    t[:G] = 0xAB; t[G + int(nbytes):] = 0xAB
    reg.append((int(nbytes), t))
    return t[G:G + int(nbytes)]
_lib._scratch = guarded
HW = int(os.environ.get("DBG_HW", "256"))
canon, faces, batch = full_frame(hw=HW)
sd = state(os.environ.get("DBG_W", "x_w4"))
r = renderer_with(sd, canon, faces, density_screen=False)
r.eval()
if os.environ.get("DBG_STOP") != "1": r.early_stop = False
def fresh():
    b = dict(batch); b["near"], b["far"] = batch["near"].clone(), batch["far"].clone(); return b
for i in range(3):
    r.render_view(fresh())
imgs = r.render_views([fresh() for _ in range(6)], frames_in_flight=3, device_output=True)
torch.cuda.synchronize()
for n, t in reg:
    lo, hi = t[:G], t[G + n:]
    bl, bh = (lo != 0xAB).nonzero().flatten(), (hi != 0xAB).nonzero().flatten()
    print("alloc", n, "before:", int(bl.numel()), (int(bl.min()) - G, int(bl.max()) - G) if bl.numel() else "", "after:", int(bh.numel()), (int(bh.min()), int(bh.max())) if bh.numel() else "")
