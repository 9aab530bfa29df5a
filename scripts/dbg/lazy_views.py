import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
canon, faces, batch = full_frame(hw=256)
sd = state("x_w4")
r1 = renderer_with(sd, canon, faces, density_screen=False)
r2 = renderer_with(sd, canon, faces, density_screen=False)
r2.lazy_lists = False
if os.environ.get("DBG_NOLAZY"): r1.lazy_lists = False
for r in (r1, r2):
    r.eval(); r.early_stop = False
def fresh():
    b = dict(batch); b["near"], b["far"] = batch["near"].clone(), batch["far"].clone(); return b
b = r2.render_view(fresh())
def cmp(tag, v):
    d = (torch.nan_to_num(v["coarse_color"], nan=-1) != torch.nan_to_num(b["coarse_color"], nan=-1)).any(-1)
    print(tag, "pixels differing:", int(d.sum()), "rows", (d.any(1).nonzero().flatten()[:5].tolist() if d.any() else []))
cmp("single 1", r1.render_view(fresh()))
cmp("single 2", r1.render_view(fresh()))
for n in (1, 2, 3):
    for i, v in enumerate(r1.render_views([fresh() for _ in range(4)], frames_in_flight=n, device_output=False)):
        cmp(f"views n={n} frame {i}", v)
for i, v in enumerate(r1.render_views([fresh() for _ in range(4)], frames_in_flight=3, device_output=True)):
    torch.cuda.synchronize(); cmp(f"views dev n=3 frame {i}", {k: x.cpu() for k, x in v.items()})
