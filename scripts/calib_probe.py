"""density-screen calibration statistics against the size of the calibration shell (DSN_CALIB_BOX) -> lines of JSON"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
from helpers import state
from test_gpu_round2 import full_frame, renderer_with
for wname in ("x_w4", "x", "x_w3"):
    canon, faces, batch = full_frame(hw=64)
    r = renderer_with(state(wname) if wname != "x" else state(), canon, faces)
    r.eval(); r._set_frame(batch)
    pk = r.net.packed(r.device)
    for box in ("0.3", "0.2", "0.12", "0.08", "0.04"):
        os.environ["DSN_CALIB_BOX"] = box
        i = pk.calibrate_screen(r.scene)
        print(wname, "box", box, {k: (round(i[k], 5) if isinstance(i[k], float) else i[k]) for k in ("deviation", "margin_statistic", "margin", "dropped_fraction")}, flush=True)
