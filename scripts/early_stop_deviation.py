import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch, numpy as np
from tests.test_gpu_round2 import _stop_pair
from helpers import state
for nm in ("x_w3", "x_w2"):
    run = _stop_pair(state(nm), hw=256, screen=False)
    ref, st0, _ = run(stop_stats=True)
    got, st1, _ = run(early_stop=True)
    print(nm, "skipped", st1["skipped"] / st1["active"], "max|colour|", float(ref["color"].abs().max()),
          "d colour", float((ref["color"] - got["color"]).abs().max()), "d acc", float((ref["acc_map"] - got["acc_map"]).abs().max()),
          "d depth", float((ref["depth_map"] - got["depth_map"]).abs().max()), "d weights", float((ref["weights"] - got["weights"]).abs().max()))
