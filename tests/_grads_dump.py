"""Helper of test_gpu_round5.py::test_paired_weight_gradient_launches...: one training forward + backward of a golden case in THIS process
(whose environment the test sets), every parameter gradient into an .npz.     python tests/_grads_dump.py CASE OUT.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for q in (os.path.join(os.path.dirname(HERE), "oracle"), os.path.dirname(HERE), HERE):      # (what tests/conftest.py puts on the path)
    sys.path.insert(0, q)
from helpers import load                                          # noqa: E402
from test_gpu_render import make_batch, make_renderer            # noqa: E402
from test_gpu_train import reference_loss                         # noqa: E402

name, out_path = sys.argv[1], sys.argv[2]
g = load(name)
r = make_renderer(g, name)
r.cfg.MODEL.raw_noise_std = float(g["raw_noise_std"])
r.train()
torch.manual_seed(int(g["seed"]))
out = r.render(make_batch(g))["coarse"]
loss = reference_loss(out, torch.from_numpy(g["target_rgb"]).cuda(), torch.from_numpy(g["occupancy"]).cuda())
r.net.zero_grad()
loss.backward()
np.savez(out_path, **{k: p.grad.detach().cpu().numpy() for k, p in r.net.named_parameters()})
