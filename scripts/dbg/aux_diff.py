"""which gradient tensors differ between the one-stream and the two-stream backward (DSN_TRAIN_AUX)?"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
from cases import load, state, make_renderer, make_batch
from dsnerf_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "full_train_grads"
g = dict(load(name).items()); sd = state(name)
r = make_renderer(g, name); dev = r.device
z = g["render:z_vals"]; R, S = z.shape
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
r._set_frame(make_batch(g))
rng = np.random.default_rng(4)
d_rgb = rng.standard_normal((R, 3)).astype(np.float32)
params = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
ws = _lib.GradWorkspace(dev)
def run():
    out = _lib.render_rays_grad(r.scene, params, T(g["poses"]), int(g["frame"]), False, T(g["ray_o"]), T(g["ray_d"]), T(z), T(g["noise"]), T(d_rgb), ws=ws)
    torch.cuda.synchronize()
    return [x.double().cpu().numpy() for x in out]
os.environ["DSN_TRAIN_AUX"] = "0"; a = run(); a2 = run()
os.environ["DSN_TRAIN_AUX"] = "1"; b = run(); b2 = run()
for k, x, x2, y, y2 in zip(_lib.PARAM_ORDER, a, a2, b, b2):
    n = max(np.linalg.norm(x), 1e-30)
    print("%-40s one-stream twice %.1e   two-stream vs one %.1e   two-stream twice %.1e" % (k, np.linalg.norm(x - x2) / n, np.linalg.norm(x - y) / n, np.linalg.norm(y - y2) / n))
