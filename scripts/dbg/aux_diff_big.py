"""one-stream vs two-stream backward at 8192 x 64 (direct C-ABI call, forward recomputed inside)"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
from dsnerf_amd import _lib, synth
dev = torch.device("cuda:0")
R, S = 8192, 64
canon, faces = synth.make_body(); sd = synth.make_state_dict(); xyz = synth.pose_body(canon)
rays = synth.make_rays(512, 512, xyz, fit_box=True)
sel = np.linspace(0, 512 * 512 - 1, R).astype(np.int64)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
params = {k: T(v) for k, v in sd.items()}
packed = _lib.PackedParams(dev).update(params)
sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
poses = T(synth.make_poses()); sc.set_frame(packed, T(xyz), poses, 5)
o, d = T(rays["ray_o"][sel]), T(rays["ray_d"][sel]); near, far = T(rays["near"][sel]), T(rays["far"][sel])
jit = T(synth.hash_uniform(R * S, 21).reshape(R, S).astype(np.float32))
_, z = _lib.sample(sc, o, d, near, far, S, torch.linspace(0.0, 1.0, steps=S).to(dev), jit, want_pts=False)
noise = T((synth.hash_uniform(R * S, 22).reshape(R, S).astype(np.float32) - 0.5) * 2.0)
d_rgb = T(synth.hash_uniform(R * 3, 23).reshape(R, 3).astype(np.float32) - 0.5)
ws = _lib.GradWorkspace(dev)
def run():
    g = _lib.render_rays_grad(sc, params, poses, 5, False, o, d, z, noise, d_rgb, ws=ws)
    torch.cuda.synchronize()
    return [x.double().cpu().numpy() for x in g]
os.environ["DSN_TRAIN_AUX"] = "0"; a = run(); a2 = run()
os.environ["DSN_TRAIN_AUX"] = "1"; b = run(); b2 = run()
for k, x, x2, y, y2 in zip(_lib.PARAM_ORDER, a, a2, b, b2):
    n = max(np.linalg.norm(x), 1e-30)
    print("%-40s one twice %.1e   two vs one %.1e   two twice %.1e" % (k, np.linalg.norm(x - x2) / n, np.linalg.norm(x - y) / n, np.linalg.norm(y - y2) / n))
