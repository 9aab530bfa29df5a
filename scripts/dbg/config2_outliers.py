"""configs[2]-size training forward against the reference's golden per-ray colours: where are the largest differences?"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in ("../../tests", "../..", "../../oracle"):
    sys.path.insert(0, os.path.join(HERE, p))
import numpy as np, torch
import test_gpu_train as TT
from cases import load, make_batch, make_renderer
import oracle as O, train_oracle as TO
from helpers import state
g = load("full_train_grads_8192")
R, S = int(g["rays"]), int(g["S"])
b = TT.config2_batch(R, S)
r = make_renderer(b, "full_train_grads_8192")
r.cfg.MODEL.raw_noise_std = float(g["raw_noise_std"])
r.train()
torch.manual_seed(int(g["seed"]))
out = r.render(make_batch(b))["coarse"]
c = out["color"].detach().cpu().numpy(); ref = g["render:color"]
e = np.abs(c - ref).max(1)
print("rays", R, "max", e.max(), "p99.9", np.quantile(e, 0.999), "p99", np.quantile(e, 0.99), "median", np.median(e), "count > 1e-4:", int((e > 1e-4).sum()), "> 1e-5:", int((e > 1e-5).sum()))
w = out["weights"].detach().cpu().numpy(); acc = out["acc_map"].detach().cpu().numpy()
worst = np.argsort(e)[-8:][::-1]
# the same rays through the float32 oracle's autograd restatement and its float64 twin (CPU): is the reference itself this noisy here?
torch.manual_seed(int(g["seed"]))
jit = torch.rand(1, R, S).numpy()[0]; noise = (torch.randn(R, S) * float(g["raw_noise_std"])).numpy()
z = out["z_vals"].detach().cpu().numpy()
sd = state("full_train_grads_8192")
sub = {k: (v[worst] if k in ("ray_o", "ray_d", "near", "far") else v) for k, v in b.items()}
for dt in (torch.float32, torch.float64):
    params = {k: torch.from_numpy(v.copy()).to(dt) for k, v in sd.items()}
    try:
        o = TO.render(params, sub, jitter_z=z[worst], noise=noise[worst])
        oc = o["color"].detach().numpy()
        print(str(dt), "oracle-vs-reference on the 8 worst rays:", np.abs(oc - ref[worst]).max(1))
    except Exception as ex:
        print("oracle", dt, "failed:", repr(ex)[:200])
for i in worst:
    print("ray", i, "err", e[i], "hip", c[i], "ref", ref[i], "acc", acc[i], "max weight", w[i].max(), "samples with weight > 1e-3:", int((w[i] > 1e-3).sum()))
