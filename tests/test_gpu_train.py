"""GPU parity of the training row: parameter gradients of Renderer.render through the C ABI
(dsn_render_rays_grad) against (a) gradients captured from the reference's own loss.backward()
(tests/golden/*_grads.npz) and (b) the differentiable CPU oracle (oracle/train_oracle.py)."""
import json
import os

import numpy as np
import pytest
import torch

import train_oracle as TO
from helpers import load, maxdiff, state
from test_gpu_render import make_batch, make_renderer

pytestmark = pytest.mark.gpu

GRAD_CASES = ["small_train_grads", "small_train_grads_nonoise", "full_train_grads", "small_train_grads_w2", "full_train_grads_w2",
              "small_train_grads_w4", "full_train_grads_w4",
              "full_train_grads_nu"]        # the SMPL-like body (dense caps at head / hands / feet): make_golden_grads.py --nonuniform
from cases import FULL_LIMIT, SAMPLE, reference_loss, rel, sample_index  # noqa: E402,F401


# ---- what the kernels ACHIEVE, and the bars derived from it ---------------------------------------------------
# tests/golden/achieved_grad_errors.json holds, per case and per parameter tensor, the relative L2 error measured on the MI355X
# (scripts/record_grad_errors.py; the numbers are in the file, with the commit and the box they come from).  A test passes when
# every tensor stays within HEADROOM x its recorded error (or the float32 floor below, for tensors whose error is already there):
# a regression of one order of magnitude fails - rounds 3-5 accepted 2e-3 / 5e-3 where 3e-7 ... 3e-5 were achieved.
ACHIEVED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "achieved_grad_errors.json")
HEADROOM = 10.0
FLOOR = 2e-6          # relative L2 below which two float32 summation orders differ anyway (atomics in the small head products)


def achieved(kind, case):
    if not os.path.exists(ACHIEVED):
        pytest.fail("tests/golden/achieved_grad_errors.json is missing: run scripts/record_grad_errors.py on the GPU box and commit it")
    with open(ACHIEVED) as f:
        rec = json.load(f)
    if case not in rec.get(kind, {}):
        pytest.fail(f"no recorded errors for {kind} / {case}: run scripts/record_grad_errors.py on the GPU box and commit the file")
    return rec[kind][case]


def bar(recorded):
    return max(HEADROOM * float(recorded), FLOOR)


def reference_case_errors(name):
    """one training forward + backward of a golden gradient case -> (loss, reference loss, {tensor: rel. L2 vs the reference's float32
    autograd on the stored elements}, {tensor: |norm - reference norm| / reference norm}, {tensor: the reference's own f32-vs-f64 spread})"""
    g = load(name)
    r = make_renderer(g, name)
    r.cfg.MODEL.raw_noise_std = float(g["raw_noise_std"])
    r.train()
    torch.manual_seed(int(g["seed"]))
    out = r.render(make_batch(g))["coarse"]
    assert np.array_equal(out["z_vals"].cpu().numpy(), g["render:z_vals"])
    loss = reference_loss(out, torch.from_numpy(g["target_rgb"]).cuda(), torch.from_numpy(g["occupancy"]).cuda())
    r.net.zero_grad()
    loss.backward()
    err, nerr, spread = {}, {}, {}
    for k, p in r.net.named_parameters():
        assert p.grad is not None, k
        full = p.grad.detach().cpu().numpy().reshape(-1)
        a = full if full.size <= FULL_LIMIT else full[sample_index(full.size)]
        b32, b64 = g["grad:" + k], g["grad:" + k + "_f64"]
        err[k] = rel(a, b32)
        spread[k] = rel(b32, b64)
        nerr[k] = abs(float(np.linalg.norm(full.astype(np.float64))) - float(g["norm:" + k])) / max(float(g["norm:" + k]), 1e-30)
    assert r.range_overflow_count() == 0
    return float(loss), float(g["loss"]), err, nerr, spread


@pytest.mark.parametrize("name", GRAD_CASES)
def test_backward_matches_reference_autograd(name):
    loss, ref, err, nerr, spread = reference_case_errors(name)
    assert abs(loss - ref) < 2e-6 * max(1.0, abs(ref)), (loss, ref)
    rec = achieved("reference", name)
    for k in err:
        # within 10 x the error recorded for this tensor (3e-7 ... 3e-5; the reference's own float32 and float64 runs differ by
        # 1e-3 ... 4e-2 per tensor - ReLU-kink flips, PE x512 - which is how far ANY float32 implementation is from the truth)
        assert err[k] <= bar(rec[k]), (k, err[k], rec[k])
        assert err[k] < max(2e-3, 2.0 * spread[k]), (k, err[k], spread[k])        # (and never beyond rounds 3-5's bound)
        assert nerr[k] <= bar(rec[k]), (k, nerr[k], rec[k])
    print({k: "%.1e" % v for k, v in err.items()})


ORACLE_CASES = [("small_train_grads", None, None), ("full_train_grads", None, None), ("full_train_grads", 37, None),
                ("full_train_grads", 37, 21), ("small_train_grads_w2", None, None), ("full_train_grads_w2", None, None),
                ("small_train_grads_w4", None, None), ("full_train_grads_w4", None, None), ("full_train_grads_nu", None, None)]


def oracle_case_errors(name, nrays, nsamp):
    """dsn_render_rays_grad with cotangents on every output (colour, disp, acc, depth, weights) against autograd of the CPU oracle on
    the same inputs, full tensors -> {tensor: rel. L2}"""
    from dsnerf_amd import _lib
    g = dict(load(name).items())
    if nrays is not None:                      # a ray count that is no multiple of the 32-point wave tiles / 128-point blocks
        for k in ("ray_o", "ray_d", "near", "far", "render:z_vals", "noise", "jitter"):
            g[k] = g[k][:nrays]
    if nsamp is not None:                      # 37 x 21 = 777 samples: not a multiple of 16 (the weight-gradient kernels' ragged tail)
        for k in ("render:z_vals", "noise", "jitter"):
            g[k] = np.ascontiguousarray(g[k][:, :nsamp])
    sd = state(name)
    r = make_renderer(g, name)
    z = g["render:z_vals"]
    R, S = z.shape
    noise = g["noise"] if float(g["raw_noise_std"]) > 0 else None
    rng = np.random.default_rng(4)
    cot = {k: rng.standard_normal(s).astype(np.float32) for k, s in
           (("color", (R, 3)), ("disp_map", (R,)), ("acc_map", (R,)), ("depth_map", (R,)), ("weights", (R, S)))}
    # oracle: L = sum(cotangent * output); disparity only where the ray hit something (NaN elsewhere, as the reference)
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd.items()}
    out = TO.render(params, g, jitter_z=z, noise=noise)
    ok = out["acc_map"].detach() > 1e-3
    cot["disp_map"] = np.where(ok.numpy(), cot["disp_map"] * 1e-2, 0.0).astype(np.float32)
    L = sum((torch.from_numpy(cot[k]) * out[k]).sum() for k in ("color", "acc_map", "depth_map", "weights"))
    L = L + (torch.from_numpy(cot["disp_map"])[ok] * out["disp_map"][ok]).sum()
    L.backward()
    dev = r.device
    b = make_batch(g)
    r._set_frame(b)
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    grads = _lib.render_rays_grad(r.scene, {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, T(g["poses"]),
                                  int(g["frame"]), False, T(g["ray_o"]), T(g["ray_d"]), T(z), T(noise), T(cot["color"]),
                                  T(cot["disp_map"]), T(cot["acc_map"]), T(cot["depth_map"]), T(cot["weights"]))
    err = {}
    for k, gr in zip(_lib.PARAM_ORDER, grads):
        want = params[k].grad.numpy() if params[k].grad is not None else np.zeros_like(sd[k])
        err[k] = rel(gr.cpu().numpy(), want)
    return err


def oracle_case_key(name, nrays, nsamp):
    return f"{name}|{nrays}|{nsamp}"


@pytest.mark.parametrize("name,nrays,nsamp", ORACLE_CASES)
def test_backward_matches_oracle_all_cotangents(name, nrays, nsamp):
    err = oracle_case_errors(name, nrays, nsamp)
    rec = achieved("oracle", oracle_case_key(name, nrays, nsamp))
    for k, e in err.items():
        assert e <= bar(rec[k]), (k, e, rec[k])
        assert e < 5e-3, (k, e)


def config2_batch(R=8192, S=64, hw=512):
    """BASELINE configs[2]'s own batch, as tests/golden/make_golden_grads.py --config2 builds it for the reference: 8192 rays spread
    over the 512 x 512 view of the synthetic body, hash-generated targets / occupancy (nothing but seeds is stored)"""
    from dsnerf_amd import synth
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(hw, hw, xyz, fit_box=True)
    sel = np.linspace(0, hw * hw - 1, R).astype(np.int64)
    g = {"canonical_vertex": canon, "faces": faces.astype(np.int32), "xyz": xyz, "poses": synth.make_poses(),
         "Th": np.asarray((0.2, -0.1, 1.0), np.float32), "frame": np.int64(5), "S": np.int64(S),
         "ray_o": rays["ray_o"][sel], "ray_d": rays["ray_d"][sel], "near": rays["near"][sel], "far": rays["far"][sel],
         "target_rgb": synth.hash_uniform(R * 3, 91).reshape(R, 3).astype(np.float32),
         "occupancy": (synth.hash_uniform(R, 92) > 0.5).astype(np.float32)}
    return g


def config2_errors(name="full_train_grads_8192"):
    g = load(name)
    R, S = int(g["rays"]), int(g["S"])
    b = config2_batch(R, S)
    r = make_renderer(b, name)
    r.cfg.MODEL.raw_noise_std = float(g["raw_noise_std"])
    r.train()
    torch.manual_seed(int(g["seed"]))
    out = r.render(make_batch(b))["coarse"]
    fwd = {k: maxdiff(out[k].detach().cpu().numpy(), g["render:" + k]) for k in ("color", "acc_map", "depth_map")}
    per_ray = np.abs(out["color"].detach().cpu().numpy().astype(np.float64) - g["render:color"]).max(1)
    fwd["color_rays_above_1e-4"] = int((per_ray > 1e-4).sum())
    fwd["color_p99.9"] = float(np.quantile(per_ray, 0.999))
    fwd["color_median"] = float(np.median(per_ray))
    zsum = float(out["z_vals"].double().sum())
    loss = reference_loss(out, torch.from_numpy(b["target_rgb"]).cuda(), torch.from_numpy(b["occupancy"]).cuda())
    r.net.zero_grad()
    loss.backward()
    err, nerr = {}, {}
    for k, p in r.net.named_parameters():
        full = p.grad.detach().cpu().numpy().reshape(-1)
        a = full if full.size <= FULL_LIMIT else full[sample_index(full.size)]
        err[k] = rel(a, g["grad:" + k])
        nerr[k] = abs(float(np.linalg.norm(full.astype(np.float64))) - float(g["norm:" + k])) / max(float(g["norm:" + k]), 1e-30)
    assert r.range_overflow_count() == 0
    return float(loss), float(g["loss"]), fwd, (zsum, float(g["z_vals_sum"])), err, nerr


@pytest.mark.parametrize("name", ["full_train_grads_8192", "full_train_grads_8192_w4"])
def test_full_batch_matches_the_reference_at_8192x64(name):
    """(default = hash-random parameters; _w4 = the CONVERGED checkpoint, the state late in training: most rows have alpha = 0.)
    BASELINE configs[2] AT ITS OWN SIZE against the real reference (rounds 3-5 pinned the training step by
    the reference at 64 / 128 rays and checked 8192 x 64 through additivity / linearity only).  The fixture is the reference's
    float32 loss.backward() on the 8192-ray batch, run in the build container (tests/golden/make_golden_grads.py --config2: loss,
    per-ray outputs, 33 norms and sub-sampled gradients; its float64 twin does not fit the container's memory)."""
    loss, ref, fwd, (zsum, zref), err, nerr = config2_errors(name)
    assert abs(zsum - zref) <= 1e-9 * abs(zref), (zsum, zref)             # the sampler (jitter included) is bit-exact: same sum
    # Per-ray colours: the bar (1e-4 absolute) on all but a handful of the 8192 rays.  Measured (scripts/dbg/config2_outliers.py,
    # profiles/r06_config2_outliers.txt): median 7e-8, 99.9 % of the rays within 2.3e-5, THREE rays above 1e-4 (6.4e-4, 3.4e-4, 2.7e-4).
    # On two of those three the float32 CPU oracle - a third implementation - agrees with this library and differs from the reference by
    # the same 3.4e-4 / 2.7e-4; on the third it agrees with the reference: single ill-conditioned samples (the normal map's difference
    # of two projections, model/spacenet.py:278-298) on which float32 implementations scatter, the tail SURVEY 8c(4) measured on the
    # reference itself (float32 vs float64 rgb: p99 1.1e-4, max 1.6e-3).  128-ray fixtures never met it.  So: at most 0.1 % of the
    # rays above the bar, none beyond the reference's own float32-vs-float64 maximum, and the loss - the mean over all rays - to 2e-6.
    assert fwd["color_rays_above_1e-4"] <= 8 and fwd["color"] < 1.6e-3 and fwd["color_p99.9"] < 1e-4 and fwd["color_median"] < 1e-6, fwd
    assert fwd["acc_map"] < 1e-4 and fwd["depth_map"] < 5e-4, fwd
    assert abs(loss - ref) < 2e-6 * max(1.0, abs(ref)), (loss, ref)
    rec = achieved("reference", name)
    for k in err:
        assert err[k] <= bar(rec[k]), (k, err[k], rec[k])
        assert nerr[k] <= bar(rec[k]), (k, nerr[k], rec[k])
    print({k: "%.1e" % v for k, v in err.items()})


def test_training_steps_reduce_the_loss():
    """trainer.py:66-81 in miniature: Adam on the mirror's parameters through render() -> loss -> backward()."""
    g = load("small_train_grads")
    r = make_renderer(g)
    r.train()
    target = torch.from_numpy(g["target_rgb"]).cuda()
    opt = torch.optim.Adam(r.net.parameters(), lr=5e-4)
    torch.manual_seed(0)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        out = r.render(make_batch(g))["coarse"]
        loss = torch.nn.functional.mse_loss(out["color"], target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < 0.9 * np.mean(losses[:3]), losses


def test_full_batch_gradients_are_additive_and_linear():
    """BASELINE configs[2] size (8192 rays x 64 samples, dense): the gradient of a batch is the sum of the gradients of its
    two halves, and doubling every cotangent doubles it - size-independent properties that exercise the large-N code paths
    (one workgroup per CU in the weight-gradient kernels, LDS-DMA rings, batch-wide magnitudes) against the small-N ones"""
    import dsnerf_amd
    from dsnerf_amd import _lib, synth
    dev = torch.device("cuda:0")
    R, S = 8192, 64
    canon, faces = synth.make_body()
    sd = synth.make_state_dict()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(512, 512, xyz, fit_box=True)
    sel = np.linspace(0, 512 * 512 - 1, R).astype(np.int64)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    params = {k: T(v) for k, v in sd.items()}
    packed = _lib.PackedParams(dev).update(params)
    sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    poses = T(synth.make_poses())
    sc.set_frame(packed, T(xyz), poses, 5)
    o, d = T(rays["ray_o"][sel]), T(rays["ray_d"][sel])
    near, far = T(rays["near"][sel]), T(rays["far"][sel])
    jit = T(synth.hash_uniform(R * S, 21).reshape(R, S).astype(np.float32))
    _, z = _lib.sample(sc, o, d, near, far, S, torch.linspace(0.0, 1.0, steps=S).to(dev), jit, want_pts=False)
    noise = T((synth.hash_uniform(R * S, 22).reshape(R, S).astype(np.float32) - 0.5) * 2.0)
    d_rgb = T(synth.hash_uniform(R * 3, 23).reshape(R, 3).astype(np.float32) - 0.5)
    d_acc = T(synth.hash_uniform(R, 24).astype(np.float32) - 0.5)

    def grads(lo, hi, scale=1.0):
        g = _lib.render_rays_grad(sc, params, poses, 5, False, o[lo:hi].contiguous(), d[lo:hi].contiguous(), z[lo:hi].contiguous(),
                                  noise[lo:hi].contiguous(), (d_rgb[lo:hi] * scale).contiguous(), None, (d_acc[lo:hi] * scale).contiguous())
        return [x.double().cpu().numpy() for x in g]

    whole, a, b, twice = grads(0, R), grads(0, R // 2), grads(R // 2, R), grads(0, R, 2.0)
    for k, w, x, y, t in zip(_lib.PARAM_ORDER, whole, a, b, twice):
        n = max(np.linalg.norm(w), 1e-30)
        assert np.linalg.norm(w - (x + y)) / n < 2e-5, (k, np.linalg.norm(w - (x + y)) / n)
        assert np.linalg.norm(t - 2.0 * w) / n < 2e-5, (k, np.linalg.norm(t - 2.0 * w) / n)


def test_underflowing_cotangents_leave_finite_gradients():
    """Regression (round 3, found by training w4): rays whose cotangent has underflowed to ~1e-38 gave some samples a denormal
    u = dL/d(grad sigma); k_tangent16 divided by max|u| (1 / 1e-39 = inf) and the second-order weight gradients of the whole
    batch came out NaN.  Such samples now count as u = 0: the gradients are finite and equal those of the batch with these
    rays' cotangents set to exactly zero."""
    from dsnerf_amd import _lib
    g = dict(load("full_train_grads").items())
    sd = state("full_train_grads")
    r = make_renderer(g, "full_train_grads")
    dev = r.device
    z = g["render:z_vals"]
    R, S = z.shape
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r._set_frame(make_batch(g))
    rng = np.random.default_rng(7)
    d_rgb = rng.standard_normal((R, 3)).astype(np.float32)
    tiny = d_rgb.copy()
    tiny[::2] *= np.float32(1e-36)                       # every other ray: a cotangent at the edge of float32
    zero = d_rgb.copy()
    zero[::2] = 0.0
    params = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    ws = _lib.GradWorkspace(dev)

    def run(c):
        out = _lib.render_rays_grad(r.scene, params, T(g["poses"]), int(g["frame"]), False, T(g["ray_o"]), T(g["ray_d"]), T(z),
                                    T(g["noise"]), T(c), ws=ws)
        return [x.double().cpu().numpy() for x in out], _lib.grad_range_overflow(ws, R, S)

    a, ovf_a = run(tiny)
    b, ovf_b = run(zero)
    assert ovf_a == 0 and ovf_b == 0
    for k, x, y in zip(_lib.PARAM_ORDER, a, b):
        assert np.isfinite(x).all(), k
        assert np.linalg.norm(x - y) <= 1e-6 * max(np.linalg.norm(y), 1e-30), (k, np.linalg.norm(x - y), np.linalg.norm(y))
