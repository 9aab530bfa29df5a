"""The oracle (oracle/dsn_oracle.c) is PINNED here: every function is checked against the golden
vectors the real reference produced (tests/golden/make_golden.py).  Geometry stages are bit-exact."""
import numpy as np
import pytest
import torch

import oracle as O
from helpers import ALL_CASES, CASES, code_for, light_kw, load, maxdiff, per_point_dirs, ref_tol, state


@pytest.fixture(scope="module")
def params():
    cache = {}

    def get(name):
        sd = state(name)
        if id(sd) not in cache:
            cache[id(sd)] = (sd, O.Params(sd))
        return cache[id(sd)]
    return get


@pytest.mark.parametrize("name", ALL_CASES)
def test_sampler_bit_exact(name):
    g = load(name)
    S = int(g["S"])
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()   # utils/pts_utils.py:4 draws it on the host
    jit = g["jitter"][0] if "jitter" in g.files else None
    sm = O.sample_gg(g["ray_o"], g["ray_d"], g["near"], g["far"], g["xyz"], S, jit, tv)
    assert np.array_equal(sm["near"], g["near_gg"]) and np.array_equal(sm["far"], g["far_gg"])
    assert np.array_equal(sm["z_vals"], g["z_vals"])
    assert np.array_equal(sm["pts"], g["pts"])
    # the built-in linspace differs from torch's vectorised one by at most 1 ulp
    assert maxdiff(O.linspace01(S), tv) <= 1.2e-7


@pytest.mark.parametrize("name", ALL_CASES)
def test_warp_bit_exact(name):
    g = load(name)
    assert np.array_equal(O.centroids(g["xyz"], g["faces"]), g["centroid_world"])
    w = O.warp(g["pts"], per_point_dirs(g), g["xyz"], g["canonical_vertex"], g["faces"])
    assert np.array_equal(w["idx"], g["idx_world"])
    assert np.array_equal(w["uv"], g["uv"]) and np.array_equal(w["h"], g["h"])
    assert np.array_equal(w["transparent"], g["transparent"])
    assert np.array_equal(w["x_c"], g["x_c"])
    assert np.array_equal(w["ray_d_can"], g["ray_d_can"])


@pytest.mark.parametrize("name", ALL_CASES)
def test_field(name, params):
    sd, P = params(name)
    g = load(name)
    q, pf = O.pose_feat(g["poses"], P)
    assert np.array_equal(q, g["pose_quat"][0])
    assert maxdiff(pf, g["pose_feat"][0]) < 5e-7 * max(1.0, float(np.abs(g["pose_feat"]).max()))     # summation order of a 3-layer MLP
    sig, ess, gr = O.field(g["x_c"], P, code_for(g, sd, name), g["pose_feat"][0])
    assert maxdiff(sig, g["sigma"]) < ref_tol(g, "sigma", 5e-5, 2e-6)       # GEMM-order noise only (|sigma| <= 64: 5e-5)
    assert maxdiff(ess, g["essence"]) < ref_tol(g, "essence", 5e-6, 2e-6)
    # d sigma/dx is ill-conditioned (PE x512, ReLU kinks: a pre-activation within rounding of 0 flips a mask and moves the
    # gradient discretely), so it is judged per point with a small outlier budget, like the GPU kernels are
    gn = np.linalg.norm(g["grad_sigma"], axis=-1)
    rel = np.linalg.norm(gr - g["grad_sigma"], axis=-1) / np.maximum(gn, 1.0)
    assert np.median(rel) < 2e-6 and np.mean(rel > 1e-4) < 2e-3, (np.median(rel), np.mean(rel > 1e-4))
    if name in CASES:
        assert maxdiff(gr, g["grad_sigma"]) < 2e-4 * np.abs(g["grad_sigma"]).max()     # no flip at all on the default set's first fixtures


@pytest.mark.parametrize("name", ALL_CASES)
def test_normal_and_lighting(name, params):
    sd, P = params(name)
    g = load(name)
    idc, nw = O.normal_world(g["x_c"], g["grad_sigma"], g["canonical_vertex"], g["xyz"], g["faces"])
    assert np.array_equal(idc, g["idx_canon"])
    assert np.array_equal(nw, g["n_w"])
    col = O.lighting(g["n_w"], g["pts"], per_point_dirs(g), g["essence"], P, **light_kw(g))
    assert maxdiff(col, g["colour"]) < 1e-6 * max(1.0, float(np.abs(g["colour"]).max()))


@pytest.mark.parametrize("name", ALL_CASES)
def test_composite(name):
    g = load(name)
    noise = g["noise"] if "noise" in g.files else None
    cm = O.composite(g["raw"], g["z_vals"], g["ray_d"], noise)
    for k in ("rgb_map", "acc_map", "weights", "depth_map"):
        assert maxdiff(cm[k], g[k]) < 2e-6 * max(1.0, float(np.abs(g[k]).max())), k
    fin = np.isfinite(g["disp_map"])
    assert np.array_equal(np.isnan(cm["disp_map"]), np.isnan(g["disp_map"]))   # NaN where acc == 0
    assert np.allclose(cm["disp_map"][fin], g["disp_map"][fin], rtol=1e-5)


@pytest.mark.parametrize("name", ALL_CASES)
def test_render_end_to_end(name, params):
    """Whole path vs Renderer.render of the reference (trainer.py:70's call)."""
    sd, P = params(name)
    g = load(name)
    S = int(g["S"])
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    jit = g["jitter"][0] if "jitter" in g.files else None
    noise = g["noise"] if "noise" in g.files else None
    e = O.render(g["ray_o"], g["ray_d"], g["near"], g["far"], S, g["xyz"], g["canonical_vertex"], g["faces"], P,
                 g["poses"], code_for(g, sd, name), jitter=jit, noise=noise, t_vals=tv, **light_kw(g))
    assert np.array_equal(e["z_vals"], g["render:z_vals"])
    for k in ("color", "depth_map", "acc_map", "weights"):
        big = float(np.abs(g["render:" + k]).max())
        assert maxdiff(e[k], g["render:" + k]) < ref_tol(g, "render:" + k, 1e-4), k       # north_star tolerance
        achieved = 5e-6 if "_w" not in name else 3e-5      # (w3: a sigma error of 1e-3 at |sigma| ~ 1e3 is 1e-5 in alpha)
        assert maxdiff(e[k], g["render:" + k]) < achieved * max(1.0, big), k              # what is actually achieved


def test_reference_noise_floor_documented():
    """The reference disagrees with ITSELF (float32 vs float64) by far more than 1e-4 end to end on the
    full body: parity has to be judged stage-wise / against same-precision outputs."""
    g = load("full_eval")
    assert maxdiff(g["rgb_map"], g["rgb_map_f64"]) > 1e-3
    assert maxdiff(g["sigma"], g["sigma_f64"]) > 1e-2


def test_camera_rays_restatement():
    """get_rays + get_near_far (whole-image path) - numpy restatement vs the reference's own output"""
    g = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "camera_rays.npz"))
    ro, rd, near, far, mask = O.camera_rays_np(g["K"], g["R"], g["T"], g["bounds"], int(g["H"]), int(g["W"]))
    assert np.array_equal(mask, g["mask_at_box"]) and 0 < mask.sum() < mask.size
    assert np.array_equal(ro, g["ray_o"]) and np.array_equal(rd, g["ray_d"])
    assert np.array_equal(near[mask], g["near"]) and np.array_equal(far[mask], g["far"])


def test_camera_rays_h36m_restatement():
    """SURVEY.md 8 f-2, Human3.6M convention (BASELINE configs[3]): utils/h36m_utils.py get_rays + get_near_far - numpy
    restatement vs the reference's own get_rays_within_bounds output, small camera in full and 1024 x 1024 on every 37th pixel"""
    g = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "camera_rays_h36m.npz"))
    ro, rd, near, far, mask = O.camera_rays_h36m_np(g["K"], g["R"], g["T"], g["bounds"], int(g["H"]), int(g["W"]))
    assert np.array_equal(mask, g["mask_at_box"]) and 0 < mask.sum() < mask.size
    assert np.array_equal(ro, g["ray_o"]) and maxdiff(rd, g["ray_d"]) <= 6e-8
    assert maxdiff(near[mask], g["near"]) <= 4.8e-7 and maxdiff(far[mask], g["far"]) <= 4.8e-7
    assert abs(float(np.linalg.norm(rd, axis=1).mean()) - 1.0) < 1e-6          # unit directions (h36m_utils.py:26)
    ro, rd, near, far, mask = O.camera_rays_h36m_np(g["K2"], g["R"], g["T2"], g["bounds"], int(g["H2"]), int(g["W2"]))
    p = g["pick2"]
    assert int(mask.sum()) == int(g["mask2_count"]) and np.array_equal(mask[p], g["mask2"])
    assert maxdiff(rd[p], g["ray_d2"]) <= 6e-8
    assert maxdiff(near[p], g["near2"]) <= 4.8e-7 and maxdiff(far[p], g["far2"]) <= 4.8e-7


@pytest.mark.parametrize("name", ["small_train_grads", "small_train_grads_nonoise", "full_train_grads", "small_train_grads_w2",
                                  "full_train_grads_w2", "small_train_grads_w4", "full_train_grads_w4"])
def test_train_oracle_reproduces_reference_autograd(name):
    """oracle/train_oracle.py (differentiable restatement, torch CPU) against the loss and the parameter gradients the
    reference's own loss.backward() produced (tests/golden/make_golden_grads.py)."""
    import train_oracle as TO
    g = load(name)
    noise = g["noise"] if float(g["raw_noise_std"]) > 0 else None
    loss, grads, out = TO.loss_and_grads(state(name), g, g["render:z_vals"], noise, g["target_rgb"], g["occupancy"])
    assert abs(loss - float(g["loss"])) < 1e-6
    assert maxdiff(out["color"], g["render:color"]) < 1e-5
    idx = lambda n: (np.arange(4096, dtype=np.int64) * 2654435761 + 12345) % n
    for k, a in grads.items():
        a = a.reshape(-1).astype(np.float64)
        a = a if a.size <= 20000 else a[idx(a.size)]
        b = g["grad:" + k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b) + 1e-12, k


@pytest.mark.parametrize("name", ["lbs_small", "lbs_full"])
@pytest.mark.parametrize("bw_type", ["rigid_center", "rigid_interp"])
def test_lbs_alternate_oracle_matches_reference_functions(name, bw_type):
    """orc_lbs_warp against utils/render_utils.py:352-403 compute_nn_mesh + utils/blend_utils.py:72-81 ppts_to_pts
    (tests/golden/make_golden_lbs.py)"""
    import dsnerf_amd.synth as synth
    g = load(name)
    canon, faces = synth.make_small_body() if int(g["small"]) else synth.make_body()
    xyz = synth.pose_body(canon)
    W = synth.make_skin_weights(xyz.shape[0], int(g["seed_weights"]))
    o = O.lbs_warp(g["pts"], xyz, faces, W, g["A"], 0 if bw_type == "rigid_center" else 1)
    assert np.array_equal(o["transparent"], g["transparent:" + bw_type])
    assert maxdiff(o["weights"], g["weights:" + bw_type]) < 1e-7
    assert maxdiff(o["pts_zero"], g["pts_zero:" + bw_type]) < 2e-6

