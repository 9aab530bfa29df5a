"""GPU parity of the fused path and the Python boundary (Renderer / DualSpaceNeRF mirrors)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle as O
from helpers import ALL_CASES, CASES, code_for, light_kw, load, maxdiff, ref_tol, state

pytestmark = pytest.mark.gpu


from cases import make_batch, make_cfg, make_renderer  # noqa: E402,F401  (oracle/cases.py: shared with __graft_entry__.smoke())


@pytest.mark.parametrize("name", ALL_CASES)
def test_render_matches_reference_and_oracle(name):
    g = load(name)
    r = make_renderer(g, name)
    train = name.startswith("small_train")
    if train:
        r.train()
        torch.manual_seed(233)       # main.py:21-26; the mirror draws rand then randn like the reference
    else:
        r.eval()
    out = r.render(make_batch(g))["coarse"]
    out = {k: v.detach().cpu().numpy() for k, v in out.items()}   # train mode: attached to the autograd node
    assert np.array_equal(out["z_vals"], g["render:z_vals"])
    # reference float32 Renderer.render on the same batch
    # 1e-4 abs (north_star); for the large-magnitude parameter set (w3: |sigma| ~ 1e3, colours in the hundreds) relative to the
    # magnitude, helpers.ref_tol
    for k, tol in (("color", 1e-4), ("acc_map", 1e-4), ("weights", 1e-4), ("depth_map", 3e-4)):
        tol = ref_tol(g, "render:" + k, tol) if "_w3" not in name else max(tol, 2e-5 * float(np.abs(g["render:" + k]).max()), 3e-4 if k != "color" else 0)
        assert maxdiff(out[k], g["render:" + k]) < tol, (k, maxdiff(out[k], g["render:" + k]), tol)
    # oracle on the same inputs (same geometry bit for bit; only MLP rounding differs)
    S = int(g["S"])
    sd = state(name)
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    jit = g["jitter"][0] if "jitter" in g.files else None
    noise = g["noise"] if "noise" in g.files else None
    e = O.render(g["ray_o"], g["ray_d"], g["near"], g["far"], S, g["xyz"], g["canonical_vertex"], g["faces"],
                 O.Params(sd), g["poses"], code_for(g, sd, name), jitter=jit, noise=noise, t_vals=tv, **light_kw(g))
    assert maxdiff(out["color"], e["color"]) < (1e-4 if "_w3" not in name else 2e-5 * float(np.abs(e["color"]).max()))
    d, dg = out["disp_map"], g["render:disp_map"]
    assert np.array_equal(np.isnan(d), np.isnan(dg))


@pytest.mark.parametrize("name", ["small_eval", "full_eval", "full_eval_w2", "full_eval_w3", "full_eval_w4"])
def test_skip_transparent_is_exact(name):
    g = load(name)
    r = make_renderer(g, name)
    r.eval()
    r.skip_transparent = True
    a = r.render(make_batch(g))["coarse"]
    a = {k: v.clone() for k, v in a.items()}
    r.skip_transparent = False
    b = r.render(make_batch(g))["coarse"]
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert torch.equal(a[k], b[k]), k


def test_render_view_matches_reference():
    g = load("small_view")
    r = make_renderer(g)
    r.eval()
    H, W = int(g["H"]), int(g["W"])
    b = make_batch(g)
    b["img"] = torch.zeros(1, H, W, 3, dtype=torch.float64)
    b["mask_at_box"] = torch.from_numpy(g["mask_at_box"])[None]
    v = r.render_view(b)
    for k in ("coarse_color", "coarse_acc", "coarse_depth"):
        assert tuple(v[k].shape) == tuple(g[k].shape) and not v[k].is_cuda
        assert maxdiff(v[k].numpy(), g[k]) < 1e-4, k
    outside = ~g["mask_at_box"].reshape(H, W)
    assert float(v["coarse_color"].numpy()[outside].sum()) == 0.0


def test_module_forward_matches_reference():
    """DualSpaceNeRF.forward(pos[N,6], rays[N,6], frame_idx, batch_info) like can_render.py:113"""
    g = load("small_eval")
    r = make_renderer(g)
    r.eval()
    b = make_batch(g)
    b["canonical_model"] = r.canonical_model
    b["face_idx"] = r.face_idx
    S = int(g["S"])
    dirs = np.repeat(g["ray_d"][:, None, :], S, 1).reshape(-1, 3)
    pos = torch.from_numpy(np.concatenate([g["pts"].reshape(-1, 3), g["x_c"]], 1))
    rays = torch.from_numpy(np.concatenate([dirs, g["ray_d_can"]], 1))
    col, den, _ = r.net(pos, rays, torch.full((g["ray_o"].shape[0], S), int(g["frame"])), batch_info=b)
    assert maxdiff(den.cpu().numpy()[:, 0], g["sigma"]) < 1e-4
    act = ~g["transparent"]
    dc = np.abs(col.cpu().numpy()[act] - g["colour"][act]).max(-1)
    # colour goes through normalize(grad sigma): per point, with the same small outlier budget as the gradient
    assert np.median(dc) < 1e-6 and np.mean(dc > 1e-4) < 5e-3, (np.median(dc), np.mean(dc > 1e-4), dc.max())
    den2 = r.net(pos, rays, int(g["frame"]), b, density_only=True)
    # density-only queries run the exact-fp32 kernel, the full forward the split-fp16 one: equal to rounding
    assert float((den2 - den).abs().max()) < 3e-5
    # query_volume (utils/visualizer.py:47-66's call)
    q = r.query_volume(torch.from_numpy(g["x_c"])[None], torch.tensor([int(g["frame"])]),
                       torch.from_numpy(g["transparent"])[None], b)
    assert q.shape == (1, g["x_c"].shape[0], 1)
    assert float(q.reshape(-1)[torch.from_numpy(g["transparent"]).to(q.device)].abs().sum()) == 0.0


def test_w2l_boundary_methods():
    g = load("small_eval")
    r = make_renderer(g)
    r.eval()
    b = make_batch(g)
    near, far = b["near"].clone(), b["far"].clone()
    pts, z = r.get_sampling_points(b["ray_o"], b["ray_d"], near, far, b["xyz"], mode="GG")
    assert np.array_equal(z[0].cpu().numpy(), g["z_vals"]) and np.array_equal(near[0].numpy(), g["near_gg"])
    p6, rays, tm = r.w2l(pts, b["ray_o"], b["ray_d"], b)
    assert np.array_equal(p6[..., 3:].reshape(-1, 3).cpu().numpy(), g["x_c"])
    assert np.array_equal(rays[..., 3:].reshape(-1, 3).cpu().numpy(), g["ray_d_can"])
    assert np.array_equal(tm.reshape(-1).cpu().numpy(), g["transparent"])


def test_full_size_properties():
    """512x512x64 (BASELINE config 2): size-independent properties + oracle on a ray subset."""
    import dsnerf_amd
    from dsnerf_amd import synth
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(512, 512, xyz, fit_box=True)
    S = 64
    g = {"canonical_vertex": canon, "faces": faces, "S": S}
    r = make_renderer(g)
    r.eval()
    poses = synth.make_poses()
    batch = {"ray_o": torch.from_numpy(rays["ray_o"])[None], "ray_d": torch.from_numpy(rays["ray_d"])[None],
             "near": torch.from_numpy(rays["near"].copy())[None], "far": torch.from_numpy(rays["far"].copy())[None],
             "xyz": torch.from_numpy(xyz)[None], "poses": torch.from_numpy(poses)[None],
             "Th": torch.tensor([0.2, -0.1, 1.0]).reshape(1, 1, 3), "frame": torch.tensor([5])}
    out = r.render(batch)["coarse"]
    w, acc, z = out["weights"], out["acc_map"], out["z_vals"]
    assert torch.isfinite(out["color"]).all() and torch.isfinite(w).all()
    assert float((w.sum(-1) - acc).abs().max()) < 1e-5                # acc is the sum of weights
    assert float(acc.max()) <= 1.0 + 1e-5 and float(w.min()) >= 0.0
    assert bool((z[:, 1:] >= z[:, :-1]).all())                           # samples are ordered
    dep = out["depth_map"]
    hit = acc > 0.99
    assert bool(((dep[hit] >= z[hit, 0] - 1e-4) & (dep[hit] <= z[hit, -1] + 1e-4)).all())
    # exact nearest-face lists vs exhaustive search: identical frame
    from dsnerf_amd import _lib
    r._set_frame(batch)
    dev = r.device
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    n2, f2 = r._dev(torch.from_numpy(rays["near"].copy())), r._dev(torch.from_numpy(rays["far"].copy()))
    ex = _lib.render_rays(r.scene, r.net.packed(dev), _lib.RenderWorkspace(dev), o, d, n2, f2, S, r._t_vals(S),
                          exhaustive=True)
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert torch.equal(ex[k], out[k]), k
    # oracle on 1024 rays spread over the image
    sel = np.linspace(0, 512 * 512 - 1, 1024).astype(np.int64)
    sd = state()
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    # same first-ray origin for the sampler: put ray 0 first (all rays share the camera origin anyway)
    e = O.render(rays["ray_o"][sel], rays["ray_d"][sel], rays["near"][sel], rays["far"][sel], S, xyz, canon, faces,
                 O.Params(sd), poses, sd["nerf.embedding.weight"][5], t_vals=tv)
    got = out["color"][torch.from_numpy(sel).cuda()].cpu().numpy()
    assert np.array_equal(out["z_vals"][torch.from_numpy(sel).cuda()].cpu().numpy(), e["z_vals"])
    assert maxdiff(got, e["color"]) < 1e-4


@pytest.mark.parametrize("R,S", [(37, 128), (1, 64), (130, 32), (65, 200)])
def test_ragged_shapes_match_oracle(R, S):
    """ragged ray counts (not multiples of the 256-thread / 128-point tiles), S = 128 (BASELINE config 4) and an S that is
    not a multiple of the wavefront width, against the oracle"""
    import dsnerf_amd
    from dsnerf_amd import synth
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(96, 96, xyz, fit_box=True)
    sel = np.linspace(0, 96 * 96 - 1, R).astype(np.int64)
    g = {"canonical_vertex": canon, "faces": faces, "S": S}
    r = make_renderer(g)
    r.eval()
    poses = synth.make_poses()
    batch = {"ray_o": torch.from_numpy(rays["ray_o"][sel])[None], "ray_d": torch.from_numpy(rays["ray_d"][sel])[None],
             "near": torch.from_numpy(rays["near"][sel].copy())[None], "far": torch.from_numpy(rays["far"][sel].copy())[None],
             "xyz": torch.from_numpy(xyz)[None], "poses": torch.from_numpy(poses)[None],
             "Th": torch.tensor([0.2, -0.1, 1.0]).reshape(1, 1, 3), "frame": torch.tensor([5])}
    out = {k: v.cpu().numpy() for k, v in r.render(batch)["coarse"].items()}
    sd = state()
    e = O.render(rays["ray_o"][sel], rays["ray_d"][sel], rays["near"][sel], rays["far"][sel], S, xyz, canon, faces,
                 O.Params(sd), poses, sd["nerf.embedding.weight"][5], t_vals=torch.linspace(0.0, 1.0, steps=S).numpy())
    assert np.array_equal(out["z_vals"], e["z_vals"])
    assert maxdiff(out["color"], e["color"]) < 1e-4
    assert maxdiff(out["acc_map"], e["acc_map"]) < 1e-4 and maxdiff(out["weights"], e["weights"]) < 1e-4


def test_all_transparent_frame():
    """rays that never come near the body: empty active list, networks not evaluated, zero image, NaN disparity"""
    import dsnerf_amd
    from dsnerf_amd import synth
    canon, faces = synth.make_small_body()
    xyz = synth.pose_body(canon)
    g = {"canonical_vertex": canon, "faces": faces, "S": 16}
    r = make_renderer(g)
    r.eval()
    R = 70
    o = np.tile(np.array([[5.0, 5.0, 5.0]], np.float32), (R, 1))
    d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (R, 1))
    batch = {"ray_o": torch.from_numpy(o)[None], "ray_d": torch.from_numpy(d)[None],
             "near": torch.full((1, R), 1.0), "far": torch.full((1, R), 2.0),
             "xyz": torch.from_numpy(xyz)[None], "poses": torch.from_numpy(synth.make_poses())[None],
             "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([0])}
    out = r.render(batch)["coarse"]
    assert float(out["color"].abs().max()) == 0.0 and float(out["acc_map"].abs().max()) == 0.0
    assert bool(torch.isnan(out["disp_map"]).all())


@pytest.mark.parametrize("train", [False, True])
def test_uniform_sampling_mode_matches_reference(train):
    """cfg.MODEL.sample_points_mode = 'uniform' end to end against the REAL reference's Renderer.render in that mode
    (tests/golden/make_golden.py --uniform): z_vals bit-exact (eval: the plain lerp; train: stratified jitter from the CPU
    generator, seed 233), colour / acc / weights within 1e-4"""
    g = load("small_uniform")
    r = make_renderer(g)
    r.sample_points_mode = "uniform"
    tag = "train:" if train else "eval:"
    if train:
        r.train()
        torch.manual_seed(233)
    else:
        r.eval()
    out = {k: v.detach().cpu().numpy() for k, v in r.render(make_batch(g))["coarse"].items()}
    assert np.array_equal(out["z_vals"], g[tag + "z_vals"])
    for k, tol in (("color", 1e-4), ("acc_map", 1e-4), ("weights", 1e-4), ("depth_map", 3e-4)):
        assert maxdiff(out[k], g[tag + k]) < tol, (k, maxdiff(out[k], g[tag + k]))
    assert np.array_equal(np.isnan(out["disp_map"]), np.isnan(g[tag + "disp_map"]))


def test_uniform_sampling_mode():
    """cfg.MODEL.sample_points_mode = 'uniform' (can_render.py:42-51): z_vals are the plain lerp of the given near/far"""
    g = load("small_eval")
    r = make_renderer(g)
    r.eval()
    r.sample_points_mode = "uniform"
    b = make_batch(g)
    out = r.render(b)["coarse"]
    S = int(g["S"])
    tv = torch.linspace(0.0, 1.0, steps=S)
    z = b["near"][0][:, None] * (1.0 - tv) + b["far"][0][:, None] * tv
    assert torch.equal(out["z_vals"].cpu(), z)
    assert torch.isfinite(out["color"]).all()


def test_train_forward_loss_parity():
    """BASELINE config 3 (forward part): train-mode render (jitter + noise from the CPU generator, seed 233, dense
    evaluation) reproduces the reference's MSE loss (trainer.py:70-72, utils/loss.py:17) on the captured batch"""
    g = load("small_train")
    r = make_renderer(g, "small_train")
    r.train()
    torch.manual_seed(233)
    out = r.render(make_batch(g))["coarse"]
    assert out["color"].requires_grad and not out["z_vals"].requires_grad
    loss = torch.nn.functional.mse_loss(out["color"].detach().cpu(), torch.from_numpy(g["target_rgb"]))
    ref = float(g["render:loss"])
    assert abs(float(loss) - ref) < 1e-6 * max(1.0, abs(ref)), (float(loss), ref)
    assert maxdiff(out["color"].detach().cpu().numpy(), g["render:color"]) < 1e-4


@pytest.mark.parametrize("H,W,frac", [(12, 12, 0.6), (37, 53, 0.3), (512, 512, 0.9), (300, 1, 0.5), (64, 64, 0.0), (64, 64, 1.0)])
def test_image_scatter_is_exact(H, W, frac):
    """f-3: post_process on the device (utils/render_utils.py:466-472) == numpy boolean-mask assignment, bit for bit,
    incl. empty and full masks, sizes that are not multiples of the block, and the clamp of test.py:62-63"""
    from dsnerf_amd import _lib
    rng = np.random.default_rng(H * 1000 + W)
    mask = rng.random(H * W) < frac
    R = int(mask.sum())
    src = {"color": rng.standard_normal((R, 3)).astype(np.float32), "disp_map": rng.random(R).astype(np.float32),
           "acc_map": rng.random(R).astype(np.float32), "depth_map": rng.random(R).astype(np.float32)}
    if R:
        src["disp_map"][0] = np.nan                      # disparity is NaN where a ray hit nothing
    dev = torch.device("cuda:0")
    out = {k: torch.from_numpy(v).to(dev) for k, v in src.items()}
    for clamp in (False, True):
        img = _lib.image_scatter(out, torch.from_numpy(mask), H, W, clamp=clamp)
        for key, name, c in (("color", "coarse_color", 3), ("disp_map", "coarse_disp", 1), ("acc_map", "coarse_acc", 1),
                             ("depth_map", "coarse_depth", 1)):
            want = np.zeros((H * W, c), np.float32)
            v = src[key].reshape(R, c)
            want[mask] = np.clip(v, 0.0, 1.0) if (clamp and key == "color") else v
            got = img[name].cpu().numpy().reshape(H * W, c)
            assert np.array_equal(got, want, equal_nan=True), (name, clamp)


def test_image_psnr_matches_float64_reference_formula():
    """metrics.py:8-21: -10 log10(mean((pred - gt)^2)) over all pixels and over mask_at_box, float64 accumulation"""
    from dsnerf_amd import _lib
    rng = np.random.default_rng(3)
    H, W = 96, 80
    pred = rng.random((H, W, 3)).astype(np.float32)
    gt = rng.random((H, W, 3))                          # float64 like batch["img"]
    mask = rng.random((H, W)) < 0.4
    got = _lib.image_psnr(torch.from_numpy(pred).cuda(), torch.from_numpy(gt), torch.from_numpy(mask)).cpu().numpy()
    d2 = (pred.astype(np.float64) - gt) ** 2
    want = np.array([d2.mean(), d2[mask].mean(), -10 * np.log10(d2.mean()), -10 * np.log10(d2[mask].mean())])
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    got32 = _lib.image_psnr(torch.from_numpy(pred).cuda(), torch.from_numpy(gt.astype(np.float32)), torch.from_numpy(mask))
    d2 = (pred.astype(np.float64) - gt.astype(np.float32).astype(np.float64)) ** 2
    assert np.allclose(got32.cpu().numpy()[:2], [d2.mean(), d2[mask].mean()], rtol=1e-12)


def test_render_view_device_output_and_metrics():
    g = load("small_view")
    r = make_renderer(g)
    r.eval()
    H, W = int(g["H"]), int(g["W"])
    b = make_batch(g)
    rng = np.random.default_rng(0)
    b["img"] = torch.from_numpy(rng.random((1, H, W, 3)))
    b["mask_at_box"] = torch.from_numpy(g["mask_at_box"])[None]
    host = r.render_view(b)
    dev = r.render_view(b, device_output=True)
    for k in host:
        assert dev[k].is_cuda and np.array_equal(dev[k].cpu().numpy(), host[k].numpy(), equal_nan=True)
    assert maxdiff(host["coarse_color"].numpy(), g["coarse_color"]) < 1e-4
    m = r.image_metrics(dev["coarse_color"], b)
    c = np.clip(host["coarse_color"].numpy().astype(np.float64), 0, 1)
    d2 = (c - b["img"][0].numpy()) ** 2
    mk = g["mask_at_box"].reshape(H, W)
    assert abs(m["psnr_woMask"] - (-10 * np.log10(d2.mean()))) < 1e-9
    assert abs(m["psnr_wMask"] - (-10 * np.log10(d2[mk].mean()))) < 1e-9


def test_density_screen_margin():
    """the plain-fp16 density screen (k_screen16) may only declare a sample empty when its accurate density is negative:
    on a whole 512 x 512 x 64 frame no empty-declared sample has sigma > 0, the margin (1 % of the term magnitude S1 + 0.01)
    is >= 10x the largest deviation actually observed, and the rendered frame is bit-identical with the screen on / off"""
    import dsnerf_amd
    from dsnerf_amd import _lib, synth
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(512, 512, xyz, fit_box=True)
    S = 64
    r = make_renderer({"canonical_vertex": canon, "faces": faces, "S": S})
    r.eval()
    batch = {"ray_o": torch.from_numpy(rays["ray_o"])[None], "ray_d": torch.from_numpy(rays["ray_d"])[None],
             "near": torch.from_numpy(rays["near"].copy())[None], "far": torch.from_numpy(rays["far"].copy())[None],
             "xyz": torch.from_numpy(xyz)[None], "poses": torch.from_numpy(synth.make_poses())[None],
             "Th": torch.tensor([0.2, -0.1, 1.0]).reshape(1, 1, 3), "frame": torch.tensor([5])}
    r._set_frame(batch)
    dev = r.device
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    n, f = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
    pts, z = _lib.sample(r.scene, o, d, n, f, S, r._t_vals(S), None)
    w = _lib.warp(r.scene, pts, d, S, want_dir=False, want_active=True)
    packed = r.net.packed(dev)
    act = w["active_list"][: int(w["active_count"][0])].long()
    sig, _, _ = _lib.field(r.scene, packed, w["x_c"], want_essence=False, want_grad=False, active=(w["active_list"], w["active_count"]), fp32=True)
    sg, s1 = _lib.screen_debug(r.scene, packed, w["x_c"])
    sig, sg, s1 = sig[act], sg[act], s1[act]
    dev_rel = ((sg - sig).abs() / s1).max().item()
    dev_abs = (sg - sig).abs().max().item()
    empty = sg < -(0.01 * s1 + 0.01)
    assert int((empty & (sig > 0)).sum()) == 0
    assert float(sig[empty].max()) < 0.0
    assert 0.01 >= 10 * dev_rel, (dev_rel, dev_abs)
    print(f"screen: {float(empty.float().mean()):.3f} of the evaluated samples declared empty; max |sigma~ - sigma| = {dev_abs:.2e}"
          f" = {dev_rel:.2e} S1; smallest slack of an empty sample {float((-sig[empty]).min()):.3e}")
    outs = []
    for screen in (True, False):
        n2, f2 = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
        outs.append(_lib.render_rays(r.scene, packed, _lib.RenderWorkspace(dev), o, d, n2, f2, S, r._t_vals(S), screen=screen))
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert torch.equal(torch.isnan(outs[0]["disp_map"]), torch.isnan(outs[1]["disp_map"]))


@pytest.mark.parametrize("name", ALL_CASES)
def test_density_screen_is_invisible_on_the_golden_cases(name):
    g = load(name)
    r = make_renderer(g, name)
    r.eval()
    from dsnerf_amd import _lib
    b = make_batch(g)
    r._set_frame(b)
    dev = r.device
    S = int(g["S"])
    o, d = r._dev(b["ray_o"][0]), r._dev(b["ray_d"][0])
    outs = []
    for screen in (True, False):
        n2, f2 = r._dev(b["near"][0]).clone(), r._dev(b["far"][0]).clone()
        outs.append(_lib.render_rays(r.scene, r.net.packed(dev), _lib.RenderWorkspace(dev), o, d, n2, f2, S, r._t_vals(S), screen=screen))
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("seed,unit_dirs,S", [(1, False, 64), (2, True, 64), (3, False, 48), (4, True, 96)])
def test_fast_paths_equal_plain_paths_on_random_frames(seed, unit_dirs, S):
    """every exact shortcut at once (cell-major + super-cell nearest face, sampler bundle cull, density screen, forward /
    reverse split) against the plain path (exhaustive search, no screen) on other poses, cameras and sample counts:
    bit-identical outputs"""
    import dsnerf_amd
    from dsnerf_amd import _lib, synth
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon, seed=20 + seed)
    rays = synth.make_rays(160, 160, xyz, fit_box=True, unit_dirs=unit_dirs)
    r = make_renderer({"canonical_vertex": canon, "faces": faces, "S": S})
    r.eval()
    batch = {"ray_o": torch.from_numpy(rays["ray_o"])[None], "ray_d": torch.from_numpy(rays["ray_d"])[None],
             "near": torch.from_numpy(rays["near"].copy())[None], "far": torch.from_numpy(rays["far"].copy())[None],
             "xyz": torch.from_numpy(xyz)[None], "poses": torch.from_numpy(synth.make_poses(seed=40 + seed))[None],
             "Th": torch.zeros(1, 1, 3), "frame": torch.tensor([7 * seed])}
    r._set_frame(batch)
    dev = r.device
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    outs = []
    for fast in (True, False):
        n2, f2 = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
        outs.append(_lib.render_rays(r.scene, r.net.packed(dev), _lib.RenderWorkspace(dev), o, d, n2, f2, S, r._t_vals(S),
                                     screen=fast, exhaustive=not fast))
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[0]["acc_map"].max()) > 0.05          # the frame is not empty


@pytest.mark.parametrize("name", ["small_eval", "full_eval"])
def test_cellmajor_search_forced_on_golden_cases(name, monkeypatch):
    """the cell-major nearest-face path (normally only above ~1 M samples) forced on the golden cases: same pixels as the
    reference and as the per-lane path, bit for bit"""
    from dsnerf_amd import _lib
    g = load(name)
    r = make_renderer(g, name)
    r.eval()
    b = make_batch(g)
    r._set_frame(b)
    dev = r.device
    S = int(g["S"])
    o, d = r._dev(b["ray_o"][0]), r._dev(b["ray_d"][0])
    outs = []
    for forced in (True, False):
        if forced:
            monkeypatch.setenv("DSN_CELLMAJOR_MIN", "1")
        else:
            monkeypatch.delenv("DSN_CELLMAJOR_MIN")
        n2, f2 = r._dev(b["near"][0]).clone(), r._dev(b["far"][0]).clone()
        outs.append(_lib.render_rays(r.scene, r.net.packed(dev), _lib.RenderWorkspace(dev), o, d, n2, f2, S, r._t_vals(S)))
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert maxdiff(outs[0]["color"].cpu().numpy(), g["render:color"]) < 1e-4


@pytest.mark.parametrize("seed,gain", [(11, 1.0), (11, 2.5), (11, 4.0), (3, 1.6), (29, 2.0)])
def test_density_screen_margin_other_weights(seed, gain):
    """the screen's safety margin on other networks: weight seeds and init gains from 1.0 (activations die out, tiny S1) to 4.0
    (|sigma| in the thousands) on a 160 x 160 x 64 frame - no sample the screen would declare empty has an accurate density
    >= 0, and the largest deviation |sigma~ - sigma| / S1 stays at least 5x below the 1 % margin"""
    from dsnerf_amd import _lib, synth
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon, seed=seed)
    rays = synth.make_rays(160, 160, xyz, fit_box=True)
    S = 64
    sd = synth.make_state_dict(seed=seed, gain=gain)
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    sc.set_frame(packed, torch.from_numpy(xyz), torch.from_numpy(synth.make_poses(seed=seed)), 5)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d, n, f = T(rays["ray_o"]), T(rays["ray_d"]), T(rays["near"]), T(rays["far"])
    pts, z = _lib.sample(sc, o, d, n, f, S, torch.linspace(0.0, 1.0, steps=S).to(dev), None)
    w = _lib.warp(sc, pts, d, S, want_dir=False, want_active=True)
    act = w["active_list"][: int(w["active_count"][0])].long()
    sig, _, _ = _lib.field(sc, packed, w["x_c"], want_essence=False, want_grad=False, active=(w["active_list"], w["active_count"]), fp32=True)
    sg, s1 = _lib.screen_debug(sc, packed, w["x_c"])
    sig, sg, s1 = sig[act], sg[act], s1[act]
    ok = torch.isfinite(sg) & torch.isfinite(s1)           # an fp16 overflow inside the screen keeps the sample (never drops it)
    empty = ok & (sg < -(0.01 * s1 + 0.01))
    assert int((empty & (sig >= 0)).sum()) == 0
    dev_rel = float(((sg - sig).abs() / s1)[ok].max())
    assert 0.01 >= 5 * dev_rel, dev_rel
    print(f"seed {seed} gain {gain}: {int(act.numel())} samples, {float(empty.float().mean()):.3f} declared empty, "
          f"max dev/S1 {dev_rel:.2e}, |sigma| max {float(sig.abs().max()):.1f}")


def test_config4_share_131072_rays_x_128_samples():
    """BASELINE configs[3] (1024 x 1024 x 128 over 8 GPUs): one GPU's share, 131 072 rays x 128 samples = 16.8 M samples.
    Size-independent properties: weights sum to acc, acc in [0, 1], depth inside [near, far] of the ray, the frame is
    bit-identical with the density screen on / off, and 4096 rays rendered on their own (rays are independent; that call is
    below the cell-major threshold, i.e. the per-lane nearest-face search) give the same pixels bit for bit"""
    from dsnerf_amd import _lib, synth
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    rays = synth.make_rays(1024, 1024, xyz, fit_box=True)
    S, R = 128, 131072
    sel = np.arange(R) + 3 * R                    # the fourth of eight contiguous blocks of the frame
    sd = synth.make_state_dict()
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    sc.set_frame(packed, torch.from_numpy(xyz), torch.from_numpy(synth.make_poses()), 5)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    o, d = T(rays["ray_o"][sel]), T(rays["ray_d"][sel])
    tv = torch.linspace(0.0, 1.0, steps=S).to(dev)

    def render(idx=slice(None), **kw):
        n, f = T(rays["near"][sel][idx]), T(rays["far"][sel][idx])
        out = _lib.render_rays(sc, packed, _lib.RenderWorkspace(dev), o[idx].contiguous(), d[idx].contiguous(), n, f, S, tv, **kw)
        return out, n, f

    a, n, f = render()
    acc, w, dep = a["acc_map"], a["weights"], a["depth_map"]
    assert torch.isfinite(a["color"]).all() and torch.isfinite(w).all()
    assert float((w.sum(-1) - acc).abs().max()) < 1e-5
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    hit = acc > 1e-3
    zz = dep[hit] / acc[hit]
    assert bool(((zz >= n[hit] - 1e-4) & (zz <= f[hit] + 1e-4)).all())
    b, _, _ = render(screen=False)
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(a[k], b[k]), k
    sub = slice(70000, 74096)
    c, _, _ = render(sub)
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(a[k][sub], c[k]), k


def test_empty_and_degenerate_ray_batches():
    """R = 0 is rejected with the library's 'empty ray batch' message (a RuntimeError, like every bad argument); R = 1 and
    S = 1 / 2 (below every tile size, S < 2 takes the sampler's single-sweep path) render finite values"""
    from dsnerf_amd import _lib, synth
    dev = torch.device("cuda:0")
    canon, faces = synth.make_body()
    xyz = synth.pose_body(canon)
    sd = synth.make_state_dict()
    packed = _lib.PackedParams(dev).update({k: torch.from_numpy(v) for k, v in sd.items()})
    sc = _lib.Scene(torch.from_numpy(canon), torch.from_numpy(faces), dev)
    sc.set_frame(packed, torch.from_numpy(xyz), torch.from_numpy(synth.make_poses()), 5)
    rays = synth.make_rays(64, 64, xyz, fit_box=True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def render(R, S):
        o, d, n, f = T(rays["ray_o"][:R]), T(rays["ray_d"][:R]), T(rays["near"][:R]), T(rays["far"][:R])
        return _lib.render_rays(sc, packed, _lib.RenderWorkspace(dev), o, d, n, f, S, torch.linspace(0.0, 1.0, steps=S).to(dev))

    with pytest.raises(RuntimeError, match="empty ray batch"):
        render(0, 64)
    for R, S in ((1, 64), (1, 1), (3, 2), (5, 1)):
        out = render(R, S)
        assert out["color"].shape == (R, 3) and out["weights"].shape == (R, S)
        assert torch.isfinite(out["color"]).all() and torch.isfinite(out["weights"]).all()
        assert float((out["weights"].sum(-1) - out["acc_map"]).abs().max()) < 1e-6
