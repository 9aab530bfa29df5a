"""Round-4 GPU tests: the early-stop threshold follows the colour scale of the loaded parameters (1e-4 ABSOLUTE on w3), the
transmittance is advanced inside the per-slice list build, the screen's frame calibration (minimum list size, cube folded in, the
list of non-transparent samples survives a sliced frame).  All through the C ABI (ctypes), as everywhere."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

from helpers import load, state
from test_gpu_round2 import T, _stop_pair, full_frame, renderer_with

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def test_early_stop_threshold_follows_the_colour_scale_w3_at_1e_4_absolute():
    """colour = (ELU + 1) x essence is unbounded, w3's colours reach the hundreds, and the bound of DSN_EARLY_STOP is
    (S + 1) eps x max|colour|.  With the colour scale measured on a plain frame (what Renderer's probe frame does) the sliced w3 frame
    is within 1e-4 ABSOLUTE of the one-pass frame on colour, acc and weights; with the scale left at 1 it is not required to be."""
    from dsnerf_amd import _lib
    S = 64
    run = _stop_pair(state("x_w3"), hw=160, screen=False)
    ref, st0, _ = run(stop_stats=True)
    cmax = st0["colour_max"]
    # the compositor's own maximum: at least the largest pixel, and of the magnitude the test expects of w3
    assert cmax >= float(ref["color"].abs().max()) * (1 - 1e-6) and cmax > 20.0, cmax
    r = run.renderer
    pk = r.net.packed(r.device)
    # (the helper follows the callers' protocol: the statistics frame has set the scale to 2 x its largest colour)
    assert abs(pk.colour_scale - _lib.EARLY_STOP_COLOUR_HEADROOM * cmax) <= 1e-5 * pk.colour_scale
    scale = pk.set_early_stop_colour_scale(_lib.EARLY_STOP_COLOUR_HEADROOM * cmax)
    eps = _lib.early_stop_eps(S, scale)
    assert eps < _lib.early_stop_eps(S) and (S + 1) * eps * scale <= 0.5e-4 * (1 + 1e-5)
    got, st1, _ = run(early_stop=True)
    assert st1["skipped"] > 0.3 * st1["active"] and st1["unshaded"] > 0          # termination still pays with the smaller threshold
    for k in ("color", "acc_map", "weights"):
        err = float((ref[k] - got[k]).abs().max())
        # + the float32 summation order of a pixel of magnitude cmax (the sliced frame shades fewer samples: other partial sums)
        assert err < 1e-4 + (2e-6 * cmax if k == "color" else 0.0), (k, err)
    assert float((ref["color"] - got["color"]).abs().max()) <= (S + 1) * eps * cmax + 2e-6 * cmax
    # a re-pack puts the scale back to 1 (dsn_pack_params)
    r.net.packed(r.device, force=True)
    assert r.net.packed(r.device).colour_scale == 1.0


def test_renderer_measures_the_colour_scale_on_its_probe_frame():
    """Renderer.early_stop = "auto" on w3: the probe frame leaves the largest weighed colour, the scale becomes 2 x that, the
    frames after it are sliced with the scaled threshold and stay within 1e-4 absolute of the probe frame (one-pass)"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)
    r = renderer_with(state("x_w3"), canon, faces)
    r.eval()
    r.density_screen = False
    a = r.render_view(batch, device_output=True)
    assert r.last_frame_info["early_stop"] is False
    r._read_stop_probe(wait=True)
    pk = r.net.packed(r.device)
    info = pk.early_stop
    assert info["usable"] and info["colour_max"] > 20.0
    assert abs(pk.colour_scale - _lib.EARLY_STOP_COLOUR_HEADROOM * info["colour_max"]) <= 1e-5 * pk.colour_scale
    b = r.render_view(batch, device_output=True)
    fi = r.last_frame_info
    assert fi["early_stop"] and fi["early_stop_colour_scale"] == pk.colour_scale and fi["early_stop_bound_abs"] <= 0.5e-4 * (1 + 1e-5)
    torch.cuda.synchronize()
    assert _lib.read_stop_stats(r._ws)["skipped"] > 0
    cmax = info["colour_max"]
    for k in ("coarse_color", "coarse_acc"):
        assert float((a[k] - b[k]).abs().max()) < 1e-4 + 2e-6 * cmax, k
    # the hand-over check of the sliced frame (round 5: every sliced frame, before it is returned) never lowers the scale
    assert not r._guards and pk.colour_scale >= _lib.EARLY_STOP_COLOUR_HEADROOM * info["colour_max"] * (1 - 1e-6)
    assert "rendered_again_in_one_pass" not in r.last_frame_info


@pytest.mark.parametrize("S", [64, 40])
def test_transmittance_advanced_by_the_list_build_equals_the_compositors(S):
    """Round 4: no k_advance_T launches - k_slice_alive multiplies the missing slices' factors in itself.  What it leaves out must be
    what a plain frame's statistics predict (same densities, same formula, same slice borders), on a solid body and at a ray length
    that is not a multiple of the slice."""
    run = _stop_pair(state("x_w3"), hw=128, S=S, screen=False)
    ref, st0, _ = run(stop_stats=True)
    got, st1, _ = run(early_stop=True)
    assert st0["would_skip"] > 0.2 * st0["active"]
    assert abs(st1["skipped"] - st0["would_skip"]) <= 0.005 * st0["would_skip"] + 64, (st0, st1)
    got2, st2, _ = run(early_stop=True)          # and it is reproducible: the lists do not depend on which wave advanced a ray first
    assert st2["skipped"] == st1["skipped"] and st2["unshaded"] == st1["unshaded"]
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(got[k], got2[k]), k


def test_frame_calibration_needs_a_list_and_folds_the_cube_in():
    """the screen's margin was calibrated on the non-transparent samples of whatever the first eval call
    rendered - a small ray batch or the first chunk of a chunked frame gives a handful of samples.  Now (1) fewer than 65 536 listed
    samples fall back to the cube, (2) the cube's statistic is folded into every frame calibration: the margin is never below the
    cube's own, (3) the list of non-transparent samples survives a whole DSN_EARLY_STOP frame (it used to be
    overwritten by the shading slots)."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)
    r = renderer_with(state(), canon, faces)
    r.eval()
    r._set_frame(batch)
    pk = r.net.packed(r.device)
    S = 64
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    n, f = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
    cube = dict(pk.calibrate_screen(r.scene))
    cube_own = dict(pk.calibrate_screen(r.scene, n_points=(1 << 20) // 4, other_frames=()))      # what a frame calibration folds in
    # (1) a 256-ray batch: ~6 k listed samples - the "frame" calibration must equal the cube's (same points, same statistic)
    ws_small = _lib.RenderWorkspace(r.device)
    _lib.render_rays(r.scene, pk, ws_small, o[:256].contiguous(), d[:256].contiguous(), n[:256].clone(), f[:256].clone(), S, r._t_vals(S),
                     phases=_lib.PHASE_GEOMETRY)
    torch.cuda.synchronize()
    assert 0 < int(ws_small.buf[:4].view(torch.int32)[0]) < 65536
    small = dict(pk.calibrate_screen(r.scene, frame=(ws_small, 256, S)))
    assert small["margin_statistic"] == pytest.approx(cube["margin_statistic"], rel=1e-6)
    # (2) the whole frame: margin >= the cube's
    ws = _lib.RenderWorkspace(r.device)
    _lib.render_rays(r.scene, pk, ws, o, d, n.clone(), f.clone(), S, r._t_vals(S), phases=_lib.PHASE_GEOMETRY)
    frame = dict(pk.calibrate_screen(r.scene, frame=(ws, o.shape[0], S)))
    assert frame["margin_statistic"] >= cube_own["margin_statistic"] * (1 - 1e-6) and frame["points_from"] == "frame + centroid cube"
    # (3) after a whole sliced frame the active list is still the geometry phase's
    before = ws.buf.clone()
    cnt = int(before[:4].view(torch.int32)[0])
    w3 = renderer_with(state("x_w3"), canon, faces)
    w3.eval()
    w3._set_frame(batch)
    ws3 = _lib.RenderWorkspace(w3.device)
    pk3 = w3.net.packed(w3.device)
    _lib.render_rays(w3.scene, pk3, ws3, o, d, n.clone(), f.clone(), S, w3._t_vals(S), phases=_lib.PHASE_GEOMETRY)
    torch.cuda.synchronize()
    N = o.shape[0] * S
    act0 = ws3.buf[_lib.CNT_BYTES:_lib.CNT_BYTES + 4 * N].view(torch.int32).clone()
    c0 = int(ws3.buf[:4].view(torch.int32)[0])
    _lib.render_rays(w3.scene, pk3, ws3, o, d, n.clone(), f.clone(), S, w3._t_vals(S), early_stop=True, screen=False)
    torch.cuda.synchronize()
    assert _lib.read_stop_stats(ws3)["skipped"] > 0
    assert int(ws3.buf[:4].view(torch.int32)[0]) == c0 == cnt      # (same rays, same mesh: the same geometry)
    act1 = ws3.buf[_lib.CNT_BYTES:_lib.CNT_BYTES + 4 * N].view(torch.int32)
    assert torch.equal(torch.sort(act0[:c0]).values, torch.sort(act1[:c0]).values)


# ------------------------------------------------------------------------------------------------------------------------
# SMPL-like tessellation: synth.make_body(nonuniform=True) - half of the vertices in dense caps at head / hands / feet
# ------------------------------------------------------------------------------------------------------------------------
def test_nonuniform_body_has_smpl_like_density():
    from dsnerf_amd import synth
    from scipy.spatial import cKDTree
    canon, faces = synth.make_body(nonuniform=True)
    assert canon.shape == (6890, 3) and faces.shape == (13776, 3) and len(np.unique(faces)) == 6890
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), 1)
    _, c = np.unique(e, axis=0, return_counts=True)
    assert (c == 2).all()                                      # closed 2-manifold: every edge in exactly two faces
    tri = canon[faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert np.percentile(area, 99) / np.percentile(area, 1) >= 30.0
    lobes = (np.abs(canon[:, 0]) > 0.68) | (canon[:, 1] > 0.08)        # hands + head
    assert lobes.mean() >= 0.35, lobes.mean()
    near = np.array([len(x) for x in cKDTree(canon).query_ball_point(canon, 0.05)])
    assert near.max() >= 300                                   # (the reference's X-pose SMPL fixture: 372; the uniform lattice body: 117)


def test_nonuniform_body_lists_fit_and_equal_the_exhaustive_search():
    """every nearest-face level of the SMPL-like body fits its capacity (ok = 1, no warning), and the exact-list search equals the
    exhaustive sweep bit for bit on 284 k points - in the fine grid, the coarse shell, far outside, and ON the dense hands / head
    (where thousands of centroids lie within a cell's bound) - for the posed and the canonical mesh; the sweep itself equals the oracle"""
    from dsnerf_amd import _lib
    g = load("full_eval_nu")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sc = _lib.Scene(T(g["canonical_vertex"]), T(g["faces"].astype(np.int64)), "cuda:0")
        pk = _lib.PackedParams("cuda:0").update({k: torch.from_numpy(v) for k, v in state().items()})
        sc.set_frame(pk, T(g["xyz"]), T(g["poses"]), 5)
        st = _lib.nn_stats(sc)
        assert sc.nn_watch(wait=True) == {}
    for name, (ncell, ok, total, cap) in st.items():
        assert ok == 1 and 0 < total <= cap, (name, st)
    print("nearest-face levels of the non-uniform body (cells, ok, entries, capacity):", st)
    rng = np.random.default_rng(11)
    xyz = g["xyz"]
    lo, hi = xyz.min(0), xyz.max(0)
    dense = np.nonzero((np.abs(g["canonical_vertex"][:, 0]) > 0.68) | (g["canonical_vertex"][:, 1] > 0.25))[0]
    pts = np.concatenate([rng.uniform(lo - 0.1, hi + 0.1, size=(160000, 3)), rng.uniform(lo - 0.9, hi + 0.9, size=(60000, 3)),
                          rng.uniform(lo - 3.0, hi + 3.0, size=(4000, 3)),
                          xyz[dense[rng.integers(0, dense.size, 60000)]] + rng.normal(0, 2e-3, (60000, 3))]).astype(np.float32)
    a = _lib.warp(sc, T(pts), None, 1, want_dir=False, want_uvh=True, exhaustive=False)
    b = _lib.warp(sc, T(pts), None, 1, want_dir=False, want_uvh=True, exhaustive=True)
    for k in ("face_idx", "x_c", "uv", "h", "transparent"):
        assert torch.equal(a[k], b[k]), k
    sub = pts[::97]
    assert np.array_equal(b["face_idx"].cpu().numpy()[::97], O.nearest_face(sub, O.centroids(xyz, g["faces"])))
    # canonical mesh (k_normal): the canonical points of those samples + a gradient
    x_c = a["x_c"]
    grad = T(rng.standard_normal((pts.shape[0], 3)).astype(np.float32))
    xw, rd, ess = T(pts), T(rng.standard_normal((pts.shape[0], 3)).astype(np.float32)), T(rng.random((pts.shape[0], 3)).astype(np.float32))
    ia, na, _ = _lib.shade(sc, pk, x_c, grad, xw, rd, ess, 1, exhaustive=False)
    ib, nb, _ = _lib.shade(sc, pk, x_c, grad, xw, rd, ess, 1, exhaustive=True)
    assert torch.equal(ia, ib) and torch.equal(torch.nan_to_num(na, nan=-7.0), torch.nan_to_num(nb, nan=-7.0))


@pytest.mark.parametrize("wname", ["", "x_w4"])
def test_nonuniform_body_frame(wname):
    """a 256 x 256 x 64 frame of the SMPL-like body through the fused path: the cell-major list search equals the exhaustive sweep bit
    for bit, the frame equals the oracle on a ray subset, and it costs about what the uniform body's frame costs (no silent cliff)"""
    import time
    from dsnerf_amd import _lib
    times = {}
    for nu in (True, False):
        canon, faces, batch = full_frame(hw=256, nonuniform=nu)
        r = renderer_with(state(wname) if wname else state(), canon, faces)
        r.eval()
        r.early_stop = False
        r._set_frame(batch)
        S = 64
        o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
        pk = r.net.packed(r.device)

        def run(ws=None, **kw):
            n, f = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
            ws = _lib.RenderWorkspace(r.device) if ws is None else ws
            out = _lib.render_rays(r.scene, pk, ws, o, d, n, f, S, r._t_vals(S), screen=False, **kw)
            torch.cuda.synchronize()
            return out, ws

        ref, ws = run()
        best = float("inf")
        for _ in range(4):      # (the same workspace: a fresh one costs its first touch; the best of four: other tenants of the box)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(ws)
            best = min(best, time.perf_counter() - t0)
        times[nu] = best
        if not nu:
            continue
        assert r.scene.nn_watch(wait=True) == {}
        exh, _ = run(exhaustive=True)
        for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
            assert torch.equal(torch.nan_to_num(ref[k], nan=-1.0), torch.nan_to_num(exh[k], nan=-1.0)), k
        assert float(ref["acc_map"].max()) > 0.05
        sel = np.arange(0, 256 * 256, 509)[:96]
        sd = state(wname) if wname else state()
        e = O.render(batch["ray_o"][0].numpy()[sel], batch["ray_d"][0].numpy()[sel], batch["near"][0].numpy()[sel].copy(),
                     batch["far"][0].numpy()[sel].copy(), S, batch["xyz"][0].numpy(), canon, faces, O.Params(sd), batch["poses"][0].numpy(),
                     sd["nerf.embedding.weight"][5], t_vals=torch.linspace(0.0, 1.0, steps=S).numpy())
        # the GG sampler takes the batch's FIRST ray origin for every ray (utils/pts_utils.py:31): one camera, same origin - subset is exact
        assert np.array_equal(ref["z_vals"].cpu().numpy()[sel], e["z_vals"])
        big = max(1.0, float(np.abs(e["color"]).max()))
        assert float(np.abs(ref["color"].cpu().numpy()[sel] - e["color"]).max()) < 1e-4 * big
        assert float(np.abs(ref["acc_map"].cpu().numpy()[sel] - e["acc_map"]).max()) < 1e-4
    # (the whole frame; the geometry share is what differs: +5-15 % measured.  A level that silently fell to the exhaustive sweep costs
    #  the nearest-face search 10-50 x, i.e. the frame several times its time - that is what this line is for, not a benchmark)
    assert times[True] < 2.0 * times[False] + 3e-3, times


def test_list_capacity_overflow_warns_and_stays_exact(monkeypatch):
    """A level whose lists do not fit is switched off by the build; queries fall through (same index).  That used to be silent: now
    Scene warns - at construction for the canonical mesh, a few frames later (asynchronously) for the posed one."""
    from dsnerf_amd import _lib
    g = load("full_eval_nu")
    cv, fc = T(g["canonical_vertex"]), T(g["faces"].astype(np.int64))
    pk = _lib.PackedParams("cuda:0").update({k: torch.from_numpy(v) for k, v in state().items()})
    ok_scene = _lib.Scene(cv, fc, "cuda:0")
    ok_scene.set_frame(pk, T(g["xyz"]), T(g["poses"]), 5)
    pts = T(g["pts"].reshape(-1, 3))
    want = _lib.warp(ok_scene, pts, None, 1, want_dir=False)
    monkeypatch.setenv("DSN_NN_FINE_CAP", "200000")
    with pytest.warns(UserWarning, match="canon fine nearest-face level"):
        sc = _lib.Scene(cv, fc, "cuda:0")
    assert "canon_fine" in sc.nn_overflow and sc.nn_overflow["canon_fine"][1] == 200000
    with pytest.warns(UserWarning, match="world fine nearest-face level"):
        sc.set_frame(pk, T(g["xyz"]), T(g["poses"]), 5)
        sc.nn_watch(wait=True)
    st = _lib.nn_stats(sc)
    assert st["world_fine"][1] == 0 and st["world_fine"][2] > 200000
    got = _lib.warp(sc, pts, None, 1, want_dir=False)
    for k in ("face_idx", "x_c", "transparent"):
        assert torch.equal(got[k], want[k]), k
    assert np.array_equal(got["face_idx"].cpu().numpy(), g["idx_world"].reshape(-1))


def test_density_screen_is_opt_in():
    """the plain-fp16 density screen is statistically safe (calibrated margin + audit), not proven exact - so it is off
    unless asked for: a new Renderer renders default-parameter frames without it (every non-transparent sample takes the accurate
    pass), dsn_render_rays without DSN_DENSITY_SCREEN likewise; opted in, it runs and the frame keeps its bits."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)
    r = renderer_with(state(), canon, faces, density_screen=False)
    r.eval()
    assert r.density_screen is False
    a = r.render_view(batch, device_output=True)
    torch.cuda.synchronize()
    assert r.last_frame_info["density_screen"] is False and r.screen_info is None
    c = r._ws.buf[:256].view(torch.int32).cpu()
    assert int(c[_lib.CNT_KEEP]) == 0 and int(c[_lib.CNT_ACTIVE]) > 0           # no screen ran
    r2 = renderer_with(state(), canon, faces, density_screen=True)
    r2.eval()
    b = r2.render_view(batch, device_output=True)
    torch.cuda.synchronize()
    assert r2.last_frame_info["density_screen"] is True and r2.screen_info["usable"]
    c2 = r2._ws.buf[:256].view(torch.int32).cpu()
    assert 0 < int(c2[_lib.CNT_KEEP]) < int(c2[_lib.CNT_ACTIVE])
    for k in a:
        assert torch.equal(torch.nan_to_num(a[k], nan=-1.0), torch.nan_to_num(b[k], nan=-1.0)), k


def test_stop_schedule_from_the_probe_frame():
    """dsn_render_rays_ex: slice lengths chosen from a probe frame's histogram (longer slices where few rays end).  The histogram is
    consistent with the scalar statistic; the chosen schedule has fewer slices, evaluates what the histogram predicts (within the
    borderline rays) and a little more than the uniform schedule; the frame stays within the early-stop bound of the one-pass frame;
    a schedule that does not add up to S is refused."""
    from dsnerf_amd import _lib
    S = 64
    run = _stop_pair(state("x_w4"), hw=256, S=S, screen=False)
    r = run.renderer
    ref, st0, ws0 = run(stop_stats=True)
    R = ref["color"].shape[0]
    hist, L = _lib.read_stop_hist(ws0, R, S)
    K = hist.shape[1]
    assert hist.shape == (K + 1, K) and L == _lib.lib().dsn_stop_stats_slice_len(R, S) and L in (_lib.stop_slice_len(R, S), _lib.stop_slice_len(R, S) // 2)
    assert int(hist.sum()) == st0["active"]                               # every non-transparent sample is in exactly one bin
    dead = sum(int(hist[g][k]) for g in range(K + 1) for k in range(K) if k >= g)
    # samples of slices at / behind the ray's first dead slice: the histogram's slices are half the uniform ones since round 6's last
    # session, the scalar statistic still counts by the uniform slices (what DSN_EARLY_STOP without a schedule leaves out)
    Lu = _lib.stop_slice_len(R, S)
    assert dead >= st0["would_skip"] and (L != Lu or dead == st0["would_skip"])
    if L != Lu:      # merged pairs of the histogram's slices = the uniform slices
        m = Lu // L
        dead_u = sum(int(hist[g][k]) for g in range(K + 1) for k in range(K) if k >= m * ((g + m - 1) // m))
        assert dead_u == st0["would_skip"], (dead_u, st0["would_skip"])
    lens, ev, un = _lib.choose_stop_schedule(hist, L, S)
    assert sum(lens) == S and all(1 <= x <= 64 for x in lens) and len(lens) < K
    assert un == st0["active"] - dead and un <= ev <= 1.15 * un
    uni, st_u, _ = run(early_stop=True)
    got, st_s, ws_s = run(early_stop=True, stop_schedule=lens)
    evaluated = st_s["active"] - st_s["skipped"]
    assert abs(evaluated - ev) <= 0.005 * ev + 64, (evaluated, ev)
    # (the schedule's borders are a subset of the histogram's - which are finer than the uniform slices since round 6: it may leave out
    #  more than the uniform slicing, never more than the statistics' own count, borderline rays aside)
    assert st_s["skipped"] <= dead + 64 and st_u["skipped"] <= st0["would_skip"] + 64
    eps = _lib.early_stop_eps(S, r.net.packed(r.device).colour_scale)
    cmax = max(1.0, float(ref["color"].abs().max()))
    for out in (uni, got):
        assert float((ref["color"] - out["color"]).abs().max()) <= (S + 1) * eps * cmax + 2e-6 * cmax
        assert float((ref["acc_map"] - out["acc_map"]).abs().max()) <= 2 * eps
        assert float((ref["weights"] - out["weights"]).abs().max()) <= eps
    with pytest.raises(RuntimeError, match="add up to S"):
        run(early_stop=True, stop_schedule=[4] * 15)
    with pytest.raises(RuntimeError, match="1 to 64"):
        run(early_stop=True, stop_schedule=[0] + [4] * 16)
    # the Renderer picks a schedule by itself from its probe frame and reports it
    canon, faces, batch = full_frame(hw=256)
    rr = renderer_with(state("x_w4"), canon, faces, density_screen=False)
    rr.eval()
    a = rr.render_view(batch, device_output=True)
    rr._read_stop_probe(wait=True)
    info = rr.net.packed(rr.device).early_stop
    assert info["usable"] and sum(info["schedule"]) == S and len(info["schedule"]) < K
    b = rr.render_view(batch, device_output=True)
    assert rr.last_frame_info["early_stop"] and rr.last_frame_info["early_stop_schedule"] == info["schedule"]
    assert float((a["coarse_color"] - b["coarse_color"]).abs().max()) < 1e-4
