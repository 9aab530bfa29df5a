"""Round-4 GPU tests: the early-stop threshold follows the colour scale of the loaded parameters (1e-4 ABSOLUTE on w3), the
transmittance is advanced inside the per-slice list build, the screen's frame calibration (minimum list size, cube folded in, the
list of non-transparent samples survives a sliced frame).  All through the C ABI (ctypes), as everywhere."""
import numpy as np
import pytest
import torch

from helpers import state
from test_gpu_round2 import _stop_pair, full_frame, renderer_with

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
    assert pk.colour_scale == 1.0
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
    # the watch: a sliced frame's colours are looked at again and never lower the scale
    r._read_colour_probe(wait=True)
    assert pk.colour_scale >= _lib.EARLY_STOP_COLOUR_HEADROOM * info["colour_max"] * (1 - 1e-6)


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
    act0 = ws3.buf[1024:1024 + 4 * N].view(torch.int32).clone()
    c0 = int(ws3.buf[:4].view(torch.int32)[0])
    _lib.render_rays(w3.scene, pk3, ws3, o, d, n.clone(), f.clone(), S, w3._t_vals(S), early_stop=True, screen=False)
    torch.cuda.synchronize()
    assert _lib.read_stop_stats(ws3)["skipped"] > 0
    assert int(ws3.buf[:4].view(torch.int32)[0]) == c0 == cnt      # (same rays, same mesh: the same geometry)
    act1 = ws3.buf[1024:1024 + 4 * N].view(torch.int32)
    assert torch.equal(torch.sort(act0[:c0]).values, torch.sort(act1[:c0]).values)
