"""Round-5 GPU tests: nearest-face lists built lazily for the cells a frame visits, per-workspace relu-record capacity, the
cost-balanced block partition of the metric's own frame, the bench's own frame against the oracle, early stop that enforces its
bound.  All through the C ABI (ctypes), as everywhere."""
import os
import warnings

import numpy as np
import pytest
import torch

import oracle as O
from helpers import maxdiff, state
from test_gpu_round2 import full_frame, renderer_with

pytestmark = pytest.mark.gpu


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))


def _frame_inputs(r, batch):
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    return o, d, r._dev(batch["near"][0]), r._dev(batch["far"][0])


# ------------------------------------------------------------------------------------------------------------------------
# DSN_FRAME_LAZY_LISTS / DSN_LAZY_LISTS: the lists of the cells a frame's samples visit == every cell's lists, for that frame
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nonuniform,wname", [(False, ""), (True, "x_w4")])
@pytest.mark.parametrize("hw", [256, 48])
def test_lazy_lists_give_the_same_frame(nonuniform, wname, hw):
    """hw = 256: 4.2 M samples, the fused cell-major search (visited cells only); hw = 48: below DSN_CELLMAJOR_MIN, where a lazily set
    frame has every cell's lists built by the render call instead.  Bit-identical outputs, near / far included; a contiguous
    eighth of the rays (a rank's block of a partitioned frame) likewise; the exhaustive search agrees."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=hw, nonuniform=nonuniform)
    r = renderer_with(state(wname) if wname else state(), canon, faces, density_screen=False)
    r.eval()
    S = 64
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])

    def run(lazy, sl=slice(None), **kw):
        r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=lazy)
        assert r.scene.lazy == lazy
        n, f = n0[sl].clone(), f0[sl].clone()
        out = _lib.render_rays(r.scene, pk, _lib.RenderWorkspace(r.device), o[sl].contiguous(), d[sl].contiguous(), n, f, S, r._t_vals(S), **kw)
        return out, n, f

    R = hw * hw
    for sl in (slice(None), slice(3 * R // 8, R // 2)):
        (a, na, fa), (b, nb, fb) = run(False, sl), run(True, sl)
        assert torch.equal(na, nb) and torch.equal(fa, fb)
        for k in a:
            assert _same(a[k], b[k]), (k, sl)
        assert float(a["acc_map"].max()) > 0.05
    ex, _, _ = run(True, exhaustive=True)
    full, _, _ = run(False)
    for k in ex:
        assert _same(ex[k], full[k]), k
    # a lazily set frame's level answers no other query: the stage call stays exact (exhaustive sweep) ...
    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=True)
    pts, _ = _lib.sample(r.scene, o[:4096].contiguous(), d[:4096].contiguous(), n0[:4096].clone(), f0[:4096].clone(), S, r._t_vals(S), None, want_pts=True)
    wl = _lib.warp(r.scene, pts[:64], d[:64], S, want_dir=False)
    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=False)
    wf = _lib.warp(r.scene, pts[:64], d[:64], S, want_dir=False)
    assert torch.equal(wl["x_c"], wf["x_c"]) and torch.equal(wl["transparent"], wf["transparent"])


def test_lazy_lists_through_the_renderer():
    """Renderer sets eval frames lazily (lazy_lists = True, the default): render_view / render / render_views == the same with every
    cell's lists; a stage call after a lazily rendered frame re-sets the frame and answers from full lists; the training forward takes
    lazily set frames too since round 6 (a batch below DSN_TRAIN_CELLMAJOR_MIN completes every cell's lists itself)"""
    canon, faces, batch = full_frame(hw=256)
    sd = state("x_w4")
    r1 = renderer_with(sd, canon, faces, density_screen=False)
    r2 = renderer_with(sd, canon, faces, density_screen=False)
    r2.lazy_lists = False
    for r in (r1, r2):
        r.eval()
        r.early_stop = False

    def fresh():
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        return b

    a, b = r1.render_view(fresh()), r2.render_view(fresh())
    assert r1.scene.lazy and not r2.scene.lazy
    for k in a:
        assert _same(a[k], b[k]), k
    va = r1.render_views([fresh(), fresh(), fresh()], frames_in_flight=3, device_output=False)
    for v in va:
        for k in a:
            assert _same(v[k], b[k]), k
    # stage call on the batch the renderer has just rendered lazily
    pts = torch.rand(1, 16, 8, 3) * 0.4 + batch["xyz"][0].mean(0)
    x1, t1 = r1.w2l_without_lbs(pts, batch, r1.canonical_model)
    assert not r1.scene.lazy
    x2, t2 = r2.w2l_without_lbs(pts, batch, r2.canonical_model)
    assert torch.equal(x1, x2) and torch.equal(t1, t2)
    r1.train()
    sub = {k: (v[:, :2048].contiguous() if k in ("ray_o", "ray_d", "near", "far") else v) for k, v in fresh().items()}
    torch.manual_seed(1)
    out = r1.render(sub)["coarse"]
    assert r1.scene.lazy and out["color"].requires_grad
    r2.train()
    torch.manual_seed(1)
    out2 = r2.render(sub)["coarse"]
    assert not r2.scene.lazy                                  # (lazy_lists = False switches it off for training as well)
    for k in ("color", "acc_map", "depth_map", "weights", "z_vals"):
        assert _same(out[k].detach(), out2[k].detach()), k


def test_lazy_lists_that_do_not_fit_stay_exact_and_warn(monkeypatch):
    """DSN_NN_FINE_CAP (a smaller logical capacity) provokes the overflow of a lazily built level: the cell-major search hands every
    sample to the exhaustive pass - same frame bit for bit - and the host mirror reports the level"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)          # 1.6 M samples: the fused path
    r = renderer_with(state(), canon, faces, density_screen=False)
    r.eval()
    S = 64
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])

    def run(lazy):
        r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=lazy)
        return _lib.render_rays(r.scene, pk, _lib.RenderWorkspace(r.device), o, d, n0.clone(), f0.clone(), S, r._t_vals(S))

    ref = run(False)
    monkeypatch.setenv("DSN_NN_FINE_CAP", "200000")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = run(True)
        torch.cuda.synchronize()
        r.scene._nn_frames = 1                     # (the header of the frame just rendered is looked at when the next one is set)
        r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=True)
        r.scene.nn_watch(wait=True)
    monkeypatch.delenv("DSN_NN_FINE_CAP")
    for k in ref:
        assert _same(ref[k], got[k]), k
    assert "world_fine" in r.scene.nn_overflow and any("nearest-face level" in str(x.message) for x in w)


# ------------------------------------------------------------------------------------------------------------------------
# ABI 6: the relu-record capacity belongs to the workspace
# ------------------------------------------------------------------------------------------------------------------------
def test_workspaces_size_their_own_records():
    """VERDICT r04 #8 / ADVICE r04: (1) a workspace asked for more records does not touch another one; (2) a request made between
    the phase calls of a frame takes effect at the NEXT frame - the buffer the geometry phase filled is the one the field and
    shading phases read; (3) frames are bit-identical whatever the capacity (overflow pass)"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=384)          # 9.4 M samples: the default capacity is the 2 M-record floor
    r = renderer_with(state("x_w2"), canon, faces, density_screen=False)      # w2: solid, 39 % of the samples have sigma > 0
    r.eval()
    S = 64
    R = 384 * 384
    r._set_frame(batch)
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    a, b = _lib.RenderWorkspace(r.device), _lib.RenderWorkspace(r.device)
    ref = _lib.render_rays(r.scene, pk, a, o, d, n0.clone(), f0.clone(), S, r._t_vals(S))
    n_pos = int(a.buf[:256].view(torch.int32)[_lib.CNT_POS])
    cap0 = a.record_capacity(R, S)
    assert cap0 == max(1 << 21, R * S // 8) and n_pos > cap0          # (this frame overflows the default capacity)
    size_b = b.bytes_for(R, S)
    a.fit_records(n_pos / float(R * S))
    assert b.bytes_for(R, S) == size_b and b.want_fraction == 0.125   # (1)
    # (2) geometry phase, THEN the request, then field + shading on the same buffer
    out = _lib.render_rays(r.scene, pk, b, o, d, n0.clone(), f0.clone(), S, r._t_vals(S), phases=_lib.PHASE_GEOMETRY)
    buf = b.buf.data_ptr()
    b.fit_records(0.9)
    for ph in (_lib.PHASE_FIELD, _lib.PHASE_SHADE):
        out = _lib.render_rays(r.scene, pk, b, o, d, n0.clone(), f0.clone(), S, r._t_vals(S), phases=ph, out=out)
    assert b.buf.data_ptr() == buf and b.record_capacity(R, S) == cap0
    for k in ref:
        assert _same(ref[k], out[k]), k
    big = _lib.render_rays(r.scene, pk, b, o, d, n0.clone(), f0.clone(), S, r._t_vals(S))
    assert b.record_capacity(R, S) >= int(0.9 * R * S) - 256 and b.buf.data_ptr() != buf
    fitted = _lib.render_rays(r.scene, pk, a, o, d, n0.clone(), f0.clone(), S, r._t_vals(S))
    assert a.record_capacity(R, S) >= n_pos
    for k in ref:                                                        # (3)
        assert _same(ref[k], big[k]) and _same(ref[k], fitted[k]), k


# ------------------------------------------------------------------------------------------------------------------------
# the strong-scaling partition of the metric's own frame
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("partition", ["blocks", "tiles"])
def test_partitioned_512_frame_reassembles_bit_for_bit(partition):
    """bench.py --strong: the 512 x 512 x 64 frame cut into 8 cost-balanced contiguous blocks (or dealt in
    round-robin tiles), every share rendered alone - lazily built lists of ITS cells - and put back into ray order through the
    plan's permutation == the frame rendered in one piece, bit for bit (one pass: every ray's pixel is independent of which other
    rays share its launch)"""
    import dsnerf_amd
    from dsnerf_amd import _lib
    H = 512
    canon, faces, batch = full_frame(hw=H)
    r = renderer_with(state("x_w4"), canon, faces, density_screen=False)
    r.eval()
    S = 64
    R = H * H
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
    ws = _lib.RenderWorkspace(r.device)

    def render(idx, want_weights=False):
        r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=True)
        out = _lib.render_rays(r.scene, pk, ws, o[idx].contiguous(), d[idx].contiguous(), n0[idx].clone(), f0[idx].clone(), S,
                               r._t_vals(S), want_weights=want_weights)
        px = torch.cat([out["color"], out["disp_map"][:, None], out["acc_map"][:, None], out["depth_map"][:, None]], 1)
        return (px, out["weights"]) if want_weights else px

    whole, w = render(torch.arange(R, device=r.device), want_weights=True)
    rp = dsnerf_amd.RayParallel()
    Nw = 8
    if partition == "blocks":
        cost = (w > 1e-6).sum(1).double().cpu() + 3.0
        bounds = rp.balanced_bounds(cost, Nw)
        shares = [float(cost[bounds[k]:bounds[k + 1]].sum() / cost.sum()) for k in range(Nw)]
        assert max(shares) < 1.05 / Nw and len({bounds[k + 1] - bounds[k] for k in range(Nw)}) > 1      # balanced in cost, ragged in rays
        plan = rp.block_plan(R, bounds, r.device)
        mine = [torch.arange(bounds[k], bounds[k + 1], device=r.device) for k in range(Nw)]
    else:
        plan = rp.tile_plan(R, 2979, r.device, world=Nw)
        mine = [rp.tile_indices(R, 2979, k, Nw).to(r.device) for k in range(Nw)]
        assert len({-(-m.numel() // 2979) for m in mine}) == 1         # every rank owns the same number of tiles (the last one is short)
    slab = plan["slab"]
    allp = torch.zeros(Nw * slab, 6, device=r.device)
    for k in range(Nw):
        allp[k * slab: k * slab + mine[k].numel()] = render(mine[k])
    full = rp.undeal(allp, plan)
    assert _same(full, whole)
    assert float(whole[:, 4].max()) > 0.9


# ------------------------------------------------------------------------------------------------------------------------
# frames in flight at the size where kernels of different frames really share compute units
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["one_pass", "sliced", "screen"])
def test_frames_in_flight_are_bit_identical_at_frame_size(mode):
    """Round 5 found that frames in flight were NOT bit-identical from 1 M samples up (the round-2 test renders 96 x 96 frames, whose
    kernels never overlap): waves of one frame's small kernels that shared a SIMD with another frame's split-fp16 MFMA kernels
    (k_field16, k_light16, k_screen16: 256-448 registers) read registers before their own loads had landed - 0.2-1 % of the samples
    got a slightly different canonical point or normal, a few hundred pixels per frame moved.  Those kernels now allocate their
    SIMD's whole register file (DSN_OWN_SIMD).  Here: 256 x 256 x 64 frames, three in flight, several rounds, against the frame
    rendered alone - every image bit for bit, in the one-pass form, with front-to-back slices and with the density screen."""
    canon, faces, batch = full_frame(hw=256)
    sd = state("x_w4") if mode != "screen" else state()
    r1 = renderer_with(sd, canon, faces, density_screen=(mode == "screen"))
    r2 = renderer_with(sd, canon, faces, density_screen=(mode == "screen"))
    for r in (r1, r2):
        r.eval()
        r.early_stop = True if mode == "sliced" else False
        r.stop_schedule = None                         # (uniform slices: the same evaluation order in both renderers)

    def fresh():
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        return b

    for _ in range(3):                                  # (calibration / probe frames of a new renderer)
        ref = r2.render_view(fresh())
        r1.render_view(fresh())
    assert r1.last_frame_info["early_stop"] == (mode == "sliced") and r1.last_frame_info["density_screen"] == (mode == "screen")
    bad = 0
    for rnd in range(3):
        for v in r1.render_views([fresh() for _ in range(6)], frames_in_flight=3, device_output=False):
            for k in ref:
                bad += int((torch.nan_to_num(v[k], nan=-1.0) != torch.nan_to_num(ref[k], nan=-1.0)).sum())
    assert bad == 0, bad
    assert float(ref["coarse_acc"].max()) > 0.5


# ------------------------------------------------------------------------------------------------------------------------
# the bench's own frame, the bench's own way, against the oracle
# ------------------------------------------------------------------------------------------------------------------------
def _oracle_rows(batch, sel, S, canon, faces, sd):
    tv = torch.linspace(0.0, 1.0, steps=S).numpy()
    g = lambda k: batch[k][0].numpy()
    return O.render(g("ray_o")[sel], g("ray_d")[sel], g("near")[sel].copy(), g("far")[sel].copy(), S, g("xyz"), canon, faces, O.Params(sd),
                    g("poses"), sd["nerf.embedding.weight"][5], t_vals=tv)


def test_bench_frame_rendered_the_bench_way_matches_the_oracle():
    """bench.py's headline frame - synth body, pose seeds 3 / 5, 512 x 512 rays through the padded AABB, 64 samples, the CONVERGED
    parameters - rendered the way the timed loop renders it: Renderer defaults, a probe frame, then front-to-back slices with the
    probe-chosen schedule, three frames in flight, lazily built lists.  1 024 rays spread over the image against the C oracle
    (utils/nerf_net_utils.py:18-51 end to end) within 1e-4 ABSOLUTE on rgb / acc (images) and weights (render()), and the whole
    sliced frame within the bound the renderer states of the one-pass frame."""
    H, S = 512, 64
    canon, faces, batch = full_frame(hw=H)
    sd = state("x_w4")
    r = renderer_with(sd, canon, faces, density_screen=False)      # (Renderer's defaults: screen off, early stop "auto", schedule "auto")
    r.eval()

    def fresh():
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        return b

    for _ in range(3):
        r.render_view(fresh())
    imgs = r.render_views([fresh() for _ in range(4)], frames_in_flight=3, device_output=False)
    info = dict(r.last_frame_info)
    assert info["early_stop"] and info["early_stop_schedule"] is not None and len(info["early_stop_schedule"]) < 16, info
    assert info["early_stop_bound_abs"] <= 5.0e-5 * (1 + 1e-5)
    sel = np.linspace(0, H * H - 1, 1024).astype(np.int64)
    e = _oracle_rows(batch, sel, S, canon, faces, sd)
    for img in imgs:
        rgb = img["coarse_color"].reshape(-1, 3).numpy()[sel]
        acc = img["coarse_acc"].reshape(-1).numpy()[sel]
        assert maxdiff(rgb, e["color"]) < 1e-4 and maxdiff(acc, e["acc_map"]) < 1e-4, (maxdiff(rgb, e["color"]), maxdiff(acc, e["acc_map"]))
    assert float(e["acc_map"].max()) > 0.99 and float(e["color"].max()) > 0.3          # (the subset sees the body)
    # weights of the same rays (render() takes the same sliced path and returns them)
    out = r.render(fresh())["coarse"]
    assert r.last_frame_info["early_stop"]
    ts = torch.from_numpy(sel).cuda()
    assert np.array_equal(out["z_vals"][ts].cpu().numpy(), e["z_vals"])
    assert maxdiff(out["weights"][ts].cpu().numpy(), e["weights"]) < 1e-4
    assert maxdiff(out["color"][ts].cpu().numpy(), e["color"]) < 1e-4
    # the whole sliced frame against the whole one-pass frame: inside the stated bound
    r.early_stop = False
    one = r.render_view(fresh())
    assert not r.last_frame_info["early_stop"]
    bound = info["early_stop_bound_abs"] + 1e-6
    assert maxdiff(imgs[-1]["coarse_color"].numpy(), one["coarse_color"].numpy()) <= bound
    assert maxdiff(imgs[-1]["coarse_acc"].numpy(), one["coarse_acc"].numpy()) <= bound


def test_configs3_share_sliced_matches_the_oracle():
    """BASELINE configs[3]: one rank's block of the 1024 x 1024 x 128 frame (an eighth of the rays), sliced with its own probe-chosen
    schedule at S = 128: 256 of its rays against the oracle at 1e-4 absolute, the share within its bound of the one-pass share"""
    H, S = 1024, 128
    canon, faces, batch = full_frame(hw=H)
    sd = state("x_w4")
    r = renderer_with(sd, canon, faces, S=S, density_screen=False)
    r.eval()
    R = H * H
    lo, hi = 3 * R // 8, R // 2                                   # (a block through the torso)

    def share():
        b = dict(batch)
        for k in ("ray_o", "ray_d"):
            b[k] = batch[k][:, lo:hi].contiguous()
        for k in ("near", "far"):
            b[k] = batch[k][:, lo:hi].clone()
        return b

    for _ in range(3):
        out = r.render(share())["coarse"]
    torch.cuda.synchronize()
    info = dict(r.last_frame_info)
    assert info["early_stop"] and info["early_stop_bound_abs"] <= 5.0e-5 * (1 + 1e-5), info
    sel = np.linspace(0, hi - lo - 1, 256).astype(np.int64)
    b0 = share()
    # (the geometry-guided sampler takes the batch's FIRST ray origin: all rays of the frame share the camera origin)
    e = _oracle_rows(b0, sel, S, canon, faces, sd)
    ts = torch.from_numpy(sel).cuda()
    assert np.array_equal(out["z_vals"][ts].cpu().numpy(), e["z_vals"])
    for k in ("color", "acc_map", "weights"):
        assert maxdiff(out[k][ts].cpu().numpy(), e[k]) < 1e-4, (k, maxdiff(out[k][ts].cpu().numpy(), e[k]))
    r.early_stop = False
    one = r.render(share())["coarse"]
    bound = info["early_stop_bound_abs"] + 1e-6
    assert maxdiff(out["color"].cpu().numpy(), one["color"].cpu().numpy()) <= bound
    assert maxdiff(out["acc_map"].cpu().numpy(), one["acc_map"].cpu().numpy()) <= bound


# ------------------------------------------------------------------------------------------------------------------------
# early stop enforces its own bound
# ------------------------------------------------------------------------------------------------------------------------
def test_sliced_frame_that_breaks_its_bound_is_rendered_again():
    """The threshold of early stop assumes colours up to a scale measured on ONE probe frame.  A later frame that weighs a larger
    colour (another pose, a light edit) is noticed at its hand-over - its compositor leaves the largest colour it weighed - and
    rendered again in one pass before the caller gets it.  Provoked here by handing the threshold a scale six times too small: the
    frame comes back bit-identical to a one-pass frame, with a warning, and the scale is put right for the frames after it;
    render_views does the same for every frame of a sequence.  A light-centre edit (test.py:193-196) afterwards: whatever it does to
    the colours, every frame stays within 1e-4 absolute of its one-pass twin."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)
    r = renderer_with(state("x_w3"), canon, faces, density_screen=False)      # w3: colours up to ~2000
    r.eval()

    def fresh():
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        return b

    r.render_view(fresh())
    r._read_stop_probe(wait=True)
    pk = r.net.packed(r.device)
    cmax = pk.early_stop["colour_max"]
    assert pk.early_stop["usable"] and cmax > 20.0
    ok = r.render_view(fresh())
    assert r.last_frame_info["early_stop"] and "rendered_again_in_one_pass" not in r.last_frame_info
    r.early_stop = False
    one = r.render_view(fresh())
    r.early_stop = "auto"
    for via in ("render_view", "render_views", "render"):
        pk.set_early_stop_colour_scale(cmax / 3.0)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            if via == "render_view":
                got = r.render_view(fresh())
            elif via == "render_views":
                got = r.render_views([fresh(), fresh(), fresh()], frames_in_flight=3, device_output=False)[1]
            else:
                o = r.render(fresh())["coarse"]
                torch.cuda.synchronize()
                got = None
        assert any("rendered again in one pass" in str(x.message) for x in w), via
        assert pk.colour_scale > cmax                     # (2 x the largest colour of the frame that broke the bound)
        if got is not None:
            for k in one:
                assert _same(got[k], one[k]), (via, k)
        else:
            assert r.last_frame_info.get("rendered_again_in_one_pass") and not r.last_frame_info["early_stop"]
    # the scale is right again: the next frame is sliced and not rendered twice
    again = r.render_view(fresh())
    assert r.last_frame_info["early_stop"] and "rendered_again_in_one_pass" not in r.last_frame_info
    for k in one:
        assert maxdiff(again[k].numpy(), one[k].numpy()) < 1e-4 + 2e-6 * cmax
    # a light-centre edit between frames of the same parameters
    for shift in (0.5, 2.0):
        r.net.set_light_center(torch.tensor([0.2 + shift, -0.1, 1.0 - shift]))
        a = r.render_view(fresh())
        r.early_stop = False
        b = r.render_view(fresh())
        r.early_stop = "auto"
        c = float(b["coarse_color"].abs().max())
        assert maxdiff(a["coarse_color"].numpy(), b["coarse_color"].numpy()) < 1e-4 + 2e-6 * max(c, 1.0), shift
        assert maxdiff(a["coarse_acc"].numpy(), b["coarse_acc"].numpy()) < 1e-4, shift


def test_host_pool_fit_is_undone_with_the_last_renderer():
    """Renderer(host_pool="fit") lowers a PROCESS-wide setting; when the last Renderer is gone and nobody has
    touched the setting since, the caller's value is back"""
    import gc
    from dsnerf_amd import _lib, can_render
    quota = _lib.cpu_quota_cores()
    if quota is None:
        pytest.skip("no cgroup CPU quota on this box: the fit does nothing")
    gc.collect()
    if can_render._Renderers.alive:
        pytest.skip("renderers of earlier tests are still alive")
    canon, faces, _ = full_frame(hw=16)
    caller = max(64, int(quota) * 4)
    torch.set_num_threads(caller)
    r1 = renderer_with(state(), canon, faces)
    r2 = renderer_with(state(), canon, faces)
    lowered = torch.get_num_threads()
    assert lowered < caller and r1.host_pool[0] == caller and r2.host_pool[0] == lowered
    del r1
    gc.collect()
    assert torch.get_num_threads() == lowered                  # one Renderer is still using it
    del r2
    gc.collect()
    assert torch.get_num_threads() == caller


@pytest.mark.gpu
@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("noisy", [False, True])
def test_sixteen_lane_compositor_agrees_with_the_wave_form(S, noisy):
    """S = 64 / 128 composite with 16 lanes per ray (k_composite16); arrays that are not 16-byte aligned fall back to the one-wave-per-ray
    kernel.  The two differ only in the ORDER of the transmittance products and of the five sums; both stay within the golden tolerance
    of the reference's cumprod (the oracle's composite)."""
    from importlib import import_module
    L = import_module("dual-space-nerf_amd._lib")
    dev = torch.device("cuda:0")
    gen = np.random.default_rng(5)
    R = 1003                                  # not a multiple of 16: the last workgroup's last rows are empty
    z = np.sort(gen.uniform(2.0, 6.0, (R, S)).astype(np.float32), axis=1)
    sigma = (gen.normal(0.0, 30.0, (R, S))).astype(np.float32)
    sigma[gen.random((R, S)) < 0.5] = -1.0
    sigma[:7] = -1.0                          # empty rays: acc == 0, disp NaN
    colour = gen.uniform(0.0, 1.0, (R, S, 3)).astype(np.float32)
    tr = (gen.random((R, S)) < 0.2).astype(np.uint8)
    ray_d = gen.normal(size=(R, 3)).astype(np.float32)
    noise = gen.normal(0.0, 1.0, (R, S)).astype(np.float32) if noisy else None
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def off4(a):                              # the same values, 4 bytes past a 16-byte boundary
        flat = torch.empty(a.size + 1, dtype=torch.float32, device=dev)
        flat[1:] = T(a).reshape(-1)
        return flat[1:].view(*a.shape)

    fast = L.composite(T(colour), T(sigma), T(tr), T(z), T(ray_d), None if noise is None else T(noise))
    wave = L.composite(T(colour), T(sigma), T(tr), off4(z), T(ray_d), None if noise is None else T(noise))
    raw = np.concatenate([colour, np.where(tr != 0, 0.0, sigma).astype(np.float32)[..., None]], -1)
    ref = O.composite(raw, z, ray_d, noise)
    names = ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]
    for a, b, k in zip(fast, wave, names):
        a, b, e = a.cpu().numpy(), b.cpu().numpy(), np.asarray(ref[k])
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isnan(a), np.isnan(e)), k
        ok = np.isfinite(e)
        if k == "disp_map":
            assert np.allclose(a[ok], b[ok], rtol=2e-5) and np.allclose(a[ok], e[ok], rtol=2e-5)
        else:
            tol = 5e-6 if k == "depth_map" else 2e-6
            assert maxdiff(a[ok], b[ok]) < tol and maxdiff(a[ok], e[ok]) < tol, k
    assert not torch.equal(fast[3], wave[3])      # different kernels, different product order: bit-equal weights would mean the fallback was not taken


@pytest.mark.gpu
def test_paired_weight_gradient_launches_match_the_single_ones(tmp_path):
    """Round 5: the two weight-gradient products of a trunk layer (a_l^T hdot_{l-1} and ahat_l^T h_{l-1}) run as ONE launch that changes
    the accumulators' units between the pairs by a power of two.  DSN_WGRAD_PAIRS=0 keeps a launch per product (the switch is read once
    per process: sub-processes).  Same sums up to the order of two additions; and the paired form's trunk gradients are bit-reproducible from run to run."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for tag, env_extra in (("pairs", {}), ("pairs_again", {}), ("single", {"DSN_WGRAD_PAIRS": "0"}), ("lin_fp32", {"DSN_TRAIN_LIN": "fp32"}),
                           ("two_pass_heads", {"DSN_TRAIN_UNFUSED_HEADS": "1"})):
        env = dict(os.environ, **env_extra)
        path = str(tmp_path / (tag + ".npz"))
        p = subprocess.run([sys.executable, os.path.join(here, "_grads_dump.py"), "full_train_grads_w4", path], env=env, capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[tag] = dict(np.load(path))
    # (bit-reproducible: what the two-stage split-fp16 products make - the trunk; the heads' exact-fp32 products and column sums still
    #  end in float atomics)
    differ = [k for k, a in outs["pairs"].items() if not np.array_equal(a, outs["pairs_again"][k])]
    assert differ == [k for k in differ if "stage" not in k] and any("stage" in k for k in outs["pairs"]), differ
    for k, a in outs["pairs"].items():
        b = outs["single"][k].astype(np.float64)
        err = np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)
        assert err < 2e-6, (k, err)
        # the heads' data gradients and the backward's encoding written inside the sweeps that read their inputs (default) or by
        # kernels of their own (DSN_TRAIN_UNFUSED_HEADS=1): the same values, so the trunk's gradients agree bit for bit - with the
        # heads' two data-gradient products on the exact-fp32 kernel in both (DSN_TRAIN_LIN=fp32: round 6 moved them to the split-fp16
        # k_t_lin16, which needs the sweeps' operand scales and is not used by the unfused form)
        c, d = outs["two_pass_heads"][k], outs["lin_fp32"][k]
        if "stage" in k:
            assert np.array_equal(d, c), k
        else:
            assert np.linalg.norm(d.astype(np.float64) - c) / max(np.linalg.norm(c.astype(np.float64)), 1e-30) < 2e-6, k
        # k_t_lin16 against k_t_lin: two roundings of 2^-22 of the operand's batch-wide magnitude per product
        assert np.linalg.norm(a.astype(np.float64) - d) / max(np.linalg.norm(d.astype(np.float64)), 1e-30) < 2e-6, k
