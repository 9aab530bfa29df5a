"""Round-3 GPU tests: the host path under the reference caller's own torch CPU ops, the converged parameter set (w4) on the
bench frame, the strong-scaling tile deal.  All through the C ABI (ctypes), as everywhere."""
import time

import numpy as np
import pytest
import torch

from helpers import state
from test_gpu_round2 import full_frame, renderer_with

pytestmark = pytest.mark.gpu


def _test_py_caller_ops(out, gt, mask):
    """test.py:61-76 on the previous frame's host images: clamp, psnr with / without mask_at_box (utils/metrics.py), the
    lpips-style permute / flip - all torch CPU ops on the main thread"""
    c = torch.clamp(out["coarse_color"], min=0.0, max=1.0)
    v = (c - gt) ** 2
    a = -10 * torch.log10(torch.mean(v[mask]))
    b = -10 * torch.log10(torch.mean(v))
    pred = (2 * c - 1).permute(2, 0, 1)[None].float().flip(1)
    return float(a) + float(b) + float(pred.sum())


def test_render_view_between_the_callers_torch_cpu_ops():
    """test.py runs torch CPU ops on 512 x 512 host images between render_view calls.  Whatever the caller does
    between the frames, (1) the frames are bit-identical, (2) the caller's thread setting survives a frame (the guard restores
    it; the one-off fit to the cgroup quota happens at construction), (3) render_view costs about what it costs alone: the
    median is compared (a throttled frame takes 60-90 ms and used to hit one frame in three, profiles/r03a_h2h_guard.json)."""
    from dsnerf_amd import _lib
    H = 512
    canon, faces, batch = full_frame(hw=H)
    r = renderer_with(state(), canon, faces)
    r.eval()
    quota = _lib.cpu_quota_cores()
    before, now, q = r.host_pool
    assert q == quota and (quota is None or now <= max(1, int(quota) - 2) or now == before <= max(1, int(quota) - 2))
    threads = torch.get_num_threads()
    gt = torch.rand(H, H, 3, dtype=torch.float64)
    mask = batch["mask_at_box"][0].reshape(H, H)

    def fresh():
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        return b

    for _ in range(4):                        # one-off work of a new Renderer: calibration, early-stop probe, staging buffers
        ref = r.render_view(fresh())
    plain, busy = [], []
    for i in range(10):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = r.render_view(fresh())
        plain.append(time.perf_counter() - t)
    out = ref
    for i in range(10):
        _test_py_caller_ops(out, gt, mask)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = r.render_view(fresh())
        busy.append(time.perf_counter() - t)
        assert torch.get_num_threads() == threads
        for k in ref:
            assert torch.equal(torch.nan_to_num(out[k], nan=-1.0), torch.nan_to_num(ref[k], nan=-1.0)), k
    # (a throttled frame takes 5-8 x: the bound only has to tell that from noise of a shared box)
    assert np.median(busy) <= 1.5 * np.median(plain) + 2e-3, (np.median(plain), np.median(busy), sorted(busy))


@pytest.mark.parametrize("wname", ["", "x_w4"])
def test_render_view_in_ray_chunks_equals_the_whole_frame(wname):
    """The per-sample workspace is the caller's to size: render_view(batch, chunk=...) renders the frame in ray
    chunks on a workspace sized for ONE chunk - the same images bit for bit (rays are independent; the exact-by-margin screen and
    the exact fallbacks do not see chunk boundaries; early stop stays off here: its probe is per call)."""
    from dsnerf_amd import _lib
    H = 256
    canon, faces, batch = full_frame(hw=H)

    def run(chunk):
        r = renderer_with(state(wname) if wname else state(), canon, faces)
        r.eval()
        r.early_stop = False
        b = dict(batch)
        b["near"], b["far"] = batch["near"].clone(), batch["far"].clone()
        out = r.render_view(b, chunk=chunk)
        torch.cuda.synchronize()
        return out, r._ws.cap

    whole, cap_whole = run(None)
    parts, cap_parts = run(H * H // 4)
    ragged, _ = run(H * H // 4 + 777)
    S = 64
    assert cap_parts == _lib.lib().dsn_render_workspace_bytes(H * H // 4, S) < 0.5 * cap_whole      # (a quarter at 512 x 512)
    assert float(whole["coarse_color"].max()) > 0.05
    for k in whole:
        assert torch.equal(torch.nan_to_num(whole[k], nan=-1.0), torch.nan_to_num(parts[k], nan=-1.0)), k
        assert torch.equal(torch.nan_to_num(whole[k], nan=-1.0), torch.nan_to_num(ragged[k], nan=-1.0)), k


def test_converged_checkpoint_on_the_bench_frame():
    """w4 (trained to convergence by the HIP trainer, pinned by reference-generated goldens) on a 256 x 256 frame of the bench
    camera: the field is bimodal (most rays are either empty or opaque), termination pays, and the sliced frame stays within its
    bound of the one-pass frame; screen on / off is bit-identical"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=256)
    r = renderer_with(state("x_w4"), canon, faces)
    r.eval()
    r._set_frame(batch)
    S = 64
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    cws = _lib.RenderWorkspace(r.device)
    _lib.render_rays(r.scene, r.net.packed(r.device), cws, o, d, r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone(), S,
                     r._t_vals(S), phases=_lib.PHASE_GEOMETRY)
    info = r.net.packed(r.device).calibrate_screen(r.scene, frame=(cws, o.shape[0], S))      # on the frame's own points, as Renderer does
    assert not info["safe"] and info["points_from"].startswith("frame"), info      # (2-5 % deviation around sigma = 0: a margin of 0.2-0.5, beyond the cap of 0.15 - screen off)

    def run(**kw):
        n, f = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
        ws = _lib.RenderWorkspace(r.device)
        out = _lib.render_rays(r.scene, r.net.packed(r.device), ws, o, d, n, f, S, r._t_vals(S), **kw)
        torch.cuda.synchronize()
        return out, _lib.read_stop_stats(ws)

    ref, st0 = run(screen=False, stop_stats=True)
    scr, _ = run(screen=True)
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(ref[k], scr[k]), k
    acc = ref["acc_map"]
    assert float(((acc < 0.02) | (acc > 0.98)).float().mean()) > 0.9          # bimodal: a trained solid, not a fog
    assert float((acc > 0.98).float().mean()) > 0.15
    assert st0["would_skip"] > 0.3 * st0["active"], st0
    got, st1 = run(screen=False, early_stop=True)
    assert st1["skipped"] > 0.3 * st1["active"]
    eps = _lib.early_stop_eps(S)
    cmax = max(1.0, float(ref["color"].abs().max()))
    assert float((ref["color"] - got["color"]).abs().max()) <= (S + 1) * eps * cmax + 2e-6 * cmax
    assert float((ref["acc_map"] - got["acc_map"]).abs().max()) <= 2 * eps
    assert float((ref["weights"] - got["weights"]).abs().max()) <= eps


def test_tile_deal_reassembles_the_frame_bit_for_bit():
    """bench.py --strong [--emulate-world N]: the N round-robin tile shares of ONE frame, rendered one after the other with the
    whole-frame kernels and un-dealt into frame order, equal the frame rendered in one piece - every ray's pixel is independent of
    which other rays share its launch (the geometry-guided sampler takes the batch's first ray origin: one camera, one origin)"""
    import dsnerf_amd
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=192)
    r = renderer_with(state(), canon, faces)
    r.eval()
    r._set_frame(batch)
    r._screen_usable()
    S = 64
    R = 192 * 192
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    n0, f0 = r._dev(batch["near"][0]), r._dev(batch["far"][0])
    pk = r.net.packed(r.device)

    def render(idx):
        ws = _lib.RenderWorkspace(r.device)
        out = _lib.render_rays(r.scene, pk, ws, o[idx].contiguous(), d[idx].contiguous(), n0[idx].clone(), f0[idx].clone(), S,
                               r._t_vals(S), want_weights=False)
        return torch.cat([out["color"], out["disp_map"][:, None], out["acc_map"][:, None], out["depth_map"][:, None]], 1)

    whole = render(torch.arange(R, device=r.device))
    rp = dsnerf_amd.RayParallel()
    Nw, tile = 8, 1024
    slab = max(rp.tile_indices(R, tile, k, Nw).numel() for k in range(Nw))
    allp = torch.zeros(Nw * slab, 6, device=r.device)
    seen = torch.zeros(R, dtype=torch.int32)
    for k in range(Nw):
        idx = rp.tile_indices(R, tile, k, Nw)
        seen[idx] += 1
        allp[k * slab: k * slab + idx.numel()] = render(idx.to(r.device))
    assert bool((seen == 1).all())                                     # a partition: every ray on exactly one rank
    full = torch.empty(R, 6, device=r.device)
    for k in range(Nw):                                                # RayParallel.undeal_tiles for an emulated world
        idx = rp.tile_indices(R, tile, k, Nw).to(r.device)
        full[idx] = allp[k * slab: k * slab + idx.numel()]
    assert torch.equal(torch.nan_to_num(full, nan=-1.0), torch.nan_to_num(whole, nan=-1.0))


@pytest.mark.parametrize("mode", ["plain", "no_screen", "early_stop", "fp32", "dense"])
def test_phase_calls_equal_the_whole_frame(mode):
    """dsn_render_rays with DSN_PHASE_GEOMETRY, then _FIELD, then _SHADE (three calls, here on one stream) == one call without phase
    bits, bit for bit, in every mode of the fused path - what lets a caller put the phases on different streams"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=128)
    r = renderer_with(state("x_w3") if mode == "early_stop" else state(), canon, faces)
    r.eval()
    r._set_frame(batch)
    r._screen_usable()
    S = 64
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    kw = {"plain": {}, "no_screen": {"screen": False}, "early_stop": {"early_stop": True}, "fp32": {"fp32": True},
          "dense": {"skip_transparent": False}}[mode]
    pk = r.net.packed(r.device)

    def run(split):
        n, f = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
        ws = _lib.RenderWorkspace(r.device)
        if not split:
            return _lib.render_rays(r.scene, pk, ws, o, d, n, f, S, r._t_vals(S), **kw), n
        out = None
        for ph in (_lib.PHASE_GEOMETRY, _lib.PHASE_FIELD, _lib.PHASE_SHADE):
            out = _lib.render_rays(r.scene, pk, ws, o, d, n, f, S, r._t_vals(S), phases=ph, out=out, **kw)
        return out, n

    (a, na), (b, nb) = run(False), run(True)
    assert torch.equal(na, nb)
    for k in a:
        assert torch.equal(torch.nan_to_num(a[k], nan=-1.0), torch.nan_to_num(b[k], nan=-1.0)), k
    assert float(a["acc_map"].max()) > 0.05


@pytest.mark.parametrize("name", ["full_train_grads", "full_train_grads_w4"])
def test_training_row_skip_is_exact(name, monkeypatch):
    """Round 3: the training step evaluates only rows that can reach an output (all but transparent samples with noise <= 0) and
    back-propagates only rows with a non-zero cotangent.  Against the dense evaluation of round 2 (DSN_TRAIN_ALL_ROWS=1): same
    outputs bit for bit, same gradients up to the summation order of the weight-gradient products (the skipped rows are exact
    zeros in every sum), and a fair share of the rows really is skipped"""
    import test_gpu_render as TR
    from helpers import load
    from dsnerf_amd import _lib
    g = load(name)

    def run():
        r = TR.make_renderer(g, name)
        r.cfg.MODEL.raw_noise_std = float(g["raw_noise_std"])
        r.train()
        torch.manual_seed(int(g["seed"]))
        out = r.render(TR.make_batch(g))["coarse"]
        loss = ((out["color"] - torch.from_numpy(g["target_rgb"]).cuda()) ** 2).mean() + 0.1 * out["acc_map"].mean() \
            + 1e-3 * (out["weights"] * out["weights"]).sum() + 1e-2 * out["depth_map"].mean()
        loss.backward()
        torch.cuda.synchronize()
        R, S = out["z_vals"].shape
        rows = _lib.grad_row_counts(r._grad_ws, R, S)
        return ({k: v.detach().clone() for k, v in out.items()}, {k: p.grad.detach().clone() for k, p in r.net.named_parameters()}, rows,
                R * S)

    out_s, grad_s, rows_s, N = run()
    monkeypatch.setenv("DSN_TRAIN_ALL_ROWS", "1")
    out_d, grad_d, rows_d, _ = run()
    monkeypatch.delenv("DSN_TRAIN_ALL_ROWS")
    assert rows_d == (N, N) and rows_s[1] <= rows_s[0] < N and rows_s[1] < 0.8 * N, (rows_s, rows_d, N)
    for k in out_s:
        assert torch.equal(torch.nan_to_num(out_s[k], nan=-1.0), torch.nan_to_num(out_d[k], nan=-1.0)), k
    for k in grad_s:
        a, b = grad_s[k].double(), grad_d[k].double()
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-12, (k, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("name", ["full_train_grads", "full_train_grads_w4"])
def test_training_far_point_search_is_bit_identical(name, monkeypatch):
    """Round 3: a training batch evaluates transparent samples with positive noise; their canonical points lie outside the fine
    nearest-face grid, and big batches send them through a coarse-level cell-major search before k_normal (a wave shares one
    coarse list instead of every lane gathering its own).  Same lists, same order, same tie rule: outputs equal the per-lane
    walk's bit for bit, gradients up to the summation order of the atomics - and the batch really has such points"""
    import test_gpu_render as TR
    from helpers import load
    g = load(name)

    def run():
        r = TR.make_renderer(g, name)
        r.cfg.MODEL.raw_noise_std = max(float(g["raw_noise_std"]), 1.0)
        r.train()
        torch.manual_seed(int(g["seed"]))
        out = r.render(TR.make_batch(g))["coarse"]
        loss = ((out["color"] - torch.from_numpy(g["target_rgb"]).cuda()) ** 2).mean() + 0.1 * out["acc_map"].mean()
        loss.backward()
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in out.items()}, {k: p.grad.detach().clone() for k, p in r.net.named_parameters()})

    monkeypatch.setenv("DSN_TRAIN_FAR_SEARCH_MIN", "1")
    out_a, grad_a = run()
    monkeypatch.setenv("DSN_TRAIN_FAR_SEARCH_MIN", str(1 << 40))
    out_b, grad_b = run()
    monkeypatch.delenv("DSN_TRAIN_FAR_SEARCH_MIN")
    assert float(out_a["acc_map"].max()) > 0.01
    for k in out_a:
        assert torch.equal(torch.nan_to_num(out_a[k], nan=-1.0), torch.nan_to_num(out_b[k], nan=-1.0)), k
    for k in grad_a:
        a, b = grad_a[k].double(), grad_b[k].double()
        assert float((a - b).norm()) <= 1e-6 * float(b.norm()) + 1e-12, (k, float((a - b).norm()), float(b.norm()))
    assert float(grad_a["lighting_mlp.lights_encoding.0.weight"].abs().max()) > 0.0


@pytest.mark.parametrize("wname,far_rays", [("", False), ("x_w4", False), ("", True)])
def test_fused_search_and_warp_equals_the_two_kernel_form(wname, far_rays, monkeypatch):
    """Round 3: in the fused path the cell-major nearest-face kernel also does the rest of the warp stage (projection, transparency,
    canonical point, active list).  Against round 2's form (DSN_NN_UNFUSED: search -> nn[] -> k_warp) the frame is bit-identical -
    also when some rays run far outside the fine grid (their samples take the second, k_warp pass)"""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=160)
    r = renderer_with(state(wname) if wname else state(), canon, faces)
    r.eval()
    r._set_frame(batch)
    r._screen_usable()
    S = 64
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    n0, f0 = r._dev(batch["near"][0]).clone(), r._dev(batch["far"][0]).clone()
    if far_rays:                                   # every 7th ray is sampled uniformly up to 2 m behind the body (uniform mode keeps near / far)
        f0[::7] += 2.0
    pk = r.net.packed(r.device)

    def run():
        ws = _lib.RenderWorkspace(r.device)
        out = _lib.render_rays(r.scene, pk, ws, o, d, n0.clone(), f0.clone(), S, r._t_vals(S), uniform=far_rays)
        torch.cuda.synchronize()
        return out, int(ws.buf[:256].view(torch.int32)[_lib.CNT_ACTIVE])

    a, na = run()
    monkeypatch.setenv("DSN_NN_UNFUSED", "1")
    b, nb = run()
    monkeypatch.delenv("DSN_NN_UNFUSED")
    assert na == nb > 0
    for k in a:
        assert torch.equal(torch.nan_to_num(a[k], nan=-1.0), torch.nan_to_num(b[k], nan=-1.0)), k


def test_screen_audit_runs_by_itself():
    """Renderer.screen_audit = "auto" (the default): the first eval frame and every 8th after it carry the audit, the counters are read
    back without a wait at a later frame, audited and plain frames are bit-identical - and a margin that drops positive densities
    (set by hand) is caught within one audit period and switches the screen off, after which the frames are exact again"""
    import warnings
    from dsnerf_amd import can_render
    canon, faces, batch = full_frame(hw=256)
    r = renderer_with(state(), canon, faces)
    r.eval()
    assert r.screen_audit == "auto"
    frames = [r.render(dict(batch))["coarse"] for _ in range(can_render.SCREEN_AUDIT_EVERY + 2)]
    torch.cuda.synchronize()
    res = r.last_screen_audit()
    assert res is not None and res["audited"] > 100 and res["violations"] == 0 and r.density_screen, res
    for f in frames[1:]:
        for k in ("color", "acc_map", "depth_map", "weights"):
            assert torch.equal(frames[0][k], f[k]), k
    r.net.packed(r.device).set_screen_margin(-0.02)         # "empty" up to sigma~ < 0.02 (S1 + 1): drops positive densities
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        bad = [r.render(dict(batch))["coarse"] for _ in range(2 * can_render.SCREEN_AUDIT_EVERY + 2)]
        torch.cuda.synchronize()
        r.last_screen_audit()
    assert any("switched off" in str(w.message) for w in caught) and r.density_screen is False
    again = r.render(dict(batch))["coarse"]
    for k in ("color", "acc_map", "depth_map", "weights"):
        assert torch.equal(frames[0][k], again[k]), k
    assert not all(torch.equal(frames[0]["color"], b["color"]) for b in bad[:3])      # (the bad margin really did change pixels)
