"""C-ABI checks that need no GPU: the library loads, exports every symbol include/dsnerf.h declares,
its size helpers are sane and argument errors are reported (never silently ignored)."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    import dsnerf_amd
    return dsnerf_amd._lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsnerf.h")).read()
    return sorted(set(re.findall(r"DSN_EXPORT\s+[\w\s\*]+?\b(dsn_\w+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dsnerf.h but not exported"
    import dsnerf_amd
    assert set(dsnerf_amd._lib.EXPORTS) == set(syms)


def test_version_and_sizes(lib):
    assert lib.dsn_abi_version() == 8
    assert lib.dsn_pose_state_bytes() >= 256 + 4 * (64 + 256)          # header + DsnFrameState
    assert lib.dsn_calibrate_workspace_bytes(C.c_int64(1 << 20)) >= (1 << 20) * 28
    assert lib.dsn_packed_param_bytes() > 3_000_000            # fwd + transposed images of ~0.5 M params
    assert lib.dsn_scene_bytes(6890, 13776) > 13776 * (64 * 2 + 16 * 2)
    assert lib.dsn_scene_bytes(0, 0) == 0
    n = lib.dsn_render_workspace_bytes(1024, 64)
    assert n >= 1024 * 64 * (4 + 1 + 4 + 12 * 5 + 4)
    assert lib.dsn_render_workspace_bytes(0, 64) == 0
    # the per-sample workspace of the benchmark frame and of a quarter of its rays (a regression guard.
    # Round 3: 3.44 GB = 205 B per sample; round 4: 1.50 GB = 89 B per sample - relu records for an eighth of the samples until a probe
    # frame asks for more (dsn_record_capacity_fraction), the normal kept where the gradient was and the colour where the essence was,
    # per-slice lists sized for the slice length in use, the screen's keep list inside the gradient buffer - and 12 GB for the
    # 1024 x 1024 x 128 frame of configs[3] on ONE GPU (27.3).  The arrays stay indexed by sample.)
    whole, quarter = lib.dsn_render_workspace_bytes(512 * 512, 64), lib.dsn_render_workspace_bytes(512 * 512 // 4, 64)
    # (ABI 6: the default fraction is a constant of the library - no process-wide setting another test could have raised)
    assert 1.45e9 < whole < 1.6e9 and whole / (512 * 512 * 64) < 90
    assert lib.dsn_render_workspace_bytes(1024 * 1024, 128) < 13e9
    assert 0.24 * whole < quarter < 0.50 * whole              # (at 4 M samples the record array keeps its 2 M-sample floor: half of them)


def test_errors_are_loud(lib):
    assert lib.dsn_pack_params(None, None, None) != 0
    assert b"dsn_pack_params" in lib.dsn_last_error()
    assert lib.dsn_composite(None, None, None, None, None, None, 4, 4, None, None, None, None, None, None) != 0
    assert lib.dsn_render_rays(None, 1, 1, None, None, None, None, None, 0, 0, None, None, None, 0, None, None, None,
                               None, None, None, None, None) != 0
    assert b"dsn_render_rays" in lib.dsn_last_error()


def test_every_entry_point_rejects_null_arguments(lib):
    """each compute entry point checks its arguments before touching the device: all-NULL calls return non-zero and
    leave a message that names the function (the wrapper turns it into RuntimeError)"""
    z, i64 = None, C.c_int64
    calls = {
        "dsn_set_body": lambda: lib.dsn_set_body(z, z, z, 0, 0, z),
        "dsn_set_frame": lambda: lib.dsn_set_frame(z, 1, 1, z, z, z, 0, 0, z, z, z, z),
        "dsn_set_frame_ex": lambda: lib.dsn_set_frame_ex(z, 1, 1, z, z, z, 0, 0, z, z, z, 0, z),
        "dsn_sample_gg": lambda: lib.dsn_sample_gg(z, 1, 1, z, z, z, z, 0, 0, z, z, z, z, z),
        "dsn_warp": lambda: lib.dsn_warp(z, 1, 1, z, z, i64(0), 1, z, z, z, z, z, z, z, z, 0, z),
        "dsn_field": lambda: lib.dsn_field(z, 1, 1, z, z, i64(0), z, z, z, z, z, 0, z),
        "dsn_field_screen": lambda: lib.dsn_field_screen(z, 1, 1, z, z, i64(0), z, z, z, z, z, z),
        "dsn_field_forward": lambda: lib.dsn_field_forward(z, 1, 1, z, z, i64(0), z, z, z, z, z, z, z, z),
        "dsn_field_reverse": lambda: lib.dsn_field_reverse(z, 1, 1, z, z, i64(0), z, z, z, z, z, z, z),
        "dsn_shade": lambda: lib.dsn_shade(z, 1, 1, z, z, z, z, z, z, i64(0), 1, z, z, z, z, z, 0, z),
        "dsn_camera_rays": lambda: lib.dsn_camera_rays(z, z, z, z, 0, 0, 0, z, z, z, z, z, z),
        "dsn_set_pose": lambda: lib.dsn_set_pose(z, z, z, z, 0, 0, z, z, z, z),
        "dsn_light": lambda: lib.dsn_light(z, z, z, z, z, i64(0), z, z, 0, z),
        "dsn_calibrate_screen": lambda: lib.dsn_calibrate_screen(z, 1, 1, z, i64(0), z, z, z),
        "dsn_set_screen_margin": lambda: lib.dsn_set_screen_margin(z, C.c_float(0.01), z),
        "dsn_module_grad": lambda: lib.dsn_module_grad(z, 1, 1, z, z, z, 0, 0, z, z, z, z, i64(0), z, z, z, z, z),
        "dsn_image_scatter": lambda: lib.dsn_image_scatter(z, z, z, z, 1, z, 0, 0, 0, z, z, z, z, z, z),
        "dsn_image_psnr": lambda: lib.dsn_image_psnr(z, z, z, z, 0, 0, z, z, z),
        "dsn_render_rays_grad": lambda: lib.dsn_render_rays_grad(z, 1, 1, z, z, z, 0, 0, z, z, z, z, 0, 0, z, z, z, z, z, z, z, 0, z),
        "dsn_render_rays_train": lambda: lib.dsn_render_rays_train(z, 1, 1, z, z, z, z, z, 0, 0, z, z, z, 0, z, z, z, z, z, z, z, z, z),
    }
    for name, call in calls.items():
        assert call() != 0, name
        assert name.encode() in lib.dsn_last_error(), (name, lib.dsn_last_error())
    # frame index outside the embedding table (model/spacenet.py:84: maxFrame = 500) is an argument error, not a crash
    one = C.c_void_p(1)
    assert lib.dsn_set_frame(one, 1, 1, one, one, one, 500, 0, z, z, z, z) != 0
    assert b"frame index" in lib.dsn_last_error()


def test_round4_host_functions(lib):
    """the host-side entry points of ABI 5 / 6 need no GPU: threshold of the early stop with a colour scale, the slice-schedule checks of
    dsn_render_rays_ex, the offsets of the nearest-face level headers, the per-workspace relu-record capacity"""
    lib.dsn_early_stop_eps_scaled.restype = C.c_float
    cap = 2.0 ** -20
    e1 = lib.dsn_early_stop_eps(64)
    assert e1 == pytest.approx(min(cap, 1e-4 / (2 * 65)), rel=1e-6)
    assert lib.dsn_early_stop_eps_scaled(64, C.c_float(1.0)) == e1 == lib.dsn_early_stop_eps_scaled(64, C.c_float(0.25))   # scales < 1 count as 1
    for S, c in ((64, 2.64), (128, 300.0), (16, 4527.0)):
        e = lib.dsn_early_stop_eps_scaled(S, C.c_float(c))
        assert e == pytest.approx(min(cap, 1e-4 / (2 * (S + 1) * c)), rel=1e-5)
        assert (S + 1) * e * c <= 0.5e-4 * (1 + 1e-5)                      # the absolute bound for colours up to the scale
    assert lib.dsn_set_early_stop_colour_scale(None, C.c_float(2.0), None) != 0 and b"dsn_set_early_stop_colour_scale" in lib.dsn_last_error()
    # the schedule of dsn_render_rays_ex is checked before anything else is touched
    one = C.c_void_p(64)
    args = lambda lens: (one, 1, 1, one, one, one, one, one, 4, 16, one, None, None, 1 | 64, one, one, one, one, None, None, one,
                         C.c_size_t(0), (C.c_int32 * len(lens))(*lens), len(lens), None)
    for lens, msg in (([4, 4, 4], b"add up to S"), ([0, 8, 8], b"1 to 64"), ([1] * 33, b"32 slices")):
        assert lib.dsn_render_rays_ex(*args(lens)) != 0 and msg in lib.dsn_last_error(), (lens, lib.dsn_last_error())
    # ABI 7 / 8: the auxiliary stream and its two events go together, and the stream must not be the call's own - checked before any
    # device work (dsn_render_rays_train_ex, dsn_render_rays_grad_ex)
    two = C.c_void_p(128)
    targs = (one, 1, 1, one, one, one, one, one, 4, 16, one, None, None, 0, one, one, one, one, None, None, one, one)
    assert lib.dsn_render_rays_train_ex(*targs, None, two, None, None) != 0 and b"go together" in lib.dsn_last_error()
    assert lib.dsn_render_rays_train_ex(*targs, None, two, two, None) != 0 and b"go together" in lib.dsn_last_error()
    assert lib.dsn_render_rays_train_ex(*targs, two, two, two, two) != 0 and b"must not be the call's own stream" in lib.dsn_last_error()
    # level headers: four distinct 64-byte slots inside the scene blob, in the order of dsn_debug_nn_stats
    off = (C.c_size_t * 4)()
    assert lib.dsn_nn_header_offsets(6890, 13776, off) == 0
    offs = [int(o) for o in off]
    assert offs == sorted(offs) and len(set(offs)) == 4 and offs[0] > 256 and offs[-1] + 64 <= lib.dsn_scene_bytes(6890, 13776)
    assert all(o % 256 == 0 for o in offs)
    assert lib.dsn_nn_header_offsets(0, 0, off) != 0
    # ABI 6: the relu-record capacity belongs to the WORKSPACE - its size says what it holds, the library keeps no setting
    lib.dsn_render_workspace_bytes_for.restype = C.c_size_t
    lib.dsn_render_workspace_bytes_for.argtypes = [C.c_int, C.c_int, C.c_float]
    lib.dsn_render_workspace_record_capacity.restype = C.c_int64
    lib.dsn_render_workspace_record_capacity.argtypes = [C.c_int, C.c_int, C.c_size_t]
    R, S = 512 * 512, 64
    N = R * S
    base = lib.dsn_render_workspace_bytes(R, S)
    assert base >= 1.45e9
    assert lib.dsn_render_workspace_bytes_for(R, S, 0.125) == base == lib.dsn_render_workspace_bytes_for(R, S, 0.0) \
        == lib.dsn_render_workspace_bytes_for(R, S, float("nan"))
    half, full = lib.dsn_render_workspace_bytes_for(R, S, 0.5), lib.dsn_render_workspace_bytes_for(R, S, 7.0)
    assert base < half < full == lib.dsn_render_workspace_bytes_for(R, S, 1.0)
    assert half - base == pytest.approx(224 * (0.5 - 0.125) * N, rel=1e-3)          # nothing but the records moves with the fraction
    cap = lib.dsn_render_workspace_record_capacity
    assert cap(R, S, base) == cap(R, S, 0) == N // 8 and cap(R, S, half) == N // 2 and cap(R, S, full) == N == cap(R, S, full + (1 << 30))
    assert cap(R, S, base - 224 * 1000) == N // 8 - 1000 or cap(R, S, base - 224 * 1000) == N // 8 - 1001      # (256-byte rounding)
    fixed = base - 224 * (N // 8)
    assert cap(R, S, fixed - 4096) == -1 and cap(0, S, base) == -1
    # small frames hold a record for every sample whatever the fraction
    assert cap(1024, 64, lib.dsn_render_workspace_bytes(1024, 64)) == 1024 * 64
    # a workspace below the fixed part is refused before anything is touched
    one = C.c_void_p(64)
    rr = lambda nbytes: lib.dsn_render_rays_ex(one, 1, 1, one, one, one, one, one, R, S, one, None, None, 1, one, one, one, one, None, None,
                                               one, C.c_size_t(nbytes), None, 0, None)
    assert rr(fixed - 4096) != 0 and b"workspace_bytes" in lib.dsn_last_error()
    lib.dsn_early_stop_colour_headroom.restype = C.c_float
    import dsnerf_amd
    assert lib.dsn_early_stop_colour_headroom() == dsnerf_amd._lib.EARLY_STOP_COLOUR_HEADROOM == 2.0
    # the uniform slice length comes from the library (the binding mirrored it by hand)
    assert lib.dsn_stop_slice_len(512 * 512, 64) == 4 and lib.dsn_stop_slice_len(4096, 64) == 8 and lib.dsn_stop_slice_len(4096, 512) == 16
    assert lib.dsn_stop_slice_len(0, 64) == 0
    # the DSN_STOP_STATS histogram counts in half slices where that keeps K <= 32 (finer borders for a caller's schedule)
    assert lib.dsn_stop_stats_slice_len(512 * 512, 64) == 2 and lib.dsn_stop_stats_slice_len(4096, 64) == 4
    assert lib.dsn_stop_stats_slice_len(1024 * 1024, 128) == 4 and lib.dsn_stop_stats_slice_len(4096, 512) == 16 and lib.dsn_stop_stats_slice_len(0, 64) == 0


def test_no_fallback_without_gpu():
    import torch
    import dsnerf_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        dsnerf_amd._lib.PackedParams("cuda")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dual-space-nerf_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "liboracle" not in src and "import oracle" not in src and "orc_" not in src, f
