"""Round-6 GPU tests: a lazily built nearest-face level belongs to the call that built it; a lazily set frame rendered in chunks
completes its lists once; geometry that follows the early-stop slices; the training step's re-written small kernels.  All through the
C ABI (ctypes), as everywhere."""
import numpy as np
import pytest
import torch

from helpers import state
from test_gpu_round2 import full_frame, renderer_with

pytestmark = pytest.mark.gpu


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))


def _frame_inputs(r, batch):
    o, d = r._dev(batch["ray_o"][0]), r._dev(batch["ray_d"][0])
    return o, d, r._dev(batch["near"][0]), r._dev(batch["far"][0])


# ------------------------------------------------------------------------------------------------------------------------
# lists in state lazy = 2 are the lists of the cells ONE call's samples visited
# ------------------------------------------------------------------------------------------------------------------------
def test_lazily_built_lists_are_not_walked_by_a_later_call_without_the_flag():
    """Rays A rendered WITH DSN_LAZY_LISTS on a lazily set frame leave the level holding the lists of the cells A visits.  Disjoint
    rays B rendered on the same frame WITHOUT the flag (a direct C-API caller; the Python binding always passes it) must not walk
    them - round 5 did, and cells only B visits had empty lists: wrong faces, no error.  Now such a call takes the exhaustive sweep:
    bit-identical to B on a fully built frame, and to DSN_NN_EXHAUSTIVE."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=256)          # halves of 2.1 M samples: the fused cell-major path
    r = renderer_with(state("x_w4"), canon, faces, density_screen=False)
    r.eval()
    S = 64
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
    R = 256 * 256
    A, B = slice(0, R // 2), slice(R // 2, R)

    def render(sl, **kw):
        return _lib.render_rays(r.scene, pk, _lib.RenderWorkspace(r.device), o[sl].contiguous(), d[sl].contiguous(), n0[sl].clone(),
                                f0[sl].clone(), S, r._t_vals(S), **kw)

    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=False)
    want_a, want_b = render(A), render(B)
    ex_b = render(B, exhaustive=True)
    assert float(want_b["acc_map"].max()) > 0.05
    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=True)
    got_a = render(A)                                  # (Scene.lazy is set: the binding passes DSN_LAZY_LISTS -> the level is in state 2)
    r.scene.lazy = False                               # what a C caller that forgets the flag does: same device state, no flag
    got_b = render(B)
    r.scene.lazy = True
    again_a = render(A)                                # with the flag the lists are rebuilt for the rays at hand, whatever was there
    for k in want_a:
        assert _same(want_a[k], got_a[k]), k
        assert _same(want_b[k], got_b[k]), k
        assert _same(ex_b[k], got_b[k]), k
        assert _same(want_a[k], again_a[k]), k


def test_lazily_set_frame_rendered_in_small_chunks_completes_its_lists_once():
    """Below DSN_CELLMAJOR_MIN samples a render call on a lazily set frame completes EVERY cell's lists - decided by the level's
    device header, so only the first chunk pays for it (round 5 rebuilt grid parameters and lists per chunk and
    switched the coarse level off again each time).  Chunks == the frame with full lists; afterwards the level is an ordinary one:
    the fused path with the flag and a stage call answer from it."""
    from dsnerf_amd import _lib
    canon, faces, batch = full_frame(hw=128)
    r = renderer_with(state(), canon, faces, density_screen=False)
    r.eval()
    S = 64
    o, d, n0, f0 = _frame_inputs(r, batch)
    pk = r.net.packed(r.device)
    xyz, poses = r._dev(batch["xyz"][0]), r._dev(batch["poses"][0])
    R = 128 * 128
    chunks = [slice(k, k + 4096) for k in range(0, R, 4096)]      # 262 144 samples per call: not the fused path

    def render(sl):
        return _lib.render_rays(r.scene, pk, _lib.RenderWorkspace(r.device), o[sl].contiguous(), d[sl].contiguous(), n0[sl].clone(),
                                f0[sl].clone(), S, r._t_vals(S))

    def header():
        torch.cuda.synchronize()
        off = r.scene._nn_off[0]                          # the posed mesh's fine level (dsn_nn_header_offsets)
        h = r.scene.buf[off:off + 64].view(torch.int32).cpu().numpy()
        return {"ok": int(h[9]), "total": int(h[10]), "lazy": int(h[13])}

    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=False)
    want = [render(sl) for sl in chunks]
    full = header()
    assert full["ok"] == 1 and full["lazy"] == 0
    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=True)
    assert header()["lazy"] == 1
    got0 = render(chunks[0])
    h1 = header()
    assert h1 == full, (h1, full)                        # complete after the first chunk: same entry count as the full build
    got = [got0] + [render(sl) for sl in chunks[1:]]
    assert header() == full
    for a, b in zip(want, got):
        for k in a:
            assert _same(a[k], b[k]), k
    pts, _ = _lib.sample(r.scene, o[:64].contiguous(), d[:64].contiguous(), n0[:64].clone(), f0[:64].clone(), S, r._t_vals(S), None, want_pts=True)
    wl = _lib.warp(r.scene, pts[:64], d[:64], S, want_dir=False)
    r.scene.set_frame(pk, xyz, poses, 5, False, None, None, None, fine_only=True, lazy=False)
    wf = _lib.warp(r.scene, pts[:64], d[:64], S, want_dir=False)
    assert torch.equal(wl["x_c"], wf["x_c"]) and torch.equal(wl["transparent"], wf["transparent"])
