"""Host-side logic of the binding that needs no GPU: the per-workspace relu-record capacity (ABI 6) and the slice schedule of the
front-to-back evaluation priced in rounds of the persistent grid."""
import numpy as np
import pytest


def test_render_workspace_capacity_is_its_own():
    from dsnerf_amd import _lib
    a, b = _lib.RenderWorkspace("cuda"), _lib.RenderWorkspace("cuda", fraction=0.5)
    R, S = 512 * 512, 64
    assert a.fraction == _lib.RECORD_FRACTION_DEFAULT == 0.125 and b.fraction == 0.5
    base, half = a.bytes_for(R, S), b.bytes_for(R, S)
    assert base == _lib.lib().dsn_render_workspace_bytes(R, S) < half
    # a request is remembered, not applied: sizes change at begin_frame() only, and never shrink
    assert a.fit_records(0.30, 1.25) == pytest.approx(0.375) and a.bytes_for(R, S) == base
    assert a.fit_records(0.10) == pytest.approx(0.375)
    a.begin_frame()
    assert a.fraction == pytest.approx(0.375) and base < a.bytes_for(R, S) < half
    assert b.bytes_for(R, S) == half and b.want_fraction == 0.5            # the other workspace is untouched
    assert a.fit_records(3.0) == 1.0
    a.begin_frame()
    assert a.bytes_for(R, S) == _lib.lib().dsn_render_workspace_bytes_for(R, S, 1.0)
    # small frames hold a record per sample whatever the fraction
    assert a.bytes_for(1024, 64) == b.bytes_for(1024, 64) == _lib.RenderWorkspace("cuda").bytes_for(1024, 64)


def _hist(K, alive_per_slice, die_at):
    """hist[g][k]: rays that are found finished at the start of slice g contribute to every slice k; alive_per_slice[k] samples each"""
    h = np.zeros((K + 1, K), np.int64)
    for g, share in die_at.items():
        for k in range(K):
            h[g][k] = int(share * alive_per_slice[k])
    return h


def test_slice_schedule_is_priced_in_rounds():
    from dsnerf_amd import _lib
    S, L = 64, 4
    K = S // L
    round_ = 128 * 224
    # a frame whose rays end between slices 3 and 6, then nothing happens: 40 rounds per uniform slice at the front
    alive = [40 * round_] * K
    h = _hist(K, alive, {3: 0.3, 4: 0.2, 5: 0.1, 6: 0.1, K: 0.3})
    lens, ev, un = _lib.choose_stop_schedule(h, L, S)
    assert sum(lens) == S and all(1 <= x <= 64 and x % L == 0 for x in lens)
    assert un <= ev <= 1.2 * un and len(lens) < K                     # fewer launches for a few more samples
    assert lens[-1] >= 16                                             # the tail, where no ray ends any more, is one long slice
    # a rank's eighth of it: the same shape at an eighth of the samples - slices that hold about a round each are merged
    h8 = _hist(K, [a // 320 for a in alive], {3: 0.3, 4: 0.2, 5: 0.1, 6: 0.1, K: 0.3})      # 1/8 round per uniform slice
    lens8, ev8, un8 = _lib.choose_stop_schedule(h8, L, S, round_weight=1.0)      # (every started round priced as a whole one)
    assert sum(lens8) == S and len(lens8) <= len(lens) and len(lens8) <= 4
    # the default weighs whole rounds and plain samples half and half (a started round does not cost a whole one while the neighbours'
    # kernels fill it): never fewer slices than the whole-round model, never more samples evaluated
    lens8h, ev8h, _ = _lib.choose_stop_schedule(h8, L, S)
    assert sum(lens8h) == S and len(lens8) <= len(lens8h) <= len(lens) and ev8h <= ev8
    # round 4's model (samples + 0.9 rounds per slice) keeps more, shorter slices there: a half-empty round each
    lens_old, _, _ = _lib.choose_stop_schedule(h8, L, S, quantise=False)
    rounds = lambda ls, hh: sum(-(-int(hh[a + 1:, a:b].sum()) // round_) for a, b in zip(np.cumsum([0] + ls[:-1]) // L, np.cumsum(ls) // L))
    assert rounds(lens8, h8) <= rounds(lens_old, h8)
    # nothing ever ends: one launch per 64 samples
    h0 = _hist(K, alive, {K: 1.0})
    assert _lib.choose_stop_schedule(h0, L, S)[0] == [64]
    assert _lib.choose_stop_schedule(_hist(32, [round_] * 32, {32: 1.0}), 4, 128)[0] == [64, 64]
