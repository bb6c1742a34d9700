"""Host side of the windowed (LDS-tiled) kernel: windows and the tiling buffer (layout in include/sparc_hip.h).
The kernel trusts these invariants (every (row, slab) produced exactly once, windows within the LDS
budget, records sorted by length inside a workgroup), so they are pinned here without a GPU."""
import numpy as np
import pytest

from sparc_amd.graph import TILE_LONG_ROW, TILE_MAX_ROWS, plan_tiles, plan_windows


@pytest.mark.parametrize("sizes", [[130, 40, 333, 700, 1, 96], [490] * 50, [5], [2000], [0, 7, 0, 9]])
def test_windows_cover_rows_once_and_fit_lds(sizes):
    m = int(sum(sizes))
    wins = plan_windows(m, sizes)
    assert wins[0][0] == 0 and wins[-1][1] == m
    for (a, b), (c, d) in zip(wins, wins[1:]):
        assert b == c
    assert all(0 < b - a <= TILE_MAX_ROWS for a, b in wins)


def test_windows_without_hint_are_equal_cuts():
    wins = plan_windows(1300, None)
    assert wins[0][0] == 0 and wins[-1][1] == 1300 and all(b - a <= TILE_MAX_ROWS for a, b in wins)
    assert max(b - a for a, b in wins) - min(b - a for a, b in wins) <= 1
    assert plan_windows(1300, [5, 5]) == wins     # a hint that does not add up is ignored


def _random_rows(m, rng, p_out=0.1, max_len=150):
    """Per row: stored entries [p0, p1), of which [lo, hi) inside the window."""
    p0 = np.zeros(m, np.int64); lo = np.zeros(m, np.int64); hi = np.zeros(m, np.int64); p1 = np.zeros(m, np.int64)
    pos = 0
    for r in range(m):
        before = int(rng.integers(0, 4)) if rng.random() < p_out else 0
        after = int(rng.integers(0, 40)) if rng.random() < p_out else 0
        ln = int(rng.integers(0, max_len)) if rng.random() < 0.1 else int(rng.integers(0, 30))
        p0[r], lo[r], hi[r], p1[r] = pos, pos + before, pos + before + ln, pos + before + ln + after
        pos = p1[r]
    return p0, lo, hi, p1


@pytest.mark.parametrize("n_slabs,threads", [(1, 1024), (2, 512), (3, 512), (4, 256)])
def test_tiling_buffer_invariants(n_slabs, threads):
    rng = np.random.default_rng(3)
    sizes = [300, 512, 45, 700, 130]
    m = sum(sizes)
    wins = plan_windows(m, sizes)
    p0, lo, hi, p1 = _random_rows(m, rng)
    desc, recs = plan_tiles(wins, p0, lo, hi, p1, n_slabs, threads)
    assert desc.dtype == np.int32 and recs.dtype == np.int32 and desc.shape[1] == 8 and recs.shape == (m, 4)
    assert sorted(recs[:, 0].tolist()) == list(range(m))                      # every row has exactly one record
    produced = np.zeros((m, n_slabs), np.int64)
    xcd_of_window = {}
    for pos, (q0, q1, w0, w1, slab, n_long, _, _) in enumerate(desc):
        if q1 == q0:                                                          # padding: a workgroup without rows
            assert (w0, w1, slab, n_long) == (0, 0, 0, 0)
            continue
        rows = recs[q0:q1, 0]
        assert xcd_of_window.setdefault((w0, w1), pos % 8) == pos % 8         # all workgroups of a window on one XCD
        assert 0 <= slab < n_slabs and 0 < w1 - w0 <= TILE_MAX_ROWS
        assert np.all((rows >= w0) & (rows < w1))                             # a workgroup's rows lie in its window
        length = recs[q0:q1, 2]
        assert np.all(np.diff(length) <= 0)                                   # longest first
        assert n_long == int((length > TILE_LONG_ROW).sum())
        assert np.array_equal(recs[q0:q1, 1], p0[rows]) and np.array_equal(length, (hi - lo)[rows])
        assert np.array_equal(recs[q0:q1, 3] >> 16, (lo - p0)[rows])          # entries before the window
        assert np.array_equal(recs[q0:q1, 3] & 0xFFFF, (p1 - hi)[rows])       # ... and behind it
        produced[rows, slab] += 1
    assert np.all(produced == 1)                                              # every (row, slab) exactly once


def test_tiling_fills_the_chip_when_windows_are_few():
    wins = plan_windows(980, [490, 490])
    rng = np.random.default_rng(0)
    p0, lo, hi, p1 = _random_rows(980, rng, p_out=0.0)
    desc, _ = plan_tiles(wins, p0, lo, hi, p1, 2, 512)
    per_window = {}
    for q0, q1, w0, w1, slab, *_ in desc:
        if q1 > q0:
            per_window.setdefault((w0, slab), 0)
            per_window[(w0, slab)] += 1
    assert all(v > 1 for v in per_window.values())       # windows are split into parts: more workgroups than windows x slabs
