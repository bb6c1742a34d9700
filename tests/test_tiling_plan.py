"""Host side of the windowed (LDS-tiled) kernel: windows and the tiling buffer (layout in include/sparc_hip.h).
The kernel trusts these invariants (every (row, slab) produced exactly once, windows within the LDS
budget, records sorted by length inside a workgroup), so they are pinned here without a GPU."""
import numpy as np
import pytest

from sparc_amd.graph import TILE_LONG_ROW, TILE_MAX_ROWS, TILE_MAX_SLOTS, plan_tiles, plan_windows


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
    """Per row: the number of stored entries inside and outside its window."""
    n_in = np.where(rng.random(m) < 0.1, rng.integers(0, max_len, m), rng.integers(0, 30, m)).astype(np.int64)
    n_out = np.where(rng.random(m) < p_out, rng.integers(1, 90, m), 0).astype(np.int64)
    return n_in, n_out


@pytest.mark.parametrize("n_slabs,threads", [(1, 1024), (2, 512), (3, 512), (4, 256)])
def test_tiling_plan_invariants(n_slabs, threads):
    rng = np.random.default_rng(3)
    sizes = [300, 512, 45, 700, 130]
    m = sum(sizes)
    wins = plan_windows(m, sizes)
    n_in, n_out = _random_rows(m, rng)
    pl = plan_tiles(wins, n_in, n_out, n_slabs, threads)
    desc, slot_rows = pl["desc"], pl["slot_rows"]
    assert desc.dtype == np.int32 and desc.shape[1] == 8 and slot_rows.shape[1] == 4
    listed = slot_rows[slot_rows >= 0]
    assert sorted(listed.tolist()) == list(range(m))                          # every row sits in exactly one slot
    produced = np.zeros((m, n_slabs), np.int64)
    xcd_of_window = {}
    for pos, (s0, s1, w0, w1, slab, n_long, _, _) in enumerate(desc):
        if s1 == s0:                                                          # padding: a workgroup without slots
            assert (w0, w1, slab, n_long) == (0, 0, 0, 0)
            continue
        assert xcd_of_window.setdefault((w0, w1), pos % 8) == pos % 8         # all workgroups of a window on one XCD
        assert 0 <= slab < n_slabs and 0 < w1 - w0 <= TILE_MAX_ROWS and 0 < s1 - s0 <= TILE_MAX_SLOTS
        rows = slot_rows[s0:s1]
        assert np.all((rows[rows >= 0] >= w0) & (rows[rows >= 0] < w1))       # a workgroup's rows lie in its window
        assert np.all(rows[:n_long, 1:] == -1) and np.all(n_in[rows[:n_long, 0]] > TILE_LONG_ROW)
        assert np.all(n_in[rows[n_long:][rows[n_long:] >= 0]] <= TILE_LONG_ROW)
        work = np.where(rows >= 0, n_in[np.maximum(rows, 0)], -1)
        assert np.all(work[:, 0] == work.max(axis=1))                         # the first row of a slot is its longest
        assert np.all(np.diff(work[:, 0]) <= 0)                               # slots by decreasing work
        produced[rows[rows >= 0], slab] += 1
    assert np.all(produced == 1)                                              # every (row, slab) exactly once
    # the entry streams: every entry has its own place inside its slot's share, as the kernel indexes it
    L, O, base, obase = pl["slot_L"], pl["slot_O"], pl["slot_base"], pl["slot_obase"]
    assert np.all(base % 16 == 0) and np.all(np.diff(base) == L[:-1] * 16) and pl["stream_len"] == base[-1] + L[-1] * 16
    assert np.all(np.diff(obase) == O[:-1] * 4) and pl["ostream_len"] == obase[-1] + O[-1] * 4
    seen, oseen = np.zeros(pl["stream_len"], np.int64), np.zeros(pl["ostream_len"], np.int64)
    for r in range(m):
        sl, g, lng = pl["row_slot"][r], pl["row_g"][r], pl["row_long"][r]
        assert slot_rows[sl, g] == r and lng == (n_in[r] > TILE_LONG_ROW)
        e = np.arange(n_in[r])
        p = base[sl] + (e if lng else 16 * (e // 4) + 4 * g + e % 4)
        assert np.all(p < base[sl] + 16 * L[sl])
        seen[p] += 1
        t = np.arange(n_out[r])
        q = obase[sl] + (t if lng else 4 * t + g)
        assert np.all(q < obase[sl] + 4 * O[sl])
        oseen[q] += 1
    assert seen.max() <= 1 and oseen.max() <= 1 and seen.sum() == n_in.sum() and oseen.sum() == n_out.sum()


def test_tiling_fills_the_chip_when_windows_are_few():
    wins = plan_windows(980, [490, 490])
    rng = np.random.default_rng(0)
    n_in, n_out = _random_rows(980, rng, p_out=0.0)
    desc = plan_tiles(wins, n_in, n_out, 2, 512)["desc"]
    per_window = {}
    for s0, s1, w0, w1, slab, *_ in desc:
        if s1 > s0:
            per_window.setdefault((w0, slab), 0)
            per_window[(w0, slab)] += 1
    assert all(v > 1 for v in per_window.values())       # windows are split into parts: more workgroups than windows x slabs


def test_parts_never_exceed_the_record_budget():
    wins = plan_windows(512, [512])
    n_in = np.full(512, 100, np.int64)                   # every row is a long row: one slot each
    pl = plan_tiles(wins, n_in, np.zeros(512, np.int64), 1, 1024, target_wgs=1)
    assert all(s1 - s0 <= TILE_MAX_SLOTS for s0, s1, *_ in pl["desc"])
