"""The partitioner on the MI355X: the per-sweep label-score kernel (sparc_label_scores) against the torch
formulation of the same sweep, which the CPU tests pin (exactly: integer weights), in the three sweep modes
and on nodes of every degree class (isolated, up to 64 neighbours: register path, more: LDS table), and the
partitioner's contract and quality end to end on the device (METIS itself is absent: parity unpinned)."""
import numpy as np
import pytest
import torch

from sparc_amd import partition as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def random_graph(rng, n, avg_deg, hubs=0, hub_deg=0):
    u = rng.integers(0, n, n * avg_deg // 2)
    v = rng.integers(0, n, n * avg_deg // 2)
    if hubs:
        hu = np.repeat(rng.choice(n, hubs, replace=False), hub_deg)
        u = np.concatenate([u, hu])
        v = np.concatenate([v, rng.integers(0, n, hubs * hub_deg)])
    keep = u != v
    u, v = u[keep], v[keep]
    key = np.unique(np.concatenate([u * n + v, v * n + u]))
    return torch.from_numpy(key // n), torch.from_numpy(key % n)


@pytest.mark.parametrize("mode", ["move", "evict", "pull"])
@pytest.mark.parametrize("n_labels", [7, 300, 5000])
def test_label_scores_kernel_equals_torch_sweep(dev, mode, n_labels):
    rng = np.random.default_rng(n_labels)
    n = 6000
    src, dst = random_graph(rng, n, 12, hubs=12, hub_deg=900)       # hubs: several chunks, hundreds of distinct labels
    src, dst = src.to(dev), dst.to(dev)
    w = torch.from_numpy(rng.integers(1, 2 ** 33, src.numel())).to(dev)      # weights beyond 32 bits
    nw = torch.from_numpy(rng.integers(1, 4, n)).to(dev)
    labels = torch.from_numpy(rng.integers(0, n_labels, n)).to(dev)
    labels[:50] = torch.arange(50, device=dev) % n_labels
    deg = torch.bincount(src, minlength=n)
    assert int(deg.max()) > 64 * 8 and int((deg == 0).sum()) >= 0 and int(((deg > 0) & (deg <= 64)).sum()) > 1000
    parity = P._parity(n, 1, dev)
    mean = int(nw.sum()) // n_labels
    kw = {"move": {}, "evict": {"evict_to": mean}, "pull": {"pull_to": mean}}[mode]
    cap = mean + 5
    rp = P._rowptr(src, n)
    assert rp is not None
    out_k, moved_k = P._sweep(labels, src, dst, w, nw, cap, n_labels, parity, rowptr=rp, **kw)
    out_t, moved_t = P._sweep(labels, src, dst, w, nw, cap, n_labels, parity, rowptr=None, **kw)
    assert moved_k == moved_t and torch.equal(out_k, out_t)
    assert moved_k > 0
    # the scores themselves
    size = torch.zeros(n_labels, dtype=torch.int64, device=dev).scatter_add_(0, labels, nw)
    ok = None if mode == "move" else (size < mean)
    best, best_lab, cur = P._label_scores_kernel(labels, rp, dst, w, n_labels, ok, mode != "move")
    A = torch.zeros(n, n_labels, dtype=torch.int64, device=dev)
    A.index_put_((src, labels[dst]), w, accumulate=True)
    assert torch.equal(cur, A[torch.arange(n, device=dev), labels])
    cand = A.clone()
    if mode != "move":
        cand[torch.arange(n, device=dev), labels] = 0
        cand[:, ~ok] = 0
    want_best = cand.max(dim=1).values
    assert torch.equal(best, want_best)
    first = torch.where(cand == want_best[:, None], torch.arange(n_labels, device=dev)[None, :], n_labels).min(dim=1).values
    assert torch.equal(best_lab, torch.where(want_best > 0, first, torch.full_like(first, n_labels)))


def test_more_distinct_labels_than_the_table_gives_no_proposal(dev):
    n = 3000
    hub = torch.zeros(n - 1, dtype=torch.int64)
    spokes = torch.arange(1, n, dtype=torch.int64)
    src = torch.cat([hub, spokes]).to(dev)
    dst = torch.cat([spokes, hub]).to(dev)
    w = torch.ones_like(src)
    labels = torch.arange(n, device=dev)            # 2999 distinct labels around node 0
    rp = P._rowptr(src, n)
    best, best_lab, cur = P._label_scores_kernel(labels, rp, dst, w, n, None, False)
    assert int(best[0]) == 0 and int(best_lab[0]) == n and int(cur[0]) == 0
    assert torch.equal(best[1:], torch.ones(n - 1, dtype=torch.int64, device=dev)) and bool((best_lab[1:] == 0).all())


def test_partitioner_on_the_device(dev):
    rng = np.random.default_rng(0)
    n, k = 40000, 80
    lab = np.repeat(np.arange(k), n // k)
    e = 400000
    u = rng.integers(0, n, e)
    same = rng.random(e) < 0.9
    v = np.where(same, lab[u] * (n // k) + rng.integers(0, n // k, e), rng.integers(0, n, e))
    keep = u != v
    key = np.unique(np.concatenate([u[keep] * n + v[keep], v[keep] * n + u[keep]]))
    src, dst = torch.from_numpy(key // n).to(dev), torch.from_numpy(key % n).to(dev)
    part = P.kway_partition(n, src, dst, k)
    sizes = torch.bincount(part, minlength=k)
    cap = max(int(np.ceil(n / k * 1.03)), int(np.ceil(n / k)) + 1)
    assert part.shape == (n,) and int(sizes.sum()) == n and int(sizes.min()) > 0 and int(sizes.max()) <= cap
    planted_cut = P.edge_cut(torch.from_numpy(lab).to(dev), src, dst)
    assert P.edge_cut(part, src, dst) <= 1.15 * planted_cut
    again = P.kway_partition(n, src, dst, k)
    assert torch.equal(part, again)                 # deterministic
