"""Parity of the LDS-tiled cluster product (spmm_tiled) against the CPU oracle, on fixtures that
provably reach it: every test asserts that the batch carries a tiling and that the tiling holds the
row classes the kernel treats differently (second window onwards, 17-32 / 33-64 / 65-128 / >128
in-window entries, 1 / 2 / more than 2 entries outside the window, padding descriptors).

Reference semantics: affinity src/SpectralTrainer.py:284-293, loss :130-137, on a multi-cluster batch
as src/utils.py:76-90 concatenates it.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import sparc_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def windowed_pattern(rng, sizes, cross_per_row=0.3, hubs=6, hub_out=5, dense_tail=True):
    """Symmetric 0/1/2-weighted adjacency + identity of a batch of clusters (`sizes` rows each) whose
    in-cluster degrees span one entry to a few hundred, with a few entries between clusters:
    `cross_per_row` expected per row plus `hubs` rows with `hub_out` of them."""
    m = int(sum(sizes))
    start = np.concatenate([[0], np.cumsum(sizes)])
    rows, cols = [], []
    for c, sz in enumerate(sizes):
        if sz < 2:
            continue
        u = rng.random(sz)
        # target degrees: most rows short, a tail up to ~60 % of the cluster
        t = np.where(u < 0.55, rng.integers(1, 14, sz),
            np.where(u < 0.72, rng.integers(15, 30, sz),
            np.where(u < 0.86, rng.integers(30, 60, sz),
            np.where(u < 0.95, rng.integers(60, 120, sz), rng.integers(120, 330, sz)))))
        if not dense_tail:
            t = np.minimum(t, 12)
        t = np.minimum(t, max(1, int(0.6 * sz)))
        for i in range(sz):
            nb = rng.choice(sz, size=int(t[i] // 2 + 1), replace=False)
            nb = nb[nb != i]
            rows.append(np.full(len(nb), start[c] + i))
            cols.append(start[c] + nb)
    r, c_ = np.concatenate(rows), np.concatenate(cols)
    n_cross = int(cross_per_row * m / 2)
    cu, cv = rng.integers(0, m, n_cross), rng.integers(0, m, n_cross)
    hub_rows = rng.choice(m, size=hubs, replace=False)
    hu = np.repeat(hub_rows, hub_out)
    hv = rng.integers(0, m, hubs * hub_out)
    r = np.concatenate([r, cu, hu])
    c_ = np.concatenate([c_, cv, hv])
    keep = r != c_
    w = np.where(rng.random(keep.sum()) < 0.3, 2.0, 1.0)
    D = sp.coo_matrix((w, (r[keep], c_[keep])), shape=(m, m)).tocsr()
    D.data[:] = np.minimum(D.data, 2.0)
    D = D.maximum(D.T)
    return sp.csr_matrix(D + sp.eye(m))


def make_support(D, flavour):
    if flavour == "enh":      # unsymmetric values: val != valT
        return O.as_support_tuple(O.row_normalize_diag_enhance(D, 0.01))
    if flavour == "rownorm":
        return O.as_support_tuple(O.row_normalize(sp.csr_matrix((D > 0) * 1.0)))
    return O.as_support_tuple(D)


def tiling_census(b, n_slabs):
    """What the tiling of batch b holds (asserts that there is one)."""
    tiles, ntiles, tile_rows = b.tiles(n_slabs)
    assert tiles is not None and ntiles > 0, "the batch has no tiling: the tiled kernel would not run"
    lo, hi, n_out, _ = b._in_window_ranges()
    inwin = hi - lo
    desc = tiles[:ntiles * 8].cpu().numpy().reshape(-1, 8)
    return {
        "windows": len(b._windows), "max_w0": max(a for a, _ in b._windows),
        "len_17_32": int(((inwin >= 17) & (inwin <= 32)).sum()), "len_33_64": int(((inwin >= 33) & (inwin <= 64)).sum()),
        "len_65_128": int(((inwin >= 65) & (inwin <= 128)).sum()), "len_gt128": int((inwin > 128).sum()),
        "out1": int((n_out == 1).sum()), "out2": int((n_out == 2).sum()), "out_gt2": int((n_out > 2).sum()),
        "loose_wgs": int((desc[:, 4] < 0).sum()), "pad_desc": int(((desc[:, 1] == desc[:, 0]) & (desc[:, 4] >= 0)).sum()),
        "slabs": int(desc[:, 4].max()) + 1, "ndesc": ntiles, "tile_rows": tile_rows,
    }


SIZES = [400, 110, 380, 300, 200, 450, 60, 512, 333, 1, 96, 280]   # 12 clusters, 3122 rows, >= 7 windows


@pytest.fixture(scope="module")
def big_fixture():
    rng = np.random.default_rng(2024)
    D = windowed_pattern(rng, SIZES)
    out = {}
    for flavour in ("enh", "raw"):
        s = make_support(D, flavour)
        A, At = O.csr_from_support(s)
        out[flavour] = (s, A, At, O.affinity_prep_operator(A))
    return out


@pytest.mark.parametrize("flavour", ["enh", "raw"])
@pytest.mark.parametrize("k", [20, 32, 47, 48, 64])
def test_tiled_lap_matches_oracle_all_row_classes(dev, big_fixture, flavour, k):
    from sparc_amd.graph import ClusterBatch
    s, A, _, prep = big_fixture[flavour]
    m = A.shape[0]
    rng = np.random.default_rng(k)
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, prep, Y.astype(np.float64))
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    assert (b.valT is None) == (flavour == "raw")
    cen = tiling_census(b, b.train_slabs(k))
    assert cen["windows"] >= 5 and cen["max_w0"] > 0
    for key in ("len_17_32", "len_33_64", "len_65_128", "len_gt128", "out1", "out2", "out_gt2", "loose_wgs", "pad_desc"):
        assert cen[key] > 0, (key, cen)
    Yd = torch.tensor(Y, device=dev)
    loss, grad = b.lap(Yd, True)
    assert rel(loss.item(), loss_o) < TOL, cen
    assert rel(grad.cpu().numpy(), g_o) < TOL, cen
    # rows of every class individually (a wrong long row would hide behind the largest gradient entry)
    gn = np.abs(g_o).max(axis=1)
    err_rows = np.abs(grad.cpu().numpy() - g_o).max(axis=1) / np.maximum(gn, 1e-3 * gn.max())
    assert err_rows.max() < 5e-4, int(err_rows.argmax())
    if k == 64:   # forward-only (validation) variant of the tiled kernel
        tiling_census(b, 1)
        lv, none = b.lap(Yd, False)
        assert none is None and rel(lv.item(), loss_o) < TOL
    # the plain row kernel on the same batch: both are oracle-checked, and agree closely
    try:
        ClusterBatch.use_tiles = False
        loss_r, grad_r = b.lap(Yd, True)
    finally:
        ClusterBatch.use_tiles = True
    assert rel(loss_r.item(), loss.item()) < 1e-5 and rel(grad_r.cpu().numpy(), grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("flavour", ["enh", "raw"])
def test_tiled_spmm_matches_scipy(dev, big_fixture, flavour):
    from sparc_amd.graph import ClusterBatch
    s, A, At, _ = big_fixture[flavour]
    m = A.shape[0]
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    cen = tiling_census(b, 1)
    assert cen["len_gt128"] > 0 and cen["out2"] > 0 and cen["loose_wgs"] > 0
    X = np.random.default_rng(1).standard_normal((m, 64)).astype(np.float32)
    Xd = torch.tensor(X, device=dev)
    assert rel(b.spmm(Xd, tiled=True).cpu().numpy(), A @ X.astype(np.float64)) < 1e-5
    assert rel(b.spmm(Xd, add=Xd, transpose=True, tiled=True).cpu().numpy(), X + At @ X.astype(np.float64)) < 1e-5
    # row by row: every row class
    Z = b.spmm(Xd, tiled=True).cpu().numpy()
    ref = A @ X.astype(np.float64)
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-3)
    assert (np.abs(Z - ref).max(axis=1) / scale).max() < 1e-4


@pytest.mark.parametrize("hint", ["clusters", "none", "halves"])
def test_tiled_windows_that_cut_clusters(dev, hint):
    """Windows that do not follow the clusters (no hint: equal cuts; two oversized clusters: split evenly)
    put many entries outside their row's window; short rows keep the batch under the planner's limit."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(11)
    sizes = [500, 480, 510, 470, 490]
    D = windowed_pattern(rng, sizes, cross_per_row=0.05, hubs=3, dense_tail=False)
    m = D.shape[0]
    s = make_support(D, "enh")
    A, _ = O.csr_from_support(s)
    k = 47
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    cs = {"clusters": sizes, "none": None, "halves": [m // 2, m - m // 2]}[hint]
    b = ClusterBatch.from_support(s, dev, cluster_sizes=cs)
    loss, grad = b.lap(torch.tensor(Y, device=dev), True)
    assert rel(loss.item(), loss_o) < TOL and rel(grad.cpu().numpy(), g_o) < TOL
    if hint != "halves":   # "halves" cuts every cluster: no tiling, the row kernel runs
        tiling_census(b, b.train_slabs(k))


def test_untileable_batch_falls_back_to_row_kernel(dev):
    """A batch without cluster structure has no tiling (more than 5 % of its rows reach outside their
    window more than twice): lap() then runs the row kernel, which must match the oracle too."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(3)
    m, k = 1500, 47
    R = sp.random(m, m, density=0.01, random_state=np.random.RandomState(5), format="csr")
    R.data[:] = 1.0
    D = sp.csr_matrix(sp.triu(R, 1) + sp.triu(R, 1).T + sp.eye(m))
    s = make_support(D, "enh")
    b = ClusterBatch.from_support(s, dev, cluster_sizes=[500, 400, 600])
    assert b.tiles(b.train_slabs(k))[0] is None
    A, _ = O.csr_from_support(s)
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    loss, grad = b.lap(torch.tensor(Y, device=dev), True)
    assert rel(loss.item(), loss_o) < TOL and rel(grad.cpu().numpy(), g_o) < TOL


def _shape_batch(dev, workload):
    """One training batch with the statistics of a BASELINE shape without generating the whole graph: twice
    the batch's clusters at their full size and in-cluster edge count, and as many uniform edges as put the
    full graph's number of cross-cluster entries into the batch (a uniform edge of the full graph has both
    ends in one batch with probability (m/n)^2, of this graph with probability 1/4)."""
    from sparc_amd import synth
    from sparc_amd.graph import ClusterBatch
    shape = synth.SHAPES[workload]
    n_clusters = 2 * shape["bsize"]
    n = int(shape["n"] * n_clusters / shape["clusters"])
    e_in = 0.7 * shape["edges"] * n_clusters / shape["clusters"]
    e_cross = 4 * 0.3 * shape["edges"] * (shape["bsize"] / shape["clusters"]) ** 2
    graph, bounds = synth.planted_cluster_graph(n, int(e_in + e_cross), n_clusters, dev,
                                                frac_in_cluster=e_in / (e_in + e_cross), seed=0)
    nodes_of, cluster_ids = synth.cluster_batches(bounds, shape["bsize"], seed=0, return_clusters=True)
    lam = shape["diag_lambda"]
    coords, values, shp, _ = graph.subblock(nodes_of[0], mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
    csz = np.diff(bounds.cpu().numpy())[cluster_ids[0]]
    return ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz), shape["k"]


@pytest.mark.parametrize("workload,min_m,min_nnz", [("amazon2m", 24000, 600_000)])
def test_full_size_batch_tiled_equals_row_kernel(dev, workload, min_m, min_nnz):
    """The benched batch shapes: the tiled kernel against the (oracle-checked) row kernel, plus the
    size-independent properties of the loss."""
    from sparc_amd.graph import ClusterBatch
    b, k = _shape_batch(dev, workload)
    assert b.m >= min_m and b.nnz >= min_nnz, (b.m, b.nnz)
    cen = tiling_census(b, b.train_slabs(k))
    assert cen["windows"] >= 16 and cen["len_33_64"] > 0, cen
    gen = torch.Generator(device=dev).manual_seed(1)
    Y = torch.randn(b.m, k, device=dev, generator=gen)
    loss, grad = b.lap(Y, True)
    lv, _ = b.lap(Y, False)
    try:
        ClusterBatch.use_tiles = False
        loss_r, grad_r = b.lap(Y, True)
    finally:
        ClusterBatch.use_tiles = True
    assert rel(loss.item(), loss_r.item()) < 1e-5
    assert rel(grad.cpu().numpy(), grad_r.cpu().numpy()) < 1e-5
    assert rel(lv.item(), loss_r.item()) < 1e-5
    assert loss.item() >= 0
    inner = (grad.double() * Y.double()).sum().item()       # gradient of a quadratic form: <grad, Y> = 2 loss
    assert abs(inner - 2 * loss.item()) / loss.item() < 1e-4
