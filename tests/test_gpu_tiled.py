"""Parity of the windowed (LDS-tiled) cluster product (spmm_win) against the CPU oracle, on fixtures that
provably reach it: every test asserts that the batch carries a tiling and that the tiling holds the
row classes the kernel treats differently (second window onwards; 17-32 / 33-64 / 65-128 in-window
entries: one or two chunks of a four-row slot; > 128 and > 256: whole-wave rows with one or more chunks;
1, 2, 5-16, > 16 entries outside the window, also on whole-wave rows; padding descriptors).

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


def windowed_pattern(rng, sizes, cross_per_row=0.3, hubs=6, hub_out=5, dense_tail=True, big_hubs=0):
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
    if big_hubs:   # two rows linked to most of their (largest) cluster: more than 256 in-window entries
        c = int(np.argmax(sizes))
        for i in range(2):
            nb = rng.choice(sizes[c], size=int(0.8 * sizes[c]), replace=False)
            r = np.concatenate([r, np.full(len(nb), start[c] + 3 + i)])
            c_ = np.concatenate([c_, start[c] + nb])
    if big_hubs:   # the rows with the most in-cluster entries also get many entries all over the batch
        deg = np.bincount(r, minlength=m)
        top = np.argsort(-deg)[:big_hubs]
        for i, h in enumerate(top):
            far = rng.choice(m, size=30 + 60 * i, replace=False)
            r = np.concatenate([r, np.full(len(far), h)])
            c_ = np.concatenate([c_, far])
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


def tiling_census(b, k, training):
    """What the tiling of batch b for this pass holds (asserts that there is one)."""
    t = b.tiling(k, training)
    assert t is not None and t.ndesc > 0, "the batch has no tiling: the windowed kernel would not run"
    p0, lo, hi, p1 = b._in_window_ranges()
    inwin, n_out = hi - lo, (lo - p0) + (p1 - hi)
    desc = t.desc
    return {
        "windows": len(b._windows), "max_w0": max(a for a, _ in b._windows),
        "len_17_32": int(((inwin >= 17) & (inwin <= 32)).sum()), "len_33_64": int(((inwin >= 33) & (inwin <= 64)).sum()),
        "len_65_128": int(((inwin >= 65) & (inwin <= 128)).sum()), "len_gt128": int((inwin > 128).sum()),
        "len_gt256": int((inwin > 256).sum()),
        "out1": int((n_out == 1).sum()), "out2": int((n_out == 2).sum()), "out_5_16": int(((n_out > 4) & (n_out <= 16)).sum()),
        "out_gt16": int((n_out > 16).sum()), "long_with_out": int(((inwin > 128) & (n_out > 0)).sum()),
        "long_out_gt64": int(((inwin > 128) & (n_out > 64)).sum()),
        "before": int(((lo - p0) > 0).sum()), "after": int(((p1 - hi) > 0).sum()),
        "pad_desc": int((desc[:, 1] == desc[:, 0]).sum()),
        "slabs": int(desc[:, 4].max()) + 1, "ndesc": t.ndesc, "tile_rows": t.rows, "vec": t.vec, "kp": t.kp,
    }


SIZES = [400, 110, 380, 300, 200, 450, 60, 512, 333, 1, 96, 280]   # 12 clusters, 3122 rows, >= 7 windows


@pytest.fixture(scope="module")
def big_fixture():
    rng = np.random.default_rng(2024)
    D = windowed_pattern(rng, SIZES, big_hubs=3)
    out = {}
    for flavour in ("enh", "raw"):
        s = make_support(D, flavour)
        A, At = O.csr_from_support(s)
        out[flavour] = (s, A, At, O.affinity_prep_operator(A))
    return out


@pytest.mark.parametrize("flavour", ["enh", "raw"])
@pytest.mark.parametrize("k", [7, 20, 32, 47, 48, 64])
@pytest.mark.parametrize("vec", [2, 4])
def test_tiled_lap_matches_oracle_all_row_classes(dev, big_fixture, flavour, k, vec, monkeypatch):
    from sparc_amd.graph import ClusterBatch
    s, A, _, prep = big_fixture[flavour]
    m = A.shape[0]
    rng = np.random.default_rng(k)
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, prep, Y.astype(np.float64))
    monkeypatch.setattr(ClusterBatch, "tile_vec", vec)
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    assert (b.valT is None) == (flavour == "raw")
    cen = tiling_census(b, k, True)
    assert cen["vec"] == vec and cen["kp"] >= k and cen["kp"] - k < 8 * vec
    assert cen["windows"] >= 5 and cen["max_w0"] > 0
    for key in ("len_17_32", "len_33_64", "len_65_128", "len_gt128", "len_gt256", "out1", "out2", "out_5_16", "out_gt16",
                "long_with_out", "long_out_gt64", "before", "after", "pad_desc"):
        assert cen[key] > 0, (key, cen)
    Yd = torch.tensor(Y, device=dev)
    loss, grad = b.lap(Yd, True)
    assert rel(loss.item(), loss_o) < TOL, cen
    assert rel(grad.cpu().numpy(), g_o) < TOL, cen
    # rows of every class individually (a wrong long row would hide behind the largest gradient entry)
    gn = np.abs(g_o).max(axis=1)
    err_rows = np.abs(grad.cpu().numpy() - g_o).max(axis=1) / np.maximum(gn, 1e-3 * gn.max())
    assert err_rows.max() < 5e-4, int(err_rows.argmax())
    if k > 16:   # forward-only (validation) variant of the windowed kernel
        tiling_census(b, k, False)
        lv, none = b.lap(Yd, False)
        assert none is None and rel(lv.item(), loss_o) < TOL
    # the plain row kernel on the same batch: both are oracle-checked, and agree closely
    monkeypatch.setattr(ClusterBatch, "use_tiles", False)
    assert b.tiling(k, True) is None
    loss_r, grad_r = b.lap(Yd, True)
    assert rel(loss_r.item(), loss.item()) < 1e-5 and rel(grad_r.cpu().numpy(), grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("flavour", ["enh", "raw"])
@pytest.mark.parametrize("width", [32, 64])
def test_tiled_spmm_matches_scipy(dev, big_fixture, flavour, width):
    from sparc_amd.graph import ClusterBatch
    s, A, At, _ = big_fixture[flavour]
    m = A.shape[0]
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    X = np.random.default_rng(1).standard_normal((m, width)).astype(np.float32)
    Xd = torch.tensor(X, device=dev)
    assert rel(b.spmm(Xd, tiled=True).cpu().numpy(), A @ X.astype(np.float64)) < 1e-5
    assert rel(b.spmm(Xd, add=Xd, transpose=True, tiled=True).cpu().numpy(), X + At @ X.astype(np.float64)) < 1e-5
    # row by row: every row class
    Z = b.spmm(Xd, tiled=True).cpu().numpy()
    ref = A @ X.astype(np.float64)
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-3)
    assert (np.abs(Z - ref).max(axis=1) / scale).max() < 1e-4


@pytest.mark.parametrize("hint", ["clusters", "none", "halves"])
def test_tiled_windows_that_cut_clusters(dev, hint, monkeypatch):
    """Windows that do not follow the clusters (no hint: equal cuts; two oversized clusters: split evenly)
    put many entries outside their row's window; the planner's limit on that share is lifted here so that
    the windowed kernel's outside path carries them."""
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
    monkeypatch.setattr(ClusterBatch, "tile_max_outside_share", 1.0)   # the windowed kernel whatever the share
    b = ClusterBatch.from_support(s, dev, cluster_sizes=cs)
    loss, grad = b.lap(torch.tensor(Y, device=dev), True)
    assert rel(loss.item(), loss_o) < TOL and rel(grad.cpu().numpy(), g_o) < TOL
    cen = tiling_census(b, k, True)   # tiled under every hint; "halves" leaves many entries outside the windows
    if hint == "halves":
        assert cen["out_5_16"] > 100


def test_untileable_batch_falls_back_to_row_kernel(dev):
    """A batch without cluster structure has no tiling (most of its entries lie outside their row's
    window): lap() then runs the row kernel, which must match the oracle too."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(3)
    m, k = 1500, 47
    R = sp.random(m, m, density=0.01, random_state=np.random.RandomState(5), format="csr")
    R.data[:] = 1.0
    D = sp.csr_matrix(sp.triu(R, 1) + sp.triu(R, 1).T + sp.eye(m))
    s = make_support(D, "enh")
    b = ClusterBatch.from_support(s, dev, cluster_sizes=[500, 400, 600])
    assert b.tiling(k, True) is None and b.tiling(k, False) is None
    A, _ = O.csr_from_support(s)
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    loss, grad = b.lap(torch.tensor(Y, device=dev), True)
    assert rel(loss.item(), loss_o) < TOL and rel(grad.cpu().numpy(), g_o) < TOL


@pytest.mark.parametrize("workload,min_m,min_nnz", [("amazon2m", 24000, 600_000), ("reddit", 7500, 700_000)])
def test_full_size_batch_tiled_equals_row_kernel(dev, workload, min_m, min_nnz, monkeypatch):
    """The benched batch shapes: the tiled kernel against the (oracle-checked) row kernel, plus the
    size-independent properties of the loss."""
    from sparc_amd.graph import ClusterBatch
    from sparc_amd import synth
    monkeypatch.setattr(ClusterBatch, "tile_max_outside_share", 1.0)   # Reddit's share is above the planner's limit
    monkeypatch.setattr(ClusterBatch, "use_blocks", False)             # ... and its clusters would run as dense blocks
    b, k = synth.shape_batch(workload, dev)
    assert b.m >= min_m and b.nnz >= min_nnz, (b.m, b.nnz)
    cen = tiling_census(b, k, True)
    assert cen["windows"] >= 16 and cen["len_33_64"] + cen["len_65_128"] + cen["len_gt128"] > 0, cen
    gen = torch.Generator(device=dev).manual_seed(1)
    Y = torch.randn(b.m, k, device=dev, generator=gen)
    loss, grad = b.lap(Y, True)
    lv, _ = b.lap(Y, False)
    monkeypatch.setattr(ClusterBatch, "use_tiles", False)
    loss_r, grad_r = b.lap(Y, True)
    assert rel(loss.item(), loss_r.item()) < 1e-5
    assert rel(grad.cpu().numpy(), grad_r.cpu().numpy()) < 1e-5
    assert rel(lv.item(), loss_r.item()) < 1e-5
    assert loss.item() >= 0
    inner = (grad.double() * Y.double()).sum().item()       # gradient of a quadratic form: <grad, Y> = 2 loss
    assert abs(inner - 2 * loss.item()) / loss.item() < 1e-4


def test_kernel_choice_by_share_and_by_measurement(dev, monkeypatch):
    """The planner's rule (share of entries outside the windows) sends the Amazon2M-shape batch to the
    windowed kernel and the Reddit-shape batch to the row kernel; warm(measure=True) times both kernels on
    a batch that has a tiling and keeps the faster, and lap() then follows that choice."""
    from sparc_amd.graph import ClusterBatch
    from sparc_amd import synth
    ba, ka = synth.shape_batch("amazon2m", dev)
    br, kr = synth.shape_batch("reddit", dev)
    assert ba.tiling(ka, True) is not None and br.tiling(kr, True) is None
    ba.warm(ka, True, measure=True)
    choice = ba._measured[(ka, True)]
    assert choice in ("tiles", "rows") and set(ba.measured_ms) == {"tiles", "rows"}
    assert choice == min(ba.measured_ms, key=ba.measured_ms.get)
    assert (ba.tiling(ka, True) is None) == (choice == "rows")
    monkeypatch.setattr(ClusterBatch, "tile_max_outside_share", 1.0)
    br2, _ = synth.shape_batch("reddit", dev)
    br2.warm(kr, True, measure=True)
    Y = torch.randn(br2.m, kr, device=dev)
    l1, g1 = br2.lap(Y, True)                    # whichever kernel was kept
    l0, g0 = br.lap(Y, True)                     # the row kernel
    assert rel(l1.item(), l0.item()) < 1e-5 and rel(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-5


# ---- pattern mode: supports built from a 0/1 adjacency, whose off-diagonal values are row factors -------------
@pytest.fixture(scope="module")
def binary_fixture():
    rng = np.random.default_rng(77)
    D = windowed_pattern(rng, SIZES, big_hubs=3)
    D = sp.csr_matrix((D > 0) * 1.0)                    # 0/1 adjacency + identity
    nodiag = sp.csr_matrix(D - sp.diags(D.diagonal()))
    out = {}
    for flavour, M in (("raw_bin", D), ("rownorm_bin", O.row_normalize(D)), ("enh_bin", O.row_normalize_diag_enhance(D, 0.01)),
                       ("nodiag_bin", nodiag)):
        s = O.as_support_tuple(sp.csr_matrix(M))
        A, At = O.csr_from_support(s)
        out[flavour] = (s, A, At, O.affinity_prep_operator(A))
    return out


@pytest.mark.parametrize("flavour", ["raw_bin", "rownorm_bin", "enh_bin", "nodiag_bin"])
@pytest.mark.parametrize("k", [7, 32, 47, 64])
def test_pattern_mode_lap_matches_oracle(dev, binary_fixture, flavour, k, monkeypatch):
    """Weight-free gathers (row factors in the epilogue, A^T half of the operand pre-multiplied): every row class of
    the windowed kernel again, on the three support flavours of the reference built from a 0/1 adjacency and on
    a support without stored diagonal; against the oracle, and against the weighted streams on the same batch."""
    from sparc_amd.graph import ClusterBatch
    s, A, _, prep = binary_fixture[flavour]
    m = A.shape[0]
    rng = np.random.default_rng(k)
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, prep, Y.astype(np.float64))
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    rf = b.row_factors()
    assert rf is not None and rf.shape == (m, 4)
    if flavour == "enh_bin":
        assert float(rf[:, 1].abs().max()) > 0          # the enhanced diagonal is not the row factor
    cen = tiling_census(b, k, True)
    assert b.tiling(k, True).pattern
    for key in ("len_17_32", "len_33_64", "len_65_128", "len_gt128", "len_gt256", "out1", "out_5_16", "out_gt16", "long_with_out"):
        assert cen[key] > 0, (key, cen)
    Yd = torch.tensor(Y, device=dev)
    loss, grad = b.lap(Yd, True)
    assert rel(loss.item(), loss_o) < TOL, cen
    assert rel(grad.cpu().numpy(), g_o) < TOL, cen
    gn = np.abs(g_o).max(axis=1)
    err_rows = np.abs(grad.cpu().numpy() - g_o).max(axis=1) / np.maximum(gn, 1e-3 * gn.max())
    assert err_rows.max() < 5e-4, int(err_rows.argmax())
    if k > 16:
        assert b.tiling(k, False).pattern
        lv, none = b.lap(Yd, False)
        assert none is None and rel(lv.item(), loss_o) < TOL
    monkeypatch.setattr(ClusterBatch, "use_pattern", False)     # the same batch through the weight streams
    assert not b.tiling(k, True).pattern
    loss_w, grad_w = b.lap(Yd, True)
    assert rel(loss_w.item(), loss.item()) < 1e-5 and rel(grad_w.cpu().numpy(), grad.cpu().numpy()) < 2e-5


def test_weighted_supports_have_no_row_factors(dev, big_fixture):
    from sparc_amd.graph import ClusterBatch
    for flavour in ("enh", "raw"):                      # weights 1 and 2 inside a row
        b = ClusterBatch.from_support(big_fixture[flavour][0], dev, cluster_sizes=SIZES)
        assert b.row_factors() is None and not b.tiling(47, True).pattern
