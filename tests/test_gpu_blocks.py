"""Dense diagonal blocks on the matrix cores (sparc_amd/csrc/blocks.hip, sparc_lap_fwd_blocks): batches whose
clusters are small, nearly complete sub-graphs.  Checked against SciPy (the block product alone), the CPU oracle
(loss and gradient of the whole pass, src/SpectralTrainer.py:116-139 and :284-293) and the row kernel."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import sparc_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


SIZES = [155, 32, 1, 97, 160, 33, 128, 190, 64, 7]      # cluster rows: below, at and above multiples of 32, one single row


def dense_cluster_graph(rng, sizes, fill=0.7, cross_per_row=3.0):
    """Symmetric 0/1/2 adjacency + identity: every cluster a random graph with `fill` of its pairs connected,
    plus a few entries between clusters."""
    m = int(sum(sizes))
    start = np.concatenate([[0], np.cumsum(sizes)])
    blocks = []
    for sz in sizes:
        B = (rng.random((sz, sz)) < fill) * np.where(rng.random((sz, sz)) < 0.3, 2.0, 1.0)
        blocks.append(np.triu(B, 1))
    D = sp.block_diag(blocks, format="csr")
    n_cross = int(cross_per_row * m / 2)
    cu, cv = rng.integers(0, m, n_cross), rng.integers(0, m, n_cross)
    keep = cu != cv
    C = sp.coo_matrix((np.ones(keep.sum()), (np.minimum(cu, cv)[keep], np.maximum(cu, cv)[keep])), shape=(m, m)).tocsr()
    C.data[:] = 1.0
    U = sp.triu(D + C, 1)
    U.data[:] = np.minimum(U.data, 2.0)
    return sp.csr_matrix(U + U.T + sp.eye(m)), start


def make_support(D, flavour):
    if flavour == "enh":      # unsymmetric values: val != valT
        return O.as_support_tuple(O.row_normalize_diag_enhance(D, 0.01))
    return O.as_support_tuple(D)


@pytest.mark.parametrize("width", [32, 64, 96, 128])
def test_block_product_matches_scipy(dev, width):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(width)
    D, start = dense_cluster_graph(rng, SIZES)
    s = make_support(D, "enh")
    A, _ = O.csr_from_support(s)
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    assert b.blocks(64, True) is not None and b._blocks["fill"] > 0.5
    cl = np.repeat(np.arange(len(SIZES)), SIZES)
    Ac = A.tocoo()
    same = cl[Ac.row] == cl[Ac.col]
    Dblk = sp.csr_matrix((Ac.data[same], (Ac.row[same], Ac.col[same])), shape=A.shape)
    assert b._blocks["r_nnz"] == int((~same).sum()) and b._blocks["ndesc"] == sum((n + 31) // 32 for n in SIZES)
    X = rng.standard_normal((A.shape[0], width)).astype(np.float32)
    Xd = torch.tensor(X, device=dev)
    want = Dblk @ X.astype(np.float64)
    assert rel(b.block_spmm(Xd).cpu().numpy(), want) < 2e-6
    assert rel(b.block_spmm(Xd, add=Xd).cpu().numpy(), want + X) < 2e-6
    scale = np.maximum(np.abs(want).max(axis=1), 1e-3)
    assert (np.abs(b.block_spmm(Xd).cpu().numpy() - want).max(axis=1) / scale).max() < 2e-5      # row by row


@pytest.mark.parametrize("flavour", ["raw", "enh"])
@pytest.mark.parametrize("k", [9, 16, 32, 47, 64])
def test_block_lap_matches_oracle(dev, flavour, k, monkeypatch):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(100 + k)
    D, _ = dense_cluster_graph(rng, SIZES)
    s = make_support(D, flavour)
    A, _ = O.csr_from_support(s)
    m = A.shape[0]
    Y = rng.standard_normal((m, k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    b = ClusterBatch.from_support(s, dev, cluster_sizes=SIZES)
    assert (b.valT is None) == (flavour == "raw")
    Yd = torch.tensor(Y, device=dev)
    if flavour == "enh" and k <= 16:              # mirrored values: the two halves (16 columns each) would share a 32-column tile
        assert b.blocks(k, True) is None
        return
    assert b.blocks(k, True) is not None          # training width 2 kp is a multiple of 32 for every k
    loss, grad = b.lap(Yd, True)
    assert rel(loss.item(), loss_o) < 1e-4 and rel(grad.cpu().numpy(), g_o) < 1e-4
    gn = np.abs(g_o).max(axis=1)
    err_rows = np.abs(grad.cpu().numpy() - g_o).max(axis=1) / np.maximum(gn, 1e-3 * gn.max())
    assert err_rows.max() < 5e-4, int(err_rows.argmax())
    has_fwd = b.blocks(k, False) is not None      # forward-only width kp: 32 or 64 only
    assert has_fwd == (k > 16)
    lv, none = b.lap(Yd, False)
    assert none is None and rel(lv.item(), loss_o) < 1e-4
    monkeypatch.setattr(ClusterBatch, "use_blocks", False)
    assert b.blocks(k, True) is None
    loss_r, grad_r = b.lap(Yd, True)              # the row kernel on the whole matrix
    assert rel(loss_r.item(), loss.item()) < 1e-5 and rel(grad_r.cpu().numpy(), grad.cpu().numpy()) < 2e-5


def test_sparse_or_large_clusters_have_no_blocks(dev):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(3)
    D, _ = dense_cluster_graph(rng, [150, 160, 140], fill=0.1)
    b = ClusterBatch.from_support(make_support(D, "raw"), dev, cluster_sizes=[150, 160, 140])
    assert b.blocks(64, True) is None             # 10 % fill: a gather moves fewer bytes
    D, _ = dense_cluster_graph(rng, [300, 200], fill=0.7)
    b = ClusterBatch.from_support(make_support(D, "raw"), dev, cluster_sizes=[300, 200])
    assert b.blocks(64, True) is None             # a 300-row cluster does not fit LDS with its operand
    b = ClusterBatch.from_support(make_support(D, "raw"), dev)
    assert b.blocks(64, True) is None             # no cluster sizes


def test_reddit_shape_batch_blocks_equal_row_kernel(dev, monkeypatch):
    """The Reddit-shape batch of the bench: dense blocks + remainder against the (oracle-checked) row kernel,
    the size-independent loss properties, and the measured choice."""
    from sparc_amd.graph import ClusterBatch
    from sparc_amd import synth
    b, k = synth.shape_batch("reddit", dev)
    bl = b.blocks(k, True)
    assert bl is not None and bl["fill"] > 0.5 and bl["r_nnz"] < 0.4 * b.nnz, (bl and bl["fill"], bl and bl["r_nnz"], b.nnz)
    gen = torch.Generator(device=dev).manual_seed(1)
    Y = torch.randn(b.m, k, device=dev, generator=gen)
    loss, grad = b.lap(Y, True)
    lv, _ = b.lap(Y, False)
    b.warm(k, True, measure=True)
    assert set(b.measured_ms) == {"blocks", "rows"} and b._measured[(k, True)] == min(b.measured_ms, key=b.measured_ms.get)
    monkeypatch.setattr(ClusterBatch, "use_blocks", False)
    b._measured.clear()
    loss_r, grad_r = b.lap(Y, True)
    assert rel(loss.item(), loss_r.item()) < 1e-5
    assert rel(grad.cpu().numpy(), grad_r.cpu().numpy()) < 2e-5
    assert rel(lv.item(), loss_r.item()) < 1e-5
    inner = (grad.double() * Y.double()).sum().item()       # gradient of a quadratic form: <grad, Y> = 2 loss
    assert abs(inner - 2 * loss.item()) / loss.item() < 1e-4


def test_block_diagonal_batch_without_other_entries(dev):
    """A batch whose clusters are not connected at all: the CSR of the other entries is empty (every row)."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(9)
    sizes = [40, 64, 17]
    D, _ = dense_cluster_graph(rng, sizes, cross_per_row=0.0)
    s = make_support(D, "raw")
    A, _ = O.csr_from_support(s)
    k = 20
    Y = rng.standard_normal((A.shape[0], k)).astype(np.float32)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    b = ClusterBatch.from_support(s, dev, cluster_sizes=sizes)
    bl = b.blocks(k, True)
    assert bl is not None and bl["r_nnz"] == 0
    loss, grad = b.lap(torch.tensor(Y, device=dev), True)
    assert rel(loss.item(), loss_o) < 1e-4 and rel(grad.cpu().numpy(), g_o) < 1e-4
