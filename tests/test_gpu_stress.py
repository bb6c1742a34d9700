"""The plain whole-graph product Z = A X (the SpMM roofline stress case: an R-MAT graph of 10 M nodes and 500 M edges at full size) at a size SciPy
checks in a second: a scaled-down R-MAT graph (same quadrant probabilities, power-law degrees, hub rows
of thousands of entries) in natural order and relabelled by decreasing degree, through the C ABI."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def _spmm(lib, n, rowptr, colidx, val, X, into=None):
    """Z = A X, or into += A X in place (the own-row operand X0 of the C ABI is the result itself)."""
    Z = torch.empty_like(X) if into is None else into
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.sparc_spmm(n, int(colidx.numel()), P(rowptr), P(colidx), P(val), int(X.shape[1]), P(X),
                        None if into is None else P(into), P(Z), None,
                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return Z


@pytest.mark.parametrize("width", [64, 128])
def test_rmat_product_matches_scipy_in_both_orders(width):
    from sparc_amd import synth
    from sparc_amd._lib import lib
    dev = torch.device("cuda:0")
    n, e = 20000, 400000                      # not a power of two: exercises the rejection of ids >= n
    rowptr, colidx = synth.rmat_graph(n, e, dev, seed=3)
    assert int(colidx.numel()) == 2 * e + n   # exactly the named number of edges, plus the unit diagonal
    deg = (rowptr[1:] - rowptr[:-1]).cpu().numpy()
    assert deg.max() > 50 * np.median(deg)    # power-law: hub rows far above the typical degree
    val = torch.rand(colidx.numel(), device=dev) + 0.5
    X = torch.randn(n, width, device=dev)
    A = sp.csr_matrix((val.cpu().numpy().astype(np.float64), colidx.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n))
    ref = A @ X.cpu().numpy().astype(np.float64)
    Z = _spmm(lib, n, rowptr, colidx, val, X).cpu().numpy()
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-3)
    assert (np.abs(Z - ref) / scale).max() < 1e-5
    # relabelled by decreasing degree: P A P^T (P X) = P (A X)
    rp2, ci2, perm = synth.degree_order(rowptr, colidx)
    d2 = (rp2[1:] - rp2[:-1]).cpu().numpy()
    assert np.all(np.diff(d2) <= 0) and sorted(d2.tolist()) == sorted(deg.tolist())
    Ap = A[perm.cpu().numpy()][:, perm.cpu().numpy()].tocsr()
    Ap.sort_indices()
    assert np.array_equal(Ap.indptr, rp2.cpu().numpy()) and np.array_equal(Ap.indices, ci2.cpu().numpy())
    val2 = torch.from_numpy(Ap.data.astype(np.float32)).to(dev)
    Z2 = _spmm(lib, n, rp2, ci2, val2, X[perm].contiguous()).cpu().numpy()
    assert (np.abs(Z2 - ref[perm.cpu().numpy()]) / scale[perm.cpu().numpy()]).max() < 1e-5


def test_product_accumulated_in_place_over_column_blocks():
    """The whole-graph product as one pass per block of source nodes, the result accumulated in place (X0 = Z):
    what tools/spmm_stress.py's rmat-blocked case times.  Blocks without entries for most rows, an empty block,
    and rows that are empty in every block included."""
    from sparc_amd import synth
    from sparc_amd._lib import lib
    dev = torch.device("cuda:0")
    n, e, width = 6000, 60000, 64
    rowptr, colidx = synth.rmat_graph(n, e, dev, seed=5)
    val = torch.rand(colidx.numel(), device=dev) + 0.5
    A = sp.csr_matrix((val.cpu().numpy(), colidx.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n)).tolil()
    A[100:140, :] = 0                      # rows without entries
    A[:, 4096:5120] = 0                    # a block of source nodes nobody points to
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    X = torch.randn(n, width, device=dev)
    ref = A.astype(np.float64) @ X.cpu().numpy().astype(np.float64)
    Z, empty_blocks = None, 0
    for lo in range(0, n, 1024):
        mask = (A.indices >= lo) & (A.indices < lo + 1024)
        rows = np.repeat(np.arange(n), np.diff(A.indptr))[mask]
        rp = np.zeros(n + 1, np.int64)
        rp[1:] = np.cumsum(np.bincount(rows, minlength=n))
        args = (torch.from_numpy(rp).to(dev), torch.from_numpy(A.indices[mask].astype(np.int32)).to(dev),
                torch.from_numpy(A.data[mask].astype(np.float32)).to(dev))
        empty_blocks += int(args[1].numel() == 0)   # an empty pattern is a valid argument (null entry pointers)
        Z = _spmm(lib, n, *args, X, into=Z) if Z is not None else _spmm(lib, n, *args, X)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-3)
    assert (np.abs(Z.cpu().numpy() - ref) / scale).max() < 1e-5
    assert np.all(Z.cpu().numpy()[100:140] == 0) and empty_blocks == 1


@pytest.mark.parametrize("width", [64, 128, 16])
def test_split_product_with_hub_rows_matches_scipy(width):
    """sparc_spmm_split: a graph whose first node is linked to 120 000 others (a row of more than 100 k entries, 59
    chunks) among R-MAT rows, several rows just above and just below the limit, through the C ABI against SciPy; the
    accumulation in place (X0 = Z); and the same bits on every run (the chunks are added in their order)."""
    from sparc_amd import synth
    from sparc_amd._lib import lib
    from sparc_amd.graph import RowSplit, spmm_split
    dev = torch.device("cuda:0")
    n, e = 150000, 600000
    rowptr, colidx = synth.rmat_graph(n, e, dev, seed=7)
    A = sp.csr_matrix((np.ones(colidx.numel(), np.float32), colidx.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n)).tolil()
    rng = np.random.default_rng(0)
    hub = rng.choice(np.arange(1, n), size=120000, replace=False)
    A[0, hub] = 1.0
    A[hub, 0] = 1.0
    for r, d in ((5, 2048), (6, 2049), (7, 4097), (8, 2047)):       # at, just above (two chunks), three chunks, below
        A[r, :] = 0
        A[r, rng.choice(n, size=d, replace=False)] = 1.0
    A = A.tocsr()
    A.sort_indices()
    A.data = (rng.random(A.nnz) + 0.5).astype(np.float32)
    deg = np.diff(A.indptr)
    assert deg[0] > 100000 and deg[5] == 2048 and deg[6] == 2049 and deg[7] == 4097
    rp = torch.from_numpy(A.indptr.astype(np.int64)).to(dev)
    ci = torch.from_numpy(A.indices.astype(np.int32)).to(dev)
    val = torch.from_numpy(A.data).to(dev)
    plan = RowSplit(rp, limit=2048)
    assert plan.nlong == int((deg > 2048).sum()) and 0 in plan.long_row.cpu().numpy() and 5 not in plan.long_row.cpu().numpy()
    per_row = np.diff(plan.long_first.cpu().numpy())
    assert per_row[0] == -(-deg[0] // 2048) and plan.nchunks == per_row.sum()
    cb, ce = plan.chunk_begin.cpu().numpy(), plan.chunk_end.cpu().numpy()
    assert (ce - cb).max() <= 2048 and (ce - cb).min() >= 1
    for i, r in enumerate(plan.long_row.cpu().numpy()):             # the chunks of a row tile its entries in order
        a, b = plan.long_first[i].item(), plan.long_first[i + 1].item()
        assert cb[a] == A.indptr[r] and ce[b - 1] == A.indptr[r + 1] and np.array_equal(cb[a + 1:b], ce[a:b - 1])
    X = torch.randn(n, width, device=dev)
    ref = A.astype(np.float64) @ X.cpu().numpy().astype(np.float64)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-3)
    Z = spmm_split(plan, n, rp, ci, val, X)
    assert (np.abs(Z.cpu().numpy() - ref) / scale).max() < 1e-5
    assert torch.equal(Z, spmm_split(plan, n, rp, ci, val, X))      # the same bits, whatever the schedule was
    Zu = _spmm(lib, n, rp, ci, val, X)                              # the unsplit product: one wave per row
    assert (np.abs(Zu.cpu().numpy() - ref) / scale).max() < 1e-5
    acc = X.clone()                                                 # in place: acc = acc + A X
    spmm_split(plan, n, rp, ci, val, X, add=acc, out=acc)
    assert (np.abs(acc.cpu().numpy() - (ref + X.cpu().numpy())) / np.maximum(scale, 1.0)).max() < 1e-5
    none = RowSplit(rp, limit=1 << 20)                              # no row above the limit: the plain row kernel
    assert none.nlong == 0 and none.nchunks == 0
    assert torch.equal(spmm_split(none, n, rp, ci, val, X), Zu)
