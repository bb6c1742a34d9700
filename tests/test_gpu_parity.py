"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference-generated golden vectors.  Tolerances: bit-exact for index/degree work,
1e-4 relative (float32) for embeddings, loss and gradients.
"""
import ctypes

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


def sup(g, prefix):
    return g[prefix + "coords"], g[prefix + "values"], tuple(int(x) for x in g[prefix + "shape"])


def random_support(m, density, rng, flavour="raw", weight2=0.3):
    A = sp.random(m, m, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="csr")
    A.data[:] = np.where(rng.random(A.nnz) < weight2, 2.0, 1.0)
    A = sp.triu(A, 1)
    A = A + A.T + sp.eye(m)
    if flavour == "rownorm":
        A = O.row_normalize(sp.csr_matrix((A > 0) * 1.0))
    elif flavour == "enh":
        A = O.row_normalize_diag_enhance(A, 0.01)
    return O.as_support_tuple(sp.csr_matrix(A))


# ---------------------------------------------------------------- CSR build
@pytest.mark.parametrize("flavour", ["raw", "rownorm", "enh"])
def test_csr_build_exact(dev, flavour):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(1)
    s = random_support(300, 0.03, rng, flavour)
    # shuffle the tuple and duplicate a few entries: the build must sort and merge
    order = rng.permutation(len(s[1]))
    coords = np.concatenate([s[0][order], s[0][order][:7]])
    values = np.concatenate([s[1][order], s[1][order][:7]])
    b = ClusterBatch.from_support((coords, values, s[2]), dev)
    A, At = O.csr_from_support((coords, values, s[2]))
    assert b.nnz == A.nnz
    assert np.array_equal(b.rowptr.cpu().numpy(), A.indptr.astype(np.int64))
    assert np.array_equal(b.colidx.cpu().numpy(), A.indices)
    assert np.array_equal(b.val.cpu().numpy(), A.data.astype(np.float32))
    assert b.valT is not None   # the duplicated entries broke the symmetry
    assert np.array_equal(b.valT.cpu().numpy(), np.asarray(At[A.nonzero()]).ravel().astype(np.float32))
    if flavour == "raw":  # numerically symmetric input: the mirrored values are dropped
        assert ClusterBatch.from_support(s, dev).valT is None


def test_csr_build_unsymmetric_pattern(dev):
    from sparc_amd.graph import ClusterBatch
    coords = np.array([[0, 1], [2, 0], [1, 1], [3, 2]], dtype=np.int32)
    values = np.array([1.5, 2.0, 3.0, 4.0])
    b = ClusterBatch.from_support((coords, values, (4, 4)), dev)
    # pattern is symmetrised with explicit zeros
    dense = np.zeros((4, 4)); denseT = np.zeros((4, 4))
    rp, ci = b.rowptr.cpu().numpy(), b.colidx.cpu().numpy()
    for i in range(4):
        for p in range(rp[i], rp[i + 1]):
            dense[i, ci[p]] = b.val[p].item(); denseT[i, ci[p]] = b.valT[p].item()
    ref = np.zeros((4, 4)); ref[coords[:, 0], coords[:, 1]] = values
    assert np.array_equal(dense, ref) and np.array_equal(denseT, ref.T)
    assert b.nnz == 7


# ---------------------------------------------------------------- affinity prep + loss
@pytest.mark.parametrize("flavour", ["raw", "rownorm", "enh"])
@pytest.mark.parametrize("m", [1, 2, 63, 64, 65, 200, 1000])
def test_affinity_prep_matches_oracle(dev, flavour, m):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(m)
    s = random_support(m, min(1.0, 6.0 / max(m, 1)), rng, flavour)
    b = ClusterBatch.from_support(s, dev)
    A, _ = O.csr_from_support(s)
    if m == 1:
        # a single node: W is the 1x1 zero matrix; the reference divides by the clamped norm
        assert torch.isfinite(b.rho).all()
        return
    prep = O.affinity_prep_operator(A)
    assert rel(b.rho.cpu().numpy(), prep["rho"]) < 1e-5
    assert rel(b.d.cpu().numpy(), prep["d"]) < 1e-5
    assert rel(b.cw.cpu().numpy(), prep["cw"]) < 1e-5
    assert rel(b.rowc[:, 1].cpu().numpy(), prep["sdiag"]) < 1e-6
    assert rel(b.rowc[:, 0].abs().cpu().numpy(), 1.0 / prep["rho"]) < 1e-5
    assert rel(b.coupling[:9].cpu().numpy(), prep["C"].ravel()) < 1e-12


@pytest.mark.parametrize("name", ["raw", "rownorm", "enh"])
def test_affinity_and_loss_match_reference_golden(dev, golden, name):
    from sparc_amd.SpectralTrainer import SpectralNetLoss
    from sparc_amd.graph import ClusterBatch
    g = golden("g2_affinity_loss.npz")
    b = ClusterBatch.from_support(sup(g, name + "_"), dev)
    assert rel(b.d.cpu().numpy(), g[name + "_d"]) < TOL
    assert rel(b.cw.cpu().numpy(), g[name + "_colsum"]) < TOL
    Y = torch.tensor(g["Y"], device=dev)
    loss = SpectralNetLoss()(b, Y, is_normalized=True)
    assert loss.dtype == torch.float32
    assert rel(loss.item(), g[name + "_loss_norm"]) < TOL
    plain = SpectralNetLoss()(b, Y)                      # is_normalized defaults to False, as in the reference
    assert rel(plain.item(), g[name + "_loss_plain"]) < TOL
    # gradient of the plain branch against autograd through the dense oracle
    W = O.valid_affinity(O.support_to_dense(sup(g, name + "_"), torch.float64))
    Yd = torch.tensor(g["Y"], dtype=torch.float64, requires_grad=True)
    (gref,) = torch.autograd.grad(O.spectral_loss(W, Yd, False), Yd)
    Yg = Y.clone().requires_grad_(True)
    SpectralNetLoss()(b, Yg, is_normalized=False).backward()
    assert rel(Yg.grad.cpu().numpy(), gref.numpy()) < TOL


@pytest.mark.parametrize("name", ["raw", "rownorm", "enh"])
def test_loss_accepts_the_dense_affinity_of_the_reference_signature(dev, golden, name):
    """SpectralNetLoss.forward(W, Y, is_normalized) with W the dense m x m affinity, as src/SpectralTrainer.py:116
    receives it from _get_valid_affinity_matrix (:284-293): both branches and the gradient with respect to Y, against
    the losses the reference printed for the same support, and the dense helper against the oracle's dense affinity."""
    from sparc_amd.SpectralTrainer import SpectralNetLoss, SpectralTrainer
    g = golden("g2_affinity_loss.npz")
    A = O.support_to_dense(sup(g, name + "_"), torch.float32)
    W_ref = O.valid_affinity(A.double())
    cfg = {"dataset": "x", "spectral": {"lr": 1e-3, "epochs": 1, "lr_decay": 0.1, "patience": 1, "batch_size": 1,
                                        "architecture": {}, "n_clusters": 1}}
    W = SpectralTrainer(cfg, dev)._get_valid_affinity_matrix(A.to(dev))
    assert W.dtype == torch.float32 and rel(W.cpu().numpy(), W_ref.numpy()) < 1e-5
    assert rel(W.sum(1).cpu().numpy(), g[name + "_d"]) < TOL
    Y = torch.tensor(g["Y"], device=dev, requires_grad=True)
    loss = SpectralNetLoss()(W, Y, is_normalized=True)
    assert loss.dtype == torch.float32 and rel(loss.item(), g[name + "_loss_norm"]) < TOL
    loss.backward()
    Yd = torch.tensor(g["Y"], dtype=torch.float64, requires_grad=True)
    (gref,) = torch.autograd.grad(O.spectral_loss(W_ref, Yd, True), Yd)
    assert rel(Y.grad.cpu().numpy(), gref.numpy()) < TOL
    plain = SpectralNetLoss()(W, Y.detach())
    assert rel(plain.item(), g[name + "_loss_plain"]) < TOL
    with pytest.raises(TypeError):
        SpectralNetLoss()(W[:-1], Y.detach())          # not the batch's m x m matrix
    with pytest.raises(TypeError):
        SpectralNetLoss()("W", Y.detach())


@pytest.mark.parametrize("k", [1, 8, 16, 17, 32, 47, 64])
@pytest.mark.parametrize("flavour", ["raw", "enh"])
def test_lap_loss_and_grad_match_oracle(dev, k, flavour):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(100 + k)
    m = 333
    s = random_support(m, 0.02, rng, flavour)
    b = ClusterBatch.from_support(s, dev)
    A, _ = O.csr_from_support(s)
    prep = O.affinity_prep_operator(A)
    Y = rng.standard_normal((m, k))
    loss_o, g_o, _, _ = O.lap_forward_operator(A, prep, Y.astype(np.float32).astype(np.float64))
    Yd = torch.tensor(Y, dtype=torch.float32, device=dev)
    loss, grad = b.lap(Yd, True)
    assert rel(loss.item(), loss_o) < TOL
    assert rel(grad.cpu().numpy(), g_o) < TOL
    loss1, none = b.lap(Yd, False)   # forward-only (validation) kernel variant
    assert none is None and rel(loss1.item(), loss_o) < TOL
    # dense oracle too (what the reference evaluates)
    W = O.valid_affinity(O.support_to_dense(s, torch.float64))
    ld = O.spectral_loss(W, torch.tensor(Y, dtype=torch.float32).double(), True).item()
    assert rel(loss.item(), ld) < TOL


@pytest.mark.parametrize("k", [20, 32, 47, 64])
def test_lap_with_any_window_hint_matches_oracle(dev, k):
    """Cluster-size hints only steer the windows: with the clusters as windows, with equal cuts and with windows
    that cut across clusters (many entries outside their row's window) the result is the oracle's.  The
    windowed kernel's row classes are covered in test_gpu_tiled.py."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(7 + k)
    sizes = [130, 40, 333, 700, 1, 96]            # one cluster larger than a window
    m = sum(sizes)
    lab = np.repeat(np.arange(len(sizes)), sizes)
    P = np.where(lab[:, None] == lab[None, :], 0.05, 0.002)
    D = np.triu(rng.random((m, m)) < P, 1).astype(np.float64)
    D = D + D.T + np.eye(m)
    s = O.as_support_tuple(O.row_normalize_diag_enhance(sp.csr_matrix(D), 0.01))
    Y = rng.standard_normal((m, k)).astype(np.float32)
    Yd = torch.tensor(Y, device=dev)
    A, _ = O.csr_from_support(s)
    loss_o, g_o, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.astype(np.float64))
    for hint in (sizes, None, [m // 3, m - m // 3]):
        b = ClusterBatch.from_support(s, dev, cluster_sizes=hint)
        assert len(b._windows) >= 2
        loss, grad = b.lap(Yd, True)
        assert rel(loss.item(), loss_o) < TOL and rel(grad.cpu().numpy(), g_o) < TOL
        lv, _ = b.lap(Yd, False)
        assert rel(lv.item(), loss_o) < TOL


def test_lap_isolated_nodes_and_empty_rows(dev):
    from sparc_amd.graph import ClusterBatch
    m = 40
    coords = np.array([[i, i] for i in range(m) if i % 3] + [[1, 2], [2, 1], [4, 5], [5, 4]], dtype=np.int32)
    values = np.ones(len(coords))
    s = (coords, values, (m, m))   # rows 0, 3, 6, ... have no stored entry at all
    b = ClusterBatch.from_support(s, dev)
    Y = np.random.default_rng(0).standard_normal((m, 8))
    W = O.valid_affinity(O.support_to_dense(s, torch.float64))
    Yt = torch.tensor(Y, dtype=torch.float64, requires_grad=True)
    ld = O.spectral_loss(W, Yt, True)
    (gd,) = torch.autograd.grad(ld, Yt)
    loss, grad = b.lap(torch.tensor(Y, dtype=torch.float32, device=dev), True)
    assert rel(loss.item(), ld.item()) < TOL
    assert rel(grad.cpu().numpy(), gd.numpy()) < TOL


def test_one_node_batch_behaves_like_the_reference(dev):
    """A batch of one node: W is the 1 x 1 zero matrix (its only entry is the removed diagonal), so the plain loss is
    exactly zero with a zero gradient, and the normalised loss divides by sqrt(0): nan, for the reference as well
    (src/SpectralTrainer.py:130-137)."""
    from sparc_amd.SpectralTrainer import SpectralNetLoss
    from sparc_amd.graph import ClusterBatch
    s = O.as_support_tuple(sp.csr_matrix(np.eye(1)))
    b = ClusterBatch.from_support(s, dev)
    W = O.valid_affinity(O.support_to_dense(s, torch.float64))
    assert float(W.abs().max()) == 0.0
    for k in (3, 31):
        Y = torch.randn(1, k, device=dev, requires_grad=True)
        plain = SpectralNetLoss()(b, Y, is_normalized=False)
        plain.backward()
        assert float(plain) == 0.0 == float(O.spectral_loss(W, Y.detach().cpu().double(), False))
        assert float(Y.grad.abs().max()) == 0.0
        assert torch.isnan(SpectralNetLoss()(b, Y.detach(), is_normalized=True))
        assert torch.isnan(O.spectral_loss(W, Y.detach().cpu().double(), True))


def test_lap_permutation_invariance_large(dev):
    """Size-independent property at a batch size the dense oracle cannot reach quickly:
    relabelling the nodes leaves the loss unchanged and permutes the gradient."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(5)
    m, k = 20000, 47
    s = random_support(m, 12.0 / m, rng, "raw")
    perm = rng.permutation(m)
    inv = np.empty_like(perm); inv[perm] = np.arange(m)
    s2 = (np.stack([inv[s[0][:, 0]], inv[s[0][:, 1]]], 1), s[1], s[2])
    Y = rng.standard_normal((m, k)).astype(np.float32)
    b1, b2 = ClusterBatch.from_support(s, dev), ClusterBatch.from_support(s2, dev)
    l1, g1 = b1.lap(torch.tensor(Y, device=dev), True)
    l2, g2 = b2.lap(torch.tensor(Y[perm], device=dev), True)
    assert abs(l1.item() - l2.item()) / abs(l1.item()) < 1e-5
    assert rel(g2.cpu().numpy(), g1.cpu().numpy()[perm]) < 1e-4
    assert l1.item() >= 0
    # gradient of a quadratic form: <grad, Y> = 2 loss
    assert abs((g1.double() * torch.tensor(Y, device=dev).double()).sum().item() - 2 * l1.item()) / l1.item() < 1e-4


@pytest.mark.parametrize("width", [16, 32, 64, 128])
def test_spmm_matches_scipy(dev, width):
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(width)
    m = 777
    s = random_support(m, 0.01, rng, "enh")
    b = ClusterBatch.from_support(s, dev)
    A, At = O.csr_from_support(s)
    X = rng.standard_normal((m, width)).astype(np.float32)
    Z = b.spmm(torch.tensor(X, device=dev)).cpu().numpy()
    assert rel(Z, A @ X.astype(np.float64)) < 1e-5
    Zt = b.spmm(torch.tensor(X, device=dev), add=torch.tensor(X, device=dev), transpose=True).cpu().numpy()
    assert rel(Zt, X + At @ X.astype(np.float64)) < 1e-5


def test_spmm_long_rows(dev):
    """Rows longer than one 64-entry chunk and a dense hub row."""
    from sparc_amd.graph import ClusterBatch
    rng = np.random.default_rng(9)
    m = 500
    D = (rng.random((m, m)) < 0.3).astype(np.float64)
    D[7, :] = 1.0
    D = np.triu(D, 1); D = D + D.T + np.eye(m)
    s = O.as_support_tuple(sp.csr_matrix(D))
    b = ClusterBatch.from_support(s, dev)
    X = rng.standard_normal((m, 64)).astype(np.float32)
    assert rel(b.spmm(torch.tensor(X, device=dev)).cpu().numpy(), D @ X.astype(np.float64)) < 1e-5


# ---------------------------------------------------------------- orthonormalisation
@pytest.mark.parametrize("m,k", [(200, 8), (1000, 64), (65, 47), (4097, 32), (70, 64)])
def test_gram_exact(dev, m, k):
    from sparc_amd.ops import gram
    rng = np.random.default_rng(m + k)
    A = rng.standard_normal((m, k)).astype(np.float32)
    B = rng.standard_normal((m, k)).astype(np.float32)
    B[0, :] = np.arange(k)  # asymmetric marker
    G = gram(torch.tensor(A, device=dev), torch.tensor(B, device=dev)).cpu().numpy()
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    assert rel(G, ref) < 1e-13
    G2 = gram(torch.tensor(A, device=dev), torch.tensor(B, device=dev)).cpu().numpy()
    assert np.array_equal(G, G2)  # fixed summation order


@pytest.mark.parametrize("name", ["gauss", "mlp", "wide"])
def test_orth_weights_match_reference_golden(dev, golden, name):
    from sparc_amd.SpectralTrainer import SpectralNetModel
    g = golden("g3_orth.npz")
    Yt = torch.tensor(g[name + "_Yt"], device=dev)
    k = Yt.shape[1]
    model = SpectralNetModel({"n_layers": 2, "h": 4, "output_dim": k}, 3, dev)
    Wo = model._make_orthonorm_weights(Yt)
    cond = float(g[name + "_cond"])
    # the reference's float32 QR is itself only cond*eps accurate; ours is float64 inside
    assert rel(Wo.cpu().numpy(), g[name + "_orth"]) < max(TOL, 5e-7 * cond ** 2)
    assert rel(Wo.cpu().numpy(), O.orthonorm_weights_qr(torch.tensor(g[name + "_Yt"]).double()).numpy()) < 1e-5
    Y = (Yt.double() @ Wo.double())
    err = (Y.t() @ Y / Yt.shape[0] - torch.eye(k, device=dev, dtype=torch.float64)).abs().max().item()
    assert abs(err - float(g[name + "_ortho_err"])) <= 1e-3 * max(1.0, float(g[name + "_ortho_err"]))


def test_orth_signs_follow_top_rows(dev):
    from sparc_amd.ops import OrthWeights
    rng = np.random.default_rng(2)
    Yt = torch.tensor(rng.standard_normal((300, 16)), dtype=torch.float32)
    perm = rng.permutation(300)
    ref = O.orthonorm_weights_qr(Yt[torch.as_tensor(perm)].double()).numpy()
    Wo = OrthWeights.apply(Yt.to(dev), perm[:16], 0).cpu().numpy()
    assert rel(Wo, ref) < 1e-5
    # row numbers on the host travel in the kernel's arguments, row numbers on the device are read there: same bits;
    # a list shorter than k and row numbers out of range stand for rows of zeros either way
    on_dev = torch.as_tensor(perm[:16].astype(np.int32), device=dev)
    assert np.array_equal(OrthWeights.apply(Yt.to(dev), on_dev, 0).cpu().numpy(), Wo)
    short = np.concatenate([perm[:11], [300, 300, 300, 300, 300]])
    a = OrthWeights.apply(Yt.to(dev), perm[:11], 0).cpu().numpy()
    b = OrthWeights.apply(Yt.to(dev), torch.as_tensor(short.astype(np.int32), device=dev), 0).cpu().numpy()
    c = OrthWeights.apply(Yt.to(dev), np.concatenate([perm[:11], [-1, 300, 99999, 301, 300]]), 0).cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(a, c)
    Wc = OrthWeights.apply(Yt.to(dev), None, 1).cpu().numpy()
    assert rel(Wc, O.orthonorm_weights_chol(Yt.double(), torch.ones(16, dtype=torch.float64)).numpy()) < 1e-6


@pytest.mark.parametrize("k", [8, 47, 64])
def test_orth_layer_backward_matches_autograd(dev, k):
    from sparc_amd.ops import OrthApply, OrthWeights
    rng = np.random.default_rng(k)
    m1, m2 = 150, 260
    Y1 = torch.tensor(rng.standard_normal((m1, k)), dtype=torch.float32)
    Y2 = torch.tensor(rng.standard_normal((m2, k)), dtype=torch.float32)
    T = torch.tensor(rng.standard_normal((m1, k)), dtype=torch.float32)
    a1, a2 = Y1.double().requires_grad_(True), Y2.double().requires_grad_(True)
    out = (a1 @ O.orthonorm_weights_qr(a2) * T.double()).sum()
    r1, r2 = torch.autograd.grad(out, [a1, a2])
    b1, b2 = Y1.to(dev).requires_grad_(True), Y2.to(dev).requires_grad_(True)
    Z = OrthApply.apply(b1, OrthWeights.apply(b2, None, 0))
    (Z * T.to(dev)).sum().backward()
    assert rel(Z.detach().cpu().numpy(), (a1 @ O.orthonorm_weights_qr(a2)).detach().numpy()) < TOL
    assert rel(b1.grad.cpu().numpy(), r1.numpy()) < TOL
    assert rel(b2.grad.cpu().numpy(), r2.numpy()) < TOL


def test_orth_not_positive_definite_is_reported(dev):
    from sparc_amd.ops import OrthWeights
    Yt = torch.randn(50, 4, device=dev)
    Yt[:, 2] = 0.0  # a dead output unit: zero pivot
    OrthWeights.apply(Yt, None, 0)
    assert int(OrthWeights.last_info) != 0


# ---------------------------------------------------------------- MLP backward helpers
@pytest.mark.parametrize("m,n,act", [(1000, 1024, 0), (777, 47, 1), (5, 3, 0), (4097, 512, 0), (33, 2052, 1), (1, 1, 1)])
def test_act_bwd_bias_matches_torch(dev, m, n, act):
    from sparc_amd.ops import act_bwd_bias
    gen = torch.Generator().manual_seed(m + n)
    g = torch.randn(m, n, generator=gen)
    pre = torch.randn(m, n, generator=gen)
    h = torch.relu(pre) if act == 0 else torch.tanh(pre)
    want = torch.where(h > 0, g, torch.zeros(())) if act == 0 else g * (1 - h * h)
    gout, db = act_bwd_bias(g.to(dev), h.to(dev), act)
    if act == 0:
        assert torch.equal(gout.cpu(), want)  # a select: exact
    else:
        assert rel(gout.cpu().numpy(), want.numpy()) < 1e-6  # 1 - h*h may be contracted to an fma on the device
    assert rel(db.cpu().numpy(), want.double().sum(0).numpy()) < 1e-5


@pytest.mark.parametrize("m,n_out,n_in", [(3001, 47, 512), (100, 16, 64), (3, 64, 128), (24491, 33, 512)])
def test_wgrad_small_matches_float64(dev, m, n_out, n_in):
    from sparc_amd.ops import wgrad_small
    gen = torch.Generator().manual_seed(m)
    g = torch.randn(m, n_out, generator=gen)
    h = torch.randn(m, n_in, generator=gen)
    dW = wgrad_small(g.to(dev), h.to(dev))
    assert rel(dW.cpu().numpy(), (g.double().t() @ h.double()).numpy()) < 1e-5


@pytest.mark.parametrize("m,din,n", [(3001, 10, 1024), (70, 63, 64), (5, 1, 128), (24491, 10, 256)])
def test_first_layer_backward_matches_float64(dev, m, din, n):
    from sparc_amd.ops import first_layer_bwd
    gen = torch.Generator().manual_seed(m + din)
    x = torch.randn(m, din, generator=gen)
    g = torch.randn(m, n, generator=gen)
    h = torch.relu(torch.randn(m, n, generator=gen))
    gp = torch.where(h > 0, g, torch.zeros(())).double()
    dW, db = first_layer_bwd(x.to(dev), g.to(dev), h.to(dev))
    assert dW.shape == (n, din) and db.shape == (n,)
    assert rel(dW.cpu().numpy(), (gp.t() @ x.double()).numpy()) < 1e-5
    assert rel(db.cpu().numpy(), gp.sum(0).numpy()) < 1e-5


@pytest.mark.parametrize("m,din,n", [(3001, 10, 1024), (70, 16, 64), (5, 1, 128), (24491, 10, 256), (4099, 15, 192)])
def test_first_layer_backward_with_recomputed_mask_is_bitwise_the_same(dev, m, din, n):
    """The backward variant that does not read the layer's activations (it recomputes the sign of every pre-activation
    with the forward kernel's own arithmetic) against the variant that reads them: the same bits, on inputs where many
    pre-activations sit next to zero (small integers times powers of two, so that exact zeros and ties occur too)."""
    from sparc_amd.ops import ACT_RELU, first_layer_bwd, linear_narrow_in_fwd
    gen = torch.Generator(device=dev).manual_seed(m + din)
    x = torch.randint(-3, 4, (m, din), device=dev, generator=gen).float() * 0.25
    W = torch.randint(-3, 4, (n, din), device=dev, generator=gen).float() * 0.5 + 1e-3 * torch.randn(n, din, device=dev, generator=gen)
    b = torch.randint(-2, 3, (n,), device=dev, generator=gen).float() * 0.125
    W[: n // 4] = W[: n // 4].round()       # a quarter of the units with exactly representable sums: pre-activations of exactly zero
    g = torch.randn(m, n, device=dev, generator=gen)
    h = linear_narrow_in_fwd(x, W, b, ACT_RELU)
    assert h is not None and float((h == 0).float().mean()) > 0.2
    dW_read, db_read = first_layer_bwd(x, g, h)
    dW_re, db_re = first_layer_bwd(x, g, None, W, b)
    assert torch.equal(dW_read, dW_re) and torch.equal(db_read, db_re)
    ref = (g.double() * (h > 0)).t() @ x.double()
    assert rel(dW_re.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("m,K,n_out,act", [(3001, 512, 47, 1), (17, 16, 1, 0), (24491, 512, 64, 2), (100, 768, 33, 1)])
def test_linear_narrow_fwd_matches_float64(dev, m, K, n_out, act):
    from sparc_amd.ops import linear_narrow_fwd
    gen = torch.Generator().manual_seed(m + K)
    x = torch.randn(m, K, generator=gen)
    W = torch.randn(n_out, K, generator=gen) / K ** 0.5
    b = torch.randn(n_out, generator=gen)
    pre = x.double() @ W.double().t() + b.double()
    want = (torch.relu(pre), torch.tanh(pre), pre)[act]
    got = linear_narrow_fwd(x.to(dev), W.to(dev), b.to(dev), act)
    assert got is not None and got.shape == (m, n_out)
    assert rel(got.cpu().numpy(), want.numpy()) < 1e-5
    # a weight matrix that does not fit the LDS of one CU is left to the library product
    assert linear_narrow_fwd(torch.randn(8, 1024, device=dev), torch.randn(33, 1024, device=dev), torch.randn(33, device=dev), 1) is None


@pytest.mark.parametrize("m,K,n_out,m2", [(48980, 512, 47, 24490), (3001, 512, 47, 1500), (3001, 512, 47, 3001), (333, 32, 12, 1),
                                          (15600, 512, 64, 7801), (2000, 64, 32, 17), (40, 16, 1, 33)])
def test_linear_narrow_fwd_with_gram_in_its_epilogue(dev, m, K, n_out, m2):
    """The output layer's forward that also leaves out[:m2]^T out[:m2]: the same `out` bit for bit as the plain call, and
    the Gram matrix of exactly those float32 values (the same exact products as sparc_gram, summed in another order in
    float64)."""
    from sparc_amd.ops import gram, linear_narrow_fwd
    gen = torch.Generator().manual_seed(m + K + m2)
    x = torch.randn(m, K, generator=gen).to(dev)
    W = (torch.randn(n_out, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(n_out, generator=gen).to(dev)
    plain = linear_narrow_fwd(x, W, b, 1)
    out, G = linear_narrow_fwd(x, W, b, 1, gram_rows=m2)
    assert G is not None and G.dtype == torch.float64 and G.shape == (n_out, n_out)
    assert torch.equal(out, plain)
    want = out[:m2].double().t() @ out[:m2].double()
    assert float((G - want).abs().max()) <= 1e-12 * float(want.abs().max())
    assert float((G - gram(out[:m2], out[:m2])).abs().max()) <= 1e-12 * float(want.abs().max())
    assert torch.equal(G, G.t())


@pytest.mark.parametrize("m,k", [(24490, 47), (1000, 64), (37, 12), (5, 3), (7801, 32)])
def test_gram_apply_equals_the_two_separate_calls(dev, m, k):
    from sparc_amd.ops import gram, gram_apply, orth_apply
    gen = torch.Generator().manual_seed(m + k)
    A = torch.randn(m, k, generator=gen).to(dev)
    B = torch.randn(m, k, generator=gen).to(dev)
    W = torch.randn(k, k, generator=gen).to(dev)
    for transpose in (True, False):
        buf = torch.full((m + 3, k + 5), 7.0, device=dev)      # rows of another stride, guard cells around them
        out = buf[1:m + 1, :k]
        G = gram_apply(A, B, W, transpose, out)
        assert torch.equal(G, gram(A, B))
        assert torch.equal(out, orth_apply(B, W, transpose=transpose))
        assert float(buf[0].min()) == 7.0 and float(buf[m + 1:].min()) == 7.0 and float(buf[:, k:].min()) == 7.0


def test_failure_codes_survive_a_lap_of_the_ring(dev):
    from sparc_amd.ops import OrthWeights
    OrthWeights.check_and_reset(dev)
    good = torch.randn(200, 8, device=dev)
    bad = torch.zeros(200, 8, device=dev)
    OrthWeights.weights(bad, None, 0)
    for _ in range(OrthWeights.RING + 5):
        OrthWeights.weights(good, None, 0)
    assert int(OrthWeights.worst(dev)) != 0
    assert OrthWeights.check_and_reset(dev) != 0 and OrthWeights.check_and_reset(dev) == 0
    OrthWeights.weights(good, None, 0)
    assert OrthWeights.check_and_reset(dev) == 0


def test_linear_narrow_fwd_with_gram_leaves_large_batches_to_two_calls(dev):
    from sparc_amd.ops import linear_narrow_fwd
    m = 256 * 16 * 16 + 1
    x = torch.randn(m, 32, device=dev)
    out, G = linear_narrow_fwd(x, torch.randn(8, 32, device=dev), torch.randn(8, device=dev), 1, gram_rows=100)
    assert out.shape == (m, 8) and G is None
    from sparc_amd._lib import lib
    G = torch.empty(8, 8, dtype=torch.float64, device=dev)
    ws = torch.empty(lib.sparc_linear_narrow_fwd_gram_workspace_bytes(8), dtype=torch.uint8, device=dev)
    rc = lib.sparc_linear_narrow_fwd_gram(m, 32, 8, 1, x.data_ptr(), 32, x.data_ptr(), x.data_ptr(), out.data_ptr(), 8, 100,
                                          G.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -4


@pytest.mark.parametrize("m,K,N", [(3001, 512, 1024), (256, 16, 128), (257, 48, 256), (5, 64, 128), (24491, 512, 1024), (1024, 1024, 1024)])
def test_dgrad_wide_relu_matches_float64(dev, m, K, N):
    """The hidden layer's input gradient with the previous ReLU's derivative and bias gradient in its epilogue: against
    float64, and bit for bit the mask (a zero wherever hprev is not positive)."""
    from sparc_amd.ops import dgrad_wide_relu
    gen = torch.Generator().manual_seed(m + K + N)
    g = torch.randn(m, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    hprev = torch.relu(torch.randn(m, N, generator=gen)).to(dev)
    got = dgrad_wide_relu(g, W, hprev)
    assert got is not None
    dx, db = got
    want = (g.double() @ W.double()) * (hprev > 0)
    assert rel(dx.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert bool((dx[hprev <= 0] == 0).all())
    assert rel(db.cpu().numpy(), want.sum(0).cpu().numpy()) < 1e-5
    assert rel(db.cpu().numpy(), dx.double().sum(0).cpu().numpy()) < 1e-6
    # shapes the kernel does not take are left to the library product
    assert dgrad_wide_relu(g[:, :K - 1].contiguous(), W[:K - 1].contiguous(), hprev) is None or (K - 1) % 16 == 0


@pytest.mark.parametrize("m,K,N,din", [(3001, 64, 128, 10), (255, 16, 256, 1), (12345, 512, 1024, 16), (770, 48, 128, 5),
                                         (256, 32, 128, 13), (48979, 1024, 1024, 10)])
def test_dgrad_wide_first_matches_float64(dev, m, K, N, din):
    """The first layer's weight and bias gradient out of the epilogue of the product that computes the gradient
    reaching its output (which is never stored), the ReLU derivative recomputed: against float64 with the mask of the
    forward kernel's own output, and against the two-step path's layout."""
    from sparc_amd.ops import ACT_RELU, dgrad_wide_first, linear_narrow_in_fwd
    gen = torch.Generator().manual_seed(m + K + N + din)
    g = torch.randn(m, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    x = torch.randn(m, din + 2, generator=gen).to(dev)[:, :din]     # (a row stride that is not the width)
    W0 = torch.randn(N, din, generator=gen).to(dev)
    b0 = torch.randn(N, generator=gen).to(dev)
    h0 = linear_narrow_in_fwd(x, W0, b0, ACT_RELU)
    got = dgrad_wide_first(g, W, x, W0, b0)
    assert got is not None
    dW, db = got
    assert dW.shape == (N, din) and db.shape == (N,)
    dx = (g.double() @ W.double()) * (h0 > 0)
    assert rel(dW.cpu().numpy(), (dx.t() @ x.double()).cpu().numpy()) < 1e-5
    assert rel(db.cpu().numpy(), dx.sum(0).cpu().numpy()) < 1e-5
    # into a caller's buffer (a reducer's bucket): weight gradient, then bias gradient
    out = torch.full(((din + 1) * N,), float("nan"), device=dev)
    dW2, db2 = dgrad_wide_first(g, W, x, W0, b0, out)
    assert dW2.data_ptr() == out.data_ptr() and db2.data_ptr() == out.data_ptr() + 4 * din * N
    assert torch.equal(dW2, dW) and torch.equal(db2, db)
    assert dgrad_wide_first(g, W, torch.randn(m, 17, device=dev), torch.randn(N, 17, device=dev), b0) is None


@pytest.mark.parametrize("m,n_out,N", [(3001, 47, 512), (70, 64, 64), (5, 1, 128), (24491, 33, 256)])
def test_dgrad_narrow_relu_matches_float64(dev, m, n_out, N):
    from sparc_amd.ops import dgrad_narrow_relu
    gen = torch.Generator().manual_seed(m + N)
    g = torch.randn(m, n_out, generator=gen)
    W = torch.randn(n_out, N, generator=gen)
    h = torch.relu(torch.randn(m, N, generator=gen))
    want = torch.where(h > 0, g.double() @ W.double(), torch.zeros((), dtype=torch.float64))
    dx, db = dgrad_narrow_relu(g.to(dev), W.to(dev), h.to(dev))
    assert rel(dx.cpu().numpy(), want.numpy()) < 1e-5
    assert torch.equal(dx.cpu() == 0, want == 0) or float(((dx.cpu() == 0) != (want == 0)).float().mean()) < 1e-6
    assert rel(db.cpu().numpy(), want.sum(0).numpy()) < 1e-5


@pytest.mark.parametrize("dims,x_grad", [((10, 128, 64, 64, 47), False), ((100, 64, 8), True), ((7, 5, 3), False),
                                          ((12, 9), True)])
def test_spectral_mlp_matches_torch_layers(dev, dims, x_grad):
    """The whole-MLP function (every fused path and every fallback) against torch.nn layers."""
    from sparc_amd.ops import SpectralMLP
    torch.manual_seed(len(dims))
    lins = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    m = 1537
    x = torch.randn(m, dims[0], device=dev, requires_grad=x_grad)
    t = torch.randn(m, dims[-1], device=dev)
    h = x
    for lin in lins[:-1]:
        h = torch.relu(lin(h))
    ref = torch.tanh(lins[-1](h))
    params = [p for lin in lins for p in (lin.weight, lin.bias)]
    wrt = params + ([x] if x_grad else [])
    rg = torch.autograd.grad((ref * t).sum(), wrt)
    out = SpectralMLP.apply(x, None, *params)
    og = torch.autograd.grad((out * t).sum(), wrt)
    assert rel(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    for a, b in zip(og, rg):
        assert a.shape == b.shape and rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


def test_permutation_heads_from_the_host_travel_asynchronously_and_intact(dev):
    """Host index lists reach the orthonormalisation inside its kernel arguments (no copy and no host stop inside the
    forward pass): forty calls in a row without a synchronisation in between give what the same heads give as device
    tensors; a list longer than k counts with its first k entries."""
    from sparc_amd import ops
    torch.manual_seed(2)
    rng = np.random.default_rng(2)
    m, k = 3000, 24
    Y = torch.randn(m, k, device=dev)
    heads = [rng.permutation(m)[:k] for _ in range(40)]
    got = [ops.OrthWeights.apply(Y, h, 0) for h in heads]            # enqueued back to back
    torch.cuda.synchronize()
    for h, w in zip(heads, got):
        ref = ops.OrthWeights.apply(Y, torch.as_tensor(h, device=dev, dtype=torch.int32), 0)
        assert torch.equal(w, ref)
    long_list = np.concatenate([heads[0], rng.permutation(m)[:70]])
    assert torch.equal(ops.OrthWeights.apply(Y, long_list, 0), got[0])
    assert torch.equal(ops.OrthWeights.apply(Y, torch.as_tensor(heads[0]), 0), got[0])   # a host tensor
    assert ops._index_to_device(torch.arange(5, device=dev), dev).dtype == torch.int32
    assert np.array_equal(ops._index_to_device(torch.arange(7), dev).cpu().numpy(), np.arange(7))


@pytest.mark.parametrize("m,K,N,act", [(1537, 10, 1024, 0), (1, 3, 4, 2), (5000, 16, 2052, 1), (333, 1, 64, 0),
                                        (2049, 12, 512, 2), (40000, 10, 1024, 0)])
def test_first_layer_forward_on_few_input_columns(dev, m, K, N, act):
    """act(x W^T + b) for a layer with at most 16 inputs through its own kernel (a store stream with the weights
    in registers) against torch in float64; rows of x may be longer than K (a view of a wider matrix)."""
    from sparc_amd._lib import lib
    from sparc_amd import ops
    torch.manual_seed(m + K)
    xw = torch.randn(m, K + 3, device=dev)
    x = xw[:, :K]
    lin = torch.nn.Linear(K, N).to(dev)
    pre = x.double() @ lin.weight.double().t() + lin.bias.double()
    ref = (torch.relu(pre) if act == 0 else torch.tanh(pre) if act == 1 else pre).detach().cpu().numpy()
    out = ops.linear_narrow_in_fwd(x, lin.weight.detach(), lin.bias.detach(), act)
    assert out is not None and out.shape == (m, N)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
    # the MLP's layers take it: same values as the library product to rounding, gradients as before
    y = ops.LinearAct.apply(x, lin.weight, lin.bias, act)
    assert torch.equal(y.detach(), out)
    assert ops.linear_narrow_in_fwd(torch.randn(8, 17, device=dev), torch.randn(8, 17, device=dev), torch.randn(8, device=dev), 0) is None
    assert ops.linear_narrow_in_fwd(torch.randn(8, 4, device=dev), torch.randn(6, 4, device=dev), torch.randn(6, device=dev), 0) is None
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    bad = lib.sparc_linear_narrow_in_fwd(m, 17, N, 0, P(x), x.stride(0), P(lin.weight), P(lin.bias), P(out), N, None)
    assert bad == -1


def test_weight_gradient_summed_in_parts(dev, monkeypatch):
    """Long batches sum the weight gradient of a wide layer whose result is small in four parts as one batched
    product (ops.wgrad_wide): against float64, with rows left over after the four parts, and through the whole
    MLP's backward against the same backward without the split."""
    from sparc_amd import ops
    torch.manual_seed(8)
    m = 16384 + 4 * 611 + 3
    g, x = torch.randn(m, 128, device=dev), torch.randn(m, 192, device=dev)
    ref = (g.double().t() @ x.double()).cpu().numpy()
    calls = []
    real_bmm = torch.bmm
    monkeypatch.setattr(torch, "bmm", lambda *a, **k: (calls.append(a[0].shape), real_bmm(*a, **k))[1])
    out = ops.wgrad_wide(g, x)
    assert calls == [(4, 128, (m - 3) // 4)]              # the split path ran
    assert rel(out.cpu().numpy(), ref) < 5e-6              # (the single product: about 5e-6 as well)
    assert rel(g.t().mm(x).cpu().numpy(), ref) < 2e-5
    assert ops.wgrad_wide(g[:16000], x[:16000]).shape == (128, 192) and len(calls) == 1   # short batch: one product
    big = torch.randn(m, 1024, device=dev)
    ops.wgrad_wide(big, big)                               # 256 tiles: one product
    assert len(calls) == 1
    dims = (12, 192, 128, 64, 9)
    lins = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    params = [p for lin in lins for p in (lin.weight, lin.bias)]
    xin, t = torch.randn(m, dims[0], device=dev), torch.randn(m, dims[-1], device=dev)
    grads = {}
    for split_rows in (16384, 1 << 40):
        monkeypatch.setattr(ops, "WGRAD_SPLIT_ROWS", split_rows)
        n0 = len(calls)
        grads[split_rows] = torch.autograd.grad((ops.SpectralMLP.apply(xin, None, *params) * t).sum(), params)
        assert (len(calls) > n0) == (split_rows == 16384)
    for a, b in zip(grads[16384], grads[1 << 40]):
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


def test_linear_act_first_layer_path(dev):
    """A Linear+ReLU layer on an input that needs no gradient takes the one-pass backward."""
    from sparc_amd.ops import LinearAct
    torch.manual_seed(4)
    lin = torch.nn.Linear(10, 128).to(dev)
    x = torch.randn(777, 10, device=dev)
    t = torch.randn(777, 128, device=dev)
    rg = torch.autograd.grad((torch.relu(lin(x)) * t).sum(), [lin.weight, lin.bias])
    og = torch.autograd.grad((LinearAct.apply(x, lin.weight, lin.bias, 0) * t).sum(), [lin.weight, lin.bias])
    for a, b in zip(og, rg):
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_linear_act_matches_torch_layers(dev):
    from sparc_amd.ops import LinearAct
    torch.manual_seed(3)
    for (m, din, dout, act) in [(513, 10, 128, 0), (513, 128, 47, 1), (64, 7, 5, 0)]:
        lin = torch.nn.Linear(din, dout).to(dev)
        x = torch.randn(m, din, device=dev, requires_grad=True)
        t = torch.randn(m, dout, device=dev)
        ref = (torch.relu if act == 0 else torch.tanh)(lin(x))
        rg = torch.autograd.grad((ref * t).sum(), [x, lin.weight, lin.bias])
        out = LinearAct.apply(x, lin.weight, lin.bias, act)
        og = torch.autograd.grad((out * t).sum(), [x, lin.weight, lin.bias])
        assert rel(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
        for a, b in zip(og, rg):
            assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


# ---------------------------------------------------------------- one optimisation step
def _load_model(g, prefix, din, k, dev):
    from sparc_amd.SpectralTrainer import SpectralNetModel
    arch = {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": k}
    net = SpectralNetModel(arch, din, dev).to(dev)
    sd = {f"layers.{i}.0.{w}": torch.tensor(g[f"{prefix}layers.{i}.0.{w}"]) for i in range(4) for w in ("weight", "bias")}
    net.load_state_dict(sd)
    return net


def test_train_step_matches_reference_golden(dev, golden):
    from sparc_amd.SpectralTrainer import SpectralNetLoss
    from sparc_amd.graph import ClusterBatch
    g = golden("g5_step.npz")
    net = _load_model(g, "init.", 12, 8, dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x1, x2 = torch.tensor(g["x1"], device=dev), torch.tensor(g["x2"], device=dev)
    batch = ClusterBatch.from_support(sup(g, "sup1_"), dev)
    net.eval()
    net(x2, should_update_orth_weights=True, top_idx=g["perm2"][:8])
    net.train()
    opt.zero_grad()
    Y = net(x1, should_update_orth_weights=False)
    loss = SpectralNetLoss()(batch, Y, is_normalized=True)
    loss.backward()
    assert net.orthonorm_weights.requires_grad
    assert rel(net.orthonorm_weights.detach().cpu().numpy(), g["orth"]) < TOL
    assert rel(Y.detach().cpu().numpy()[g["perm1"]], g["Y"]) < TOL      # the reference permuted the batch
    assert rel(loss.item(), g["loss"]) < TOL
    for name, p in net.named_parameters():
        # observed: <= 1.4e-5, except the output layer's bias (1.8e-4): a sum over the rows of terms that nearly
        # cancel, which the reference's own float32 summation resolves no better
        assert rel(p.grad.cpu().numpy(), g["grad." + name]) < (5e-4 if name == "layers.3.0.bias" else 5e-5), name
    opt.step()
    for name, p in net.state_dict().items():
        assert rel(p.cpu().numpy(), g["after." + name]) < TOL, name
    # validation with the carried weights (no grad, no update)
    net.eval()
    with torch.no_grad():
        vb = ClusterBatch.from_support(sup(g, "supv_"), dev)
        Yv = net(x2, should_update_orth_weights=False)
        vl = SpectralNetLoss()(vb, Yv, is_normalized=True)
    assert rel(Yv.cpu().numpy(), g["val_Y"]) < TOL
    assert rel(vl.item(), g["val_loss"]) < TOL


def test_forward_pair_equals_two_forwards(dev, golden):
    """One MLP pass over both batches of a step gives the reference's two forwards and its gradients."""
    from sparc_amd.SpectralTrainer import SpectralNetLoss
    from sparc_amd.graph import ClusterBatch
    g = golden("g5_step.npz")
    x1, x2 = torch.tensor(g["x1"], device=dev), torch.tensor(g["x2"], device=dev)
    batch = ClusterBatch.from_support(sup(g, "sup1_"), dev)
    net = _load_model(g, "init.", 12, 8, dev)
    Y = net.forward_pair(x2, x1, top_idx=g["perm2"][:8])
    loss = SpectralNetLoss()(batch, Y, is_normalized=True)
    loss.backward()
    assert rel(net.orthonorm_weights.detach().cpu().numpy(), g["orth"]) < TOL
    assert rel(Y.detach().cpu().numpy()[g["perm1"]], g["Y"]) < TOL
    assert rel(loss.item(), g["loss"]) < TOL
    for name, p in net.named_parameters():
        # observed: <= 1.4e-5, except the output layer's bias (1.8e-4): a sum over the rows of terms that nearly
        # cancel, which the reference's own float32 summation resolves no better
        assert rel(p.grad.cpu().numpy(), g["grad." + name]) < (5e-4 if name == "layers.3.0.bias" else 5e-5), name


def test_step_gradient_survives_an_ill_conditioned_orthonormalisation_batch(dev):
    """An untrained MLP now and then maps the orthonormalisation batch to nearly collinear outputs (condition number
    1.7e4 in this problem, number 0 of tests/fuzz_step.py).  The gradient with respect to that batch is then a product
    Yt H whose terms cancel over several orders of magnitude; with H and the sums in float32 the parameter gradient was
    0.27 off, where the reference's float32 QR backward keeps 4e-4.  H and the product are float64 now."""
    from tests.fuzz_step import allowance, step_errors
    line, errs, e32, cond = step_errors(0)
    assert cond > 1e4, line
    assert errs["grads"] < 2e-3 and errs["last bias"] < 2e-2, line
    allow = allowance(e32, cond)
    assert all(errs[a] <= allow[a] for a in errs), line


@pytest.mark.parametrize("hip_graph", [False, True])
def test_trainer_end_to_end_matches_reference(dev, golden, tmp_path, monkeypatch, capsys, hip_graph):
    """Two epochs of SpectralTrainer.train with the reference's seeds and data; eagerly, and with every step
    replayed as a HIP graph (spectral.hip_graph)."""
    from sparc_amd.SpectralTrainer import SpectralTrainer
    g = golden("g8_trainer.npz")
    monkeypatch.chdir(tmp_path)
    feats = [torch.tensor(g[f"feat{b}"]) for b in range(3)]
    sups = [(g[f"coords{b}"], g[f"values{b}"], tuple(int(x) for x in g[f"shape{b}"])) for b in range(3)]
    vf = [torch.tensor(g[f"vfeat{b}"]) for b in range(2)]
    vs = [(g[f"vcoords{b}"], g[f"vvalues{b}"], tuple(int(x) for x in g[f"vshape{b}"])) for b in range(2)]
    cfg = {"dataset": "golden", "spectral": {
        "architecture": {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": 6},
        "epochs": 2, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024, "n_clusters": 6, "patience": 10,
        "hip_graph": hip_graph}}
    torch.manual_seed(0)
    np.random.seed(0)
    tr = SpectralTrainer(cfg, dev)
    net = tr.train(([np.arange(f.shape[0]) for f in feats], feats, sups, [None] * 3, [None] * 3),
                   (list(range(2)), vf, vs, [None] * 2, [None] * 2))
    out = capsys.readouterr().out
    ref_lines = [l for l in str(g["stdout"]).splitlines() if l.startswith("Epoch")]
    our_lines = [l for l in out.splitlines() if l.startswith("Epoch")]
    assert len(ref_lines) == len(our_lines) == 2

    def nums(line):
        import re
        return [float(x) for x in re.findall(r"[-+]?\d+\.\d+", line)]
    for a, b in zip(our_lines, ref_lines):
        na, nb = nums(a), nums(b)
        # observed on an MI355X: 3e-7 (tests/tol_probe.py); the printed losses carry 7 decimals
        assert abs(na[0] - nb[0]) / nb[0] < 1e-5 and abs(na[1] - nb[1]) / nb[1] < 1e-5, (a, b)
    # initialisation used the same torch RNG stream as the reference
    for name, p in net.state_dict().items():
        assert rel(p.cpu().numpy(), g["final." + name]) < 1e-5, name      # observed: 5e-7 after the six Adam steps
    assert (tmp_path / "weights" / "golden" / "spectralnet_weights.pth").exists()


def test_predict_matches_reference_facade(dev, golden):
    """SpectralNet.predict on all nodes with the reference's trained weights, pushed through in chunks."""
    from sparc_amd.SpectralNet import SpectralNet
    from sparc_amd.SpectralTrainer import SpectralNetModel
    g = golden("g8_trainer.npz")
    arch = {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": 6}
    net = SpectralNetModel(arch, 10, dev).to(dev)
    net.load_state_dict({k[len("final."):]: torch.tensor(g[k]) for k in g.files if k.startswith("final.")})
    net.orthonorm_weights = torch.tensor(g["final_orth"], device=dev)
    facade = SpectralNet(n_clusters=6, config={"should_use_ae": False, "should_use_siamese": False})
    facade.spec_net = net
    X = torch.tensor(g["predict_X"])                      # stays on the host: predict moves chunks
    Y = facade.predict(X, chunk_rows=100)
    assert Y.dtype == np.float32 and Y.shape == g["predict_Y"].shape
    assert rel(Y, g["predict_Y"]) < TOL
    assert facade.embeddings_ is Y
    assert rel(Y, facade.predict(X)) < 1e-5               # chunking only changes the GEMM's rounding


# ---------------------------------------------------------------- support / degree construction
def _csr(g, prefix):
    n = int(g["n"])
    return sp.csr_matrix((g[prefix + "data"], g[prefix + "indices"], g[prefix + "indptr"]), shape=(n, n))


@pytest.mark.parametrize("li,lam", [(0, -1), (1, 0.01)])
def test_device_multicluster_supports_bit_exact(dev, golden, li, lam):
    from sparc_amd import utils as U
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    sizes = g[f"mc{li}_parts_sizes"]
    parts = np.split(g[f"mc{li}_parts_after_shuffle"], np.cumsum(sizes)[:-1])
    graph = U.DeviceGraph.from_scipy(train_adj, dev)
    for b in range(int(g[f"mc{li}_nbatch"])):
        pt = np.concatenate(parts[2 * b:2 * b + 2]).astype(np.int32)
        coords, values, shape, rowsum = graph.subblock(pt, mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
        r0, c0, v0, sh0 = O.canonical_support((g[f"mc{li}_coords{b}"], g[f"mc{li}_values{b}"], g[f"mc{li}_shape{b}"]))
        assert tuple(shape) == sh0
        assert np.array_equal(coords.cpu().numpy()[:, 0], r0) and np.array_equal(coords.cpu().numpy()[:, 1], c0)
        assert np.array_equal(values.cpu().numpy(), v0)     # bit-exact float64


@pytest.mark.parametrize("li,lam", [(0, -1), (1, 0.01)])
def test_device_validation_supports_bit_exact(dev, golden, li, lam):
    from sparc_amd import utils as U
    g = golden("g1_support.npz")
    adj = _csr(g, "adj_")
    graph = U.DeviceGraph.from_scipy(adj, dev)
    for b in range(3):
        pt = g[f"val{li}_part{b}"].astype(np.int32)
        coords, values, shape, rowsum = graph.subblock(pt, mode=(1 if lam == -1 else 2), binarize=True, diag_lambda=lam)
        r0, c0, v0, sh0 = O.canonical_support((g[f"val{li}_coords{b}"], g[f"val{li}_values{b}"], g[f"val{li}_shape{b}"]))
        assert tuple(shape) == sh0
        assert np.array_equal(coords.cpu().numpy()[:, 0], r0) and np.array_equal(coords.cpu().numpy()[:, 1], c0)
        assert np.array_equal(values.cpu().numpy(), v0)


def test_device_supports_round_like_scipy_on_weighted_graphs(dev):
    """Integer-weighted graphs (the Cora loader's multigraph gives weight 2): every reference operation rounds once.
    The first case is one where a fused multiply-add in `d + lambda * d` is one unit in the last place off
    (row sum 37, diagonal 5, lambda 0.01); then random graphs, subsets, flavours (tests/fuzz_support.py in small)."""
    from sparc_amd import utils as U
    A = sp.csr_matrix(np.array([[4.0, 32.0], [32.0, 0.0]]))
    coords, values, shape, rowsum = U.DeviceGraph.from_scipy(A, dev).subblock(np.array([0, 1], np.int32), mode=2, diag_lambda=0.01)
    d = (1.0 / (37.0 + 1e-20)) * 5.0
    assert rowsum.cpu().numpy()[0] == 37.0 and values.cpu().numpy()[0] == d + 0.01 * d
    for seed in range(25):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 400))
        A = sp.random(n, n, density=min(1.0, rng.uniform(0.2, 30) / n), random_state=np.random.RandomState(seed), format="csr")
        A.data[:] = rng.integers(1, 4, A.nnz).astype(np.float64)
        A = sp.csr_matrix(A + A.T + sp.eye(n))
        graph = U.DeviceGraph.from_scipy(A, dev)
        for mode, binarize, lam in ((0, False, -1), (1, True, -1), (2, False, 0.01), (2, True, 0.5)):
            pt = rng.permutation(n)[:int(rng.integers(1, n + 1))].astype(np.int32)
            sub = sp.csr_matrix(A[pt, :][:, pt])
            if binarize:
                sub = sp.csr_matrix((sub != 0) * 1.0)
            ref = sub if mode == 0 else O.row_normalize(sub) if mode == 1 else O.row_normalize_diag_enhance(sub, lam)
            coords, values, shape, _ = graph.subblock(pt, mode=mode, binarize=binarize, diag_lambda=lam)
            r0, c0, v0, sh0 = O.canonical_support(O.as_support_tuple(sp.csr_matrix(ref)))
            c = coords.cpu().numpy().reshape(-1, 2)
            assert tuple(shape) == sh0 and np.array_equal(c[:, 0], r0) and np.array_equal(c[:, 1], c0)
            assert np.array_equal(values.cpu().numpy(), v0), (seed, mode, binarize, lam)


def test_device_degree_arrays_bit_exact(dev, golden):
    from sparc_amd import utils as U
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    graph = U.DeviceGraph.from_scipy(train_adj, dev)
    n = int(g["n"])
    _, _, _, rowsum = graph.subblock(np.arange(n, dtype=np.int32), mode=0, binarize=False)
    assert np.array_equal(rowsum.cpu().numpy(), g["rowsum_raw"])
    _, _, _, rowsum = graph.subblock(np.arange(n, dtype=np.int32), mode=2, binarize=False, diag_lambda=0.01)
    assert np.array_equal(rowsum.cpu().numpy(), g["rowsum_enh"])


def test_preprocess_partitions_on_the_device(dev):
    """preprocess with num_clusters > 1 and no parts given: the device partitioner supplies them."""
    from sparc_amd.partition import edge_cut
    from sparc_amd.utils import DeviceGraph, preprocess
    rng = np.random.default_rng(11)
    n, k = 2000, 8
    lab = np.repeat(np.arange(k), n // k)
    u = rng.integers(0, n, 20000)
    v = np.where(rng.random(20000) < 0.85, lab[u] * (n // k) + rng.integers(0, n // k, 20000), rng.integers(0, n, 20000))
    A = sp.coo_matrix((np.ones(len(u)), (u, v)), shape=(n, n))
    A = sp.csr_matrix(((A + A.T + sp.eye(n)) > 0) * 1.0)
    feats = rng.standard_normal((n, 5)).astype(np.float32)
    y = np.zeros((n, 3), dtype=np.float32)
    mask = np.ones(n, dtype=bool)
    visible = rng.permutation(n)[:1800]           # a subset of the nodes, as the reference's train split
    parts, fb, sb, yb, mb = preprocess(DeviceGraph.from_scipy(A, dev), feats, y, mask, visible, k)
    assert len(parts) == k and sorted(np.concatenate(parts).tolist()) == sorted(visible.tolist())
    assert max(len(p) for p in parts) <= int(np.ceil(1800 / k * 1.03)) + 1
    where = np.full(n, -1)
    for g_, p in enumerate(parts):
        where[p] = g_
    sub = sp.coo_matrix(A[visible][:, visible])
    cut = int((where[visible[sub.row]] != where[visible[sub.col]]).sum()) // 2
    planted_cut = int((lab[visible[sub.row]] != lab[visible[sub.col]]).sum()) // 2
    assert cut <= 1.3 * planted_cut
    for p, (coords, values, shape) in zip(parts, sb):   # supports are the oracle's for these parts
        ref = O.as_support_tuple(O.row_normalize(sp.csr_matrix((A[p][:, p] > 0) * 1.0)))
        ours = O.canonical_support((coords.cpu().numpy(), values.cpu().numpy(), shape))
        assert np.array_equal(ours[0], O.canonical_support(ref)[0]) and np.array_equal(ours[1], O.canonical_support(ref)[1])


# ---------------------------------------------------------------- autoencoder stage
AE_CFG = {"dataset": "golden_ae", "should_use_ae": True, "should_use_siamese": False,
          "ae": {"architecture": {"hidden_dim1": 24, "hidden_dim2": 24, "hidden_dim3": 12, "output_dim": 5},
                 "epochs": 3, "n_samples": 1000, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "patience": 10,
                 "batch_size": 256}}


def test_autoencoder_stage_matches_reference(dev, golden, tmp_path, monkeypatch, capsys):
    """Three epochs of AETrainer.train with the reference's seed and data, then embed."""
    from sparc_amd.AETrainer import AETrainer
    g = golden("g9_ae.npz")
    monkeypatch.chdir(tmp_path)
    fb = [torch.tensor(g[f"feat{b}"]) for b in range(3)]
    vb = [torch.tensor(g[f"vfeat{b}"]) for b in range(2)]
    torch.manual_seed(1)
    tr = AETrainer(AE_CFG, dev)
    ae = tr.train(fb, vb)
    enc, enc_val = tr.embed(fb, vb)
    out = capsys.readouterr().out
    ref_lines = [l for l in str(g["stdout"]).splitlines() if l.startswith("Epoch")]
    our_lines = [l for l in out.splitlines() if l.startswith("Epoch")]
    assert our_lines == ref_lines                     # losses printed with 4 decimals
    for name, p in ae.state_dict().items():
        assert rel(p.cpu().numpy(), g["final." + name]) < 1e-5, name      # observed: 1.3e-7
    for b in range(3):
        assert rel(enc[b].cpu().numpy(), g[f"enc{b}"]) < 1e-5
    for b in range(2):
        assert rel(enc_val[b].cpu().numpy(), g[f"venc{b}"]) < 1e-5
    assert (tmp_path / "weights" / "golden_ae" / "ae_weights_5.pth").exists()
    # a second trainer finds the checkpoint and loads it instead of training (src/AETrainer.py:88-90)
    ae2 = AETrainer(AE_CFG, dev).train(fb, vb)
    for (n1, p1), (n2, p2) in zip(ae.state_dict().items(), ae2.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)
    # single-row helper and the module's own forward
    x = fb[0].to(dev)
    assert rel(tr.embed_single(fb[0]).cpu().numpy(), g["enc0"]) < 1e-5
    rec = ae(x)
    assert rec.shape == x.shape


# ---------------------------------------------------------------- token lists of the downstream models
def test_token_lists_match_reference_golden(dev, golden):
    from sparc_amd.tokens import spectral_token_lists
    g = golden("g10_tokens.npz")
    t = spectral_token_lists(g["encoding"], g["features"], int(g["K"]), device=dev, chunk_rows=64)
    assert t.shape == g["tokens"].shape and t.dtype == torch.float32 and t.device.type == "cpu"
    assert np.abs(t.numpy() - g["tokens"]).max() < 1e-5


@pytest.mark.parametrize("kind", ["normal", "ties", "tiny", "negatives"])
def test_row_rank_values_bit_exact(dev, kind):
    """The selection kernel against numpy's sort, bit for bit: continuous values (buckets refined once or
    twice), long runs of equal values straddling the ranks (refined down to the last key bit), rows shorter than the
    ranks, negative entries and zeros of both signs (the integer image of the doubles must keep their order)."""
    from sparc_amd.tokens import row_rank_values
    rng = np.random.default_rng(5)
    rows, n = 37, 23011
    if kind == "normal":
        D = rng.standard_normal((rows, n)) ** 2
        D[np.arange(rows), np.arange(rows)] = -1.0
    elif kind == "ties":
        D = rng.integers(0, 7, (rows, n)).astype(np.float64) * 0.125      # seven distinct values: every rank in a long run
        D[:, :5] = rng.standard_normal((rows, 5))
    elif kind == "tiny":
        n = 9
        D = rng.standard_normal((rows, n))
    else:
        D = rng.standard_normal((rows, n)) * 10.0 ** rng.integers(-300, 300, (rows, n))
        D[:, 7] = 0.0
        D[:, 8] = -0.0
        D[:, 9] = np.inf
    ranks = [1, 2, 4, 8, 16, 300, 2048, 4097, 16384, n]
    ranks = sorted(set(min(r, n) for r in ranks)) if kind != "tiny" else [1, 2, 4, 8, 16, 300]
    got = row_rank_values(torch.tensor(D, device=dev), ranks).cpu().numpy()
    srt = np.sort(D, axis=1)
    want = np.stack([srt[:, min(r, n) - 1] for r in ranks], axis=1)
    assert np.array_equal(got, want)      # -0.0 == 0.0: the two zeros compare equal, as they do in the membership test
    # a strided view (rows of a larger matrix)
    big = torch.tensor(np.concatenate([D, D], axis=1), device=dev)
    got2 = row_rank_values(big[:, :n], ranks).cpu().numpy()
    assert np.array_equal(got2, want)


def test_token_lists_match_oracle_larger(dev):
    from sparc_amd.tokens import spectral_token_lists
    rng = np.random.default_rng(8)
    n, k, nf, K = 3000, 16, 24, 13          # prefixes run past n: the last tokens average every node
    E = rng.standard_normal((n, k))
    Fm = rng.standard_normal((n, nf)).astype(np.float32)
    want = O.spectral_token_lists(E, Fm, K)
    got = spectral_token_lists(E, Fm, K, device=dev, chunk_rows=1000).numpy()
    assert np.abs(got - want).max() < 2e-5
    assert np.allclose(got[:, -1, :], Fm.mean(0, keepdims=True), atol=1e-5)


def test_token_lists_beyond_the_neighbour_limit(dev):
    """More nodes than the 16 384 neighbours the reference asks scikit-learn for
    (implementations/SPARCphormer/utils.py:268-298): the last prefixes stop at 16 384 nodes, not at n.  Checked on a
    sample of query rows against a direct sort of their distances."""
    from sparc_amd.tokens import spectral_token_lists
    rng = np.random.default_rng(3)
    n, k, nf, K = 17000, 12, 6, 15
    E = rng.standard_normal((n, k))
    Fm = rng.standard_normal((n, nf)).astype(np.float32)
    got = spectral_token_lists(E, Fm, K, device=dev).numpy()
    rows = rng.choice(n, 200, replace=False)
    d2 = ((E[rows, None, :] - E[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")
    for j in range(K):
        e = min(2 ** (j + 1), 2 ** 14, n)
        want = Fm[order[:, :e]].astype(np.float64).mean(1)
        assert np.abs(got[rows, j + 1, :] - want).max() < 1e-5, j
    assert np.array_equal(got[:, 0, :], Fm)


def test_facade_fit_predict_with_autoencoder(dev, golden, tmp_path, monkeypatch, capsys):
    """SpectralNet.fit with should_use_ae: autoencoder stage, SpectralNet stage on its codes, predict / embed
    on all nodes (src/SpectralNet.py:25-98), and the results writer."""
    from sparc_amd.SpectralNet import SpectralNet, save_results
    g = golden("g8_trainer.npz")
    monkeypatch.chdir(tmp_path)
    feats = [torch.tensor(g[f"feat{b}"]) for b in range(3)]
    sups = [(g[f"coords{b}"], g[f"values{b}"], tuple(int(x) for x in g[f"shape{b}"])) for b in range(3)]
    vf = [torch.tensor(g[f"vfeat{b}"]) for b in range(2)]
    vs = [(g[f"vcoords{b}"], g[f"vvalues{b}"], tuple(int(x) for x in g[f"vshape{b}"])) for b in range(2)]
    cfg = {"dataset": "facade", "n_clusters": 6, "should_use_ae": True, "should_use_siamese": False,
           "ae": dict(AE_CFG["ae"], epochs=2),
           "spectral": {"architecture": {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": 6},
                        "epochs": 2, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024, "n_clusters": 6,
                        "patience": 10, "bsize": 1}}
    torch.manual_seed(0)
    np.random.seed(0)
    net = SpectralNet(n_clusters=6, config=cfg)
    net.fit(([np.arange(f.shape[0]) for f in feats], feats, sups, [None] * 3, [None] * 3),
            (list(range(2)), vf, vs, [None] * 2, [None] * 2))
    out = capsys.readouterr().out
    assert "Training Autoencoder:" in out and "Training SpectralNet:" in out
    X = torch.cat(feats + vf, dim=0)
    Y = net.predict(X, chunk_rows=128)
    E = net.embed(X)
    assert Y.shape == (X.shape[0], 6) and E.shape == (X.shape[0], 5) and np.isfinite(Y).all()
    # the embedding of the orthonormalisation batch is orthonormal up to the weights' batch: finite and non-trivial
    assert np.abs(Y).max() > 0
    # predict = MLP(encoder(X)) @ orthonormalisation weights
    with torch.no_grad():
        code = net.ae_net.encode(X.to(dev))
        want = net.spec_net(code, should_update_orth_weights=False).cpu().numpy()
    assert rel(Y, want) < 1e-5 and rel(E, code.cpu().numpy()) < 1e-6
    save_results(str(tmp_path / "sparc_results" / "facade"), Y, X, np.zeros(len(X), int), np.ones(len(X), bool),
                 np.ones(len(X), bool), np.ones(len(X), bool), E)
    assert np.allclose(np.load(tmp_path / "sparc_results" / "facade" / "embeddings.npy"), Y / np.sqrt(len(X)))


# ---------------------------------------------------------------- real graph: Cora through the reference's loader
def test_cora_batch_matches_reference(dev, golden):
    """The shipped Cora graph (m = 2437, 11 041 stored entries, off-diagonal weight 2, k = 64): support built
    on the device bit-exact, affinity sums, orthonormalisation weights, embedding, both loss branches and the
    gradient against the imported reference (tools/gen_golden_cora.py)."""
    from sparc_amd import utils as U
    from sparc_amd.SpectralTrainer import SpectralNetLoss, SpectralNetModel
    from sparc_amd.graph import ClusterBatch
    g = golden("g7_cora.npz")
    n = int(g["num_data"])
    train_adj = sp.csr_matrix((g["train_data_values"], g["train_indices"], g["train_indptr"]), shape=(n, n))
    graph = U.DeviceGraph.from_scipy(train_adj, dev)
    coords, values, shape, _ = graph.subblock(g["train_part"].astype(np.int32), mode=0, binarize=False)
    r0, c0, v0, sh0 = O.canonical_support(sup(g, "sup_"))
    assert tuple(shape) == sh0 and np.array_equal(coords.cpu().numpy()[:, 0], r0) and np.array_equal(coords.cpu().numpy()[:, 1], c0)
    assert np.array_equal(values.cpu().numpy(), v0)
    full = sp.csr_matrix((g["full_data"], g["full_indices"], g["full_indptr"]), shape=(n, n))
    vcoords, vvalues, vshape, _ = U.DeviceGraph.from_scipy(full, dev).subblock(np.arange(n, dtype=np.int32), mode=1, binarize=True)
    r1, c1, v1, _ = O.canonical_support(sup(g, "val_"))
    assert np.array_equal(vcoords.cpu().numpy()[:, 0], r1) and np.array_equal(vcoords.cpu().numpy()[:, 1], c1)
    assert np.array_equal(vvalues.cpu().numpy(), v1)

    b = ClusterBatch.from_device_coo(shape[0], coords, values)
    assert rel(b.d.cpu().numpy(), g["W_rowsum"]) < TOL and rel(b.cw.cpu().numpy(), g["W_colsum"]) < TOL
    m, k = shape[0], 64
    rng = np.random.default_rng(7)
    rng.integers(0, m, size=(256, 2))
    Yt = torch.tensor(rng.standard_normal((m, k)).astype(np.float32), device=dev)
    model = SpectralNetModel({"n_layers": 2, "h": 4, "output_dim": k}, 3, dev)
    Wo = model._make_orthonorm_weights(Yt)
    assert rel(Wo.cpu().numpy(), g["orth"]) < TOL
    Y = (Yt @ torch.tensor(g["orth"], device=dev)).requires_grad_(True)
    assert rel(Y.detach().cpu().numpy()[g["grad_rows"]], g["Y_rows"]) < TOL
    for use_tiles in (True, False):          # one window of 512 rows at a time, or the row kernel
        ClusterBatch.use_tiles = use_tiles
        try:
            Y.grad = None
            loss = SpectralNetLoss()(b, Y, is_normalized=True)
            loss.backward()
            assert rel(loss.item(), g["loss"]) < TOL
            assert rel(Y.grad.cpu().numpy()[g["grad_rows"]], g["gradY_rows"]) < TOL
            assert rel(Y.grad.norm(dim=1).cpu().numpy(), g["gradY_rownorm"]) < TOL
            assert rel(SpectralNetLoss()(b, Y.detach(), is_normalized=False).item(), g["loss_plain"]) < TOL
        finally:
            ClusterBatch.use_tiles = True


def test_hip_graph_steps_match_eager_steps(dev, monkeypatch):
    """Whole steps replayed as HIP graphs (sparc_amd/stepgraph.py) follow the eager trainer: same losses and
    weights over eight steps that revisit every (gradient batch, orthogonality batch) pair of two batches,
    with a different permutation head every step; likewise when only two of the four pairs may be captured and
    the others keep running eagerly between the replays."""
    from sparc_amd import synth
    from sparc_amd.graph import ClusterBatch
    from sparc_amd.parallel import ShardedSpectralTrainer
    k, din = 12, 20
    arch = {"n_layers": 6, "hidden_dim1": 64, "hidden_dim2": 64, "hidden_dim3": 32, "output_dim": k}
    graph, bounds = synth.planted_cluster_graph(3000, 12000, 12, dev, seed=3)
    nodes_of, cluster_ids = synth.cluster_batches(bounds, 6, seed=0, return_clusters=True)
    from sparc_amd.stepgraph import GraphedSteps
    runs = {}
    for mode in (False, "same optimiser", True, "capped"):   # capped: two pairs captured at most, the other two stay eager
        monkeypatch.setattr(GraphedSteps, "MAX_GRAPHS", 2 if mode == "capped" else 64)
        tr = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=5, hip_graph=mode in (True, "capped"))
        if mode == "same optimiser":   # eager steps with the Adam variant the graphs need (learning rate in a tensor)
            tr.optimizer = torch.optim.Adam(tr.net.parameters(), lr=torch.tensor(1e-3, device=dev), fused=True, capturable=True)
        for b in range(2):
            coords, values, shp, _ = graph.subblock(nodes_of[b], mode=2, binarize=False, diag_lambda=0.01)
            tr.add_batch(synth.block_features(nodes_of[b], bounds, din, 12),
                         ClusterBatch.from_device_coo(shp[0], coords, values).warm(k))
        rng = np.random.RandomState(0)
        losses = []
        for s in range(8):
            pid, pid_2 = s % 2, (s // 2) % 2
            head = rng.permutation(tr.batches[pid_2].m)[:k]
            losses.append(float(tr.step(pid, pid_2, head)))
        runs[mode] = (np.array(losses), torch.cat([p.detach().flatten() for p in tr.params]).cpu().numpy())
        if mode in (True, "capped"):
            assert len(tr._graphed.graphs) == (2 if mode == "capped" else 4)
    for mode in (True, "capped"):
        assert np.all(np.isfinite(runs[mode][0]))
        assert rel(runs[mode][0], runs[False][0]) < 1e-5, (mode, runs[mode][0], runs[False][0])
        assert rel(runs[mode][1], runs[False][1]) < 1e-5
        # with the same optimiser arithmetic the replayed steps are the eager steps bit for bit
        assert np.array_equal(runs[mode][0], runs["same optimiser"][0])
        assert np.array_equal(runs[mode][1], runs["same optimiser"][1])


def test_hip_graph_step_with_fewer_rows_than_dimensions(dev):
    """An orthogonality batch of fewer than k rows (its Gram matrix cannot be positive definite: the reference's
    torch.inverse fails on it) goes through the graphed step as through the eager one: the permutation head is padded
    with the out-of-range row that reads as zeros, and the failure is reported by the same flag, not by an exception
    of the staging."""
    from sparc_amd.graph import ClusterBatch
    from sparc_amd.ops import OrthWeights
    from sparc_amd.parallel import ShardedSpectralTrainer
    k, din = 12, 6
    arch = {"n_layers": 4, "hidden_dim1": 32, "hidden_dim2": 32, "hidden_dim3": 16, "output_dim": k}
    rng = np.random.default_rng(0)
    sizes = (40, 8)                                   # the second batch has 8 < k rows
    for graphed in (False, True):
        tr = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=1, hip_graph=graphed)
        for m in sizes:
            A = sp.random(m, m, density=0.3, random_state=np.random.RandomState(m), format="csr")
            A.data[:] = 1.0
            A = sp.csr_matrix(sp.triu(A, 1) + sp.triu(A, 1).T + sp.eye(m))
            tr.add_batch(torch.tensor(rng.standard_normal((m, din)), dtype=torch.float32), ClusterBatch.from_support(O.as_support_tuple(A), dev))
        OrthWeights.check_and_reset(dev)
        loss_ok = tr.step(0, 0, np.random.RandomState(1).permutation(sizes[0])[:k])
        assert np.isfinite(float(loss_ok)) and OrthWeights.check_and_reset(dev) == 0
        for _ in range(2):                            # eager first visit, then the replay
            tr.step(0, 1, np.random.RandomState(2).permutation(sizes[1])[:k])
        assert OrthWeights.check_and_reset(dev) != 0


def test_training_separates_planted_clusters(dev, tmp_path, monkeypatch, capsys):
    """End to end on a graph with a known answer: six planted clusters, every batch holds a share of each (the
    reference's training input for num_clusters = 1, src/partition_utils.py:57-58), features that tell the clusters
    apart only through noise.  After 150 epochs the loss has dropped and the embedding of unseen nodes (predict)
    puts them next to their own cluster's centre: the step does what SpectralNet is for, not only what one reference
    step computes."""
    from sparc_amd.SpectralNet import SpectralNet
    from sparc_amd.SpectralTrainer import SpectralTrainer
    rng = np.random.default_rng(3)
    ncl, per, din = 6, 180, 16
    n = ncl * per
    label = np.repeat(np.arange(ncl), per)
    P = np.where(label[:, None] == label[None, :], 0.12, 0.002)
    A = sp.csr_matrix(np.triu(rng.random((n, n)) < P, 1).astype(np.float64))
    A = A + A.T + sp.eye(n)
    centres = rng.normal(size=(ncl, din))
    X = (centres[label] + 0.8 * rng.normal(size=(n, din))).astype(np.float32)
    order = rng.permutation(n)
    parts = [np.sort(order[0:n // 3]), np.sort(order[n // 3:2 * n // 3])]      # two training batches
    held = np.sort(order[2 * n // 3:])                                          # never trained on
    def batch(idx):
        return O.as_support_tuple(O.row_normalize(sp.csr_matrix(A[idx][:, idx])))
    monkeypatch.chdir(tmp_path)
    cfg = {"dataset": "planted", "spectral": {
        "architecture": {"n_layers": 4, "hidden_dim1": 64, "hidden_dim2": 64, "hidden_dim3": 32, "output_dim": ncl},
        "epochs": 150, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024, "n_clusters": ncl, "patience": 50}}
    torch.manual_seed(0)
    np.random.seed(0)
    net = SpectralTrainer(cfg, dev).train(
        ([p for p in parts], [torch.tensor(X[p]) for p in parts], [batch(p) for p in parts], [None] * 2, [None] * 2),
        ([0], [torch.tensor(X[held])], [batch(held)], [None], [None]))
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Epoch")]
    import re
    first, last = (float(re.findall(r"[-+]?\d+\.\d+", l)[0]) for l in (lines[0], lines[-1]))
    assert last < 0.6 * first, (first, last)
    facade = SpectralNet(n_clusters=ncl, config={"should_use_ae": False, "should_use_siamese": False})
    facade.spec_net = net
    Y = facade.predict(torch.tensor(X))
    cen = np.stack([Y[parts[0]][label[parts[0]] == c].mean(0) for c in range(ncl)])     # centres from trained nodes
    pred = np.argmin(((Y[held][:, None, :] - cen[None]) ** 2).sum(-1), axis=1)
    acc = float((pred == label[held]).mean())
    print("planted clusters: loss", first, "->", last, "held-out accuracy", acc)
    assert acc > 0.9, acc


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_one_launch_adam_matches_torch_adam(dev, weight_decay):
    """sparc_adam_step against torch.optim.Adam (the optimiser of src/SpectralTrainer.py:188) over several steps with
    a learning-rate change between them, on tensors of ragged sizes (tails that are no multiple of four, a view at an
    address that is no multiple of sixteen bytes, an empty tensor, one tensor without a gradient)."""
    from sparc_amd.optim import Adam
    torch.manual_seed(3)
    shapes = [(1024, 500), (1024,), (513, 7), (47,), (1,), (0,), (32, 33)]
    base = [torch.randn(s, device=dev) for s in shapes]
    odd = torch.randn(1 + 301, device=dev)
    mine = [torch.nn.Parameter(b.clone()) for b in base] + [torch.nn.Parameter(odd.clone()[1:])]
    ref = [torch.nn.Parameter(b.clone()) for b in base] + [torch.nn.Parameter(odd.clone()[1:])]
    assert mine[-1].data_ptr() % 16 != 0
    skipped = torch.nn.Parameter(torch.randn(5, device=dev))
    keep = skipped.detach().clone()
    a = Adam(mine + [skipped], lr=1e-3, weight_decay=weight_decay)
    b = torch.optim.Adam(ref, lr=1e-3, weight_decay=weight_decay)
    for step in range(6):
        if step == 3:
            for o in (a, b):
                o.param_groups[0]["lr"] = 1e-4
        for p, q in zip(mine, ref):
            g = torch.randn_like(p) * (10.0 ** (step - 3))
            p.grad, q.grad = g.clone(), g.clone()
        a.step()
        b.step()
        for i, (p, q) in enumerate(zip(mine, ref)):
            if p.numel():
                assert rel(p.detach().cpu().numpy(), q.detach().cpu().numpy()) < 1e-6, (step, i)
    for p, q in zip(mine, ref):
        if p.numel():
            assert rel(a.state[p]["exp_avg"].cpu().numpy(), b.state[q]["exp_avg"].cpu().numpy()) < 1e-6
            assert rel(a.state[p]["exp_avg_sq"].cpu().numpy(), b.state[q]["exp_avg_sq"].cpu().numpy()) < 1e-6
    assert torch.equal(skipped.detach(), keep)
    sd = a.state_dict()                                   # a torch Optimizer: its state round-trips
    a2 = Adam(mine + [skipped], lr=1.0)
    a2.load_state_dict(sd)
    assert a2.param_groups[0]["lr"] == 1e-4 and a2.param_groups[0]["step"] == 6
