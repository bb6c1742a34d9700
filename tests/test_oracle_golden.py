"""The CPU oracle against vectors produced by the imported reference.

Every golden file under tests/golden was written by tools/gen_golden.py from
the reference's own classes; these tests pin oracle/sparc_oracle.py to them.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import sparc_oracle as O


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def sup(g, prefix):
    return g[prefix + "coords"], g[prefix + "values"], tuple(int(x) for x in g[prefix + "shape"])


@pytest.mark.parametrize("name", ["raw", "rownorm", "enh"])
def test_affinity_dense_matches_reference(golden, name):
    g = golden("g2_affinity_loss.npz")
    W = O.valid_affinity(O.support_to_dense(sup(g, name + "_"), torch.float32))
    assert rel(W.numpy(), g[name + "_W"]) < 2e-6
    W64 = O.valid_affinity(O.support_to_dense(sup(g, name + "_"), torch.float64))
    assert rel(W64.numpy(), g[name + "_W"]) < 5e-6
    assert rel(W64.sum(1).numpy(), g[name + "_d"]) < 5e-6
    assert rel(W64.sum(0).numpy(), g[name + "_colsum"]) < 5e-6


@pytest.mark.parametrize("name", ["raw", "rownorm", "enh"])
def test_loss_matches_reference(golden, name):
    g = golden("g2_affinity_loss.npz")
    W = torch.tensor(g[name + "_W"], dtype=torch.float64)
    Y = torch.tensor(g["Y"], dtype=torch.float64)
    assert rel(O.spectral_loss(W, Y, True).numpy(), g[name + "_loss_norm"]) < 2e-5
    assert rel(O.spectral_loss(W, Y, False).numpy(), g[name + "_loss_plain"]) < 2e-5


@pytest.mark.parametrize("name", ["raw", "rownorm", "enh"])
def test_operator_form_equals_dense(golden, name):
    """The sparse operator form (what the kernels compute) vs the dense path."""
    g = golden("g2_affinity_loss.npz")
    s = sup(g, name + "_")
    A, _ = O.csr_from_support(s)
    prep = O.affinity_prep_operator(A)
    W = O.valid_affinity(O.support_to_dense(s, torch.float64))
    assert rel(prep["d"], W.sum(1).numpy()) < 1e-10
    assert rel(prep["cw"], W.sum(0).numpy()) < 1e-10
    Y = torch.tensor(g["Y"], dtype=torch.float64, requires_grad=True)
    loss = O.spectral_loss(W, Y, True)
    (gY,) = torch.autograd.grad(loss, Y)
    l_op, gY_op, WY, WtY = O.lap_forward_operator(A, prep, g["Y"].astype(np.float64))
    assert abs(l_op - loss.item()) / abs(loss.item()) < 1e-10
    assert rel(gY_op, gY.numpy()) < 1e-9
    Yn = (Y / torch.sqrt(W.sum(1))[:, None]).detach()
    assert rel(WY, (W @ Yn).numpy()) < 1e-10
    assert rel(WtY, (W.t() @ Yn).numpy()) < 1e-10
    # and against the reference's float32 numbers
    assert rel(l_op, g[name + "_loss_norm"]) < 2e-5


@pytest.mark.parametrize("name", ["gauss", "mlp", "wide"])
def test_orth_weights_match_reference(golden, name):
    g = golden("g3_orth.npz")
    Yt = torch.tensor(g[name + "_Yt"])
    tol = 5e-7 * float(g[name + "_cond"]) ** 2 + 1e-5
    assert rel(O.orthonorm_weights_qr(Yt).numpy(), g[name + "_orth"]) < 1e-5 + 2e-7 * float(g[name + "_cond"])
    s = O.householder_signs(Yt)
    ref_s = g[name + "_signs"]
    assert np.array_equal(s.numpy(), ref_s.astype(np.float64))
    Wo = O.orthonorm_weights_chol(Yt.double(), s)
    # the float32 Householder QR of the reference carries ~cond * eps error itself
    assert rel(Wo.numpy(), g[name + "_orth"]) < tol
    Yo = Yt.double() @ Wo
    err = (Yo.t() @ Yo / Yt.shape[0] - torch.eye(Yt.shape[1], dtype=torch.float64)).abs().max().item()
    assert abs(err - float(g[name + "_ortho_err"])) <= 1e-3 * max(1.0, float(g[name + "_ortho_err"]))


def test_orth_signs_are_mixed_somewhere(golden):
    g = golden("g3_orth.npz")
    assert any((g[n + "_signs"] < 0).any() and (g[n + "_signs"] > 0).any() for n in ["gauss", "mlp", "wide"])


def _params(g, prefix):
    return [torch.tensor(g[f"{prefix}layers.{i}.0.{w}"]) for i in range(4) for w in ("weight", "bias")]


def test_step_matches_reference(golden):
    g = golden("g5_step.npz")
    ps = _params(g, "init.")
    out = O.train_step_dense(ps, torch.tensor(g["x1"]), sup(g, "sup1_"), torch.tensor(g["x2"]),
                             g["perm1"], g["perm2"], dtype=torch.float64)
    assert bool(g["orth_requires_grad"])
    assert rel(out["loss"].numpy(), g["loss"]) < 1e-4
    assert rel(out["orth"].numpy(), g["orth"]) < 1e-4
    assert rel(out["Y"].numpy(), g["Y"]) < 1e-4
    names = [f"layers.{i}.0.{w}" for i in range(4) for w in ("weight", "bias")]
    for n, gr in zip(names, out["grads"]):
        assert rel(gr.numpy(), g["grad." + n]) < 5e-4, n
    # detached variant must NOT match (the reference differentiates through the orth batch)
    det = O.train_step_dense(ps, torch.tensor(g["x1"]), sup(g, "sup1_"), torch.tensor(g["x2"]),
                             g["perm1"], g["perm2"], dtype=torch.float64, detach_orth=True)
    assert rel(det["grads"][0].numpy(), g["grad.layers.0.0.weight"]) > 1e-2
    # Adam
    new = O.adam_update([p.double() for p in ps], out["grads"], {}, lr=1e-3)
    for n, p in zip(names, new):
        assert rel(p.numpy(), g["after." + n]) < 1e-5, n
    # validation with the carried orth weights
    after = _params(g, "after.")
    vl = O.validation_loss_dense(after, torch.tensor(g["x2"]), sup(g, "supv_"), torch.tensor(g["orth"]), torch.float64)
    assert rel(vl.numpy(), g["val_loss"]) < 1e-4


def test_step_operator_form_matches_dense(golden):
    """Gradient through the operator form + Cholesky backward == dense autograd."""
    g = golden("g5_step.npz")
    ps = [p.double() for p in _params(g, "init.")]
    s1 = sup(g, "sup1_")
    out = O.train_step_dense(ps, torch.tensor(g["x1"]), s1, torch.tensor(g["x2"]), g["perm1"], g["perm2"], dtype=torch.float64)
    # operator form on the permuted support
    perm1 = g["perm1"]
    inv = np.empty_like(perm1)
    inv[perm1] = np.arange(len(perm1))
    coords = s1[0]
    sp_perm = (np.stack([inv[coords[:, 0]], inv[coords[:, 1]]], 1), s1[1], s1[2])
    A, _ = O.csr_from_support(sp_perm)
    prep = O.affinity_prep_operator(A)
    loss, gY, _, _ = O.lap_forward_operator(A, prep, out["Y"].numpy())
    assert abs(loss - out["loss"].item()) / out["loss"].item() < 1e-9
    Yt1, Yt2, Wo = out["Yt1"].numpy(), out["Yt2"].numpy(), out["orth"].numpy()
    gYt1 = gY @ Wo.T
    gWo = Yt1.T @ gY
    signs = O.householder_signs(out["Yt2"]).numpy()
    gYt2 = O.orth_backward_operator(Yt2, signs, gWo)
    # push both through the MLP with autograd and compare parameter grads
    ps_r = [p.clone().requires_grad_(True) for p in ps]
    x1 = torch.tensor(g["x1"]).double()[torch.as_tensor(perm1)]
    x2 = torch.tensor(g["x2"]).double()[torch.as_tensor(g["perm2"])]
    y1 = O.mlp_forward(ps_r, x1)
    y2 = O.mlp_forward(ps_r, x2)
    grads = torch.autograd.grad([y1, y2], ps_r, [torch.tensor(gYt1), torch.tensor(gYt2)])
    for a, b in zip(grads, out["grads"]):
        assert rel(a.numpy(), b.numpy()) < 1e-8


def _csr(g, prefix):
    n = int(g["n"])
    return sp.csr_matrix((g[prefix + "data"], g[prefix + "indices"], g[prefix + "indptr"]), shape=(n, n))


@pytest.mark.parametrize("li,lam", [(0, -1), (1, 0.01)])
def test_multicluster_supports_bit_exact(golden, li, lam):
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    sizes = g[f"mc{li}_parts_sizes"]
    flat = g[f"mc{li}_parts_after_shuffle"]
    parts = np.split(flat, np.cumsum(sizes)[:-1])

    class NoShuffle:
        def shuffle(self, x):
            pass
    nodes, sups = O.multicluster_batches(train_adj, list(parts), 6, 2, lam, shuffle_rng=NoShuffle())
    assert len(sups) == int(g[f"mc{li}_nbatch"])
    for b, s in enumerate(sups):
        r0, c0, v0, sh0 = O.canonical_support((g[f"mc{li}_coords{b}"], g[f"mc{li}_values{b}"], g[f"mc{li}_shape{b}"]))
        r1, c1, v1, sh1 = O.canonical_support(s)
        assert sh0 == sh1 and np.array_equal(r0, r1) and np.array_equal(c0, c1)
        assert np.array_equal(v0, v1)  # bit-exact float64


def test_multicluster_shuffle_order(golden):
    """np.random.shuffle(parts) under seed 5 reproduces the reference's batch order."""
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    idx, groups = g["idx_nodes"], g["groups"]
    parts = [np.array(idx[groups == c]) for c in range(6)]
    np.random.seed(5)
    nodes, sups = O.multicluster_batches(train_adj, parts, 6, 2, -1)
    assert np.array_equal(np.concatenate(parts), g["mc0_parts_after_shuffle"])
    assert np.array_equal(np.asarray(sups[0][0]), g["mc0_coords0"])  # same storage order too


@pytest.mark.parametrize("li,lam", [(0, -1), (1, 0.01)])
def test_validation_supports_bit_exact(golden, li, lam):
    g = golden("g1_support.npz")
    adj = _csr(g, "adj_")
    n = int(g["n"])
    groups = g["labels"] % 3
    part_adj, parts = O.intra_part_adjacency(adj, np.arange(n), groups, 3)
    sups = O.partition_batches(part_adj, [np.array(p) for p in parts], lam)
    for b, s in enumerate(sups):
        assert np.array_equal(np.asarray(parts[b]), g[f"val{li}_part{b}"])
        r0, c0, v0, sh0 = O.canonical_support((g[f"val{li}_coords{b}"], g[f"val{li}_values{b}"], g[f"val{li}_shape{b}"]))
        r1, c1, v1, sh1 = O.canonical_support(s)
        assert sh0 == sh1 and np.array_equal(r0, r1) and np.array_equal(c0, c1)
        assert np.array_equal(v0, v1)


def test_intra_part_adjacency_bit_exact(golden):
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    part_adj, parts = O.intra_part_adjacency(train_adj, g["idx_nodes"], g["groups"], 6)
    ref = _csr(g, "part_adj_")
    assert (part_adj != ref).nnz == 0
    assert part_adj.dtype == ref.dtype


def test_degree_arrays_bit_exact(golden):
    g = golden("g1_support.npz")
    train_adj = _csr(g, "train_")
    assert np.array_equal(np.asarray(train_adj.sum(axis=1)).ravel(), g["rowsum_raw"])


def test_token_lists_oracle_matches_reference(golden):
    """The consumer-side token lists (implementations/SPARCphormer/utils.py:268-298)."""
    g = golden("g10_tokens.npz")
    t = O.spectral_token_lists(g["encoding"], g["features"], int(g["K"]))
    assert t.shape == g["tokens"].shape and t.dtype == np.float32
    assert np.abs(t - g["tokens"]).max() < 1e-6


# ---------------------------------------------------------------- real graph: Cora through the reference's loader
def _cora_yt(m, k=64):
    """The pre-orthonormalisation output tools/gen_golden_cora.py drew (same generator, same call order)."""
    rng = np.random.default_rng(7)
    rng.integers(0, m, size=(256, 2))
    return rng.standard_normal((m, k)).astype(np.float32)


def test_cora_plumbing_counts(golden):
    """The facts of the shipped Cora graph as the reference's loader sees it (src/utils.py:140-255): the
    split, the stored-entry counts, off-diagonal weight 2 (both link directions of the multigraph add up)."""
    g = golden("g7_cora.npz")
    assert (int(g["num_data"]), int(g["n_train"]), int(g["n_val"]), int(g["n_test"])) == (2708, 2437, 189, 82)
    assert int(g["train_adj_nnz"]) == 11312 and int(g["full_adj_nnz"]) == 13264
    assert g["offdiag_values"].tolist() == [2.0] and g["diag_values"].tolist() == [1.0]
    assert tuple(g["sup_shape"]) == (2437, 2437) and len(g["sup_values"]) == 11041
    assert np.allclose(g["val_rowsums"], 1.0, atol=1e-12)       # validation supports are row-normalised


def test_cora_oracle_support_and_loss(golden):
    g = golden("g7_cora.npz")
    n = int(g["num_data"])
    train_adj = sp.csr_matrix((g["train_data_values"], g["train_indices"], g["train_indptr"]), shape=(n, n))
    parts = [g["train_part"].copy()]
    nodes, sups = O.multicluster_batches(train_adj, parts, 1, 100000, -1, shuffle_rng=np.random.RandomState(0))
    ours, ref = O.canonical_support(sups[0]), O.canonical_support(sup(g, "sup_"))
    for a, b in zip(ours[:3], ref[:3]):
        assert np.array_equal(a, b)
    full = sp.csr_matrix((g["full_data"], g["full_indices"], g["full_indptr"]), shape=(n, n))
    part_adj, vparts = O.intra_part_adjacency(full, np.arange(n), np.zeros(n, int), 1)
    vs = O.partition_batches(part_adj, [np.asarray(p) for p in vparts], -1)[0]
    ours, ref = O.canonical_support(vs), O.canonical_support(sup(g, "val_"))
    assert np.array_equal(ours[0], ref[0]) and np.array_equal(ours[1], ref[1]) and np.array_equal(ours[2], ref[2])
    # affinity, orthonormalisation, loss and gradient at m = 2437, k = 64
    s = sup(g, "sup_")
    A, _ = O.csr_from_support(s)
    prep = O.affinity_prep_operator(A)
    assert rel(prep["d"], g["W_rowsum"]) < 2e-5 and rel(prep["cw"], g["W_colsum"]) < 2e-5
    Yt = _cora_yt(A.shape[0])
    assert abs(Yt.astype(np.float64).sum() - float(g["Yt_sum"])) < 1e-6 * float(g["Yt_abs_sum"])
    Wo = O.orthonorm_weights_qr(torch.tensor(Yt, dtype=torch.float64)).numpy()
    assert rel(Wo, g["orth"]) < 1e-4
    Y = Yt.astype(np.float64) @ g["orth"].astype(np.float64)
    loss, gY, _, _ = O.lap_forward_operator(A, prep, Y)
    assert rel(loss, g["loss"]) < 2e-5
    assert rel(gY[g["grad_rows"]], g["gradY_rows"]) < 1e-4
    assert rel(np.linalg.norm(gY, axis=1), g["gradY_rownorm"]) < 1e-4
