"""CPU oracle for the SpectralNet training step of shaham-lab/SPARC.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

It restates, with dense torch/numpy ops on the CPU, what the reference does on
its hot path (all citations relative to the reference checkout):

  * COO support tuple -> dense matrix            src/SpectralTrainer.py:275-281
  * dense affinity  W+1e-6, W+W@W+W@W@W, ...     src/SpectralTrainer.py:284-293
  * MLP forward                                  src/SpectralTrainer.py:28-39, 96-97
  * orthonormalisation weights (QR, sign, inv)   src/SpectralTrainer.py:42-70
  * spectral loss                                src/SpectralTrainer.py:116-139
  * one optimisation step (orth step + grad)     src/SpectralTrainer.py:203-239
  * validation loss                              src/SpectralTrainer.py:242-258
  * support / degree construction                src/utils.py:22-137

and, separately, the sparse *operator form* of the same mathematics that the
HIP kernels implement (chained CSR products + rank-3 correction), so that each
kernel can be checked in isolation.  The dense functions are pinned against
vectors produced by the imported reference (tests/golden, tools/gen_golden.py);
the operator-form functions are pinned against the dense ones.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import torch

EPS_BACKGROUND = 1e-6  # src/SpectralTrainer.py:288


# --------------------------------------------------------------------------
# dense restatement (what the reference literally evaluates)
# --------------------------------------------------------------------------
def support_to_dense(support, dtype=torch.float32) -> torch.Tensor:
    """COO (coords[nnz,2], values[nnz], shape) -> dense; duplicates add up.

    src/SpectralTrainer.py:275-281 builds an uncoalesced sparse tensor from
    float32-cast values and densifies it, which sums repeated coordinates.
    """
    coords, values, shape = support
    coords = np.asarray(coords).reshape(-1, 2)
    m0, m1 = int(shape[0]), int(shape[1])
    vals32 = torch.as_tensor(np.asarray(values, dtype=np.float64)).to(torch.float32)
    dense = torch.zeros(m0 * m1, dtype=dtype)
    flat = torch.as_tensor(coords[:, 0].astype(np.int64) * m1 + coords[:, 1].astype(np.int64))
    dense.index_add_(0, flat, vals32.to(dtype))
    return dense.view(m0, m1)


def valid_affinity(A: torch.Tensor) -> torch.Tensor:
    """src/SpectralTrainer.py:284-293 on a dense support matrix."""
    W0 = A + EPS_BACKGROUND
    W2 = W0 @ W0
    Wp = W0 + W2 + W2 @ W0
    Wp.fill_diagonal_(0)
    rho = Wp.norm(dim=1).clamp_min(1e-12)  # F.normalize(p=2, dim=1, eps=1e-12)
    return Wp / rho[:, None]


def mlp_forward(params: Sequence[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Linear+ReLU for all but the last pair, Linear+Tanh for the last.

    params = [w0, b0, w1, b1, ...] in nn.Linear layout (out, in).
    src/SpectralTrainer.py:28-39, 96-97.
    """
    n = len(params) // 2
    for i in range(n):
        x = x @ params[2 * i].t() + params[2 * i + 1]
        x = torch.tanh(x) if i == n - 1 else torch.relu(x)
    return x


def orthonorm_weights_qr(Yt: torch.Tensor) -> torch.Tensor:
    """src/SpectralTrainer.py:62-68: sqrt(m) * inverse(R @ diag(sign(diag R)))."""
    m = Yt.shape[0]
    _, R = torch.linalg.qr(Yt)
    D = torch.diag(torch.sign(torch.diag(R)))
    return math.sqrt(m) * torch.inverse(R @ D)


def householder_signs(Yt: torch.Tensor) -> torch.Tensor:
    """sign(diag R) of LAPACK's Householder QR, computed without a QR.

    With Rc the positive-diagonal Cholesky factor of Yt^T Yt, the Householder R
    equals diag(s) Rc, and s is the sign sequence of the 'modified LU' of the
    leading k x k block of the orthonormal factor Yt Rc^-1.
    """
    k = Yt.shape[1]
    Y64 = Yt.detach().to(torch.float64)
    Rc = torch.linalg.cholesky(Y64.t() @ Y64).t()
    Q = torch.linalg.solve_triangular(Rc, Y64[:k], upper=True, left=False).clone()
    s = torch.ones(k, dtype=torch.float64)
    for i in range(k):
        s[i] = -1.0 if Q[i, i] > 0 else 1.0  # s_i = -sign(Q_ii), sign(0) -> +1
        Q[i, i] -= s[i]
        if i + 1 < k:
            Q[i + 1:, i] /= Q[i, i]
            Q[i + 1:, i + 1:] -= torch.outer(Q[i + 1:, i], Q[i, i + 1:])
    return s


def orthonorm_weights_chol(Yt: torch.Tensor, signs: torch.Tensor | None = None) -> torch.Tensor:
    """sqrt(m) * S Rc^-1 S  (algebraically equal to orthonorm_weights_qr).

    Differentiable through the Gram matrix; S is treated as a constant.
    """
    m, k = Yt.shape
    if signs is None:
        signs = householder_signs(Yt)
    G = Yt.t() @ Yt
    Rc = torch.linalg.cholesky(G).t()
    Rinv = torch.linalg.solve_triangular(Rc, torch.eye(k, dtype=Yt.dtype), upper=True)
    s = signs.to(Yt.dtype)
    return math.sqrt(m) * (s[:, None] * Rinv * s[None, :])


def spectral_loss(W: torch.Tensor, Y: torch.Tensor, is_normalized: bool = False) -> torch.Tensor:
    """src/SpectralTrainer.py:130-137: sum(W * cdist(Y,Y)^2) / (2m)."""
    m = Y.shape[0]
    if is_normalized:
        d = W.sum(dim=1)
        Y = Y / torch.sqrt(d[:, None])
    sq = (Y * Y).sum(dim=1)
    D2 = (sq[:, None] + sq[None, :] - 2.0 * (Y @ Y.t())).clamp_min(0)
    return (W * D2).sum() / (2 * m)


def train_step_dense(params: Sequence[torch.Tensor], x1: torch.Tensor, support1,
                     x2: torch.Tensor, perm1=None, perm2=None, dtype=torch.float32,
                     detach_orth: bool = False) -> Dict[str, torch.Tensor]:
    """One pass of the loop body src/SpectralTrainer.py:203-239 (without Adam).

    Returns loss, Y, orth weights and the gradient of every parameter.  The
    orthonormalisation batch is *not* detached (the reference runs it with
    autograd enabled, :222-223), so its graph contributes to the gradients.
    """
    ps = [p.detach().to(dtype).clone().requires_grad_(True) for p in params]
    A = support_to_dense(support1, dtype)
    x1 = x1.to(dtype)
    x2 = x2.to(dtype)
    if perm1 is not None:
        p1 = torch.as_tensor(np.asarray(perm1), dtype=torch.long)
        x1 = x1[p1]
        A = A[p1][:, p1]
    if perm2 is not None:
        x2 = x2[torch.as_tensor(np.asarray(perm2), dtype=torch.long)]
    Yt2 = mlp_forward(ps, x2)
    Wo = orthonorm_weights_qr(Yt2)
    if detach_orth:
        Wo = Wo.detach()
    Yt1 = mlp_forward(ps, x1)
    Y = Yt1 @ Wo
    W = valid_affinity(A)
    loss = spectral_loss(W, Y, is_normalized=True)
    grads = torch.autograd.grad(loss, ps)
    return {"loss": loss.detach(), "Y": Y.detach(), "orth": Wo.detach(),
            "Yt1": Yt1.detach(), "Yt2": Yt2.detach(), "grads": [g.detach() for g in grads]}


def adam_update(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (src/SpectralTrainer.py:188, 238)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = []
    for i, (p, g) in enumerate(zip(params, grads)):
        mkey, vkey = ("m", i), ("v", i)
        state[mkey] = betas[0] * state.get(mkey, torch.zeros_like(p)) + (1 - betas[0]) * g
        state[vkey] = betas[1] * state.get(vkey, torch.zeros_like(p)) + (1 - betas[1]) * g * g
        mhat = state[mkey] / (1 - betas[0] ** t)
        denom = (state[vkey].sqrt() / math.sqrt(1 - betas[1] ** t)) + eps
        out.append(p - lr * mhat / denom)
    return out


def validation_loss_dense(params, x, support, orth, dtype=torch.float32) -> torch.Tensor:
    """src/SpectralTrainer.py:246-258 for one validation batch."""
    with torch.no_grad():
        ps = [p.to(dtype) for p in params]
        Y = mlp_forward(ps, x.to(dtype)) @ orth.to(dtype)
        W = valid_affinity(support_to_dense(support, dtype))
        return spectral_loss(W, Y, is_normalized=True)


# --------------------------------------------------------------------------
# support / degree construction  (src/utils.py:22-137)
# --------------------------------------------------------------------------
def row_normalize(adj: sp.spmatrix) -> sp.csr_matrix:
    """diag(1/max(1,rowsum)) @ adj          src/utils.py:22-27."""
    adj = sp.csr_matrix(adj)
    rowsum = np.asarray(adj.sum(axis=1)).ravel()
    scale = 1.0 / np.maximum(1.0, rowsum)
    return sp.csr_matrix(sp.diags(scale, 0) @ adj)


def row_normalize_diag_enhance(adj: sp.spmatrix, lam: float) -> sp.csr_matrix:
    """(D+I)^-1 (A+I), then add lam*diag of the result   src/utils.py:30-38."""
    adj = sp.csr_matrix(adj) + sp.eye(adj.shape[0])
    rowsum = np.asarray(adj.sum(axis=1)).ravel()
    scale = 1.0 / (rowsum + 1e-20)
    adj = sp.diags(scale, 0) @ adj
    return sp.csr_matrix(adj + lam * sp.diags(adj.diagonal(), 0))


def as_support_tuple(mx: sp.spmatrix):
    """COO triple (coords[nnz,2] (row,col), values, shape)   src/utils.py:41-47."""
    coo = mx.tocoo()
    return np.stack([coo.row, coo.col], axis=1), coo.data, coo.shape


def canonical_support(support):
    """Sort a support tuple by (row, col) and merge duplicates (order-free view)."""
    coords, values, shape = support
    coords = np.asarray(coords).reshape(-1, 2).astype(np.int64)
    key = coords[:, 0] * int(shape[1]) + coords[:, 1]
    order = np.argsort(key, kind="stable")
    key = key[order]
    vals = np.asarray(values, dtype=np.float64)[order]
    uniq, start = np.unique(key, return_index=True)
    merged = np.add.reduceat(vals, start) if len(vals) else vals
    rows = (uniq // int(shape[1])).astype(np.int32)
    cols = (uniq % int(shape[1])).astype(np.int32)
    return rows, cols, merged, (int(shape[0]), int(shape[1]))


def multicluster_batches(adj: sp.csr_matrix, parts: List[np.ndarray], num_clusters: int,
                         block_size: int, diag_lambda: float = -1, shuffle_rng=None):
    """Cluster concatenation + induced sub-block per batch   src/utils.py:75-90.

    `parts` is shuffled in place with numpy's global RNG unless shuffle_rng is
    given (the reference calls np.random.shuffle(parts), :75).
    Returns (node lists, support tuples).
    """
    (shuffle_rng or np.random).shuffle(parts)
    node_lists, supports = [], []
    for st in range(0, num_clusters, block_size):
        pt = np.concatenate([np.asarray(p) for p in parts[st:min(st + block_size, num_clusters)]])
        if min(st + block_size, num_clusters) - st > 1:
            pt = pt.astype(np.int32)  # the reference casts only when it concatenated (:78-80)
        sub = adj[pt, :][:, pt]
        if diag_lambda != -1:
            sub = row_normalize_diag_enhance(sub, diag_lambda)
        node_lists.append(pt)
        supports.append(as_support_tuple(sub))
    return node_lists, supports


def intra_part_adjacency(adj: sp.csr_matrix, idx_nodes: np.ndarray, groups: Sequence[int], num_clusters: int):
    """Parts and the 0/1 adjacency restricted to intra-part edges.

    src/partition_utils.py:60-78 (everything after the METIS call; `groups` is
    the cluster id of idx_nodes[i]).  Every stored entry of `adj` between two
    nodes of one part contributes 1; the matrix is sized by a zero-valued
    (n-1, n-1) sentinel.
    """
    n = adj.shape[0]
    groups = np.asarray(groups)
    idx_nodes = np.asarray(idx_nodes)
    grp_of = np.full(n, -1, dtype=np.int64)
    grp_of[idx_nodes] = groups
    parts = [idx_nodes[groups == g].tolist() for g in range(num_clusters)]
    coo = sp.csr_matrix(adj)[idx_nodes].tocoo()
    rows = idx_nodes[coo.row]
    keep = grp_of[coo.col] == grp_of[rows]
    r = np.concatenate([rows[keep], [n - 1]])
    c = np.concatenate([coo.col[keep], [n - 1]])
    v = np.concatenate([np.ones(int(keep.sum()), dtype=np.int64), [0]])
    return sp.coo_matrix((v, (r, c))).tocsr(), parts


def partition_batches(part_adj: sp.csr_matrix, parts: List[np.ndarray], diag_lambda: float = -1):
    """Per-part supports of a normalised intra-part adjacency  src/utils.py:113-128."""
    norm = row_normalize(part_adj) if diag_lambda == -1 else row_normalize_diag_enhance(part_adj, diag_lambda)
    return [as_support_tuple(norm[np.asarray(pt), :][:, np.asarray(pt)]) for pt in parts]


# --------------------------------------------------------------------------
# operator form (what the HIP kernels compute), float64 numpy
# --------------------------------------------------------------------------
def csr_from_support(support) -> Tuple[sp.csr_matrix, sp.csr_matrix]:
    """Canonical CSR of the float32-cast support and of its transpose."""
    v32 = np.asarray(support[1], np.float64).astype(np.float32)
    rows, cols, vals, shape = canonical_support((support[0], v32, support[2]))
    # repeated coordinates: torch's sparse-to-dense (src/SpectralTrainer.py:277) adds the float32 values one after the
    # other in input order; canonical_support's float64 sum agrees up to two copies, beyond that the order matters
    coords = np.asarray(support[0]).reshape(-1, 2).astype(np.int64)
    key = coords[:, 0] * int(shape[1]) + coords[:, 1]
    uniq = rows.astype(np.int64) * int(shape[1]) + cols
    if len(key) != len(uniq):
        acc = np.zeros(len(uniq), dtype=np.float32)
        np.add.at(acc, np.searchsorted(uniq, key), v32)       # unbuffered: sequential float32 additions
        vals = acc
    A = sp.csr_matrix((np.asarray(vals, np.float32).astype(np.float64), (rows, cols)), shape=shape)
    return A, sp.csr_matrix(A.T)


def _coupling(m: int, s: float, eps: float = EPS_BACKGROUND) -> np.ndarray:
    e, e2, e3 = eps, eps * eps, eps ** 3
    return np.array([[e + e2 * m + e2 * s + e3 * m * m, e + e2 * m, e],
                     [e + e2 * m, e, 0.0],
                     [e, 0.0, 0.0]])


def affinity_prep_operator(A: sp.csr_matrix, eps: float = EPS_BACKGROUND) -> Dict[str, np.ndarray]:
    """Per-batch vectors such that, with S = A + A^2 + A^3,

        W = diag(1/rho) (S + U C V^T - diag(g)),   U=[1 r r2], V=[1 c c2].

    The explicit rows of S are only used for rho and g (done once per batch).
    """
    m = A.shape[0]
    one = np.ones(m)
    At = sp.csr_matrix(A.T)
    r, c = A @ one, At @ one
    r2, c2 = A @ r, At @ c
    s = float(r.sum())
    C = _coupling(m, s, eps)
    U = np.stack([one, r, r2], axis=1)
    V = np.stack([one, c, c2], axis=1)
    S = sp.csr_matrix(A + A @ A + A @ A @ A)
    Sdiag = S.diagonal()
    Soff = S - sp.diags(Sdiag, 0)
    low_diag = np.einsum("ia,ab,ib->i", U, C, V)
    g = Sdiag + low_diag
    UC = U @ C
    sumsq = np.asarray(Soff.multiply(Soff).sum(axis=1)).ravel()
    SV = Soff @ V
    cross = np.einsum("ia,ia->i", UC, SV)
    low_sq = np.einsum("ia,ab,ib->i", UC, V.T @ V, UC) - low_diag ** 2
    rho = np.sqrt(sumsq + 2 * cross + low_sq)
    chain1 = S @ one
    d = (chain1 + UC @ (V.T @ one) - g) / rho
    St = sp.csr_matrix(S.T)
    x = 1.0 / rho
    cw = St @ x + V @ (C.T @ (U.T @ x)) - g * x
    return {"r": r, "c": c, "r2": r2, "c2": c2, "s": s, "C": C, "g": g, "sdiag": Sdiag, "rho": rho, "d": d, "cw": cw}


def lap_forward_operator(A: sp.csr_matrix, prep: Dict[str, np.ndarray], Y: np.ndarray):
    """loss = tr(Yn^T (1/2 (D_r + D_c) - W) Yn)/m  and  dloss/dY, Yn = Y/sqrt(d).

    Equal to spectral_loss(valid_affinity(A), Y, True) and its gradient.
    """
    m = A.shape[0]
    At = sp.csr_matrix(A.T)
    one = np.ones(m)
    U = np.stack([one, prep["r"], prep["r2"]], axis=1)
    V = np.stack([one, prep["c"], prep["c2"]], axis=1)
    C, g, rho, d, cw = prep["C"], prep["g"], prep["rho"], prep["d"], prep["cw"]
    Yn = Y / np.sqrt(d)[:, None]

    def chain(M, X):
        P = X + M @ X
        P = X + M @ P
        return M @ P

    WY = (chain(A, Yn) + U @ (C @ (V.T @ Yn)) - g[:, None] * Yn) / rho[:, None]
    X2 = Yn / rho[:, None]
    WtY = chain(At, X2) + V @ (C.T @ (U.T @ X2)) - g[:, None] * X2
    LY = 0.5 * (d + cw)[:, None] * Yn - WY
    loss = float((Yn * LY).sum() / m)
    gYn = ((d + cw)[:, None] * Yn - WY - WtY) / m
    return loss, gYn / np.sqrt(d)[:, None], WY, WtY


def orth_backward_operator(Yt2: np.ndarray, signs: np.ndarray, gWo: np.ndarray) -> np.ndarray:
    """d loss / d Yt2 given d loss / d Wo, for Wo = sqrt(m) S Rc^-1 S, Rc = chol(Yt2^T Yt2)^T."""
    m, k = Yt2.shape
    Y64 = Yt2.astype(np.float64)
    G = Y64.T @ Y64
    L = np.linalg.cholesky(G)               # lower, G = L L^T, Rc = L^T
    Linv = np.linalg.inv(L)
    M = Linv.T                              # Rc^-1
    Mbar = math.sqrt(m) * (signs[:, None] * gWo * signs[None, :])
    Rbar = -M.T @ Mbar @ M.T                # grad wrt Rc (upper)
    Lbar = np.tril(Rbar.T)                  # grad wrt L = Rc^T, lower part only
    P = L.T @ Lbar
    Phi = np.tril(P)
    Phi[np.diag_indices(k)] *= 0.5
    Gbar = Linv.T @ Phi @ Linv
    Gsym = 0.5 * (Gbar + Gbar.T)
    return Y64 @ (2.0 * Gsym)


# --------------------------------------------------------------------------- token lists (consumer side)
def spectral_token_lists(spectral_encoding: np.ndarray, features: np.ndarray, K: int) -> np.ndarray:
    """Per-node token sequence of the downstream transformer models
    (implementations/SPARCphormer/utils.py:268-298, same function in implementations/SAMBA/utils.py:239):
    token 0 is the node's own feature row, token j+1 the mean feature row of its 2^(j+1) nearest
    nodes in the spectral embedding (Euclidean, the node itself included), at most min(2^14, n)
    neighbours; beyond the listed prefix lengths the mean runs over all of them."""
    from sklearn.neighbors import NearestNeighbors
    n = len(spectral_encoding)
    knn = NearestNeighbors(n_neighbors=min(2 ** 14, n), algorithm="auto")
    knn.fit(spectral_encoding)
    _, indices = knn.kneighbors(spectral_encoding)
    feats = np.asarray(features, dtype=np.float32)
    out = np.empty((n, K + 1, feats.shape[1]), dtype=np.float32)
    out[:, 0, :] = feats
    starts = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
    for j in range(K):
        end = starts[j + 1] if j + 1 < len(starts) else indices.shape[1]
        out[:, j + 1, :] = torch.mean(torch.from_numpy(feats)[torch.from_numpy(indices[:, :end])], dim=1).numpy()
    return out
