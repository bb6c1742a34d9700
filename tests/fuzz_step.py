#!/usr/bin/env python3
"""Randomised check of one whole optimisation step (MLP on both batches, orthonormalisation with its Householder signs,
affinity, loss, and the backward through all of it including the orthonormalisation batch) against the oracle's dense
float64 restatement of src/SpectralTrainer.py:203-239, on random small problems.  A script (it uses the oracle as its
checker), not collected by pytest: python tests/fuzz_step.py [seeds] [first seed]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparc_oracle as O  # noqa: E402
from sparc_amd.SpectralTrainer import SpectralNetLoss, SpectralNetModel  # noqa: E402
from sparc_amd.graph import ClusterBatch  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def step_errors(seed):
    """One random problem: (description, errors of this package's step, errors of the dense float32 restatement,
    condition number of the orthonormalisation batch), all errors against the dense float64 restatement."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    k = int(rng.integers(2, 17))
    m1, m2 = int(rng.integers(k + 5, 500)), int(rng.integers(4 * k, 500))
    din = int(rng.integers(3, 40))
    arch = {"n_layers": 4, "hidden_dim1": int(rng.integers(8, 65)), "hidden_dim2": int(rng.integers(8, 65)),
            "hidden_dim3": int(rng.integers(8, 65)), "output_dim": k}
    ncl = int(rng.integers(1, 6))
    lab = np.sort(rng.integers(0, ncl, m1))
    P = np.where(lab[:, None] == lab[None, :], rng.uniform(0.03, 0.3), rng.uniform(0.0, 0.01))
    A = sp.csr_matrix(np.triu(rng.random((m1, m1)) < P, 1).astype(np.float64))
    A = A + A.T + sp.eye(m1)
    flavour = ["raw", "rownorm", "enh"][rng.integers(0, 3)]
    if flavour == "rownorm":
        A = O.row_normalize(A)
    elif flavour == "enh":
        A = O.row_normalize_diag_enhance(A, 0.01)
    support = O.as_support_tuple(sp.csr_matrix(A))
    x1 = torch.tensor(rng.standard_normal((m1, din)), dtype=torch.float32)
    x2 = torch.tensor(rng.standard_normal((m2, din)), dtype=torch.float32)
    perm2 = rng.permutation(m2)
    net = SpectralNetModel(arch, din, dev).to(dev)
    params = [p.detach().cpu() for p in net.parameters()]
    ref = O.train_step_dense(params, x1, support, x2, perm1=None, perm2=perm2, dtype=torch.float64)
    batch = ClusterBatch.from_support(support, dev, cluster_sizes=np.bincount(lab, minlength=ncl))
    Y = net.forward_pair(x2.to(dev), x1.to(dev), top_idx=perm2[:k])
    loss = SpectralNetLoss()(batch, Y, is_normalized=True)
    loss.backward()
    cond = float(np.linalg.cond(ref["Yt2"].numpy()))
    errs = {"loss": rel(loss.item(), ref["loss"].item()), "Y": rel(Y.detach().cpu().numpy(), ref["Y"].numpy()),
            "orth": rel(net.orthonorm_weights.detach().cpu().numpy(), ref["orth"].numpy())}
    # the output layer's bias gradient is a sum over the rows of terms that nearly cancel: its own, wider allowance
    gall = [rel(p.grad.cpu().numpy(), g.numpy()) for p, g in zip(net.parameters(), ref["grads"])]
    errs["grads"] = max(gall[:-1])
    errs["last bias"] = gall[-1]
    r32 = O.train_step_dense(params, x1, support, x2, perm1=None, perm2=perm2, dtype=torch.float32)
    e32 = {"loss": rel(r32["loss"].item(), ref["loss"].item()), "Y": rel(r32["Y"].numpy(), ref["Y"].numpy()),
           "orth": rel(r32["orth"].numpy(), ref["orth"].numpy()),
           "grads": max(rel(a.numpy(), b.numpy()) for a, b in zip(r32["grads"][:-1], ref["grads"][:-1])),
           "last bias": rel(r32["grads"][-1].numpy(), ref["grads"][-1].numpy())}
    # a ReLU pre-activation within float32 rounding of zero may fall on either side (seen: one flipped unit in one row
    # moves a gradient by 2.6e-3): such problems say nothing about the gradients
    tiny = np.inf
    for x in (x1, x2):
        h = x.double()
        for li in range(3):
            pre = h @ params[2 * li].double().t() + params[2 * li + 1].double()
            tiny = min(tiny, float((pre.abs() / pre.abs().max()).min()))
            h = torch.relu(pre)
    if tiny < 2e-7:
        errs["grads"] = errs["last bias"] = 0.0
    line = f"seed {seed}: m {m1}/{m2} k {k} din {din} {flavour} cond {cond:.0f}{' (a pre-activation next to zero: gradients not compared)' if tiny < 2e-7 else ''}: " + " ".join(f"{a} {b:.1e}" for a, b in errs.items())
    line += " | dense float32: " + " ".join(f"{a} {b:.1e}" for a, b in e32.items())
    return line, errs, e32, cond


def allowance(e32, cond):
    """float32 against float64: the errors grow with the condition number of the orthonormalisation batch (an untrained
    MLP's outputs are nearly collinear now and then; the float32 MLP outputs differ from the float64 ones by rounding,
    and the orthonormalisation passes that on times the condition number), for the reference's own float32 arithmetic
    as for these kernels: the yardstick is the dense float32 restatement's own error, three times over, and not below
    a floor that grows with the condition number."""
    grow = 1.0 + cond / 1000.0
    floor = {"loss": 2e-5 * grow, "Y": 2e-5 * grow, "orth": 2e-5 * grow, "grads": 2e-4 * grow, "last bias": 2e-3 * grow}
    return {a: max(floor[a], 3.0 * e32[a]) for a in floor}


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worst = {}
    for seed in range(first, first + n_seeds):
        line, errs, e32, cond = step_errors(seed)
        allow = allowance(e32, cond)
        bad = [a for a in errs if errs[a] > allow[a]]
        for a, b in errs.items():
            worst[a] = max(worst.get(a, 0.0), b / allow[a])
        if bad:
            print("MISMATCH", bad, line, flush=True)
            sys.exit(1)
        if seed % 10 == 0:
            print(line, flush=True)
    print("worst, as a fraction of the allowance max(floor, 3 x the dense float32 restatement's own error):", worst)


if __name__ == "__main__":
    main()
