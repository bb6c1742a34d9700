#!/usr/bin/env python3
"""Randomised bit-exactness check of the device support construction (DeviceGraph.subblock) against the oracle's SciPy
restatement of src/utils.py:22-47, :83, :126: random weighted graphs, random node subsets in random order, the three
flavours with and without binarisation.  A script (it uses the oracle as its checker), not collected by pytest:
python tests/fuzz_support.py [seeds] [first seed]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparc_oracle as O  # noqa: E402
from sparc_amd.utils import DeviceGraph  # noqa: E402

dev = torch.device("cuda:0")


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 3000))
        deg = rng.uniform(0.2, 30)
        A = sp.random(n, n, density=min(1.0, deg / n), random_state=np.random.RandomState(seed), format="csr")
        A.data[:] = rng.integers(1, 4, A.nnz).astype(np.float64)
        if rng.random() < 0.7:
            A = A + A.T
        if rng.random() < 0.7:
            A = A + sp.eye(n)
        A = sp.csr_matrix(A)
        graph = DeviceGraph.from_scipy(A, dev)
        for _ in range(4):
            m = int(rng.integers(1, n + 1))
            pt = rng.permutation(n)[:m].astype(np.int32)
            mode = int(rng.integers(0, 3))
            binarize = bool(rng.integers(0, 2))
            lam = float(rng.choice([0.01, 0.5, 1.0, 0.0]))
            sub = sp.csr_matrix(A[pt, :][:, pt])
            if binarize:
                sub = sp.csr_matrix((sub != 0) * 1.0)
            if mode == 1:
                rowsum_ref = np.asarray(sub.sum(axis=1)).ravel()
                ref = O.row_normalize(sub)
            elif mode == 2:
                rowsum_ref = np.asarray((sub + sp.eye(m)).sum(axis=1)).ravel()
                ref = O.row_normalize_diag_enhance(sub, lam)
            else:
                rowsum_ref = np.asarray(sub.sum(axis=1)).ravel()
                ref = sub
            coords, values, shape, rowsum = graph.subblock(pt, mode=mode, binarize=binarize, diag_lambda=lam)
            r0, c0, v0, sh0 = O.canonical_support(O.as_support_tuple(sp.csr_matrix(ref)))
            c = coords.cpu().numpy().reshape(-1, 2)
            ok = tuple(shape) == sh0 and np.array_equal(c[:, 0], r0) and np.array_equal(c[:, 1], c0) \
                and np.array_equal(values.cpu().numpy(), v0) and np.array_equal(rowsum.cpu().numpy(), rowsum_ref)
            if not ok:
                print(f"MISMATCH seed {seed}: n {n} m {m} mode {mode} binarize {binarize} lam {lam}: nnz {len(v0)} vs {c.shape[0]}", flush=True)
                if c.shape[0] == len(v0):
                    v = values.cpu().numpy()
                    bad = np.nonzero(v != v0)[0]
                    print("  coords equal:", np.array_equal(c[:, 0], r0) and np.array_equal(c[:, 1], c0), " values differing:", len(bad),
                          " rowsum equal:", np.array_equal(rowsum.cpu().numpy(), rowsum_ref))
                    for i in bad[:5]:
                        print(f"   entry ({r0[i]}, {c0[i]}): ours {v[i]!r} reference {v0[i]!r} rowsum {rowsum_ref[r0[i]]!r}")
                sys.exit(1)
        if seed % 20 == 0:
            print(f"seed {seed}: n {n} ok", flush=True)
    print("all bit-exact")


if __name__ == "__main__":
    main()
