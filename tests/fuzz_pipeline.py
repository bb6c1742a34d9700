#!/usr/bin/env python3
"""Randomised bit-exactness check of the data preparation around the step against the oracle's restatement of the
reference (src/utils.py:60-137, src/partition_utils.py:60-78), given the same parts: preprocess_multicluster (shuffle with
numpy's global RNG, cluster concatenation, induced sub-blocks, optional diagonal enhancement), partition_graph with
given groups (parts + intra-part adjacency), preprocess with given parts (normalised intra-part adjacency per part).
A script (it uses the oracle as its checker), not collected by pytest: python tests/fuzz_pipeline.py [seeds] [first]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparc_oracle as O  # noqa: E402
from sparc_amd import utils as U  # noqa: E402
from sparc_amd.partition import partition_graph  # noqa: E402

dev = torch.device("cuda:0")


def same_support(ours, ref):
    coords, values, shape = ours
    r0, c0, v0, sh0 = O.canonical_support(ref)
    r1, c1, v1, sh1 = O.canonical_support((coords.cpu().numpy(), values.cpu().numpy(), tuple(shape)))
    return sh0 == sh1 and np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(4, 1500))
        A = sp.random(n, n, density=min(1.0, rng.uniform(0.5, 12) / n), random_state=np.random.RandomState(seed), format="csr")
        A.data[:] = 1.0
        A = A + A.T
        if rng.random() < 0.5:
            A.data[:] = np.minimum(A.data, 2.0)            # weight 2 where both directions were drawn (the Cora loader)
        else:
            A.data[:] = 1.0
        A = sp.csr_matrix(A + sp.eye(n))
        nvis = int(rng.integers(2, n + 1))
        idx = np.sort(rng.permutation(n)[:nvis])
        ncl = int(rng.integers(1, max(2, min(nvis, 40))))
        groups = rng.integers(0, ncl, nvis)
        groups[:ncl] = np.arange(ncl)                      # no empty part
        feats = rng.standard_normal((n, 5)).astype(np.float32)
        y = rng.standard_normal((n, 3)).astype(np.float32)
        mask = rng.random(n) < 0.5
        lam = float(rng.choice([-1, 0.01, 0.5]))
        # partition_graph with the groups given
        part_adj, parts = partition_graph(A, idx, ncl, groups=groups)
        ref_adj, ref_parts = O.intra_part_adjacency(A, idx, groups, ncl)
        if (part_adj != ref_adj).nnz != 0 or part_adj.shape != ref_adj.shape or parts != ref_parts:
            print(f"MISMATCH partition_graph seed {seed}", flush=True); sys.exit(1)
        # preprocess with these parts: per-part supports of the normalised intra-part adjacency
        _, fb, sb, yb, mb = U.preprocess(part_adj, feats, y, mask, idx, ncl, diag_lambda=lam, parts=[np.array(p) for p in parts], device=dev)
        ref_sb = O.partition_batches(ref_adj, [np.array(p) for p in ref_parts], lam)
        for b, (ours, ref) in enumerate(zip(sb, ref_sb)):
            if not same_support(ours, ref):
                print(f"MISMATCH preprocess seed {seed} part {b} lam {lam}", flush=True); sys.exit(1)
            if not (np.array_equal(fb[b].numpy(), feats[parts[b]]) and np.array_equal(mb[b], mask[parts[b]])):
                print(f"MISMATCH preprocess features seed {seed} part {b}", flush=True); sys.exit(1)
        # preprocess_multicluster: the same shuffle, concatenation and sub-blocks
        bsize = int(rng.integers(1, ncl + 1))
        p_ours = [np.array(p) for p in parts]; p_ref = [np.array(p) for p in parts]
        np.random.seed(seed)
        fb, sb, yb, mb = U.preprocess_multicluster(A, p_ours, feats, y, mask, ncl, bsize, diag_lambda=lam, device=dev)
        np.random.seed(seed)
        node_lists, ref_sb = O.multicluster_batches(A, p_ref, ncl, bsize, lam)
        if len(sb) != len(ref_sb):
            print(f"MISMATCH multicluster batch count seed {seed}", flush=True); sys.exit(1)
        for b, (ours, ref) in enumerate(zip(sb, ref_sb)):
            if not same_support(ours, ref) or not np.array_equal(fb[b].numpy(), feats[node_lists[b]]):
                print(f"MISMATCH multicluster seed {seed} batch {b} lam {lam} bsize {bsize}", flush=True); sys.exit(1)
        if seed % 20 == 0:
            print(f"seed {seed} ok (n {n}, {ncl} parts)", flush=True)
    print("all bit-exact")


if __name__ == "__main__":
    main()
