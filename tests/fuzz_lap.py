#!/usr/bin/env python3
"""Randomised cross-check of the Laplacian pass on the GPU: random cluster batches (cluster counts and sizes, densities,
support flavours, k, training / forward-only) through every kernel path the batch offers (row kernel, windowed kernel
with value streams and in row-factor form, dense blocks), all against the row kernel, and the small ones against the
CPU oracle as well.  Not a test (minutes of GPU time): python tests/fuzz_lap.py [seeds] [first seed]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparc_oracle as O  # noqa: E402  (checker only)
from sparc_amd.graph import ClusterBatch  # noqa: E402
from tests.test_gpu_tiled import windowed_pattern, make_support  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if np.isnan(a).all() and np.isnan(b).all():
        return 0.0          # a single-node batch: W is the 1 x 1 zero matrix and the loss 0 / 0, for the reference too
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def batch_of(rng):
    kind = rng.integers(0, 5)
    if kind == 4:      # loose ends: isolated nodes, rows with a diagonal entry only, no stored diagonal at all, free weights
        m = int(rng.integers(1, 1400))
        sizes = [m]
        D = sp.random(m, m, density=min(1.0, rng.uniform(0.0, 6.0) / m), random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
        D.data[:] = rng.uniform(0.2, 3.0, D.nnz)
        D = sp.triu(D, 1)
        D = D + D.T
        keep = rng.random(m) < 0.8                     # a fifth of the nodes lose all their entries
        Kd = sp.diags(keep.astype(np.float64))
        D = Kd @ D @ Kd
        if rng.random() < 0.7:
            D = D + sp.diags((rng.random(m) < 0.7).astype(np.float64))    # a diagonal entry on most rows
        D = sp.csr_matrix(D); D.eliminate_zeros()
        if D.nnz == 0:
            D = sp.csr_matrix(sp.eye(m))
        return sizes, D, "raw"
    if kind == 0:      # many small, dense clusters (dense-block territory)
        sizes = [int(x) for x in rng.integers(20, 190, rng.integers(3, 30))]
        m = sum(sizes); start = np.concatenate([[0], np.cumsum(sizes)])
        fill = rng.uniform(0.3, 0.95)
        blocks = [sp.csr_matrix((rng.random((s, s)) < fill).astype(np.float64)) for s in sizes]
        D = sp.block_diag(blocks, format="csr")
        cross = sp.random(m, m, density=rng.uniform(0, 3.0) / m, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
        cross.data[:] = 1.0
        D = D + cross
        D = ((D + D.T) > 0).astype(np.float64)
        D.setdiag(0); D.eliminate_zeros()
        D = sp.csr_matrix(D + sp.eye(m))
    else:
        many = 80 if os.environ.get("BIG") else 14          # BIG=1: batches of up to 48 k rows (more windows than CUs x 2)
        sizes = [int(x) for x in rng.integers(1, 600 if kind == 1 else 300, rng.integers(1, many))]
        D = windowed_pattern(rng, sizes, cross_per_row=float(rng.uniform(0, 0.6)), hubs=min(int(rng.integers(0, 8)), sum(sizes)),
                             hub_out=int(rng.integers(1, 8)), dense_tail=bool(rng.integers(0, 2)), big_hubs=0)
        if rng.random() < 0.5:
            D = sp.csr_matrix(((D > 0) * 1.0))      # 0/1 adjacency: row-factor form applies
    flavour = ["raw", "rownorm", "enh"][rng.integers(0, 3)]
    return sizes, D, flavour


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worst = {}
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        sizes, D, flavour = batch_of(rng)
        m = D.shape[0]
        k = int(rng.choice([1, 3, 8, 9, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 63, 64]))
        training = bool(rng.integers(0, 4) > 0)
        s = make_support(D, flavour)
        Y = torch.tensor(rng.standard_normal((m, k)).astype(np.float32), device=dev)
        results = {}
        for path in ("rows", "tiles", "tiles_values", "blocks"):
            ClusterBatch.use_tiles = path in ("tiles", "tiles_values")
            ClusterBatch.use_pattern = path != "tiles_values"
            ClusterBatch.use_blocks = path == "blocks"
            ClusterBatch.tile_max_outside_share = 1.0 if seed % 3 == 0 else 0.08
            b = ClusterBatch.from_support(s, dev, cluster_sizes=sizes)
            if path == "blocks" and b.blocks(k, training) is None:
                continue
            if path.startswith("tiles") and b.tiling(k, training) is None:
                continue
            loss, grad = b.lap(Y, training)
            results[path] = (float(loss), None if grad is None else grad.cpu().numpy(), b.pass_kernel(k, training) if hasattr(b, "pass_kernel") else "")
        ref_loss, ref_grad, _ = results["rows"]
        line = f"seed {seed}: m {m} clusters {len(sizes)} nnz {D.nnz} {flavour} k {k} {'train' if training else 'fwd'}:"
        for path, (l, g, kern) in results.items():
            if path == "rows":
                continue
            e = rel(l, ref_loss)
            if g is not None:
                e = max(e, rel(g, ref_grad))
            worst[path] = max(worst.get(path, 0.0), e)
            line += f" {path} {e:.1e}"
            if not (e < 2e-5):
                print("MISMATCH", line, flush=True)
                sys.exit(1)
        if m <= 1500:
            A, _ = O.csr_from_support(s)
            lo, go, _, _ = O.lap_forward_operator(A, O.affinity_prep_operator(A), Y.cpu().numpy().astype(np.float64))
            e = rel(ref_loss, lo)
            if training:
                e = max(e, rel(ref_grad, go))
            worst["oracle"] = max(worst.get("oracle", 0.0), e)
            line += f" oracle {e:.1e}"
            if not (e < 1e-4):
                print("MISMATCH", line, flush=True)
                sys.exit(1)
        if m <= 500:   # both branches of the loss module and autograd through it, against the dense float64 restatement
            from sparc_amd.SpectralTrainer import SpectralNetLoss
            ClusterBatch.use_tiles = ClusterBatch.use_blocks = ClusterBatch.use_pattern = True
            b = ClusterBatch.from_support(s, dev, cluster_sizes=sizes)
            Wd = O.valid_affinity(O.support_to_dense(s, torch.float64))
            Yd = Y.detach().cpu().double().requires_grad_(True)
            for norm in (True, False):
                want = O.spectral_loss(Wd, Yd, norm)
                (gw,) = torch.autograd.grad(want, Yd)
                Yg = Y.detach().clone().requires_grad_(True)
                got = SpectralNetLoss()(b, Yg, is_normalized=norm)
                got.backward()
                e = max(rel(got.item(), want.item()), rel(Yg.grad.cpu().numpy(), gw.numpy()) if float(gw.abs().max()) > 0 else 0.0)
                worst["dense, normalised" if norm else "dense, plain"] = max(worst.get("dense, normalised" if norm else "dense, plain", 0.0), e)
                line += f" dense{'N' if norm else 'P'} {e:.1e}"
                if not (e < 1e-4):
                    print("MISMATCH", line, flush=True)
                    sys.exit(1)
        print(line, flush=True)
    print("worst:", worst)


if __name__ == "__main__":
    main()
