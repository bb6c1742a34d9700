#!/usr/bin/env python3
"""Randomised checks of the remaining integer / bit-exact kernels on the GPU, each against an independent formulation:
  csr     sparc_csr_build (ClusterBatch.from_support) against the oracle's SciPy CSR, bit for bit;
  rank    sparc_row_rank_values against numpy's sort, bit for bit;
  sweep   the partitioner's label-score kernel sweep against the torch formulation of the same sweep, exactly;
  gram    sparc_gram (strided operands, every k) against float64 numpy, and sparc_gram_static against it bit for bit;
  spmm    the plain products of ClusterBatch.spmm (widths, transpose, addend, windowed or not) against SciPy.
A script (it uses the oracle as a checker), not collected by pytest: python tests/fuzz_misc.py [seeds] [first seed]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sparc_oracle as O  # noqa: E402
from sparc_amd import partition as P  # noqa: E402
from sparc_amd.graph import ClusterBatch  # noqa: E402
from sparc_amd.tokens import row_rank_values  # noqa: E402

dev = torch.device("cuda:0")


def fail(what, seed, detail=""):
    print(f"MISMATCH {what} seed {seed} {detail}", flush=True)
    sys.exit(1)


def check_csr(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.integers(1, 2500))
    nnz = int(rng.integers(0, 12 * m + 1))
    rows = rng.integers(0, m, nnz).astype(np.int32)
    cols = rng.integers(0, m, nnz).astype(np.int32)
    vals = rng.choice([1.0, 2.0, 0.5, 1e-3, 0.0, 1 / 3, -1.25], nnz)
    kind = rng.integers(0, 3)
    if kind == 0 and nnz:            # symmetric values
        rows, cols, vals = np.concatenate([rows, cols]), np.concatenate([cols, rows]), np.concatenate([vals, vals])
    if kind == 1 and nnz:            # duplicates
        dup = np.unique(rng.integers(0, len(rows), len(rows) // 3 + 1))   # at most two copies: the order of longer sums is not defined
        rows, cols, vals = np.concatenate([rows, rows[dup]]), np.concatenate([cols, cols[dup]]), np.concatenate([vals, vals[dup]])
    coords = np.stack([rows, cols], axis=1).astype(np.int32).reshape(-1, 2)
    s = (coords, vals.astype(np.float64), (m, m))
    b = ClusterBatch.from_support(s, dev)
    # expected: the stored keys united with their mirror images, sorted by (row, col); val = the (float32-cast,
    # duplicate-merged) value or 0, valT = the value of the mirrored key or 0
    # (duplicates are summed in float32 in input order, as torch's sparse-to-dense does for the reference,
    # src/SpectralTrainer.py:275-281; the oracle's float64 sum agrees up to two copies)
    k_in = coords[:, 0].astype(np.int64) * m + coords[:, 1]
    order = np.argsort(k_in, kind="stable")
    uniq, start = np.unique(k_in[order], return_index=True)
    v32 = vals.astype(np.float32)[order]
    v0 = v32[start].copy() if len(k_in) else np.zeros(0, np.float32)
    ends = np.append(start[1:], len(v32)) if len(k_in) else start
    for u in np.nonzero(ends - start > 1)[0]:          # sequential float32 sums of the runs of duplicates
        acc = np.float32(0.0)
        for x in v32[start[u]:ends[u]]:
            acc = np.float32(acc + x)
        v0[u] = acc
    r0, c0 = (uniq // m).astype(np.int32), (uniq % m).astype(np.int32)
    key = r0.astype(np.int64) * m + c0
    keyT = c0.astype(np.int64) * m + r0
    allk = np.union1d(key, keyT)
    val = np.zeros(len(allk), np.float32); valT = np.zeros(len(allk), np.float32)
    val[np.searchsorted(allk, key)] = v0.astype(np.float32)
    valT[np.searchsorted(allk, keyT)] = v0.astype(np.float32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(allk // m, minlength=m))]).astype(np.int64)
    ok = b.nnz == len(allk) and np.array_equal(b.rowptr.cpu().numpy(), rowptr) \
        and np.array_equal(b.colidx.cpu().numpy()[:b.nnz], (allk % m).astype(np.int32)) \
        and np.array_equal(b.val.cpu().numpy()[:b.nnz], val)
    if ok:
        ok = np.array_equal(b.valT.cpu().numpy()[:b.nnz], valT) if b.valT is not None else np.array_equal(val, valT)
    if not ok:
        d = f"m {m} nnz {nnz} kind {kind}: nnz {b.nnz} vs {len(allk)}, valT {b.valT is not None}"
        if b.nnz == len(allk):
            bv = b.val.cpu().numpy()[:b.nnz]
            d += f" rowptr {np.array_equal(b.rowptr.cpu().numpy(), rowptr)} col {np.array_equal(b.colidx.cpu().numpy()[:b.nnz], (allk % m).astype(np.int32))}"
            dv = np.nonzero(bv != val)[0]
            d += f" val differing {len(dv)}"
            for i in dv[:4]:
                cnt = int((k_in == allk[i]).sum())
                d += f" [ours {bv[i]!r} expected {val[i]!r} copies {cnt} inputs {vals[k_in == allk[i]]}]"
            if b.valT is not None:
                d += f" valT differing {int((b.valT.cpu().numpy()[:b.nnz] != valT).sum())}"
        fail("csr", seed, d)


def check_rank(seed):
    rng = np.random.default_rng(seed)
    rows, n = int(rng.integers(1, 40)), int(rng.integers(1, 40000))
    kind = rng.integers(0, 4)
    if kind == 0:
        D = rng.standard_normal((rows, n)) ** 2
    elif kind == 1:
        D = rng.integers(0, int(rng.integers(1, 50)), (rows, n)).astype(np.float64) * 0.125
    elif kind == 2:
        D = rng.standard_normal((rows, n)) * 10.0 ** rng.integers(-300, 300, (rows, n))
    else:
        D = np.abs(rng.standard_normal((rows, 1))) * np.ones((rows, n))       # one value per row
        D[:, : n // 2] *= 2
    ranks = sorted(set(int(r) for r in np.minimum(rng.integers(1, 40001, int(rng.integers(1, 14))), n)))
    got = row_rank_values(torch.tensor(D, device=dev), ranks).cpu().numpy()
    srt = np.sort(D, axis=1)
    want = np.stack([srt[:, r - 1] for r in ranks], axis=1)
    if not np.array_equal(got, want):
        fail("rank", seed, f"rows {rows} n {n} kind {kind} ranks {ranks}")


def check_sweep(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 5000))
    deg = int(rng.integers(1, 40))
    u = rng.integers(0, n, n * deg // 2 + 1); v = rng.integers(0, n, n * deg // 2 + 1)
    if rng.random() < 0.5:           # a few hubs (the LDS table path)
        hubs = rng.choice(n, min(4, n), replace=False)
        hu = np.repeat(hubs, min(n, 700))
        u = np.concatenate([u, hu]); v = np.concatenate([v, rng.integers(0, n, len(hu))])
    keep = u != v
    key = np.unique(np.concatenate([u[keep] * n + v[keep], v[keep] * n + u[keep]]))
    if len(key) == 0:
        return
    src, dst = torch.from_numpy(key // n).to(dev), torch.from_numpy(key % n).to(dev)
    n_labels = int(rng.integers(1, max(2, n)))
    w = torch.from_numpy(rng.integers(1, 2 ** int(rng.integers(1, 34)), src.numel())).to(dev)
    nw = torch.from_numpy(rng.integers(1, 4, n)).to(dev)
    labels = torch.from_numpy(rng.integers(0, n_labels, n)).to(dev)
    parity = P._parity(n, int(rng.integers(0, 5)), dev)
    mean = max(1, int(nw.sum()) // n_labels)
    mode = ["move", "evict", "pull"][rng.integers(0, 3)]
    kw = {"move": {}, "evict": {"evict_to": mean}, "pull": {"pull_to": mean}}[mode]
    rp = P._rowptr(src, n)
    out_k, moved_k = P._sweep(labels, src, dst, w, nw, mean + int(rng.integers(0, 9)), n_labels, parity, rowptr=rp, **kw)
    out_t, moved_t = P._sweep(labels, src, dst, w, nw, mean + 0, n_labels, parity, rowptr=None, **kw) if False else (None, None)
    return (labels, src, dst, w, nw, n_labels, parity, kw, rp, out_k, moved_k, mean)


def check_gram(seed):
    import ctypes
    from sparc_amd._lib import lib
    from sparc_amd.graph import _ptr
    from sparc_amd.ops import gram
    rng = np.random.default_rng(seed)
    m, k = int(rng.integers(1, 30000)), int(rng.integers(1, 65))
    pad = int(rng.integers(0, 9))
    A = torch.tensor(rng.standard_normal((m, k + pad)), dtype=torch.float32, device=dev)[:, :k]      # strided rows
    B = A if rng.random() < 0.5 else torch.tensor(rng.standard_normal((m, k)), dtype=torch.float32, device=dev)
    G = gram(A, B)
    want = A.double().cpu().numpy().T @ B.double().cpu().numpy()
    e = float(np.abs(G.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-30))
    if not (e < 1e-12):
        fail("gram", seed, f"m {m} k {k} error {e:.2e}")
    G2 = torch.empty_like(G)
    with torch.cuda.device(dev):
        rc = lib.sparc_gram_static(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G2), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0 or not torch.equal(G, G2):
        fail("gram_static", seed, f"m {m} k {k} rc {rc}")


def check_spmm(seed):
    rng = np.random.default_rng(seed)
    ncl = int(rng.integers(1, 9))
    sizes = [int(x) for x in rng.integers(1, 400, ncl)]
    m = sum(sizes)
    lab = np.repeat(np.arange(ncl), sizes)
    dens = rng.uniform(0.01, 0.2)
    Pm = np.where(lab[:, None] == lab[None, :], dens, rng.uniform(0, 0.003))
    A = sp.csr_matrix((rng.random((m, m)) < Pm) * rng.uniform(0.5, 2.0, (m, m)))       # unsymmetric values and pattern
    if rng.random() < 0.5:
        A = sp.csr_matrix(A + A.T)
    A = sp.csr_matrix(A + sp.eye(m))
    s = O.as_support_tuple(A)
    b = ClusterBatch.from_support(s, dev, cluster_sizes=sizes)
    A32, At32 = O.csr_from_support(s)
    for width in (16, 32, 64, 128):
        X = rng.standard_normal((m, width)).astype(np.float32)
        Xd = torch.tensor(X, device=dev)
        for transpose in (False, True):
            M = At32 if transpose else A32
            ref = M @ X.astype(np.float64)
            variants = [False] + ([True] if width in (32, 64) else [])
            for tiled in variants:
                try:
                    Z = b.spmm(Xd, transpose=transpose, tiled=tiled)
                except ValueError:
                    continue            # no tiling for this batch
                e = float(np.abs(Z.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30))
                Z2 = b.spmm(Xd, add=Xd, transpose=transpose, tiled=tiled)
                e = max(e, float(np.abs(Z2.cpu().numpy() - (ref + X)).max() / max(np.abs(ref + X).max(), 1e-30)))
                if not (e < 2e-6):
                    fail("spmm", seed, f"m {m} width {width} transpose {transpose} tiled {tiled}: {e:.2e}")


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for seed in range(first, first + n_seeds):
        check_csr(seed)
        check_rank(seed)
        check_gram(seed)
        check_spmm(seed)
        # the sweep twice with the same cap: kernel against torch
        rng = np.random.default_rng(seed)
        got = check_sweep(seed)
        if got is not None:
            labels, src, dst, w, nw, n_labels, parity, kw, rp, _, _, mean = got
            cap = mean + seed % 7
            out_k, moved_k = P._sweep(labels, src, dst, w, nw, cap, n_labels, parity, rowptr=rp, **kw)
            out_t, moved_t = P._sweep(labels, src, dst, w, nw, cap, n_labels, parity, rowptr=None, **kw)
            if moved_k != moved_t or not torch.equal(out_k, out_t):
                fail("sweep", seed, f"n {labels.numel()} labels {n_labels} moved {moved_k} vs {moved_t}")
        if seed % 20 == 0:
            print(f"seed {seed} ok", flush=True)
    print("all exact")


if __name__ == "__main__":
    main()
