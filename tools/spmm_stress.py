#!/usr/bin/env python3
"""Whole-graph CSR SpMM (k = 64) in the HBM-bound regime: an R-MAT graph (a, b, c, d = 0.57, 0.19,
0.19, 0.05; 2^23 nodes, about 500 M stored entries) in natural order, and the cluster-ordered
Amazon2M-shape planted-cluster graph.  One launch of the row-gather kernel per product
(sparc_spmm); prints one JSON line per graph with the algorithmic bytes, the kernel time (HIP
events) and the fraction of the 8 TB/s HBM peak."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth  # noqa: E402
from sparc_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
HBM_PEAK_GBS = 8000.0


def rmat_csr(scale: int, n_edges: int, a=0.57, b=0.19, c=0.19, seed=0, chunk=1 << 26):
    """Symmetrised, deduplicated R-MAT adjacency with unit diagonal as (rowptr int64, colidx int32)."""
    n = 1 << scale
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    keys = []
    done = 0
    while done < n_edges:
        e = min(chunk, n_edges - done)
        u = torch.zeros(e, dtype=torch.int64, device=dev)
        v = torch.zeros(e, dtype=torch.int64, device=dev)
        for _ in range(scale):
            r = torch.rand(e, generator=gen, device=dev)
            ubit = (r >= a + b).to(torch.int64)                      # quadrants c, d: lower half
            vbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)  # quadrants b, d: right half
            u = (u << 1) | ubit
            v = (v << 1) | vbit
        keep = u != v
        u, v = u[keep], v[keep]
        keys.append(torch.unique(torch.cat([u * n + v, v * n + u])))
        done += e
        del u, v, keep
    diag = torch.arange(n, device=dev, dtype=torch.int64)
    keys.append(diag * n + diag)
    keys = torch.unique(torch.cat(keys))
    rows = torch.div(keys, n, rounding_mode="floor")
    cols = (keys - rows * n).to(torch.int32)
    del keys
    counts = torch.bincount(rows, minlength=n)
    del rows
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    return n, rowptr, cols.contiguous()


def run(name, n, rowptr, colidx, width=64, reps=5):
    nnz = int(colidx.numel())
    val = torch.ones(nnz, dtype=torch.float32, device=dev)
    X = torch.randn(n, width, device=dev)
    Z = torch.empty_like(X)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def launch():
        check(lib.sparc_spmm(n, nnz, P(rowptr), P(colidx), P(val), width, P(X), None, P(Z), None, 0, 0, 0, stream), "sparc_spmm")

    launch()
    torch.cuda.synchronize()
    # spot check against float64 on sampled rows
    rows = torch.randint(n, (512,), device=dev)
    worst = 0.0
    for r in rows.tolist():
        cols = colidx[int(rowptr[r]):int(rowptr[r + 1])].long()
        ref = X[cols].double().sum(0)
        worst = max(worst, float((Z[r].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(reps):
        ev0.record()
        launch()
        ev1.record()
        torch.cuda.synchronize()
        times.append(ev0.elapsed_time(ev1))
    ms = sorted(times)[len(times) // 2]
    alg = nnz * 8 + (n + 1) * 8 + 2 * n * width * 4
    out = {"graph": name, "nodes": n, "stored_entries": nnz, "width": width, "kernel": "spmm_rows", "ms": ms,
           "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6, "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
           "max_rel_err_sampled_rows": worst}
    print(json.dumps(out), flush=True)
    return out


def main():
    which = sys.argv[1:] or ["amazon2m", "rmat"]
    if "amazon2m" in which:
        t0 = time.perf_counter()
        g, _ = synth.planted_cluster_graph(2449029, 61859140, 5000, dev, seed=0)
        print(f"[stress] amazon2m-shape graph built in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
        run("amazon2m-shape, cluster-ordered", g.n, g.indptr, g.indices)
        del g
        torch.cuda.empty_cache()
    if "rmat" in which:
        t0 = time.perf_counter()
        n, rowptr, colidx = rmat_csr(23, int(os.environ.get("RMAT_EDGES", 330_000_000)))
        print(f"[stress] R-MAT graph built in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
        run("R-MAT scale 23, natural order", n, rowptr, colidx)


if __name__ == "__main__":
    main()
