#!/usr/bin/env python3
"""Whole-graph CSR SpMM Z = A X (k = 64, values 1) in the HBM-bound regime: the R-MAT stress graph
(a, b, c, d = 0.57, 0.19, 0.19, 0.05; 10 M nodes, 500 M stored off-diagonal entries + unit diagonal) in
natural and in degree order, and the cluster-ordered Amazon2M-shape planted-cluster graph with the row
kernel and with the windowed kernel.  One launch per product (sparc_spmm); prints one JSON line per case
with the algorithmic bytes, the kernel time (HIP events on the launch stream) and the fraction of the
8 TB/s HBM peak.  Cases: python tools/spmm_stress.py [rmat] [rmat-blocked[=N]] [rmat-degree] [amazon2m] [amazon2m-windowed]
rmat-split[=LIMIT] / rmat-degree-split[=LIMIT]: the same products with every row above LIMIT entries cut into
chunks (sparc_spmm_split).  (RMAT_NODES / RMAT_EDGES shrink the R-MAT graph)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth  # noqa: E402
from sparc_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
HBM_PEAK_GBS = 8000.0
P = lambda t: ctypes.c_void_p(t.data_ptr())


def run(name, n, rowptr, colidx, width=64, reps=5, tiling=None, kernel="spmm_rows", split=0):
    nnz = int(colidx.numel())
    val = torch.ones(nnz, dtype=torch.float32, device=dev)
    X = torch.randn(n, width, device=dev)
    Z = torch.empty_like(X)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = None
    if split:
        from sparc_amd.graph import RowSplit, spmm_split
        plan = RowSplit(rowptr, limit=split)
        deg = rowptr[1:] - rowptr[:-1]
        kernel = (f"spmm_rows + spmm_chunks / spmm_merge: {plan.nlong} rows above {split} entries (longest {int(deg.max())}) "
                  f"in {plan.nchunks} chunks holding {int(deg[plan.long_row.long()].sum())} entries")

    def launch():
        if plan is not None:
            spmm_split(plan, n, rowptr, colidx, val, X, out=Z)
            return
        check(lib.sparc_spmm(n, nnz, P(rowptr), P(colidx), P(val), width, P(X), None, P(Z),
                             ctypes.byref(tiling) if tiling is not None else None, stream), "sparc_spmm")

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
    out = {"graph": name, "nodes": n, "stored_entries": nnz, "width": width, "kernel": kernel, "ms": ms,
           "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6, "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
           "gathered_row_GBps": nnz * width * 4 / ms / 1e6, "GFLOPs": 2.0 * nnz * width / ms / 1e6,
           "max_rel_err_sampled_rows": worst}
    if "decreasing degree" in name:   # how much of the gather traffic the hottest operand rows carry (they are the first rows)
        for top in (16384, 131072, 1048576):
            out[f"share_of_entries_pointing_at_the_{top}_rows_of_highest_degree"] = float((colidx < top).sum()) / nnz
    print(json.dumps(out), flush=True)
    return out


def column_blocks(n, rowptr, colidx, block_rows):
    """The CSR pattern split by source-node range: block p holds the entries whose column lies in
    [p * block_rows, (p + 1) * block_rows), as a CSR over all n rows.  Returns [(rowptr_p, colidx_p)]."""
    row = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), rowptr[1:] - rowptr[:-1])
    blk = torch.div(colidx, block_rows, rounding_mode="floor").to(torch.int16)
    nb = (n + block_rows - 1) // block_rows
    order = torch.argsort(blk, stable=True)       # rows stay ascending inside a block, columns inside a row
    counts = torch.bincount(blk.long(), minlength=nb).cpu().tolist()
    del blk
    row, col = row[order], colidx[order]
    del order
    out, start = [], 0
    for p in range(nb):
        r = row[start:start + counts[p]]
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(r.long(), minlength=n), 0)
        out.append((rp, col[start:start + counts[p]].contiguous()))
        start += counts[p]
    return out


def run_blocked(name, n, rowptr, colidx, block_rows, width=64, reps=5):
    """Z = A X as one pass per block of source nodes, Z accumulated in place (X0 = Z): the operand rows a pass
    gathers (block_rows * width * 4 bytes) fit the Infinity Cache, the result is read and written once per pass."""
    nnz = int(colidx.numel())
    blocks = column_blocks(n, rowptr, colidx, block_rows)
    val = torch.ones(max(int(c.numel()) for _, c in blocks), dtype=torch.float32, device=dev)
    X = torch.randn(n, width, device=dev)
    Z = torch.empty_like(X)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        for p, (rp, ci) in enumerate(blocks):
            check(lib.sparc_spmm(n, int(ci.numel()), P(rp), P(ci), P(val), width, P(X), P(Z) if p else None, P(Z), None, stream),
                  "sparc_spmm")

    launch()
    torch.cuda.synchronize()
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
    nb = len(blocks)
    moved = nnz * 8 + nb * (n + 1) * 8 + (2 * nb - 1) * n * width * 4 + n * width * 4
    out = {"graph": name, "nodes": n, "stored_entries": nnz, "width": width,
           "kernel": f"spmm_rows, {nb} passes over blocks of {block_rows} source nodes, result accumulated in place",
           "ms": ms, "algorithmic_bytes": alg, "achieved_GBps": alg / ms / 1e6, "frac_of_hbm_peak": alg / ms / 1e6 / HBM_PEAK_GBS,
           "streamed_bytes_of_the_passes": moved, "entries_in_largest_block": max(int(c.numel()) for _, c in blocks),
           "GFLOPs": 2.0 * nnz * width / ms / 1e6, "max_rel_err_sampled_rows": worst}
    print(json.dumps(out), flush=True)
    return out


def main():
    which = sys.argv[1:] or ["rmat", "rmat-degree", "amazon2m", "amazon2m-windowed"]
    if any(w.startswith("rmat") for w in which):
        n = int(os.environ.get("RMAT_NODES", 10_000_000))
        e = int(os.environ.get("RMAT_EDGES", 250_000_000))
        t0 = time.perf_counter()
        rowptr, colidx = synth.rmat_graph(n, e, dev)
        torch.cuda.synchronize()
        print(f"[stress] R-MAT graph ({n} nodes, {e} undirected edges) built in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
        tag = f"R-MAT {n / 1e6:g}M / {2 * e / 1e6:g}M"
        if "rmat" in which:
            run(f"{tag}, natural order", n, rowptr, colidx)
        for w in which:
            if w.startswith("rmat-split"):        # rmat-split or rmat-split=<row limit>
                run(f"{tag}, natural order", n, rowptr, colidx, split=int(w.split("=")[1]) if "=" in w else 2048)
        for w in which:
            if w.startswith("rmat-blocked"):      # rmat-blocked or rmat-blocked=<source nodes per block>
                br = int(w.split("=")[1]) if "=" in w else 1 << 19
                run_blocked(f"{tag}, natural order", n, rowptr, colidx, br)
                torch.cuda.empty_cache()
        if any(w.startswith("rmat-degree") for w in which):
            t0 = time.perf_counter()
            rp2, ci2, _ = synth.degree_order(rowptr, colidx)
            torch.cuda.synchronize()
            print(f"[stress] degree ordering in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
            del rowptr, colidx
            if "rmat-degree" in which:
                run(f"{tag}, nodes relabelled by decreasing degree", n, rp2, ci2)
            for w in which:
                if w.startswith("rmat-degree-split"):
                    run(f"{tag}, nodes relabelled by decreasing degree", n, rp2, ci2, split=int(w.split("=")[1]) if "=" in w else 2048)
            del rp2, ci2
        torch.cuda.empty_cache()
    if "amazon2m" in which or "amazon2m-windowed" in which:
        t0 = time.perf_counter()
        g, bounds = synth.planted_cluster_graph(2449029, 61859140, 5000, dev, seed=0)
        print(f"[stress] amazon2m-shape graph built in {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
        if "amazon2m" in which:
            run("amazon2m-shape, cluster-ordered", g.n, g.indptr, g.indices)
        if "amazon2m-windowed" in which:
            from sparc_amd.graph import ClusterBatch, plan_windows
            b = ClusterBatch()                       # the whole graph as one batch, its clusters as windows
            b.m, b.nnz, b.device = g.n, int(g.indices.numel()), dev
            b.rowptr, b.colidx = g.indptr, g.indices
            b.val = torch.ones(b.nnz, dtype=torch.float32, device=dev)
            b.valT = None
            b._windows = plan_windows(g.n, np.diff(bounds.cpu().numpy()))
            b._tiles, b._measured, b._window_ranges = {}, {}, None
            b.tile_max_outside_share = 1.0           # the windowed kernel whatever the share of cross-cluster entries
            t0 = time.perf_counter()
            t = b._plan(4, 64, 1, False, 1024, 256)
            torch.cuda.synchronize()
            print(f"[stress] tiling in {time.perf_counter() - t0:.1f}s: {t.ndesc if t else None} workgroups", file=sys.stderr, flush=True)
            if t is not None:
                run("amazon2m-shape, cluster-ordered", g.n, g.indptr, g.indices, tiling=t, kernel="spmm_win<vec 4, single>")


if __name__ == "__main__":
    main()
