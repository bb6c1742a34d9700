#!/usr/bin/env python3
"""Kernel-level timing of the Laplacian chain passes on one Amazon2M-shape cluster batch.
Run under rocprofv3 --kernel-trace --stats to read per-kernel durations; SPARC_DBG selects
ablations of the tiled kernel (1: no row loop, 2: no gather steps, 4: no staging)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch

dev = torch.device("cuda:0")
scale = float(os.environ.get("SCALE", "1.0"))
n = int(2449029 * scale)
graph, bounds = synth.planted_cluster_graph(n, int(61859140 * scale), int(5000 * scale), dev, seed=0)
nodes_of, cids = synth.cluster_batches(bounds, int(os.environ.get("BSIZE", "50")), seed=0, return_clusters=True)
pt = nodes_of[0]
coords, values, shp, _ = graph.subblock(pt, mode=2, binarize=False, diag_lambda=0.01)
csz = np.diff(bounds.cpu().numpy())[cids[0]]
b = ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz)
k = int(os.environ.get("K", "47"))
lo, hi, n_out, _ = b._in_window_ranges()
print(f"rows with entries outside their window: {int((n_out > 0).sum())} (more than two: {int((n_out > 2).sum())}) of {b.m}; windows {len(b._windows)}; descriptors {b.tiles(2)[1]} tiled={b.tiles(2)[0] is not None}", flush=True)
Y = torch.randn(b.m, k, device=dev)
reps = int(os.environ.get("REPS", "20"))
for tiles in (True, False):
    ClusterBatch.use_tiles = tiles
    for _ in range(3):
        b.lap(Y, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.lap(Y, True)
    torch.cuda.synchronize()
    alg = b.nnz * 12 + (b.m + 1) * 8 + 3 * b.m * 128 * 4
    print(f"tiles={tiles} m={b.m} nnz={b.nnz} algorithmic bytes per inner pass {alg} lap_fwd(dual) {1e6 * (time.perf_counter() - t0) / reps:.1f} us/call", flush=True)
