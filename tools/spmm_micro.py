#!/usr/bin/env python3
"""Kernel-level timing of the Laplacian chain passes on one cluster batch of a named shape
(WORKLOAD=amazon2m|reddit|pubmed, BSIZE clusters per batch, K spectral dimension).  Run under
rocprofv3 --kernel-trace --stats to read per-kernel durations.  VEC / THREADS / WGS select the
windowed kernel's geometry (floats per lane, workgroup size, workgroups to aim for)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch

dev = torch.device("cuda:0")
workload = os.environ.get("WORKLOAD", "amazon2m")
bsize = int(os.environ.get("BSIZE", "0")) or None
if os.environ.get("LAM"):      # another support flavour of the same graph (-1: raw symmetric values)
    synth.SHAPES[workload] = dict(synth.SHAPES[workload], diag_lambda=float(os.environ["LAM"]))
b, k = synth.shape_batch(workload, dev, bsize=bsize)
k = int(os.environ.get("K", k))
ClusterBatch.tile_vec = int(os.environ.get("VEC", "0")) or None
ClusterBatch.tile_threads = int(os.environ.get("THREADS", "0")) or None
ClusterBatch.tile_target_wgs = int(os.environ.get("WGS", "0")) or None
ClusterBatch.tile_parity_order = os.environ.get("PARITY", "0") == "1"
p0, lo, hi, p1 = b._in_window_ranges()
n_out = (lo - p0) + (p1 - hi)
t = b.tiling(k, True)
print(f"{workload}: m={b.m} nnz={b.nnz} k={k} windows={len(b._windows)} entries outside their window: "
      f"{int(n_out.sum())} ({n_out.sum() / b.nnz:.3f}); tiling: "
      + (f"vec={t.vec} kp={t.kp} threads={t.threads} descriptors={t.ndesc} rows={t.rows}" if t else "none"), flush=True)
Y = torch.randn(b.m, k, device=dev)
reps = int(os.environ.get("REPS", "20"))
if b.blocks(k, True) is not None:
    bl = b.blocks(k, True)
    print("dense blocks:", "none" if bl is None else f"fill {bl['fill']:.3f}, {bl['ndesc']} workgroups, {bl['r_nnz']} entries left to the row kernel", flush=True)
    for _ in range(3):
        b.lap(Y, True)
        b.lap(Y, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.lap(Y, True)
    torch.cuda.synchronize()
    print(f"blocks lap_fwd(dual) {1e6 * (time.perf_counter() - t0) / reps:.1f} us/call", flush=True)
ClusterBatch.use_blocks = False
for tiles in (True, False):
    ClusterBatch.use_tiles = tiles
    for _ in range(3):
        b.lap(Y, True)
        b.lap(Y, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.lap(Y, True)
    torch.cuda.synchronize()
    alg = b.nnz * (12 if b.valT is not None else 8) + (b.m + 1) * 8 + 3 * b.m * 2 * k * 4
    print(f"tiles={tiles} algorithmic bytes per inner pass {alg} lap_fwd(dual) {1e6 * (time.perf_counter() - t0) / reps:.1f} us/call", flush=True)
