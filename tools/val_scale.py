#!/usr/bin/env python3
"""One Amazon2M-shape VALIDATION batch (n / num_clusters_val = 122 k nodes; the reference's dense path
would need a 60 GB m x m matrix for it): preparation time and the forward-only loss."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth  # noqa: E402
from sparc_amd.graph import ClusterBatch  # noqa: E402

dev = torch.device("cuda:0")
g, bounds = synth.planted_cluster_graph(2449029, 61859140, 5000, dev, seed=0)
b = bounds.cpu().numpy()
n_cl = 250
# a small batch first: the first use of every torch and library kernel loads its code object, which is not
# part of preparing a batch
wp = np.arange(b[n_cl], b[n_cl + 5], dtype=np.int32)
wc, wv, ws, _ = g.subblock(wp, mode=2, binarize=True, diag_lambda=0.01)
ClusterBatch.from_device_coo(ws[0], wc, wv, cluster_sizes=np.diff(b[n_cl:n_cl + 6])).warm(47, training=False)
pt = np.arange(b[0], b[n_cl], dtype=np.int32)
torch.cuda.synchronize()
t0 = time.perf_counter()
coords, values, shp, _ = g.subblock(pt, mode=2, binarize=True, diag_lambda=0.01)
cb = ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=np.diff(b[:n_cl + 1])).warm(47, training=False)
torch.cuda.synchronize()
t_prep = time.perf_counter() - t0
Y = torch.randn(cb.m, 47, device=dev)
for _ in range(2):
    loss, _ = cb.lap(Y, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    loss, _ = cb.lap(Y, False)
torch.cuda.synchronize()
t_lap = (time.perf_counter() - t0) / 10
print(json.dumps({"val_batch_nodes": cb.m, "stored_entries": cb.nnz, "prep_seconds": t_prep, "loss_forward_ms": t_lap * 1e3,
                  "loss": float(loss), "finite": bool(torch.isfinite(loss))}), flush=True)
