#!/usr/bin/env python3
"""How the stored entries of the bench's first training batch split into in-cluster, in-window and the rest
(WORKLOAD=amazon2m|reddit|pubmed|cora)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch
dev = torch.device("cuda:0")
w = os.environ.get("WORKLOAD", "reddit")
shape = synth.SHAPES[w]
graph, bounds = synth.planted_cluster_graph(shape["n"], shape["edges"], shape["clusters"], dev, seed=0)
nodes_of, cluster_ids = synth.cluster_batches(bounds, shape["bsize"], seed=0, return_clusters=True)
lam = shape["diag_lambda"]
coords, values, shp, _ = graph.subblock(nodes_of[0], mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
csz = np.diff(bounds.cpu().numpy())[cluster_ids[0]]
b = ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz)
start = np.concatenate([[0], np.cumsum(csz)])
cl_of = np.repeat(np.arange(len(csz)), csz)
rp = b.rowptr.cpu().numpy(); ci = b.colidx.cpu().numpy()
rows = np.repeat(np.arange(b.m), np.diff(rp))
same = cl_of[rows] == cl_of[ci]
p0, lo, hi, p1 = b._in_window_ranges()
out = {"workload": w, "m": b.m, "nnz": b.nnz, "clusters": len(csz), "cluster_rows_mean": float(csz.mean()),
       "in_cluster_share": float(same.mean()), "in_window_share": float((hi - lo).sum() / b.nnz),
       "cluster_block_fill": float(same.sum() / (csz.astype(np.float64) ** 2).sum()),
       "entries_per_row": b.nnz / b.m, "symmetric_values": b.valT is None}
print(json.dumps(out))
