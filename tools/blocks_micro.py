#!/usr/bin/env python3
"""Kernel-level timing of the dense-block product alone on the bench's Reddit-shape batch (run under rocprofv3
--kernel-trace --stats; SPARC_DBG_BLOCKS = 1 no matrix instructions, 2 no staging, 4 no result stores)."""
import os, sys
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
k = shape["k"]
bl = b.blocks(k, True)
print(f"m={b.m} nnz={b.nnz} fill={bl['fill']:.3f} workgroups={bl['ndesc']} rest={bl['r_nnz']}", flush=True)
X = torch.randn(b.m, 128, device=dev)
Y = torch.randn(b.m, k, device=dev)
for _ in range(30):
    b.block_spmm(X)
    b.block_spmm(X, add=X)
for _ in range(10):
    b.lap(Y, True)
torch.cuda.synchronize()
