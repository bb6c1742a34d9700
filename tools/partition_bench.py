#!/usr/bin/env python3
"""Device partitioner on the Amazon2M-shape planted-cluster graph: time, balance and edge cut
against the planted clusters (the cut METIS would be asked to find)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth  # noqa: E402
from sparc_amd.partition import edge_cut, edge_list_of_csr, kway_partition  # noqa: E402

dev = torch.device("cuda:0")
scale = float(os.environ.get("SCALE", "1.0"))
n, e, k = int(2449029 * scale), int(61859140 * scale), max(int(5000 * scale), 2)
g, bounds = synth.planted_cluster_graph(n, e, k, dev, seed=0)
src, dst = edge_list_of_csr(g.indptr, g.indices)
planted = torch.searchsorted(bounds, torch.arange(n, device=dev), right=True) - 1
torch.cuda.synchronize()
t0 = time.perf_counter()
part = kway_partition(n, src, dst, k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sizes = torch.bincount(part, minlength=k)
print(json.dumps({"nodes": n, "undirected_edges": int(src.numel()) // 2, "parts": k, "seconds": dt,
                  "min_part": int(sizes.min()), "max_part": int(sizes.max()), "mean_part": n / k,
                  "edge_cut": edge_cut(part, src, dst), "planted_cut": edge_cut(planted, src, dst)}), flush=True)
