#!/usr/bin/env python3
"""Token-list builder on a PubMed-size problem (19 717 nodes, 32-d embedding, 500 features, 14 prefix
means): device time, and the oracle (scikit-learn k-NN + gathers, as the reference does it) on a
row sample of the same problem."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.tokens import spectral_token_lists  # noqa: E402

rng = np.random.default_rng(0)
n, k, nf, K = 19717, 32, 500, 14
E = rng.standard_normal((n, k))
Fm = rng.standard_normal((n, nf)).astype(np.float32)
spectral_token_lists(E[:2048], Fm[:2048], K)      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
tok = spectral_token_lists(E, Fm, K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"nodes": n, "embedding_dim": k, "features": nf, "prefix_means": K, "seconds": dt,
                  "nodes_per_s": n / dt}), flush=True)

# the selection step alone: the kernel against torch.topk (sorted) on one chunk of distance rows
from sparc_amd.tokens import row_rank_values  # noqa: E402
dev = torch.device("cuda:0")
Ed = torch.tensor(E, device=dev)
sq = (Ed * Ed).sum(1)
d2 = (sq[:2048, None] + sq[None, :] - 2.0 * (Ed[:2048] @ Ed.t())).clamp_(min=0.0)
ends = [min(2 ** (j + 1), 2 ** 14) for j in range(K)]
for name, fn in (("kernel", lambda: row_rank_values(d2, ends)),
                 ("torch.topk", lambda: torch.topk(d2, 2 ** 14, dim=1, largest=False, sorted=True).values)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(json.dumps({"selection": name, "rows": 2048, "ms": (time.perf_counter() - t0) / 3 * 1e3}), flush=True)
