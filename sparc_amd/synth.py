"""Synthetic graphs of the reference's dataset shapes (planted clusters), generated on the
device.  There is no network for the real datasets; cluster membership is planted (the
reference obtains it from METIS, src/partition_utils.py:56, which is an input of the path).
"""
from __future__ import annotations

import numpy as np
import torch

from .utils import DeviceGraph

SHAPES = {
    # name: nodes, undirected edges, METIS clusters, clusters per batch, feature dim, k, diag_lambda
    "cora": dict(n=2708, edges=5278, clusters=1, bsize=1, din=32, k=64, diag_lambda=-1),
    "pubmed": dict(n=19717, edges=44338, clusters=15, bsize=5, din=500, k=32, diag_lambda=-1),
    "reddit": dict(n=232965, edges=57_000_000, clusters=1500, bsize=50, din=256, k=64, diag_lambda=-1),
    "amazon2m": dict(n=2449029, edges=61859140, clusters=5000, bsize=50, din=10, k=47, diag_lambda=0.01),
}


def planted_cluster_graph(n: int, n_edges: int, n_clusters: int, device, frac_in_cluster: float = 0.7,
                          skew: float = 2.0, seed: int = 0):
    """Symmetric adjacency A + A^T + I (values 1, unit diagonal, as src/utils.py:238-243 builds it)
    with EXACTLY n_edges distinct undirected edges (when the graph can hold them): `frac_in_cluster`
    of the draws fall inside contiguous node ranges (the planted clusters), the rest are uniform;
    `skew` > 1 concentrates endpoints on the first nodes of a cluster (heavier-tailed degrees).
    Repeated draws are dropped and drawn again (from the third round on uniformly: a cluster that is
    already complete cannot take more), and a surplus is thinned at random.
    Returns (DeviceGraph, bounds int64[n_clusters+1])."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    bounds = torch.div(torch.arange(n_clusters + 1, device=dev, dtype=torch.int64) * n, n_clusters, rounding_mode="floor")
    n_edges = min(int(n_edges), n * (n - 1) // 2)

    def draw(count, frac_in):
        e_in = int(count * frac_in)
        e_out = count - e_in
        c = torch.randint(n_clusters, (e_in,), generator=gen, device=dev)
        lo = bounds[c]
        sz = (bounds[c + 1] - lo).double()
        u = lo + (torch.rand(e_in, generator=gen, device=dev, dtype=torch.float64).pow(skew) * sz).long()
        v = lo + (torch.rand(e_in, generator=gen, device=dev, dtype=torch.float64).pow(skew) * sz).long()
        u = torch.cat([u, torch.randint(n, (e_out,), generator=gen, device=dev)]).clamp_(max=n - 1)
        v = torch.cat([v, torch.randint(n, (e_out,), generator=gen, device=dev)]).clamp_(max=n - 1)
        keep = u != v
        u, v = u[keep], v[keep]
        return torch.minimum(u, v) * n + torch.maximum(u, v)      # undirected key, smaller end first

    und = torch.unique(draw(n_edges, frac_in_cluster))
    for it in range(40):
        deficit = n_edges - und.numel()
        if deficit <= 0:
            break
        extra = draw(int(deficit * 1.2) + 16, frac_in_cluster if it < 2 else 0.0)
        und = torch.unique(torch.cat([und, extra]))
    if und.numel() > n_edges:                                       # thin the surplus at random
        keep = torch.randperm(und.numel(), generator=gen, device=dev)[:n_edges]
        und = und[keep.sort().values]
    u = torch.div(und, n, rounding_mode="floor")
    v = und - u * n
    del und
    diag = torch.arange(n, device=dev, dtype=torch.int64)
    keys = torch.cat([u * n + v, v * n + u, diag * n + diag])
    del u, v
    keys = torch.sort(keys).values
    rows = torch.div(keys, n, rounding_mode="floor")
    cols = (keys - rows * n).to(torch.int32)
    counts = torch.bincount(rows, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(counts, 0)
    data = torch.ones(cols.numel(), dtype=torch.float64, device=dev)
    return DeviceGraph(n, indptr, cols.contiguous(), data), bounds


def cluster_batches(bounds: torch.Tensor, bsize: int, seed: int = 0, return_clusters: bool = False):
    """Node lists of the training batches: the clusters are shuffled (np.random.shuffle in
    src/utils.py:75) and concatenated `bsize` at a time (:76-80)."""
    b = bounds.cpu().numpy()
    n_clusters = len(b) - 1
    order = np.random.RandomState(seed).permutation(n_clusters)
    batches, clusters = [], []
    for st in range(0, n_clusters, bsize):
        ids = order[st:st + bsize]
        clusters.append(ids)
        batches.append(np.concatenate([np.arange(b[c], b[c + 1], dtype=np.int32) for c in ids]))
    return (batches, clusters) if return_clusters else batches


def block_features(node_ids: np.ndarray, bounds: torch.Tensor, din: int, n_blocks: int, seed: int = 0) -> torch.Tensor:
    """N(0,1) features plus a per-block mean shift (block = cluster id mod n_blocks)."""
    b = bounds.cpu().numpy()
    cluster = np.searchsorted(b, node_ids, side="right") - 1
    block = cluster % n_blocks
    means = np.random.RandomState(seed + 1).standard_normal((n_blocks, din)).astype(np.float32)
    rng = np.random.RandomState(seed + 2 + int(node_ids[0]))
    return torch.from_numpy(means[block] + rng.standard_normal((len(node_ids), din)).astype(np.float32))


def shape_batch(workload: str, device, bsize: int | None = None, seed: int = 0):
    """(ClusterBatch, k) with the statistics of one training batch of a named shape, without generating
    the whole graph: twice the batch's clusters at their full size and in-cluster edge count, and as
    many uniform edges as put the full graph's number of cross-cluster entries into the batch (a
    uniform edge of the full graph has both ends in one batch with probability (m/n)^2, of this graph
    with probability 1/4)."""
    from .graph import ClusterBatch
    shape = SHAPES[workload]
    bsize = bsize or shape["bsize"]
    n_clusters = 2 * bsize
    n = int(shape["n"] * n_clusters / shape["clusters"])
    sz = shape["n"] / shape["clusters"]
    # in-cluster edges of the full graph: 70 % of all, but a cluster cannot hold more than its pairs (the
    # skewed draws fill about three quarters of them before the generator turns to uniform edges)
    e_in_full = min(0.7 * shape["edges"], shape["clusters"] * 0.75 * sz * (sz - 1) / 2)
    e_in = e_in_full * n_clusters / shape["clusters"]
    e_cross = 4 * (shape["edges"] - e_in_full) * (bsize / shape["clusters"]) ** 2
    graph, bounds = planted_cluster_graph(n, int(e_in + e_cross), n_clusters, device,
                                          frac_in_cluster=e_in / (e_in + e_cross), seed=seed)
    nodes_of, cluster_ids = cluster_batches(bounds, bsize, seed=seed, return_clusters=True)
    lam = shape["diag_lambda"]
    coords, values, shp, _ = graph.subblock(nodes_of[0], mode=(0 if lam == -1 else 2), binarize=False, diag_lambda=lam)
    csz = np.diff(bounds.cpu().numpy())[cluster_ids[0]]
    return ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz), shape["k"]


def rmat_graph(n: int, n_edges: int, device, a: float = 0.57, b: float = 0.19, c: float = 0.19, seed: int = 0,
               chunk: int = 1 << 26):
    """R-MAT graph (quadrant probabilities a, b, c, 1 - a - b - c) on n nodes with EXACTLY n_edges distinct
    undirected edges, as symmetric CSR pattern with unit diagonal: (rowptr int64[n+1], colidx int32),
    columns sorted inside a row.  n need not be a power of two: ids are drawn on the next power of two
    and draws with an end >= n are dropped.  Repeated edges are drawn again."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    scale = max(1, int(np.ceil(np.log2(n))))
    n_edges = min(int(n_edges), n * (n - 1) // 2)

    def draw(count):
        out = []
        for lo in range(0, count, chunk):
            e = min(chunk, count - lo)
            u = torch.zeros(e, dtype=torch.int64, device=dev)
            v = torch.zeros(e, dtype=torch.int64, device=dev)
            for _ in range(scale):
                r = torch.rand(e, generator=gen, device=dev)
                u = (u << 1) | (r >= a + b).to(torch.int64)                                   # quadrants c, d: lower half
                v = (v << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)  # quadrants b, d: right half
            keep = (u != v) & (u < n) & (v < n)
            u, v = u[keep], v[keep]
            out.append(torch.unique(torch.minimum(u, v) * n + torch.maximum(u, v)))
        return torch.unique(torch.cat(out)) if len(out) > 1 else out[0]

    und = draw(n_edges)
    for _ in range(60):
        deficit = n_edges - und.numel()
        if deficit <= 0:
            break
        und = torch.unique(torch.cat([und, draw(int(deficit * 1.5) + 1024)]))
    if und.numel() > n_edges:
        keep = torch.randperm(und.numel(), generator=gen, device=dev)[:n_edges]
        und = und[keep.sort().values]
    u = torch.div(und, n, rounding_mode="floor")
    v = und - u * n
    del und
    return csr_pattern(n, u, v)


def csr_pattern(n: int, u: torch.Tensor, v: torch.Tensor):
    """Symmetric CSR pattern with unit diagonal of the undirected edges (u, v): (rowptr, colidx)."""
    dev = u.device
    diag = torch.arange(n, device=dev, dtype=torch.int64)
    keys = torch.sort(torch.cat([u * n + v, v * n + u, diag * n + diag])).values
    rows = torch.div(keys, n, rounding_mode="floor")
    cols = (keys - rows * n).to(torch.int32)
    del keys
    counts = torch.bincount(rows, minlength=n)
    del rows
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, cols.contiguous()


def degree_order(rowptr: torch.Tensor, colidx: torch.Tensor):
    """Relabel the nodes by decreasing degree (ties by old id): the rows every other row gathers most often
    become one contiguous, cache-resident block.  Returns (rowptr, colidx, perm) of P A P^T with
    perm[new] = old, columns sorted inside a row."""
    n = rowptr.numel() - 1
    deg = rowptr[1:] - rowptr[:-1]
    perm = torch.sort(deg, descending=True, stable=True).indices          # new -> old
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=perm.device)
    rows = torch.repeat_interleave(torch.arange(n, device=perm.device), deg)
    keys = torch.sort(inv[rows] * n + inv[colidx.long()]).values
    del rows
    nrows = torch.div(keys, n, rounding_mode="floor")
    ncols = (keys - nrows * n).to(torch.int32)
    del keys
    nptr = torch.zeros(n + 1, dtype=torch.int64, device=perm.device)
    nptr[1:] = torch.cumsum(torch.bincount(nrows, minlength=n), 0)
    return nptr, ncols.contiguous(), perm
