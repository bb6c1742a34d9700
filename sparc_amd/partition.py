"""Balanced k-way graph partitioning on the device, standing in for the METIS call of the
reference (metis.part_graph in src/partition_utils.py:56), and partition_graph itself
(src/partition_utils.py:30-81) with the same arguments and return value.

METIS is a native dependency the reference does not ship; its cut cannot be reproduced bit for
bit, and nothing downstream depends on the particular cut: the training batches are unions of
parts (src/utils.py:76-80).  What is kept is the contract: exactly num_clusters parts, every
node in one part, parts of near-equal size, few edges between parts.

Method: multilevel size-constrained label propagation.  Coarsening sweeps merge every node into
the neighbouring cluster it is connected to most strongly (subject to a size cap) and contract;
the coarsest graph is packed into k parts; the way back up refines with the same sweep restricted
to moves that reduce the cut and respect the part capacity.  On the GPU the per-node label totals of a
sweep come from one wavefront-per-node kernel over the CSR adjacency (sparc_label_scores); the choice of the
moves that fit the capacities is a few sorts over the candidate nodes.  The same sweep is also written with
sorts and segmented reductions over the edge list for any torch device (integer weights: results are
deterministic and the two agree).
"""
from __future__ import annotations

import heapq
from typing import List, Optional, Tuple

import numpy as np
import scipy.sparse as sp
import torch


# --------------------------------------------------------------------------- one propagation sweep
def _label_scores_kernel(labels, rowptr, dst, w, n_labels, label_ok, exclude_own):
    """(best, best_lab, cur) of every node from the wavefront-per-node kernel of libsparc_hip.so (sparc_label_scores)."""
    import ctypes
    from ._lib import check, lib
    n = labels.numel()
    dev = labels.device
    best = torch.empty(n, dtype=torch.int64, device=dev)
    best_lab = torch.empty(n, dtype=torch.int64, device=dev)
    cur = torch.empty(n, dtype=torch.int64, device=dev)
    ok = label_ok.to(torch.uint8).contiguous() if label_ok is not None else None
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    with torch.cuda.device(dev):
        check(lib.sparc_label_scores(n, ptr(rowptr), ptr(dst), ptr(w), ptr(labels), int(n_labels), ptr(ok), int(exclude_own),
                                     ptr(best), ptr(best_lab), ptr(cur),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "sparc_label_scores")
    return best, best_lab, cur


def _sweep(labels, src, dst, w, nw, cap, n_labels, parity, evict_to=None, pull_to=None, rowptr=None):
    """Every node selected by `parity` moves to the label its neighbours carry with the largest
    total weight if that beats its current label's weight and the target has room.
    With evict_to (the size every label must come down to) the sweep rebalances instead: only
    nodes of oversized labels move, to their best-connected label that has room, the least
    costly first, whatever the sign of the gain.  With pull_to (the size every label should reach)
    labels below it draw in their best-connected neighbours from labels above it.
    Returns the new labels and the number of nodes moved."""
    n = labels.numel()
    dev = labels.device
    size = torch.zeros(n_labels, dtype=torch.int64, device=dev).scatter_add_(0, labels, nw)
    if rowptr is not None and labels.is_cuda:
        # on the GPU the per-node label totals come from one kernel pass over the CSR adjacency (rowptr: the
        # edge list is sorted by source); the torch formulation below is the same computation on any device
        label_ok = None
        if evict_to is not None:
            label_ok = size < evict_to
        if pull_to is not None:
            label_ok = size < pull_to
        best, best_lab, cur = _label_scores_kernel(labels.contiguous(), rowptr, dst.contiguous(), w.contiguous(), n_labels,
                                                   label_ok, evict_to is not None or pull_to is not None)
        if pull_to is not None:   # only nodes of labels that can spare one are drawn
            spare = size[labels] > pull_to
            best = torch.where(spare, best, torch.zeros_like(best))
            best_lab = torch.where(spare, best_lab, torch.full_like(best_lab, n_labels))
        return _sweep_moves(labels, nw, cap, n_labels, parity, evict_to, pull_to, size, best, best_lab, cur)
    key = src * n_labels + labels[dst]
    key, order = torch.sort(key)
    uniq, seg = torch.unique_consecutive(key, return_inverse=True)
    score = torch.zeros(uniq.numel(), dtype=torch.int64, device=dev).scatter_add_(0, seg, w[order])
    u_of = torch.div(uniq, n_labels, rounding_mode="floor")
    lab_of = uniq - u_of * n_labels
    cur_key = torch.arange(n, device=dev, dtype=torch.int64) * n_labels + labels
    pos = torch.searchsorted(uniq, cur_key).clamp_(max=max(uniq.numel() - 1, 0))
    cur = torch.where(uniq[pos] == cur_key, score[pos], torch.zeros(n, dtype=torch.int64, device=dev)) if uniq.numel() \
        else torch.zeros(n, dtype=torch.int64, device=dev)
    if evict_to is not None:   # candidates: other labels with room
        ok = (lab_of != labels[u_of]) & (size[lab_of] < evict_to)
        score = torch.where(ok, score, torch.zeros_like(score))
    if pull_to is not None:    # candidates: undersized labels, for nodes of labels that can spare one
        ok = (lab_of != labels[u_of]) & (size[lab_of] < pull_to) & (size[labels[u_of]] > pull_to)
        score = torch.where(ok, score, torch.zeros_like(score))
    best = torch.zeros(n, dtype=torch.int64, device=dev).scatter_reduce_(0, u_of, score, "amax", include_self=True)
    cand = torch.where((score == best[u_of]) & (score > 0), lab_of, torch.full_like(lab_of, n_labels))
    best_lab = torch.full((n,), n_labels, dtype=torch.int64, device=dev).scatter_reduce_(0, u_of, cand, "amin", include_self=True)
    return _sweep_moves(labels, nw, cap, n_labels, parity, evict_to, pull_to, size, best, best_lab, cur)


def _sweep_moves(labels, nw, cap, n_labels, parity, evict_to, pull_to, size, best, best_lab, cur):
    """Second half of a sweep: which of the proposed moves are carried out (gain, capacity, one target at a time)."""
    n = labels.numel()
    dev = labels.device
    gain = best - cur
    if pull_to is not None:
        want = (best_lab < n_labels) & parity
        cap = pull_to
    elif evict_to is None:
        want = (best_lab < n_labels) & (best_lab != labels) & (gain > 0) & parity
    else:
        over = size > evict_to
        want = over[labels] & parity
        # nodes of an oversized label without a connected label that has room go to the emptiest label
        lonely = want & (best_lab >= n_labels)
        if bool(lonely.any()):
            best_lab = torch.where(lonely, torch.argmin(size).expand_as(best_lab), best_lab)
    idx = torch.nonzero(want).flatten()
    if idx.numel() == 0:
        return labels, 0
    if evict_to is not None:   # per oversized label only as many as its excess, least costly first
        o1 = torch.sort(gain[idx], descending=True, stable=True).indices
        idx = idx[o1]
        o2 = torch.sort(labels[idx], stable=True).indices
        idx = idx[o2]
        lab_s = labels[idx]
        wt = nw[idx]
        cum = torch.cumsum(wt, 0)
        first = torch.ones_like(lab_s, dtype=torch.bool)
        first[1:] = lab_s[1:] != lab_s[:-1]
        start = torch.cummax(torch.where(first, cum - wt, torch.zeros_like(cum)), 0).values
        idx = idx[(cum - start - wt) < (size[lab_s] - evict_to)]   # leaves while still above the target
        if idx.numel() == 0:
            return labels, 0
        cap = evict_to
    # capacity: candidates of one target in order of decreasing gain, accepted while they fit
    o1 = torch.sort(gain[idx], descending=True, stable=True).indices
    idx = idx[o1]
    o2 = torch.sort(best_lab[idx], stable=True).indices
    idx = idx[o2]
    tgt = best_lab[idx]
    wt = nw[idx]
    cum = torch.cumsum(wt, 0)
    first = torch.ones_like(tgt, dtype=torch.bool)
    first[1:] = tgt[1:] != tgt[:-1]
    start = torch.where(first, cum - wt, torch.zeros_like(cum))
    start = torch.cummax(start, 0).values            # running offset of the candidate's segment
    fits = size[tgt] + (cum - start) <= cap
    idx, tgt = idx[fits], tgt[fits]
    out = labels.clone()
    out[idx] = tgt
    return out, int(idx.numel())


def _rowptr(src, n):
    """Row pointers of an edge list sorted by source (the kernel's CSR view of it); None on the CPU, where the
    torch formulation of the sweep runs."""
    if not src.is_cuda:
        return None
    return torch.searchsorted(src, torch.arange(n + 1, device=src.device, dtype=torch.int64))


def _parity(n, t, dev):
    """Half of the nodes per sweep (a hash bit): two neighbours never swap labels in the same sweep."""
    h = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761) >> 7
    return ((h + t) & 1) == 0


def _contract(labels, src, dst, w, nw):
    """Clusters become nodes: returns (map old label -> coarse id per node, coarse src, dst, w, nw)."""
    dev = labels.device
    uniq, cid = torch.unique(labels, return_inverse=True)
    nc = int(uniq.numel())
    cnw = torch.zeros(nc, dtype=torch.int64, device=dev).scatter_add_(0, cid, nw)
    cs, cd = cid[src], cid[dst]
    keep = cs != cd
    cs, cd, cw = cs[keep], cd[keep], w[keep]
    key = cs * nc + cd
    ukey, seg = torch.unique(key, return_inverse=True)
    cw2 = torch.zeros(ukey.numel(), dtype=torch.int64, device=dev).scatter_add_(0, seg, cw)
    cs2 = torch.div(ukey, nc, rounding_mode="floor")
    return cid, cs2, ukey - cs2 * nc, cw2, cnw


def _pack(weights: np.ndarray, k: int, cap: int, conn=None) -> np.ndarray:
    """Clusters of the coarsest graph -> k parts: the k heaviest open the parts, the others go,
    heaviest first, to the part they are connected to most strongly among those with room
    (conn: dict cluster -> {cluster: weight}), else to the lightest part."""
    nc = len(weights)
    order = np.argsort(-weights, kind="stable")
    part = np.full(nc, -1, dtype=np.int64)
    load = np.zeros(k, dtype=np.int64)
    for p, c in enumerate(order[:k]):
        part[c] = p
        load[p] = weights[c]
    heap = [(int(load[p]), p) for p in range(k)]
    heapq.heapify(heap)
    for c in order[k:]:
        placed = False
        if conn is not None and c in conn:
            pull = {}
            for nb, wt in conn[c].items():
                if part[nb] >= 0:
                    pull[part[nb]] = pull.get(part[nb], 0) + wt
            for p, _ in sorted(pull.items(), key=lambda kv: (-kv[1], kv[0])):
                if load[p] + weights[c] <= cap:
                    part[c] = p
                    load[p] += weights[c]
                    heapq.heappush(heap, (int(load[p]), p))   # the entry with the old load is skipped when it surfaces
                    placed = True
                    break
        if not placed:
            while True:
                ld, p = heapq.heappop(heap)
                if ld == load[p]:
                    break
            part[c] = p
            load[p] += weights[c]
            heapq.heappush(heap, (int(load[p]), p))
    return part


def _cap(n: int, k: int, imbalance: float) -> int:
    return max(int(np.ceil(n / k * (1.0 + imbalance))), int(np.ceil(n / k)) + 1)


def kway_partition(n: int, src: torch.Tensor, dst: torch.Tensor, k: int, imbalance: float = 0.03,
                   coarsen_sweeps: int = 8, refine_sweeps: int = 6) -> torch.Tensor:
    """Part id (int64 [n], values 0..k-1) of every node of the undirected graph given as a
    symmetric edge list (both directions present, no self loops), on the device of `src`."""
    dev = src.device
    if k < 1:
        raise ValueError("k must be positive")
    if k > max(n, 1):
        raise ValueError("more parts than nodes")
    if k == 1 or n == 0:
        return torch.zeros(n, dtype=torch.int64, device=dev)
    cap = _cap(n, k, imbalance)
    src, dst = src.to(torch.int64), dst.to(torch.int64)
    if src.numel() > 1 and not bool((src[1:] >= src[:-1]).all()):   # the sweeps want the edges grouped by source
        order = torch.sort(src, stable=True).indices
        src, dst = src[order], dst[order]
    graphs = [(src, dst, torch.ones(src.numel(), dtype=torch.int64, device=dev),
               torch.ones(n, dtype=torch.int64, device=dev))]
    maps = []

    # ---- coarsening: merge along the heaviest connections, contract, repeat
    while graphs[-1][3].numel() > 8 * k and graphs[-1][0].numel() > 0:
        s_, d_, w_, nw_ = graphs[-1]
        cur_n = int(nw_.numel())
        labels = torch.arange(cur_n, device=dev, dtype=torch.int64)
        ccap = max(2, cap // 8) if cur_n > 32 * k else max(2, cap // 2)
        idle = 0
        rp = _rowptr(s_, cur_n)
        for t in range(coarsen_sweeps):
            labels, moved = _sweep(labels, s_, d_, w_, nw_, ccap, cur_n, _parity(cur_n, t, dev), rowptr=rp)
            idle = idle + 1 if moved == 0 else 0
            if idle == 2:   # both halves of the nodes had their turn
                break
        cid, cs, cd, cw, cnw = _contract(labels, s_, d_, w_, nw_)
        if int(cnw.numel()) > 0.9 * cur_n:   # no longer shrinking (isolated nodes, capacity)
            break
        maps.append(cid)
        graphs.append((cs, cd, cw, cnw))

    # ---- coarsest graph: merge up to the part capacity, then pack into exactly k parts
    s_, d_, w_, nw_ = graphs[-1]
    cur_n = int(nw_.numel())
    labels = torch.arange(cur_n, device=dev, dtype=torch.int64)
    if s_.numel() > 0:
        idle = 0
        rp = _rowptr(s_, cur_n)
        for t in range(4 * coarsen_sweeps):
            labels, moved = _sweep(labels, s_, d_, w_, nw_, cap, cur_n, _parity(cur_n, t, dev), rowptr=rp)
            idle = idle + 1 if moved == 0 else 0
            if idle == 2:
                break
    uniq, cid = torch.unique(labels, return_inverse=True)
    nc = int(uniq.numel())
    if nc > k:
        cweights = torch.zeros(nc, dtype=torch.int64, device=dev).scatter_add_(0, cid, nw_).cpu().numpy()
        conn = None
        if s_.numel() > 0:
            cs, cd = cid[s_], cid[d_]
            keep = cs != cd
            key, seg = torch.unique(cs[keep] * nc + cd[keep], return_inverse=True)
            if key.numel() <= 4_000_000:
                cw2 = torch.zeros(key.numel(), dtype=torch.int64, device=dev).scatter_add_(0, seg, w_[keep])
                kh, wh = key.cpu().numpy(), cw2.cpu().numpy()
                conn = {}
                for a, b, x in zip((kh // nc).tolist(), (kh % nc).tolist(), wh.tolist()):
                    conn.setdefault(a, {})[b] = x
        part = torch.from_numpy(_pack(cweights, k, cap, conn)).to(dev)[cid]
    else:
        part = cid

    loose = cap + max(2, cap // 10)   # refinement may overfill a part for a while: full parts could not trade nodes otherwise

    def refine(part, graph, t0):
        s_, d_, w_, nw_ = graph
        if s_.numel() == 0:
            return part
        nn = int(part.numel())
        idle = 0
        rp = _rowptr(s_, nn)
        for t in range(refine_sweeps):
            part, moved = _sweep(part, s_, d_, w_, nw_, loose, k, _parity(nn, t0 + t, dev), rowptr=rp)
            idle = idle + 1 if moved == 0 else 0
            if idle == 2:
                break
        return part

    # ---- way back up: project to the finer level and refine there
    part = refine(part, graphs[-1], 0)
    for lvl in range(len(maps) - 1, -1, -1):
        part = refine(part[maps[lvl]], graphs[lvl], lvl + 1)
    # ---- bring every part back under the capacity (finest level: unit node weights) ...
    s_, d_, w_, nw_ = graphs[0]
    everyone = torch.ones(n, dtype=torch.bool, device=dev)
    rp = _rowptr(s_, n)
    for _ in range(16):
        if int(torch.bincount(part, minlength=k).max()) <= cap or s_.numel() == 0:
            break
        part, moved = _sweep(part, s_, d_, w_, nw_, cap, k, everyone, evict_to=cap, rowptr=rp)
        if moved == 0:
            break
    # ---- ... and no part far below the mean: small parts draw in neighbours, then take what is left
    low = max(1, (n // k) // 2)
    for _ in range(8):
        if int(torch.bincount(part, minlength=k).min()) >= low or s_.numel() == 0:
            break
        part, moved = _sweep(part, s_, d_, w_, nw_, low, k, everyone, pull_to=low, rowptr=rp)
        if moved == 0:
            break
    return _fill_small_parts(_trim_large_parts(part, k, cap), k, low)


def _trim_large_parts(part: torch.Tensor, k: int, cap: int) -> torch.Tensor:
    """No part above `cap`: the last nodes (in node order) of a part beyond its capacity go to the parts with room,
    fullest first.  Only reached when the eviction sweeps found no neighbouring part with room (graphs without
    community structure cut into many parts); k * cap >= n, so there is always room."""
    sizes = torch.bincount(part, minlength=k)
    if int(sizes.max()) <= cap:
        return part
    order = torch.sort(part, stable=True).indices                      # nodes grouped by part, node order inside
    start = torch.cumsum(sizes, 0) - sizes
    rank = torch.arange(part.numel(), device=part.device) - start[part[order]]
    surplus = order[rank >= cap]                                        # the nodes beyond the capacity of their part
    room = (cap - sizes).clamp_(min=0)
    by_room = torch.sort(room, stable=True).indices                    # least room (fullest) first, part order among equals
    by_room = by_room[room[by_room] > 0]
    slots = torch.repeat_interleave(by_room, room[by_room])
    part = part.clone()
    part[surplus] = slots[:surplus.numel()]
    return part


def _fill_small_parts(part: torch.Tensor, k: int, low: int) -> torch.Tensor:
    """Every part gets at least `low` nodes: a smaller part takes the last nodes (in node order) of
    the currently largest part.  Only reached for parts that nothing connects to the rest."""
    sizes = torch.bincount(part, minlength=k)
    small = torch.nonzero(sizes < low).flatten().tolist()
    if not small:
        return part
    part = part.clone()
    for e in small:
        sizes = torch.bincount(part, minlength=k)
        need = low - int(sizes[e])
        while need > 0:
            big = int(torch.argmax(sizes))
            spare = int(sizes[big]) - low
            if big == e or spare <= 0:
                break
            take = min(need, spare)
            members = torch.nonzero(part == big).flatten()
            part[members[members.numel() - take:]] = e
            sizes[big] -= take
            sizes[e] += take
            need -= take
    return part


def edge_cut(part: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> int:
    """Number of undirected edges whose end points lie in different parts."""
    return int((part[src.long()] != part[dst.long()]).sum().item()) // 2


def edge_list_of_csr(indptr: torch.Tensor, indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(src, dst) of a CSR pattern without its diagonal."""
    n = indptr.numel() - 1
    src = torch.repeat_interleave(torch.arange(n, device=indptr.device, dtype=torch.int64), indptr[1:] - indptr[:-1])
    dst = indices.to(torch.int64)
    keep = src != dst
    return src[keep], dst[keep]


def partition_graph(adj, idx_nodes, num_clusters: int, device=None, groups=None):
    """partition_graph of the reference (src/partition_utils.py:30-81): returns (part_adj, parts).

    parts[g] lists the nodes (original ids, in idx_nodes order) of part g; part_adj is the
    adjacency restricted to node pairs of the same part, entries 1, in the shape of `adj`
    (the reference's zero-valued sentinel entry at the last node keeps that shape).  `groups`
    (one part id per entry of idx_nodes) replaces the partitioner, e.g. a METIS result
    computed elsewhere."""
    idx_nodes = np.asarray(idx_nodes)
    num_nodes, num_all = len(idx_nodes), adj.shape[0]
    sub = sp.coo_matrix(adj.tocsr()[idx_nodes, :][:, idx_nodes])
    if groups is None:
        if num_clusters > 1:
            if device is None:
                if not torch.cuda.is_available():
                    raise RuntimeError("partition_graph runs on the MI355X; pass device= explicitly to run it elsewhere")
                device = torch.device("cuda", torch.cuda.current_device())
            off = sub.row != sub.col
            r, c = sub.row[off].astype(np.int64), sub.col[off].astype(np.int64)
            key = np.unique(np.concatenate([r * num_nodes + c, c * num_nodes + r]))   # symmetric, no repeats
            src = torch.from_numpy(key // num_nodes).to(device)
            dst = torch.from_numpy(key % num_nodes).to(device)
            groups = kway_partition(num_nodes, src, dst, num_clusters).cpu().numpy()
        else:
            groups = np.zeros(num_nodes, dtype=np.int64)
    groups = np.asarray(groups)
    order = np.argsort(groups, kind="stable")
    bounds = np.searchsorted(groups[order], np.arange(num_clusters + 1))
    parts = [idx_nodes[order[bounds[g]:bounds[g + 1]]].tolist() for g in range(num_clusters)]
    same = groups[sub.row] == groups[sub.col]
    rows = np.concatenate([idx_nodes[sub.row[same]], [num_all - 1]])
    cols = np.concatenate([idx_nodes[sub.col[same]], [num_all - 1]])
    data = np.concatenate([np.ones(int(same.sum()), dtype=np.int64), [0]])
    part_adj = sp.coo_matrix((data, (rows, cols)), shape=(num_all, num_all)).tocsr()
    return part_adj, parts


def partition_nodes(graph, idx_nodes, num_clusters: int) -> List[np.ndarray]:
    """Parts (arrays of original node ids, in idx_nodes order) of the sub-graph induced by idx_nodes
    of a device-resident graph (sparc_amd.utils.DeviceGraph): the `parts` of partition_graph without
    a round trip of the edge list through the host."""
    dev = graph.device
    idx = torch.as_tensor(np.asarray(idx_nodes), device=dev).to(torch.int64)
    local = torch.full((graph.n,), -1, dtype=torch.int64, device=dev)
    local[idx] = torch.arange(idx.numel(), device=dev, dtype=torch.int64)
    src, dst = edge_list_of_csr(graph.indptr, graph.indices)
    ls, ld = local[src], local[dst]
    keep = (ls >= 0) & (ld >= 0)
    ls, ld = ls[keep], ld[keep]
    m = int(idx.numel())
    key = torch.unique(torch.cat([ls * m + ld, ld * m + ls]))    # symmetric pattern, no repeats
    ls = torch.div(key, m, rounding_mode="floor")
    groups = kway_partition(m, ls, key - ls * m, num_clusters)
    order = torch.sort(groups, stable=True).indices
    bounds = torch.searchsorted(groups[order], torch.arange(num_clusters + 1, device=dev)).cpu().numpy()
    ids = idx[order].cpu().numpy()
    return [ids[bounds[g]:bounds[g + 1]] for g in range(num_clusters)]
