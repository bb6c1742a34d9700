"""Support / degree construction on the device, with the reference's function names
(src/utils.py:22-137): cluster concatenation, induced sub-blocks, the two row
normalisations and the COO tuple, produced by the HIP kernels in float64 with the
reference's rounding sequence (values and indices are bit-identical to SciPy's once a
tuple is sorted by (row, col); the tuples here are always emitted in that order).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ._lib import check, lib
from .graph import Workspace, _ptr, _stream, require_gpu

MODE_RAW, MODE_ROWNORM, MODE_ENHANCE = 0, 1, 2


def sample_mask(idx, l):
    """src/utils.py:15-19."""
    mask = np.zeros(l)
    mask[idx] = 1
    return np.array(mask, dtype=bool)


class DeviceGraph:
    """A whole graph's adjacency as CSR in HBM: indptr int64[n+1], indices int32 (sorted
    inside a row), data float64."""

    def __init__(self, n, indptr, indices, data):
        self.n, self.indptr, self.indices, self.data = int(n), indptr, indices, data
        self.device = indptr.device

    @classmethod
    def from_scipy(cls, adj, device) -> "DeviceGraph":
        device = require_gpu(device)
        adj = adj.tocsr()
        if not adj.has_sorted_indices:
            adj = adj.sorted_indices()
        adj.sum_duplicates()
        return cls(adj.shape[0],
                   torch.from_numpy(adj.indptr.astype(np.int64)).to(device),
                   torch.from_numpy(adj.indices.astype(np.int32)).to(device),
                   torch.from_numpy(adj.data.astype(np.float64)).to(device))

    def subblock(self, pt, mode=MODE_RAW, binarize=False, diag_lambda=-1.0):
        """adj[pt,:][:,pt] (optionally binarised / normalised) as a device COO sorted by (row, col).

        Returns (coords int32 [nnz,2], values float64 [nnz], (m, m), rowsum float64 [m])."""
        dev = self.device
        pt_d = torch.as_tensor(np.ascontiguousarray(pt, dtype=np.int32)).to(dev) if not torch.is_tensor(pt) \
            else pt.to(device=dev, dtype=torch.int32).contiguous()
        m = int(pt_d.numel())
        flavour = int(mode) | (16 if binarize else 0)
        row_nnz = torch.empty(m + 1, dtype=torch.int64, device=dev)
        rowsum = torch.empty(m, dtype=torch.float64, device=dev)
        nbytes = lib.sparc_subblock_workspace_bytes(self.n, m)
        ws = Workspace.get(nbytes, dev)
        with torch.cuda.device(dev):
            check(lib.sparc_subblock_count(self.n, _ptr(self.indptr), _ptr(self.indices), _ptr(self.data), m, _ptr(pt_d),
                                           flavour, _ptr(row_nnz), _ptr(rowsum), _ptr(ws), nbytes, _stream()),
                  "sparc_subblock_count")
            total = int(row_nnz[m])
            coords = torch.empty(max(total, 1), 2, dtype=torch.int32, device=dev)
            values = torch.empty(max(total, 1), dtype=torch.float64, device=dev)
            check(lib.sparc_subblock_fill(self.n, _ptr(self.indptr), _ptr(self.indices), _ptr(self.data), m, _ptr(pt_d),
                                          flavour, float(diag_lambda), _ptr(row_nnz), _ptr(rowsum), _ptr(coords),
                                          _ptr(values), _ptr(ws), nbytes, _stream()), "sparc_subblock_fill")
        return coords[:total], values[:total], (m, m), rowsum


def _as_graph(adj, device):
    return adj if isinstance(adj, DeviceGraph) else DeviceGraph.from_scipy(adj, device)


def _default_device():
    return torch.device("cuda", torch.cuda.current_device())


def preprocess_multicluster(adj, parts, features, y_train, train_mask, num_clusters, block_size, diag_lambda=-1,
                            device=None):
    """Generate the batches for multiple clusters (src/utils.py:60-98).

    Same arguments and return value as the reference; `adj` may be a SciPy matrix or a
    DeviceGraph.  Support tuples hold device tensors (coords int32, values float64)."""
    device = device or _default_device()
    graph = _as_graph(adj, device)
    features_batches, support_batches, y_train_batches, train_mask_batches = [], [], [], []
    np.random.shuffle(parts)
    for st in range(0, num_clusters, block_size):
        pt = parts[st]
        for pt_idx in range(st + 1, min(st + block_size, num_clusters)):
            pt = np.concatenate((pt, parts[pt_idx]), axis=0)
            pt = pt.astype(np.int32)
        features_batches.append(torch.tensor(features[pt, :]).float())
        y_train_batches.append(torch.tensor(y_train[pt, :]).float())
        mode = MODE_RAW if diag_lambda == -1 else MODE_ENHANCE
        coords, values, shape, _ = graph.subblock(pt, mode=mode, binarize=False, diag_lambda=diag_lambda)
        support_batches.append((coords, values, shape))
        train_mask_batches.append(np.asarray(train_mask)[pt].astype(bool))
    return features_batches, support_batches, y_train_batches, train_mask_batches


def preprocess(adj, features, y_train, train_mask, visible_data, num_clusters, diag_lambda=-1, parts=None,
               device=None):
    """Per-partition batches for validation / test (src/utils.py:101-137).

    The reference obtains `parts` from METIS inside this function (partition_graph,
    src/partition_utils.py:56); here they come from the device partitioner
    (sparc_amd.partition) unless the caller passes `parts` (e.g. a METIS result)."""
    device = device or _default_device()
    graph = _as_graph(adj, device)
    if parts is None:
        visible = np.asarray(visible_data)
        if num_clusters == 1:
            parts = [visible]
        else:
            from .partition import partition_nodes
            parts = partition_nodes(graph, visible, num_clusters)
    parts = [np.array(pt) for pt in parts]
    features_batches, support_batches, y_train_batches, train_mask_batches = [], [], [], []
    mode = MODE_ROWNORM if diag_lambda == -1 else MODE_ENHANCE
    for pt in parts:
        features_batches.append(torch.tensor(features[pt, :]).float())
        coords, values, shape, _ = graph.subblock(pt, mode=mode, binarize=True, diag_lambda=diag_lambda)
        support_batches.append((coords, values, shape))
        y_train_batches.append(torch.tensor(y_train[pt, :]).float())
        train_mask_batches.append(np.asarray(train_mask)[pt].astype(bool))
    return parts, features_batches, support_batches, y_train_batches, train_mask_batches
