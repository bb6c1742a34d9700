"""Whole training steps replayed as HIP graphs.

A SpectralNet step on a small batch (Cora: 2.4 k rows, PubMed-shape: 6 k) is about 45 kernel launches of a few
microseconds each, so a host that cannot launch that fast bounds it (the hosts measured so far can: replay and eager
step take the same time there).  Every launch of the step is stream-ordered and
allocation-free once warm (the kernels of libsparc_hip.so take the stream as an argument and a workspace from
the caller), so the step -- forward of both batches, orthonormalisation, Laplacian chain, backward, fused Adam --
is captured once per (gradient batch, orthogonality batch) pair and replayed with one launch.  What changes from
step to step is handed over in static device buffers: the k leading rows of the orthogonality batch's
permutation (src/SpectralTrainer.py:218-223) and the learning rate.

A pair's first visit costs an eager step plus the capture, and there are (number of batches) squared pairs, so the
mode pays where the batches are few and visited often (Cora: one, PubMed: a handful).  With twelve Amazon2M-shape
batches most of the first hundred steps meet a new pair and the run is slower than eager (7.5 against 4.2 ms per
step); MAX_GRAPHS bounds what is captured, later new pairs run eagerly.
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import Workspace
from .ops import LapLoss


class GraphedSteps:
    """step(pid, pid_2, top_idx) with the semantics of ShardedSpectralTrainer.step on one GPU.  The optimiser
    must be capturable (torch.optim.Adam(..., capturable=True)) with its learning rate in a device tensor."""

    RING = 8   # pinned staging buffers for the permutation heads: the host may run this many steps ahead
    MAX_GRAPHS = 64   # pairs captured at most; a pair met after that runs eagerly every time

    def __init__(self, net, criterion, optimizer, features, batches, k: int, device):
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.features, self.batches, self.k = features, batches, int(k)
        self.device = torch.device(device)
        self.top_idx = torch.zeros(self.k, dtype=torch.int32, device=self.device)
        # the orthonormalisation matrix of the last step outlives the step (validation and predict use it,
        # src/SpectralTrainer.py:69-70), so it is copied out of the graphs' shared memory pool
        self.orth = torch.zeros(self.k, self.k, dtype=torch.float32, device=self.device)
        self._host = [torch.zeros(self.k, dtype=torch.int32).pin_memory() for _ in range(self.RING)]
        self._free = [None] * self.RING       # event recorded after the copy out of each staging buffer
        self._turn = 0
        self.graphs = {}
        self.pool = None
        self._keep = []                       # the shared workspace buffers the graphs were captured with

    def _eager(self, pid, pid_2):
        self.optimizer.zero_grad(set_to_none=True)
        Y = self.net.forward_pair(self.features[pid_2], self.features[pid], top_idx=self.top_idx)
        loss = self.criterion(self.batches[pid], Y, is_normalized=True)
        loss.backward(gradient=LapLoss.unit_gradient(loss))
        self.optimizer.step()
        self.orth.copy_(self.net.orthonorm_weights.detach())
        return loss.detach()

    def _stage(self, top_idx, m2: int):
        slot = self._turn % self.RING
        self._turn += 1
        if self._free[slot] is not None:
            self._free[slot].synchronize()
        head = np.arange(self.k, dtype=np.int32) if top_idx is None else np.asarray(top_idx, dtype=np.int32)[:self.k]
        buf = self._host[slot].numpy()
        # an orthogonality batch of fewer than k rows leaves the tail without a row: its row count is the index that
        # reads as a row of zeros (as ops.OrthWeights pads it)
        buf[:] = m2
        buf[:head.shape[0]] = head
        self.top_idx.copy_(self._host[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._free[slot] = ev

    def step(self, pid: int, pid_2: int, top_idx=None) -> torch.Tensor:
        self.net.train()
        self._stage(top_idx, int(self.features[pid_2].shape[0]))
        key = (pid, pid_2)
        hit = self.graphs.get(key)
        if hit is None and len(self.graphs) >= self.MAX_GRAPHS:
            loss = self._eager(pid, pid_2)
            self.net.orthonorm_weights = self.orth
            return loss.clone()
        if hit is None:
            # the pair's first step runs eagerly on a side stream (it sizes the workspace and configures the
            # kernels); the capture that follows records the pair's NEXT step without executing it
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                loss = self._eager(pid, pid_2)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(g, pool=self.pool):
                out = self._eager(pid, pid_2)
            if self.pool is None:
                self.pool = g.pool()
            self._keep.append(Workspace._buffers.get(self.device.index or 0))
            self.graphs[key] = (g, out)
            self.net.orthonorm_weights = self.orth
            return loss.clone()
        g, out = hit
        g.replay()
        self.net.orthonorm_weights = self.orth
        return out.clone()
