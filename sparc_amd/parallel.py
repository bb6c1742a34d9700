"""Cluster batches sharded over the GPUs of one node: one process per GPU, the MLP
replicated, every rank training on its own batches, and one collective per step, the
all-reduce of the MLP gradient packed into one bucket (RCCL over xGMI through torch.distributed).

The reference is single-device (src/SpectralNet.py:21); its batches are independent
units (each support is the induced sub-block of its own nodes, src/utils.py:83), which
is what makes this sharding exact per batch.  With P ranks one optimiser step averages
the gradients of P batches; at P = 1 it is the reference's step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .ops import LapLoss


def _staged(t: torch.Tensor, group) -> bool:
    """gloo cannot reduce device tensors reliably: rehearsals of the multi-rank path on a GPU
    stage through the host (RCCL, the production backend, reduces in place)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None):
    if _staged(t, group):
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def broadcast(t: torch.Tensor, src: int = 0, group=None):
    if _staged(t, group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)


def shard_batches(n_batches: int, world_size: int, rank: int) -> List[int]:
    """Round-robin ownership: batch b lives on rank b % world_size."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    return list(range(rank, n_batches, world_size))


def all_reduce_mean_gradients(params: Sequence[torch.nn.Parameter], group=None) -> Optional[torch.Tensor]:
    """Average the gradients autograd left in .grad over the ranks with ONE collective: they are packed
    into one bucket (a single copy), reduced, and every .grad becomes a view into the bucket.  Compared
    with gradients that live in a persistent flat buffer this needs no zero-fill and no accumulation
    kernel per parameter.  A no-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return None
    owners = [p for p in params if p.grad is not None]
    if not owners:
        return None
    flat = torch.cat([p.grad.reshape(-1) for p in owners])
    all_reduce_sum(flat, group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for p in owners:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return flat


class LayerwiseReducer:
    """Gradient all-reduce that starts while the backward is still running: every MLP layer's weight and bias
    gradient are copied into that layer's bucket (allocated once, reused by every step) the moment the backward has
    produced them and the bucket's all-reduce is launched asynchronously (RCCL runs it on its own stream), so it
    overlaps the backward products of the layers below.  Autograd keeps the local gradients it was going to keep;
    finish() waits for all buckets, turns the sums into means and makes every reported parameter's .grad a view into
    its bucket -- by assignment, so nothing depends on how autograd stores what the backward returned.

    params: [W0, b0, W1, b1, ...] in the order of the layer numbers the MLP reports.  Their .grad must be None when
    the backward starts (zero_grad(set_to_none=True)): gradients that autograd accumulates in place would be added
    into a bucket whose collective is in flight."""

    def __init__(self, params: Optional[Sequence[torch.nn.Parameter]] = None, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.params = list(params) if params is not None else None
        self.buckets = {}
        self.pending = []
        self.reduced = {}

    def _bucket(self, layer: int, n: int, like: torch.Tensor) -> torch.Tensor:
        flat = self.buckets.get(layer)
        if flat is None or flat.numel() != n or flat.device != like.device or flat.dtype != like.dtype:
            flat = torch.empty(n, dtype=like.dtype, device=like.device)
            self.buckets[layer] = flat
        return flat

    def destination(self, layer: int, w_shape, b_shape, like: torch.Tensor):
        """(weight part, bias part, whole bucket) of layer's bucket, for a backward that can write its gradients there
        itself: on_layer then finds them in place and copies nothing.  The previous step's collective on this bucket
        has been waited for in finish(), and whatever still reads the old means (the optimiser's update) was enqueued
        before the kernels that will overwrite them."""
        n_w, n_b = int(np.prod(w_shape)), int(np.prod(b_shape))
        flat = self._bucket(layer, n_w + n_b, like)
        return flat[:n_w].view(w_shape), flat[n_w:].view(b_shape), flat

    def __call__(self, layer: int, dW: torch.Tensor, db: torch.Tensor):
        return self.on_layer(layer, dW, db)

    def on_layer(self, layer: int, dW: torch.Tensor, db: torch.Tensor):
        if self.params is not None and (self.params[2 * layer].grad is not None or self.params[2 * layer + 1].grad is not None):
            raise RuntimeError("LayerwiseReducer: clear the gradients with zero_grad(set_to_none=True) before the backward")
        if any(layer == p[0] for p in self.pending):
            raise RuntimeError("LayerwiseReducer: a layer reported its gradients twice in one backward (run both batches "
                               "of a step through the MLP in one pass: SpectralNetModel.forward_pair)")
        n = dW.numel()
        flat = self._bucket(layer, n + db.numel(), dW)
        in_place = lambda t, at: t.is_contiguous() and t.data_ptr() == flat.data_ptr() + 4 * at and t.dtype == flat.dtype
        if not in_place(dW, 0):
            flat[:n].view(dW.shape).copy_(dW)          # a transposed view is laid out as the parameter on the way
        if not in_place(db, n):
            flat[n:].view(db.shape).copy_(db)
        averaged = False
        if _staged(flat, self.group):          # gloo rehearsal on a GPU: through the host, synchronously
            all_reduce_sum(flat, self.group)
            work = None
        elif dist.get_backend(self.group) == "nccl":   # RCCL averages inside the collective: no scaling kernel per bucket
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            averaged = True
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((layer, work, flat, dW.shape, db.shape, averaged))
        return dW, db

    def finish(self) -> int:
        """Wait for every bucket, average, and hand the means to the parameters; returns the number of buckets
        reduced.  reduced[layer] = (dW, db) holds the averaged pair (views into the layer's bucket) until the next
        step's backward overwrites the bucket."""
        done = len(self.pending)
        self.reduced = {}
        for layer, work, flat, w_shape, b_shape, averaged in self.pending:
            if work is not None:
                work.wait()
            if not averaged:
                flat.div_(self.world)
            n = flat.numel() - int(np.prod(b_shape))
            gW, gb = flat[:n].view(w_shape), flat[n:].view(b_shape)
            self.reduced[layer] = (gW, gb)
            if self.params is not None:
                self.params[2 * layer].grad = gW
                self.params[2 * layer + 1].grad = gb
        self.pending = []
        return done


class ShardedSpectralTrainer:
    """Data-parallel SpectralNet steps over this rank's shard of cluster batches.  net / criterion can be
    injected (anything with forward_pair(x2, x1, top_idx=) and __call__(batch, Y, is_normalized=)); by
    default they are the MI355X SpectralNetModel and SpectralNetLoss."""

    def __init__(self, architecture: dict, input_dim: int, device, lr: float = 1e-3, orth_mode: str = "reference",
                 group=None, seed: Optional[int] = 0, net=None, criterion=None, fused_adam: bool = True,
                 hip_graph: bool = False, reduce_in_a_world_of_one: bool = False):
        self.device = torch.device(device)
        if seed is not None:
            torch.manual_seed(seed)  # identical initial weights on every rank
        if net is None:
            from .SpectralTrainer import SpectralNetModel
            net = SpectralNetModel(architecture, input_dim, self.device, orth_mode=orth_mode).to(self.device)
        if criterion is None:
            from .SpectralTrainer import SpectralNetLoss
            criterion = SpectralNetLoss()
        self.net = net
        self.k = architecture["output_dim"]
        self.group = group
        # reduce_in_a_world_of_one: run the bucket copies and the collectives even with a single rank (a measurement of
        # what the data-parallel step adds on one GPU; the result is the single-process step)
        self.distributed = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(group) > 1 or reduce_in_a_world_of_one)
        if self.distributed:
            for p in self.net.parameters():
                broadcast(p.data, 0, group)
        self.params = list(self.net.parameters())
        self.hip_graph = bool(hip_graph) and not self.distributed   # whole steps replayed as HIP graphs (one GPU)
        if self.hip_graph:
            self.lr = torch.tensor(float(lr), device=self.device)
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=self.lr, fused=True, capturable=True)
        elif fused_adam and self.device.type == "cuda":
            from .optim import Adam
            self.optimizer = Adam(self.net.parameters(), lr=lr)
        else:
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr)
        self._graphed = None
        self.criterion = criterion
        self.reducer = None
        if self.distributed:
            self.attach_reducer(LayerwiseReducer(self.params, group))
        self.features = []
        self.batches = []

    def attach_reducer(self, reducer: LayerwiseReducer):
        """Average the gradients through this reducer: a model that reports its layers during the backward
        (SpectralNetModel.grad_ready_hook) feeds it layer by layer; for any other model step() falls back to one
        bucket after the backward."""
        self.reducer = reducer
        if hasattr(self.net, "grad_ready_hook"):
            self.net.grad_ready_hook = reducer   # (called as hook(layer, dW, db); its destination() lets the backward write into the buckets)

    def add_batch(self, features: torch.Tensor, batch) -> int:
        self.features.append(features.to(self.device, dtype=torch.float32).contiguous())
        self.batches.append(batch)
        return len(self.batches) - 1

    def step(self, pid: int, pid_2: int, top_idx=None) -> torch.Tensor:
        """Orthogonality step on local batch pid_2, gradient step on local batch pid
        (src/SpectralTrainer.py:216-238), gradient averaged over the ranks before Adam: layer by layer
        while the backward runs when the MLP reports its layers (SpectralNetModel.grad_ready_hook), in one bucket after
        the backward otherwise."""
        if self.hip_graph:
            if self._graphed is None:
                from .stepgraph import GraphedSteps
                self._graphed = GraphedSteps(self.net, self.criterion, self.optimizer, self.features, self.batches,
                                             self.k, self.device)
            return self._graphed.step(pid, pid_2, top_idx)
        net = self.net
        net.train()
        self.optimizer.zero_grad(set_to_none=True)   # autograd hands its gradients over as they are: no accumulation
        Y = net.forward_pair(self.features[pid_2], self.features[pid], top_idx=top_idx)
        loss = self.criterion(self.batches[pid], Y, is_normalized=True)
        loss.backward(gradient=LapLoss.unit_gradient(loss))
        if self.distributed and self.reducer.finish() == 0:
            all_reduce_mean_gradients(self.params, self.group)   # no layer reported: one bucket, one collective
        self.optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def validation_loss(self, features: Sequence[torch.Tensor], batches: Sequence) -> torch.Tensor:
        """Sum of the validation losses of this rank's validation batches, summed over the ranks."""
        self.net.eval()
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        for x, b in zip(features, batches):
            Y = self.net(x.to(self.device), should_update_orth_weights=False)
            total += self.criterion(b, Y, is_normalized=True)
        if self.distributed:
            all_reduce_sum(total, self.group)
        return total
