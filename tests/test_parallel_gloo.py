"""world_size-2 CPU (gloo) tests of the data-parallel plumbing: batch sharding and the
bucketed gradient all-reduce.  The kernels need a GPU; here a plain torch MLP stands in so that
the collective logic is exercised end to end."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparc_amd.parallel import LayerwiseReducer, ShardedSpectralTrainer, all_reduce_mean_gradients, shard_batches


def test_shard_batches_partition():
    for n, w in [(100, 8), (7, 2), (3, 4), (0, 2)]:
        seen = sorted(b for r in range(w) for b in shard_batches(n, w, r))
        assert seen == list(range(n))
        sizes = [len(shard_batches(n, w, r)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_batches(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    params = list(net.parameters())
    opt = torch.optim.Adam(params, lr=1e-2)
    x = torch.randn(16, 5, generator=torch.Generator().manual_seed(100 + rank))
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        net(x).pow(2).sum().backward()
        local = torch.cat([p.grad.flatten() for p in params]).clone()
        flat = all_reduce_mean_gradients(params)
        assert all(p.grad.data_ptr() >= flat.data_ptr() for p in params)   # views into the bucket
        opt.step()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    weights = torch.cat([p.detach().flatten() for p in params])
    allw = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(allw, weights)
    if rank == 0:
        out.put((torch.stack(gathered).mean(0).numpy(), flat.clone().numpy(), torch.stack(allw).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mean_local, reduced, weights = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(mean_local, reduced, atol=1e-6)
    assert np.array_equal(weights[0], weights[1])
    net = torch.nn.Linear(4, 2)
    net(torch.ones(3, 4)).sum().backward()
    assert all_reduce_mean_gradients(list(net.parameters())) is None   # no process group: a no-op


class _StubNet(torch.nn.Module):
    """Stands in for SpectralNetModel on the CPU: an MLP with the trainer-facing forward_pair (orthogonality batch
    first, gradient batch second; the orthogonality batch feeds the output through its column scale, so the
    gradient flows through both batches as in the reference)."""

    def __init__(self):
        super().__init__()
        self.layers = torch.nn.Sequential(torch.nn.Linear(5, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3), torch.nn.Tanh())

    def forward_pair(self, x2, x1, top_idx=None):
        y = self.layers(torch.cat([x2, x1]))
        y2, y1 = y[:x2.shape[0]], y[x2.shape[0]:]
        return y1 / y2.pow(2).mean(0, keepdim=True).sqrt()

    def forward(self, x, should_update_orth_weights=False):
        return self.layers(x)


def _stub_loss(batch, Y, is_normalized=True):
    return ((batch @ Y) * Y).sum().double() / Y.shape[0]     # a quadratic form of the batch's matrix, like the spectral loss


def _trainer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234 + rank)            # different seeds: the trainer must broadcast rank 0's weights
    tr = ShardedSpectralTrainer({"output_dim": 3}, 5, "cpu", lr=1e-2, seed=None, net=_StubNet(), criterion=_stub_loss,
                                fused_adam=False)
    gen = torch.Generator().manual_seed(7)
    feats = [torch.randn(12 + b, 5, generator=gen) for b in range(4)]
    mats = [torch.randn(12 + b, 12 + b, generator=gen) for b in range(4)]
    mine = shard_batches(4, world, rank)
    for b in mine:
        tr.add_batch(feats[b], mats[b] @ mats[b].t())
    w0 = torch.cat([p.detach().flatten() for p in tr.params]).clone()
    losses = [float(tr.step(s % len(mine), (s + 1) % len(mine))) for s in range(3)]
    w = torch.cat([p.detach().flatten() for p in tr.params])
    val = float(tr.validation_loss([feats[b] for b in mine], [mats[b] @ mats[b].t() for b in mine]))
    # the layer-wise reducer on its own: buckets launched one by one, averaged on finish
    red = LayerwiseReducer()
    gW, gb = torch.full((3, 4), float(rank + 1)), torch.full((3,), float(10 * (rank + 1)))
    vW, vb = red.on_layer(0, gW, gb)
    vW2, _ = red.on_layer(1, gW * 2, gb)
    n_done = red.finish()
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    g0 = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(g0, w0)
    if rank == 0:
        out.put({"w": torch.stack(gathered).numpy(), "w0": torch.stack(g0).numpy(), "losses": losses, "val": val,
                 "vW": vW.numpy(), "vb": vb.numpy(), "vW2": vW2.numpy(), "n_done": n_done})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_trainer_step_world2():
    """ShardedSpectralTrainer.step end to end on two gloo ranks (stub model and loss): same initial weights,
    averaged gradients, identical weights after three steps, and the result of one process that averages
    the two ranks' gradients by hand."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res["w0"][0], res["w0"][1])            # broadcast of the initial weights
    assert np.array_equal(res["w"][0], res["w"][1])              # replicated Adam on averaged gradients
    assert not np.array_equal(res["w"][0], res["w0"][0])
    assert np.allclose(res["vW"], 1.5) and np.allclose(res["vb"], 15.0) and np.allclose(res["vW2"], 3.0) and res["n_done"] == 2
    # the same three steps in one process, gradients of the two ranks' batches averaged by hand
    torch.manual_seed(1234)
    net = _StubNet()
    assert np.allclose(torch.cat([p.detach().flatten() for p in net.parameters()]).numpy(), res["w0"][0])
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    gen = torch.Generator().manual_seed(7)
    feats = [torch.randn(12 + b, 5, generator=gen) for b in range(4)]
    mats = [torch.randn(12 + b, 12 + b, generator=gen) for b in range(4)]
    shards = [shard_batches(4, 2, r) for r in range(2)]
    for s in range(3):
        opt.zero_grad()
        total = 0
        for r in range(2):
            mine = shards[r]
            b1, b2 = mine[s % len(mine)], mine[(s + 1) % len(mine)]
            total = total + _stub_loss(mats[b1] @ mats[b1].t(), net.forward_pair(feats[b2], feats[b1])) / 2
        total.backward()
        opt.step()
    want = torch.cat([p.detach().flatten() for p in net.parameters()]).numpy()
    assert np.allclose(res["w"][0], want, rtol=1e-5, atol=1e-6)
    assert np.isfinite(res["val"]) and all(np.isfinite(res["losses"]))
