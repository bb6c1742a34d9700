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

from sparc_amd.parallel import all_reduce_mean_gradients, shard_batches


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
