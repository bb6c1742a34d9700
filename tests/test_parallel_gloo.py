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
    lin = [torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)]
    red = LayerwiseReducer([p for l in lin for p in (l.weight, l.bias)])
    gW, gb = torch.full((3, 4), float(rank + 1)), torch.full((3,), float(10 * (rank + 1)))
    kept = red.on_layer(0, gW, gb)
    assert kept[0] is gW and kept[1] is gb          # autograd keeps the local pair; the means arrive through finish()
    red.on_layer(1, (gW * 2).t().contiguous().t(), gb)   # a gradient that arrives as a transposed view
    try:
        red.on_layer(1, gW, gb)
        raise AssertionError("a layer reported twice must be refused")
    except RuntimeError:
        pass
    n_done = red.finish()
    (vW, vb), (vW2, _) = red.reduced[0], red.reduced[1]
    assert lin[0].weight.grad.data_ptr() == vW.data_ptr() and lin[1].bias.grad is red.reduced[1][1]
    first_bucket = red.buckets[0].data_ptr()
    lin[0].weight.grad = lin[0].bias.grad = None
    red.on_layer(0, gW, gb)                          # the next step reuses the layer's bucket
    red.finish()
    assert red.buckets[0].data_ptr() == first_bucket and torch.equal(lin[0].weight.grad, vW)
    try:
        red.on_layer(0, gW, gb)                      # gradients not cleared: refused
        raise AssertionError("a stale .grad must be refused")
    except RuntimeError:
        pass
    # a backward that writes its gradients into the bucket itself: destination() names the place, on_layer finds them
    # there, and the collective averages them like copied ones
    vW, vb = vW.clone(), vb.clone()                  # (views into the bucket that is about to be written again)
    lin[0].weight.grad = lin[0].bias.grad = None
    dW_to, db_to, whole = red.destination(0, (3, 4), (3,), gW)
    assert whole.data_ptr() == first_bucket and dW_to.data_ptr() == first_bucket and db_to.data_ptr() == first_bucket + 4 * 12
    dW_to.fill_(float(3 * (rank + 1)))
    db_to.fill_(float(5 * (rank + 1)))
    got = red(0, dW_to, db_to)                       # (the reducer itself is the hook)
    assert got[0] is dW_to and got[1] is db_to
    red.finish()
    mean = (1 + world) / 2.0
    assert torch.equal(lin[0].weight.grad, torch.full((3, 4), 3 * mean)) and torch.equal(lin[0].bias.grad, torch.full((3,), 5 * mean))
    assert lin[0].weight.grad.data_ptr() == first_bucket
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


# ---- SpectralTrainer.train inside a process group (the epoch loop on CPU stand-ins) ------------------------------
def _make_problem():
    gen = torch.Generator().manual_seed(11)
    feats = [torch.randn(10 + b, 5, generator=gen) for b in range(5)]
    mats = [torch.randn(10 + b, 10 + b, generator=gen) for b in range(5)]
    vfeats = [torch.randn(9 + b, 5, generator=gen) for b in range(3)]
    vmats = [torch.randn(9 + b, 9 + b, generator=gen) for b in range(3)]
    psd = lambda m: m @ m.t()
    return feats, [psd(m) for m in mats], vfeats, [psd(m) for m in vmats]


_EPOCH_CFG = {"dataset": "gloo_epochs", "spectral": {"lr": 1e-2, "epochs": 3, "lr_decay": 0.1, "patience": 10, "min_lr": 1e-8,
                                                   "batch_size": 1, "n_clusters": 3, "architecture": {"output_dim": 3}}}


def _epoch_trainer(tmpdir):
    from sparc_amd.SpectralTrainer import SpectralTrainer

    class StubTrainer(SpectralTrainer):
        def _setup_model(self, input_dim):
            self.criterion = _stub_loss
            self.spectral_net = _StubNet()

        def _get_valid_affinity_matrix(self, W, X=None):
            return W

    tr = StubTrainer(_EPOCH_CFG, torch.device("cpu"))
    tr.weights_path = os.path.join(tmpdir, "w", "spectralnet_weights.pth")
    tr.orth_weights_path = os.path.join(tmpdir, "w", "orth.pth")
    return tr


def _epochs_worker(rank, world, port, tmpdir, out):
    import contextlib
    import io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, mats, vfeats, vmats = _make_problem()
    torch.manual_seed(50 + rank)                 # different initial weights: rank 0's must win
    np.random.seed(0)
    tr = _epoch_trainer(os.path.join(tmpdir, f"rank{rank}"))
    tr._setup_model(5)
    tr.spectral_net.orthonorm_weights = torch.eye(3)
    tr._setup_optimizer(False)
    mine, mine_val = tr._my_share(5, 3, world, rank)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        net = tr._run_epochs([feats[b] for b in mine], [mats[b] for b in mine], [vfeats[b] for b in mine_val],
                             [vmats[b] for b in mine_val], 3, n_val_total=3)
    w = torch.cat([p.detach().flatten() for p in net.parameters()])
    allw = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(allw, w)
    out.put((rank, buf.getvalue(), torch.stack(allw).numpy(), os.path.exists(tr.weights_path), mine, mine_val))
    dist.barrier()
    dist.destroy_process_group()


def test_spectral_trainer_epochs_world2(tmp_path):
    """SpectralTrainer's epoch loop on two gloo ranks (stand-in model and loss on the CPU): five training batches dealt
    3 + 2, three validation batches dealt 2 + 1; both replicas end with the same weights, which are those of one
    process that plays both ranks (same draws, gradients averaged by hand); the printed losses are the sums over
    both ranks; rank 0 alone prints and writes the checkpoint."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epochs_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, out0, w0, saved0, mine0, val0), (_, out1, w1, saved1, mine1, val1) = got
    assert mine0 == [0, 2, 4] and mine1 == [1, 3] and val0 == [0, 2] and val1 == [1]
    assert np.array_equal(w0[0], w0[1]) and np.array_equal(w0, w1)
    assert saved0 and not saved1 and out1 == ""
    lines = [l for l in out0.splitlines() if l.startswith("Epoch")]
    assert out0.startswith("Training SpectralNet:") and len(lines) == 3

    # one process playing both ranks
    feats, mats, vfeats, vmats = _make_problem()
    torch.manual_seed(50)
    net = _StubNet()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    shares = [[0, 2, 4], [1, 3]]
    rs = [np.random.RandomState(0), np.random.RandomState(0)]
    want_lines = []
    for epoch in range(3):
        total = 0.0
        for step in range(3):
            opt.zero_grad()
            loss_sum = 0
            for r in range(2):
                nb = len(shares[r])
                pid = step % nb
                pid_2 = rs[r].choice(nb)
                b1, b2 = shares[r][pid], shares[r][pid_2]
                rs[r].permutation(len(feats[b1]))
                rs[r].permutation(len(feats[b2]))
                l = _stub_loss(mats[b1], net.forward_pair(feats[b2], feats[b1]))
                total += float(l.detach())
                loss_sum = loss_sum + l / 2
            loss_sum.backward()
            opt.step()
        with torch.no_grad():
            vl = sum(float(_stub_loss(vmats[b], net(vfeats[b]))) for b in range(3))
        want_lines.append((total / 6, vl / 3))
    want = torch.cat([p.detach().flatten() for p in net.parameters()]).numpy()
    assert np.allclose(w0[0], want, rtol=1e-5, atol=1e-6)
    import re
    for line, (tl, vl) in zip(lines, want_lines):
        nums = [float(x) for x in re.findall(r"[-+]?\d+\.\d+", line)]
        assert abs(nums[0] - tl) <= 1e-5 * max(1, abs(tl)) and abs(nums[1] - vl) <= 1e-5 * max(1, abs(vl)), (line, tl, vl)


def test_more_ranks_than_batches_still_gives_every_rank_work():
    from sparc_amd.SpectralTrainer import SpectralTrainer
    assert SpectralTrainer._my_share(2, 1, 4, 3) == ([1], [])
    assert SpectralTrainer._my_share(2, 1, 4, 0) == ([0], [0])
    assert SpectralTrainer._my_share(100, 20, 8, 7) == (list(range(7, 100, 8)), [7, 15])
