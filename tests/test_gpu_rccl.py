"""The data-parallel step over RCCL on the one GPU a test box has: a world of ONE rank with the `nccl`
backend runs the layer-wise asynchronous all-reduce of ShardedSpectralTrainer.step for real (RCCL launches
on its own stream, the buckets are handed to autograd as views, finish() waits and averages), and must
reproduce the single-process trainer: the mean over one rank is the rank's own gradient.  The multi-rank
arithmetic itself is covered on CPU by tests/test_parallel_gloo.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SPARC_ROOT"])
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch
from sparc_amd.parallel import LayerwiseReducer, ShardedSpectralTrainer

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
k, din = 12, 20
arch = {"n_layers": 6, "hidden_dim1": 64, "hidden_dim2": 64, "hidden_dim3": 32, "output_dim": k}
graph, bounds = synth.planted_cluster_graph(3000, 12000, 12, dev, seed=3)
nodes_of, _ = synth.cluster_batches(bounds, 6, seed=0, return_clusters=True)
runs = {}
for mode in ("single", "rccl"):
    tr = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=5)
    assert not tr.distributed                      # a world of one: the trainer would skip the collective
    if mode == "rccl":                             # ... so it is switched on by hand
        tr.distributed = True
        red = LayerwiseReducer(tr.params)
        seen = []
        inner = red.on_layer
        def spy(layer, dW, db, inner=inner, seen=seen):
            seen.append(layer)
            return inner(layer, dW, db)
        red.on_layer = spy
        tr.attach_reducer(red)
    for b in range(2):
        coords, values, shp, _ = graph.subblock(nodes_of[b], mode=2, binarize=False, diag_lambda=0.01)
        tr.add_batch(synth.block_features(nodes_of[b], bounds, din, 12), ClusterBatch.from_device_coo(shp[0], coords, values).warm(k))
    rng = np.random.RandomState(0)
    losses = []
    for s in range(6):
        pid, pid_2 = s % 2, (s // 2) % 2
        losses.append(float(tr.step(pid, pid_2, rng.permutation(tr.batches[pid_2].m)[:k])))
    runs[mode] = (np.array(losses), torch.cat([p.detach().flatten() for p in tr.params]).cpu().numpy())
    if mode == "rccl":
        assert len(seen) == 6 * 4, seen      # four Linear layers reported per step
        for l in range(4):                   # every gradient Adam read was a view into its layer's bucket
            lo, hi = red.buckets[l].data_ptr(), red.buckets[l].data_ptr() + red.buckets[l].numel() * 4
            assert lo <= tr.params[2 * l].grad.data_ptr() < hi and lo <= tr.params[2 * l + 1].grad.data_ptr() < hi
        val = float(tr.validation_loss([tr.features[0]], [tr.batches[0]]))  # the scalar all-reduce of the validation loss
        assert np.isfinite(val)
dist.barrier()
dist.destroy_process_group()
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
assert rel(runs["rccl"][0], runs["single"][0]) < 1e-6, (runs["rccl"][0], runs["single"][0])
assert rel(runs["rccl"][1], runs["single"][1]) < 1e-6
print("RCCL_STEP_OK")
"""


def test_sharded_step_over_rccl_world_of_one():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPARC_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "RCCL_STEP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


WORKER2 = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SPARC_ROOT"])
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch
from sparc_amd.parallel import ShardedSpectralTrainer

rank = int(os.environ["RANK"])
dev = torch.device("cuda:0")                       # both ranks share the one GPU of the test box
torch.cuda.set_device(dev)
k, din = 12, 20
arch = {"n_layers": 6, "hidden_dim1": 64, "hidden_dim2": 64, "hidden_dim3": 32, "output_dim": k}
graph, bounds = synth.planted_cluster_graph(3000, 12000, 12, dev, seed=3)
nodes_of, _ = synth.cluster_batches(bounds, 6, seed=0, return_clusters=True)

def run(distributed, same_batches):
    tr = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=5)
    assert tr.distributed == distributed
    for b in range(2):
        nb = b if same_batches else (2 * b + rank) % len(nodes_of)
        coords, values, shp, _ = graph.subblock(nodes_of[nb], mode=2, binarize=False, diag_lambda=0.01)
        tr.add_batch(synth.block_features(nodes_of[nb], bounds, din, 12), ClusterBatch.from_device_coo(shp[0], coords, values).warm(k))
    rng = np.random.RandomState(0)
    losses = []
    for s in range(6):
        pid, pid_2 = s % 2, (s // 2) % 2
        losses.append(float(tr.step(pid, pid_2, rng.permutation(tr.batches[pid_2].m)[:k])))
    return np.array(losses), torch.cat([p.detach().flatten() for p in tr.params])

def one_backward(distributed):
    # the gradients one step leaves in .grad (before Adam), on this rank's own batch
    tr = ShardedSpectralTrainer(arch, din, dev, lr=0.0, seed=5)
    assert tr.distributed == distributed
    coords, values, shp, _ = graph.subblock(nodes_of[rank], mode=2, binarize=False, diag_lambda=0.01)
    tr.add_batch(synth.block_features(nodes_of[rank], bounds, din, 12), ClusterBatch.from_device_coo(shp[0], coords, values).warm(k))
    tr.step(0, 0, np.arange(k))
    return torch.cat([p.grad.detach().flatten() for p in tr.params]).clone(), tr

single = run(False, True)                          # before the process group exists: the single-process trainer
local_grad, _ = one_backward(False)
dist.init_process_group("gloo", rank=rank, world_size=2)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
# the gradients themselves (not the weights after Adam, which hardly notice a factor of two): what the layer-wise
# reducer leaves in .grad is the mean of the two ranks' local gradients, and lives in the layers' buckets
mean_grad, tr_d = one_backward(True)
both_local = [torch.empty_like(local_grad) for _ in range(2)]
dist.all_gather(both_local, local_grad)
want = (both_local[0].double() + both_local[1].double()) / 2
assert not torch.equal(both_local[0], both_local[1])
assert rel(mean_grad.double().cpu().numpy(), want.cpu().numpy()) < 1e-6
assert len(tr_d.reducer.buckets) == 4
for l in range(4):
    lo, hi = tr_d.reducer.buckets[l].data_ptr(), tr_d.reducer.buckets[l].data_ptr() + tr_d.reducer.buckets[l].numel() * 4
    assert lo <= tr_d.params[2 * l].grad.data_ptr() < hi and lo <= tr_d.params[2 * l + 1].grad.data_ptr() < hi
# both ranks hold the same batches: the mean of two equal gradients is the gradient, so the trajectory is the
# single-process one (the all-reduce, the bucket views and the averaging all take part)
same = run(True, True)
assert rel(same[0], single[0]) < 1e-6, (same[0], single[0])
assert rel(same[1].cpu().numpy(), single[1].cpu().numpy()) < 1e-6
# different batches per rank: the replicas stay identical bit for bit, and the step still descends
diff = run(True, False)
both = [torch.empty_like(diff[1]) for _ in range(2)]
dist.all_gather(both, diff[1])
assert torch.equal(both[0], both[1])
assert np.isfinite(diff[0]).all()
dist.barrier()
dist.destroy_process_group()
print("GLOO2_STEP_OK", rank)
"""


def test_sharded_step_two_ranks_on_one_gpu():
    """Two ranks (gloo) drive the real kernels on the one GPU: with equal batches on both ranks the averaged step is the
    single-process step; with different batches the replicas stay bitwise identical."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPARC_ROOT=ROOT, RANK=str(r), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER2], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"GLOO2_STEP_OK {r}" in o, o[-2000:] + e[-4000:]


WORKER3 = r"""
import contextlib, io, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SPARC_ROOT"])
from sparc_amd import synth
from sparc_amd.SpectralTrainer import SpectralTrainer

rank = int(os.environ["RANK"])
dev = torch.device("cuda:0")                       # both ranks share the one GPU of the test box
torch.cuda.set_device(dev)
os.chdir(os.environ["SPARC_TMP"])
k, din = 12, 20
graph, bounds = synth.planted_cluster_graph(4000, 16000, 20, dev, seed=3)
nodes_of, cluster_ids = synth.cluster_batches(bounds, 4, seed=0, return_clusters=True)   # five training batches
sups, feats = [], []
for b in range(len(nodes_of)):
    coords, values, shp, _ = graph.subblock(nodes_of[b], mode=2, binarize=False, diag_lambda=0.01)
    sups.append((coords.cpu().numpy(), values.cpu().numpy(), shp))
    feats.append(synth.block_features(nodes_of[b], bounds, din, 12))
train = ([np.arange(f.shape[0]) for f in feats], feats, sups, [None] * 5, [None] * 5)
val = (list(range(3)), feats[:3], sups[:3], [None] * 3, [None] * 3)
cfg = {"dataset": f"dp_rank{rank}", "spectral": {
    "architecture": {"n_layers": 4, "hidden_dim1": 64, "hidden_dim2": 64, "hidden_dim3": 32, "output_dim": k},
    "epochs": 3, "lr": 1e-3, "lr_decay": 0.1, "min_lr": 1e-7, "batch_size": 1024, "n_clusters": k, "patience": 10}}

def run():
    torch.manual_seed(7 + (rank if dist.is_initialized() else 0))      # rank 0's initial weights must win
    np.random.seed(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr = SpectralTrainer(cfg, dev)
        net = tr.train(train, val)
    return tr, net, buf.getvalue()

dist.init_process_group("gloo", rank=rank, world_size=2)
tr, net, out = run()
w = torch.cat([p.detach().flatten() for p in net.parameters()])
both = [torch.empty_like(w) for _ in range(2)]
dist.all_gather(both, w)
assert torch.equal(both[0], both[1])               # the replicas end bitwise equal
orth = [torch.empty_like(net.orthonorm_weights) for _ in range(2)]
lines = [l for l in out.splitlines() if l.startswith("Epoch")]
assert (len(lines) == 3 and out.startswith("Training SpectralNet:")) if rank == 0 else out == "", out
assert os.path.exists(tr.weights_path) == (rank == 0) and os.path.exists(tr.orth_weights_path) == (rank == 0)
assert net.grad_ready_hook is not None
import re
nums = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in lines]
assert all(np.isfinite(n).all() for n in nums)
if rank == 0:
    assert nums[-1][0] < nums[0][0], lines         # the averaged steps descend
    # the checkpoint restores a model that predicts: MLP weights and the orthonormalisation weights beside them
    tr2 = SpectralTrainer(cfg, dev)
    net2 = tr2.load_model(din)
    assert net2.orthonorm_weights.shape == (k, k) and torch.isfinite(net2.orthonorm_weights).all()
    with torch.no_grad():
        net2.eval()
        Y = net2(feats[0].to(dev), should_update_orth_weights=False)
    assert Y.shape == (feats[0].shape[0], k) and torch.isfinite(Y).all()
dist.barrier()
dist.destroy_process_group()
print("DP_TRAIN_OK", rank)
"""


def test_spectral_trainer_train_two_ranks_on_one_gpu(tmp_path):
    """SpectralTrainer.train inside a process group of two gloo ranks that share the one GPU, through the real kernels:
    five training batches dealt 3 + 2, three validation batches 2 + 1; the replicas end bitwise equal, the loss falls,
    rank 0 alone prints and writes the checkpoint, which reloads into a model that predicts."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPARC_ROOT=ROOT, RANK=str(r), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", SPARC_TMP=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER3], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DP_TRAIN_OK {r}" in o, o[-2000:] + e[-4000:]


def test_bench_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks (here both on the one GPU,
    over gloo), they run a shrunken Amazon2M-shape workload and rank 0 prints the JSON line for two GPUs."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPARC_BENCH_BACKEND="gloo", SPARC_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.04", "--steps", "4",
                          "--warmup", "2", "--batches-per-rank", "2", "--no-cpu-baseline"], env=env, capture_output=True,
                         text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-12000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert "rank 0 of 2 (gloo)" in out.stderr and "rank 1 of 2 (gloo)" in out.stderr
