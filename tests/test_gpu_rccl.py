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
from sparc_amd import ops, synth
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
        tr.reducer = LayerwiseReducer()
        seen = []
        inner = tr.reducer.on_layer
        def spy(layer, dW, db, inner=inner, seen=seen):
            seen.append(layer)
            return inner(layer, dW, db)
        tr.reducer.on_layer = spy
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
        assert len(seen) == 6 * 4 and ops.GRAD_READY_HOOK is None, seen      # four Linear layers reported per step
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

single = run(False, True)                          # before the process group exists: the single-process trainer
dist.init_process_group("gloo", rank=rank, world_size=2)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
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
