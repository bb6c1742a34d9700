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
