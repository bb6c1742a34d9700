"""Where the host spends a training step (no device sync inside the loop): per-phase wall time of the enqueueing
thread at a named batch shape.  A phase that takes longer than its launches explain is waiting for the device.
    python tools/host_phase_probe.py [amazon2m|reddit|pubmed|cora]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth                                   # noqa: E402
from sparc_amd.parallel import ShardedSpectralTrainer         # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "amazon2m"
dev = torch.device("cuda:0")
b, k = synth.shape_batch(workload, dev)
shape = synth.SHAPES[workload]
din = shape["din"]
arch = {"n_layers": 6, "hidden_dim1": 1024, "hidden_dim2": 1024, "hidden_dim3": 512, "output_dim": k}
tr = ShardedSpectralTrainer(arch, din, dev, lr=1e-3, seed=0)
b.warm(k, measure=True)
tr.add_batch(torch.randn(b.m, din, device=dev), b)
rng = np.random.RandomState(0)
names = ("zero_grad", "forward_pair", "loss", "backward", "adam")
acc = np.zeros(len(names))
steps, warm = 60, 10
torch.cuda.synchronize()
for s in range(steps + warm):
    if s == warm:
        torch.cuda.synchronize()
        acc[:] = 0
        t_all = time.perf_counter()
    head = rng.permutation(b.m)[:k]
    t = [time.perf_counter()]
    tr.net.train()
    tr.optimizer.zero_grad(set_to_none=True)
    t.append(time.perf_counter())
    Y = tr.net.forward_pair(tr.features[0], tr.features[0], top_idx=head)
    t.append(time.perf_counter())
    loss = tr.criterion(b, Y, is_normalized=True)
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    tr.optimizer.step()
    t.append(time.perf_counter())
    acc += np.diff(t)
t_host = time.perf_counter() - t_all
torch.cuda.synchronize()
t_dev = time.perf_counter() - t_all
print(f"{workload}: m = {b.m}, host {1e3 * t_host / steps:.2f} ms/step, device {1e3 * t_dev / steps:.2f} ms/step")
for n, a in zip(names, acc):
    print(f"  {n:14s} {1e3 * a / steps:7.3f} ms")
