#!/usr/bin/env python3
"""The memory-bound helper kernels of the MLP at the bench shape (M rows = both batches of a step), each launched REPS
times; run under rocprofv3 --kernel-trace --stats to read their durations (tools/run_mlp_micro.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import ops

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "48980"))
reps = int(os.environ.get("REPS", "20"))
torch.manual_seed(0)
x = torch.randn(M, 10, device=dev)
W1 = torch.randn(1024, 10, device=dev) / 3
b1 = torch.randn(1024, device=dev)
g1 = torch.randn(M, 1024, device=dev)
h1 = ops.linear_narrow_in_fwd(x, W1, b1, ops.ACT_RELU)
h3 = torch.relu(torch.randn(M, 512, device=dev))
W4 = torch.randn(47, 512, device=dev) / 22
b4 = torch.randn(47, device=dev)
g4 = torch.randn(M, 47, device=dev)
y = ops.linear_narrow_fwd(h3, W4, b4, ops.ACT_TANH)
h2 = torch.relu(torch.randn(M, 1024, device=dev))
g2 = torch.randn(M, 1024, device=dev)
for _ in range(reps):
    ops.linear_narrow_in_fwd(x, W1, b1, ops.ACT_RELU)
    ops.first_layer_bwd(x, g1, h1, W1, b1)
    ops.linear_narrow_fwd(h3, W4, b4, ops.ACT_TANH)
    ops.wgrad_small(g4, h3)
    ops.dgrad_narrow_relu(g4, W4, h3)
    ops.act_bwd_bias(g2, h2, ops.ACT_RELU)
    ops.act_bwd_bias(g4, y, ops.ACT_TANH)
torch.cuda.synchronize()
print("done", M, reps)
