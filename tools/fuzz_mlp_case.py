#!/usr/bin/env python3
"""One MLP problem of tests/fuzz_float.py (by seed) looked at closely: every gradient against torch's float32 layers,
with the hidden layers' input gradients on the fused GEMMs and on the library path.
    python tools/fuzz_mlp_case.py SEED [SEED ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.ops import SpectralMLP   # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    nl = int(rng.integers(1, 5))
    pick = lambda: int(rng.choice([1, 2, 3, 5, 8, 10, 16, 31, 32, 33, 47, 48, 63, 64, 65, 100, 128, 200, 256, 500, 512, 1024, 1100]))
    dims = [pick() for _ in range(nl + 1)]
    m = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 1537, 4097]))
    x_grad = bool(rng.integers(0, 2))
    if seed % 4 == 0:
        dims = [int(rng.choice([1, 3, 10, 13, 16])), int(rng.choice([128, 256, 1024])), int(rng.choice([16, 64, 128, 512])),
                int(rng.choice([5, 47, 64, 128]))] + ([pick()] if rng.integers(0, 2) else [])
        x_grad = False
    lins = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(m, dims[0], device=dev, requires_grad=x_grad)
    t = torch.randn(m, dims[-1], device=dev)
    h = x
    near = []
    for lin in lins[:-1]:
        pre = lin(h)
        near.append(int((pre.abs() < 1e-6 * pre.abs().max()).sum()))
        h = torch.relu(pre)
    ref = torch.tanh(lins[-1](h))
    params = [p for lin in lins for p in (lin.weight, lin.bias)]
    rg = torch.autograd.grad((ref * t).sum(), params)
    print(f"seed {seed}: dims {dims} m {m}, pre-activations within 1e-6 of zero per hidden layer: {near}")
    res = {}
    for fused in (True, False):
        SpectralMLP.fuse_wide_dgrad = fused
        out = SpectralMLP.apply(x, None, *params)
        og = torch.autograd.grad((out * t).sum(), params)
        res[fused] = og
        print("  fused" if fused else "  library", [f"{rel(a, b):.2e}" for a, b in zip(og, rg)])
    print("  fused against library", [f"{rel(a, b):.2e}" for a, b in zip(res[True], res[False])])
    # the first layer's ReLU units on which the own forward kernel and torch's layer disagree
    from sparc_amd.ops import ACT_RELU, linear_narrow_in_fwd
    if dims[0] <= 16:
        own = linear_narrow_in_fwd(x.detach(), lins[0].weight.detach(), lins[0].bias.detach(), ACT_RELU)
        theirs = torch.relu(lins[0](x.detach()))
        print("  first-layer units with different ReLU decisions:", int(((own > 0) != (theirs > 0)).sum()))
