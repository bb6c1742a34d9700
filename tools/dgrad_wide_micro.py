#!/usr/bin/env python3
"""The hidden layer's input gradient with the previous ReLU's derivative: library GEMM + sparc_act_bwd_bias against
sparc_dgrad_wide_relu, at the MLP's shapes (M rows, K -> N).  Run under rocprofv3 --kernel-trace --stats for kernel times.
    python tools/dgrad_wide_micro.py [M] [K] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.ops import ACT_RELU, act_bwd_bias, dgrad_wide_relu   # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 48980
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = torch.randn(M, K, device=dev)
W = torch.randn(K, N, device=dev) / K ** 0.5
h = torch.relu(torch.randn(M, N, device=dev))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def two_steps():
    return act_bwd_bias(g.mm(W), h, ACT_RELU)


a, da = two_steps()
b, db = dgrad_wide_relu(g, W, h)
print("max |difference| of the results:", float((a - b).abs().max()), float((da - db).abs().max()))
t2, t1 = timed(two_steps), timed(lambda: dgrad_wide_relu(g, W, h))
flop = 2.0 * M * K * N
print(f"M={M} K={K} N={N}: library GEMM + act_bwd_bias {t2:.1f} us, fused {t1:.1f} us ({flop / t1 / 1e6:.1f} TFLOP/s)")
