#!/usr/bin/env python3
"""Timing of the Gram product of the orthonormalisation layer (gram_partial + gram_reduce) at the batch sizes of
the named shapes; run under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.ops import gram
dev = torch.device("cuda:0")
for m, k in [(24490, 47), (24490, 64), (7766, 64), (98000, 47)]:
    Y = torch.randn(m, k, device=dev)
    for _ in range(5):
        gram(Y, Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gram(Y, Y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"m={m} k={k}: {us:.1f} us per product, {2.0 * m * k * k / us / 1e6:.2f} TFLOP/s f64, {2.0 * m * k * 4 / us / 1e3:.0f} GB/s", flush=True)
