#!/usr/bin/env python3
"""The wide GEMMs of one training step (both batches of a step in one pass, M rows) under torch's
TunableOp: `tune` searches the rocBLAS / hipBLASLt solutions for every shape and writes the result
file; `run` executes every GEMM REPS times with the default picks (USE_TUNED=0) or the tuned ones
(USE_TUNED=1) -- run it under `rocprofv3 --kernel-trace --stats` and compare per-kernel durations.
fp32 in, fp32 accumulate, fp32 out either way: tuning only selects among kernels of the same arithmetic.

    python tools/tune_gemms.py tune [M] [file]
    USE_TUNED=1 python tools/tune_gemms.py run [M] [file]
"""
import os
import sys
import time

import torch
import torch.cuda.tunable as tun

mode = sys.argv[1] if len(sys.argv) > 1 else "run"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 48979
path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/tunableop_results.csv"
DIN = int(os.environ.get("DIN", "10"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def cases():
    out = {}
    x0 = torch.randn(M, DIN, device=dev)
    W0 = torch.randn(1024, DIN, device=dev)
    b0 = torch.randn(1024, device=dev)
    out[f"fwd {DIN}->1024"] = lambda: torch._addmm_activation(b0, x0, W0.t())
    for (K, N) in [(1024, 1024), (1024, 512)]:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        g = torch.randn(M, N, device=dev)
        out[f"fwd {K}->{N}"] = (lambda x=x, W=W, b=b: torch._addmm_activation(b, x, W.t()))
        out[f"dgrad {K}->{N}"] = (lambda g=g, W=W: g.mm(W))
        out[f"wgrad {K}->{N}"] = (lambda g=g, x=x: g.t().mm(x))
    return out


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cs = cases()
if mode == "tune":
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "30")))
    tun.set_max_tuning_iterations(int(os.environ.get("TUNE_IT", "10")))
    tun.set_filename(path)
    t0 = time.perf_counter()
    for k, f in cs.items():
        t1 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        print(f"tuned {k} in {time.perf_counter() - t1:.1f}s", flush=True)
    for r in tun.get_results():
        print("   ", r, flush=True)
    print(f"(the result file is written when the process exits) tuning took {time.perf_counter() - t0:.1f}s -> {path}", flush=True)
else:
    if os.environ.get("USE_TUNED", "0") == "1":
        tun.enable(True)
        tun.tuning_enable(False)
        tun.set_filename(path)
        ok = tun.read_file(path)
        print("tuned picks loaded:", ok, flush=True)
    reps = int(os.environ.get("REPS", "20"))
    res = {k: round(timeit(f, reps), 1) for k, f in cs.items()}
    print(f"M={M}", "tuned" if os.environ.get("USE_TUNED", "0") == "1" else "default", res, "sum", round(sum(res.values()), 1), flush=True)
