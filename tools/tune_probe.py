"""Does GEMM tuning (torch TunableOp over rocBLAS/hipBLASLt solutions) beat the default heuristic on the
fixed GEMM shapes of a padded training step?"""
import os
import time

import torch
import torch.cuda.tunable as tun

dev = torch.device("cuda", 0)
M = 49408
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cases = {}
for (K, N) in [(1024, 1024), (1024, 512)]:
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    g = torch.randn(M, N, device=dev)
    cases[f"fwd {K}->{N}"] = (lambda x=x, W=W, b=b: torch._addmm_activation(b, x, W.t()))
    cases[f"dgrad {K}->{N}"] = (lambda g=g, W=W: g.mm(W))
    cases[f"wgrad {K}->{N}"] = (lambda g=g, x=x: g.t().mm(x))

base = {k: timeit(f) for k, f in cases.items()}
print("default:", {k: round(v, 1) for k, v in base.items()}, flush=True)
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "30")))
tun.set_max_tuning_iterations(int(os.environ.get("TUNE_IT", "10")))
tun.set_filename("gpurun_out/tunableop_probe.csv")
t0 = time.perf_counter()
for k, f in cases.items():
    t1 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    print(f"tuned {k} in {time.perf_counter() - t1:.1f}s", flush=True)
print(f"tuning took {time.perf_counter() - t0:.1f}s", flush=True)
tun.tuning_enable(False)
tuned = {k: timeit(f) for k, f in cases.items()}
print("tuned:  ", {k: round(v, 1) for k, v in tuned.items()}, flush=True)
print("sum default %.1f us, tuned %.1f us" % (sum(base.values()), sum(tuned.values())))

