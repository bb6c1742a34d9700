"""Probe, not product code: how long the host takes to enqueue the library GEMMs of one MLP step at the bench's sizes
(does each call return before the product has run?)."""
import time
import torch

dev = torch.device("cuda:0")
M = 48980
x0 = torch.randn(M, 10, device=dev)
h1 = torch.randn(M, 1024, device=dev)
h2 = torch.randn(M, 1024, device=dev)
g3 = torch.randn(M, 512, device=dev)
W1 = torch.randn(1024, 10, device=dev)
W2 = torch.randn(1024, 1024, device=dev)
W3 = torch.randn(512, 1024, device=dev)
b1 = torch.randn(1024, device=dev)
b3 = torch.randn(512, device=dev)
cases = (("fwd layer 1 (K=10)", lambda: torch._addmm_activation(b1, x0, W1.t())),
         ("fwd layer 2", lambda: torch._addmm_activation(b1, h1, W2.t())),
         ("fwd layer 3", lambda: torch._addmm_activation(b3, h2, W3.t())),
         ("wgrad 3", lambda: g3.t().mm(h2)),
         ("dgrad 3", lambda: g3.mm(W3)),
         ("wgrad 2", lambda: h1.t().mm(h2)),
         ("dgrad 2", lambda: h1.mm(W2)),
         ("cat", lambda: torch.cat([h1, h2], 0)),
         ("empty", lambda: torch.empty(M, 1024, device=dev)))
for name, f in cases:
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:20s}: host {1e6 * (t1 - t0) / n:8.1f} us/call, device {1e6 * (t2 - t0) / n:8.1f} us/call")
