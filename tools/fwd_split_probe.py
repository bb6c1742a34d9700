"""Feasibility probe: the MLP's forward products on both batches at once (M rows) against the same products on the two
halves one after the other (what overlapping the orthonormalisation of the first half with the second half's forward
would need)."""
import time
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 48980
h = M // 2


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for K, N in [(1024, 1024), (1024, 512)]:
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    whole = timed(lambda: torch._addmm_activation(b, x, W.t(), out=out))
    halves = timed(lambda: (torch._addmm_activation(b, x[:h], W.t(), out=out[:h]), torch._addmm_activation(b, x[h:], W.t(), out=out[h:])))
    print(f"{K} -> {N}: whole {whole:.1f} us, two halves {halves:.1f} us")

# the backward's products the same way: input gradient g W, weight gradient g^T x (halves: one batched product + a sum)
for K, N in [(1024, 1024)]:
    g = torch.randn(M, N, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    x = torch.randn(M, K, device=dev)
    out = torch.empty(M, K, device=dev)
    whole = timed(lambda: torch.mm(g, W, out=out))
    halves = timed(lambda: (torch.mm(g[:h], W, out=out[:h]), torch.mm(g[h:], W, out=out[h:])))
    print(f"dgrad {N} -> {K}: whole {whole:.1f} us, two halves {halves:.1f} us")
    whole = timed(lambda: g.t().mm(x))
    two = timed(lambda: torch.bmm(g.view(2, h, N).transpose(1, 2), x.view(2, h, K)).sum(0))
    four = timed(lambda: torch.bmm(g[:M // 4 * 4].view(4, M // 4, N).transpose(1, 2), x[:M // 4 * 4].view(4, M // 4, K)).sum(0))
    print(f"wgrad {N} x {K}: whole {whole:.1f} us, two parts {two:.1f} us, four parts {four:.1f} us")
for Mx in (48980, 49152, 24490, 24576, 16384, 12245):
    x = torch.randn(Mx, 1024, device=dev)
    W = torch.randn(1024, 1024, device=dev) / 32
    b = torch.randn(1024, device=dev)
    out = torch.empty(Mx, 1024, device=dev)
    t = timed(lambda: torch._addmm_activation(b, x, W.t(), out=out))
    print(f"forward 1024 -> 1024 at M = {Mx}: {t:.1f} us, {2.0 * Mx * 1024 * 1024 / t / 1e6:.1f} TFLOP/s")
