"""Feasibility probe, not product code: an fp32 GEMM of the MLP's size against the same product written as six bf16
products (x = hi + mid + lo, the terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid stacked along K), library GEMMs
both.  Prints the times and the error of each against a float64 product."""
import time
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, N = 48980, 1024, 1024


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = torch.randn(M, K, device=dev)
b = torch.randn(K, N, device=dev) / 32
ref = (a[:2048].double() @ b.double())
c32 = a @ b
print("fp32 GEMM        %.3f ms  err %.2e" % (timed(lambda: a @ b), ((c32[:2048].double() - ref).abs().max() / ref.abs().max()).item()))
ah, am, al = split3(a)
bh, bm, bl = split3(b)
A6 = torch.cat([ah, ah, am, ah, al, am], 1).contiguous()
B6 = torch.cat([bh, bm, bh, bl, bh, bm], 0).contiguous()
A3 = torch.cat([ah, ah, am], 1).contiguous()
B3 = torch.cat([bh, bm, bh], 0).contiguous()
print("bf16 GEMM K      %.3f ms" % timed(lambda: ah @ bh))
print("bf16 GEMM 3K     %.3f ms" % timed(lambda: A3 @ B3))
print("bf16 GEMM 6K     %.3f ms" % timed(lambda: A6 @ B6))
print("split of a       %.3f ms" % timed(lambda: split3(a)))
try:
    c6 = torch.mm(A6, B6, out_dtype=torch.float32)
    print("bf16x6, fp32 out %.3f ms  err %.2e" % (timed(lambda: torch.mm(A6, B6, out_dtype=torch.float32)),
                                                  ((c6[:2048].double() - ref).abs().max() / ref.abs().max()).item()))
    c3 = torch.mm(A3, B3, out_dtype=torch.float32)
    print("bf16x3, fp32 out %.3f ms  err %.2e" % (timed(lambda: torch.mm(A3, B3, out_dtype=torch.float32)),
                                                  ((c3[:2048].double() - ref).abs().max() / ref.abs().max()).item()))
except Exception as e:   # noqa: BLE001
    print("mm(out_dtype=float32) not available:", repr(e)[:200])
