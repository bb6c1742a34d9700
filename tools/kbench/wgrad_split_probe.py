"""Probe, not product code: the weight gradient of the 1024 -> 512 layer (a 512 x 1024 result summed over 49 k rows) as
one library GEMM and as a batched GEMM over parts of the rows whose partial results are added."""
import torch

dev = torch.device("cuda:0")
M = 48980
g = torch.randn(M, 512, device=dev)
x = torch.randn(M, 1024, device=dev)


def timed(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = g.double().t() @ x.double()
one = lambda: g.t().mm(x)
print(f"one GEMM              {timed(one):7.1f} us  err {(one().double() - ref).abs().max().item() / ref.abs().max().item():.2e}")
for parts in (2, 3, 4, 5, 6, 8):
    rows = M // parts * parts
    g3 = g[:rows].view(parts, rows // parts, 512)
    x3 = x[:rows].view(parts, rows // parts, 1024)

    def split():
        out = torch.bmm(g3.transpose(1, 2), x3).sum(0)
        if rows < M:
            out.addmm_(g[rows:].t(), x[rows:])
        return out
    print(f"bmm over {parts} parts    {timed(split):7.1f} us  err {(split().double() - ref).abs().max().item() / ref.abs().max().item():.2e}")
# the other orientation: (x^T g)^T
other = lambda: x.t().mm(g)
print(f"x^T g                 {timed(other):7.1f} us")
g2 = torch.randn(M, 1024, device=dev)
print(f"1024 x 1024 result    {timed(lambda: g2.t().mm(x)):7.1f} us")

ref2 = g2.double().t() @ x.double()
for parts in (2, 4):
    rows = M // parts * parts
    a3 = g2[:rows].view(parts, rows // parts, 1024)
    x3 = x[:rows].view(parts, rows // parts, 1024)
    f = lambda: torch.bmm(a3.transpose(1, 2), x3).sum(0)
    print(f"1024 x 1024 over {parts} parts {timed(f):7.1f} us  err {(f().double() - ref2).abs().max().item() / ref2.abs().max().item():.2e}")
Modd = 48979
for parts in (4,):
    rows = Modd // parts * parts
    g3 = g[:rows].view(parts, rows // parts, 512)
    x3 = x[:rows].view(parts, rows // parts, 1024)

    def split():
        out = torch.bmm(g3.transpose(1, 2), x3).sum(0)
        out.addmm_(g[rows:Modd].t(), x[rows:Modd])
        return out
    print(f"48979 rows: bmm over 4 parts + the last {Modd - rows} rows {timed(split):7.1f} us")
