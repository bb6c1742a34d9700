"""Timing probe for the MLP pieces of the training step (torch ops through hipBLASLt):
separate linear + relu against the fused bias+relu epilogue, and the elementwise backward pieces."""
import time

import torch

dev = torch.device("cuda", 0)
m = 24500
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


for (K, N) in [(10, 1024), (1024, 1024), (1024, 512), (512, 47)]:
    x = torch.randn(m, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = torch.relu(torch.nn.functional.linear(x, W, b))
    fused = torch._addmm_activation(b, x, W.t())
    err = float((ref - fused).abs().max())
    t_sep = timeit(lambda: torch.relu(torch.nn.functional.linear(x, W, b)))
    t_fus = timeit(lambda: torch._addmm_activation(b, x, W.t()))
    t_mm = timeit(lambda: torch.mm(x, W.t()))
    g = torch.randn(m, N, device=dev)
    t_dx = timeit(lambda: torch.mm(g, W))
    t_dw = timeit(lambda: torch.mm(g.t(), x))
    t_thr = timeit(lambda: torch.ops.aten.threshold_backward(g, ref, 0.0))
    t_sum = timeit(lambda: g.sum(0))
    print(f"K={K} N={N}: linear+relu {t_sep:.1f} us, addmm_activation {t_fus:.1f} us (max diff {err:.2e}), mm {t_mm:.1f}, "
          f"dx {t_dx:.1f}, dW {t_dw:.1f}, threshold_bwd {t_thr:.1f}, colsum {t_sum:.1f}", flush=True)

# Adam flavours on the SpectralNet parameter set
shapes = [(1024, 10), (1024,), (1024, 1024), (1024,), (512, 1024), (512,), (47, 512), (47,)]
for fused in (False, True):
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = torch.optim.Adam(ps, lr=1e-3, fused=fused)
    print(f"Adam fused={fused}: {timeit(opt.step):.1f} us", flush=True)

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd.ops import act_bwd_bias, wgrad_small  # noqa: E402

for n, act in [(1024, 0), (512, 0), (47, 1)]:
    g = torch.randn(m, n, device=dev)
    h = torch.relu(torch.randn(m, n, device=dev)) if act == 0 else torch.tanh(torch.randn(m, n, device=dev))
    print(f"act_bwd_bias n={n}: {timeit(lambda: act_bwd_bias(g, h, act)):.1f} us", flush=True)
g = torch.randn(m, 47, device=dev)
h = torch.randn(m, 512, device=dev)
print(f"wgrad_small 47x512: {timeit(lambda: wgrad_small(g, h)):.1f} us", flush=True)

# forward product with the weight stored transposed (NN instead of NT)
for mm_ in (24500, 49000):
    for (K, N) in [(1024, 1024), (1024, 512)]:
        x = torch.randn(mm_, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        Wt = W.t().contiguous()
        b = torch.randn(N, device=dev)
        t_nt = timeit(lambda: torch._addmm_activation(b, x, W.t()))
        t_nn = timeit(lambda: torch._addmm_activation(b, x, Wt))
        g = torch.randn(mm_, N, device=dev)
        t_dx = timeit(lambda: torch.mm(g, W))
        t_dx_t = timeit(lambda: torch.mm(g, Wt.t()))
        t_dw = timeit(lambda: torch.mm(g.t(), x))
        t_dw2 = timeit(lambda: torch.mm(x.t(), g))
        print(f"m={mm_} K={K} N={N}: fwd NT {t_nt:.1f} us, fwd NN {t_nn:.1f} us; dx {t_dx:.1f} / via Wt {t_dx_t:.1f}; "
              f"dW g^T x {t_dw:.1f} / x^T g {t_dw2:.1f}", flush=True)
