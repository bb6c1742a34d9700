"""sparc_gram_static, the entry point that allocates nothing (its scratch is an array inside the library
image): through the C ABI, against sparc_gram on the same inputs, bit for bit, on two calls in a row, the
second with a smaller input (the scratch keeps the first call's partial sums)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _gram_pair(lib, A, B, k):
    """(G from sparc_gram with a torch-allocated workspace, G from sparc_gram_static)."""
    m = A.shape[0]
    dev = A.device
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = lib.sparc_gram_workspace_bytes(m, k)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws = ws[(-ws.data_ptr()) % 256:]
    G_ref = torch.full((k, k), float("nan"), dtype=torch.float64, device=dev)
    G_new = torch.full((k, k), float("nan"), dtype=torch.float64, device=dev)
    assert lib.sparc_gram(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G_ref), _ptr(ws), nbytes, stream) == 0
    before = torch.cuda.memory_allocated(dev)
    rc = lib.sparc_gram_static(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G_new), stream)
    assert rc == 0, rc
    assert torch.cuda.memory_allocated(dev) == before
    torch.cuda.synchronize()
    return G_ref.cpu().numpy(), G_new.cpu().numpy()


def test_gram_static_equals_gram_two_calls_in_a_row():
    from sparc_amd._lib import lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(5)
    with torch.cuda.device(dev):
        for (m, k) in [(24491, 64), (3001, 47), (70, 5)]:       # each call smaller than the one before
            A = torch.randn(m, k, device=dev, generator=gen)
            B = torch.randn(m, k, device=dev, generator=gen)
            ref, got = _gram_pair(lib, A, B, k)
            assert np.isfinite(got).all()
            assert np.array_equal(ref, got), (m, k, np.abs(ref - got).max())   # same kernels, same summation order
            want = A.double().t() @ B.double()
            assert float((torch.tensor(got, device=dev) - want).abs().max() / want.abs().max()) < 1e-13


def test_gram_static_rejects_what_it_cannot_hold():
    from sparc_amd._lib import lib
    dev = torch.device("cuda:0")
    A = torch.zeros(8, 65, device=dev)
    G = torch.zeros(65, 65, dtype=torch.float64, device=dev)
    assert lib.sparc_gram_static(8, 65, _ptr(A), 65, _ptr(A), 65, _ptr(G), None) == -4      # SPARC_ERR_SIZE
    assert lib.sparc_gram_static(0, 8, _ptr(A), 65, _ptr(A), 65, _ptr(G), None) == -1       # SPARC_ERR_ARG


def test_gram_static_under_stream_capture():
    """After sparc_init() on the device the entry point makes no runtime query: captured into a HIP graph and replayed
    on new operand contents it gives sparc_gram's result bit for bit."""
    from sparc_amd._lib import init_device, lib
    dev = torch.device("cuda:0")
    init_device(0)
    m, k = 5000, 47
    gen = torch.Generator(device=dev).manual_seed(9)
    A = torch.randn(m, k, device=dev, generator=gen)
    B = torch.randn(m, k, device=dev, generator=gen)
    G = torch.zeros(k, k, dtype=torch.float64, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        rc = lib.sparc_gram_static(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    for seed in (1, 2):
        A.copy_(torch.randn(m, k, device=dev, generator=gen))
        B.copy_(torch.randn(m, k, device=dev, generator=gen))
        g.replay()
        torch.cuda.synchronize()
        ref, _ = _gram_pair(lib, A, B, k)
        assert np.array_equal(G.cpu().numpy(), ref)
