#!/usr/bin/env python3
"""Time of sparc_orth_weights and sparc_orth_weights_bwd_f64 by spectral dimension and mode (0: with the sign
recovery, 1: Cholesky only), torch events over REPS calls each."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd._lib import check, lib
from sparc_amd.ops import gram

dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
reps = int(os.environ.get("REPS", "200"))
m = 24000
for k in (16, 32, 47, 64):
    Y = torch.randn(m, k, device=dev)
    G = gram(Y, Y)
    Wo = torch.empty(k, k, device=dev)
    Rc, Ri, H, g64 = (torch.empty(k, k, dtype=torch.float64, device=dev) for _ in range(4))
    g64.normal_()
    sg = torch.empty(k, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode in (0, 1):
        def fwd():
            check(lib.sparc_orth_weights(m, k, P(G), P(Y), k, None, mode, P(Wo), P(Rc), P(Ri), P(sg), P(info), st), "fwd")
        def bwd():
            check(lib.sparc_orth_weights_bwd_f64(m, k, P(g64), P(Rc), P(Ri), P(sg), mode, P(H), st), "bwd")
        out = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / reps * 1e3
        print(f"k {k} mode {mode}: orth_weights {out['fwd']:.1f} us, orth_weights_bwd_f64 {out['bwd']:.1f} us, info {int(info)}", flush=True)
