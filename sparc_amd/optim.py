"""torch.optim.Adam as the reference configures it (src/SpectralTrainer.py:188: defaults apart from the learning rate)
with the update of all parameter tensors in one launch of libsparc_hip.so (sparc_adam_step).  A torch Optimizer:
param_groups, state_dict and learning-rate schedulers (src/SpectralTrainer.py:189-192) work as with torch's own."""
from __future__ import annotations

import ctypes

import torch

from ._lib import AdamTensor, check, lib

MAX_TENSORS = 32


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            group["step"] = group.get("step", 0) + 1
            recs = []
            for p in live:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or p.grad.is_sparse:
                    raise RuntimeError("sparc_amd.optim.Adam updates dense float32 parameters on an MI355X")
                if not p.is_contiguous():
                    raise RuntimeError("sparc_amd.optim.Adam needs contiguous parameters")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                recs.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            dev = live[0].device
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            with torch.cuda.device(dev):
                for a in range(0, len(recs), MAX_TENSORS):
                    part = recs[a:a + MAX_TENSORS]
                    arr = (AdamTensor * len(part))(*[AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                                                     for p, g, m, v in part])
                    check(lib.sparc_adam_step(len(part), arr, float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                              float(group["eps"]), float(group["weight_decay"]), int(group["step"]), stream),
                          "sparc_adam_step")
        return loss
