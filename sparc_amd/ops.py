"""torch.autograd glue around the HIP kernels of the orthonormalisation layer and the
spectral loss.  torch supplies device memory, the stream and the autograd tape only;
every numerical operation here runs in libsparc_hip.so.
"""
from __future__ import annotations

import torch

from ._lib import check, lib
from .graph import ClusterBatch, Workspace, _ptr, _stream, require_gpu

ORTH_REFERENCE = 0  # sqrt(m) S Rc^-1 S, S = signs of the Householder diagonal (src/SpectralTrainer.py:64-68)
ORTH_CHOLESKY = 1   # S = I: the orthonormalisation the paper describes


def _f32c(t):
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def gram(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A^T B in float64 (MFMA f64 kernel), A and B float32 [m, k]."""
    m, k = A.shape
    G = torch.empty(k, k, dtype=torch.float64, device=A.device)
    nbytes = lib.sparc_gram_workspace_bytes(m, k)
    ws = Workspace.get(nbytes, A.device)
    with torch.cuda.device(A.device):
        check(lib.sparc_gram(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G), _ptr(ws), nbytes, _stream()),
              "sparc_gram")
    return G


def orth_apply(Yt: torch.Tensor, Wo: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    m, k = Yt.shape
    Z = torch.empty(m, k, dtype=torch.float32, device=Yt.device)
    with torch.cuda.device(Yt.device):
        check(lib.sparc_orth_apply(m, k, _ptr(Yt), Yt.stride(0), _ptr(Wo), 1 if transpose else 0, _ptr(Z), k, _stream()),
              "sparc_orth_apply")
    return Z


class OrthWeights(torch.autograd.Function):
    """Yt [m, k] -> orthonormalisation weights [k, k]  (src/SpectralTrainer.py:42-70)."""

    last_info = None

    @staticmethod
    def forward(ctx, Yt, top_idx, mode):
        require_gpu(Yt.device)
        Y = _f32c(Yt)
        m, k = Y.shape
        if k > 64:
            raise ValueError("spectral dimension k must be <= 64")
        dev = Y.device
        G = gram(Y, Y)
        Wo = torch.empty(k, k, dtype=torch.float32, device=dev)
        saved = torch.empty(2 * k * k + k, dtype=torch.float64, device=dev)
        Rc, Rinv, signs = saved[:k * k], saved[k * k:2 * k * k], saved[2 * k * k:]
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        tidx = None
        if top_idx is not None:
            tidx = torch.as_tensor(top_idx, device=dev).to(torch.int32).contiguous()
            if tidx.numel() < k:  # batches with fewer than k rows: pad with out-of-range rows (read as zero)
                tidx = torch.cat([tidx, torch.full((k - tidx.numel(),), m, dtype=torch.int32, device=dev)])
        with torch.cuda.device(dev):
            check(lib.sparc_orth_weights(m, k, _ptr(G), _ptr(Y), Y.stride(0), _ptr(tidx), int(mode), _ptr(Wo), _ptr(Rc),
                                         _ptr(Rinv), _ptr(signs), _ptr(info), _stream()), "sparc_orth_weights")
        ctx.save_for_backward(Y, saved)
        ctx.mode = int(mode)
        OrthWeights.last_info = info  # device int; non-zero = Gram matrix not positive definite (checked lazily)
        return Wo

    @staticmethod
    def backward(ctx, gWo):
        Y, saved = ctx.saved_tensors
        m, k = Y.shape
        Rc, Rinv, signs = saved[:k * k], saved[k * k:2 * k * k], saved[2 * k * k:]
        g64 = gWo.detach().to(torch.float64).contiguous()
        H = torch.empty(k, k, dtype=torch.float32, device=Y.device)
        with torch.cuda.device(Y.device):
            check(lib.sparc_orth_weights_bwd(m, k, _ptr(g64), _ptr(Rc), _ptr(Rinv), _ptr(signs), ctx.mode, _ptr(H),
                                             _stream()), "sparc_orth_weights_bwd")
        return orth_apply(Y, H), None, None


class OrthApply(torch.autograd.Function):
    """Y = Yt @ Wo  (src/SpectralTrainer.py:103) with gradients for both operands."""

    @staticmethod
    def forward(ctx, Yt, Wo):
        require_gpu(Yt.device)
        Y, W = _f32c(Yt), _f32c(Wo)
        if W.shape != (Y.shape[1], Y.shape[1]):
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({Y.shape[0]}x{Y.shape[1]} and "
                               f"{W.shape[0]}x{W.shape[1]})")
        ctx.save_for_backward(Y, W)
        return orth_apply(Y, W)

    @staticmethod
    def backward(ctx, gY):
        Y, W = ctx.saved_tensors
        g = _f32c(gY)
        gYt = orth_apply(g, W, transpose=True) if ctx.needs_input_grad[0] else None
        gW = gram(Y, g).to(torch.float32) if ctx.needs_input_grad[1] else None
        return gYt, gW


class LapLoss(torch.autograd.Function):
    """Spectral loss with is_normalized=True on a ClusterBatch (src/SpectralTrainer.py:130-137).

    The gradient with respect to Y falls out of the same pass as the loss (the forward
    and the transposed product share one sweep over the CSR), so backward only scales it.
    """

    @staticmethod
    def forward(ctx, Y, batch: ClusterBatch):
        want = bool(ctx.needs_input_grad[0])
        loss, grad = batch.lap(Y, want)
        if want:
            ctx.save_for_backward(grad)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        return grad * gl, None
