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


ACT_RELU = 0
ACT_TANH = 1
ACT_NONE = 2  # a Linear layer without activation (autoencoder bottleneck / output)


def act_bwd_bias(g: torch.Tensor, h: torch.Tensor, act: int):
    """(g * act'(h), column sums of it): the gradient entering a Linear layer and its bias gradient."""
    m, n = h.shape
    gout = torch.empty(m, n, dtype=torch.float32, device=h.device)
    db = torch.empty(n, dtype=torch.float32, device=h.device)
    nbytes = lib.sparc_act_bwd_bias_workspace_bytes(m, n)
    ws = Workspace.get(nbytes, h.device)
    with torch.cuda.device(h.device):
        check(lib.sparc_act_bwd_bias(m, n, int(act), _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(gout), n, _ptr(db),
                                     _ptr(ws), nbytes, _stream()), "sparc_act_bwd_bias")
    return gout, db


def wgrad_small(g: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """g^T h for a narrow g [m, n_out <= 64] and h [m, n_in], n_in a multiple of 64."""
    m, n_out = g.shape
    n_in = h.shape[1]
    dW = torch.empty(n_out, n_in, dtype=torch.float32, device=h.device)
    nbytes = lib.sparc_wgrad_small_workspace_bytes(n_out, n_in)
    ws = Workspace.get(nbytes, h.device)
    with torch.cuda.device(h.device):
        check(lib.sparc_wgrad_small(m, n_out, n_in, _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(dW), _ptr(ws), nbytes,
                                    _stream()), "sparc_wgrad_small")
    return dW


def first_layer_bwd(x: torch.Tensor, g: torch.Tensor, h: torch.Tensor):
    """(dW [n, din] as a transposed view, db [n]) of a Linear+ReLU layer whose input needs no gradient,
    in one pass over g and h: the masked gradient is never written out."""
    m, din = x.shape
    n = h.shape[1]
    out = torch.empty(din + 1, n, dtype=torch.float32, device=h.device)
    nbytes = lib.sparc_first_layer_bwd_workspace_bytes(din, n)
    ws = Workspace.get(nbytes, h.device)
    with torch.cuda.device(h.device):
        check(lib.sparc_first_layer_bwd(m, din, n, _ptr(x), x.stride(0), _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(out),
                                        _ptr(ws), nbytes, _stream()), "sparc_first_layer_bwd")
    return out[:din].t(), out[din]


class LinearAct(torch.autograd.Function):
    """act(x W^T + b) of one MLP layer (src/SpectralTrainer.py:28-39): the products are library
    GEMMs (bias and ReLU in the forward product's epilogue); the activation derivative, the bias
    gradient and the narrow output layer's weight gradient run in libsparc_hip.so."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        require_gpu(x.device)
        if act == ACT_RELU:
            h = torch._addmm_activation(b, x, W.t())
        elif act == ACT_TANH:
            h = torch.addmm(b, x, W.t()).tanh_()
        else:
            h = torch.addmm(b, x, W.t())
        ctx.save_for_backward(x, W, h if act != ACT_NONE else None)
        ctx.act = act
        return h

    @staticmethod
    def backward(ctx, g):
        x, W, h = ctx.saved_tensors
        if (ctx.act == ACT_RELU and not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and x.shape[1] < 64
                and W.shape[0] % 64 == 0 and x.dtype == torch.float32 and x.stride(1) == 1):
            g = _f32c(g)
            if g.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0:
                dW, db = first_layer_bwd(x, g, h)
                return None, dW, (db if ctx.needs_input_grad[2] else None), None
        if ctx.act == ACT_NONE:
            gp = _f32c(g)
            db = gp.sum(0) if ctx.needs_input_grad[2] else None
        else:
            gp, db = act_bwd_bias(_f32c(g), h, ctx.act)
        dx = gp.mm(W) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            n_out, n_in = W.shape
            if n_out <= 64 and n_in % 64 == 0 and x.is_contiguous() and x.data_ptr() % 16 == 0:
                dW = wgrad_small(gp, x)           # narrow output layer
            elif n_in <= 64 and n_out % 64 == 0:
                dW = wgrad_small(x, gp).t()       # narrow input layer: (x^T gp)^T
            else:
                dW = gp.t().mm(x)
        return dx, dW, (db if ctx.needs_input_grad[2] else None), None


class SplitRows(torch.autograd.Function):
    """(Y[:m2], Y[m2:]) as views; the backward concatenates the two gradients with one copy
    instead of two zero-filled buffers and an add."""

    @staticmethod
    def forward(ctx, Y, m2):
        ctx.m2 = m2
        ctx.rows = Y.shape[0]
        return Y[:m2], Y[m2:]

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None, None
        ref = g1 if g1 is not None else g2
        if g1 is None:
            g1 = ref.new_zeros(ctx.m2, ref.shape[1])
        if g2 is None:
            g2 = ref.new_zeros(ctx.rows - ctx.m2, ref.shape[1])
        return torch.cat([g1, g2], dim=0), None


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
