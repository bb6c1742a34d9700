"""torch.autograd glue around the HIP kernels: the MLP, the orthonormalisation layer and the
spectral loss.  torch supplies device memory, the stream, the autograd tape and the wide
dense products of the MLP (library GEMMs); every other numerical operation here runs in
libsparc_hip.so.
"""
from __future__ import annotations

import numpy as np
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


def gram_apply(A: torch.Tensor, B: torch.Tensor, W: torch.Tensor, transpose: bool, out: torch.Tensor) -> torch.Tensor:
    """A^T B in float64, and out = B W (or B W^T) in float32, in one launch (the two halves of the backward of Yt Wo)."""
    m, k = A.shape
    G = torch.empty(k, k, dtype=torch.float64, device=A.device)
    nbytes = lib.sparc_gram_workspace_bytes(m, k)
    ws = Workspace.get(nbytes, A.device)
    with torch.cuda.device(A.device):
        check(lib.sparc_gram_apply(m, k, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(G), _ptr(W), 1 if transpose else 0,
                                   _ptr(out), out.stride(0), _ptr(ws), nbytes, _stream()), "sparc_gram_apply")
    return G


def orth_apply(Yt: torch.Tensor, Wo: torch.Tensor, transpose: bool = False, out: torch.Tensor = None) -> torch.Tensor:
    """Yt Wo (or Yt Wo^T); out: float32 rows of stride out.stride(0) to write into (default: a new [m, k] tensor)."""
    m, k = Yt.shape
    Z = torch.empty(m, k, dtype=torch.float32, device=Yt.device) if out is None else out
    with torch.cuda.device(Yt.device):
        check(lib.sparc_orth_apply(m, k, _ptr(Yt), Yt.stride(0), _ptr(Wo), 1 if transpose else 0, _ptr(Z), Z.stride(0), _stream()),
              "sparc_orth_apply")
    return Z


ACT_RELU = 0
ACT_TANH = 1
ACT_NONE = 2  # a Linear layer without activation (autoencoder bottleneck / output)


def _usable(out, shape):
    """out if it is a float32 contiguous tensor of that shape to write a result into, else None."""
    ok = out is not None and tuple(out.shape) == tuple(shape) and out.dtype == torch.float32 and out.is_contiguous()
    return out if ok else None


def act_bwd_bias(g: torch.Tensor, h: torch.Tensor, act: int, db_out: torch.Tensor = None):
    """(g * act'(h), column sums of it): the gradient entering a Linear layer and its bias gradient (written into
    db_out when one is given)."""
    m, n = h.shape
    gout = torch.empty(m, n, dtype=torch.float32, device=h.device)
    db = _usable(db_out, (n,))
    if db is None:
        db = torch.empty(n, dtype=torch.float32, device=h.device)
    nbytes = lib.sparc_act_bwd_bias_workspace_bytes(m, n)
    ws = Workspace.get(nbytes, h.device)
    with torch.cuda.device(h.device):
        check(lib.sparc_act_bwd_bias(m, n, int(act), _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(gout), n, _ptr(db),
                                     _ptr(ws), nbytes, _stream()), "sparc_act_bwd_bias")
    return gout, db


def wgrad_small(g: torch.Tensor, h: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """g^T h for a narrow g [m, n_out <= 64] and h [m, n_in], n_in a multiple of 64 (written into out when one is given)."""
    m, n_out = g.shape
    n_in = h.shape[1]
    dW = _usable(out, (n_out, n_in))
    if dW is None:
        dW = torch.empty(n_out, n_in, dtype=torch.float32, device=h.device)
    nbytes = lib.sparc_wgrad_small_workspace_bytes(n_out, n_in)
    ws = Workspace.get(nbytes, h.device)
    with torch.cuda.device(h.device):
        check(lib.sparc_wgrad_small(m, n_out, n_in, _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(dW), _ptr(ws), nbytes,
                                    _stream()), "sparc_wgrad_small")
    return dW


def first_layer_bwd(x: torch.Tensor, g: torch.Tensor, h: torch.Tensor, W: torch.Tensor | None = None,
                    b: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """(dW [n, din], db [n]) of a Linear+ReLU layer whose input needs no gradient,
    in one pass over g and h: the masked gradient is never written out.  With W and b (the layer's own
    parameters, for a layer whose forward ran on linear_narrow_in_fwd: at most 16 inputs) the activations h
    are not read either: the kernel recomputes the pre-activations' signs with the forward's arithmetic."""
    m, din = x.shape
    n = g.shape[1]
    out = _usable(out, ((din + 1) * n,))     # (the weight gradient followed by the bias gradient: a reducer's bucket)
    if out is None:
        out = torch.empty((din + 1) * n, dtype=torch.float32, device=g.device)
    nbytes = lib.sparc_first_layer_bwd_workspace_bytes(din, n)
    ws = Workspace.get(nbytes, g.device)
    with torch.cuda.device(g.device):
        if W is not None:
            check(lib.sparc_first_layer_bwd_recompute(m, din, n, _ptr(x), x.stride(0), _ptr(g), g.stride(0), _ptr(W), _ptr(b),
                                                      _ptr(out), _ptr(ws), nbytes, _stream()), "sparc_first_layer_bwd_recompute")
        else:
            check(lib.sparc_first_layer_bwd(m, din, n, _ptr(x), x.stride(0), _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(out),
                                            _ptr(ws), nbytes, _stream()), "sparc_first_layer_bwd")
    return out[:din * n].view(n, din), out[din * n:]


FWD_GRAM_MAX_ROWS = {True: 256 * 8 * 16, False: 256 * 16 * 16}   # one 16-row tile per wave, one workgroup per CU (wide: 8 waves)


def linear_narrow_fwd(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, act: int, gram_rows: int = 0):
    """act(x W^T + b) on the MFMA kernel for a narrow output layer, or None when the shape does not fit it.
    gram_rows > 0: returns (out, G) with G = out[:gram_rows]^T out[:gram_rows] in float64, from the same launch when the
    batch is small enough for it (G is None otherwise: the caller then asks for the Gram product itself)."""
    m, K = x.shape
    n_out = W.shape[0]
    if (K % 16 or n_out > 64 or ((n_out + 15) // 16) * 16 * (K + 4) * 4 > 150 * 1024 or not x.is_contiguous()
            or not W.is_contiguous() or x.data_ptr() % 16 or W.data_ptr() % 16 or x.dtype != torch.float32):
        return None
    out = torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    fused = 0 < gram_rows <= m <= FWD_GRAM_MAX_ROWS[n_out > 48]
    with torch.cuda.device(x.device):
        if fused:
            G = torch.empty(n_out, n_out, dtype=torch.float64, device=x.device)
            nbytes = lib.sparc_linear_narrow_fwd_gram_workspace_bytes(n_out)
            ws = Workspace.get(nbytes, x.device)
            check(lib.sparc_linear_narrow_fwd_gram(m, K, n_out, int(act), _ptr(x), x.stride(0), _ptr(W), _ptr(b), _ptr(out), n_out,
                                                   int(gram_rows), _ptr(G), _ptr(ws), nbytes, _stream()),
                  "sparc_linear_narrow_fwd_gram")
            return out, G
        check(lib.sparc_linear_narrow_fwd(m, K, n_out, int(act), _ptr(x), x.stride(0), _ptr(W), _ptr(b), _ptr(out), n_out,
                                          _stream()), "sparc_linear_narrow_fwd")
    return (out, None) if gram_rows > 0 else out


def linear_narrow_in_fwd(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, act: int):
    """act(x W^T + b) for a layer with at most 16 inputs (the first layer on a handful of node features): a store
    stream with the weights in registers, or None when the shape does not fit it."""
    m, K = x.shape
    N = W.shape[0]
    if (K > 16 or N % 4 or m == 0 or x.stride(1) != 1 or not W.is_contiguous() or not b.is_contiguous()
            or x.dtype != torch.float32 or W.dtype != torch.float32 or b.dtype != torch.float32):
        return None
    out = torch.empty(m, N, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.sparc_linear_narrow_in_fwd(m, K, N, int(act), _ptr(x), x.stride(0), _ptr(W), _ptr(b), _ptr(out), N,
                                             _stream()), "sparc_linear_narrow_in_fwd")
    return out


def dgrad_narrow_relu(g: torch.Tensor, W: torch.Tensor, hprev: torch.Tensor, db_out: torch.Tensor = None):
    """((g W) * [hprev > 0], its column sums): input gradient of a narrow layer fused with the ReLU
    derivative and bias gradient of the layer before it (the sums written into db_out when one is given)."""
    m, n_out = g.shape
    N = W.shape[1]
    dx = torch.empty(m, N, dtype=torch.float32, device=g.device)
    db = _usable(db_out, (N,))
    if db is None:
        db = torch.empty(N, dtype=torch.float32, device=g.device)
    nbytes = lib.sparc_dgrad_narrow_relu_workspace_bytes(m, N)
    ws = Workspace.get(nbytes, g.device)
    with torch.cuda.device(g.device):
        check(lib.sparc_dgrad_narrow_relu(m, n_out, N, _ptr(g), g.stride(0), _ptr(W), _ptr(hprev), hprev.stride(0), _ptr(dx), N,
                                          _ptr(db), _ptr(ws), nbytes, _stream()), "sparc_dgrad_narrow_relu")
    return dx, db


def dgrad_wide_relu(g: torch.Tensor, W: torch.Tensor, hprev: torch.Tensor, db_out: torch.Tensor = None):
    """((g W) * [hprev > 0], its column sums) for a wide g: the input gradient of a hidden layer fused with the ReLU
    derivative and bias gradient of the layer before it, or None when the shapes do not fit the kernel."""
    m, K = g.shape
    N = W.shape[1]
    if (K % 16 or N % 128 or m == 0 or not g.is_contiguous() or not W.is_contiguous() or not hprev.is_contiguous()
            or g.dtype != torch.float32 or W.dtype != torch.float32 or g.data_ptr() % 16 or W.data_ptr() % 16):
        return None
    dx = torch.empty(m, N, dtype=torch.float32, device=g.device)
    db = _usable(db_out, (N,))
    if db is None:
        db = torch.empty(N, dtype=torch.float32, device=g.device)
    nbytes = lib.sparc_dgrad_wide_relu_workspace_bytes(m, N)
    ws = Workspace.get(nbytes, g.device)
    with torch.cuda.device(g.device):
        check(lib.sparc_dgrad_wide_relu(m, K, N, _ptr(g), g.stride(0), _ptr(W), _ptr(hprev), hprev.stride(0), _ptr(dx), N,
                                        _ptr(db), _ptr(ws), nbytes, _stream()), "sparc_dgrad_wide_relu")
    return dx, db


def dgrad_wide_first(g: torch.Tensor, W: torch.Tensor, x: torch.Tensor, W0: torch.Tensor, b0: torch.Tensor,
                     out: torch.Tensor = None):
    """(dW [n, din], db [n]) of the network's first Linear+ReLU layer (input x [m, din], din <= 16, weights W0 [n, din], bias
    b0, forward on linear_narrow_in_fwd) from the gradient g reaching the NEXT layer's pre-activations and that layer's
    weights W [K, n]: the product (g W) * [x W0^T + b0 > 0] is summed into the two gradients inside the GEMM's epilogue
    and never stored, the ReLU derivative recomputed with the forward's arithmetic.  None when the shapes do not fit."""
    m, K = g.shape
    N = W.shape[1]
    din = x.shape[1]
    if (K % 16 or N % 128 or m == 0 or din > 16 or x.stride(1) != 1 or not g.is_contiguous() or not W.is_contiguous()
            or not W0.is_contiguous() or not b0.is_contiguous() or tuple(W0.shape) != (N, din)
            or any(t.dtype != torch.float32 for t in (g, W, x, W0, b0)) or g.data_ptr() % 16 or W.data_ptr() % 16):
        return None
    out = _usable(out, ((din + 1) * N,))     # (the weight gradient followed by the bias gradient: a reducer's bucket)
    if out is None:
        out = torch.empty((din + 1) * N, dtype=torch.float32, device=g.device)
    nbytes = lib.sparc_dgrad_wide_first_workspace_bytes(m, N, din)
    ws = Workspace.get(nbytes, g.device)
    with torch.cuda.device(g.device):
        check(lib.sparc_dgrad_wide_first(m, K, N, din, _ptr(g), g.stride(0), _ptr(W), _ptr(x), x.stride(0), _ptr(W0), _ptr(b0),
                                         _ptr(out), _ptr(ws), nbytes, _stream()), "sparc_dgrad_wide_first")
    return out[:din * N].view(N, din), out[din * N:]


def _destination(hook, l, W, cache):
    """(dW_out, db_out, both) where layer l's gradients should be written so that the hook need not copy them: a hook
    object with a destination(layer, w_shape, b_shape, like) method (a data-parallel reducer's bucket) names them;
    `both` is the [weight | bias] buffer as one flat tensor.  (None, None, None) otherwise."""
    if l not in cache:
        f = getattr(hook, "destination", None)
        cache[l] = f(l, tuple(W.shape), (W.shape[0],), W) if f is not None else (None, None, None)
    return cache[l]


def _layer_done(grads, l, hook):
    """hook(layer, dW, db), the second argument of SpectralMLP, is called as soon as a layer's weight and bias gradient
    exist and returns the pair to hand to autograd (a data-parallel trainer copies them into a bucket and starts its
    all-reduce there and then, so the collective overlaps the backward products of the layers below)."""
    if hook is not None and grads[2 * l] is not None and grads[2 * l + 1] is not None:
        grads[2 * l], grads[2 * l + 1] = hook(l, grads[2 * l], grads[2 * l + 1])


class SpectralMLP(torch.autograd.Function):
    """The whole SpectralNet MLP (src/SpectralTrainer.py:28-39, :96-97): Linear+ReLU layers and a final
    Linear+Tanh, forward and backward written out so that neighbouring steps share a pass:
    bias+ReLU ride in the epilogue of the wide forward products; the narrow output layer runs on the
    MFMA kernels (forward, weight gradient, input gradient fused with the previous layer's ReLU
    derivative and bias gradient); the first layer's backward is one pass that never writes the
    masked gradient.  The wide products stay library GEMMs.  Arguments: x, hook, W1, b1, ..., WL, bL (hook: None, or
    a callable told of every layer's gradients during the backward, see _layer_done)."""

    fuse_wide_dgrad = True   # class-wide switch: a hidden layer's input gradient on sparc_dgrad_wide_relu

    @staticmethod
    def forward(ctx, x, hook, *params):
        return SpectralMLP._forward(ctx, x, hook, params, 0)[0]

    @staticmethod
    def _forward(ctx, x, hook, params, gram_rows):
        """(y, G): G = y[:gram_rows]^T y[:gram_rows] (float64) when gram_rows > 0 and the output layer's kernel could
        take the product along, None otherwise."""
        require_gpu(x.device)
        ctx.hook = hook
        Ws, bs = params[0::2], params[1::2]
        acts = [x]
        ctx.first_own = False   # the first layer ran on linear_narrow_in_fwd: its backward can recompute the ReLU mask
        for W, b in zip(Ws[:-1], bs[:-1]):
            h = linear_narrow_in_fwd(acts[-1], W, b, ACT_RELU) if len(acts) == 1 else None
            if len(acts) == 1 and h is not None:
                ctx.first_own = True
            acts.append(h if h is not None else torch._addmm_activation(b, acts[-1], W.t()))
        G = None
        y = linear_narrow_fwd(acts[-1], Ws[-1], bs[-1], ACT_TANH, gram_rows)
        if y is not None and gram_rows > 0:
            y, G = y
        if y is None:
            y = torch.addmm(bs[-1], acts[-1], Ws[-1].t()).tanh_()
        ctx.save_for_backward(*acts, y, *Ws, bs[0])
        ctx.n_layers = len(Ws)
        return y, G

    @staticmethod
    def backward(ctx, gy):
        L = ctx.n_layers
        hook = ctx.hook
        saved = ctx.saved_tensors
        acts, y, Ws, b_first = saved[:L], saved[L], saved[L + 1:2 * L + 1], saved[2 * L + 1]
        grads = [None] * (2 * L)
        need_x = ctx.needs_input_grad[0]
        # output layer: tanh derivative + bias gradient, weight gradient, input gradient (+ previous ReLU)
        dest = {}
        dW_to, db_to, _ = _destination(hook, L - 1, Ws[L - 1], dest)
        gp, db = act_bwd_bias(_f32c(gy), y, ACT_TANH, db_to)
        grads[2 * (L - 1)] = _wgrad(gp, acts[L - 1], Ws[L - 1], dW_to)
        grads[2 * (L - 1) + 1] = db
        _layer_done(grads, L - 1, hook)
        gx = None
        if L == 1:
            gx = gp.mm(Ws[0]) if need_x else None
            return (gx, None, *grads)
        # gradient reaching the output of hidden layer L-2: already multiplied by its ReLU derivative
        # (with its bias gradient) when the fused kernel applies, raw otherwise
        W, h = Ws[L - 1], acts[L - 1]
        if W.shape[0] <= 64 and W.shape[1] % 64 == 0 and W.is_contiguous() and W.data_ptr() % 16 == 0:
            pending = dgrad_narrow_relu(gp, W, h, _destination(hook, L - 2, Ws[L - 2], dest)[1])
        else:
            pending = gp.mm(W)
        for l in range(L - 2, -1, -1):   # hidden layers, last to first; acts[l] is layer l's input, acts[l + 1] its output
            if isinstance(pending, tuple):
                gp, db = pending
            else:
                xin, hout = acts[l], acts[l + 1]
                if (l == 0 and not need_x and xin.shape[1] < 64 and hout.shape[1] % 64 == 0 and xin.stride(1) == 1
                        and xin.dtype == torch.float32 and pending.data_ptr() % 16 == 0):
                    own = ctx.first_own and Ws[0].is_contiguous() and xin.shape[1] <= 16
                    dWt, db = first_layer_bwd(xin, pending, hout, Ws[0] if own else None, b_first if own else None,
                                              _destination(hook, 0, Ws[0], dest)[2])
                    grads[0], grads[1] = dWt, db
                    _layer_done(grads, 0, hook)
                    return (None, None, *grads)
                gp, db = act_bwd_bias(pending, hout, ACT_RELU, _destination(hook, l, Ws[l], dest)[1])
            grads[2 * l] = _wgrad(gp, acts[l], Ws[l], _destination(hook, l, Ws[l], dest)[0])
            grads[2 * l + 1] = db
            _layer_done(grads, l, hook)
            if l == 1 and SpectralMLP.fuse_wide_dgrad and not need_x and ctx.first_own:
                # layer 0 is the first layer: its gradients are column sums of this product, which is not stored
                first = dgrad_wide_first(gp, Ws[1], acts[0], Ws[0], b_first, _destination(hook, 0, Ws[0], dest)[2])
                if first is not None:
                    grads[0], grads[1] = first
                    _layer_done(grads, 0, hook)
                    return (None, None, *grads)
            if l > 1 and SpectralMLP.fuse_wide_dgrad:   # layer l - 1 is a hidden layer behind the first: its ReLU rides along
                pending = dgrad_wide_relu(gp, Ws[l], acts[l], _destination(hook, l - 1, Ws[l - 1], dest)[1])
                if pending is None:
                    pending = gp.mm(Ws[l])
            elif l > 0:
                pending = gp.mm(Ws[l])
            elif need_x:
                gx = gp.mm(Ws[0])
        return (gx, None, *grads)


class SpectralMLPGram(torch.autograd.Function):
    """SpectralMLP for the two batches of a training step in one pass: arguments x, hook, gram_rows, W1, b1, ...; returns
    (y, G) with G = y[:gram_rows]^T y[:gram_rows] in float64 (not differentiable: the orthonormalisation's backward has
    its own formula in terms of y), an empty tensor when the output layer's kernel could not take the product along."""

    @staticmethod
    def forward(ctx, x, hook, gram_rows, *params):
        y, G = SpectralMLP._forward(ctx, x, hook, params, int(gram_rows))
        if G is None:
            G = torch.empty(0, dtype=torch.float64, device=y.device)
        ctx.mark_non_differentiable(G)
        ctx.set_materialize_grads(False)   # no zero-filled gradient for G on every backward
        return y, G

    @staticmethod
    def backward(ctx, gy, _gG):
        if gy is None:
            return (None,) * (3 + 2 * ctx.n_layers)
        gx, _, *grads = SpectralMLP.backward(ctx, gy)
        return (gx, None, None, *grads)


def _wgrad(gp: torch.Tensor, xin: torch.Tensor, W: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """gp^T xin, on the MFMA kernel when one side is narrow (written into out, a contiguous tensor of W's shape, where
    the kernel produces that layout)."""
    n_out, n_in = W.shape
    if n_out <= 64 and n_in % 64 == 0 and xin.is_contiguous() and xin.data_ptr() % 16 == 0:
        return wgrad_small(gp, xin, out)
    if n_in <= 64 and n_out % 64 == 0 and gp.data_ptr() % 16 == 0:
        return wgrad_small(xin, gp).t()
    return wgrad_wide(gp, xin, out)


WGRAD_SPLIT_ROWS = 16384   # batches at least this long have their weight gradients summed in four parts


def wgrad_wide(gp: torch.Tensor, xin: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """gp^T xin, the weight gradient of a wide layer: a library GEMM.  A result of fewer than 256 tiles of 64 x 64
    (1024 -> 512: 128) leaves half of the CUs idle, because the library does not split the long sum over the rows:
    such a product runs as ONE batched GEMM over four parts of the rows whose partial results are added (489 -> 358 us
    at 49 k rows; the summation order changes, towards the smaller error)."""
    m, n_out = gp.shape
    n_in = xin.shape[1]
    tiles = ((n_out + 63) // 64) * ((n_in + 63) // 64)
    out = _usable(out, (n_out, n_in))
    if tiles >= 256 or m < WGRAD_SPLIT_ROWS or not (gp.is_contiguous() and xin.is_contiguous()):
        return torch.mm(gp.t(), xin, out=out) if out is not None else gp.t().mm(xin)
    parts = 4
    rows = m // parts * parts
    four = torch.bmm(gp[:rows].view(parts, rows // parts, n_out).transpose(1, 2), xin[:rows].view(parts, rows // parts, n_in))
    out = torch.sum(four, 0, out=out) if out is not None else four.sum(0)
    if rows < m:
        out.addmm_(gp[rows:].t(), xin[rows:])
    return out


class LinearAct(torch.autograd.Function):
    """act(x W^T + b) of one MLP layer (src/SpectralTrainer.py:28-39): the products are library
    GEMMs (bias and ReLU in the forward product's epilogue); the activation derivative, the bias
    gradient and the narrow output layer's weight gradient run in libsparc_hip.so."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        require_gpu(x.device)
        h = linear_narrow_in_fwd(x, W, b, act) if x.dim() == 2 else None   # a handful of input columns
        if h is not None:
            pass
        elif act == ACT_RELU:
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
                dW = wgrad_wide(gp, x)
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


def _index_to_device(idx, dev) -> torch.Tensor:
    """int32 device copy of an index list (a device tensor stays where it is)."""
    if torch.is_tensor(idx) and idx.device == dev:
        return idx.to(torch.int32).contiguous()
    arr = np.ascontiguousarray(idx.cpu().numpy() if torch.is_tensor(idx) else np.asarray(idx)).astype(np.int32, copy=False)
    return torch.as_tensor(arr, device=dev)


class OrthWeights(torch.autograd.Function):
    """Yt [m, k] -> orthonormalisation weights [k, k]  (src/SpectralTrainer.py:42-70)."""

    last_info = None   # device int of the latest call
    # Per device: the failure codes of the calls since the last check.  Every call writes its code into the next slot of
    # a ring (no kernel of its own to fold it into a running maximum); slot RING carries the maximum of the laps before.
    RING = 1024
    _rings = {}
    _calls = {}

    @classmethod
    def _info_slot(cls, dev):
        key = dev.index or 0
        ring = cls._rings.get(key)
        if ring is None:
            ring = cls._rings[key] = torch.zeros(cls.RING + 1, dtype=torch.int32, device=dev)
            cls._calls[key] = 0
        n = cls._calls[key]
        cls._calls[key] = n + 1
        if n and n % cls.RING == 0:
            ring[cls.RING:] = ring.max()
        return ring[n % cls.RING:n % cls.RING + 1]

    @classmethod
    def worst(cls, device):
        """int32 [1] on the device: the largest failure code since the previous check (no host sync); None before the
        first call on this device."""
        ring = cls._rings.get(torch.device(device).index or 0)
        return None if ring is None else ring.max().reshape(1)

    @classmethod
    def set_worst(cls, device, code: int):
        ring = cls._rings.get(torch.device(device).index or 0)
        if ring is not None:
            ring.zero_()
            ring[cls.RING] = int(code)

    @classmethod
    def check_and_reset(cls, device):
        """Largest failure code since the previous check on this device (one host sync), then cleared.
        Non-zero: some call met a Gram matrix that is not positive definite."""
        ring = cls._rings.get(torch.device(device).index or 0)
        if ring is None:
            return 0
        code = int(ring.max())
        ring.zero_()
        return code

    @staticmethod
    def weights(Y, top_idx, mode, G=None):
        """(Wo float32 [k, k], saved float64 [2 k k + k] = Rc, Rinv, signs) for the contiguous-row batch Y [m, k].
        G: Y^T Y in float64 when the caller has it already."""
        m, k = Y.shape
        if k > 64:
            raise ValueError("spectral dimension k must be <= 64")
        dev = Y.device
        if G is None:
            G = gram(Y, Y)
        Wo = torch.empty(k, k, dtype=torch.float32, device=dev)
        saved = torch.empty(2 * k * k + k, dtype=torch.float64, device=dev)
        Rc, Rinv, signs = saved[:k * k], saved[k * k:2 * k * k], saved[2 * k * k:]
        info = OrthWeights._info_slot(dev)   # written by every call
        tidx = None
        if top_idx is not None and not (torch.is_tensor(top_idx) and top_idx.is_cuda) and not torch.cuda.is_current_stream_capturing():
            # row numbers on the host (the reference's numpy permutation): they travel in the kernel's arguments
            arr = np.ascontiguousarray(top_idx.numpy() if torch.is_tensor(top_idx) else np.asarray(top_idx))
            if arr.ndim == 1:
                arr = np.clip(arr[:k], -1, m).astype(np.int32)   # (only the first k are used; out of range = a row of zeros)
                with torch.cuda.device(dev):
                    check(lib.sparc_orth_weights_host_idx(m, k, _ptr(G), _ptr(Y), Y.stride(0), arr.ctypes.data, int(arr.shape[0]),
                                                          int(mode), _ptr(Wo), _ptr(Rc), _ptr(Rinv), _ptr(signs), _ptr(info),
                                                          _stream()), "sparc_orth_weights_host_idx")
                OrthWeights.last_info = info
                return Wo, saved
        if top_idx is not None:
            tidx = _index_to_device(top_idx, dev)
            if tidx.numel() < k:  # batches with fewer than k rows: pad with out-of-range rows (read as zero)
                tidx = torch.cat([tidx, torch.full((k - tidx.numel(),), m, dtype=torch.int32, device=dev)])
        with torch.cuda.device(dev):
            check(lib.sparc_orth_weights(m, k, _ptr(G), _ptr(Y), Y.stride(0), _ptr(tidx), int(mode), _ptr(Wo), _ptr(Rc),
                                         _ptr(Rinv), _ptr(signs), _ptr(info), _stream()), "sparc_orth_weights")
        OrthWeights.last_info = info  # device int; non-zero = Gram matrix not positive definite (checked lazily)
        return Wo, saved

    @staticmethod
    def weights_backward(Y, saved, gW64, mode, out):
        """out [m, k] (float32 rows) = the gradient with respect to Y of a scalar whose gradient with respect to Wo is
        gW64 (float64 [k, k]).  H stays in float64 and Y H is accumulated in float64: for an ill-conditioned batch the
        entries of H cancel down to a product many orders of magnitude smaller."""
        m, k = Y.shape
        Rc, Rinv, signs = saved[:k * k], saved[k * k:2 * k * k], saved[2 * k * k:]
        H = torch.empty(k, k, dtype=torch.float64, device=Y.device)
        with torch.cuda.device(Y.device):
            check(lib.sparc_orth_weights_bwd_f64(m, k, _ptr(gW64), _ptr(Rc), _ptr(Rinv), _ptr(signs), int(mode), _ptr(H),
                                                 _stream()), "sparc_orth_weights_bwd_f64")
            check(lib.sparc_orth_apply_f64(m, k, _ptr(Y), Y.stride(0), _ptr(H), _ptr(out), out.stride(0), _stream()),
                  "sparc_orth_apply_f64")
        return out

    @staticmethod
    def forward(ctx, Yt, top_idx, mode):
        require_gpu(Yt.device)
        Y = _f32c(Yt)
        Wo, saved = OrthWeights.weights(Y, top_idx, mode)
        ctx.save_for_backward(Y, saved)
        ctx.mode = int(mode)
        return Wo

    @staticmethod
    def backward(ctx, gWo):
        Y, saved = ctx.saved_tensors
        m, k = Y.shape
        gY = torch.empty(m, k, dtype=torch.float32, device=Y.device)
        OrthWeights.weights_backward(Y, saved, gWo.detach().to(torch.float64).contiguous(), ctx.mode, gY)
        return gY, None, None


class OrthPair(torch.autograd.Function):
    """The orthonormalisation of one training step as ONE node of the autograd graph: Yt [m2 + m1, k] holds the MLP's
    output for the orthogonality batch (first m2 rows) and the gradient batch; the weights come from the first
    (src/SpectralTrainer.py:222-223, :42-70) and are applied to the second (:232, :103).  Returns (Y [m1, k], Wo).
    Same kernels as OrthWeights + OrthApply + SplitRows; as one node the backward writes both parts of the gradient
    straight into one [m2 + m1, k] buffer (no concatenation) and the k x k gradient of the weights stays in float64
    from the Gram product to the factorisation's backward (no casts)."""

    @staticmethod
    def forward(ctx, Yt, m2, top_idx, mode, G=None):
        require_gpu(Yt.device)
        Y = _f32c(Yt)
        if G is not None and (G.shape != (Y.shape[1], Y.shape[1]) or G.dtype != torch.float64):
            G = None   # (the producer of Yt had no Gram matrix to give)
        Wo, saved = OrthWeights.weights(Y[:m2], top_idx, mode, G)
        out = orth_apply(Y[m2:], Wo)
        ctx.save_for_backward(Y, Wo, saved)
        ctx.m2, ctx.mode = int(m2), int(mode)
        ctx.mark_non_differentiable(Wo)
        ctx.set_materialize_grads(False)   # no zero-filled gradient for Wo on every backward
        return out, Wo

    @staticmethod
    def backward(ctx, gY, _gWo):
        if gY is None:
            return None, None, None, None, None
        Y, Wo, saved = ctx.saved_tensors
        m2 = ctx.m2
        g = _f32c(gY)
        gYt = torch.empty_like(Y)
        gW = gram_apply(Y[m2:], g, Wo, True, gYt[m2:])     # Yt^T gY, and gY Wo^T into the gradient batch's rows
        OrthWeights.weights_backward(Y[:m2], saved, gW, ctx.mode, gYt[:m2])
        return gYt, None, None, None, None


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
    """Spectral loss on a ClusterBatch (src/SpectralTrainer.py:130-137), either branch of is_normalized.

    The gradient with respect to Y falls out of the same pass as the loss (the forward
    and the transposed product share one sweep over the CSR), so backward only scales it.
    """

    @staticmethod
    def forward(ctx, Y, batch: ClusterBatch, normalized: bool = True):
        want = bool(ctx.needs_input_grad[0])
        loss, grad = batch.lap(Y, want, normalized)
        if want:
            ctx.save_for_backward(grad)
        return loss.to(torch.float32)

    _units = {}   # per device: the float32 scalar 1 that unit_gradient hands out

    @classmethod
    def unit_gradient(cls, loss: torch.Tensor) -> torch.Tensor:
        """The scalar 1 to start the backward from, `loss.backward(gradient=LapLoss.unit_gradient(loss))`: the same
        gradients as `loss.backward()` without the kernel that fills a fresh tensor of ones and without the one that
        multiplies d loss / d Y by it (the backward recognises this very tensor)."""
        key = (loss.device, loss.dtype)
        if key not in cls._units:
            cls._units[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
        return cls._units[key]

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        unit = LapLoss._units.get((gl.device, gl.dtype))
        if unit is not None and gl.data_ptr() == unit.data_ptr() and gl.dim() == 0:
            return grad, None, None
        return grad * gl, None, None
