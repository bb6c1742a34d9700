"""Device-resident cluster batches: the support matrix as dual-value CSR plus the
per-batch affinity vectors, built once and reused by every training step.

Stands in for what the reference recomputes on each step from the COO tuple:
SpectralTrainer._get_support_matrix (src/SpectralTrainer.py:275-281) and
SpectralTrainer._get_valid_affinity_matrix (:284-293).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

EPS_BACKGROUND = 1e-6  # src/SpectralTrainer.py:288


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Workspace:
    """One growing scratch buffer per device; every kernel call on the current stream shares it."""
    _buffers = {}

    @classmethod
    def get(cls, nbytes: int, device) -> torch.Tensor:
        key = torch.device(device).index or 0
        buf = cls._buffers.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20) + 256, dtype=torch.uint8, device=device)
            cls._buffers[key] = buf
        off = (-buf.data_ptr()) % 256
        return buf[off:]


def require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SparcHipError("sparc_amd runs on an MI355X only (device must be a ROCm 'cuda' device); "
                                 "there is no CPU path")
    return device


class ClusterBatch:
    """Support of one cluster batch on the device.

    Attributes: m, nnz, rowptr (int64[m+1]), colidx (int32[nnz]), val / valT (float32[nnz],
    valT is None for a numerically symmetric support), rowc (float32[m,8]), d, cw, rho
    (float32[m]: row sums, column sums and pre-normalisation row norms of the affinity W),
    coupling (float64[16]).
    """

    def __init__(self):
        self.m = 0
        self.nnz = 0

    @classmethod
    def from_support(cls, support, device, eps: float = EPS_BACKGROUND) -> "ClusterBatch":
        """support = (coords[nnz,2] (row, col), values[nnz], shape) as written by the
        reference's to_tuple (src/utils.py:41-47); values are cast to float32 as
        torch.FloatTensor(values) does (src/SpectralTrainer.py:277)."""
        device = require_gpu(device)
        coords, values, shape = support
        m = int(shape[0])
        if int(shape[1]) != m:
            raise ValueError("support matrix must be square")
        if torch.is_tensor(coords):
            coords_d = coords.to(device=device, dtype=torch.int32).contiguous()
        else:
            coords_d = torch.from_numpy(np.ascontiguousarray(np.asarray(coords).reshape(-1, 2), dtype=np.int32)).to(device)
        if torch.is_tensor(values):
            values_d = values.to(device=device, dtype=torch.float64).contiguous()
        else:
            values_d = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(device)
        return cls.from_device_coo(m, coords_d, values_d, eps)

    @classmethod
    def from_device_coo(cls, m: int, coords: torch.Tensor, values: torch.Tensor, eps: float = EPS_BACKGROUND):
        self = cls()
        device = coords.device
        nnz_in = int(coords.shape[0])
        if nnz_in and (int(coords.min()) < 0 or int(coords.max()) >= m):
            raise ValueError("support coordinates out of range")
        cap = max(2 * nnz_in, 1)
        rowptr = torch.empty(m + 1, dtype=torch.int64, device=device)
        colidx = torch.empty(cap, dtype=torch.int32, device=device)
        val = torch.empty(cap, dtype=torch.float32, device=device)
        valT = torch.empty(cap, dtype=torch.float32, device=device)
        meta = torch.zeros(2, dtype=torch.int64, device=device)  # [nnz_out, is_symmetric (int32 in the low word)]
        nbytes = lib.sparc_csr_build_workspace_bytes(m, nnz_in)
        ws = Workspace.get(nbytes, device)
        with torch.cuda.device(device):
            check(lib.sparc_csr_build(m, nnz_in, _ptr(coords), _ptr(values), _ptr(rowptr), _ptr(colidx), _ptr(val),
                                      _ptr(valT), _ptr(meta[0:1]), _ptr(meta[1:2]), _ptr(ws), nbytes, _stream()),
                  "sparc_csr_build")
        meta_h = meta.cpu()
        nnz = int(meta_h[0])
        symmetric = bool(int(meta_h[1]) & 0xFFFFFFFF)
        self.m, self.nnz = m, nnz
        self.rowptr = rowptr
        self.colidx = colidx[:nnz].clone()
        self.val = val[:nnz].clone()
        self.valT = None if symmetric else valT[:nnz].clone()
        self.device = device
        self._prepare(eps)
        return self

    def _prepare(self, eps):
        m, device = self.m, self.device
        self.rowc = torch.empty(m, 8, dtype=torch.float32, device=device)
        vec = torch.empty(3, m, dtype=torch.float32, device=device)
        self.coupling = torch.empty(16, dtype=torch.float64, device=device)
        nbytes = lib.sparc_affinity_prep_workspace_bytes(m, self.nnz)
        ws = Workspace.get(nbytes, device)
        with torch.cuda.device(device):
            check(lib.sparc_affinity_prep(m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(self.val),
                                          _ptr(self.valT), eps, _ptr(self.rowc), _ptr(vec), _ptr(self.coupling),
                                          _ptr(ws), nbytes, _stream()), "sparc_affinity_prep")
        self.d, self.cw, self.rho = vec[0], vec[1], vec[2]

    # ---- per-step work -------------------------------------------------------------------
    def lap(self, Y: torch.Tensor, want_grad: bool):
        """loss (float64 0-d tensor) and d loss / d Y (or None) for is_normalized=True."""
        if Y.dim() != 2 or Y.shape[0] != self.m:
            raise ValueError(f"Y must be [{self.m}, k], got {tuple(Y.shape)}")
        k = int(Y.shape[1])
        if k < 1 or k > 64:
            raise ValueError("spectral dimension k must be in 1..64")
        Y = Y.detach()
        if Y.dtype != torch.float32 or not Y.is_contiguous():
            Y = Y.float().contiguous()
        loss = torch.empty((), dtype=torch.float64, device=self.device)
        grad = torch.empty_like(Y) if want_grad else None
        nbytes = lib.sparc_lap_workspace_bytes(self.m, k)
        ws = Workspace.get(nbytes, self.device)
        with torch.cuda.device(self.device):
            check(lib.sparc_lap_fwd(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(self.val),
                                    _ptr(self.valT), _ptr(self.rowc), _ptr(self.coupling), k, k, _ptr(Y), _ptr(loss),
                                    _ptr(grad), _ptr(ws), nbytes, _stream()), "sparc_lap_fwd")
        return loss, grad

    def spmm(self, X: torch.Tensor, add: torch.Tensor | None = None, transpose: bool = False) -> torch.Tensor:
        """Z = A X (+ add); width of X must be 16, 32, 64 or 128."""
        X = X.contiguous()
        Z = torch.empty_like(X)
        v = self.val if (not transpose or self.valT is None) else self.valT
        with torch.cuda.device(self.device):
            check(lib.sparc_spmm(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(v), int(X.shape[1]),
                                 _ptr(X), _ptr(add), _ptr(Z), _stream()), "sparc_spmm")
        return Z

    def nbytes(self) -> int:
        t = [self.rowptr, self.colidx, self.val, self.rowc, self.d, self.cw, self.rho]
        if self.valT is not None:
            t.append(self.valT)
        return sum(x.numel() * x.element_size() for x in t)
