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
TILE_MAX_ROWS = 512     # SPARC_TILE_MAX_ROWS: rows of one LDS window (256 bytes each)
TILE_LONG_ROW = 64       # SPARC_TILE_LONG_ROW: longer rows get a whole wavefront
TILE_MAX_OUTSIDE = 2    # SPARC_TILE_MAX_OUTSIDE: entries of a row outside its window that ride in the row record
TILE_TARGET_WGS = 256   # one workgroup per CU


def plan_windows(m: int, cluster_sizes=None, max_rows: int = TILE_MAX_ROWS):
    """Row windows [begin, end) for the LDS-tiled kernel: consecutive clusters are merged while
    they fit one window, clusters larger than a window are split evenly; without cluster sizes
    the rows are cut into equal windows."""
    if cluster_sizes is None or int(np.sum(cluster_sizes)) != m:
        cluster_sizes = [m]
    bounds = [0]
    for sz in cluster_sizes:
        sz = int(sz)
        if sz <= 0:
            continue
        if sz > max_rows:                         # split an oversized cluster evenly
            parts = -(-sz // max_rows)
            base, rem = divmod(sz, parts)
            start = bounds[-1]
            for i in range(parts):
                start += base + (1 if i < rem else 0)
                bounds.append(start)
            continue
        bounds.append(bounds[-1] + sz)
    # merge neighbours greedily
    wins, cur = [], 0
    for i in range(1, len(bounds)):
        if bounds[i] - cur > max_rows:
            wins.append((cur, bounds[i - 1]))
            cur = bounds[i - 1]
    wins.append((cur, bounds[-1]))
    return [(a, b) for a, b in wins if b > a]


def plan_tiles(windows, lo: np.ndarray, hi: np.ndarray, n_out: np.ndarray, out_rec: np.ndarray, n_slabs: int,
               target_wgs: int = TILE_TARGET_WGS):
    """The tiling buffer of the LDS-tiled kernel (layout: include/sparc_hip.h): descriptors and row
    records.  [lo, hi) are the positions of every row's in-window entries, n_out the number of
    its other entries and out_rec int32[m][6] = {col0, col1, val0, valT0, val1, valT1} (float
    bits) the first two of them.  Every window's rows are sorted by in-window length (longest
    first) and dealt round-robin to J parts so that windows x slabs x J fills the chip with
    equal work; rows with more than TILE_MAX_OUTSIDE outside entries go to slab = -1 workgroups,
    which gather from global memory."""
    nw = len(windows)
    J = max(1, target_wgs // max(nw * n_slabs, 1))
    desc, recs = [], []
    q = 0
    length = hi - lo
    tiled_ok = n_out <= TILE_MAX_OUTSIDE

    def records(rows):
        z = np.zeros_like(rows)
        head = np.stack([rows, lo[rows], hi[rows], np.minimum(n_out[rows], TILE_MAX_OUTSIDE)], axis=1)
        return np.concatenate([head, out_rec[rows], np.stack([z, z], axis=1)], axis=1)

    for (a, b) in windows:
        rows = np.arange(a, b)
        rows = rows[tiled_ok[a:b]]
        if len(rows) == 0:
            continue
        order = rows[np.argsort(-length[rows], kind="stable")]
        parts = max(1, min(J, (len(order) + 63) // 64))   # at least one row per 16-lane group of the workgroup
        for j in range(parts):
            mine = order[j::parts]
            if len(mine) == 0:
                continue
            recs.append(records(mine))
            n_long = int((length[mine] > TILE_LONG_ROW).sum())
            for s_ in range(n_slabs):
                desc.append((q, q + len(mine), a, b, s_, n_long, 0, 0))
            q += len(mine)
    loose = np.nonzero(~tiled_ok)[0]
    if len(loose):
        for part in np.array_split(loose, max(1, (len(loose) + 15) // 16)):
            if len(part):
                recs.append(records(part))
                desc.append((q, q + len(part), 0, 0, -1, 0, 0, 0))
                q += len(part)
    desc = np.asarray(_xcd_align(desc), dtype=np.int32).reshape(-1, 8)
    recs = np.concatenate(recs, axis=0).astype(np.int32) if recs else np.zeros((0, 12), np.int32)
    return desc, recs


N_XCD = 8  # workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD has its own L2


def _xcd_align(desc):
    """Order the descriptors so that all workgroups of one window run on the same XCD: the parts of a
    window stage the same operand rows and its slabs stream the same stored entries, so the second
    reader finds them in that XCD's L2 instead of fetching them again.  Windows are dealt to eight
    bins (heaviest first, to the lightest bin); bin x supplies the descriptors at positions x, x + 8, ...;
    shorter bins are padded with empty descriptors (a workgroup that finds no rows and returns)."""
    if len(desc) <= N_XCD:
        return desc
    groups = {}
    for d in desc:
        key = (d[2], d[3]) if d[4] >= 0 else ("loose", d[0])
        groups.setdefault(key, []).append(d)
    bins = [[] for _ in range(N_XCD)]
    load = [0] * N_XCD
    for key, ds in sorted(groups.items(), key=lambda kv: -sum(d[1] - d[0] for d in kv[1])):
        x = min(range(N_XCD), key=lambda i: (load[i], len(bins[i])))
        bins[x].extend(ds)
        load[x] += sum(d[1] - d[0] for d in ds)
    depth = max(len(b) for b in bins)
    empty = (0, 0, 0, 0, 0, 0, 0, 0)
    out = []
    for r in range(depth):
        for x in range(N_XCD):
            out.append(bins[x][r] if r < len(bins[x]) else empty)
    while out and out[-1] == empty:
        out.pop()
    return out


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

    use_tiles = True  # class-wide switch between the LDS-tiled and the plain row kernel

    def __init__(self):
        self.m = 0
        self.nnz = 0

    @classmethod
    def from_support(cls, support, device, eps: float = EPS_BACKGROUND, cluster_sizes=None) -> "ClusterBatch":
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
        return cls.from_device_coo(m, coords_d, values_d, eps, cluster_sizes)

    @classmethod
    def from_device_coo(cls, m: int, coords: torch.Tensor, values: torch.Tensor, eps: float = EPS_BACKGROUND,
                        cluster_sizes=None):
        """cluster_sizes: number of rows of each consecutive METIS cluster of the batch (a hint
        for the LDS tiling; any value is correct, good ones make every gather an LDS read)."""
        self = cls()
        self.cluster_sizes = cluster_sizes
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
        self._windows = plan_windows(m, cluster_sizes)
        self._tiles = {}
        self._window_ranges = None
        self._prepare(eps)
        return self

    def _in_window_ranges(self):
        """Per row: positions [lo, hi) of its entries whose column lies in the row's window, the
        number of its other entries and the first two of those (column, val, valT as float bits);
        computed once on the device, kept on the host."""
        if self._window_ranges is None:
            m, dev = self.m, self.device
            wb = torch.empty(m, dtype=torch.int64, device=dev)
            we = torch.empty(m, dtype=torch.int64, device=dev)
            for a, b in self._windows:
                wb[a:b] = a
                we[a:b] = b
            p0, p1 = self.rowptr[:-1], self.rowptr[1:]
            rows = torch.repeat_interleave(torch.arange(m, device=dev), p1 - p0)
            keys = rows * m + self.colidx.to(torch.int64)        # ascending: CSR order
            base = torch.arange(m, device=dev, dtype=torch.int64) * m
            lo = torch.searchsorted(keys, base + wb)
            hi = torch.searchsorted(keys, base + we)
            n_out = (lo - p0) + (p1 - hi)
            # positions of the first two outside entries: before the window first, then after it
            nb = lo - p0
            pos0 = torch.where(nb > 0, p0, hi)
            pos1 = torch.where(nb > 1, p0 + 1, torch.where(nb == 1, hi, hi + 1))
            last = max(self.nnz - 1, 0)
            pos0c, pos1c = pos0.clamp(0, last), pos1.clamp(0, last)
            valT = self.valT if self.valT is not None else self.val
            bits = lambda t: t.view(torch.int32).to(torch.int64)
            has0, has1 = n_out > 0, n_out > 1
            zero = torch.zeros_like(lo)
            rec = torch.stack([torch.where(has0, self.colidx[pos0c].to(torch.int64), zero),
                               torch.where(has1, self.colidx[pos1c].to(torch.int64), zero),
                               torch.where(has0, bits(self.val[pos0c]), zero),
                               torch.where(has0, bits(valT[pos0c]), zero),
                               torch.where(has1, bits(self.val[pos1c]), zero),
                               torch.where(has1, bits(valT[pos1c]), zero)], dim=1) if self.nnz else \
                torch.zeros(m, 6, dtype=torch.int64, device=dev)
            self._window_ranges = tuple(t.cpu().numpy() for t in (lo, hi, n_out, rec))
        return self._window_ranges

    def tiles(self, n_slabs: int):
        """(device tiling buffer, descriptor count, largest window) for the tiled kernel, cached per slab count."""
        hit = self._tiles.get(n_slabs)
        if hit is None:
            if self.nnz >= 2 ** 31:
                raise ValueError("the tiled kernel needs nnz < 2^31")
            lo, hi, n_out, out_rec = self._in_window_ranges()
            desc, recs = plan_tiles(self._windows, lo, hi, n_out, out_rec, n_slabs)
            assert recs.shape[0] == self.m
            n_loose = int(sum(d[1] - d[0] for d in desc if d[4] < 0))
            if n_loose > 0.05 * self.m:   # too many rows reach outside their window: tiling does not pay
                self._tiles[n_slabs] = (None, 0, 0)
                return self._tiles[n_slabs]
            buf = np.concatenate([desc.ravel(), recs.ravel()]).astype(np.int32)
            rows = max(b - a for a, b in self._windows)
            hit = (torch.from_numpy(buf).to(self.device), int(desc.shape[0]), int(rows))
            self._tiles[n_slabs] = hit
        return hit

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

    @staticmethod
    def train_slabs(k: int) -> int:
        """Column slabs of the tiled training pass for spectral dimension k (0: the row kernel runs)."""
        kp = 16 if k <= 16 else (32 if k <= 32 else 64)
        return kp // 32

    def warm(self, k: int, training: bool = True):
        """Build the tiling for spectral dimension k now (it is otherwise built on first use)."""
        kp = 16 if k <= 16 else (32 if k <= 32 else 64)
        if self.use_tiles and ((training and kp >= 32) or (not training and kp == 64)):
            self.tiles(kp // 32 if training else 1)
        return self

    # ---- per-step work -------------------------------------------------------------------
    def lap(self, Y: torch.Tensor, want_grad: bool, normalized: bool = True):
        """loss (float64 0-d tensor) and d loss / d Y (or None): the reference's is_normalized=True branch
        (Yn = Y / sqrt(d), src/SpectralTrainer.py:132-135), or with normalized=False the plain one (Yn = Y):
        the kernels take 1/sqrt(d) from the per-row record only, so the plain branch is the same pass over a
        copy of the records whose 1/sqrt(d) field is 1."""
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
        kp = 16 if k <= 16 else (32 if k <= 32 else 64)
        tiles, ntiles, tile_rows = None, 0, 0
        if self.use_tiles and ((want_grad and kp >= 32) or (not want_grad and kp == 64)):
            tiles, ntiles, tile_rows = self.tiles(kp // 32 if want_grad else 1)
        nbytes = lib.sparc_lap_workspace_bytes(self.m, k, ntiles)
        ws = Workspace.get(nbytes, self.device)
        with torch.cuda.device(self.device):
            rowc = self.rowc
            if not normalized:
                if getattr(self, "_rowc_plain", None) is None:
                    self._rowc_plain = self.rowc.clone()
                    self._rowc_plain[:, 3] = 1.0
                rowc = self._rowc_plain
            check(lib.sparc_lap_fwd(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(self.val),
                                    _ptr(self.valT), _ptr(rowc), _ptr(self.coupling), k, k, _ptr(Y), _ptr(loss),
                                    _ptr(grad), _ptr(tiles), ntiles, tile_rows, _ptr(ws), nbytes, _stream()),
                  "sparc_lap_fwd")
        return loss, grad

    def spmm(self, X: torch.Tensor, add: torch.Tensor | None = None, transpose: bool = False,
             tiled: bool = False) -> torch.Tensor:
        """Z = A X (+ add); width of X must be 16, 32, 64 or 128 (tiled: 64)."""
        X = X.contiguous()
        Z = torch.empty_like(X)
        v = self.val if (not transpose or self.valT is None) else self.valT
        tiles, ntiles, tile_rows = self.tiles(1) if tiled else (None, 0, 0)
        with torch.cuda.device(self.device):
            check(lib.sparc_spmm(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(v), int(X.shape[1]),
                                 _ptr(X), _ptr(add), _ptr(Z), _ptr(tiles), ntiles, tile_rows,
                                 int(transpose and self.valT is not None), _stream()), "sparc_spmm")
        return Z

    def nbytes(self) -> int:
        t = [self.rowptr, self.colidx, self.val, self.rowc, self.d, self.cw, self.rho]
        if self.valT is not None:
            t.append(self.valT)
        return sum(x.numel() * x.element_size() for x in t)
