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
TILE_LDS_BYTES = 128 * 1024   # LDS budget of one window (rows * 64 * vec bytes)
TILE_MAX_ROWS = 512           # rows of one window
TILE_LONG_ROW = 128           # SPARC_TILE_LONG_ROW: rows with more in-window entries get a whole wavefront
TILE_MAX_RUN = 65535          # entries of one row before / behind its window
N_CU = 256


def plan_windows(m: int, cluster_sizes=None, max_rows: int = TILE_MAX_ROWS):
    """Row windows [begin, end) for the windowed kernel: consecutive clusters are merged while
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


TILE_MAX_SLOTS = 128          # slots of one workgroup (four records each: SPARC_TILE_MAX_RECORDS / 4)


def plan_tiles(windows, n_in: np.ndarray, n_out: np.ndarray, n_slabs: int, threads: int = 512,
               target_wgs: int = 2 * N_CU):
    """Work plan of the windowed kernel (layout: include/sparc_hip.h).  n_in / n_out: per row, the number of
    stored entries whose column lies inside / outside the row's window.  Every window's rows are sorted
    by n_in (longest first) and dealt round-robin to J parts so that windows x slabs x J fills the chip
    with equal work; a part's rows are cut into slots: one row of more than TILE_LONG_ROW in-window
    entries, or four shorter rows.  Returns a dict: desc int32[.][8] (one per workgroup, XCD-aligned),
    slot_rows int32[nslots][4] (-1: none), slot_L / slot_O (lanes' worth of in-window entries, four per
    lane, and of outside entries, one per lane), slot_base / slot_obase (positions in the two entry
    streams), stream_len / ostream_len, and per row its slot, DPP row and long flag."""
    nw = len(windows)
    J = max(1, target_wgs // max(nw * n_slabs, 1))
    min_rows = threads // 16                      # one row per 16-lane group of the workgroup
    m = len(n_in)
    desc, slot_rows = [], []
    for (a, b) in windows:
        rows = np.arange(a, b)
        order = rows[np.argsort(-n_in[rows], kind="stable")]
        parts = max(1, min(J, (len(order) + min_rows - 1) // min_rows))
        while True:                                # keep every part within the records a workgroup can hold
            worst = 0
            for j in range(parts):
                mine = order[j::parts]
                nl = int((n_in[mine] > TILE_LONG_ROW).sum())
                worst = max(worst, nl + (len(mine) - nl + 3) // 4)
            if worst <= TILE_MAX_SLOTS:
                break
            parts += 1
        for j in range(parts):
            mine = order[j::parts]
            if len(mine) == 0:
                continue
            nl = int((n_in[mine] > TILE_LONG_ROW).sum())
            s0 = len(slot_rows)
            for r in mine[:nl]:
                slot_rows.append((r, -1, -1, -1))
            rest = mine[nl:]
            for i in range(0, len(rest), 4):
                quad = list(rest[i:i + 4]) + [-1] * (4 - len(rest[i:i + 4]))
                slot_rows.append(tuple(quad))
            s1 = len(slot_rows)
            has_out = int(n_out[mine].max() > 0)
            for s_ in range(n_slabs):
                desc.append((s0, s1, a, b, s_, nl, has_out, 0))
    slot_rows = np.asarray(slot_rows, dtype=np.int64).reshape(-1, 4)
    ns = slot_rows.shape[0]
    first = slot_rows[:, 0]
    is_long_slot = slot_rows[:, 1] < 0
    is_long_slot &= n_in[first] > TILE_LONG_ROW
    safe = np.where(slot_rows >= 0, slot_rows, 0)
    nin_s = np.where(slot_rows >= 0, n_in[safe], 0)
    nout_s = np.where(slot_rows >= 0, n_out[safe], 0)
    groups = (nin_s.max(axis=1) + 3) // 4                                  # the first row is the longest
    slot_L = np.where(is_long_slot, (groups + 3) // 4, groups)
    slot_O = np.where(is_long_slot, (nout_s[:, 0] + 3) // 4, nout_s.max(axis=1))
    slot_base = np.concatenate([[0], np.cumsum(slot_L * 16)])
    slot_obase = np.concatenate([[0], np.cumsum(slot_O * 4)])
    row_slot = np.zeros(m, np.int64); row_g = np.zeros(m, np.int64)
    for gi in range(4):
        ok = slot_rows[:, gi] >= 0
        row_slot[slot_rows[ok, gi]] = np.nonzero(ok)[0]
        row_g[slot_rows[ok, gi]] = gi
    return {"desc": np.asarray(_xcd_align(desc), dtype=np.int32).reshape(-1, 8), "slot_rows": slot_rows.astype(np.int32),
            "slot_L": slot_L, "slot_O": slot_O, "slot_base": slot_base[:-1], "slot_obase": slot_obase[:-1],
            "stream_len": int(slot_base[-1]), "ostream_len": int(slot_obase[-1]),
            "row_slot": row_slot, "row_g": row_g, "row_long": is_long_slot[row_slot] if ns else np.zeros(m, bool)}


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
        key = (d[2], d[3])
        groups.setdefault(key, []).append(d)
    bins = [[] for _ in range(N_XCD)]
    load = [0] * N_XCD
    for key, ds in sorted(groups.items(), key=lambda kv: -(kv[0][1] - kv[0][0])):
        x = min(range(N_XCD), key=lambda i: (load[i], len(bins[i])))
        bins[x].extend(ds)
        load[x] += key[1] - key[0]
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
    _lib.init_device(device.index if device.index is not None else torch.cuda.current_device())
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
        self._blocks = None
        self._row_factors = None
        self._measured = {}
        self._window_ranges = None
        self._prepare(eps)
        return self

    def _in_window_ranges(self):
        """Per row: its stored entries [p0, p1) and the positions [lo, hi) of those whose column lies in
        the row's window; computed once on the device, kept on the host (int64 arrays)."""
        if self._window_ranges is None:
            m, dev = self.m, self.device
            wb = torch.empty(m, dtype=torch.int64, device=dev)
            we = torch.empty(m, dtype=torch.int64, device=dev)
            for a, b in self._windows:
                wb[a:b] = a
                we[a:b] = b
            p0, p1 = self.rowptr[:-1], self.rowptr[1:]
            rows = torch.repeat_interleave(torch.arange(m, device=dev), p1 - p0, output_size=self.nnz)
            keys = rows * m + self.colidx.to(torch.int64)        # ascending: CSR order
            base = torch.arange(m, device=dev, dtype=torch.int64) * m
            lo = torch.searchsorted(keys, base + wb)
            hi = torch.searchsorted(keys, base + we)
            self._window_ranges = tuple(t.cpu().numpy() for t in (p0, lo, hi, p1))
        return self._window_ranges

    # ---- tiling policy ---------------------------------------------------------------------
    tile_vec = None         # floats per lane of the windowed kernel (2: 32-float slabs, 4: 64-float slabs; None: by k)
    tile_threads = None     # workgroup size (None: 512 for vec 2, 1024 for vec 4)
    tile_target_wgs = None  # workgroups to aim for (None: two per CU for vec 2, one for vec 4)
    # A batch with a larger share of its entries outside the windows runs on the row kernel: outside entries are
    # gathered from L2 by a quarter wave each, which the row kernel does better with whole waves.  Measured on an
    # MI355X, k = 47..64: 0.5 % outside, windowed 19 us vs 39 us; 4 %, a tie; 23 %, 33 us vs 26 us.
    tile_max_outside_share = 0.08

    @classmethod
    def tile_geometry(cls, k: int, training: bool):
        """(vec, kp, n_slabs) of the windowed kernel for spectral dimension k, or None when the pass
        runs on the row kernel.  Training pass: slabs of 8 vec columns of each half, kp = k rounded up to
        that; forward-only pass: slabs of 16 vec columns, for k > 16."""
        if not cls.use_tiles or (not training and k <= 16):
            return None
        vec = cls.tile_vec
        if vec is None:     # the wide lanes when they bring no extra column padding (half as many slabs and instructions)
            narrow = 8 * 2 if training else 16 * 2
            vec = 4 if (-(-k // narrow)) % 2 == 0 else 2
        hw = 8 * vec if training else 16 * vec
        kp = -(-k // hw) * hw
        return vec, kp, kp // hw

    def tiling(self, k: int, training: bool):
        """The _lib.Tiling of this batch for the given pass (cached), or None: no windowed kernel for this
        k, too large a share of the batch's entries outside the windows, or the row kernel measured faster
        (warm(measure=True))."""
        geo = self.tile_geometry(k, training)
        if geo is None or self._measured.get((k, bool(training))) in ("rows", "blocks"):
            return None
        vec, kp, n_slabs = geo
        pattern = self.use_pattern and self.row_factors() is not None
        # a 512-row window is 64 KB at vec 2: two workgroups per CU; the weight-free training kernel needs 80
        # registers, so twelve waves per workgroup (six per SIMD) fit
        threads = self.tile_threads or ((768 if pattern and training else 512) if vec == 2 else 1024)
        target = self.tile_target_wgs or (2 * N_CU if vec == 2 else N_CU)
        key = (vec, kp, bool(training), threads, target, pattern)
        if key not in self._tiles:
            self._tiles[key] = self._plan(vec, kp, n_slabs, bool(training), threads, target, pattern=pattern)
        return self._tiles[key]

    # ---- dense diagonal blocks ------------------------------------------------------------------
    # Batches whose clusters are small, nearly complete sub-graphs (Reddit shape: 155-node clusters, 86 % of the
    # node pairs connected) multiply their diagonal blocks on the matrix cores; the row kernel adds the entries
    # that join different clusters.  Below this fill a gather moves fewer bytes than the dense block.
    block_min_fill = 0.35
    block_max_rows = 192          # P x W floats of operand + the block stripes must fit 152 KiB of LDS at W = 128
    use_blocks = True

    def blocks(self, k: int, training: bool):
        """(_lib.Blocks, remainder CSR tensors) of this batch for the given pass, or None: no cluster sizes,
        clusters too large or too sparse, or an operand width the block kernel does not take."""
        kp = 16 if k <= 16 else 32 if k <= 32 else 64      # the row kernel's column padding
        W = 2 * kp if training else kp
        if not self.use_blocks or W % 32 or self.nnz == 0:
            return None
        if self._blocks is None:
            self._blocks = self._plan_blocks() or False
        if self._blocks is False:
            return None
        bl = self._blocks
        mirrored = training and self.valT is not None
        if mirrored and kp % 32:
            return None
        lds = (bl["max_p"] * W + (2 if mirrored else 1) * bl["max_p"] * 32) * 4
        return bl if lds <= 152 * 1024 else None

    def _plan_blocks(self):
        sizes = self.cluster_sizes
        if sizes is None or int(np.sum(sizes)) != self.m:
            return None
        sizes = np.asarray([int(x) for x in sizes if int(x) > 0], dtype=np.int64)
        if len(sizes) == 0 or sizes.max() > self.block_max_rows:
            return None
        dev, m = self.device, self.m
        start = np.concatenate([[0], np.cumsum(sizes)])
        cl_of = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).to(dev)
        counts = self.rowptr[1:] - self.rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(m, device=dev), counts, output_size=self.nnz)
        col = self.colidx.to(torch.int64)
        same = cl_of[row] == cl_of[col]
        fill = float(same.sum()) / float((sizes.astype(np.float64) ** 2).sum())
        if fill < self.block_min_fill:
            return None
        P = (sizes + 31) // 32 * 32
        off = np.concatenate([[0], np.cumsum(P * P)])
        if off[-1] >= 2 ** 40:
            return None
        desc = []
        for c in range(len(sizes)):
            for st in range(int(P[c]) // 32):
                if st * 32 < sizes[c]:
                    desc.append((start[c], sizes[c], P[c], st, off[c] & 0x7fffffff, off[c] >> 31, 0, 0))
        desc_d = torch.from_numpy(np.asarray(desc, dtype=np.int32)).to(dev)
        start_d, P_d, off_d = (torch.from_numpy(x.astype(np.int64)).to(dev) for x in (start[:-1], P, off[:-1]))
        r_in, c_in = row[same], col[same]
        cl = cl_of[r_in]
        i, j = r_in - start_d[cl], c_in - start_d[cl]
        pos = off_d[cl] + torch.div(i, 32, rounding_mode="floor") * (32 * P_d[cl]) + j * 32 + i % 32
        blk = torch.zeros(int(off[-1]), dtype=torch.float32, device=dev)
        blk[pos] = self.val[same]
        blkT = None
        if self.valT is not None:
            blkT = torch.zeros_like(blk)
            blkT[pos] = self.valT[same]
        # the remaining entries as CSR
        rest = ~same
        r_rowptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        r_rowptr[1:] = torch.cumsum(torch.bincount(row[rest], minlength=m), 0)
        r_col = self.colidx[rest].contiguous()
        r_val = self.val[rest].contiguous()
        r_valT = self.valT[rest].contiguous() if self.valT is not None else None
        b = _lib.Blocks(desc_d.data_ptr(), int(desc_d.shape[0]), int(P.max()), blk.data_ptr(),
                        blkT.data_ptr() if blkT is not None else None)
        return {"struct": b, "keep": (desc_d, blk, blkT), "max_p": int(P.max()), "fill": fill, "ndesc": int(desc_d.shape[0]),
                "r_rowptr": r_rowptr, "r_col": r_col, "r_val": r_val, "r_valT": r_valT, "r_nnz": int(rest.sum())}

    use_pattern = True   # class-wide switch: weight-free gathers for supports whose off-diagonal values are row factors

    def row_factors(self):
        """float32 [m, 4] = {r, delta, delta / r, 1 / r} when every off-diagonal value of the support is its row's
        factor, A[i][j] = r_i (and the mirrored value therefore r_j): the supports src/utils.py builds from a 0/1
        adjacency (A + I, D^-1 (A + I), with or without the diagonal enhancement); delta_i = A[i][i] - r_i (0 without
        a stored diagonal).  None otherwise.  Checked exactly, once per batch."""
        if self._row_factors is None:
            self._row_factors = self._find_row_factors()
        return self._row_factors if self._row_factors is not False else None

    def _find_row_factors(self):
        if self.nnz == 0:
            return False
        dev, m = self.device, self.m
        counts = self.rowptr[1:] - self.rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(m, device=dev), counts, output_size=self.nnz)
        col = self.colidx.to(torch.int64)
        off = row != col
        inf = torch.full((m,), float("inf"), dtype=torch.float32, device=dev)
        lo = inf.clone().scatter_reduce_(0, row[off], self.val[off], "amin", include_self=True)
        hi = (-inf).scatter_reduce_(0, row[off], self.val[off], "amax", include_self=True)
        has = torch.isfinite(lo)
        diag = torch.zeros(m, dtype=torch.float32, device=dev)
        has_diag = torch.zeros(m, dtype=torch.bool, device=dev)
        diag[row[~off]] = self.val[~off]
        has_diag[row[~off]] = True
        if bool((lo[has] != hi[has]).any()):
            return False
        r = torch.where(has, lo, torch.where(has_diag & (diag != 0), diag, torch.ones_like(lo)))   # rows without neighbours
        if bool((r == 0).any()) or not bool(torch.isfinite(r).all()):
            return False
        valT = self.valT if self.valT is not None else self.val
        if bool((valT[off] != r[col[off]]).any()) or bool((valT[~off] != self.val[~off]).any()):
            return False
        delta = torch.where(has_diag, diag - r, torch.zeros_like(r))
        return torch.stack([r, delta, delta / r, 1.0 / r], dim=1).contiguous()

    # class-wide switch (measurements, tools/run_parity_exp.sh): a row's in-window entries ordered by operand-row parity.
    # Off: it takes 70 % of the LDS bank conflicts of the dual-width kernel away (SQ_LDS_BANK_CONFLICT 658 k -> 198 k
    # cycles per launch, SQ_LDS_IDX_ACTIVE 2.25 M -> 1.79 M) and the kernel takes as long as before.
    tile_parity_order = False

    def _parity_order(self, row, wcol, inside, ein, g, r_p0):
        """A new place inside its row's in-window entries for every stored entry (the other entries keep `ein`).
        With two floats per lane an operand row of the LDS window covers half of the 64 banks, the half chosen by
        the parity of its index in the window, and one ds_read_b64 serves two DPP rows (two matrix rows of a slot)
        per LDS cycle: they collide when their operand rows have the same parity.  Matrix rows on even DPP rows
        therefore take their even operand rows first, those on odd DPP rows their odd ones: the t-th entries of the
        two rows then differ in parity except where the counts do not match."""
        odd = (wcol & 1).bool()
        is_e, is_o = inside & ~odd, inside & odd
        ce, co = torch.cumsum(is_e, 0), torch.cumsum(is_o, 0)
        zero = torch.zeros(1, dtype=ce.dtype, device=ce.device)
        start = r_p0[row]                                   # first stored entry of the row
        be = torch.cat([zero, ce])[start]                   # evens / odds stored before the row begins
        bo = torch.cat([zero, co])[start]
        rank_e, rank_o = ce - 1 - be, co - 1 - bo           # place among the row's evens / odds (where it is one)
        m = self.m
        ne = torch.zeros(m, dtype=ce.dtype, device=ce.device).scatter_add_(0, row, is_e.to(ce.dtype))
        no = torch.zeros(m, dtype=ce.dtype, device=ce.device).scatter_add_(0, row, is_o.to(ce.dtype))
        evens_first = (g & 1) == 0
        new = torch.where(evens_first, torch.where(is_e, rank_e, ne[row] + rank_o), torch.where(is_o, rank_o, no[row] + rank_e))
        return torch.where(inside, new, ein)

    def _plan(self, vec, kp, n_slabs, dual, threads, target, mirrored=False, pattern=False):
        """Tiling for one kernel geometry: work descriptors and slot records (planned on the host) and the
        entry streams in consumption order (scattered on the device).  mirrored: bake the mirrored values
        into the single-width stream (for A^T X)."""
        if self.nnz >= 2 ** 31 or self.m == 0:
            return None
        rows = max(b - a for a, b in self._windows)
        if rows * 64 * vec > TILE_LDS_BYTES:
            return None
        p0, lo, hi, p1 = self._in_window_ranges()
        n_in, n_out = hi - lo, (lo - p0) + (p1 - hi)
        if n_out.sum() > self.tile_max_outside_share * max(self.nnz, 1):
            return None
        pl = plan_tiles(self._windows, n_in, n_out, n_slabs, threads, target)
        if len(pl["slot_L"]) and (pl["slot_L"].max() > 0xFFFF or pl["slot_O"].max() > 0x7FFF) or \
                max(pl["stream_len"], pl["ostream_len"]) >= 2 ** 31 - 4096:
            return None
        dev, m = self.device, self.m
        rec = np.empty((pl["slot_rows"].shape[0], 4, 4), np.int32)
        rec[:, :, 0] = pl["slot_rows"]
        rec[:, :, 1] = pl["slot_base"][:, None]
        rec[:, :, 2] = pl["slot_obase"][:, None]
        rec[:, :, 3] = (pl["slot_L"] | (pl["slot_O"] << 16))[:, None]
        buf = torch.from_numpy(np.concatenate([pl["desc"].ravel(), rec.ravel()]).astype(np.int32)).to(dev)
        # per-entry positions in the streams
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64)).to(dev)
        wb = np.empty(m, np.int64)
        for a, b in self._windows:
            wb[a:b] = a
        r_p0, r_nb, r_ni = up(p0), up(lo - p0), up(n_in)
        r_base, r_obase = up(pl["slot_base"][pl["row_slot"]]), up(pl["slot_obase"][pl["row_slot"]])
        r_g, r_long, r_w0 = up(pl["row_g"]), up(pl["row_long"]).bool(), up(wb)
        counts = self.rowptr[1:] - self.rowptr[:-1]
        row = torch.repeat_interleave(torch.arange(m, device=dev), counts, output_size=self.nnz)
        e = torch.arange(self.nnz, device=dev, dtype=torch.int64) - r_p0[row]
        ein = e - r_nb[row]
        inside = (ein >= 0) & (ein < r_ni[row])
        g, lng = r_g[row], r_long[row]
        col = self.colidx.to(torch.int64)
        if self.tile_parity_order and vec == 2:
            ein = torch.where(lng, ein, self._parity_order(row, col - r_w0[row], inside, ein, g, r_p0))
        pos_in = r_base[row] + torch.where(lng, ein, (torch.div(ein, 4, rounding_mode="floor") * 4 + g) * 4 + ein % 4)
        t = torch.where(ein < 0, e, e - r_ni[row])
        pos_out = r_obase[row] + torch.where(lng, t, t * 4 + g)
        valT = self.valT if self.valT is not None else self.val
        v_main, v_mirr = (valT, self.val) if mirrored else (self.val, valT)
        n_s, n_o = pl["stream_len"] + 16, pl["ostream_len"] + 16
        s_addr = torch.zeros(n_s, dtype=torch.int32, device=dev)
        o_row = torch.zeros(n_o, dtype=torch.int32, device=dev)
        o_val = torch.zeros(n_o, dtype=torch.float32, device=dev)
        pi, po = pos_in[inside], pos_out[~inside]
        o_row[po] = self.colidx[~inside]
        if pattern:
            # no weight streams: padding entries point at the zeroed row behind the slot's window, outside entries
            # carry a 0/1 weight
            we = np.empty(m, np.int64)
            for a, b in self._windows:
                we[a:b] = b - a
            first = np.maximum(pl["slot_rows"][:, 0], 0)
            pad = up(we[first] * (64 * vec))
            if len(pl["slot_L"]):
                s_addr[:pl["stream_len"]] = torch.repeat_interleave(pad, up(pl["slot_L"] * 16), output_size=int(pl["stream_len"])).to(torch.int32)
            s_addr[pi] = ((col[inside] - r_w0[row[inside]]) * (64 * vec)).to(torch.int32)
            o_val[po] = 1.0
            rf = self.row_factors()
            t = _lib.Tiling(buf.data_ptr(), int(pl["desc"].shape[0]), int(rows), vec, threads, kp, int(dual),
                            s_addr.data_ptr(), None, None, o_row.data_ptr(), o_val.data_ptr(), None, rf.data_ptr())
            t.keep = (buf, s_addr, o_row, o_val, rf)
            t.plan = pl
            t.desc = pl["desc"]
            t.pattern = True
            return t
        s_val = torch.zeros(n_s, dtype=torch.float32, device=dev)
        s_addr[pi] = ((col[inside] - r_w0[row[inside]]) * (64 * vec)).to(torch.int32)
        s_val[pi] = v_main[inside]
        o_val[po] = v_main[~inside]
        s_valT = o_valT = None
        if dual:
            if self.valT is None:
                s_valT, o_valT = s_val, o_val
            else:
                s_valT = torch.zeros(n_s, dtype=torch.float32, device=dev)
                o_valT = torch.zeros(n_o, dtype=torch.float32, device=dev)
                s_valT[pi] = v_mirr[inside]
                o_valT[po] = v_mirr[~inside]
        t = _lib.Tiling(buf.data_ptr(), int(pl["desc"].shape[0]), int(rows), vec, threads, kp, int(dual),
                        s_addr.data_ptr(), s_val.data_ptr(), s_valT.data_ptr() if s_valT is not None else None,
                        o_row.data_ptr(), o_val.data_ptr(), o_valT.data_ptr() if o_valT is not None else None, None)
        t.keep = (buf, s_addr, s_val, s_valT, o_row, o_val, o_valT)   # the struct points into these tensors
        t.pattern = False
        t.plan = pl
        t.desc = pl["desc"]
        return t

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

    def warm(self, k: int, training: bool = True, measure: bool = False, reps: int = 5):
        """Build the tiling for spectral dimension k now (it is otherwise built on first use).  measure: when
        the batch has a tiling, time its pass on both kernels (reps calls each, about a millisecond in all) and
        keep the faster one for this (k, training); both produce the same result to rounding, so the choice
        only moves time."""
        names = [n for n, ok in (("blocks", self.blocks(k, training) is not None), ("tiles", self.tiling(k, training) is not None),
                                 ("rows", True)) if ok]
        if len(names) > 1 and measure and (k, bool(training)) not in self._measured:
            gen = torch.Generator(device=self.device)       # a private stream of numbers: the caller's stays where it was
            gen.manual_seed(0)
            Y = torch.randn(self.m, k, device=self.device, generator=gen)
            took = {}
            for name in names:
                self._measured[(k, bool(training))] = name
                self.lap(Y, training)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(reps):
                    self.lap(Y, training)
                ev[1].record()
                ev[1].synchronize()
                took[name] = ev[0].elapsed_time(ev[1])
            self._measured[(k, bool(training))] = min(took, key=took.get)
            self.measured_ms = {n: v / reps for n, v in took.items()}
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
        choice = self._measured.get((k, bool(want_grad)))
        bl = self.blocks(k, want_grad) if choice in (None, "blocks") else None
        t = self.tiling(k, want_grad) if bl is None else None
        nbytes = lib.sparc_lap_workspace_bytes(self.m, k, t.ndesc if t is not None else 0)
        ws = Workspace.get(nbytes, self.device)
        with torch.cuda.device(self.device):
            rowc = self.rowc
            if not normalized:
                if getattr(self, "_rowc_plain", None) is None:
                    self._rowc_plain = self.rowc.clone()
                    self._rowc_plain[:, 3] = 1.0
                rowc = self._rowc_plain
            if bl is not None:
                check(lib.sparc_lap_fwd_blocks(self.m, bl["r_nnz"], _ptr(bl["r_rowptr"]), _ptr(bl["r_col"]), _ptr(bl["r_val"]),
                                               _ptr(bl["r_valT"]), _ptr(rowc), _ptr(self.coupling), k, k, _ptr(Y), _ptr(loss),
                                               _ptr(grad), ctypes.byref(bl["struct"]), _ptr(ws), nbytes, _stream()),
                      "sparc_lap_fwd_blocks")
                return loss, grad
            check(lib.sparc_lap_fwd(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(self.val),
                                    _ptr(self.valT), _ptr(rowc), _ptr(self.coupling), k, k, _ptr(Y), _ptr(loss),
                                    _ptr(grad), ctypes.byref(t) if t is not None else None, _ptr(ws), nbytes, _stream()),
                  "sparc_lap_fwd")
        return loss, grad

    def spmm(self, X: torch.Tensor, add: torch.Tensor | None = None, transpose: bool = False,
             tiled: bool = False) -> torch.Tensor:
        """Z = A X (+ add); width of X must be 16, 32, 64 or 128 (tiled: 32 or 64, one slab of the windowed
        kernel)."""
        X = X.contiguous()
        Z = torch.empty_like(X)
        v = self.val if (not transpose or self.valT is None) else self.valT
        width = int(X.shape[1])
        t = None
        if tiled:
            if width not in (32, 64):
                raise ValueError("the windowed kernel takes a plain product of width 32 or 64")
            mirrored = transpose and self.valT is not None
            key = ("spmm", width, mirrored)
            if key not in self._tiles:
                self._tiles[key] = self._plan(width // 16, width, 1, False, 512 if width == 32 else 1024,
                                              2 * N_CU if width == 32 else N_CU, mirrored=mirrored)
            t = self._tiles[key]
            if t is None:
                raise ValueError("this batch has no tiling")
        with torch.cuda.device(self.device):
            check(lib.sparc_spmm(self.m, self.nnz, _ptr(self.rowptr), _ptr(self.colidx), _ptr(v), width,
                                 _ptr(X), _ptr(add), _ptr(Z), ctypes.byref(t) if t is not None else None, _stream()),
                  "sparc_spmm")
        return Z

    def block_spmm(self, X: torch.Tensor, add: torch.Tensor | None = None) -> torch.Tensor:
        """Z = blockdiag(D_c) X (+ add): the dense-block part of the product alone (tests and measurements);
        width of X a multiple of 32 up to 128."""
        if self._blocks is None:
            self._blocks = self._plan_blocks() or False
        if self._blocks is False:
            raise ValueError("this batch has no dense blocks")
        bl = self._blocks
        X = X.contiguous()
        Z = torch.zeros_like(X)
        with torch.cuda.device(self.device):
            check(lib.sparc_blocks_spmm(self.m, int(X.shape[1]), ctypes.byref(bl["struct"]), _ptr(X), _ptr(add), _ptr(Z), _stream()),
                  "sparc_blocks_spmm")
        return Z

    def pass_kernel(self, k: int, training: bool = True) -> str:
        """Which kernels lap() runs for this (k, training): "blocks" (dense diagonal blocks + row kernel on the
        rest), "tiles" (windowed kernel) or "rows"."""
        choice = self._measured.get((k, bool(training)))
        if choice in (None, "blocks") and self.blocks(k, training) is not None:
            return "blocks"
        return "tiles" if self.tiling(k, training) is not None else "rows"

    def nbytes(self) -> int:
        t = [self.rowptr, self.colidx, self.val, self.rowc, self.d, self.cw, self.rho]
        if self.valT is not None:
            t.append(self.valT)
        return sum(x.numel() * x.element_size() for x in t)


class RowSplit:
    """Plan of a split product (sparc_spmm_split) for a CSR pattern with hub rows: the rows of more than `limit`
    stored entries, cut into chunks of at most `limit` entries.  Built once per graph from its row pointers, on
    the device."""

    def __init__(self, rowptr: torch.Tensor, limit: int = 2048):
        if limit < 64:
            raise ValueError("the row limit of a split product is at least one wavefront of entries (64)")
        self.limit = int(limit)
        dev = rowptr.device
        deg = rowptr[1:] - rowptr[:-1]
        self.long_row = torch.nonzero(deg > limit).flatten().to(torch.int32)
        self.nlong = int(self.long_row.numel())
        ldeg = deg[self.long_row.long()]
        per = torch.div(ldeg + (limit - 1), limit, rounding_mode="floor")          # chunks of every long row
        self.long_first = torch.zeros(self.nlong + 1, dtype=torch.int64, device=dev)
        self.long_first[1:] = torch.cumsum(per, 0)
        self.nchunks = int(self.long_first[-1]) if self.nlong else 0
        owner = torch.repeat_interleave(torch.arange(self.nlong, device=dev), per, output_size=self.nchunks)
        within = torch.arange(self.nchunks, device=dev, dtype=torch.int64) - self.long_first[owner]
        start = rowptr[self.long_row.long()][owner]
        # equal chunks (the last of a row no shorter than the others by more than one entry per chunk)
        size = torch.div(ldeg[owner] + per[owner] - 1, per[owner], rounding_mode="floor")
        self.chunk_begin = (start + within * size).contiguous()
        self.chunk_end = torch.minimum(self.chunk_begin + size, rowptr[self.long_row.long() + 1][owner]).contiguous()
        self._partial = None

    def partial(self, width: int) -> torch.Tensor:
        n = max(self.nchunks, 1) * width
        if self._partial is None or self._partial.numel() < n:
            self._partial = torch.empty(n, dtype=torch.float32, device=self.long_first.device)
        return self._partial


def spmm_split(plan: RowSplit, m: int, rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, X: torch.Tensor,
               add: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Z = A X (+ add) with the long rows of the plan in chunks; add may be out itself."""
    require_gpu(X.device)
    X = X.contiguous()
    Z = torch.empty_like(X) if out is None else out
    width = int(X.shape[1])
    with torch.cuda.device(X.device):
        check(lib.sparc_spmm_split(m, int(colidx.numel()), _ptr(rowptr), _ptr(colidx), _ptr(val), width, _ptr(X), _ptr(add), _ptr(Z),
                                   plan.limit, plan.nlong, _ptr(plan.long_row), _ptr(plan.long_first), plan.nchunks,
                                   _ptr(plan.chunk_begin), _ptr(plan.chunk_end), _ptr(plan.partial(width)), _stream()),
              "sparc_spmm_split")
    return Z
