/*
 * sparc_hip.h -- C ABI of libsparc_hip.so: the MI355X (gfx950) kernels behind the
 * SpectralNet training step of SPARC.
 *
 * The reference implements this path in Python on dense torch tensors and has no
 * foreign-function interface of its own; each entry point below names the lines
 * of the reference (relative to its checkout) whose work it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - every function enqueues work on `stream` and returns at once: 0 on success,
 *     a negative SPARC_ERR_* code otherwise; nothing throws, nothing synchronises
 *     except where stated, nothing allocates: scratch memory is passed in through
 *     `ws` (size from the matching *_workspace_bytes query, 256-byte aligned);
 *   - matrices are row-major float unless stated; k is the spectral dimension
 *     (<= 64); m is the number of nodes of one cluster batch;
 *   - calls on distinct streams with distinct workspaces may run concurrently.
 *
 * Sparse format ("dual-value CSR").  A batch's support matrix A is stored once:
 *     rowptr  int64[m+1], colidx int32[nnz] (sorted inside a row),
 *     val     float[nnz]  = A[i, j]      for the entry (i, j = colidx[p]),
 *     valT    float[nnz]  = A[j, i]      (the mirrored entry; 0 if absent).
 * The sparsity pattern is symmetrised at build time, so A^T x is evaluated with the
 * same row gather as A x, using valT instead of val: no transposed copy exists.
 * valT may be NULL when the matrix is numerically symmetric.
 */
#ifndef SPARC_HIP_H
#define SPARC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sparc_stream_t; /* == hipStream_t */

#define SPARC_OK 0
#define SPARC_ERR_ARG (-1)     /* bad argument (NULL pointer, k out of range, ...) */
#define SPARC_ERR_WS (-2)      /* workspace too small or misaligned */
#define SPARC_ERR_LAUNCH (-3)  /* the HIP runtime refused a launch */
#define SPARC_ERR_SIZE (-4)    /* the problem is larger than a fixed-size entry point supports */

#define SPARC_MAX_K 64
/* per-row constants used by the per-step kernels, float[m][8] */
#define SPARC_ROWC 8 /* {+-1/rho, S_ii, d + cw, 1/sqrt(d), r, r2, c, c2}; 1/rho < 0 flags an isolated row */

int sparc_abi_version(void);
/* Per-device set-up, to be called once with the device current before its first use from code that may not query
 * the runtime later (HIP graph capture): looks up the device addresses of the arrays that are part of the library
 * image (the scratch of sparc_gram_static).  No counterpart in the reference.  Calling it again is free. */
int sparc_init(void);

/* ---- support matrix: COO tuple -> dual-value CSR -----------------------------------
 * Replaces SpectralTrainer._get_support_matrix (src/SpectralTrainer.py:275-281): the
 * reference casts the float64 values to float32 and densifies the COO (repeated
 * coordinates add up) on every step; here the tuple is sorted and merged once.
 * coords is int32[nnz_in][2] = (row, col) as written by to_tuple (src/utils.py:41-47).
 * Outputs must have room for 2*nnz_in entries (the symmetrised pattern); the number
 * actually used is written to *nnz_out, and *is_symmetric is 1 iff val == valT
 * everywhere.  Both are device words; read them back after synchronising. */
size_t sparc_csr_build_workspace_bytes(int64_t m, int64_t nnz_in);
int sparc_csr_build(int64_t m, int64_t nnz_in, const int32_t* coords, const double* values,
                    int64_t* rowptr, int32_t* colidx, float* val, float* valT,
                    int64_t* nnz_out, int32_t* is_symmetric,
                    void* ws, size_t ws_bytes, sparc_stream_t stream);

/* ---- affinity: the per-batch vectors of W ---------------------------------------------
 * Replaces SpectralTrainer._get_valid_affinity_matrix (src/SpectralTrainer.py:284-293)
 * and the row sums of SpectralNetLoss (src/SpectralTrainer.py:133).  With eps = 1e-6,
 * S = A + A^2 + A^3, U = [1 r r2], V = [1 c c2] (r = A1, c = A^T 1, r2 = A r,
 * c2 = A^T c) the reference's dense matrix is
 *     W = diag(1/rho) (S + U C V^T - diag(g)),
 * g the removed diagonal and rho the row 2-norms.  Run once per batch (W depends on the
 * support only).  Outputs: rowc float[m][SPARC_ROWC] as defined above; vec float[3][m] =
 * {d = W 1, cw = W^T 1, rho}; coupling double[16] = {C row-major (9), m, s = 1^T A 1, ...}. */
size_t sparc_affinity_prep_workspace_bytes(int64_t m, int64_t nnz);
int sparc_affinity_prep(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                        const float* val, const float* valT, double eps,
                        float* rowc, float* vec, double* coupling,
                        void* ws, size_t ws_bytes, sparc_stream_t stream);

/* ---- Laplacian product + spectral loss (+ gradient) ---------------------------------------
 * Replaces SpectralNetLoss.forward with is_normalized=True (src/SpectralTrainer.py:130-137)
 * and its autograd backward (:237).  Computes, with Yn = Y / sqrt(d),
 *     loss  = tr(Yn^T (1/2 (D_r + D_c) - W) Yn) / m          (== sum(W * cdist(Yn)^2) / 2m)
 *     gradY = [(d + cw) * Yn - W Yn - W^T Yn] / (m sqrt(d))   (only if gradY != NULL)
 * by three chained CSR products (Horner form of A + A^2 + A^3) whose last pass carries the
 * rank-3 correction, the diagonal removal, the row scaling, the per-row dot products of the
 * loss and the gradient in its epilogue.  Y is [m][ldy] with k <= ldy; gradY is [m][ldy].
 * loss is one device double. */
size_t sparc_lap_workspace_bytes(int64_t m, int k, int ntiles);

/* Cluster tiling (optional; NULL selects the plain row kernel).  A HOST struct that describes a work
 * plan and a copy of the batch's stored entries in the order the windowed kernel consumes them.
 *
 * The dense operand of a tiled pass is kept slab-major: slab s is a row-major [m][16 vec] matrix
 * holding columns [s hw, (s+1) hw) of the A half and, in the training pass, the same columns of
 * the A^T half behind them (hw = 8 vec columns per half; forward-only pass: hw = 16 vec columns).
 *
 * buf (device int32) = ndesc descriptors {slot_begin, slot_end, win_begin, win_end, slab, n_long, has_outside, 0}
 * followed by the slot records, four per slot: {row, s_pos, o_pos, L | O << 16}.
 * A workgroup stages rows [win_begin, win_end) of slab `slab` in LDS and works through the slots
 * [slot_begin, slot_end) (at most SPARC_TILE_MAX_RECORDS / 4), one wavefront per slot; has_outside
 * must be 1 if any of these slots has O > 0 (0 lets the workgroup skip the outside path).  A slot holds four
 * rows of the window, one per 16-lane group of the wavefront (record g = row of group g, -1: none), or,
 * for the first n_long slots of a workgroup, ONE row (record 0) that all four groups share (the planner
 * of this package gives rows with more than SPARC_TILE_LONG_ROW in-window entries a slot of their own).  Entries
 * whose column lies inside the window are gathered from LDS, the others from global memory:
 *   s_addr / s_val / s_valT [s_pos + 16 j + 4 g + q], j < L, q < 4: entry 4 j + q of row g (four-row
 *       slot) or entry 16 j + 4 g + q of the row (one-row slot); s_addr = (column - win_begin) * 64 vec,
 *       the byte offset of the operand row inside the LDS window;
 *   o_row / o_val / o_valT [o_pos + 4 j + g], j < O: outside entry j of row g, or outside entry 4 j + g
 *       of the row; o_row = column.
 * Both streams are padded with zero-valued entries up to the slot's L and O (the same for its four
 * records; the kernel reads them from record 0's group); s_pos is a multiple of 4, and the streams
 * must be readable 16 entries past the last slot.  s_valT / o_valT are used by the training pass
 * only (the mirrored values A[j, i]; they may alias s_val / o_val for a symmetric matrix).  Slots of
 * a workgroup are ordered by decreasing work; every (row, slab) pair must be produced exactly once;
 * Pattern mode (s_val == NULL; for matrices whose off-diagonal values are all A[i][j] = r_i, which is what
 * src/utils.py builds from a 0/1 adjacency: A + I, D^-1 (A + I), with or without the diagonal enhancement):
 * there are no in-window weight streams; padding entries of s_addr point at the row behind the window
 * (s_addr = window rows * 64 vec), o_val is 1 for an entry and 0 for padding, and row_scale holds per row
 * r, delta = A[i][i] - r (0 without a stored diagonal), delta / r and 1 / r.
 * a descriptor with slot_begin == slot_end is an empty workgroup (padding that lets a caller place
 * the workgroups of one window on one XCD: workgroup b runs on XCD b % 8).  kp (a multiple of hw,
 * >= k) is the padded number of columns per half; rows = the largest window, rows * 64 * vec <=
 * 128 KiB. */
typedef struct sparc_tiling {
    const int32_t* buf; /* device */
    int32_t ndesc;
    int32_t rows;
    int32_t vec;     /* floats per lane, 2 or 4 */
    int32_t threads; /* workgroup size: a multiple of 64, at most 1024 */
    int32_t kp;
    int32_t dual;    /* 1: planned for the training pass (loss + gradient), 0: forward only */
    const int32_t* s_addr; /* device: the entry streams */
    const float* s_val;
    const float* s_valT;
    const int32_t* o_row;
    const float* o_val;
    const float* o_valT;
    const float* row_scale; /* device float[m][4] = {r, delta, delta / r, 1 / r}: pattern mode (s_val == NULL), else NULL */
} sparc_tiling_t;
#define SPARC_TILE_LONG_ROW 128
#define SPARC_TILE_MAX_RECORDS 512

int sparc_lap_fwd(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                  const float* val, const float* valT, const float* rowc, const double* coupling,
                  int k, int64_t ldy, const float* Y, double* loss, float* gradY,
                  const sparc_tiling_t* tiling,
                  void* ws, size_t ws_bytes, sparc_stream_t stream);

/* Dense diagonal blocks (optional; for batches whose clusters are small, nearly complete sub-graphs).
 * The matrix is split as A = blockdiag(D_c) + C: cluster c is rows/columns [first, first + n) of the batch,
 * D_c its n x n block padded with zeros to P x P (P = n rounded up to 32), and C the entries that join
 * different clusters.  blk (device float32) holds the blocks stripe by stripe in k-major order: entry
 * (i, j) of block c sits at offset_c + (i / 32) * (32 P) + j * 32 + i % 32; blkT holds the blocks of the
 * mirrored values A[j, i] in the same layout (NULL or == blk for a numerically symmetric matrix).
 * desc (device int32[ndesc][8]) = one record per (cluster, stripe of 32 rows):
 * {first, n, P, stripe, offset_c & 0x7fffffff, offset_c >> 31, 0, 0}; max_p = the largest P.  The operand
 * width W = 2 kp (training) or kp (forward only) must be a multiple of 32, kp a multiple of 32 when blkT
 * is used, and (max_p W + 32 max_p (2 with blkT, else 1)) * 4 bytes must fit 152 KiB of LDS.
 * sparc_lap_fwd_blocks is sparc_lap_fwd on that split: rowptr / colidx / val / valT describe C only
 * (rows of C may be empty), every pass multiplies the blocks on the matrix cores
 * (v_mfma_f32_32x32x2_f32) and the row kernel adds the gathers of C.  Reference semantics as sparc_lap_fwd
 * (src/SpectralTrainer.py:116-139, :284-293). */
typedef struct sparc_blocks {
    const int32_t* desc; /* device */
    int32_t ndesc;
    int32_t max_p;
    const float* blk;    /* device */
    const float* blkT;   /* device or NULL */
} sparc_blocks_t;

int sparc_lap_fwd_blocks(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                         const float* val, const float* valT, const float* rowc, const double* coupling,
                         int k, int64_t ldy, const float* Y, double* loss, float* gradY,
                         const sparc_blocks_t* blocks,
                         void* ws, size_t ws_bytes, sparc_stream_t stream);

/* Z = blockdiag(D_c) X (+ X0 if X0 != NULL), X, X0, Z [m][width], width a multiple of 32 up to 128:
 * the dense-block part of a product on its own (tests and measurements). */
int sparc_blocks_spmm(int64_t m, int width, const sparc_blocks_t* blocks, const float* X, const float* X0,
                      float* Z, sparc_stream_t stream);

/* Plain product Z = A X (+ X0 if X0 != NULL) on the same kernels, X, X0, Z [m][width] with
 * width in {16, 32, 64, 128}; pass the mirrored values as `val` for A^T X.  With a tiling
 * (dual = 0, kp = width = 16 vec: one slab, which is the row-major operand itself) the windowed
 * kernel runs and takes the values from the tiling's streams (built from the mirrored values
 * for A^T X).  X0 may be Z itself (a row's own operand is read before the row is written, by the
 * lanes that write it), which accumulates a product pass by pass; X must not overlap Z.  Exposed
 * for roofline measurements on whole graphs. */
int sparc_spmm(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
               const float* val, int width, const float* X, const float* X0, float* Z,
               const sparc_tiling_t* tiling, sparc_stream_t stream);
/* Z = A X (+ X0) for a matrix with hub rows: every row of more than split_above (>= 64) stored entries is cut into
 * chunks that get a wavefront each, and the chunks' partial sums are added in chunk order (with the row's X0), so
 * a hub row no longer keeps one wavefront busy long after the rest of the launch has drained, and the result does
 * not depend on the schedule.  The reference has no sparse product at all (it multiplies the dense m x m matrices,
 * src/SpectralTrainer.py:289); this is the whole-graph product of the SpMM roofline stress case (R-MAT graph, 10 M nodes, 500 M edges, k = 64).
 * The caller supplies the plan, all device arrays: long_row[nlong], the rows above the limit; the chunks of
 * long_row[i] are c in [long_first[i], long_first[i + 1]) (long_first has nlong + 1 entries, long_first[nlong] ==
 * nchunks) and chunk c covers the stored entries [chunk_begin[c], chunk_end[c]), which together are exactly the
 * row's entries in order; partial is scratch for nchunks * width floats.  X0 may be Z (in-place accumulation over
 * column blocks).  width: 16, 32, 64 or 128.  sparc_amd.graph.RowSplit builds the plan from rowptr. */
int sparc_spmm_split(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                     int width, const float* X, const float* X0, float* Z, int64_t split_above, int64_t nlong,
                     const int32_t* long_row, const int64_t* long_first, int64_t nchunks,
                     const int64_t* chunk_begin, const int64_t* chunk_end, float* partial, sparc_stream_t stream);

/* Measurement aid (no counterpart in the reference): between begin and end, every launch of the
 * row-gather kernel with the given epilogue (1 = the two inner passes of sparc_lap_fwd and
 * sparc_spmm with X0, 0 = sparc_spmm without X0, 4 / 3 = the last pass with / without gradient)
 * is bracketed by HIP events on its stream, up to max_launches event pairs (the two inner passes of one
 * sparc_lap_fwd call share a pair).  sparc_profile_end waits for them and returns the summed elapsed
 * milliseconds and the number of kernel launches they cover (host pointers).
 * Not thread-safe; allocates and frees the events in begin / end, never in a launch path. */
int sparc_profile_begin(int epilogue, int max_launches);
int sparc_profile_end(double* total_ms, int64_t* launches);

/* ---- orthonormalisation layer ------------------------------------------------------------------
 * Replaces SpectralNetModel._make_orthonorm_weights (src/SpectralTrainer.py:42-70): the
 * reference takes R from a Householder QR of Yt [m][k], multiplies its columns by
 * sign(diag R) and returns sqrt(m) inverse(R).  That equals sqrt(m) S Rc^-1 S with Rc the
 * Cholesky factor (upper, positive diagonal) of the Gram matrix Yt^T Yt and S the signs of
 * the Householder diagonal; S is reproduced from the k rows `top_idx` (the first k rows of
 * the reference's permuted batch; NULL = rows 0..k-1).
 *
 * sparc_gram:         G = A^T B, double [k][k], A and B float [m][ld*]; MFMA f64, fixed
 *                     summation order (bitwise reproducible).
 * sparc_orth_weights: mode 0 = reference (with S), 1 = S = I.  Wo float[k][k]; saved for the
 *                     backward: Rc, Rinv double[k][k], signs double[k]; *info (device int) is
 *                     set to j+1 if the Gram matrix is not positive definite at pivot j, to 0
 *                     otherwise (always written: no need to clear it).
 * sparc_orth_apply:   Z = Yt Wo (transpose_w = 0) or Yt Wo^T (1): SpectralNetModel.forward's
 *                     last line (src/SpectralTrainer.py:103) and its backward wrt Yt.
 * sparc_orth_weights_bwd: given gWo = d loss / d Wo (double [k][k]) returns H = Gbar + Gbar^T
 *                     (float [k][k]) such that d loss / d Yt = Yt H: the backward of the
 *                     reference's qr -> R@D -> inverse chain (autograd, :237). */
size_t sparc_gram_workspace_bytes(int64_t m, int k);
int sparc_gram(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb,
               double* G, void* ws, size_t ws_bytes, sparc_stream_t stream);
/* sparc_gram and sparc_orth_apply on the second matrix in one launch: G = A^T B and Z = B W (transpose_w = 0) or B W^T (1),
 * W float [k][k], Z float [m][ldz].  The backward of Y = Yt Wo (src/SpectralTrainer.py:103 under loss.backward(), :237)
 * needs both from the same pair: d Wo = Yt^T gY and d Yt = gY Wo^T.  Bitwise the results of the two separate calls. */
int sparc_gram_apply(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G,
                     const float* W, int transpose_w, float* Z, int64_t ldz, void* ws, size_t ws_bytes,
                     sparc_stream_t stream);
/* G = the sum of nparts partial Gram matrices double [nparts][k][k], in a fixed order (the second half of sparc_gram,
 * for producers that leave the partial matrices themselves: sparc_linear_narrow_fwd_gram). */
int sparc_gram_reduce(int k, int nparts, const double* partials, double* G, sparc_stream_t stream);
/* sparc_gram without a workspace argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory or carry a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels,
 * same fixed summation order and bitwise the same G as sparc_gram; the partial Gram matrices go
 * to a 4 MiB array that is part of the library image on every GPU (128 blocks x 64 x 64 doubles),
 * which covers every m and every k <= SPARC_MAX_K; anything beyond returns SPARC_ERR_SIZE.
 * sparc_init() looks the array's address up on the current device; call it once per device before
 * the first sparc_gram_static that runs under stream capture (without it the first call on a
 * device does the look-up itself, a runtime query that does not belong inside a capture).  No call
 * allocates.  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams or threads) race. */
int sparc_gram_static(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb,
                      double* G, sparc_stream_t stream);
int sparc_orth_weights(int64_t m, int k, const double* G, const float* Yt, int64_t ld,
                       const int32_t* top_idx, int mode, float* Wo,
                       double* Rc, double* Rinv, double* signs, int32_t* info,
                       sparc_stream_t stream);
/* sparc_orth_weights for a caller that holds the top rows' numbers on the HOST, as the reference does (the first k
 * entries of numpy's permutation, src/SpectralTrainer.py:216-224): top_idx_host[0 .. n_idx) is read during the call and
 * travels inside the kernel's arguments, so no copy to the device (and no wait for one) sits in the middle of the
 * forward pass.  n_idx <= 64; entries beyond n_idx, and row numbers outside [0, m), count as rows of zeros.  Not for
 * code under stream capture: the numbers would be frozen into the graph. */
int sparc_orth_weights_host_idx(int64_t m, int k, const double* G, const float* Yt, int64_t ld,
                                const int32_t* top_idx_host, int n_idx, int mode, float* Wo,
                                double* Rc, double* Rinv, double* signs, int32_t* info,
                                sparc_stream_t stream);
int sparc_orth_apply(int64_t m, int k, const float* Yt, int64_t ld, const float* Wo,
                     int transpose_w, float* Z, int64_t ldz, sparc_stream_t stream);
int sparc_orth_weights_bwd(int64_t m, int k, const double* gWo, const double* Rc,
                           const double* Rinv, const double* signs, int mode, float* H,
                           sparc_stream_t stream);
/* The same backward with H kept in float64, and Z = Yt H accumulated in float64 (Z float32): for an ill-conditioned
 * orthonormalisation batch the entries of H exceed those of Yt H by many orders of magnitude (they cancel), so the
 * float32 pair above loses the leading digits of d loss / d Yt (measured: 7e-5 of its largest entry at condition
 * number 1.7e4, which is 0.3 of the parameter gradient once the MLP's backward has cancelled the large directions;
 * the reference's QR backward in float32 keeps 4e-4 there).  The autograd functions of the package use this pair. */
int sparc_orth_weights_bwd_f64(int64_t m, int k, const double* gWo, const double* Rc,
                               const double* Rinv, const double* signs, int mode, double* H,
                               sparc_stream_t stream);
int sparc_orth_apply_f64(int64_t m, int k, const float* Yt, int64_t ld, const double* H,
                         float* Z, int64_t ldz, sparc_stream_t stream);

/* ---- MLP helpers ---------------------------------------------------------------------------------
 * The SpectralNet MLP is Linear+ReLU x3, Linear+Tanh (src/SpectralTrainer.py:28-39); its wide
 * products are library GEMMs.  These replace the narrow products of the forward (:96-97) and the
 * elementwise and narrow parts of the autograd backward (src/SpectralTrainer.py:237):
 * sparc_act_bwd_bias: gout = g * act'(h) with h the layer's OUTPUT (act 0: ReLU, h > 0 ? g : 0;
 *                     act 1: Tanh, g (1 - h^2)) and db[n] = column sums of gout (the bias gradient),
 *                     in one pass over g and h; all three [m][ld*], gout may alias g.
 * sparc_wgrad_small:  dW = g^T h, g [m][n_out] with n_out <= SPARC_MAX_K, h [m][n_in] with
 *                     n_in a multiple of 64 and 16-byte aligned rows: the weight gradient of the
 *                     output layer (f32 MFMA, fixed summation order).
 * sparc_first_layer_bwd: the whole backward of the first Linear+ReLU layer (its input needs no
 *                     gradient) in one pass over g and h [m][n], n a multiple of 64, 16-byte aligned
 *                     rows: with gp = g * [h > 0] (never written out), dW_db receives gp^T x, the weight
 *                     gradient [n][din] in the layout of the layer's weight (x [m][din], din < SPARC_MAX_K),
 *                     and behind it the bias gradient [n], the column sums of gp: (din + 1) n floats.
 * sparc_linear_narrow_fwd: out = act(x W^T + b) for the narrow output layer: x [m][K], K a multiple of
 *                     16 with 16-byte aligned rows, W [n_out][K], n_out <= SPARC_MAX_K and
 *                     16 (n_out rounded up to 16) (K + 4) floats within 150 KB of LDS; act 0 ReLU,
 *                     1 Tanh, 2 none (f32 MFMA, x read once).
 * sparc_dgrad_narrow_relu: dx = (g W) * [hprev > 0] and db_prev = column sums of dx: the input
 *                     gradient of the narrow output layer (g [m][n_out], W [n_out][N], N a multiple
 *                     of 64) fused with the ReLU derivative and the bias gradient of the layer
 *                     before it, whose output hprev [m][N] is this layer's input. */
size_t sparc_act_bwd_bias_workspace_bytes(int64_t m, int n);
int sparc_act_bwd_bias(int64_t m, int n, int act, const float* g, int64_t ldg, const float* h, int64_t ldh,
                       float* gout, int64_t ldo, float* db, void* ws, size_t ws_bytes, sparc_stream_t stream);
size_t sparc_wgrad_small_workspace_bytes(int n_out, int n_in);
int sparc_wgrad_small(int64_t m, int n_out, int n_in, const float* g, int64_t ldg, const float* h, int64_t ldh,
                      float* dW, void* ws, size_t ws_bytes, sparc_stream_t stream);
size_t sparc_first_layer_bwd_workspace_bytes(int din, int n);
int sparc_first_layer_bwd(int64_t m, int din, int n, const float* x, int64_t ldx, const float* g, int64_t ldg,
                          const float* h, int64_t ldh, float* dW_db, void* ws, size_t ws_bytes, sparc_stream_t stream);
/* The same without reading the layer's activations h: for a layer of at most 16 inputs whose forward was
 * sparc_linear_narrow_in_fwd, the sign of every pre-activation is recomputed from x, W [n][din] and b [n] with that
 * kernel's own sequence of fused multiply-adds (bit for bit the forward's value), which halves the traffic of the
 * pass.  Same part of the reference's backward as sparc_first_layer_bwd (autograd of src/SpectralTrainer.py:28-39
 * through loss.backward(), :237). */
int sparc_first_layer_bwd_recompute(int64_t m, int din, int n, const float* x, int64_t ldx, const float* g,
                                    int64_t ldg, const float* W, const float* b, float* dW_db, void* ws,
                                    size_t ws_bytes, sparc_stream_t stream);
int sparc_linear_narrow_fwd(int64_t m, int K, int n_out, int act, const float* x, int64_t ldx, const float* W,
                            const float* b, float* out, int64_t ldo, sparc_stream_t stream);
/* sparc_linear_narrow_fwd that also returns G = out[:gram_rows]^T out[:gram_rows] (double [n_out][n_out]): in a training
 * step the output layer's result for the orthogonality batch goes straight into the Gram matrix of
 * _make_orthonorm_weights (src/SpectralTrainer.py:96-97 then :62), and a wave's output tile sits in registers in the
 * layout the f64 matrix instruction takes, so the product rides in the forward's epilogue (bitwise the same `out`;
 * G sums the same products as sparc_gram in another fixed order).  1 <= gram_rows <= m; ws: the query's bytes,
 * 256-byte aligned (one partial matrix per workgroup).  Every wave takes one tile of 16 rows and every CU one
 * workgroup: m beyond 65536 rows (32768 for n_out > 48) returns SPARC_ERR_SIZE and the caller makes the two calls. */
size_t sparc_linear_narrow_fwd_gram_workspace_bytes(int n_out);
int sparc_linear_narrow_fwd_gram(int64_t m, int K, int n_out, int act, const float* x, int64_t ldx, const float* W,
                                 const float* b, float* out, int64_t ldo, int64_t gram_rows, double* G, void* ws,
                                 size_t ws_bytes, sparc_stream_t stream);
size_t sparc_dgrad_narrow_relu_workspace_bytes(int64_t m, int N);
int sparc_dgrad_narrow_relu(int64_t m, int n_out, int N, const float* g, int64_t ldg, const float* W,
                            const float* hprev, int64_t ldh, float* dx, int64_t ldx, float* db_prev,
                            void* ws, size_t ws_bytes, sparc_stream_t stream);
/* The same fusion for a WIDE g (K columns, a multiple of 16; N a multiple of 128; rows of g 16-byte aligned): the input
 * gradient of a hidden Linear layer with the ReLU derivative and the bias gradient of the layer before it in its
 * epilogue, a float32 GEMM on the f32 matrix cores whose m x N result leaves the registers once, already masked
 * (autograd of src/SpectralTrainer.py:28-39 under loss.backward(), :237; as a library GEMM followed by
 * sparc_act_bwd_bias the product is written, read back together with hprev and written again). */
size_t sparc_dgrad_wide_relu_workspace_bytes(int64_t m, int N);
int sparc_dgrad_wide_relu(int64_t m, int K, int N, const float* g, int64_t ldg, const float* W, const float* hprev,
                          int64_t ldh, float* dx, int64_t ldx, float* db_prev, void* ws, size_t ws_bytes,
                          sparc_stream_t stream);
/* The same product when the layer before is the network's FIRST layer (din <= 16 inputs xin [m][ldxi] that need no
 * gradient, weights W0 [N][din], bias b0, forward computed by sparc_linear_narrow_in_fwd): the masked product
 * dx = (g W) * [x W0^T + b0 > 0] is only ever used for that layer's weight and bias gradient, so it is not stored at all:
 * dW_db = [dW [N][din] | db [N]] with dW[n][j] = sum over rows of dx[row][n] xin[row][j] and db[n] = sum over rows of
 * dx[row][n] come out of the product's epilogue as din + 1 column sums per tile (the layout of sparc_first_layer_bwd's
 * result).  The ReLU derivative is recomputed with the forward's own sequence of fused multiply-adds (the same bits, the
 * same sign), so the epilogue reads nothing but the tile's rows of xin.  Replaces a library GEMM that writes m x N floats
 * and the pass of sparc_first_layer_bwd_recompute that reads them back (autograd of src/SpectralTrainer.py:28-39 under
 * loss.backward(), :237). */
size_t sparc_dgrad_wide_first_workspace_bytes(int64_t m, int N, int din);
int sparc_dgrad_wide_first(int64_t m, int K, int N, int din, const float* g, int64_t ldg, const float* W,
                           const float* xin, int64_t ldxi, const float* W0, const float* b0, float* dW_db, void* ws,
                           size_t ws_bytes, sparc_stream_t stream);

/* out = act(x W^T + b) for a narrow INPUT: x [m][K] with K <= 16 (the first layer of the MLP on a handful of node
 * features, src/SpectralTrainer.py:28-39), W [N][K] row-major, b [N], out [m][ldo] with N a multiple of 4 and out
 * 16-byte aligned; act as for sparc_linear_narrow_fwd (0 ReLU, 1 Tanh, 2 none).  The result is the traffic: one
 * store stream, the weights in registers. */
int sparc_linear_narrow_in_fwd(int64_t m, int K, int N, int act, const float* x, int64_t ldx, const float* W,
                               const float* b, float* out, int64_t ldo, sparc_stream_t stream);

/* ---- support / degree construction -----------------------------------------------------------
 * Replaces the sparse algebra of preprocess_multicluster / preprocess (src/utils.py:60-137):
 * the induced sub-block adj[pt,:][:,pt] (:83, :126), normalize_adj (:22-27),
 * normalize_adj_diag_enhance (:30-38) and to_tuple (:41-47), all in float64 with one
 * rounding per reference operation so that values, indices and degree arrays are
 * bit-identical to SciPy's (after sorting a tuple by (row, col)).
 *
 * Graph CSR: indptr int64[n+1], indices int32, data double (sorted inside a row).
 * pt int32[m] lists the batch's nodes (no repeats); local index = position in pt.
 * flavour = mode | (binarize << 4).  mode 0: the raw sub-block; mode 1: rows scaled by
 * 1/max(1, rowsum) (normalize_adj); mode 2: (D+I)^-1 (A+I) then diagonal += diag_lambda *
 * diagonal (normalize_adj_diag_enhance).  binarize replaces every stored value by 1 first:
 * that is partition_graph's intra-part adjacency (src/partition_utils.py:69-74), whose rows,
 * for pt = one part, equal the sub-block's rows, so the validation path (:113-116, :126)
 * is mode 1 or 2 with binarize set, and the training path (:83-89) is mode 0 or 2 without.
 * Pass 1 (count) writes row_nnz int64[m+1] (exclusive prefix, total in [m]) and the float64
 * degree array rowsum[m] (the rowsum of the matrix the flavour normalises by);
 * pass 2 (fill) writes coords int32[total][2] sorted by (row, col) and values double[total]. */
size_t sparc_subblock_workspace_bytes(int64_t n, int64_t m);
int sparc_subblock_count(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                         int64_t m, const int32_t* pt, int flavour,
                         int64_t* row_nnz, double* rowsum,
                         void* ws, size_t ws_bytes, sparc_stream_t stream);
int sparc_subblock_fill(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                        int64_t m, const int32_t* pt, int flavour, double diag_lambda,
                        const int64_t* row_nnz, const double* rowsum,
                        int32_t* coords, double* values,
                        void* ws, size_t ws_bytes, sparc_stream_t stream);

/* ---- token lists: the selection step of the brute-force nearest-neighbour search ------
 * Replaces the sort inside sklearn's NearestNeighbors.kneighbors as
 * implementations/SPARCphormer/utils.py:270-273 uses it (only prefixes of 2, 4, ... of the 16 384
 * neighbours are consumed, :288-296): for every row of the float64 distance matrix D [rows][n]
 * (row stride ld), out[row][i] = the ranks[i]-th smallest entry of the row (1-based ranks in
 * ascending order on the HOST, at most 32 of them, clamped to n; equal entries count separately).
 * A prefix of e neighbours is then {j : D[row][j] <= out[row][i]}.  No workspace; D is only read. */
int sparc_row_rank_values(int64_t rows, int64_t n, const double* D, int64_t ld, int nranks,
                          const int64_t* ranks, double* out, sparc_stream_t stream);

/* ---- graph partitioner: label scores of one propagation sweep -------------------------
 * The partitioner stands in for metis.part_graph (src/partition_utils.py:56); this is the per-sweep
 * kernel of its size-constrained label propagation.  Graph as CSR without self loops (rowptr int64[n+1],
 * nbr int64[E] neighbour ids, w int64[E] positive edge weights), labels int64[n] in [0, n_labels).
 * For every node: cur = total weight towards its own label; best / best_lab = the largest total
 * towards one label and that label (ties: the smallest), taken over the labels with label_ok[l] != 0
 * (uint8[n_labels]; NULL: all) and, with exclude_own, other than the node's own; best = 0 and
 * best_lab = n_labels when there is none, and for a node whose neighbours carry more than 1024
 * distinct labels (no proposal in this sweep). */
int sparc_label_scores(int64_t n, const int64_t* rowptr, const int64_t* nbr, const int64_t* w,
                       const int64_t* labels, int64_t n_labels, const uint8_t* label_ok, int exclude_own,
                       int64_t* best, int64_t* best_lab, int64_t* cur, sparc_stream_t stream);

/* ---- optimiser step ----------------------------------------------------------------
 * One update of torch.optim.Adam (defaults: amsgrad = False, maximize = False), as the reference steps it after
 * every backward (src/SpectralTrainer.py:188, :238; src/AETrainer.py likewise), for all parameter tensors of a
 * model in one launch.  tensors: a HOST array of n_tensors <= SPARC_ADAM_MAX_TENSORS records of device pointers
 * (parameter, its gradient, first and second moment; float32, n elements each); step: the number of this update,
 * counted from 1 (bias corrections 1 - beta^step).  The moments are updated in place. */
#define SPARC_ADAM_MAX_TENSORS 32
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} sparc_adam_tensor_t;
int sparc_adam_step(int n_tensors, const sparc_adam_tensor_t* tensors, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int64_t step, sparc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARC_HIP_H */
