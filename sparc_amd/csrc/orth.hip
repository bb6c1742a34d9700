// Orthonormalisation layer: Gram matrix on the f64 matrix cores, k x k Cholesky,
// triangular inverse and Householder sign recovery in LDS, the product with the
// orthonormalisation weights, and the backward of the whole chain.
#include <mutex>
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

typedef float float4_t __attribute__((ext_vector_type(4)));

#ifndef SPARC_GRAM_MAX_BLOCKS
#define SPARC_GRAM_MAX_BLOCKS 128
#endif
constexpr int GRAM_MAX_BLOCKS = SPARC_GRAM_MAX_BLOCKS;

// Z[r0 .. r1) = Q[r0 .. r1) W (or W^T) for one wavefront, 16 rows at a time, in float32 on v_mfma_f32_16x16x4: W (k x k,
// zero padded to 16 NT) in registers in the B-operand layout (lane (kk = lane >> 4, i = lane & 15) holds
// W[4 s + kk][16 t + i]), the rows of Q read from global memory in the A-operand layout (lane (kk, i): row i, columns
// 4 s + kk).  Result register q of a lane: row 4 kk + q, column 16 t + i.
// The wavefront takes the tiles that begin at r0, r0 + step, ... (step: a multiple of 16).
template <int NT>
__device__ __forceinline__ void apply_rows_f32(const int64_t r0, const int64_t r1, const int64_t step, const int k,
                                               const float* __restrict__ Q, const int64_t ldq, const float* __restrict__ W,
                                               const int transpose_w, float* __restrict__ Z, const int64_t ldz, const int lane) {
    constexpr int S = 4 * NT;
    const int i = lane & 15, kk = lane >> 4;
    float b[NT][S];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int l = 4 * s_ + kk, c = 16 * t + i;
            b[t][s_] = (l < k && c < k) ? (transpose_w ? W[c * k + l] : W[l * k + c]) : 0.f;
        }
    auto request = [&](float (&a)[S], const int64_t rb) {   // (rows past r1: zero, nothing read)
        const int64_t row = rb + i;
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int l = 4 * s_ + kk;
            a[s_] = (row < r1 && l < k) ? Q[row * ldq + l] : 0.f;
        }
    };
    auto tile = [&](const float (&a)[S], const int64_t rb) {
        float4_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b[t][s_], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = rb + 4 * kk + q;
                const int c = 16 * t + i;
                if (r < r1 && c < k) Z[r * ldz + c] = acc[t][q];   // (r < rb + 16 always)
            }
    };
    // two register sets in turn: a tile's rows are requested before the products of the tile before
    float a0[S], a1[S];
    if (r0 < r1) request(a0, r0);
    for (int64_t rb = r0; rb < r1; rb += 2 * step) {
        request(a1, rb + step);
        tile(a0, rb);
        if (rb + step >= r1) break;
        request(a0, rb + 2 * step);
        tile(a1, rb + step);
    }
}

// G_b = P[rows of block b]^T Q[rows of block b], accumulated in float64 by
// v_mfma_f64_16x16x4_f64.  Lane l feeds A[i = l&15][kk = l>>4] = P[r0+kk][16 ti + i] and
// B[kk][j = l&15] = Q[r0+kk][16 tj + j]; result register q of lane l is
// G[16 ti + (l>>4) + 4 q][16 tj + (l&15)].
// AW > 0: the workgroup has AW more wavefronts, which write Z = Q W (or Q W^T, float32) for the same rows, a tile of 16
// at a time, while the first four accumulate the Gram matrix: the backward of Y = Yt Wo needs Yt^T gY and gY Wo^T from
// the same two matrices.  (Four: 16.3 us at 24.5 k rows x 47; eight, in 166 registers, 18.1 us; separate launches 18.9.)
template <int NT, int AW = 0>
__global__ __launch_bounds__(256 + 64 * AW) void gram_partial(int64_t m, int k, const float* __restrict__ P, int64_t ldp,
                                                    const float* __restrict__ Q, int64_t ldq,
                                                    int64_t rows_per_block, double* __restrict__ slab,
                                                    const float* __restrict__ W = nullptr, int transpose_w = 0,
                                                    float* __restrict__ Z = nullptr, int64_t ldz = 0) {
    constexpr int KP = NT * 16;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const bool applies = AW > 0 && threadIdx.x >= 256;   // (wave-uniform)
    const int li = lane & 15, lk = lane >> 4;
    const int64_t base = blockIdx.x * rows_per_block;
    const int64_t end = min(m, base + rows_per_block);
    double4_t acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

    // every wave takes a contiguous quarter of the block's rows, 16 rows (four MFMA steps) at a time:
    // the 8 NT loads of a group are issued before its first product
    const int64_t per_wave = ((end - base + 3) / 4 + 3) / 4 * 4;
    const int64_t w0 = base + wave * per_wave;
    const int64_t w1 = min(end, w0 + per_wave);
    int col[NT];
    bool cok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        cok[t] = t * 16 + li < k;
        col[t] = cok[t] ? t * 16 + li : 0;
    }
    // 16 rows per group; the loads of the next group are requested before the products of the current one
    // (two register sets in turn)
    auto load_group = [&](float (&pf)[4][NT], float (&qf)[4][NT], const int64_t rb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + 4 * u + lk;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pf[u][t] = P[r * ldp + col[t]];
                qf[u][t] = Q[r * ldq + col[t]];
            }
        }
    };
    auto mul_group = [&](const float (&pf)[4][NT], const float (&qf)[4][NT]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double pa[NT], qb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pa[t] = cok[t] ? static_cast<double>(pf[u][t]) : 0.0;
                qb[t] = cok[t] ? static_cast<double>(qf[u][t]) : 0.0;
            }
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[a], qb[b], acc[a][b], 0, 0, 0);
        }
    };
    int64_t rb = w0;
    if (applies) {
        const int aw = (threadIdx.x >> 6) - 4;
        apply_rows_f32<NT>(base + 16 * aw, end, 16 * AW, k, Q, ldq, W, transpose_w, Z, ldz, lane);
        rb = w1;
    }
    if (rb + 16 <= w1) {
        float pf0[4][NT], qf0[4][NT], pf1[4][NT], qf1[4][NT];
        load_group(pf0, qf0, rb);
        for (;;) {
            const bool more1 = rb + 32 <= w1;
            if (more1) load_group(pf1, qf1, rb + 16);
            mul_group(pf0, qf0);
            rb += 16;
            if (!more1) break;
            const bool more0 = rb + 32 <= w1;
            if (more0) load_group(pf0, qf0, rb + 16);
            mul_group(pf1, qf1);
            rb += 16;
            if (!more0) break;
        }
    }
    for (; rb < w1; rb += 4) {  // last rows of the wave's range
        const int64_t r = rb + lk;
        const bool live = r < w1;
        const int64_t rc = live ? r : w1 - 1;
        double pa[NT], qb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = live && cok[t];
            const float pv = P[rc * ldp + col[t]], qv = Q[rc * ldq + col[t]];
            pa[t] = ok ? static_cast<double>(pv) : 0.0;
            qb[t] = ok ? static_cast<double>(qv) : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[a], qb[b], acc[a][b], 0, 0, 0);
    }

    // waves add into one LDS image in wave order (fixed summation order)
    __shared__ double sm[KP * KP];
    for (int w = 0; w < 4; ++w) {
        if (wave == w && !applies) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gi = a * 16 + lk + 4 * q, gj = b * 16 + li;
                        const double v = acc[a][b][q];
                        if (w == 0) sm[gi * KP + gj] = v; else sm[gi * KP + gj] += v;
                    }
        }
        __syncthreads();
    }
    double* out = slab + static_cast<int64_t>(blockIdx.x) * k * k;
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) out[e] = sm[(e / k) * KP + (e % k)];
}

// G = sum of the block partials, in a fixed order: element e is summed by GRAM_SPLIT threads (thread h takes the
// partials h, h + GRAM_SPLIT, ... in eight independent chains) and combined in thread order.
constexpr int GRAM_SPLIT = 4;
__global__ __launch_bounds__(256) void gram_reduce(int kk, int nparts, const double* __restrict__ slab, double* __restrict__ G) {
    __shared__ double part[256];
    const int local = threadIdx.x / GRAM_SPLIT, h = threadIdx.x % GRAM_SPLIT;
    const int e = blockIdx.x * (256 / GRAM_SPLIT) + local;
    double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (e < kk) {
        int b = h;
        for (; b + 7 * GRAM_SPLIT < nparts; b += 8 * GRAM_SPLIT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] += slab[static_cast<int64_t>(b + u * GRAM_SPLIT) * kk + e];
        }
        for (; b < nparts; b += GRAM_SPLIT) t[0] += slab[static_cast<int64_t>(b) * kk + e];
    }
    part[threadIdx.x] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    __syncthreads();
    if (e < kk && h == 0) {
        double sum = part[threadIdx.x];
        for (int j = 1; j < GRAM_SPLIT; ++j) sum += part[threadIdx.x + j];
        G[e] = sum;
    }
}

// 1 / sqrt(x) in float64: the hardware estimate refined by two Newton steps
__device__ __forceinline__ double fast_rsqrt(const double x) {
#ifdef SPARC_OW_FAKE
    return x * 0.5;   // timing experiment
#endif
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// 1 / x in float64: the hardware estimate refined by two Newton steps
__device__ __forceinline__ double fast_rcp(const double x) {
#ifdef SPARC_OW_FAKE
    return x * 0.25;   // timing experiment
#endif
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

constexpr int KLD = SPARC_MAX_K + 1;  // leading dimension of the k x k LDS images (bank-conflict free columns)

// C = scale * op(A) op(B) for k x k matrices in LDS (row-major images of SPARC_MAX_K rows, leading dimension KLD,
// ZERO beyond k up to the next multiple of 16) on the f64 matrix cores: wave w of the workgroup's four takes the
// 16 x 16 tiles w, w + 4, ... of the result and walks the inner dimension four at a time (v_mfma_f64_16x16x4:
// lane l feeds opA[16 ti + (l & 15)][kk + (l >> 4)] and opB[kk + (l >> 4)][16 tj + (l & 15)]; result register q of
// lane l is C[16 ti + (l >> 4) + 4 q][16 tj + (l & 15)]).  Whole tiles are written, so C is zero beyond k too.
// C may not alias A or B.  Ends with a barrier.
__device__ void mm_kk(int k, const double* A, bool ta, const double* B, bool tb, double* C, double scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (k + 15) / 16, n4 = nt * 4;   // inner steps of four (the images are zero up to 16 nt)
    for (int t = wave; t < nt * nt; t += nwaves) {
        const int ti = t / nt, tj = t % nt;
        const int ar = 16 * ti + li, bc = 16 * tj + li;
        // all operands of the tile are requested before the first product; two accumulators in turn
        double av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            av[i] = bv[i] = 0.0;
            if (i < n4) {
                const int l = 4 * i + lk;
                av[i] = ta ? A[l * KLD + ar] : A[ar * KLD + l];
                bv[i] = tb ? B[bc * KLD + l] : B[l * KLD + bc];
            }
        }
        double4_t acc0 = double4_t{0.0, 0.0, 0.0, 0.0}, acc1 = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            if (i < n4) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i + 1], bv[i + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(16 * ti + lk + 4 * q) * KLD + bc] = scale * (acc0[q] + acc1[q]);
    }
    __syncthreads();
}

// Zero the part of a k x k LDS image that mm_kk may read: rows and columns up to the next multiple of 16.
__device__ __forceinline__ void zero_image(double* M, int k) {
    const int kp = (k + 15) / 16 * 16;
    for (int e = threadIdx.x; e < kp * kp; e += blockDim.x) M[(e / kp) * KLD + e % kp] = 0.0;
}

constexpr int OW_THREADS = 512;
#ifdef SPARC_OW_STAMPS
__device__ long long g_ow_stamps[8];   // developer timeline of orth_weights_kernel (cycle counter at its phase boundaries)
#endif

// ---- blocked factorisations of a k x k LDS image, 16 columns at a time ------------------------------------------
// A column-at-a-time factorisation costs a barrier and two LDS round trips per column (1200-1450 cycles each at
// k = 47).  Here a 16 x 16 diagonal block is factored by ONE wave with the block in its registers (a column's
// elimination is one matrix instruction: no memory, no barrier), the panels beside
// it are solved by one thread per column (or row) with the block's factor read from LDS (the same address in all
// lanes: broadcast reads), and the trailing matrix is updated on the f64 matrix cores: three barriers per 16 columns.

template <int L>
__device__ __forceinline__ double readlane_d(const double x) {   // lane L of the wave, in every lane (through scalar registers)
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), L), hi = __builtin_amdgcn_readlane(__double2hiint(x), L);
    return __hiloint2double(hi, lo);
}

// The 16 x 16 diagonal block lives in the registers of ONE wave in the layout of the f64 matrix instruction's
// accumulator: register q of lane (r = lane >> 4, c = lane & 15) is element (4 q + r, c).  A column's elimination is
// then ONE v_mfma_f64_16x16x4 (a rank-one update through inner index r_J = J % 4, whose operands are exactly where
// the pivot row already sits: the lanes of DPP row r_J, register J / 4), and the serial part of a step is the pivot
// (read through scalar registers), its reciprocal and that instruction: no LDS, no barrier, ~35 instructions.

// Cholesky (upper) of the symmetric block.  Step J: with u = e(J, :), rows below J take e(I, c) -= u_I u_c / pivot and
// row J becomes the factor's row u / sqrt(pivot).  On return e holds F on and above the diagonal (below: not used);
// lane J writes rd[J] = 1 / F[J][J]; bad = 1 + the first column whose pivot is not positive, else 0.
template <int J>
__device__ __forceinline__ void chol_block(double4_t& e, double* rd, const int lane, int& bad) {
    if constexpr (J < 16) {
        constexpr int qJ = J / 4, rJ = J % 4;
        const bool src = (lane >> 4) == rJ;
        double piv = readlane_d<rJ * 16 + J>(e[qJ]);
        if (!(piv > 0.0)) {
            if (!bad) bad = J + 1;
            piv = 1e-300;
        }
        const double x = src ? e[qJ] : 0.0;                                      // B operand: u_c
        const double y = (src && (lane & 15) > J) ? -x * fast_rcp(piv) : 0.0;    // A operand: -u_I / pivot, rows I > J only
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, e, 0, 0, 0);
        const double rs = fast_rsqrt(piv);
        if (src) e[qJ] = x * rs;
        if (lane == J) rd[J] = rs;
        chol_block<J + 1>(e, rd, lane, bad);
    }
}

// The sign-tracking LU of the block, kept twice: e = the block, et = its transpose, so that column J (the
// multipliers) is a ROW of et and sits in the same lanes as row J of e.  Step J: sgn = -sign(q_JJ) (the reference's
// Householder sign), pivot q_JJ - sgn, l = column J / pivot, e -= l u^T and et -= u l^T (u = row J right of the
// diagonal).  On return e holds U on and above the diagonal and the multipliers below; lane J writes the sign of
// column J to sgn_out[J] (if that column exists: J < ncols) and the reciprocal pivot to rd[J].
template <int J>
__device__ __forceinline__ void lu_block(double4_t& e, double4_t& et, double* sgn_out, double* rd, const int lane,
                                         const int ncols) {
    if constexpr (J < 16) {
        constexpr int qJ = J / 4, rJ = J % 4;
        const int r = lane >> 4, c = lane & 15;
        const double qjj = readlane_d<rJ * 16 + J>(e[qJ]);
        const double sgn = qjj > 0.0 ? -1.0 : 1.0;
        const double ip = fast_rcp(qjj - sgn);
        if (lane == J) {
            if (J < ncols) sgn_out[J] = sgn;
            rd[J] = ip;
        }
        const bool live = r == rJ && c > J;
        const double x = live ? e[qJ] : 0.0;            // U[J][c], c > J
        const double y = live ? et[qJ] * ip : 0.0;      // L[a][J], a > J (this lane's c is the row index a)
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, x, e, 0, 0, 0);
        et = __builtin_amdgcn_mfma_f64_16x16x4f64(-x, y, et, 0, 0, 0);
        if (live) et[qJ] = y;                            // the multipliers, scaled, in both copies
        if (c == J) {
#pragma unroll
            for (int q = qJ; q < 4; ++q)
                if (4 * q + r > J) e[q] *= ip;
        }
        lu_block<J + 1>(e, et, sgn_out, rd, lane, ncols);
    }
}

// C[a][b] -= sum_l opA[a][l] opB[l][b] over the 16 inner indices of block jb, for the tiles of rows / columns
// [jb + 16, 16 nt): opA[a][l] = ta ? A[jb + l][a] : A[a][jb + l], opB[l][b] = B[jb + l][b].  upper: only the tiles on
// and above the diagonal.  One tile per wave in turn; ends with a barrier.
__device__ __forceinline__ void trailing_update(const int jb, const int nt, const double* A, const bool ta, const double* B,
                                                double* C, const bool upper) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int t0 = jb / 16 + 1, n = nt - t0;
    for (int t = wave; t < n * n; t += nwaves) {
        const int ti = t0 + t / n, tj = t0 + t % n;
        if (upper && ti > tj) continue;
        const int ar = 16 * ti + li, bc = 16 * tj + li;
        double av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = jb + 4 * i + lk;
            av[i] = ta ? A[l * KLD + ar] : A[ar * KLD + l];
            bv[i] = B[l * KLD + bc];
        }
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(16 * ti + lk + 4 * q) * KLD + bc] -= acc[q];
    }
    __syncthreads();
}

// Single workgroup of 16 waves; the two factorisations are bound by LDS latency and the barrier of every column,
// so their trailing updates give each thread at most 2 x 2 elements, fully unrolled; the inverse of the factor and
// the k x k products run on the f64 matrix cores (mm_kk).  LDS images (padded leading dimension): R (Gram, then
// its Cholesky factor, upper), X = R^-1, T (top rows of Yt), Q (scaled pivot rows during the factorisation, then
// Yt[top] X and its sign-tracking LU).  (Four waves with the trailing matrix in registers, one barrier per column
// as well: 83 us against 63 us at k = 47; a column costs 1200-1450 cycles either way: the barrier, two LDS round
// trips and the reciprocal square root chain, not the arithmetic.)
// (head: the top rows' numbers handed over inside the kernel arguments, when the caller had them on the host)
struct HeadRows {
    int32_t idx[SPARC_MAX_K];
    int given;
};

__global__ __launch_bounds__(OW_THREADS) void orth_weights_kernel(int64_t m, int k, const double* __restrict__ G,
                                                                  const float* __restrict__ Yt, int64_t ld,
                                                                  const int32_t* __restrict__ top_idx, int mode,
                                                                  float* __restrict__ Wo, double* __restrict__ Rc_out,
                                                                  double* __restrict__ Rinv_out,
                                                                  double* __restrict__ signs_out, int32_t* __restrict__ info,
                                                                  const HeadRows head) {
    __shared__ double R[SPARC_MAX_K * KLD];
    __shared__ double X[SPARC_MAX_K * KLD];
    __shared__ double Q[SPARC_MAX_K * KLD];
    __shared__ double T[SPARC_MAX_K * KLD];
    __shared__ double sg[SPARC_MAX_K];
    __shared__ double rdiag[SPARC_MAX_K];   // reciprocal pivots of the block being factored
    __shared__ int bad;
    const int tid = threadIdx.x;
#ifdef SPARC_OW_STAMPS
#define OW_STAMP(i) if (tid == 0) g_ow_stamps[i] = __builtin_readcyclecounter();
#else
#define OW_STAMP(i)
#endif
    OW_STAMP(0)
    if (tid == 0) bad = 0;
    // the kernel's reads from global memory are requested first (the Gram matrix; the top rows of Yt, which are not
    // needed before the factor has been inverted) and land while the images are cleared
    constexpr int PER = (SPARC_MAX_K * SPARC_MAX_K + OW_THREADS - 1) / OW_THREADS;
    double gv[PER];
    float tv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * OW_THREADS;
        gv[u] = e < k * k ? G[e] : 0.0;
        tv[u] = 0.f;
        if (mode == 0 && e < k * k) {
            const int64_t row = top_idx ? top_idx[e / k] : head.given ? head.idx[e / k] : e / k;
            if (row < m) tv[u] = Yt[row * ld + e % k];   // top rows of Yt
        }
    }
    zero_image(R, k);   // the products on the matrix cores read whole 16 x 16 tiles: zero beyond k
    zero_image(X, k);
    zero_image(Q, k);   // (the factor's image: its trailing updates read whole tiles too)
    zero_image(T, k);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * OW_THREADS;
        if (e < k * k) {
            R[(e / k) * KLD + e % k] = gv[u];
            T[(e / k) * KLD + e % k] = static_cast<double>(tv[u]);
        }
    }
    if (tid >= k && tid < (k + 15) / 16 * 16) R[tid * KLD + tid] = 1.0;   // identity beyond k: the last block stays regular
    __syncthreads();

    OW_STAMP(1)
    // Cholesky, upper: G = R^T R, in blocks of 16 columns (see above).  R keeps the Schur complement of the columns
    // still to come; the factor's rows go to their own image (Q doubles as that image until it is needed for the
    // top rows).
    double* F = Q;
    const int ntile = (k + 15) / 16;
    for (int jb = 0; jb < k; jb += 16) {
        if (tid < 64) {                                   // wave 0: the diagonal block in its registers
            const int r = tid >> 4, c = tid & 15;
            double4_t e;
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = R[(jb + 4 * q + r) * KLD + jb + c];
            int bd = 0;
            chol_block<0>(e, rdiag + jb, tid, bd);
#pragma unroll
            for (int q = 0; q < 4; ++q) F[(jb + 4 * q + r) * KLD + jb + c] = (4 * q + r <= c) ? e[q] : 0.0;
            if (tid == 0 && bd && !bad) bad = jb + bd;
        }
        __syncthreads();
        if (jb + 16 >= k) break;
        if (tid < k - (jb + 16)) {                        // the panel right of the block: forward substitution per column
            const int c = jb + 16 + tid;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = R[(jb + i) * KLD + c];
#pragma unroll
            for (int l = 0; l < 16; ++l) {                // (row l is final; the rows below it take their share at once:
                v[l] *= rdiag[jb + l];                    // fifteen independent updates instead of one chain per row)
#pragma unroll
                for (int i = l + 1; i < 16; ++i) v[i] -= F[(jb + l) * KLD + jb + i] * v[l];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) F[(jb + i) * KLD + c] = v[i];
        }
        __syncthreads();
        trailing_update(jb, ntile, F, true, F, R, true);   // R[a][b] -= sum_l F[jb + l][a] F[jb + l][b]
    }
    for (int e = tid; e < k * k; e += OW_THREADS) {
        const int i = e / k, j = e % k;
        R[i * KLD + j] = (i <= j) ? F[i * KLD + j] : 0.0;
    }
    __syncthreads();

    OW_STAMP(2)
    // X = R^-1 by blocks of 16: the diagonal blocks D_i^-1 first (wave b inverts block b, lane c < 16 its column c by
    // back substitution with the column in registers; the entries of R it needs are the same for all lanes: LDS
    // broadcasts), then the blocks above the diagonal on the matrix cores.  Rows beyond k are given a unit diagonal so
    // that the blocks stay regular.
    const int kp = (k + 15) / 16 * 16;
    for (int e = tid; e < k * k; e += OW_THREADS) Rc_out[e] = R[(e / k) * KLD + e % k];
    if (tid >= k && tid < kp) R[tid * KLD + tid] = 1.0;
    __syncthreads();
    if (tid < kp) sg[tid] = fast_rcp(R[tid * KLD + tid]);   // reciprocal diagonal, borrowed from the sign array
    __syncthreads();
    {
        const int lane = tid & 63, blk = tid >> 6;
        if (16 * blk < kp && lane < 16) {
            const int b0 = 16 * blk, c = lane;
            double x[16];
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int l = i + 1; l < 16; ++l) t -= R[(b0 + i) * KLD + b0 + l] * x[l];
                x[i] = t * sg[b0 + i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) X[(b0 + i) * KLD + b0 + c] = x[i];   // (zero below the diagonal by construction)
        }
    }
    __syncthreads();
    // the blocks above the diagonal, one block diagonal at a time: X_ij = -D_i^-1 (sum_{i < l <= j} R_il X_lj).  A wave
    // owns a tile; the sum leaves the matrix cores in exactly the register layout their B operand wants (register q of
    // a lane is row 4 q + (lane >> 4)), so the product with D_i^-1 follows without a trip through LDS.
    {
        const int lane = tid & 63, wave = tid >> 6, nwaves = OW_THREADS / 64;
        const int li = lane & 15, lk = lane >> 4, nt = kp / 16;
        for (int dl = 1; dl < nt; ++dl) {
            for (int t = wave; t < nt - dl; t += nwaves) {
                const int i = t, j = t + dl;
                double4_t sum = double4_t{0.0, 0.0, 0.0, 0.0};
                for (int l = i + 1; l <= j; ++l) {
                    double av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int kk = 16 * l + 4 * u + lk;
                        av[u] = R[(16 * i + li) * KLD + kk];
                        bv[u] = X[kk * KLD + 16 * j + li];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], sum, 0, 0, 0);
                }
                double dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) dv[u] = X[(16 * i + li) * KLD + 16 * i + 4 * u + lk];
                double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[u], sum[u], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) X[(16 * i + lk + 4 * q) * KLD + 16 * j + li] = -acc[q];
            }
            __syncthreads();
        }
    }

    OW_STAMP(3)
    if (mode == 0) {
        mm_kk(k, T, false, X, false, Q, 1.0);   // Q = Yt[top] X
        OW_STAMP(4)
        // Householder signs from the modified LU of Q, in blocks of 16 columns: only the sign sequence is wanted.
        // (Q is zero beyond k; a zero diagonal there gives a pivot of 1 and no multipliers.)
        for (int jb = 0; jb < k; jb += 16) {
            if (tid < 64) {                               // wave 0: the diagonal block in its registers
                const int r = tid >> 4, c = tid & 15;
                double4_t e, et;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e[q] = Q[(jb + 4 * q + r) * KLD + jb + c];
                    et[q] = Q[(jb + c) * KLD + jb + 4 * q + r];
                }
                lu_block<0>(e, et, sg + jb, rdiag + jb, tid, k - jb);
#pragma unroll
                for (int q = 0; q < 4; ++q) Q[(jb + 4 * q + r) * KLD + jb + c] = e[q];
            }
            __syncthreads();
            if (jb + 16 >= k) break;
            const int nrest = k - (jb + 16);
            if (tid < nrest) {                            // the rows of U right of the block: unit-lower solve per column
                const int c = jb + 16 + tid;
                double v[16];
#pragma unroll
                for (int a = 0; a < 16; ++a) v[a] = Q[(jb + a) * KLD + c];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
#pragma unroll
                    for (int a = i + 1; a < 16; ++a) v[a] -= Q[(jb + a) * KLD + jb + i] * v[i];
                }
#pragma unroll
                for (int a = 0; a < 16; ++a) Q[(jb + a) * KLD + c] = v[a];
            } else if (tid >= 64 && tid - 64 < nrest) {   // the multipliers below the block: one row per thread
                const int r = jb + 16 + tid - 64;
                double w[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) w[b] = Q[r * KLD + jb + b];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    w[i] *= rdiag[jb + i];
#pragma unroll
                    for (int b = i + 1; b < 16; ++b) w[b] -= w[i] * Q[(jb + i) * KLD + jb + b];
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) Q[r * KLD + jb + b] = w[b];
            }
            __syncthreads();
            trailing_update(jb, ntile, Q, false, Q, Q, false);   // Q[a][b] -= sum_l L[a][jb + l] U[jb + l][b]
        }
    } else if (tid < k) {
        sg[tid] = 1.0;
    }
    __syncthreads();

    OW_STAMP(5)
    const double sq = sqrt(static_cast<double>(m));
    for (int e = tid; e < k * k; e += OW_THREADS) {
        const int i = e / k, j = e % k;
        const double x = X[i * KLD + j];
        Wo[e] = static_cast<float>(sq * sg[i] * x * sg[j]);
        Rinv_out[e] = x;
    }
    if (tid < k) signs_out[tid] = sg[tid];
    if (tid == 0) *info = bad;
    OW_STAMP(6)
}

// Backward of Wo = sqrt(m) S Rc^-1 S through Rc = chol(G)^T:  H = Gbar + Gbar^T, so that
// d loss / d Yt = Yt H for G = Yt^T Yt.
// sixteen waves: the nine to sixteen result tiles of a product are in flight at once (four waves: 27.5 us at k = 47)
constexpr int OWB_THREADS = 1024;
template <typename TH>
__global__ __launch_bounds__(OWB_THREADS) void orth_weights_bwd_kernel(int64_t m, int k, const double* __restrict__ gWo,
                                                               const double* __restrict__ Rc, const double* __restrict__ Rinv,
                                                               const double* __restrict__ signs, int mode,
                                                               TH* __restrict__ H) {
    __shared__ double M[SPARC_MAX_K * KLD];   // Rc^-1
    __shared__ double A[SPARC_MAX_K * KLD];
    __shared__ double B[SPARC_MAX_K * KLD];
    __shared__ double L[SPARC_MAX_K * KLD];   // Rc (upper); L = Rc^T
    const int tid = threadIdx.x;
    const double sq = sqrt(static_cast<double>(m));
    zero_image(M, k);
    zero_image(A, k);
    zero_image(B, k);
    zero_image(L, k);
    __syncthreads();
    for (int e = tid; e < k * k; e += OWB_THREADS) {
        const int i = e / k, j = e % k;
        M[i * KLD + j] = Rinv[e];
        L[i * KLD + j] = Rc[e];
        const double s = (mode == 0) ? signs[i] * signs[j] : 1.0;
        A[i * KLD + j] = (i <= j) ? sq * s * gWo[e] : 0.0;  // Mbar (upper part only: M is upper triangular)
    }
    __syncthreads();
    mm_kk(k, A, false, M, true, B, 1.0);    // B = Mbar M^T
    mm_kk(k, M, true, B, false, A, -1.0);   // A = Rbar = -M^T Mbar M^T
    // Lbar = tril(Rbar^T): B[i][j] = Rbar[j][i] for i >= j
    for (int e = tid; e < k * k; e += OWB_THREADS) {
        const int i = e / k, j = e % k;
        B[i * KLD + j] = (i >= j) ? A[j * KLD + i] : 0.0;
    }
    __syncthreads();
    mm_kk(k, L, false, B, false, A, 1.0);   // A = L^T Lbar = Rc Lbar
    for (int e = tid; e < k * k; e += OWB_THREADS) {
        const int i = e / k, j = e % k;
        const double v = A[i * KLD + j];
        A[i * KLD + j] = (i > j) ? v : (i == j ? 0.5 * v : 0.0);  // Phi
    }
    __syncthreads();
    mm_kk(k, M, false, A, false, B, 1.0);   // B = M Phi
    mm_kk(k, B, false, M, true, A, 1.0);    // A = Gbar = M Phi M^T
    for (int e = tid; e < k * k; e += OWB_THREADS) {
        const int i = e / k, j = e % k;
        H[e] = static_cast<TH>(A[i * KLD + j] + A[j * KLD + i]);
    }
}

// Z[rows] = Yt[rows] W on the matrix cores, one wavefront per tile of 16 rows: W (k x k, zero padded to 16 NT) sits in
// registers in the B-operand layout (lane (kk = lane >> 4, i = lane & 15) holds W[4 s + kk][16 t + i]), the 16 rows of
// Yt are read straight from global memory in the A-operand layout (lane (kk, i): row i, columns 4 s + kk; the 16-byte
// pieces of a row are consumed by the four kk groups of the same load), no LDS, no barrier.
//   F64 = false: W = Wo or Wo^T in float32 (v_mfma_f32_16x16x4): SpectralNetModel.forward's Y = Yt Wo and its backward.
//   F64 = true:  W = H in float64, Yt widened, sums in float64 (v_mfma_f64_16x16x4), Z float32: the gradient with
//                respect to the orthonormalisation batch.  For an ill-conditioned batch the entries of H are many
//                orders of magnitude larger than those of the product (they cancel), so rounding H or the partial
//                sums to float32 would cost the gradient its leading digits.
template <int NT, bool F64>
__global__ __launch_bounds__(256) void orth_apply_mfma(int64_t m, int k, const float* __restrict__ Yt, int64_t ld,
                                                       const void* __restrict__ Wv, int transpose_w,
                                                       float* __restrict__ Z, int64_t ldz) {
    constexpr int S = 4 * NT;   // inner steps of four
    typedef typename std::conditional<F64, double, float>::type T;
    typedef typename std::conditional<F64, double4_t, float4_t>::type Acc;
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    T b[NT][S];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int l = 4 * s_ + kk, c = 16 * t + i;
            T w = 0;
            if (l < k && c < k) {
                if (F64) w = static_cast<const double*>(Wv)[l * k + c];
                else w = transpose_w ? static_cast<const float*>(Wv)[c * k + l] : static_cast<const float*>(Wv)[l * k + c];
            }
            b[t][s_] = w;
        }
    const int64_t ntiles = (m + 15) / 16;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t rt = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); rt < ntiles; rt += nwaves) {
        const int64_t row = rt * 16 + i;
        T a[S];
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int l = 4 * s_ + kk;
            a[s_] = (row < m && l < k) ? static_cast<T>(Yt[row * ld + l]) : static_cast<T>(0);
        }
        Acc acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = Acc{0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (F64) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s_], b[t][s_], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b[t][s_], acc[t], 0, 0, 0);
            }
        // result register q of a lane: row kk + 4 q (f64 instruction) or 4 kk + q (f32 instruction), column i
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = rt * 16 + (F64 ? kk + 4 * q : 4 * kk + q);
                const int c = 16 * t + i;
                if (r < m && c < k) Z[r * ldz + c] = static_cast<float>(acc[t][q]);
            }
    }
}

int gram_blocks(int64_t m) {
    int64_t nb = (m + 63) / 64;
    if (nb > GRAM_MAX_BLOCKS) nb = GRAM_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return static_cast<int>(nb);
}

template <bool F64>
int launch_orth_apply(int64_t m, int k, const float* Yt, int64_t ld, const void* W, int transpose_w, float* Z, int64_t ldz,
                      hipStream_t stream) {
    const int64_t ntiles = (m + 15) / 16;
    const dim3 g(static_cast<unsigned>(std::min<int64_t>((ntiles + 3) / 4, 1024))), b(256);
    if (k <= 16) hipLaunchKernelGGL((orth_apply_mfma<1, F64>), g, b, 0, stream, m, k, Yt, ld, W, transpose_w, Z, ldz);
    else if (k <= 32) hipLaunchKernelGGL((orth_apply_mfma<2, F64>), g, b, 0, stream, m, k, Yt, ld, W, transpose_w, Z, ldz);
    else if (k <= 48) hipLaunchKernelGGL((orth_apply_mfma<3, F64>), g, b, 0, stream, m, k, Yt, ld, W, transpose_w, Z, ldz);
    else hipLaunchKernelGGL((orth_apply_mfma<4, F64>), g, b, 0, stream, m, k, Yt, ld, W, transpose_w, Z, ldz);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

}  // namespace

extern "C" size_t sparc_gram_workspace_bytes(int64_t m, int k) {
    if (m < 0 || k < 1 || k > SPARC_MAX_K) return 0;
    return sparc_align_up(static_cast<size_t>(gram_blocks(m)) * k * k * sizeof(double));
}

namespace {
int gram_launch(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G, double* slab,
                hipStream_t stream, const float* W = nullptr, int transpose_w = 0, float* Z = nullptr, int64_t ldz = 0);

// Scratch of sparc_gram_static: the partial Gram matrices of at most GRAM_MAX_BLOCKS blocks.  Every GPU the
// library runs on holds its own copy of this array at its own address for the life of the process; in host
// code the name is only a handle for hipGetSymbolAddress, never a device pointer.
constexpr size_t GRAM_STATIC_BYTES = static_cast<size_t>(GRAM_MAX_BLOCKS) * SPARC_MAX_K * SPARC_MAX_K * sizeof(double);
static __device__ __align__(256) unsigned char scratch[GRAM_STATIC_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[MAX_DEVICES] = {};  // device address of `scratch` on each device, looked up once

// The address of `scratch` on the current device (nullptr on failure).  The first call on a device asks
// the runtime for it (sparc_init does, so that no later call has to); later calls only read the table.
void* scratch_on_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return nullptr;
        g_scratch_ptr[dev] = p;
    }
    return g_scratch_ptr[dev];
}
}  // namespace

extern "C" int sparc_init(void) {
    return scratch_on_current_device() ? SPARC_OK : SPARC_ERR_LAUNCH;
}

extern "C" int sparc_gram(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G,
                          void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !A || !B || !G || lda < k || ldb < k) return SPARC_ERR_ARG;
    if (!ws || ws_bytes < sparc_gram_workspace_bytes(m, k) || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    return gram_launch(m, k, A, lda, B, ldb, G, static_cast<double*>(ws), stream);
}

extern "C" int sparc_gram_apply(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G,
                                const float* W, int transpose_w, float* Z, int64_t ldz, void* ws, size_t ws_bytes,
                                sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !A || !B || !G || !W || !Z || lda < k || ldb < k || ldz < k) return SPARC_ERR_ARG;
    if (!ws || ws_bytes < sparc_gram_workspace_bytes(m, k) || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    return gram_launch(m, k, A, lda, B, ldb, G, static_cast<double*>(ws), stream, W, transpose_w, Z, ldz);
}

extern "C" int sparc_gram_reduce(int k, int nparts, const double* partials, double* G, sparc_stream_t stream) {
    if (k < 1 || k > SPARC_MAX_K || nparts < 1 || !partials || !G) return SPARC_ERR_ARG;
    hipLaunchKernelGGL(gram_reduce, dim3((k * k * GRAM_SPLIT + 255) / 256), dim3(256), 0, stream, k * k, nparts, partials, G);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" int sparc_gram_static(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G,
                                 sparc_stream_t stream) {
    if (m <= 0 || !A || !B || !G || lda < k || ldb < k) return SPARC_ERR_ARG;
    if (k < 1 || k > SPARC_MAX_K || sparc_gram_workspace_bytes(m, k) > GRAM_STATIC_BYTES) return SPARC_ERR_SIZE;
    void* slab = scratch_on_current_device();
    if (!slab) return SPARC_ERR_LAUNCH;
    return gram_launch(m, k, A, lda, B, ldb, G, static_cast<double*>(slab), stream);
}

namespace {
int gram_launch(int64_t m, int k, const float* A, int64_t lda, const float* B, int64_t ldb, double* G, double* slab,
                hipStream_t stream, const float* W, int transpose_w, float* Z, int64_t ldz) {
    const int nb = gram_blocks(m);
    int64_t rpb = (m + nb - 1) / nb;
    rpb = (rpb + 15) / 16 * 16;
    const int nparts = static_cast<int>((m + rpb - 1) / rpb);
    const int nt = (k + 15) / 16;
    const dim3 g(static_cast<unsigned>(nparts)), b(256);
    if (Z) {
        const dim3 b2(256 + 64 * 4);
        switch (nt) {
            case 1: hipLaunchKernelGGL((gram_partial<1, 4>), g, b2, 0, stream, m, k, A, lda, B, ldb, rpb, slab, W, transpose_w, Z, ldz); break;
            case 2: hipLaunchKernelGGL((gram_partial<2, 4>), g, b2, 0, stream, m, k, A, lda, B, ldb, rpb, slab, W, transpose_w, Z, ldz); break;
            case 3: hipLaunchKernelGGL((gram_partial<3, 4>), g, b2, 0, stream, m, k, A, lda, B, ldb, rpb, slab, W, transpose_w, Z, ldz); break;
            default: hipLaunchKernelGGL((gram_partial<4, 4>), g, b2, 0, stream, m, k, A, lda, B, ldb, rpb, slab, W, transpose_w, Z, ldz); break;
        }
    } else {
        switch (nt) {
            case 1: hipLaunchKernelGGL((gram_partial<1>), g, b, 0, stream, m, k, A, lda, B, ldb, rpb, slab); break;
            case 2: hipLaunchKernelGGL((gram_partial<2>), g, b, 0, stream, m, k, A, lda, B, ldb, rpb, slab); break;
            case 3: hipLaunchKernelGGL((gram_partial<3>), g, b, 0, stream, m, k, A, lda, B, ldb, rpb, slab); break;
            default: hipLaunchKernelGGL((gram_partial<4>), g, b, 0, stream, m, k, A, lda, B, ldb, rpb, slab); break;
        }
    }
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL(gram_reduce, dim3((k * k * GRAM_SPLIT + 255) / 256), b, 0, stream, k * k, nparts, slab, G);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_orth_weights(int64_t m, int k, const double* G, const float* Yt, int64_t ld,
                                  const int32_t* top_idx, int mode, float* Wo, double* Rc, double* Rinv,
                                  double* signs, int32_t* info, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !G || !Yt || ld < k || !Wo || !Rc || !Rinv || !signs || !info ||
        (mode != 0 && mode != 1))
        return SPARC_ERR_ARG;
    HeadRows head = {};
    hipLaunchKernelGGL(orth_weights_kernel, dim3(1), dim3(OW_THREADS), 0, stream, m, k, G, Yt, ld, top_idx, mode, Wo, Rc, Rinv,
                       signs, info, head);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" int sparc_orth_weights_host_idx(int64_t m, int k, const double* G, const float* Yt, int64_t ld,
                                           const int32_t* top_idx_host, int n_idx, int mode, float* Wo, double* Rc,
                                           double* Rinv, double* signs, int32_t* info, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !G || !Yt || ld < k || !Wo || !Rc || !Rinv || !signs || !info ||
        (mode != 0 && mode != 1) || !top_idx_host || n_idx < 0 || n_idx > SPARC_MAX_K)
        return SPARC_ERR_ARG;
    HeadRows head;
    head.given = 1;
    for (int j = 0; j < SPARC_MAX_K; ++j) {
        const int64_t r = j < n_idx ? top_idx_host[j] : m;   // beyond the list, and anything out of range: a row of zeros
        head.idx[j] = (r < 0 || r > m) ? static_cast<int32_t>(m < 0x7fffffff ? m : 0x7fffffff) : static_cast<int32_t>(r);
    }
    hipLaunchKernelGGL(orth_weights_kernel, dim3(1), dim3(OW_THREADS), 0, stream, m, k, G, Yt, ld,
                       static_cast<const int32_t*>(nullptr), mode, Wo, Rc, Rinv, signs, info, head);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" int sparc_orth_weights_bwd(int64_t m, int k, const double* gWo, const double* Rc, const double* Rinv,
                                      const double* signs, int mode, float* H, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !gWo || !Rc || !Rinv || !signs || !H) return SPARC_ERR_ARG;
    hipLaunchKernelGGL(orth_weights_bwd_kernel<float>, dim3(1), dim3(OWB_THREADS), 0, stream, m, k, gWo, Rc, Rinv, signs, mode, H);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" int sparc_orth_weights_bwd_f64(int64_t m, int k, const double* gWo, const double* Rc, const double* Rinv,
                                          const double* signs, int mode, double* H, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !gWo || !Rc || !Rinv || !signs || !H) return SPARC_ERR_ARG;
    hipLaunchKernelGGL(orth_weights_bwd_kernel<double>, dim3(1), dim3(OWB_THREADS), 0, stream, m, k, gWo, Rc, Rinv, signs, mode, H);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" int sparc_orth_apply_f64(int64_t m, int k, const float* Yt, int64_t ld, const double* H, float* Z, int64_t ldz,
                                    sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !Yt || !H || !Z || ld < k || ldz < k) return SPARC_ERR_ARG;
    return launch_orth_apply<true>(m, k, Yt, ld, H, 0, Z, ldz, stream);
}

extern "C" int sparc_orth_apply(int64_t m, int k, const float* Yt, int64_t ld, const float* Wo, int transpose_w,
                                float* Z, int64_t ldz, sparc_stream_t stream) {
    if (m <= 0 || k < 1 || k > SPARC_MAX_K || !Yt || !Wo || !Z || ld < k || ldz < k) return SPARC_ERR_ARG;
    return launch_orth_apply<false>(m, k, Yt, ld, Wo, transpose_w, Z, ldz, stream);
}

#ifdef SPARC_OW_STAMPS
extern "C" int sparc_debug_orth_stamps(long long* host_out) {   // developer builds only
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ow_stamps), 8 * sizeof(long long)) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
}
#endif
