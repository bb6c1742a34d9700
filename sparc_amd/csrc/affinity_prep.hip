// One-time per-batch preparation of the affinity operator
//     W = diag(1/rho) (S + U C V^T - diag(g)),  S = A + A^2 + A^3.
// The explicit rows of S are needed only for the row norms rho and the removed
// diagonal g; they are formed 64 columns at a time with the row-gather product
// (S[:, J] = A (E_J + A (E_J + A E_J))) and reduced on the fly, so nothing
// m x m is ever stored.  All reductions are float64 with a fixed order.
#include "spmm.h"

namespace {

constexpr int PREP_W = 64;

// r = A 1 and c = A^T 1 (row sums of val / valT); iso[i] = 1 if row i has no off-diagonal entry
__global__ void rowsums(int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                        const float* __restrict__ val, const float* __restrict__ valT, double* __restrict__ r,
                        double* __restrict__ c, double* __restrict__ iso) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= m) return;
    double a = 0.0, b = 0.0;
    bool off = false;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        a += val[p];
        b += valT[p];
        if (col[p] != i && (val[p] != 0.f || valT[p] != 0.f)) off = true;
    }
    r[i] = a;
    c[i] = b;
    iso[i] = off ? 0.0 : 1.0;
}

// S_ii = A_ii + (A^2)_ii + (A^3)_ii in float64 straight from the definition (closed walks of
// length 1..3); A_li is read from row i's mirrored values (the pattern is symmetric).  One wavefront per
// row: the walk i -> j is wave-uniform, the lanes share the entries of row j and each looks its l up in the
// sorted row i; the lanes' partial sums are combined in a fixed order.
__global__ __launch_bounds__(256) void diag_of_s(int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                 const float* __restrict__ val, const float* __restrict__ valT,
                                                 double* __restrict__ sdiag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (i >= m) return;
    const int64_t b0 = rowptr[i], b1 = rowptr[i + 1];
    double w1 = 0.0, w2 = 0.0, w3 = 0.0;
    for (int64_t p = b0 + lane; p < b1; p += 64) {
        const double aij = val[p];
        if (col[p] == i) w1 = aij;
        w2 += aij * static_cast<double>(valT[p]);
    }
    for (int64_t p = b0; p < b1; ++p) {  // wave-uniform
        const int64_t j = col[p];
        const double aij = val[p];
        if (aij == 0.0) continue;
        double inner = 0.0;  // this lane's share of sum_l A_jl A_li
        const int64_t q1 = rowptr[j + 1];
        for (int64_t q = rowptr[j] + lane; q < q1; q += 64) {
            const int32_t l = col[q];
            int64_t lo = b0, hi = b1;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (col[mid] < l) lo = mid + 1; else hi = mid;
            }
            if (lo < b1 && col[lo] == l) inner += static_cast<double>(val[q]) * static_cast<double>(valT[lo]);
        }
        w3 += aij * inner;
    }
    const double t = wave_sum(w1) + wave_sum(w2) + wave_sum(w3);
    if (lane == 0) sdiag[i] = t;
}

// y = add + M x with M given by (rowptr, col, v); add may be null
__global__ void spmv64(int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                       const float* __restrict__ v, const double* __restrict__ x, const double* __restrict__ add,
                       double* __restrict__ y) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= m) return;
    double a = add ? add[i] : 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) a += static_cast<double>(v[p]) * x[col[p]];
    y[i] = a;
}

// out[j] = sum_i f_j(i), one block, fixed order.  Which sums: see the callers.
template <int NOUT, typename F>
__device__ void block_sums(int64_t m, F f, double* out) {
    __shared__ double sm[NOUT][256];
    double t[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) t[j] = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) f(i, t);
#pragma unroll
    for (int j = 0; j < NOUT; ++j) sm[j][threadIdx.x] = t[j];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) sm[j][threadIdx.x] += sm[j][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < NOUT) out[threadIdx.x] = sm[threadIdx.x][0];
}

// scal[0] = s = sum r ; scal[1..2] = sum c, sum c2 ; scal[3..5] = sum c^2, sum c c2, sum c2^2
// then the 3x3 coupling C (eps-polynomial) into coupling[0..8], m and s behind it.
__global__ __launch_bounds__(256) void prep_scalars(int64_t m, double eps, const double* __restrict__ r,
                                                    const double* __restrict__ c, const double* __restrict__ c2,
                                                    double* __restrict__ scal, double* __restrict__ coupling) {
    block_sums<6>(m, [&](int64_t i, double* t) {
        const double ci = c[i], di = c2[i];
        t[0] += r[i]; t[1] += ci; t[2] += di; t[3] += ci * ci; t[4] += ci * di; t[5] += di * di;
    }, scal);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = scal[0], dm = static_cast<double>(m);
        const double e = eps, e2 = eps * eps, e3 = eps * eps * eps;
        coupling[0] = e + e2 * dm + e2 * s + e3 * dm * dm;
        coupling[1] = e + e2 * dm;
        coupling[2] = e;
        coupling[3] = e + e2 * dm;
        coupling[4] = e;
        coupling[5] = 0.0;
        coupling[6] = e;
        coupling[7] = 0.0;
        coupling[8] = 0.0;
        coupling[9] = dm;
        coupling[10] = s;
        for (int j = 11; j < 16; ++j) coupling[j] = 0.0;
    }
}

// P1 = E_J + A[:, J]: lane l of the row's wave looks column col0 + l up in the sorted row.  Rows without an
// entry in these columns are flagged as zero instead of stored.
__global__ __launch_bounds__(256) void build_p1(int64_t m, const int64_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ col, const float* __restrict__ val,
                                                int64_t col0, float* __restrict__ P1, uint8_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (row >= m) return;
    const int64_t target = col0 + lane;
    float out = (target == row) ? 1.f : 0.f;
    const int64_t b0 = rowptr[row], b1 = rowptr[row + 1];
    // the row's entries lie in [col[b0], col[b1 - 1]]: most rows miss the 64 columns altogether
    const bool near = b1 > b0 && col[b0] < col0 + PREP_W && col[b1 - 1] >= col0;
    if (near) {
        int64_t lo = b0, hi = b1;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (lo < b1 && col[lo] == target) out += val[lo];
    }
    const bool nz = __any(out != 0.f);
    if (nz) P1[row * PREP_W + lane] = out;
    if (lane == 0) flag[row] = nz ? 1 : 0;
}

// rho, g, d and x = 1/rho from the accumulated row statistics.
__global__ void prep_rows(int64_t m, const double* __restrict__ racc, const double* __restrict__ r,
                          const double* __restrict__ r2, const double* __restrict__ r3,
                          const double* __restrict__ c, const double* __restrict__ c2,
                          const double* __restrict__ scal, const double* __restrict__ coupling,
                          const double* __restrict__ sdiag, double* __restrict__ rho, double* __restrict__ g,
                          double* __restrict__ d, double* __restrict__ x) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= m) return;
    const double u[3] = {1.0, r[i], r2[i]};
    const double v[3] = {1.0, c[i], c2[i]};
    double uc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) uc[b] = u[0] * coupling[b] + u[1] * coupling[3 + b] + u[2] * coupling[6 + b];
    const double vtv[3][3] = {{static_cast<double>(m), scal[1], scal[2]}, {scal[1], scal[3], scal[4]}, {scal[2], scal[4], scal[5]}};
    const double vt1[3] = {static_cast<double>(m), scal[1], scal[2]};
    const double* a = racc + i * 8;
    const double low_diag = uc[0] * v[0] + uc[1] * v[1] + uc[2] * v[2];
    const double cross = uc[0] * a[1] + uc[1] * a[2] + uc[2] * a[3];
    double low_sq = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) low_sq += uc[p] * vtv[p][q] * uc[q];
    low_sq -= low_diag * low_diag;
    double rr = a[0] + 2.0 * cross + low_sq;
    rr = rr > 0.0 ? sqrt(rr) : 0.0;
    // F.normalize(eps=1e-12) clamps the norm.  Every off-diagonal entry carries the 1e-6 background, so only the
    // single row of a one-node batch gets here: its row is exactly zero and stays zero after the division, which
    // 1 / rho = 0 reproduces (the rounding residue of the row's terms times 1e12 would not)
    const bool clamped = !(rr > 1e-12);
    rr = clamped ? 1e-12 : rr;
    const double gi = sdiag[i] + low_diag;
    rho[i] = rr;
    g[i] = gi;
    x[i] = clamped ? 0.0 : 1.0 / rr;
    d[i] = clamped ? 0.0
                   : (((r[i] + r2[i] + r3[i]) - sdiag[i]) + ((uc[0] * vt1[0] + uc[1] * vt1[1] + uc[2] * vt1[2]) - low_diag)) / rr;
}

// utx = U^T x (3 sums)
__global__ __launch_bounds__(256) void prep_utx(int64_t m, const double* __restrict__ r, const double* __restrict__ r2,
                                                const double* __restrict__ x, double* __restrict__ utx) {
    block_sums<3>(m, [&](int64_t i, double* t) {
        const double xi = x[i];
        t[0] += xi; t[1] += r[i] * xi; t[2] += r2[i] * xi;
    }, utx);
}

__global__ void prep_finish(int64_t m, const double* __restrict__ chainT, const double* __restrict__ utx,
                            const double* __restrict__ coupling, const double* __restrict__ r,
                            const double* __restrict__ r2, const double* __restrict__ c, const double* __restrict__ c2,
                            const double* __restrict__ rho, const double* __restrict__ g, const double* __restrict__ d,
                            const double* __restrict__ x, const double* __restrict__ sdiag,
                            const double* __restrict__ iso, float* __restrict__ rowc, float* __restrict__ vec) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= m) return;
    double ctu[3];  // C^T (U^T x)
#pragma unroll
    for (int a = 0; a < 3; ++a) ctu[a] = coupling[a] * utx[0] + coupling[3 + a] * utx[1] + coupling[6 + a] * utx[2];
    const double cw = chainT[i] + (ctu[0] + c[i] * ctu[1] + c2[i] * ctu[2]) - g[i] * x[i];
    float* rc = rowc + i * SPARC_ROWC;
    rc[0] = static_cast<float>(iso[i] != 0.0 ? -x[i] : x[i]);  // sign flags a row without off-diagonal entries
    rc[1] = static_cast<float>(sdiag[i]);
    rc[2] = static_cast<float>(d[i] + cw);
    rc[3] = static_cast<float>(1.0 / sqrt(d[i]));
    rc[4] = static_cast<float>(r[i]);
    rc[5] = static_cast<float>(r2[i]);
    rc[6] = static_cast<float>(c[i]);
    rc[7] = static_cast<float>(c2[i]);
    vec[i] = static_cast<float>(d[i]);
    vec[m + i] = static_cast<float>(cw);
    vec[2 * m + i] = static_cast<float>(rho[i]);
}

struct PrepPlan {
    size_t bytes;
};
size_t prep_bytes(int64_t m) {
    const size_t vd = sparc_align_up(static_cast<size_t>(m) * sizeof(double));
    return 14 * vd + 2 * sparc_align_up(static_cast<size_t>(m)) + sparc_align_up(static_cast<size_t>(m) * 8 * sizeof(double)) +
           2 * sparc_align_up(static_cast<size_t>(m) * PREP_W * sizeof(float)) + 2 * sparc_align_up(16 * sizeof(double));
}

}  // namespace

extern "C" size_t sparc_affinity_prep_workspace_bytes(int64_t m, int64_t nnz) {
    (void)nnz;
    if (m < 0) return 0;
    return prep_bytes(m);
}

extern "C" int sparc_affinity_prep(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx,
                                   const float* val, const float* valT, double eps, float* rowc, float* vec,
                                   double* coupling, void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m <= 0 || nnz < 0 || !rowptr || !rowc || !vec || !coupling || (nnz > 0 && (!colidx || !val))) return SPARC_ERR_ARG;
    const float* vT = valT ? valT : val;
    WsCarver cv(ws, ws_bytes);
    double* r = cv.take<double>(m);
    double* c = cv.take<double>(m);
    double* r2 = cv.take<double>(m);
    double* c2 = cv.take<double>(m);
    double* r3 = cv.take<double>(m);
    double* rho = cv.take<double>(m);
    double* g = cv.take<double>(m);
    double* d = cv.take<double>(m);
    double* x = cv.take<double>(m);
    double* t1 = cv.take<double>(m);
    double* t2 = cv.take<double>(m);
    double* chainT = cv.take<double>(m);
    double* iso = cv.take<double>(m);
    double* sdiag = cv.take<double>(m);
    double* racc = cv.take<double>(static_cast<size_t>(m) * 8);
    float* P1 = cv.take<float>(static_cast<size_t>(m) * PREP_W);
    float* P2 = cv.take<float>(static_cast<size_t>(m) * PREP_W);
    double* scal = cv.take<double>(16);
    double* utx = cv.take<double>(16);
    uint8_t* flag1 = cv.take<uint8_t>(m);
    uint8_t* flag2 = cv.take<uint8_t>(m);
    if (!ws || !cv.ok()) return SPARC_ERR_WS;

    const dim3 bt(256), gt(static_cast<unsigned>((m + 255) / 256));
    hipLaunchKernelGGL(rowsums, gt, bt, 0, stream, m, rowptr, colidx, val, vT, r, c, iso);
    hipLaunchKernelGGL(diag_of_s, dim3(static_cast<unsigned>((m + 3) / 4)), bt, 0, stream, m, rowptr, colidx, val, vT, sdiag);
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, val, r, static_cast<const double*>(nullptr), r2);
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, vT, c, static_cast<const double*>(nullptr), c2);
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, val, r2, static_cast<const double*>(nullptr), r3);
    hipLaunchKernelGGL(prep_scalars, dim3(1), bt, 0, stream, m, eps, r, c, c2, scal, coupling);
    SPARC_CHECK_LAUNCH();

    if (hipMemsetAsync(racc, 0, static_cast<size_t>(m) * 8 * sizeof(double), stream) != hipSuccess) return SPARC_ERR_LAUNCH;
    SpmmArgs a = {};
    a.m = m; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = val;
    a.racc = racc; a.cvec = c; a.c2vec = c2;
    const dim3 gp(static_cast<unsigned>((m + 3) / 4));
    for (int64_t col0 = 0; col0 < m; col0 += PREP_W) {
        hipLaunchKernelGGL(build_p1, gp, bt, 0, stream, m, rowptr, colidx, val, col0, P1, flag1);
        a.col0 = col0;
        a.Xg = P1; a.out = P2; a.gflag = flag1; a.oflag = flag2;
        int rc = sparc_spmm_launch(EPI_ADD_IDENT, PREP_W, a, stream);
        if (rc) return rc;
        a.Xg = P2; a.out = nullptr; a.gflag = flag2; a.oflag = nullptr;
        rc = sparc_spmm_launch(EPI_PREP_ACC, PREP_W, a, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(prep_rows, gt, bt, 0, stream, m, racc, r, r2, r3, c, c2, scal, coupling, sdiag, rho, g, d, x);
    hipLaunchKernelGGL(prep_utx, dim3(1), bt, 0, stream, m, r, r2, x, utx);
    // chainT = A^T (x + A^T (x + A^T x))
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, vT, x, x, t1);
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, vT, t1, x, t2);
    hipLaunchKernelGGL(spmv64, gt, bt, 0, stream, m, rowptr, colidx, vT, t2, static_cast<const double*>(nullptr), chainT);
    hipLaunchKernelGGL(prep_finish, gt, bt, 0, stream, m, chainT, utx, coupling, r, r2, c, c2, rho, g, d, x, sdiag, iso, rowc, vec);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
