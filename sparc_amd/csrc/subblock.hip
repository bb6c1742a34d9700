// Support / degree construction on the device: induced sub-block of a graph CSR for one
// cluster batch, with the reference's float64 normalisations applied with one rounding
// per reference operation (no fused multiply-add), so that the values are bit-identical
// to what SciPy produces on the host.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

// The compiler contracts a * b + c into a fused multiply-add by default, also across HIP's __dmul_rn / __dadd_rn
// (plain operators compiled in a header, where contraction is allowed): one rounding where the reference has two.
// The operations of this file are therefore written here, with contraction off.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double add_rn(const double a, const double b) { return a + b; }
__device__ __forceinline__ double mul_rn(const double a, const double b) { return a * b; }
__device__ __forceinline__ double div_rn(const double a, const double b) { return a / b; }

constexpr int MODE_RAW = 0, MODE_ROWNORM = 1, MODE_ENHANCE = 2;

__global__ void fill_i32(int64_t n, int32_t v, int32_t* __restrict__ a) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i < n) a[i] = v;
}

__global__ void scatter_local(int64_t m, const int32_t* __restrict__ pt, int32_t* __restrict__ inv) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i < m) inv[pt[i]] = static_cast<int32_t>(i);
}

__global__ void count_rows(int64_t m, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                           const double* __restrict__ data, const int32_t* __restrict__ pt,
                           const int32_t* __restrict__ inv, int mode, int binarize, int64_t* __restrict__ cnt,
                           double* __restrict__ rowsum) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i > m) return;
    if (i == m) { cnt[m] = 0; return; }
    const int64_t gi = pt[i];
    int64_t c = 0;
    double s = 0.0;
    bool has_diag = false;
    for (int64_t p = indptr[gi]; p < indptr[gi + 1]; ++p) {
        const int32_t gj = indices[p];
        if (inv[gj] < 0) continue;
        ++c;
        s = add_rn(s, binarize ? 1.0 : data[p]);
        if (gj == gi) has_diag = true;
    }
    if (mode == MODE_ENHANCE) {  // adj + I before the row sums
        if (!has_diag) ++c;
        s = add_rn(s, 1.0);
    }
    cnt[i] = c;
    rowsum[i] = s;
}

// One thread per row: collect (local col, value), insertion-sort by local col in place.
__global__ void fill_rows(int64_t m, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                          const double* __restrict__ data, const int32_t* __restrict__ pt,
                          const int32_t* __restrict__ inv, int mode, int binarize, double diag_lambda,
                          const int64_t* __restrict__ row_nnz, const double* __restrict__ rowsum,
                          int32_t* __restrict__ coords, double* __restrict__ values) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= m) return;
    const int64_t gi = pt[i];
    const int64_t o0 = row_nnz[i];
    double scale = 1.0;
    if (mode == MODE_ROWNORM) scale = div_rn(1.0, fmax(1.0, rowsum[i]));
    else if (mode == MODE_ENHANCE) scale = div_rn(1.0, add_rn(rowsum[i], 1e-20));
    int64_t n = 0;
    bool has_diag = false;
    for (int64_t p = indptr[gi]; p < indptr[gi + 1]; ++p) {
        const int32_t gj = indices[p];
        const int32_t lj = inv[gj];
        if (lj < 0) continue;
        double v = binarize ? 1.0 : data[p];
        if (gj == gi) {
            has_diag = true;
            if (mode == MODE_ENHANCE) v = add_rn(v, 1.0);
        }
        // insert keeping local columns ascending
        int64_t q = n;
        while (q > 0 && coords[2 * (o0 + q - 1) + 1] > lj) {
            coords[2 * (o0 + q) + 1] = coords[2 * (o0 + q - 1) + 1];
            values[o0 + q] = values[o0 + q - 1];
            --q;
        }
        coords[2 * (o0 + q) + 1] = lj;
        values[o0 + q] = v;
        ++n;
    }
    if (mode == MODE_ENHANCE && !has_diag) {
        const int32_t lj = static_cast<int32_t>(i);
        int64_t q = n;
        while (q > 0 && coords[2 * (o0 + q - 1) + 1] > lj) {
            coords[2 * (o0 + q) + 1] = coords[2 * (o0 + q - 1) + 1];
            values[o0 + q] = values[o0 + q - 1];
            --q;
        }
        coords[2 * (o0 + q) + 1] = lj;
        values[o0 + q] = 1.0;
        ++n;
    }
    for (int64_t q = 0; q < n; ++q) {
        coords[2 * (o0 + q)] = static_cast<int32_t>(i);
        double v = values[o0 + q];
        if (mode != MODE_RAW) v = mul_rn(scale, v);
        if (mode == MODE_ENHANCE && coords[2 * (o0 + q) + 1] == i) v = add_rn(v, mul_rn(diag_lambda, v));
        values[o0 + q] = v;
    }
}

size_t scan_tmp_bytes(int64_t m) {
    size_t b = 0;
    int64_t* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b, p, p, static_cast<int64_t>(0), static_cast<size_t>(m + 1), rocprim::plus<int64_t>(), nullptr);
    return b;
}

}  // namespace

extern "C" size_t sparc_subblock_workspace_bytes(int64_t n, int64_t m) {
    if (n < 0 || m < 0) return 0;
    return sparc_align_up(static_cast<size_t>(n > 0 ? n : 1) * sizeof(int32_t)) +
           sparc_align_up(static_cast<size_t>(m + 1) * sizeof(int64_t)) + sparc_align_up(scan_tmp_bytes(m));
}

extern "C" int sparc_subblock_count(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                                    int64_t m, const int32_t* pt, int flavour, int64_t* row_nnz, double* rowsum,
                                    void* ws, size_t ws_bytes, sparc_stream_t stream) {
    const int mode = flavour & 15, binarize = (flavour >> 4) & 1;
    if (n <= 0 || m <= 0 || m > n || !indptr || !indices || !data || !pt || !row_nnz || !rowsum || mode > MODE_ENHANCE)
        return SPARC_ERR_ARG;
    WsCarver c(ws, ws_bytes);
    int32_t* inv = c.take<int32_t>(static_cast<size_t>(n));
    int64_t* cnt = c.take<int64_t>(static_cast<size_t>(m + 1));
    const size_t stb = scan_tmp_bytes(m);
    void* tmp = c.take<char>(stb);
    if (!ws || !c.ok()) return SPARC_ERR_WS;
    const dim3 bt(256);
    hipLaunchKernelGGL(fill_i32, dim3(static_cast<unsigned>((n + 255) / 256)), bt, 0, stream, n, -1, inv);
    hipLaunchKernelGGL(scatter_local, dim3(static_cast<unsigned>((m + 255) / 256)), bt, 0, stream, m, pt, inv);
    hipLaunchKernelGGL(count_rows, dim3(static_cast<unsigned>((m + 1 + 255) / 256)), bt, 0, stream, m, indptr, indices, data,
                       pt, inv, mode, binarize, cnt, rowsum);
    SPARC_CHECK_LAUNCH();
    size_t sb = stb;
    if (rocprim::exclusive_scan(tmp, sb, cnt, row_nnz, static_cast<int64_t>(0), static_cast<size_t>(m + 1),
                                rocprim::plus<int64_t>(), stream) != hipSuccess)
        return SPARC_ERR_LAUNCH;
    return SPARC_OK;
}

extern "C" int sparc_subblock_fill(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                                   int64_t m, const int32_t* pt, int flavour, double diag_lambda,
                                   const int64_t* row_nnz, const double* rowsum, int32_t* coords, double* values,
                                   void* ws, size_t ws_bytes, sparc_stream_t stream) {
    const int mode = flavour & 15, binarize = (flavour >> 4) & 1;
    if (n <= 0 || m <= 0 || m > n || !indptr || !indices || !data || !pt || !row_nnz || !rowsum || !coords || !values ||
        mode > MODE_ENHANCE)
        return SPARC_ERR_ARG;
    WsCarver c(ws, ws_bytes);
    int32_t* inv = c.take<int32_t>(static_cast<size_t>(n));  // filled by sparc_subblock_count on the same workspace
    if (!ws || !c.ok()) return SPARC_ERR_WS;
    hipLaunchKernelGGL(fill_rows, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, stream, m, indptr, indices, data,
                       pt, inv, mode, binarize, diag_lambda, row_nnz, rowsum, coords, values);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
