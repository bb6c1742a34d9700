// COO support tuple -> dual-value CSR (sorted, duplicates merged, pattern symmetrised).
//
// Every input entry (i, j, v) is entered twice: as (i, j) carrying v in the "val" slot
// and as (j, i) carrying v in the "valT" slot.  One stable radix sort by the key i*m + j
// brings all contributions to a coordinate together; heads of equal-key runs become the
// CSR entries.  The result holds A[i,j] and A[j,i] side by side under one column index,
// which is what lets the transposed product reuse the row gather.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

__global__ void make_keys(int64_t m, int64_t nnz, const int32_t* __restrict__ coords, int64_t* __restrict__ keys,
                          int32_t* __restrict__ idx) {
    const int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (e >= nnz) return;
    const int64_t i = coords[2 * e], j = coords[2 * e + 1];
    keys[e] = i * m + j;
    idx[e] = static_cast<int32_t>(e);
    keys[nnz + e] = j * m + i;
    idx[nnz + e] = static_cast<int32_t>(nnz + e);
}

__global__ void head_flags(int64_t n, const int64_t* __restrict__ keys, int32_t* __restrict__ flag) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// One thread per run head: sums the float32-cast values of the run in sorted (= input) order.
__global__ void emit_entries(int64_t m, int64_t nnz, int64_t n, const int64_t* __restrict__ keys,
                             const int32_t* __restrict__ idx, const int32_t* __restrict__ flag,
                             const int32_t* __restrict__ seg, const double* __restrict__ values,
                             int32_t* __restrict__ colidx, float* __restrict__ val, float* __restrict__ valT,
                             int64_t* __restrict__ ukeys, int64_t* __restrict__ nnz_out,
                             int32_t* __restrict__ is_symmetric) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *nnz_out = static_cast<int64_t>(seg[i]);  // seg = inclusive count of run heads
    if (!flag[i]) return;
    const int64_t key = keys[i];
    float a = 0.f, b = 0.f;
    for (int64_t q = i; q < n && keys[q] == key; ++q) {
        const int64_t e = idx[q];
        if (e < nnz) a += static_cast<float>(values[e]);
        else b += static_cast<float>(values[e - nnz]);
    }
    const int64_t s = static_cast<int64_t>(seg[i]) - 1;
    colidx[s] = static_cast<int32_t>(key % m);
    val[s] = a;
    valT[s] = b;
    ukeys[s] = key;
    if (a != b) *is_symmetric = 0;
}

__global__ void row_pointers(int64_t m, const int64_t* __restrict__ ukeys, const int64_t* __restrict__ nnz_out,
                             int64_t* __restrict__ rowptr) {
    const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (r > m) return;
    const int64_t n = *nnz_out, target = r * m;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ukeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void init_flags(int64_t* nnz_out, int32_t* is_symmetric) {
    *nnz_out = 0;
    *is_symmetric = 1;
}

int key_bits(int64_t m) {
    // keys are < m*m
    unsigned __int128 lim = static_cast<unsigned __int128>(m) * static_cast<unsigned __int128>(m);
    int bits = 1;
    while (bits < 64 && (static_cast<unsigned __int128>(1) << bits) < lim) ++bits;
    return bits;
}

struct BuildPlan {
    size_t sort_tmp, scan_tmp, bytes;
};

BuildPlan build_plan(int64_t m, int64_t nnz) {
    BuildPlan p = {};
    const size_t n = static_cast<size_t>(2 * nnz);
    int64_t* k0 = nullptr;
    int32_t* i0 = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, p.sort_tmp, k0, k0, i0, i0, n > 0 ? n : 1, 0, key_bits(m > 1 ? m : 2), nullptr);
    (void)rocprim::inclusive_scan(nullptr, p.scan_tmp, i0, i0, n > 0 ? n : 1, rocprim::plus<int32_t>(), nullptr);
    const size_t cnt = n > 0 ? n : 1;
    p.bytes = 3 * sparc_align_up(cnt * sizeof(int64_t)) + 4 * sparc_align_up(cnt * sizeof(int32_t)) +
              sparc_align_up(p.sort_tmp) + sparc_align_up(p.scan_tmp);
    return p;
}

}  // namespace

extern "C" int sparc_abi_version(void) { return 1; }

extern "C" size_t sparc_csr_build_workspace_bytes(int64_t m, int64_t nnz_in) {
    if (m < 0 || nnz_in < 0) return 0;
    return build_plan(m, nnz_in).bytes;
}

extern "C" int sparc_csr_build(int64_t m, int64_t nnz_in, const int32_t* coords, const double* values,
                               int64_t* rowptr, int32_t* colidx, float* val, float* valT, int64_t* nnz_out,
                               int32_t* is_symmetric, void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m <= 0 || nnz_in < 0 || 2 * nnz_in > 0x7fffffffLL || !rowptr || !nnz_out || !is_symmetric) return SPARC_ERR_ARG;
    if (nnz_in > 0 && (!coords || !values || !colidx || !val || !valT)) return SPARC_ERR_ARG;
    const BuildPlan p = build_plan(m, nnz_in);
    const int64_t n = 2 * nnz_in;
    const size_t cnt = n > 0 ? static_cast<size_t>(n) : 1;
    WsCarver c(ws, ws_bytes);
    int64_t* keys_a = c.take<int64_t>(cnt);
    int64_t* keys_b = c.take<int64_t>(cnt);
    int64_t* ukeys = c.take<int64_t>(cnt);
    int32_t* idx_a = c.take<int32_t>(cnt);
    int32_t* idx_b = c.take<int32_t>(cnt);
    int32_t* flag = c.take<int32_t>(cnt);
    int32_t* seg = c.take<int32_t>(cnt);
    void* sort_tmp = c.take<char>(p.sort_tmp);
    void* scan_tmp = c.take<char>(p.scan_tmp);
    if (!ws || !c.ok()) return SPARC_ERR_WS;

    hipLaunchKernelGGL(init_flags, dim3(1), dim3(1), 0, stream, nnz_out, is_symmetric);
    const dim3 bt(256);
    if (n > 0) {
        hipLaunchKernelGGL(make_keys, dim3(static_cast<unsigned>((nnz_in + 255) / 256)), bt, 0, stream, m, nnz_in, coords,
                           keys_a, idx_a);
        SPARC_CHECK_LAUNCH();
        size_t st = p.sort_tmp;
        if (rocprim::radix_sort_pairs(sort_tmp, st, keys_a, keys_b, idx_a, idx_b, static_cast<size_t>(n), 0, key_bits(m > 1 ? m : 2),
                                      stream) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        const dim3 gn(static_cast<unsigned>((n + 255) / 256));
        hipLaunchKernelGGL(head_flags, gn, bt, 0, stream, n, keys_b, flag);
        size_t sc = p.scan_tmp;
        if (rocprim::inclusive_scan(scan_tmp, sc, flag, seg, static_cast<size_t>(n), rocprim::plus<int32_t>(), stream) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        hipLaunchKernelGGL(emit_entries, gn, bt, 0, stream, m, nnz_in, n, keys_b, idx_b, flag, seg, values, colidx, val, valT,
                           ukeys, nnz_out, is_symmetric);
        SPARC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(row_pointers, dim3(static_cast<unsigned>((m + 1 + 255) / 256)), bt, 0, stream, m, ukeys, nnz_out, rowptr);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
