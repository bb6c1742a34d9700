// Order statistics of the rows of a float64 matrix: for every row, the values at a few ascending ranks
// (the e-th smallest entry, ties counted as separate entries).  This is the selection step of the brute-force
// nearest-neighbour search behind the token lists (implementations/SPARCphormer/utils.py:268-298 asks
// scikit-learn for the 16 384 nearest nodes of every node and then only uses prefixes of 2, 4, ... of them):
// a prefix of length e is the set {j : d(i, j) <= e-th smallest distance of row i}, so the prefix lengths
// become per-row thresholds and no row is ever sorted.
//
// One workgroup per row.  Keys are the doubles mapped to unsigned integers in the same order.  A histogram of
// the top 11 key bits is made once per row; for every requested rank the bucket holding it is refined 11 bits
// at a time (only entries of that bucket are counted) until it holds at most SEL_CAP entries, which are then
// sorted in LDS, or until all 64 bits are fixed (a rank inside a run of equal values).  The row (n x 8 bytes)
// is re-read from L2 a few times per rank; nothing is written but the results.
#include "common.h"

namespace {

constexpr int SEL_THREADS = 256;
constexpr int SEL_BINS = 2048;   // 11 bits per level
constexpr int SEL_CAP = 2048;    // bucket size that is sorted in LDS
constexpr int SEL_MAX_RANKS = 32;

struct SelRanks {
    int n;
    int64_t r[SEL_MAX_RANKS];  // 0-based, ascending
};

__device__ __forceinline__ unsigned long long ordered_key(const double d) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(d));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_value(const unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double(static_cast<long long>(b));
}

// hist[0..SEL_BINS) -> exclusive prefix sums in place; returns through *bin / *before the bucket that holds
// 0-based rank t and the number of entries in the buckets before it.  All threads call it.
__device__ void locate(unsigned* hist, unsigned* part, const long long t, int* bin, long long* before, unsigned* count) {
    const int tid = threadIdx.x;
    constexpr int PER = SEL_BINS / SEL_THREADS;
    unsigned loc[PER];
    unsigned sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        loc[q] = hist[tid * PER + q];
        sum += loc[q];
    }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < SEL_THREADS; off <<= 1) {  // inclusive scan of the per-thread sums
        const unsigned add = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    long long run = static_cast<long long>(part[tid]) - sum;  // entries in the buckets before this thread's
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (t >= run && t < run + static_cast<long long>(loc[q])) {
            *bin = tid * PER + q;
            *before = run;
            *count = loc[q];
        }
        run += loc[q];
    }
    __syncthreads();
}

__global__ __launch_bounds__(SEL_THREADS) void row_rank_values(int64_t n, const double* __restrict__ D, int64_t ld,
                                                               const SelRanks ranks, double* __restrict__ out) {
    __shared__ unsigned hist0[SEL_BINS];
    __shared__ unsigned hist[SEL_BINS];
    __shared__ unsigned part[SEL_THREADS];
    __shared__ unsigned long long cand[SEL_CAP];
    __shared__ int s_bin;
    __shared__ long long s_before;
    __shared__ unsigned s_count;
    __shared__ unsigned s_fill;
    const int tid = threadIdx.x;
    const double* __restrict__ row = D + static_cast<int64_t>(blockIdx.x) * ld;
    double* __restrict__ res = out + static_cast<int64_t>(blockIdx.x) * ranks.n;

    for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist0[b] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += SEL_THREADS) atomicAdd(&hist0[ordered_key(row[i]) >> 53], 1u);
    __syncthreads();

    for (int r = 0; r < ranks.n; ++r) {
        long long t = ranks.r[r] < n ? ranks.r[r] : n - 1;
        for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist[b] = hist0[b];
        __syncthreads();
        locate(hist, part, t, &s_bin, &s_before, &s_count);
        unsigned long long prefix = static_cast<unsigned long long>(s_bin);
        t -= s_before;
        unsigned count = s_count;
        int shift = 53;  // the key bits above `shift` are fixed to `prefix`
        __syncthreads();
        while (count > SEL_CAP && shift > 0) {
            const int nshift = shift > 11 ? shift - 11 : 0;
            const unsigned mask = (1u << (shift - nshift)) - 1u;
            for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist[b] = 0;
            __syncthreads();
            for (int64_t i = tid; i < n; i += SEL_THREADS) {
                const unsigned long long k = ordered_key(row[i]);
                if ((k >> shift) == prefix) atomicAdd(&hist[static_cast<unsigned>(k >> nshift) & mask], 1u);
            }
            __syncthreads();
            locate(hist, part, t, &s_bin, &s_before, &s_count);
            prefix = (prefix << (shift - nshift)) | static_cast<unsigned long long>(s_bin);
            t -= s_before;
            count = s_count;
            shift = nshift;
            __syncthreads();
        }
        if (shift == 0) {  // every bit is fixed: the rank lies inside a run of equal values
            if (tid == 0) res[r] = key_value(prefix);
            __syncthreads();
            continue;
        }
        // collect the bucket and sort it
        if (tid == 0) s_fill = 0;
        __syncthreads();
        for (int64_t i = tid; i < n; i += SEL_THREADS) {
            const unsigned long long k = ordered_key(row[i]);
            if ((k >> shift) == prefix) cand[atomicAdd(&s_fill, 1u)] = k;
        }
        __syncthreads();
        int len = 1;
        while (len < static_cast<int>(count)) len <<= 1;
        for (int i = static_cast<int>(count) + tid; i < len; i += SEL_THREADS) cand[i] = ~0ull;
        __syncthreads();
        for (int size = 2; size <= len; size <<= 1) {  // bitonic sort, ascending
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tid; i < len; i += SEL_THREADS) {
                    const int j = i ^ stride;
                    if (j > i) {
                        const unsigned long long a = cand[i], b = cand[j];
                        const bool up = (i & size) == 0;
                        if ((a > b) == up) {
                            cand[i] = b;
                            cand[j] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (tid == 0) res[r] = key_value(cand[t]);
        __syncthreads();
    }
}

}  // namespace

extern "C" int sparc_row_rank_values(int64_t rows, int64_t n, const double* D, int64_t ld, int nranks,
                                     const int64_t* ranks, double* out, sparc_stream_t stream) {
    if (rows < 0 || n < 1 || !D || ld < n || nranks < 1 || nranks > SEL_MAX_RANKS || !ranks || !out) return SPARC_ERR_ARG;
    if (rows > 0x7fffffff) return SPARC_ERR_SIZE;
    SelRanks r;
    r.n = nranks;
    for (int i = 0; i < nranks; ++i) {
        if (ranks[i] < 1 || (i > 0 && ranks[i] < ranks[i - 1])) return SPARC_ERR_ARG;
        r.r[i] = ranks[i] - 1;
    }
    if (rows == 0) return SPARC_OK;
    hipLaunchKernelGGL(row_rank_values, dim3(static_cast<unsigned>(rows)), dim3(SEL_THREADS), 0, stream, n, D, ld, r, out);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
