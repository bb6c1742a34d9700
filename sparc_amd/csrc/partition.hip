// Label scores of one propagation sweep of the k-way partitioner (the stand-in for metis.part_graph,
// src/partition_utils.py:56): for every node, the total edge weight towards each label among its neighbours,
// the label with the largest total among the admissible ones (ties: the smallest label) and the total towards
// the node's own label.  One wavefront per node over the CSR adjacency:
//   * up to 64 neighbours: each lane holds one neighbour; lane l sums the weights of the lanes that carry its
//     label (a loop of wave-uniform lane reads), so no table is needed;
//   * more: an open-addressing table per wavefront in LDS (label -> total, LDS atomics), scanned at the end.
//     A node whose neighbours carry more distinct labels than the table holds gets no proposal in this sweep
//     (the partitioner is a heuristic: a skipped move is a valid outcome).
#include "common.h"

namespace {

constexpr int PT_WAVES = 4;
constexpr int PT_SLOTS = 1024;        // table slots per wavefront
constexpr int PT_EMPTY = -1;

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// (largest score, smallest label among the lanes that reach it) over the wave; score 0 means no candidate
__device__ __forceinline__ void wave_best(long long& score, int& label, const int none) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long s2 = __shfl_xor(score, off);
        const int l2 = __shfl_xor(label, off);
        if (s2 > score || (s2 == score && l2 < label)) {
            score = s2;
            label = l2;
        }
    }
    if (score <= 0) label = none;
}

__global__ __launch_bounds__(PT_WAVES * 64) void label_scores(int64_t n, const int64_t* __restrict__ rowptr,
                                                              const int64_t* __restrict__ nbr, const int64_t* __restrict__ w,
                                                              const int64_t* __restrict__ labels, int64_t n_labels,
                                                              const uint8_t* __restrict__ label_ok, int exclude_own,
                                                              int64_t* __restrict__ best, int64_t* __restrict__ best_lab,
                                                              int64_t* __restrict__ cur) {
    __shared__ int keys[PT_WAVES][PT_SLOTS];
    __shared__ unsigned long long vals[PT_WAVES][PT_SLOTS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int none = static_cast<int>(n_labels);
    for (int s = lane; s < PT_SLOTS; s += 64) {
        keys[wave][s] = PT_EMPTY;
        vals[wave][s] = 0ull;
    }
    // (a wave only ever touches its own table: no workgroup barrier is needed)
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * PT_WAVES;
    for (int64_t node = static_cast<int64_t>(blockIdx.x) * PT_WAVES + wave; node < n; node += nwaves) {
        const int64_t p0 = rowptr[node], p1 = rowptr[node + 1];
        const int own = static_cast<int>(labels[node]);
        long long score = 0, mine = 0;
        int label = none;
        if (p1 - p0 <= 64) {
            const int deg = static_cast<int>(p1 - p0);
            int L = -2;
            long long W = 0;
            if (lane < deg) {
                L = static_cast<int>(labels[nbr[p0 + lane]]);
                W = w[p0 + lane];
            }
            long long sum = 0;
            for (int s = 0; s < deg; ++s) {  // wave-uniform lane reads
                const int Ls = __builtin_amdgcn_readlane(L, s);
                const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(W), s);
                const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(static_cast<unsigned long long>(W) >> 32), s);
                if (Ls == L) sum += static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
            }
            mine = wave_sum_ll(L == own ? W : 0);
            const bool ok = L >= 0 && !(exclude_own && L == own) && (!label_ok || label_ok[L]);
            score = ok ? sum : 0;
            label = ok ? L : none;
            wave_best(score, label, none);
        } else {
            bool overflow = false;
            for (int64_t p = p0; p < p1; p += 64) {
                if (p + lane < p1) {
                    const int L = static_cast<int>(labels[nbr[p + lane]]);
                    const unsigned long long W = static_cast<unsigned long long>(w[p + lane]);
                    unsigned slot = (static_cast<unsigned>(L) * 2654435761u) >> 22;   // top 10 bits: PT_SLOTS = 1024
                    int probes = 0;
                    for (; probes < PT_SLOTS; ++probes) {
                        const int old = atomicCAS(&keys[wave][slot], PT_EMPTY, L);
                        if (old == PT_EMPTY || old == L) {
                            atomicAdd(&vals[wave][slot], W);
                            break;
                        }
                        slot = (slot + 1) & (PT_SLOTS - 1);
                    }
                    if (probes == PT_SLOTS) overflow = true;
                }
            }
            overflow = __any(overflow);
            for (int s = lane; s < PT_SLOTS; s += 64) {  // scan and clear
                const int L = keys[wave][s];
                if (L != PT_EMPTY) {
                    const long long sum = static_cast<long long>(vals[wave][s]);
                    if (L == own) mine += sum;
                    const bool ok = !(exclude_own && L == own) && (!label_ok || label_ok[L]);
                    if (ok && (sum > score || (sum == score && L < label))) {
                        score = sum;
                        label = L;
                    }
                    keys[wave][s] = PT_EMPTY;
                    vals[wave][s] = 0ull;
                }
            }
            mine = wave_sum_ll(mine);
            wave_best(score, label, none);
            if (overflow) {
                score = 0;
                label = none;
            }
        }
        if (lane == 0) {
            best[node] = score;
            best_lab[node] = label;
            cur[node] = mine;
        }
    }
}

}  // namespace

extern "C" int sparc_label_scores(int64_t n, const int64_t* rowptr, const int64_t* nbr, const int64_t* w, const int64_t* labels,
                                  int64_t n_labels, const uint8_t* label_ok, int exclude_own, int64_t* best, int64_t* best_lab,
                                  int64_t* cur, sparc_stream_t stream) {
    if (n < 0 || n_labels < 1 || n_labels >= 0x7fffffff || !rowptr || !labels || !best || !best_lab || !cur) return SPARC_ERR_ARG;
    if (n == 0) return SPARC_OK;
    const int64_t blocks = (n + PT_WAVES - 1) / PT_WAVES;
    const unsigned grid = static_cast<unsigned>(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(label_scores, dim3(grid), dim3(PT_WAVES * 64), 0, stream, n, rowptr, nbr, w, labels, n_labels, label_ok,
                       exclude_own, best, best_lab, cur);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
