// The last workgroup of a grid to arrive sums what the others left: arrival counters in device memory, partial
// results passed through device-scope atomic stores and loads.  Shared by the kernels of spmm.hip and blocks.hip.
#pragma once
#include "common.h"
#include "spmm.h"

constexpr int CHAIN_SUBS = 64;                 // second-level counters of a large grid, one cache line each
constexpr int CHAIN_LINE = 32;                 // words per 128-byte line
constexpr int CHAIN_WORDS = (1 + CHAIN_SUBS) * CHAIN_LINE;   // per slot: the counter and its sub-counters

// Partial results travel between workgroups through device-scope atomic stores and loads (write-through and
// read-through the per-XCD L2s), so that no workgroup has to write back or invalidate a whole L2, which is what a
// device-scope fence costs here (measured: 17 -> 98 us for the last pass with a fence per workgroup).
static __device__ __forceinline__ void publish(double* p, const double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ double collect(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by every thread of every workgroup of the grid after the workgroup's partial results are published.
// Returns true, in all threads, in the one workgroup that arrives last; there every other workgroup's
// partial results can be collected.  `flag` is a word of LDS.
static __device__ __forceinline__ bool arrive_last(unsigned int* counter, unsigned int* subs, const unsigned int total, unsigned int* flag) {
    // this thread's published stores have completed (the fence keeps the compiler from moving them; the wait is
    // for the write-through stores themselves: vmcnt(0)) ...
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                         // ... and so have those of the rest of the workgroup
    if (threadIdx.x == 0) {
        // One atomic per workgroup costs about 7 ns when they all hit one word (measured: 6 k workgroups of the row
        // kernel, +45 us), so a grid of more than 2 CHAIN_SUBS workgroups counts in two levels: workgroup b arrives at
        // sub-counter b mod CHAIN_SUBS (`subs`: they follow `counter`, a line apart), and the last arrival of every
        // sub-counter at `counter` itself.
        bool last;
        if (subs && total > 2 * CHAIN_SUBS) {
            const unsigned int j = blockIdx.x % CHAIN_SUBS;
            const unsigned int mine = (total - j + CHAIN_SUBS - 1) / CHAIN_SUBS;   // workgroups b = j (mod CHAIN_SUBS), b < total
            unsigned int* sub = subs + j * CHAIN_LINE;
            last = __hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == mine;
            if (last) {
                __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == CHAIN_SUBS;
            }
        } else {
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == total;
        }
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody has arrived: ready for the next call
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return last;
}

// *loss_out = scale * (sum of the grid's nparts loss partials), summed in a fixed order by the last workgroup to arrive
// at `counter` (its sub-counters follow a line apart); wl: LDS for one double per wavefront, flag: a word of LDS.
static __device__ __forceinline__ void sum_loss_if_last(unsigned int* counter, const double* loss_part, const unsigned int nparts,
                                                        double* loss_out, const double loss_scale, double* wl, unsigned int* flag) {
    if (!arrive_last(counter, counter + CHAIN_LINE, gridDim.x, flag)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    double t = 0.0;
    for (unsigned int i = threadIdx.x; i < nparts; i += blockDim.x) t += collect(loss_part + i);
    t = wave_sum(t);
    if (lane == 0) wl[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < nwaves; ++w) sum += wl[w];
        *loss_out = sum * loss_scale;
    }
}

// The finish of the chain's column sums (M1 = C (V^T Yn), M2 = C^T (U^T X2) from the block partials of lap_scale)
// spread over the workgroups of the first inner pass (the last pass of the chain is the first
// reader of M12, so it has a whole pass of time): workgroup c < KP owns column c, its wave w the sum over the block
// partials of row w of the six (V^T Yn: 0-2, U^T X2: 3-5): a lane requests its partials when the kernel starts
// (at most SPARC_SCALE_MAX_BLOCKS: lap_scale's block limit) and the sums are formed after the workgroup's own work, so the
// loads cost the pass nothing.  NQ: sums per wave (workgroups of fewer than six waves take two).  One extra
// workgroup doing all of it was tried first and became the pass's tail (14.4 -> 21.8 us).
template <int NQ>
struct FinishShare {
    static constexpr int PL = SPARC_SCALE_MAX_BLOCKS / 64;   // partials per lane
    float v[NQ][PL];
    __device__ __forceinline__ void request(const ColsumFinish& f, const int wave, const int lane, const int nwaves) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int t = 0; t < PL; ++t) v[j][t] = 0.f;
            const int q = wave + j * nwaves;
            if (f.colpart && static_cast<int>(blockIdx.x) < f.kp && q < (f.dual ? 6 : 3)) {
                const float* src = f.colpart + static_cast<int64_t>(q) * f.kp + blockIdx.x;
#pragma unroll
                for (int t = 0; t < PL; ++t)
                    if (lane + 64 * t < f.nparts) v[j][t] = src[static_cast<int64_t>(lane + 64 * t) * 6 * f.kp];
            }
        }
    }
    // sums: LDS for six doubles; every thread of the workgroup calls this (it holds a barrier)
    __device__ __forceinline__ void finish(const ColsumFinish& f, const int wave, const int lane, const int nwaves, double* sums) const {
        if (!f.colpart || static_cast<int>(blockIdx.x) >= f.kp) return;   // uniform over the workgroup
        const int ns = f.dual ? 6 : 3;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = wave + j * nwaves;
            double mine = 0.0;
#pragma unroll
            for (int t = 0; t < PL; ++t) mine += static_cast<double>(v[j][t]);
            const double t = wave_sum(mine);
            if (lane == 0 && q < ns) sums[q] = t;
        }
        __syncthreads();
        if (static_cast<int>(threadIdx.x) < ns) {
            const int q = threadIdx.x;
            double t = 0.0;
            for (int b = 0; b < 3; ++b) t += (q < 3 ? f.coupling[q * 3 + b] : f.coupling[b * 3 + (q - 3)]) * sums[q < 3 ? b : 3 + b];
            f.M12[q * f.kp + blockIdx.x] = static_cast<float>(t);
        }
        if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 73) f.M12[6 * f.kp + threadIdx.x - 64] = static_cast<float>(f.coupling[threadIdx.x - 64]);
        __syncthreads();   // `sums` may be reused by the caller
    }
};
