// The last workgroup of a grid to arrive sums what the others left: arrival counters in device memory, partial
// results passed through device-scope atomic stores and loads.  Shared by the kernels of spmm.hip and blocks.hip.
#pragma once
#include "common.h"

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
