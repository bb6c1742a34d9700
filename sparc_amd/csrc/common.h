// Shared helpers for the gfx950 kernels of libsparc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sparc_hip.h"

#define SPARC_WAVE 64

#define SPARC_CHECK_LAUNCH()                                  \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return SPARC_ERR_LAUNCH; \
    } while (0)

static inline size_t sparc_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Carves consecutive 256-byte aligned regions out of one workspace.
struct WsCarver {
    char* base;
    size_t off;
    size_t cap;
    WsCarver(void* p, size_t bytes) : base(static_cast<char*>(p)), off(0), cap(bytes) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = sparc_align_up(count * sizeof(T));
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap && (reinterpret_cast<uintptr_t>(base) & 255) == 0; }
};

// Blocks b and b+8 share an XCD (round-robin dispatch): hand each XCD one contiguous
// range of logical blocks so that neighbouring rows (one cluster) meet in one L2.
// Bijective for every grid size.
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t b, int64_t nb) {
    const int64_t xcd = b & 7, q = nb >> 3, rem = nb & 7;
    const int64_t start = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return start + (b >> 3);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
