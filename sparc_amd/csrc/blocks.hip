// Dense diagonal blocks of a cluster batch on the matrix cores.
//
// When the clusters of a batch are small and nearly complete sub-graphs (Reddit-shape batches: 155-node
// clusters with 86 % of their node pairs connected), a row gather moves 512 bytes of operand per stored
// entry and is bound by the vector memory path, while the same product as a dense block is
// 2 n_c^2 W flop per cluster on data that fits LDS.  The matrix is split as A = blockdiag(D_c) + C; this
// file computes out = blockdiag(D_c) Xg (+ X0), the row kernel then adds C Xg on the (few) entries that join
// different clusters.
//
// One workgroup = (cluster, stripe of 32 rows), four waves, wave w = columns [32 w, 32 w + 32) of the
// operand.  LDS holds the cluster's operand rows Xs[P][W] (P = cluster rows rounded up to 32; rows past the
// cluster are zero) and the stripe of the block in k-major order Ds[P][32] (Ds[k][i] = D[32 stripe + i][k]),
// which is how the blocks are stored in memory, so both are staged with asynchronous global->LDS copies.
// v_mfma_f32_32x32x2_f32: lane l feeds A[i = l % 32][kk = l / 32] and B[kk][j = l % 32]; step s of lane
// half kk takes k = kk P/2 + s, so every read is 32 consecutive floats.  Result register r of lane l is
// C[8 (r / 4) + 4 (l / 32) + r % 4][l % 32].
#include <mutex>

#include "spmm.h"

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));

template <bool ADD>
__global__ __launch_bounds__(256) void dense_blocks(const BlocksArgs a) {
    extern __shared__ float4 lds4[];
    float* Xs = reinterpret_cast<float*>(lds4);
    const int32_t* d = a.desc + static_cast<int64_t>(blockIdx.x) * 8;
    const int row0 = d[0], nc = d[1], P = d[2], stripe = d[3];
    const int64_t off = (static_cast<int64_t>(d[5]) << 31) | static_cast<int64_t>(d[4]);
    const int W = a.W;
    float* Ds = Xs + static_cast<int64_t>(P) * W;
    float* DTs = Ds + P * 32;  // only when the values are not symmetric
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;

    {   // stage: 16 bytes per lane and copy instruction, straight into LDS
        const float* xsrc = a.Xg + static_cast<int64_t>(row0) * W;
        const int nx = nc * W / 4, nd = P * 32 / 4;
        for (int base = wave * 64; base < nx; base += 256) {
            const int i = base + lane;
            if (i < nx)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + i * 4),
                                                 (__attribute__((address_space(3))) void*)(lds4 + base), 16, 0, 0);
        }
        for (int i = nc * W + threadIdx.x; i < P * W; i += 256) Xs[i] = 0.f;  // rows past the cluster
        const float* dsrc = a.blk + off + static_cast<int64_t>(stripe) * P * 32;
        float4* dd = reinterpret_cast<float4*>(Ds);
        for (int base = wave * 64; base < nd; base += 256) {
            const int i = base + lane;
            if (i < nd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + i * 4),
                                                 (__attribute__((address_space(3))) void*)(dd + base), 16, 0, 0);
        }
        if (a.blkT != a.blk) {
            const float* tsrc = a.blkT + off + static_cast<int64_t>(stripe) * P * 32;
            float4* dt = reinterpret_cast<float4*>(DTs);
            for (int base = wave * 64; base < nd; base += 256) {
                const int i = base + lane;
                if (i < nd)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tsrc + i * 4),
                                                     (__attribute__((address_space(3))) void*)(dt + base), 16, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's copies into LDS have landed
    __syncthreads();

    for (int ct = wave; ct * 32 < W; ct += 4) {
        // training pass on unsymmetric values: the columns of the A^T half multiply the mirrored block
        const float* As = (a.blkT != a.blk && ct * 32 >= a.split) ? DTs : Ds;
        const int half = P >> 1;
        const float* ap = As + (lk * half) * 32 + li;
        const float* bp = Xs + static_cast<int64_t>(lk * half) * W + ct * 32 + li;
        float16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // the own-row operand of the epilogue is requested before the products
        float x0[16];
        if (ADD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = min(stripe * 32 + 8 * (r >> 2) + 4 * lk + (r & 3), nc - 1);
                x0[r] = a.X0[static_cast<int64_t>(row0 + i) * W + ct * 32 + li];
            }
        }
        // eight steps per register set, two sets in turn: the LDS reads of the next eight steps are requested
        // before the matrix instructions of the current eight (P is a multiple of 32: half is a multiple of 16)
        float av0[8], bv0[8], av1[8], bv1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av0[u] = ap[u * 32];
            bv0[u] = bp[static_cast<int64_t>(u) * W];
        }
        for (int s = 0; s < half; s += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av1[u] = ap[(s + 8 + u) * 32];
                bv1[u] = bp[static_cast<int64_t>(s + 8 + u) * W];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u], bv0[u], acc, 0, 0, 0);
            if (s + 16 < half) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av0[u] = ap[(s + 16 + u) * 32];
                    bv0[u] = bp[static_cast<int64_t>(s + 16 + u) * W];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u], bv1[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = stripe * 32 + 8 * (r >> 2) + 4 * lk + (r & 3);
            if (i < nc) {
                const int64_t p = static_cast<int64_t>(row0 + i) * W + ct * 32 + li;
                a.out[p] = ADD ? acc[r] + x0[r] : acc[r];
            }
        }
    }
}

// Per device and per instantiation: allow the large dynamic LDS allocation.
template <bool ADD>
int configure() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SPARC_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_blocks<ADD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SPARC_BLOCKS_LDS_BYTES) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        done[dev] = true;
    }
    return SPARC_OK;
}

}  // namespace

size_t sparc_blocks_lds_bytes(int max_p, int W, bool mirrored) {
    return (static_cast<size_t>(max_p) * W + static_cast<size_t>(mirrored ? 2 : 1) * max_p * 32) * sizeof(float);
}

int sparc_blocks_launch(const BlocksArgs& a, bool add, hipStream_t stream) {
    if (a.ndesc <= 0) return SPARC_OK;
    if (a.W % 32 || a.W < 32 || a.W > 128 || a.max_p % 32 || a.max_p < 32) return SPARC_ERR_ARG;
    const size_t lds = sparc_blocks_lds_bytes(a.max_p, a.W, a.blkT != a.blk);
    if (lds > SPARC_BLOCKS_LDS_BYTES) return SPARC_ERR_ARG;
    if (add) {
        if (configure<true>()) return SPARC_ERR_LAUNCH;
        hipLaunchKernelGGL((dense_blocks<true>), dim3(static_cast<unsigned>(a.ndesc)), dim3(256), lds, stream, a);
    } else {
        if (configure<false>()) return SPARC_ERR_LAUNCH;
        hipLaunchKernelGGL((dense_blocks<false>), dim3(static_cast<unsigned>(a.ndesc)), dim3(256), lds, stream, a);
    }
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
