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
#include <cstdlib>
#include <mutex>

#include "spmm.h"
#include "chain.h"

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));

// Remainder entries of one row (row-major operand of W = 4 LR floats), gathered by a whole wave: 64 / LR
// operand rows per wave instruction, U of them in flight.  Lanes of the A^T half take the mirrored value.
template <int LR, int U>
__device__ __forceinline__ void gather_some(float4& acc, const int c, const float v, const float vt, const int it, const int g,
                                            const int lr, const int split4, const float* __restrict__ Xg) {
    constexpr int W = LR * 4, G = 64 / LR;
    float4 x[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = (it + u) * G + g;
        const int cj = __shfl(c, idx);
        const float vj = __shfl(v, idx), vtj = __shfl(vt, idx);
        w[u] = lr >= split4 ? vtj : vj;
        x[u] = *reinterpret_cast<const float4*>(Xg + static_cast<int64_t>(cj) * W + lr * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc.x = fmaf(w[u], x[u].x, acc.x);
        acc.y = fmaf(w[u], x[u].y, acc.y);
        acc.z = fmaf(w[u], x[u].z, acc.z);
        acc.w = fmaf(w[u], x[u].w, acc.w);
    }
}

// One row's entries outside the blocks, gathered by a whole wave.  The row's bounds and its first 64 entries
// arrive preloaded (the caller requests them for all its rows at once, before the staging barrier).
template <int LR>
__device__ __forceinline__ void gather_row(const BlocksArgs& a, const int64_t p0, const int64_t p1, int c, float v, float vt,
                                           const int lane, float* __restrict__ dst) {
    constexpr int G = 64 / LR;
    const int g = lane / LR, lr = lane % LR;
    const int split4 = a.split / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p = p0; p < p1; p += 64) {
        const int cnt = static_cast<int>(min(static_cast<int64_t>(64), p1 - p));
        if (p > p0) {  // rows of more than 64 such entries: the next chunk
            c = 0; v = 0.f; vt = 0.f;
            if (lane < cnt) {
                c = a.r_col[p + lane];
                v = a.r_val[p + lane];
                vt = a.r_valT[p + lane];
            }
        }
        const int c_first = __shfl(c, 0);   // idle lanes point at a row that is read anyway and carry a zero weight
        if (lane >= cnt) c = c_first;
        const int iters = (cnt + G - 1) / G;
        int it = 0;
        for (; it + 8 <= iters; it += 8) gather_some<LR, 8>(acc, c, v, vt, it, g, lr, split4, a.Xg);
        if (it + 4 <= iters) { gather_some<LR, 4>(acc, c, v, vt, it, g, lr, split4, a.Xg); it += 4; }
        if (it + 2 <= iters) { gather_some<LR, 2>(acc, c, v, vt, it, g, lr, split4, a.Xg); it += 2; }
        if (it < iters) gather_some<LR, 1>(acc, c, v, vt, it, g, lr, split4, a.Xg);
    }
#pragma unroll
    for (int off = LR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    if (g == 0) *reinterpret_cast<float4*>(dst + lr * 4) = acc;
}

// FUSED (1024 threads): waves 4..15 gather the stripe's entries outside the blocks into LDS while waves 0..3
// multiply the block; the epilogue adds the two.  Otherwise 256 threads and the block product alone.
//
// LAP (with FUSED, a training pass's last product, W = 2 kp): nothing is stored; the finished rows go through the loss /
// gradient epilogue of the chain instead (the formula of spmm.hip's row_epilogue: rowc = {+-1/rho, S_ii, d + cw,
// 1/sqrt(d)}, {r, r2, c, c2}).  The lanes of the A^T half leave their corrected products in LDS, the lanes of the A half
// pick them up, write the gradient and sum the loss; the last workgroup of the grid to arrive adds the partial sums up.
template <bool ADD, bool FUSED, bool LAP = false>
__global__ __launch_bounds__(FUSED ? 1024 : 256) void dense_blocks(const BlocksArgs a) {
    extern __shared__ float4 lds4[];
    float* Xs = reinterpret_cast<float*>(lds4);
    const int32_t* d = a.desc + static_cast<int64_t>(blockIdx.x) * 8;
    const int row0 = d[0], nc = d[1], P = d[2], stripe = d[3];
    const int64_t off = (static_cast<int64_t>(d[5]) << 31) | static_cast<int64_t>(d[4]);
    const int W = a.W;
    float* Ds = Xs + static_cast<int64_t>(P) * W;
    float* DTs = Ds + P * 32;  // only when the values are not symmetric
    float* Rs = DTs + (a.blkT != a.blk ? P * 32 : 0);  // FUSED: [32][W], the stripe's gathers
    float* Ex = Rs + 32 * W;                           // LAP: [32][W / 2] corrected products of the A^T half, then [32][8] row constants
    float* Rc = Ex + 32 * (W / 2);
    if (LAP && threadIdx.x < 256) {
        const int i = stripe * 32 + (threadIdx.x >> 3);
        Rc[threadIdx.x] = i < nc ? a.lap_rowc[static_cast<int64_t>(row0 + i) * SPARC_ROWC + (threadIdx.x & 7)] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;

    // FUSED, gather waves: the bounds and the first 64 entries of this wave's rows (r = wave - 4 + 12 j) are
    // requested now and arrive behind the staging
    constexpr int GR = 3;   // ceil(32 rows / 12 gather waves)
    int64_t gp0[GR], gp1[GR];
    int gc[GR];
    float gv[GR], gvt[GR];
    // the column-sum finish, when it rides: workgroup c < kp owns column c, gather waves 4..9 one of the six sums each
    // (requested here, summed after the gathers)
    FinishShare<1> fin;
    const int fin_wave = wave >= 4 ? wave - 4 : 16;
    if (FUSED && ADD && !LAP) fin.request(a.fin, fin_wave, lane, 12);
    if (FUSED && wave >= 4) {
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const int r = wave - 4 + 12 * j, i = stripe * 32 + r;
            const bool live = r < 32 && i < nc;
            gp0[j] = live ? a.r_rowptr[row0 + i] : 0;
            gp1[j] = live ? a.r_rowptr[row0 + i + 1] : 0;
        }
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const bool has = gp0[j] + lane < gp1[j];
            gc[j] = has ? a.r_col[gp0[j] + lane] : 0;
            gv[j] = has ? a.r_val[gp0[j] + lane] : 0.f;
            gvt[j] = has ? a.r_valT[gp0[j] + lane] : 0.f;
        }
    }

    // FUSED: only the four product waves stage and wait for the staging (a counter in LDS: the gather waves
    // go straight to their gathers and meet the others at the barrier before the epilogue)
    __shared__ int staged;
    if (FUSED) {
        if (threadIdx.x == 0) staged = 0;
        __syncthreads();
    }
    const int nstage = FUSED ? 256 : blockDim.x;   // threads that take part in the staging
    if (!FUSED || wave < 4) {   // stage: 16 bytes per lane and copy instruction, straight into LDS
        const float* xsrc = a.Xg + static_cast<int64_t>(row0) * W;
        const int nx = (a.dbg & 2) ? 0 : nc * W / 4, nd = (a.dbg & 2) ? 0 : P * 32 / 4;
        for (int base = wave * 64; base < nx; base += nstage) {
            const int i = base + lane;
            if (i < nx)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + i * 4),
                                                 (__attribute__((address_space(3))) void*)(lds4 + base), 16, 0, 0);
        }
        for (int i = nc * W + threadIdx.x; i < P * W; i += nstage) Xs[i] = 0.f;  // rows past the cluster
        const float* dsrc = a.blk + off + static_cast<int64_t>(stripe) * P * 32;
        float4* dd = reinterpret_cast<float4*>(Ds);
        for (int base = wave * 64; base < nd; base += nstage) {
            const int i = base + lane;
            if (i < nd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + i * 4),
                                                 (__attribute__((address_space(3))) void*)(dd + base), 16, 0, 0);
        }
        if (a.blkT != a.blk) {
            const float* tsrc = a.blkT + off + static_cast<int64_t>(stripe) * P * 32;
            float4* dt = reinterpret_cast<float4*>(DTs);
            for (int base = wave * 64; base < nd; base += nstage) {
                const int i = base + lane;
                if (i < nd)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tsrc + i * 4),
                                                     (__attribute__((address_space(3))) void*)(dt + base), 16, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's copies into LDS (and the gather waves' entries) have landed
    if (!FUSED) {
        __syncthreads();
    } else if (wave < 4) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the zero fill
        // release / acquire on the counter: a wave that sees four arrivals also sees what the four staged
        if (lane == 0) __hip_atomic_fetch_add(&staged, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // bounded, so that the grid drains whatever happens (the four waves always arrive, within microseconds);
        // a wave that gave up must not multiply operands that were never staged: it stops the kernel instead
        int spin = 0;
        while (__hip_atomic_load(&staged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
            if (++spin > (1 << 20)) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
    }

    if (FUSED && wave >= 4) {   // rows of the stripe, dealt to the twelve gather waves
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            const int r = wave - 4 + 12 * j, i = stripe * 32 + r;
            if (r < 32 && i < nc) {
                float* dst = Rs + r * W;
                if (W == 128) gather_row<32>(a, gp0[j], gp1[j], gc[j], gv[j], gvt[j], lane, dst);
                else if (W == 64) gather_row<16>(a, gp0[j], gp1[j], gc[j], gv[j], gvt[j], lane, dst);
                else gather_row<8>(a, gp0[j], gp1[j], gc[j], gv[j], gvt[j], lane, dst);
            }
        }
    }
    // W / 32 <= 4 column tiles, one per wave
    const int ct = wave;
    const bool mine = wave < 4 && ct * 32 < W;
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x0[16];
    if (mine) {
        // training pass on unsymmetric values: the columns of the A^T half multiply the mirrored block
        const float* As = (a.blkT != a.blk && ct * 32 >= a.split) ? DTs : Ds;
        const int half = P >> 1;
        const float* ap = As + (lk * half) * 32 + li;
        const float* bp = Xs + static_cast<int64_t>(lk * half) * W + ct * 32 + li;
        if (ADD || LAP) {  // the own-row operand of the epilogue is requested before the products
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = min(stripe * 32 + 8 * (r >> 2) + 4 * lk + (r & 3), nc - 1);
                x0[r] = a.X0[static_cast<int64_t>(row0 + i) * W + ct * 32 + li];
            }
        }
        // eight steps per register set, two sets in turn: the LDS reads of the next eight steps are requested
        // before the matrix instructions of the current eight (P is a multiple of 32: half is a multiple of 16)
        float av0[8], bv0[8], av1[8], bv1[8];
        float16_t acc2;   // two accumulators in turn: consecutive matrix instructions are independent
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av0[u] = ap[u * 32];
            bv0[u] = bp[static_cast<int64_t>(u) * W];
        }
        for (int s = 0; s < ((a.dbg & 1) ? 0 : half); s += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av1[u] = ap[(s + 8 + u) * 32];
                bv1[u] = bp[static_cast<int64_t>(s + 8 + u) * W];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u], bv0[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u + 1], bv0[u + 1], acc2, 0, 0, 0);
            }
            if (s + 16 < half) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av0[u] = ap[(s + 16 + u) * 32];
                    bv0[u] = bp[static_cast<int64_t>(s + 16 + u) * W];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u], bv1[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u + 1], bv1[u + 1], acc2, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    if (FUSED) __syncthreads();  // the gathers of the stripe are in LDS
    if (FUSED && ADD && !LAP) {
        __shared__ double fin_sums[6];
        fin.finish(a.fin, fin_wave, lane, 12, fin_sums);
    }
    if constexpr (LAP) {
        const int kp = a.lap_kp;
        const int c = ct * 32 + li;
        const bool right = c >= kp;
        const int cl = right ? c - kp : c;
        float t[16];
        float ma = 0.f, mb = 0.f, mc = 0.f, cf[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) cf[q] = a.lap_M12[6 * kp + q];
        if (mine) {
            const float* Mb = a.lap_M12 + (right ? 3 * kp : 0) + cl;
            ma = Mb[0]; mb = Mb[kp]; mc = Mb[2 * kp];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rs = 8 * (r >> 2) + 4 * lk + (r & 3);
                const float* rc = Rc + rs * 8;
                const float v = acc[r] + Rs[rs * W + c];
                const float ldiag = (cf[0] + cf[1] * rc[6] + cf[2] * rc[7]) + rc[4] * (cf[3] + cf[4] * rc[6] + cf[5] * rc[7]) +
                                    rc[5] * (cf[6] + cf[7] * rc[6] + cf[8] * rc[7]);
                const float u1 = right ? rc[6] : rc[4], u2 = right ? rc[7] : rc[5];
                t[r] = (rc[0] < 0.f ? 0.f : v - rc[1] * x0[r]) + ((ma + u1 * mb + u2 * mc) - ldiag * x0[r]);
                if (right) Ex[rs * kp + cl] = t[r];
            }
        }
        __syncthreads();
        double dot = 0.0;
        if (mine && !right) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rs = 8 * (r >> 2) + 4 * lk + (r & 3);
                const int i = stripe * 32 + rs;
                if (i < nc) {
                    const float* rc = Rc + rs * 8;
                    const float wy = fabsf(rc[0]) * t[r];
                    const float ly = 0.5f * rc[2] * x0[r] - wy;
                    dot += static_cast<double>(x0[r]) * ly;
                    if (cl < a.lap_k)
                        a.lap_gradY[static_cast<int64_t>(row0 + i) * a.lap_ldy + cl] = (rc[2] * x0[r] - wy - Ex[rs * kp + cl]) * (a.lap_inv_m * rc[3]);
                }
            }
        }
        __shared__ double wl[16];
        __shared__ unsigned int last_flag;
        dot = wave_sum(dot);
        if (lane == 0) wl[wave] = dot;
        __syncthreads();
        if (threadIdx.x == 0) publish(a.lap_loss_part + blockIdx.x, ((wl[0] + wl[1]) + wl[2]) + wl[3]);   // (the product waves)
        sum_loss_if_last(a.lap_counter, a.lap_loss_part, gridDim.x, a.lap_loss_out, a.lap_loss_scale, wl, &last_flag);
        return;
    }
    if (mine) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rs = 8 * (r >> 2) + 4 * lk + (r & 3);
            const int i = stripe * 32 + rs;
            if (i < nc && !(a.dbg & 4)) {
                const int64_t p = static_cast<int64_t>(row0 + i) * W + ct * 32 + li;
                float v = acc[r];
                if (ADD) v += x0[r];
                if (FUSED) v += Rs[rs * W + ct * 32 + li];
                a.out[p] = v;
            }
        }
    }
}

// Per device and per instantiation: allow the large dynamic LDS allocation.
template <bool ADD, bool FUSED, bool LAP = false>
int configure() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SPARC_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_blocks<ADD, FUSED, LAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SPARC_BLOCKS_LDS_BYTES) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        done[dev] = true;
    }
    return SPARC_OK;
}

template <bool ADD, bool FUSED, bool LAP = false>
int launch(const BlocksArgs& a0, size_t lds, hipStream_t stream) {
    if (configure<ADD, FUSED, LAP>()) return SPARC_ERR_LAUNCH;
    BlocksArgs a = a0;
    static const int dbg = [] {  // developer switch for kernel ablation timings, read once
        const char* e = getenv("SPARC_DBG_BLOCKS");
        return e ? atoi(e) : 0;
    }();
    a.dbg = dbg;
    hipLaunchKernelGGL((dense_blocks<ADD, FUSED, LAP>), dim3(static_cast<unsigned>(a.ndesc)), dim3(FUSED ? 1024 : 256), lds, stream, a);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

}  // namespace

size_t sparc_blocks_lds_bytes(int max_p, int W, bool mirrored, bool fused, bool lap) {
    return (static_cast<size_t>(max_p) * W + static_cast<size_t>(mirrored ? 2 : 1) * max_p * 32 + (fused ? 32 * W : 0) +
            (lap ? 32 * (W / 2) + 32 * 8 : 0)) * sizeof(float);
}

// fused: the entries outside the blocks (a.r_*) are gathered by the same launch; needs W in {32, 64, 128}
bool sparc_blocks_can_fuse(int max_p, int W, bool mirrored) {
    return (W == 32 || W == 64 || W == 128) && sparc_blocks_lds_bytes(max_p, W, mirrored, true) <= SPARC_BLOCKS_LDS_BYTES;
}

int sparc_blocks_launch(const BlocksArgs& a, bool add, bool fused, hipStream_t stream) {
    if (a.ndesc <= 0) return SPARC_OK;
    if (a.W % 32 || a.W < 32 || a.W > 128 || a.max_p % 32 || a.max_p < 32) return SPARC_ERR_ARG;
    const bool mirrored = a.blkT != a.blk;
    const bool lap = a.lap_rowc != nullptr;
    const size_t lds = sparc_blocks_lds_bytes(a.max_p, a.W, mirrored, fused, lap);
    if (lds > SPARC_BLOCKS_LDS_BYTES) return SPARC_ERR_ARG;
    if (lap) {   // the last product of a training pass with its loss / gradient epilogue
        if (!fused || add || !a.r_rowptr || !a.X0 || a.W != 2 * a.lap_kp || !a.lap_M12 || !a.lap_gradY || !a.lap_loss_part ||
            !a.lap_loss_out || !a.lap_counter)
            return SPARC_ERR_ARG;
        return launch<false, true, true>(a, lds, stream);
    }
    if (fused) {
        if (!sparc_blocks_can_fuse(a.max_p, a.W, mirrored) || !a.r_rowptr) return SPARC_ERR_ARG;
        return add ? launch<true, true>(a, lds, stream) : launch<false, true>(a, lds, stream);
    }
    return add ? launch<true, false>(a, lds, stream) : launch<false, false>(a, lds, stream);
}
