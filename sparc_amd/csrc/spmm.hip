// CSR row-gather products for cluster batches and the fused Laplacian / loss epilogue.
//
// Two kernels share one epilogue:
//
//  spmm_rows   one wavefront per row.  The row's column indices and values are streamed
//              coalesced (one element per lane, 64 at a time) and handed to the consuming
//              lanes with cross-lane reads; rows of the dense operand X are gathered from
//              global memory (L2), 16 bytes per lane, 64/LR rows per wave instruction.
//
//  spmm_tiled  one workgroup per (window of rows, column slab, part of the window's rows).
//              The window's rows of X (one slab wide) are staged in LDS once; every stored
//              entry whose column falls in the window (the cluster-diagonal block, i.e.
//              nearly all of a cluster batch) is gathered from LDS, the rest from global
//              memory.  The row's entries are spread over the four 16-lane DPP rows of the
//              wave and broadcast inside each with v_mov_dpp row_newbcast, so the LDS pipe
//              carries nothing but the 16-byte gathers.
//
// In "dual" mode X is [Yn | Yn/rho]: the left half is multiplied by A (val), the right half
// by A^T (valT, same pattern), so W Yn and W^T Yn share one pass over the index stream.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "spmm.h"
#include "chain.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // spmm_rows: one row per wave
constexpr int WIN_MAX_THREADS = 1024;  // spmm_win: workgroup size is a launch parameter, up to 16 waves
constexpr int WIN_MAX_RECORDS = SPARC_TILE_MAX_RECORDS;  // rows of one workgroup (their records sit in LDS)

// ------------------------------------------------------------------ last workgroup done
// The loss partials of the windowed kernel's last pass are summed by whichever workgroup arrives last instead of
// by a kernel of their own.  The arrival counters are part of the library image (zero when it is loaded, and put
// back to zero by the workgroup that finishes); a call takes the next of CHAIN_SLOTS sets, so calls that overlap on
// one device (other streams) count separately as long as fewer than CHAIN_SLOTS of them are in flight.  (The row
// kernel does the same up to ROWS_FINISH_MAX_BLOCKS workgroups and keeps its finishing launch beyond: with one workgroup
// per four rows the arrivals then cost more than the launch, 6 k workgroups: 39 + 5 us as two launches, 50 us as one.)
constexpr int CHAIN_SLOTS = 256;
constexpr int64_t ROWS_FINISH_MAX_BLOCKS = 2048;
__device__ unsigned int g_chain_counter[CHAIN_SLOTS * CHAIN_WORDS];
std::atomic<unsigned int> g_chain_next{0};

// The device address of the counters on the current device, for kernels of other source files (nullptr on failure);
// looked up once per device.
unsigned int* chain_counters_on_current_device() {
    static std::mutex mu;
    static unsigned int* ptr[64] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_chain_counter)) != hipSuccess || !p) return nullptr;
        ptr[dev] = static_cast<unsigned int*>(p);
    }
    return ptr[dev];
}

// loss = scale * (sum of the grid's loss partials), summed in a fixed order by the last workgroup
__device__ __forceinline__ void finish_loss(const SpmmArgs& a, const unsigned int nparts, double* wl, unsigned int* flag) {
    sum_loss_if_last(g_chain_counter + static_cast<size_t>(a.chain_slot) * CHAIN_WORDS, a.loss_part, nparts, a.loss_out,
                     a.loss_scale, wl, flag);
}

// M1 = C (V^T Yn), M2 = C^T (U^T X2): the 3 x KP matrices of the rank-3 correction, from the block partials of
// lap_scale (M12 = [M1; M2; the 3 x 3 coupling]).  Work of ONE workgroup of any size: the partials of one column
// sum are dealt to `split` threads (fixed assignment, eight independent chains each) and combined in a fixed
// order; S is LDS for FINISH_SPLIT x 6 x SPARC_MAX_K doubles.  (The stand-alone kernel: the dense-block path and
// shapes that FinishShare does not take.)
constexpr int FINISH_SPLIT = 4;
__device__ __forceinline__ void colsum_finish_body(const ColsumFinish& f, double* S) {
    const int KP = f.kp, n = 6 * KP, nt = blockDim.x;
    const int split = max(1, min(FINISH_SPLIT, nt / n));
    const int ns = f.dual ? 6 : 3;
    for (int idx = threadIdx.x; idx < n * split; idx += nt) {
        const int e = idx % n, share = idx / n;
        const int q = e / KP, col = e % KP;
        if (q < ns) {
            double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // eight independent chains, fixed order
            int b = share;
            for (; b + 7 * split < f.nparts; b += 8 * split) {
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] += f.colpart[(static_cast<int64_t>(b + u * split) * 6 + q) * KP + col];
            }
            for (; b < f.nparts; b += split) t[0] += f.colpart[(static_cast<int64_t>(b) * 6 + q) * KP + col];
            S[(share * 6 + q) * SPARC_MAX_K + col] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += nt) {
        const int q = e / KP, col = e % KP;
        if (q >= ns) continue;
        double t = 0.0;
        for (int b = 0; b < 3; ++b) {
            const int src = q < 3 ? b : 3 + b;
            double sum = S[src * SPARC_MAX_K + col];
            for (int h = 1; h < split; ++h) sum += S[(h * 6 + src) * SPARC_MAX_K + col];
            t += (q < 3 ? f.coupling[q * 3 + b] : f.coupling[b * 3 + (q - 3)]) * sum;
        }
        f.M12[q * KP + col] = static_cast<float>(t);
    }
    if (threadIdx.x < 9) f.M12[6 * KP + threadIdx.x] = static_cast<float>(f.coupling[threadIdx.x]);
}


__global__ __launch_bounds__(1024) void lap_colsum_finish(const ColsumFinish f) {
    __shared__ double S[FINISH_SPLIT * 6 * SPARC_MAX_K];
    colsum_finish_body(f, S);
}

__global__ __launch_bounds__(256) void loss_finish(int64_t n, const double* __restrict__ part, double scale,
                                                   double* __restrict__ loss) {
    __shared__ double sm[256];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) t += part[i];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = sm[0] * scale;
}

// ------------------------------------------------------------------ shared epilogue
// A lane owns VEC consecutive columns of one row: `cl` is their index inside the half (0..kp-1),
// `right` tells the A^T half from the A half, `half` is the lane distance between the two halves.
template <int VEC>
struct RowPre {  // what the epilogue reads of its own row, requested before the gathers
    float x0[VEC];
    float4 rc0, rc1;
};

template <int VEC>
struct LapConst {  // per-lane constants of the Laplacian epilogue: rows of M1 / M2 and the 3x3 coupling
    float ma[VEC], mb[VEC], mc[VEC];
    float cf[9];
};

template <int VEC>
__device__ __forceinline__ void load_vec(float (&dst)[VEC], const float* __restrict__ src) {
    typedef float V __attribute__((ext_vector_type(VEC)));
    const V v = *reinterpret_cast<const V*>(src);
#pragma unroll
    for (int q = 0; q < VEC; ++q) dst[q] = v[q];
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ dst, const float (&src)[VEC]) {
    typedef float V __attribute__((ext_vector_type(VEC)));
    V v;
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] = src[q];
    *reinterpret_cast<V*>(dst) = v;
}

template <int VEC, int EPI>
__device__ __forceinline__ LapConst<VEC> lap_const(const SpmmArgs& a, const int cl, const bool right) {
    LapConst<VEC> c;
#pragma unroll
    for (int q = 0; q < VEC; ++q) c.ma[q] = c.mb[q] = c.mc[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.cf[i] = 0.f;
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        const int kp = a.kp;
        const float* Mb = a.M12 + (right ? 3 * kp : 0) + cl;
        load_vec<VEC>(c.ma, Mb);
        load_vec<VEC>(c.mb, Mb + kp);
        load_vec<VEC>(c.mc, Mb + 2 * kp);
#pragma unroll
        for (int i = 0; i < 9; ++i) c.cf[i] = a.M12[6 * kp + i];
    }
    return c;
}

// x0p: this lane's columns of the row's own operand (EPI_ADD_X0 and the Laplacian epilogues)
template <int VEC, int EPI>
__device__ __forceinline__ RowPre<VEC> row_preload(const SpmmArgs& a, const int64_t row, const float* __restrict__ x0p) {
    RowPre<VEC> r;
#pragma unroll
    for (int q = 0; q < VEC; ++q) r.x0[q] = 0.f;
    r.rc0 = r.rc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_ADD_X0 || EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) load_vec<VEC>(r.x0, x0p);
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        r.rc0 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC);
        r.rc1 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC + 4);
    }
    return r;
}

// outp: where this lane's VEC results of the row go (store epilogues); colbase: index of the lane's first
// column in the identity / statistics epilogues of the affinity preparation.
template <int VEC, bool DUAL, int EPI>
__device__ __forceinline__ void row_epilogue(const SpmmArgs& a, const int64_t row, const int lane, const bool owner,
                                             const int half, const int cl, const bool right, const RowPre<VEC>& pre,
                                             const LapConst<VEC>& lc, float (&acc)[VEC], float* __restrict__ outp,
                                             const int64_t colbase, double& wloss, const float oscale = 1.f) {
    if (EPI == EPI_STORE) {
        if (owner) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] *= oscale;
            store_vec<VEC>(outp, acc);
        }
    } else if (EPI == EPI_ADD_X0) {
        if (owner) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = (acc[q] + pre.x0[q]) * oscale;
            store_vec<VEC>(outp, acc);
        }
    } else if (EPI == EPI_ADD_IDENT) {
        bool nz = false;
        if (owner) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                acc[q] += (colbase + q == row) ? 1.f : 0.f;
                nz |= acc[q] != 0.f;
            }
        }
        nz = __any(nz);   // a zero row is flagged instead of stored
        if (owner && nz) store_vec<VEC>(outp, acc);
        if (lane == 0) a.oflag[row] = nz ? 1 : 0;
    } else if (EPI == EPI_PREP_ACC) {
        // acc holds S[row, colbase .. colbase + VEC): gather the row statistics of S
        double ss = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (owner) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const int64_t cj = colbase + q;
                if (cj < a.m && cj != row) {
                    const double sv = static_cast<double>(acc[q]);
                    ss += sv * sv;
                    s0 += sv;
                    s1 += sv * a.cvec[cj];
                    s2 += sv * a.c2vec[cj];
                }
            }
        }
        ss = wave_sum(ss); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (owner && lane == 0) {
            double* r = a.racc + row * 8;
            r[0] += ss; r[1] += s0; r[2] += s1; r[3] += s2;
        }
    } else {  // EPI_LAP_SINGLE / EPI_LAP_DUAL
        // rowc: {+-1/rho, S_ii, d+cw, 1/sqrt(d)}, {r, r2, c, c2}; a negative 1/rho marks a row without
        // off-diagonal entries, whose chain result is exactly S_ii x and is dropped instead of cancelled
        const float4 rc0 = pre.rc0, rc1 = pre.rc1;
        const bool isolated = rc0.x < 0.f;
        const float irho = fabsf(rc0.x);
        const float* Cf = lc.cf;  // the 3x3 coupling, row-major
        // diagonal of U C V^T for this row
        const float ldiag = (Cf[0] + Cf[1] * rc1.z + Cf[2] * rc1.w) + rc1.x * (Cf[3] + Cf[4] * rc1.z + Cf[5] * rc1.w) +
                            rc1.y * (Cf[6] + Cf[7] * rc1.z + Cf[8] * rc1.w);
        const float u1 = right ? rc1.z : rc1.x, u2 = right ? rc1.w : rc1.y;
        float t[VEC], wt[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            t[q] = (isolated ? 0.f : acc[q] - rc0.y * pre.x0[q]) + ((lc.ma[q] + u1 * lc.mb[q] + u2 * lc.mc[q]) - ldiag * pre.x0[q]);
            wt[q] = DUAL ? __shfl(t[q], lane + half) : 0.f;  // W^T Yn sits `half` lanes to the right of W Yn
        }
        double dot = 0.0;
        if (owner && !right) {
            const float hd = 0.5f * rc0.z;
            const float sc = a.inv_m * rc0.w;
            float* gp = DUAL ? a.gradY + row * a.ldy + cl : nullptr;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float wy = irho * t[q];
                const float ly = hd * pre.x0[q] - wy;
                dot += static_cast<double>(pre.x0[q]) * ly;
                if (DUAL && cl + q < a.k) gp[q] = (rc0.z * pre.x0[q] - wy - wt[q]) * sc;
            }
        }
        wloss += dot;  // per lane; summed over the wave once, by the caller
    }
}

// ------------------------------------------------------------------ spmm_rows
// U gathers in flight: the U rows of X are requested before the first is consumed.  FL: entries whose flag is
// zero point at operand rows that are zero and are not read (their lanes take no part in the load).
template <int LR, bool DUAL, int U, bool FL = false>
__device__ __forceinline__ void gather_rows(float4& acc, const int c, const float v, const float vt, const int it,
                                            const int g, const int lr, const float* __restrict__ Xg, const int f = 1) {
    constexpr int W = LR * 4, G = 64 / LR, HALF = LR / 2;
    float4 x[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = (it + u) * G + g;
        const int cj = __shfl(c, idx);
        float vj = __shfl(v, idx);
        if (DUAL) {
            const float vtj = __shfl(vt, idx);
            vj = (lr >= HALF) ? vtj : vj;
        }
        w[u] = vj;
        if (FL) {
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (__shfl(f, idx)) x[u] = *reinterpret_cast<const float4*>(Xg + static_cast<int64_t>(cj) * W + lr * 4);
        } else {
            x[u] = *reinterpret_cast<const float4*>(Xg + static_cast<int64_t>(cj) * W + lr * 4);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc.x = fmaf(w[u], x[u].x, acc.x);
        acc.y = fmaf(w[u], x[u].y, acc.y);
        acc.z = fmaf(w[u], x[u].z, acc.z);
        acc.w = fmaf(w[u], x[u].w, acc.w);
    }
}

// The stored entries [p0, p1) of one row, gathered from global memory by the whole wave: lane (g, lr) ends up
// with the partial sum of its group's entries over its four columns (the G groups still to be added).
// FL: the sweeps of the affinity preparation multiply operands whose rows are mostly zero: a byte per operand row
// says whether it holds anything, zero rows are neither stored nor read (the sums are the same bit for bit);
// any_live (wave-uniform) tells whether any entry pointed at a row that holds something.
template <int LR, bool DUAL, bool FL>
__device__ __forceinline__ float4 gather_range(const SpmmArgs& a, const int64_t p0, const int64_t p1, const int lane, bool& any_live) {
    constexpr int G = 64 / LR;      // rows of X gathered per wave instruction
    const int g = lane / LR, lr = lane % LR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p = p0; p < p1; p += 64) {
        const int cnt = static_cast<int>(min(static_cast<int64_t>(64), p1 - p));
        int c = 0, f = 0;
        float v = 0.f, vt = 0.f;
        if (lane < cnt) {
            c = a.col[p + lane];
            v = a.val[p + lane];
            if (DUAL) vt = a.valT[p + lane];
            if (FL) f = a.gflag[c];
        }
        // idle lanes point at a row that is read anyway and carry a zero weight
        const int c_first = __shfl(c, 0);
        if (lane >= cnt) c = c_first;
        const int iters = (cnt + G - 1) / G;  // wave-uniform; (iters - 1) * G + g <= 63
        int it = 0;
        if (FL) {
            const unsigned long long live = __ballot(f != 0);
            if (live == 0ull) continue;
            any_live = true;
            auto has = [&](const int first, const int n) {  // any flagged entry in iterations [first, first + n)
                return ((live >> (first * G)) & (n * G >= 64 ? ~0ull : ((1ull << (n * G)) - 1ull))) != 0ull;
            };
            if (G <= 8)
                for (; it + 8 <= iters; it += 8)
                    if (has(it, 8)) gather_rows<LR, DUAL, 8, true>(acc, c, v, vt, it, g, lr, a.Xg, f);
            if (it + 4 <= iters) { if (has(it, 4)) gather_rows<LR, DUAL, 4, true>(acc, c, v, vt, it, g, lr, a.Xg, f); it += 4; }
            if (it + 2 <= iters) { if (has(it, 2)) gather_rows<LR, DUAL, 2, true>(acc, c, v, vt, it, g, lr, a.Xg, f); it += 2; }
            if (it < iters && has(it, 1)) gather_rows<LR, DUAL, 1, true>(acc, c, v, vt, it, g, lr, a.Xg, f);
            continue;
        }
        if (G <= 8)
            for (; it + 8 <= iters; it += 8) gather_rows<LR, DUAL, 8>(acc, c, v, vt, it, g, lr, a.Xg);
        if (it + 4 <= iters) { gather_rows<LR, DUAL, 4>(acc, c, v, vt, it, g, lr, a.Xg); it += 4; }
        if (it + 2 <= iters) { gather_rows<LR, DUAL, 2>(acc, c, v, vt, it, g, lr, a.Xg); it += 2; }
        if (it < iters) gather_rows<LR, DUAL, 1>(acc, c, v, vt, it, g, lr, a.Xg);
    }
    return acc;
}

template <int LR>
__device__ __forceinline__ void merge_groups(float4& acc) {   // the sum over the wave's G groups, in every lane
#pragma unroll
    for (int off = LR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
}

// One row, gathered from global memory by the whole wave.
template <int LR, bool DUAL, int EPI>
__device__ __forceinline__ void wave_row_global(const SpmmArgs& a, const int64_t row, const int lane, double& wloss) {
    constexpr int W = LR * 4;       // floats per row of X
    constexpr int HALF = LR / 2;
    const int g = lane / LR, lr = lane % LR;
    const int64_t p0 = a.skip_entries ? 0 : a.rowptr[row], p1 = a.skip_entries ? 0 : a.rowptr[row + 1];
    // a split product: the rows above the limit are cut into chunks that have waves of their own (spmm_chunks)
    if ((EPI == EPI_STORE || EPI == EPI_ADD_X0) && a.split_above > 0 && p1 - p0 > a.split_above) return;
    const RowPre<4> pre = row_preload<4, EPI>(a, row, a.X0 ? a.X0 + row * W + lr * 4 : nullptr);
    constexpr bool FL = EPI == EPI_ADD_IDENT || EPI == EPI_PREP_ACC;
    bool any_live = false;  // wave-uniform
    float4 acc = gather_range<LR, DUAL, FL>(a, p0, p1, lane, any_live);
    if (EPI == EPI_PREP_ACC && !any_live) return;  // a zero row of S[:, J] adds nothing to the statistics
    merge_groups<LR>(acc);
    const bool right = DUAL && lr >= HALF;
    const int cl = (DUAL ? (lr % HALF) : lr) * 4;
    const LapConst<4> lc = lap_const<4, EPI>(a, cl, right);
    if (a.addend) {  // the part of the row's product that another kernel has formed (dense diagonal blocks)
        const float4 t = *reinterpret_cast<const float4*>(a.addend + row * W + lr * 4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    float accv[4] = {acc.x, acc.y, acc.z, acc.w};
    row_epilogue<4, DUAL, EPI>(a, row, lane, g == 0, HALF, cl, right, pre, lc, accv,
                               a.out ? a.out + row * W + lr * 4 : nullptr, a.col0 + lr * 4, wloss);
}

// ------------------------------------------------------------------ long rows in chunks
// A graph with hub rows (R-MAT: rows of several hundred thousand entries among rows of a few dozen) keeps one
// wave busy with a hub for milliseconds after every other wave has finished.  A split product cuts every row
// above a limit into chunks of at most that many entries: one wave per chunk writes the chunk's partial sum
// (spmm_chunks), and the chunks of a row are added in their order, with the row's own operand, by spmm_merge:
// the result does not depend on which wave ran when.  The row kernel skips those rows (SpmmArgs::split_above).
template <int LR>
__global__ __launch_bounds__(256) void spmm_chunks(const SpmmArgs a, const int64_t* __restrict__ chunk_begin,
                                                   const int64_t* __restrict__ chunk_end, int64_t nchunks,
                                                   float* __restrict__ partial) {
    constexpr int W = LR * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t id = static_cast<int64_t>(blockIdx.x) * 4 + wave;   // neighbouring chunks go to different XCDs
    if (id >= nchunks) return;
    bool any_live = false;
    float4 acc = gather_range<LR, false, false>(a, chunk_begin[id], chunk_end[id], lane, any_live);
    merge_groups<LR>(acc);
    if (lane < LR) *reinterpret_cast<float4*>(partial + id * W + lane * 4) = acc;
}

// thread = (long row, four columns): its chunks' partial sums in chunk order, plus the own-row operand
template <int LR>
__global__ __launch_bounds__(256) void spmm_merge(const SpmmArgs a, const int32_t* __restrict__ long_row,
                                                  const int64_t* __restrict__ long_first, int64_t nlong,
                                                  const float* __restrict__ partial) {
    constexpr int W = LR * 4;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t i = t / LR;
    const int lr = static_cast<int>(t % LR);
    if (i >= nlong) return;
    const int64_t row = long_row[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t c = long_first[i]; c < long_first[i + 1]; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(partial + c * W + lr * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (a.X0) {
        const float4 v = *reinterpret_cast<const float4*>(a.X0 + row * W + lr * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(a.out + row * W + lr * 4) = acc;
}

template <int LR, bool DUAL, int EPI>
__global__ __launch_bounds__(256) void spmm_rows(const SpmmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // neighbouring rows meet in one XCD's L2 (cluster batches), unless the caller wants them dealt to all XCDs in
    // turn: rows ordered by degree would otherwise hand one XCD all the work
    const int64_t blk = ((EPI == EPI_STORE || EPI == EPI_ADD_X0) && a.interleave_rows) ? static_cast<int64_t>(blockIdx.x)
                                                                                         : xcd_contiguous_block(blockIdx.x, gridDim.x);
    if (EPI == EPI_ADD_X0) {   // this workgroup's share of the column-sum finish, out of the way before the rows
        __shared__ double fs[6];
        FinishShare<2> fin;
        fin.request(a.fin, wave, lane, 4);
        fin.finish(a.fin, wave, lane, 4, fs);
    }
    double wloss = 0.0;
    for (int rr = wave; rr < ROWS_PER_BLOCK; rr += 4) {
        const int64_t row = blk * ROWS_PER_BLOCK + rr;
        if (row >= a.m) break;
        wave_row_global<LR, DUAL, EPI>(a, row, lane, wloss);
    }
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        __shared__ double wl[4];
        __shared__ unsigned int last_flag;
        wloss = wave_sum(wloss);
        if (lane == 0) wl[wave] = wloss;
        __syncthreads();
        if (a.rows_finish) {   // (grids of a few hundred workgroups: the arrival counter costs less than a launch)
            if (threadIdx.x == 0) publish(a.loss_part + blk, (wl[0] + wl[1]) + (wl[2] + wl[3]));
            finish_loss(a, gridDim.x, wl, &last_flag);
        } else if (threadIdx.x == 0) {
            a.loss_part[blk] = (wl[0] + wl[1]) + (wl[2] + wl[3]);
        }
    }
}

// ------------------------------------------------------------------ spmm_win
// One workgroup = (window of rows, column slab, part of the window's rows).  The dense operand is kept
// slab-major for this kernel: slab s is a row-major [m][16 VEC] matrix (dual: 8 VEC columns of the A
// half, then the same 8 VEC columns of the A^T half), so a window's rows of one slab are one contiguous
// piece of memory, staged in LDS by asynchronous global->LDS copies.  A wave works on FOUR rows at a
// time, one per 16-lane DPP row: lane (g, lr) owns VEC columns of the slab of row g.  The 16 lanes of a
// DPP row hold 16 consecutive stored entries; step u broadcasts entry u inside each DPP row
// (row_newbcast) and gathers the operand row with one ds_read (entries whose column lies in the window,
// i.e. nearly all of a cluster batch) or one global load (the others).  Rows arrive sorted by length, so
// the four rows of a wave are about equally long and the first bounds the trip count.
template <int VEC> struct VecOf;
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };

struct Chunk {  // 16 entries outside the window, one per lane of the DPP row
    int addr;   // the operand row
    float v, vt;
};

struct Chunk4 {  // 64 in-window entries, four consecutive ones per lane of the DPP row
    int addr[4];  // byte offset of the operand row inside the LDS window
    float v[4], vt[4];
};

// The entry streams of a tiling are laid out in the order the lanes consume them and padded with
// zero-weight entries, so a lane fetches its share with one 16-byte load per stream and no bounds logic.
// WT, the weights of an entry: 0 = one (val), 1 = two (val for the A half, valT for the A^T half), 2 = none
// (pattern mode: the stored value is a product of a row factor and a column factor that the caller folds into
// the epilogue and the operand, so the gather is a plain sum of operand rows).
template <int WT>
__device__ __forceinline__ Chunk4 load_entries4(const SpmmArgs& a, const unsigned pos, const bool live) {
    Chunk4 ck;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ck.addr[q] = 0;
        ck.v[q] = 0.f;
        ck.vt[q] = 0.f;
    }
    if (live) {
        const int4 c = *reinterpret_cast<const int4*>(a.s_addr + pos);
        ck.addr[0] = c.x; ck.addr[1] = c.y; ck.addr[2] = c.z; ck.addr[3] = c.w;
        if (WT != 2) {
            const float4 v = *reinterpret_cast<const float4*>(a.s_val + pos);
            ck.v[0] = v.x; ck.v[1] = v.y; ck.v[2] = v.z; ck.v[3] = v.w;
        }
        if (WT == 1) {
            const float4 vt = *reinterpret_cast<const float4*>(a.s_valT + pos);
            ck.vt[0] = vt.x; ck.vt[1] = vt.y; ck.vt[2] = vt.z; ck.vt[3] = vt.w;
        }
    }
    return ck;
}

template <int WT>
__device__ __forceinline__ Chunk load_outside(const SpmmArgs& a, const unsigned pos, const bool live) {
    Chunk ck;
    ck.addr = 0;
    ck.v = 0.f;
    ck.vt = 0.f;
    if (live) {
        ck.addr = a.o_row[pos];
        if (WT != 2) ck.v = a.o_val[pos];
        if (WT == 1) ck.vt = a.o_valT[pos];
    }
    return ck;
}

// The broadcasts are compiler-visible DPP operations (the wait states between a vector write and a DPP
// read of the same register are inserted by the compiler) and fold into the instructions that use them:
// the address is one v_add_u32 with a DPP source, and in the training pass the weight is two DPP moves
// into one register: val to every lane, then valT over the lanes of the A^T half (DPP banks 2-3).
template <int U>
__device__ __forceinline__ int bcast_add(const int x, const int base) {  // lane U of the DPP row + this lane's base
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + U, 0xf, 0xf, false) + base;
}

template <int WT, int U>
__device__ __forceinline__ float bcast_weight(const float v, const float vt) {
    int w = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + U, 0xf, 0xf, true);
    if (WT == 1) w = __builtin_amdgcn_update_dpp(w, __float_as_int(vt), 0x150 + U, 0xf, 0xc, false);
    return __int_as_float(w);
}

template <int VEC>
__device__ __forceinline__ void vec_fma(float (&acc)[VEC], const float w, const typename VecOf<VEC>::T x) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = fmaf(w, x[q], acc[q]);
}

template <int VEC>
__device__ __forceinline__ void vec_add(float (&acc)[VEC], const typename VecOf<VEC>::T x) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] += x[q];
}

template <int VEC>
__device__ __forceinline__ typename VecOf<VEC>::T lds_at(const int addr) {  // addr: LDS byte address
    typedef typename VecOf<VEC>::T V;
    return *(const __attribute__((address_space(3))) V*)(unsigned long)(unsigned)addr;
}

// The four in-window entries held by lane L of each DPP row: the four LDS reads are requested together and
// the products alternate between two accumulators.  lane_base = this lane's column offset inside a
// window row + the window's LDS base address.
template <int VEC, int WT, int L>
__device__ __forceinline__ void lds_group(float (&acc0)[VEC], float (&acc1)[VEC], const Chunk4& ck, const int lane_base) {
    typedef typename VecOf<VEC>::T V;
    typedef __attribute__((address_space(3))) const V LdsV;
    const int a0 = bcast_add<L>(ck.addr[0], lane_base), a1 = bcast_add<L>(ck.addr[1], lane_base);
    const int a2 = bcast_add<L>(ck.addr[2], lane_base), a3 = bcast_add<L>(ck.addr[3], lane_base);
    const V x0 = *(LdsV*)(unsigned long)(unsigned)a0;
    const V x1 = *(LdsV*)(unsigned long)(unsigned)a1;
    const V x2 = *(LdsV*)(unsigned long)(unsigned)a2;
    const V x3 = *(LdsV*)(unsigned long)(unsigned)a3;
    if (WT == 2) {
        vec_add<VEC>(acc0, x0);
        vec_add<VEC>(acc1, x1);
        vec_add<VEC>(acc0, x2);
        vec_add<VEC>(acc1, x3);
    } else {
        const float w0 = bcast_weight<WT, L>(ck.v[0], ck.vt[0]), w1 = bcast_weight<WT, L>(ck.v[1], ck.vt[1]);
        const float w2 = bcast_weight<WT, L>(ck.v[2], ck.vt[2]), w3 = bcast_weight<WT, L>(ck.v[3], ck.vt[3]);
        vec_fma<VEC>(acc0, w0, x0);
        vec_fma<VEC>(acc1, w1, x1);
        vec_fma<VEC>(acc0, w2, x2);
        vec_fma<VEC>(acc1, w3, x3);
    }
}

// The first `groups` (1..16, wave-uniform) lanes' entries of an in-window chunk (padding entries carry a zero weight)
template <int VEC, int WT>
__device__ __forceinline__ void lds_chunk(float (&acc0)[VEC], float (&acc1)[VEC], const Chunk4& ck, const int groups,
                                          const int lane_base) {
#define SPARC_G(L) if (groups > L) lds_group<VEC, WT, L>(acc0, acc1, ck, lane_base);
    SPARC_G(0) SPARC_G(1) SPARC_G(2) SPARC_G(3) SPARC_G(4) SPARC_G(5) SPARC_G(6) SPARC_G(7)
    SPARC_G(8) SPARC_G(9) SPARC_G(10) SPARC_G(11) SPARC_G(12) SPARC_G(13) SPARC_G(14) SPARC_G(15)
#undef SPARC_G
}

// Entries outside the window: the operand rows come from global memory (the slab's rows are 16 VEC floats
// apart), OG at a time (8 with the narrow lanes, 4 with the wide ones: registers).  Issue and use are separate
// so that the loads of a slot's first outside entries travel behind its LDS work.
#ifndef SPARC_OUTSIDE_DEPTH2
#define SPARC_OUTSIDE_DEPTH2 4
#endif
template <int VEC>
struct OutsideG {
    static constexpr int N = VEC == 2 ? SPARC_OUTSIDE_DEPTH2 : 4;
    typename VecOf<VEC>::T x[N];
};

template <int VEC, int U, int Q = 0>
__device__ __forceinline__ void outside_issue(OutsideG<VEC>& o, const Chunk& ck, const float* __restrict__ lane_ptr) {
    typedef typename VecOf<VEC>::T V;
    if constexpr (Q < OutsideG<VEC>::N) {
        const int r = bcast_add<(U + Q) & 15>(ck.addr, 0);
        o.x[Q] = *reinterpret_cast<const V*>(lane_ptr + static_cast<int64_t>(r) * (16 * VEC));
        outside_issue<VEC, U, Q + 1>(o, ck, lane_ptr);
    }
}

template <int VEC, int WT, int U, int Q = 0>
__device__ __forceinline__ void outside_use(float (&acc0)[VEC], float (&acc1)[VEC], const Chunk& ck, const OutsideG<VEC>& o) {
    if constexpr (Q < OutsideG<VEC>::N) {
        if (WT == 2) vec_add<VEC>((Q & 1) ? acc1 : acc0, o.x[Q]);
        else vec_fma<VEC>((Q & 1) ? acc1 : acc0, bcast_weight<WT, (U + Q) & 15>(ck.v, ck.vt), o.x[Q]);
        outside_use<VEC, WT, U, Q + 1>(acc0, acc1, ck, o);
    }
}

// steps [first, steps) of an outside chunk (first: a multiple of the group size), one group of loads at a time
template <int VEC, int WT>
__device__ __forceinline__ void outside_chunk(float (&acc0)[VEC], float (&acc1)[VEC], const Chunk& ck, const int first,
                                              const int steps, const float* __restrict__ lane_ptr) {
    constexpr int N = OutsideG<VEC>::N;
#define SPARC_O(U)                                               \
    if (U % N == 0 && first <= U && steps > U) {                 \
        OutsideG<VEC> o;                                         \
        outside_issue<VEC, U>(o, ck, lane_ptr);                  \
        outside_use<VEC, WT, U>(acc0, acc1, ck, o);              \
    }
    SPARC_O(0) SPARC_O(4) SPARC_O(8) SPARC_O(12)
#undef SPARC_O
}

// WM, how the stored values enter: 0 = val (and valT for the A^T half of a training pass), 1 = training pass on a
// numerically symmetric matrix (valT == val: one weight per entry for both halves), 2 = pattern mode: every
// off-diagonal value is A[i][j] = r_i (the supports the reference builds from a 0/1 adjacency: A + I,
// D^-1 (A + I), with or without the diagonal enhancement), so that
//     (A x)_i = r_i sum_j x_j + delta_i x_i,      (A^T x)_i = sum_j (r_j x_j) + delta_i x_i,      delta_i = A[i][i] - r_i
// (sums over the stored pattern of row i).  The gathered operand then carries the A^T half pre-multiplied by r
// (the epilogue writes it that way), the gather is a plain sum of operand rows (no weight streams, no
// weight broadcasts), and the row factors come back in the epilogue (a.rowpat: {r, delta, delta / r, 1 / r} per row).
// Padding entries of the in-window stream point at one extra, zeroed window row; outside entries keep a 0/1 weight.
// Ablation switches for the per-slot memory operations (SPARC_DBG bits 8 / 16 / 32: no entry / own-row / row-factor
// loads, 128: no store; timing experiments only, the results are wrong): compiled in only with -DSPARC_ABLATE
// (tools/build_variant.sh), because the widest instantiations have no register to spare for them.
#ifdef SPARC_ABLATE
#define SPARC_ABLATED(bit) ((a.dbg & (bit)) != 0)
#else
#define SPARC_ABLATED(bit) false
#endif
template <int VEC, bool DUAL, int EPI, int WM>
__global__ __launch_bounds__(WIN_MAX_THREADS) void spmm_win(const SpmmArgs a) {
    constexpr bool PAT = WM == 2;
    constexpr int WT = PAT ? 2 : (DUAL && WM == 0) ? 1 : 0;   // weights of the in-window entries
    constexpr int WTO = PAT ? 0 : WT;                          // ... and of the outside entries
    // the widest variants keep one slot's operands in registers at a time (no request one slot ahead)
    constexpr bool AHEAD = !(VEC == 4 && (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL));
    extern __shared__ float4 tile[];  // [window rows][16 lanes][VEC floats]
    __shared__ int4 info[WIN_MAX_RECORDS];  // the slot records of this workgroup
    typedef typename VecOf<VEC>::T V;
    constexpr int RW = 16 * VEC;      // floats per slab row
    constexpr int ROWB = RW * 4;      // bytes per slab row
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and known to be
    const int g = lane >> 4, lr = lane & 15;
    const int32_t* d = a.desc + static_cast<int64_t>(blockIdx.x) * 8;
    const int s_begin = d[0], s_end = d[1], w0 = d[2], wn = d[3] - d[2], slab = d[4], n_long = d[5];
    __shared__ double wl[WIN_MAX_THREADS / 64];
    __shared__ unsigned int last_flag;
    // this workgroup's share of the column-sum finish: requested now, summed behind the staging barrier (the
    // loads travel with the staging copies; held any longer they cost the slot loop its register budget: 80 -> 86
    // registers, one workgroup per CU instead of two, 14.4 -> 18.6 us)
    FinishShare<1> fin;
    if (EPI == EPI_ADD_X0) fin.request(a.fin, wave, lane, nwaves);
    if (s_end <= s_begin) {  // padding descriptor: nothing to add, but the workgroup is counted
        if (EPI == EPI_ADD_X0) fin.finish(a.fin, wave, lane, nwaves, wl);
        if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
            if (threadIdx.x == 0) publish(a.loss_part + blockIdx.x, 0.0);
            finish_loss(a, gridDim.x, wl, &last_flag);
        }
        return;
    }
    // columns of this lane inside its half (dual: 8 lanes for the A half, 8 for the A^T half)
    const bool right = DUAL && lr >= 8;
    const int cl = DUAL ? slab * (8 * VEC) + (lr & 7) * VEC : slab * RW + lr * VEC;
    const int64_t slab_off = static_cast<int64_t>(slab) * a.plane;
    const float* __restrict__ Xs = a.Xg + slab_off + lr * VEC;    // this lane's columns of operand row 0
    const float* __restrict__ X0s = a.X0 ? a.X0 + slab_off + lr * VEC : nullptr;
    float* __restrict__ outs = a.out ? a.out + slab_off + lr * VEC : nullptr;

    long long t_start = 0, t_bar = 0;
    if (a.stamps) t_start = __builtin_readcyclecounter();
    const int nrec = (s_end - s_begin) * 4;
    {   // stage the window and the slot records with asynchronous global->LDS copies (no register round trip):
        // one wave instruction moves 64 x 16 bytes to LDS base + lane * 16
        const float* src = a.Xg + slab_off + static_cast<int64_t>(w0) * RW;
        const int n16 = (a.dbg & 4) ? 0 : wn * (ROWB / 16);
        for (int base = wave * 64; base < n16; base += blockDim.x) {
            const int i = base + lane;
            if (i < n16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 4),
                                                 (__attribute__((address_space(3))) void*)(tile + base), 16, 0, 0);
        }
        const int4* rsrc = reinterpret_cast<const int4*>(a.rowinfo) + static_cast<int64_t>(s_begin) * 4;
        for (int base = wave * 64; base < nrec; base += blockDim.x) {
            const int i = base + lane;
            if (i < nrec)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rsrc + i),
                                                 (__attribute__((address_space(3))) void*)(info + base), 16, 0, 0);
        }
    }
    if (PAT && threadIdx.x < RW) reinterpret_cast<float*>(tile)[wn * RW + threadIdx.x] = 0.f;  // the row padding entries point at
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's copies into LDS have landed
    __syncthreads();                     // ... and so have everybody else's
    if (EPI == EPI_ADD_X0) fin.finish(a.fin, wave, lane, nwaves, wl);
    if (a.stamps) t_bar = __builtin_readcyclecounter();

    // A wave owns slots wave, wave + nwaves, ... sorted by work (longest first) and dealt in serpentine
    // order.  A slot is four rows, one per DPP row, or (the first n_long slots) one long row worked on by the
    // whole wave, whose four partial sums are merged at the end.  Record of (slot, DPP row) =
    // {row, position of the slot's in-window entries, position of its outside entries, L | O << 16}: every
    // lane lr < L holds four in-window entries, every lane lr < O one outside entry (both streams are padded
    // with zero-weight entries up to these wave-uniform counts).
    const int nslots = (a.dbg & 1) ? 0 : s_end - s_begin;
    auto slot_of = [&](const int round) { return round * nwaves + ((round & 1) ? nwaves - 1 - wave : wave); };
    struct Slot {
        int4 ri;
        Chunk4 c0;  // the first in-window chunk: all of a short row
        Chunk o0;   // the first outside chunk
        float x0[VEC];  // the row's own operand of the epilogue
        float4 rp;      // pattern mode: {r, delta, delta / r, 1 / r} of the row
    };
    const bool has_out = d[6] != 0;  // any of this workgroup's rows has entries outside the window
    const unsigned lane4 = (lr * 4 + g) * 4, lane1 = lr * 4 + g;  // this lane's place inside a chunk of either stream
    // Everything a slot reads from global memory is requested one slot ahead: global-memory latency hides
    // behind the LDS work of the slot before.
    auto fetch = [&](Slot& s, const int slot) {
        int4 ri = make_int4(-1, 0, 0, 0);
        if (slot < nslots) ri = info[slot * 4 + g];
        s.ri = ri;
        s.c0 = load_entries4<WT>(a, static_cast<unsigned>(ri.y) + lane4, lr < (ri.w & 0xffff) && !SPARC_ABLATED(8));
        s.o0.addr = 0; s.o0.v = 0.f; s.o0.vt = 0.f;
        if (has_out) s.o0 = load_outside<WTO>(a, static_cast<unsigned>(ri.z) + lane1, lr < (ri.w >> 16));
#pragma unroll
        for (int q = 0; q < VEC; ++q) s.x0[q] = 0.f;
        if (EPI != EPI_STORE && !SPARC_ABLATED(16)) load_vec<VEC>(s.x0, X0s + static_cast<int64_t>(ri.x >= 0 ? ri.x : 0) * RW);
        s.rp = make_float4(1.f, 0.f, 0.f, 0.f);
        if (PAT && !SPARC_ABLATED(32)) s.rp = *reinterpret_cast<const float4*>(a.rowpat + static_cast<int64_t>(ri.x >= 0 ? ri.x : 0) * 4);
    };

    const int lane_base = lr * (VEC * 4) + (int)(unsigned long)(__attribute__((address_space(3))) const char*)tile;
    const LapConst<VEC> lc = lap_const<VEC, EPI>(a, cl, right);
    double wloss = 0.0;
    auto work = [&](const Slot& cur, const int slot) {
        const bool lng = slot < n_long;  // wave-uniform
        const int4 ri = cur.ri;
        const bool valid = ri.x >= 0;
        const int groups = (a.dbg & 2) ? 0 : __builtin_amdgcn_readfirstlane(ri.w & 0xffff);
        const int64_t row = valid ? ri.x : 0;
        RowPre<VEC> pre = row_preload<VEC, EPI_STORE>(a, row, nullptr);  // the per-row constants arrive behind the LDS work
        if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
            pre.rc0 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC);
            pre.rc1 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC + 4);
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) pre.x0[q] = (PAT && right) ? cur.x0[q] * cur.rp.w : cur.x0[q];   // stored times r: back to its own scale
        float acc[VEC], acc1[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc1[q] = 0.f;
        OutsideG<VEC> og;
        int osteps = 0;
        if (has_out) {  // the first outside operand rows travel behind the LDS work
            osteps = __builtin_amdgcn_readfirstlane(ri.w >> 16);
            if (osteps > 0) outside_issue<VEC, 0>(og, cur.o0, Xs);
        }
        if (groups > 0) lds_chunk<VEC, WT>(acc, acc1, cur.c0, min(groups, 16), lane_base);
        if (groups > 16) {  // rows of more than 256 in-window entries: the next chunk is requested before this one is used
            Chunk4 ck = load_entries4<WT>(a, static_cast<unsigned>(ri.y) + 256 + lane4, lr + 16 < groups);
            for (int c = 1; c * 16 < groups; ++c) {
                const Chunk4 cn = load_entries4<WT>(a, static_cast<unsigned>(ri.y) + (c + 1) * 256 + lane4, lr + (c + 1) * 16 < groups);
                lds_chunk<VEC, WT>(acc, acc1, ck, min(groups - c * 16, 16), lane_base);
                ck = cn;
            }
        }
        if (osteps > 0) {
            outside_use<VEC, WTO, 0>(acc, acc1, cur.o0, og);
            outside_chunk<VEC, WTO>(acc, acc1, cur.o0, OutsideG<VEC>::N, min(osteps, 16), Xs);
            for (int c = 1; c * 16 < osteps; ++c) {
                const Chunk ck = load_outside<WTO>(a, static_cast<unsigned>(ri.z) + c * 64 + lane1, lr + c * 16 < osteps);
                outside_chunk<VEC, WTO>(acc, acc1, ck, 0, min(osteps - c * 16, 16), Xs);
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += acc1[q];
        if (lng) {  // merge the four DPP rows
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                acc[q] += __shfl_xor(acc[q], 16);
                acc[q] += __shfl_xor(acc[q], 32);
            }
        }
        const bool owner = valid && (!lng || g == 0);
        float oscale = 1.f;
        if (PAT) {  // back to the matrix's own scale: the row factor, and the diagonal's share beyond it
            const V own = lds_at<VEC>(lane_base + (valid ? static_cast<int>(row) - w0 : 0) * ROWB);
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = right ? fmaf(cur.rp.z, own[q], acc[q]) : fmaf(cur.rp.x, acc[q], cur.rp.y * own[q]);
            if (right) oscale = cur.rp.x;   // the next pass gathers the A^T half pre-multiplied by r
        }
        row_epilogue<VEC, DUAL, EPI>(a, row, lane, owner && !SPARC_ABLATED(128), 8, cl, right, pre, lc, acc, outs ? outs + row * RW : nullptr, 0,
                                     wloss, oscale);
    };
    if (AHEAD) {  // two slots in turn, so that no register that a load is still writing has to be copied
        Slot sa, sb;
        fetch(sa, slot_of(0));
        for (int round = 0; round * nwaves < nslots; round += 2) {  // wave-uniform
            fetch(sb, slot_of(round + 1));
            work(sa, slot_of(round));
            if ((round + 1) * nwaves >= nslots) break;
            fetch(sa, slot_of(round + 2));
            work(sb, slot_of(round + 1));
        }
    } else {
        Slot sa;
        for (int round = 0; round * nwaves < nslots; ++round) {  // wave-uniform
            fetch(sa, slot_of(round));
            work(sa, slot_of(round));
        }
    }

    if (a.stamps && lane == 0) {  // developer timeline: per wave {start, after the barrier, end, slots}
        long long* st = a.stamps + (static_cast<int64_t>(blockIdx.x) * 16 + wave) * 4;
        st[0] = t_start; st[1] = t_bar; st[2] = __builtin_readcyclecounter(); st[3] = nslots;
    }
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        wloss = wave_sum(wloss);
        if (lane == 0) wl[wave] = wloss;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < nwaves; ++w) t += wl[w];
            publish(a.loss_part + blockIdx.x, t);
        }
        finish_loss(a, gridDim.x, wl, &last_flag);
    }
}

// ------------------------------------------------------------------ small kernels of the Laplacian pass
// X = [Y/sqrt(d) | Y/(sqrt(d) rho)] (zero padded to KP columns per half) and the block partial column
// sums V^T Yn (3 rows) and U^T X2 (3 rows).  Column c of half h of row r goes to
// X[(c / hw) * plane + r * rw + h * hw + c % hw]: hw = KP, rw = row width, one slab for the row kernels;
// hw columns of each half per slab for the windowed kernel.
template <int KP, bool DUAL>
__global__ __launch_bounds__(256) void lap_scale(int64_t m, int k, int64_t ldy, const float* __restrict__ Y,
                                                 const float* __restrict__ rowc, float* __restrict__ X, int hw, int rw,
                                                 int64_t plane, float* __restrict__ colpart, int64_t rows_per_block,
                                                 const float* __restrict__ rowpat) {
    // a thread owns four consecutive columns (hw is a multiple of 4: they share a slab) of every RG-th row:
    // 16-byte stores into the slab-major operand
    constexpr int TPR = KP / 4;         // threads per row
    constexpr int RG = 256 / TPR;       // rows per sweep of the workgroup
    constexpr int NS = DUAL ? 6 : 3;
    const int c4 = (threadIdx.x % TPR) * 4, rg = threadIdx.x / TPR;
    const int64_t base = blockIdx.x * rows_per_block;
    const int64_t end = min(m, base + rows_per_block);
    const int64_t xoff = static_cast<int64_t>(c4 / hw) * plane + c4 % hw;
    float s[6][4];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) s[q][u] = 0.f;
    if (rg < RG) {
        // the pass is latency bound: the loads of B rows are requested before the first of them is used
#ifndef SPARC_SCALE_AHEAD
#define SPARC_SCALE_AHEAD 4
#endif
        constexpr int B = SPARC_SCALE_AHEAD;
        for (int64_t r0 = base + rg; r0 < end; r0 += B * RG) {
            float4 rc0[B], rc1[B];
            float yv[B][4], rf[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int64_t r = min(r0 + j * RG, end - 1);   // (rows past the end re-read the last row and are skipped below)
                rc0[j] = *reinterpret_cast<const float4*>(rowc + r * SPARC_ROWC);
                rc1[j] = *reinterpret_cast<const float4*>(rowc + r * SPARC_ROWC + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) yv[j][u] = (c4 + u < k) ? Y[r * ldy + c4 + u] : 0.f;
                rf[j] = (DUAL && rowpat) ? rowpat[r * 4] : 1.f;
            }
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int64_t r = r0 + j * RG;
                if (r >= end) break;
                float y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) y[u] = yv[j][u] * rc0[j].w;
                *reinterpret_cast<float4*>(X + xoff + r * rw) = make_float4(y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s[0][u] += y[u];
                    s[1][u] += rc1[j].z * y[u];
                    s[2][u] += rc1[j].w * y[u];
                }
                if (DUAL) {
                    // pattern mode of the windowed kernel: the A^T half is kept multiplied by the row factor, in X as
                    // in every later operand of the chain (the epilogues divide the own-row operand back)
                    const float irho = fabsf(rc0[j].x);
                    float x2s[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float x2 = y[u] * irho;
                        x2s[u] = x2 * rf[j];
                        s[3][u] += x2;
                        s[4][u] += rc1[j].x * x2;
                        s[5][u] += rc1[j].y * x2;
                    }
                    *reinterpret_cast<float4*>(X + xoff + r * rw + hw) = make_float4(x2s[0], x2s[1], x2s[2], x2s[3]);
                }
            }
        }
    }
    __shared__ float sm[RG][6][KP];
    if (rg < RG) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) sm[rg][q][c4 + u] = s[q][u];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NS * KP; e += 256) {  // fixed order over the row groups
        const int q = e / KP, col = e % KP;
        float t = 0.f;
        for (int j = 0; j < RG; ++j) t += sm[j][q][col];
        colpart[(blockIdx.x * 6 + q) * KP + col] = t;
    }
}

// ------------------------------------------------------------------ launch helpers
// Optional timing of one kernel class with HIP events recorded on the launch stream.  One profiling
// session per process at a time; the mutex only keeps concurrent launches from corrupting the table.
struct Profiler {
    std::mutex mu;
    int epi = -1;                 // which epilogue class is timed
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;
    int64_t launches = 0;         // kernel launches covered by the pairs
} g_prof;

// One start/stop pair around `count` back-to-back launches of the timed class (the two inner passes of
// the Laplacian chain share a pair, which halves the share of the event overhead in the average).
struct Timed {
    size_t slot = SIZE_MAX;
    hipStream_t s;
    Timed(int epi, hipStream_t stream, int count) : s(stream) {
        if (g_prof.epi < 0) return;  // no session: no lock on the launch path
        std::lock_guard<std::mutex> lock(g_prof.mu);
        if (g_prof.epi == epi && g_prof.used + 2 <= g_prof.ev.size()) {
            slot = g_prof.used;
            g_prof.used += 2;
            g_prof.launches += count;
            (void)hipEventRecord(g_prof.ev[slot], s);
        }
    }
    ~Timed() {
        if (slot != SIZE_MAX) (void)hipEventRecord(g_prof.ev[slot + 1], s);
    }
};

template <int LR, bool DUAL, int EPI>
int launch_rows(const SpmmArgs& a, hipStream_t stream) {
    const int64_t nb = (a.m + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (nb <= 0) return SPARC_OK;
    if (nb > 0x7fffffffLL) return SPARC_ERR_ARG;
    if (EPI == EPI_ADD_X0 && a.fin.colpart && nb < a.fin.kp) return SPARC_ERR_ARG;   // (the caller checks: fin_can_ride)
    hipLaunchKernelGGL((spmm_rows<LR, DUAL, EPI>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, stream, a);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

long long* g_stamps = nullptr;  // developer timeline buffer (SPARC_DBG & 64), set by sparc_debug_stamps

// Per device and per instantiation: allow the large dynamic LDS allocation.
template <int VEC, bool DUAL, int EPI, int WM>
int configure_win() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SPARC_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_win<VEC, DUAL, EPI, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SPARC_TILE_LDS_BYTES + 256) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        done[dev] = true;
    }
    return SPARC_OK;
}

template <int VEC, bool DUAL, int EPI, int WM = 0>
int launch_win(const SpmmArgs& a, hipStream_t stream) {
    if (a.ndesc <= 0) return SPARC_OK;
    const size_t lds = static_cast<size_t>(a.tile_rows + (WM == 2 ? 1 : 0)) * 64 * VEC;   // pattern mode: one more, zeroed row
    if (lds > SPARC_TILE_LDS_BYTES + 256 || a.threads < 64 || a.threads > WIN_MAX_THREADS || a.threads % 64) return SPARC_ERR_ARG;
    const int rc = configure_win<VEC, DUAL, EPI, WM>();
    if (rc) return rc;
    SpmmArgs b = a;
    static const int dbg = [] {  // developer switch for kernel ablation timings, read once
        const char* e = getenv("SPARC_DBG");
        return e ? atoi(e) : 0;
    }();
    b.dbg = dbg;
    b.stamps = (dbg & 64) && EPI == EPI_ADD_X0 ? g_stamps : nullptr;
    // the fused-loss variants of the narrow lanes need more registers: eight waves per workgroup keep two
    // workgroups on a CU (the plan does not depend on the width: a wave takes slots wave, wave + nwaves, ...)
    const bool lap = EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL;
    const int threads = (lap && VEC == 2 && a.threads > 512) ? 512 : a.threads;
    hipLaunchKernelGGL((spmm_win<VEC, DUAL, EPI, WM>), dim3(static_cast<unsigned>(a.ndesc)), dim3(threads), lds, stream, b);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

template <int EPI>
int launch_single(int width, const SpmmArgs& a, hipStream_t s) {
    if (a.desc) {
        if (a.rowpat) return a.vec == 2 ? launch_win<2, false, EPI, 2>(a, s) : launch_win<4, false, EPI, 2>(a, s);
        return a.vec == 2 ? launch_win<2, false, EPI>(a, s) : launch_win<4, false, EPI>(a, s);
    }
    switch (width) {
        case 16: return launch_rows<4, false, EPI>(a, s);
        case 32: return launch_rows<8, false, EPI>(a, s);
        case 64: return launch_rows<16, false, EPI>(a, s);
        case 128: return launch_rows<32, false, EPI>(a, s);
    }
    return SPARC_ERR_ARG;
}

template <int EPI>
int launch_dual(int kp, const SpmmArgs& a, hipStream_t s) {
    if (a.desc) {
        if (a.rowpat) return a.vec == 2 ? launch_win<2, true, EPI, 2>(a, s) : launch_win<4, true, EPI, 2>(a, s);
        if (a.s_valT == a.s_val && a.o_valT == a.o_val)   // numerically symmetric matrix: one weight per entry
            return a.vec == 2 ? launch_win<2, true, EPI, 1>(a, s) : launch_win<4, true, EPI, 1>(a, s);
        return a.vec == 2 ? launch_win<2, true, EPI>(a, s) : launch_win<4, true, EPI>(a, s);
    }
    switch (kp) {
        case 16: return launch_rows<8, true, EPI>(a, s);
        case 32: return launch_rows<16, true, EPI>(a, s);
        case 64: return launch_rows<32, true, EPI>(a, s);
    }
    return SPARC_ERR_ARG;
}

// Checks a tiling against the pass it is used for and fills the SpmmArgs fields.  dual: the training pass
// ([A half | A^T half], kp columns each); otherwise kp columns in all.
int set_tiling(SpmmArgs& a, const sparc_tiling_t* t, int64_t m, bool dual, int kp) {
    a.desc = nullptr;
    a.rowinfo = nullptr;
    a.rowpat = nullptr;
    a.ndesc = 0;
    if (!t) return SPARC_OK;
    const int hw = dual ? 8 * t->vec : 16 * t->vec;   // columns of a half (dual) or of the row (single) per slab
    if (!t->buf || t->ndesc <= 0 || (t->vec != 2 && t->vec != 4) || t->rows <= 0 ||
        static_cast<size_t>(t->rows) * 64 * t->vec > SPARC_TILE_LDS_BYTES || (t->dual != 0) != dual || t->kp != kp ||
        kp % hw != 0 || t->threads < 64 || t->threads > WIN_MAX_THREADS || t->threads % 64)
        return SPARC_ERR_ARG;
    const bool pattern = t->s_val == nullptr;   // no weight streams: every off-diagonal value is its row's factor
    if (!t->s_addr || !t->o_row || !t->o_val) return SPARC_ERR_ARG;
    if (pattern ? !t->row_scale : ((dual && !t->s_valT) || (dual && !t->o_valT))) return SPARC_ERR_ARG;
    a.rowpat = pattern ? t->row_scale : nullptr;
    a.desc = t->buf;
    a.rowinfo = t->buf + static_cast<int64_t>(t->ndesc) * 8;  // the slot records follow the descriptors
    a.s_addr = t->s_addr; a.s_val = t->s_val; a.s_valT = t->s_valT;
    a.o_row = t->o_row; a.o_val = t->o_val; a.o_valT = t->o_valT;
    a.ndesc = t->ndesc;
    a.tile_rows = t->rows;
    a.vec = t->vec;
    a.threads = t->threads;
    a.plane = m * 16 * t->vec;
    return SPARC_OK;
}

}  // namespace

int64_t sparc_rows_blocks(int64_t m) { return (m + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }

int sparc_spmm_launch(int epi, int width, const SpmmArgs& a, hipStream_t stream) {
    Timed t(epi, stream, 1);
    switch (epi) {
        case EPI_STORE: return launch_single<EPI_STORE>(width, a, stream);
        case EPI_ADD_X0: return launch_single<EPI_ADD_X0>(width, a, stream);
        case EPI_ADD_IDENT: return width == 64 && !a.desc && a.gflag && a.oflag ? launch_single<EPI_ADD_IDENT>(64, a, stream) : SPARC_ERR_ARG;
        case EPI_PREP_ACC: return width == 64 && !a.desc && a.gflag ? launch_single<EPI_PREP_ACC>(64, a, stream) : SPARC_ERR_ARG;
    }
    return SPARC_ERR_ARG;
}

extern "C" int sparc_debug_stamps(long long* buf) {  // not part of the public header: kernel timeline experiments
    g_stamps = buf;
    return SPARC_OK;
}

extern "C" int sparc_profile_begin(int epilogue, int max_launches) {
    if (max_launches < 0 || epilogue < 0 || epilogue > EPI_PREP_ACC) return SPARC_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign(static_cast<size_t>(max_launches) * 2, nullptr);
    for (auto& e : g_prof.ev)
        if (hipEventCreate(&e) != hipSuccess) return SPARC_ERR_LAUNCH;
    g_prof.used = 0;
    g_prof.launches = 0;
    g_prof.epi = epilogue;
    return SPARC_OK;
}

extern "C" int sparc_profile_end(double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    double total = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return SPARC_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return SPARC_ERR_LAUNCH;
        total += ms;
    }
    const int64_t n = g_prof.launches;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    g_prof.epi = -1;
    if (total_ms) *total_ms = total;
    if (launches) *launches = n;
    return SPARC_OK;
}

extern "C" int sparc_spmm(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                          int width, const float* X, const float* X0, float* Z, const sparc_tiling_t* tiling,
                          sparc_stream_t stream) {
    if (m < 0 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !val)) || !X || !Z) return SPARC_ERR_ARG;
    SpmmArgs a = {};
    a.m = m; a.nnz = nnz; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = val;
    a.Xg = X; a.X0 = X0; a.out = Z;
    a.kp = width;
    // a tiled plain product works on one slab: the row-major operand is the slab
    if (tiling && (tiling->kp != width || width != 16 * tiling->vec)) return SPARC_ERR_ARG;
    const int rc = set_tiling(a, tiling, m, false, width);
    if (rc) return rc;
    return sparc_spmm_launch(X0 ? EPI_ADD_X0 : EPI_STORE, width, a, stream);
}

namespace {
template <int LR>
int launch_split(const SpmmArgs& a, int64_t nlong, const int32_t* long_row, const int64_t* long_first, int64_t nchunks,
                 const int64_t* chunk_begin, const int64_t* chunk_end, float* partial, hipStream_t stream) {
    const int64_t cb = (nchunks + 3) / 4, mb = (nlong * LR + 255) / 256;
    if (cb > 0x7fffffffLL || mb > 0x7fffffffLL) return SPARC_ERR_ARG;
    hipLaunchKernelGGL((spmm_chunks<LR>), dim3(static_cast<unsigned>(cb)), dim3(256), 0, stream, a, chunk_begin, chunk_end, nchunks, partial);
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL((spmm_merge<LR>), dim3(static_cast<unsigned>(mb)), dim3(256), 0, stream, a, long_row, long_first, nlong, partial);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_spmm_split(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                                int width, const float* X, const float* X0, float* Z, int64_t split_above, int64_t nlong,
                                const int32_t* long_row, const int64_t* long_first, int64_t nchunks,
                                const int64_t* chunk_begin, const int64_t* chunk_end, float* partial, sparc_stream_t stream) {
    if (m < 0 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !val)) || !X || !Z || split_above < 64 || nlong < 0 || nchunks < nlong ||
        (nlong > 0 && (!long_row || !long_first || !chunk_begin || !chunk_end || !partial)) ||
        (width != 16 && width != 32 && width != 64 && width != 128))
        return SPARC_ERR_ARG;
    SpmmArgs a = {};
    a.m = m; a.nnz = nnz; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = val;
    a.Xg = X; a.X0 = X0; a.out = Z;
    a.kp = width;
    a.split_above = split_above;
    a.interleave_rows = 1;
    // the hub rows first: their chunks are the long poles, the short rows fill in behind them
    if (nlong > 0) {
        int rc = SPARC_ERR_ARG;
        switch (width) {
            case 16: rc = launch_split<4>(a, nlong, long_row, long_first, nchunks, chunk_begin, chunk_end, partial, stream); break;
            case 32: rc = launch_split<8>(a, nlong, long_row, long_first, nchunks, chunk_begin, chunk_end, partial, stream); break;
            case 64: rc = launch_split<16>(a, nlong, long_row, long_first, nchunks, chunk_begin, chunk_end, partial, stream); break;
            case 128: rc = launch_split<32>(a, nlong, long_row, long_first, nchunks, chunk_begin, chunk_end, partial, stream); break;
        }
        if (rc) return rc;
    }
    return sparc_spmm_launch(X0 ? EPI_ADD_X0 : EPI_STORE, width, a, stream);
}

namespace {
constexpr int SCALE_MAX_BLOCKS = SPARC_SCALE_MAX_BLOCKS;
// Columns per half: the windowed kernel takes multiples of 16, the row kernels 16, 32 or 64.
int padded_k(int k, bool tiled) { return tiled ? (k + 15) / 16 * 16 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64)); }

struct LapPlan {
    int64_t nscale, rows_per_block, nloss;
    size_t bytes;
};
LapPlan lap_plan(int64_t m, int kp, int ntiles) {
    LapPlan p;
    const int rg = 256 / (kp / 4);   // rows per sweep of a lap_scale workgroup
    int64_t nb = (m + 63) / 64;
    if (nb > SCALE_MAX_BLOCKS) nb = SCALE_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    int64_t rpb = (m + nb - 1) / nb;
    rpb = (rpb + rg - 1) / rg * rg;
    if (rpb < rg) rpb = rg;
    p.rows_per_block = rpb;
    p.nscale = (m + rpb - 1) / rpb;
    if (p.nscale < 1) p.nscale = 1;
    p.nloss = sparc_rows_blocks(m);
    if (ntiles > p.nloss) p.nloss = ntiles;
    p.bytes = 3 * sparc_align_up(static_cast<size_t>(m) * 2 * kp * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(SCALE_MAX_BLOCKS) * 6 * kp * sizeof(float)) +
              sparc_align_up((6 * SPARC_MAX_K + 16) * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(p.nloss > 0 ? p.nloss : 1) * sizeof(double));
    return p;
}
}  // namespace

extern "C" size_t sparc_lap_workspace_bytes(int64_t m, int k, int ntiles) {
    if (m < 0 || k < 1 || k > SPARC_MAX_K || ntiles < 0) return 0;
    return lap_plan(m, SPARC_MAX_K, ntiles).bytes;   // covers the column padding of every kernel variant
}

namespace {
// The Laplacian pass on the row kernel (tiling == blocks == NULL), the windowed kernel (tiling) or dense
// diagonal blocks plus the row kernel on the remaining entries (blocks; the CSR arrays then hold only those).
int lap_fwd_impl(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                 const float* valT, const float* rowc, const double* coupling, int k, int64_t ldy,
                 const float* Y, double* loss, float* gradY, const sparc_tiling_t* tiling, const sparc_blocks_t* blocks,
                 void* ws, size_t ws_bytes, hipStream_t stream) {
    if (m <= 0 || nnz < 0 || k < 1 || k > SPARC_MAX_K || ldy < k || !rowptr || !rowc || !coupling || !Y || !loss ||
        (nnz > 0 && (!colidx || !val)))
        return SPARC_ERR_ARG;
    const bool dual = gradY != nullptr;
    const bool tiled = tiling != nullptr;
    const int kp = tiled ? tiling->kp : padded_k(k, false);
    if (kp < k || kp > SPARC_MAX_K || kp % 16) return SPARC_ERR_ARG;
    BlocksArgs bl = {};
    bool fuse = false;   // the inner passes gather the entries outside the blocks in the block kernel's own launch
    if (blocks) {
        const int Wb = dual ? 2 * kp : kp;
        const bool mirrored = dual && blocks->blkT && blocks->blkT != blocks->blk;
        if (tiled || !blocks->desc || blocks->ndesc <= 0 || !blocks->blk || Wb % 32 || (mirrored && kp % 32) ||
            blocks->max_p < 32 || blocks->max_p % 32 || sparc_blocks_lds_bytes(blocks->max_p, Wb, mirrored, false) > SPARC_BLOCKS_LDS_BYTES)
            return SPARC_ERR_ARG;
        bl.desc = blocks->desc; bl.ndesc = blocks->ndesc; bl.max_p = blocks->max_p;
        bl.blk = blocks->blk; bl.blkT = mirrored ? blocks->blkT : blocks->blk;
        bl.W = Wb; bl.split = dual ? kp : Wb;
        bl.r_rowptr = rowptr; bl.r_col = colidx; bl.r_val = val; bl.r_valT = (dual && valT) ? valT : val;
        fuse = sparc_blocks_can_fuse(blocks->max_p, Wb, mirrored);
    }
    const LapPlan p = lap_plan(m, kp, tiled ? tiling->ndesc : 0);
    if (p.bytes > sparc_lap_workspace_bytes(m, k, tiled ? tiling->ndesc : 0)) return SPARC_ERR_ARG;
    WsCarver c(ws, ws_bytes);
    const int W = dual ? 2 * kp : kp;
    float* X = c.take<float>(static_cast<size_t>(m) * 2 * kp);
    float* P1 = c.take<float>(static_cast<size_t>(m) * 2 * kp);
    float* P2 = c.take<float>(static_cast<size_t>(m) * 2 * kp);
    float* colpart = c.take<float>(static_cast<size_t>(SCALE_MAX_BLOCKS) * 6 * kp);
    float* M12 = c.take<float>(6 * SPARC_MAX_K + 16);
    double* loss_part = c.take<double>(static_cast<size_t>(p.nloss));
    if (!ws || !c.ok()) return SPARC_ERR_WS;

    SpmmArgs a = {};
    a.m = m; a.nnz = nnz; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = valT ? valT : val;
    a.rowc = rowc; a.M12 = M12; a.loss_part = loss_part; a.gradY = gradY; a.ldy = ldy; a.k = k; a.kp = kp;
    a.inv_m = static_cast<float>(1.0 / static_cast<double>(m));
    a.loss_out = loss;
    a.loss_scale = 1.0 / static_cast<double>(m);
    a.chain_slot = static_cast<int>(g_chain_next.fetch_add(1u, std::memory_order_relaxed) % CHAIN_SLOTS);
    int rcode = set_tiling(a, tiling, m, dual, kp);
    if (rcode) return rcode;
    const int64_t nparts = tiled ? tiling->ndesc : sparc_rows_blocks(m);
    // operand layout: one slab of full rows for the row kernels, slabs of hw columns per half otherwise
    const int hw = tiled ? (dual ? 8 : 16) * tiling->vec : kp;
    const int rw = tiled ? 16 * tiling->vec : W;
    const int64_t plane = tiled ? a.plane : 0;

    const dim3 gs(static_cast<unsigned>(p.nscale)), bs(256);
#define SPARC_SCALE(KP)                                                                                              \
    if (dual)                                                                                                        \
        hipLaunchKernelGGL((lap_scale<KP, true>), gs, bs, 0, stream, m, k, ldy, Y, rowc, X, hw, rw, plane, colpart,   \
                           p.rows_per_block, a.rowpat);                                                               \
    else                                                                                                             \
        hipLaunchKernelGGL((lap_scale<KP, false>), gs, bs, 0, stream, m, k, ldy, Y, rowc, X, hw, rw, plane, colpart,  \
                           p.rows_per_block, a.rowpat);
    if (kp == 16) { SPARC_SCALE(16) } else if (kp == 32) { SPARC_SCALE(32) } else if (kp == 48) { SPARC_SCALE(48) } else { SPARC_SCALE(64) }
#undef SPARC_SCALE
    SPARC_CHECK_LAUNCH();
    ColsumFinish fin = {colpart, coupling, M12, kp, static_cast<int>(p.nscale), dual ? 1 : 0};
    // rides with the first inner pass when that pass has a workgroup per column with enough waves for the six sums
    // (the fused block kernel: six of its gather waves)
    const bool rides = p.nscale <= SCALE_MAX_BLOCKS &&
                       (blocks ? (fuse && bl.ndesc >= kp)
                               : tiled ? (tiling->ndesc >= kp && tiling->threads >= 6 * 64) : sparc_rows_blocks(m) >= kp);
    if (rides) {
        if (blocks) bl.fin = fin; else a.fin = fin;
    } else {
        hipLaunchKernelGGL(lap_colsum_finish, dim3(1), dim3(min(FINISH_SPLIT, 1024 / (6 * kp)) * 6 * kp), 0, stream, fin);
        SPARC_CHECK_LAUNCH();
    }

    a.X0 = X;
    // Horner: chain(X) = A (X + A (X + A X)).  With dense blocks a pass is two launches: the blocks' product
    // (plus the own-row operand), then the row kernel over the remaining entries, which adds its gathers to it.
    {
        Timed both(EPI_ADD_X0, stream, 2);   // the two inner passes, back to back
        a.Xg = X; a.out = P1;
        if (blocks) {
            bl.Xg = X; bl.X0 = X; bl.out = P1;
            if ((rcode = sparc_blocks_launch(bl, true, fuse, stream))) return rcode;
            a.X0 = P1;   // each lane reads and rewrites its own elements only
        }
        if (!fuse) {
            rcode = dual ? launch_dual<EPI_ADD_X0>(kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
            if (rcode) return rcode;
        }
        a.fin = ColsumFinish{};
        bl.fin = ColsumFinish{};
        a.Xg = P1; a.out = P2;
        if (blocks) {
            bl.Xg = P1; bl.X0 = X; bl.out = P2;
            if ((rcode = sparc_blocks_launch(bl, true, fuse, stream))) return rcode;
            a.X0 = P2;
        }
        if (!fuse) {
            rcode = dual ? launch_dual<EPI_ADD_X0>(kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
            if (rcode) return rcode;
        }
    }
    a.Xg = P2; a.out = nullptr; a.X0 = X;
    // the row kernel's loss: summed by its last workgroup for grids of up to ROWS_FINISH_MAX_BLOCKS workgroups (6 k
    // workgroups: 39 + 5 us as two launches, 50 us as one), by loss_finish beyond
    const bool rows_finish = !tiled && sparc_rows_blocks(m) <= ROWS_FINISH_MAX_BLOCKS;
    a.rows_finish = rows_finish ? 1 : 0;
    // dense blocks, training pass: the block kernel's own launch carries the loss / gradient epilogue when its LDS has
    // room for the exchange of the two halves
    const bool blocks_lap = blocks && fuse && dual &&
                            sparc_blocks_lds_bytes(bl.max_p, bl.W, bl.blkT != bl.blk, true, true) <= SPARC_BLOCKS_LDS_BYTES;
    if (blocks_lap) {
        unsigned int* counters = chain_counters_on_current_device();
        if (!counters) return SPARC_ERR_LAUNCH;
        Timed last(EPI_LAP_DUAL, stream, 1);
        bl.Xg = P2; bl.X0 = X; bl.out = nullptr;
        bl.lap_rowc = rowc; bl.lap_M12 = M12; bl.lap_kp = kp; bl.lap_k = k; bl.lap_inv_m = a.inv_m;
        bl.lap_gradY = gradY; bl.lap_ldy = ldy; bl.lap_loss_part = loss_part; bl.lap_loss_out = loss;
        bl.lap_loss_scale = a.loss_scale;
        bl.lap_counter = counters + static_cast<size_t>(a.chain_slot) * CHAIN_WORDS;
        return sparc_blocks_launch(bl, false, true, stream);
    }
    {
        Timed last(dual ? EPI_LAP_DUAL : EPI_LAP_SINGLE, stream, 1);
        if (blocks) {
            bl.Xg = P2; bl.X0 = nullptr; bl.out = P1;   // P1 is free again
            if ((rcode = sparc_blocks_launch(bl, false, fuse, stream))) return rcode;
            a.addend = P1;
            a.skip_entries = fuse ? 1 : 0;   // fused: P1 is the whole product, the row kernel only runs its epilogue
        }
        rcode = dual ? launch_dual<EPI_LAP_DUAL>(kp, a, stream) : launch_single<EPI_LAP_SINGLE>(W, a, stream);
        if (rcode) return rcode;
    }
    if (!tiled && !rows_finish) {   // the windowed kernel's last workgroup has written the loss; the row kernel leaves partials
        hipLaunchKernelGGL(loss_finish, dim3(1), dim3(256), 0, stream, nparts, loss_part, 1.0 / static_cast<double>(m), loss);
        SPARC_CHECK_LAUNCH();
    }
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_lap_fwd(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                             const float* valT, const float* rowc, const double* coupling, int k, int64_t ldy,
                             const float* Y, double* loss, float* gradY, const sparc_tiling_t* tiling,
                             void* ws, size_t ws_bytes, sparc_stream_t stream) {
    return lap_fwd_impl(m, nnz, rowptr, colidx, val, valT, rowc, coupling, k, ldy, Y, loss, gradY, tiling, nullptr, ws, ws_bytes,
                        stream);
}

extern "C" int sparc_lap_fwd_blocks(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                                    const float* valT, const float* rowc, const double* coupling, int k, int64_t ldy,
                                    const float* Y, double* loss, float* gradY, const sparc_blocks_t* blocks,
                                    void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (!blocks) return SPARC_ERR_ARG;
    return lap_fwd_impl(m, nnz, rowptr, colidx, val, valT, rowc, coupling, k, ldy, Y, loss, gradY, nullptr, blocks, ws, ws_bytes,
                        stream);
}

extern "C" int sparc_blocks_spmm(int64_t m, int width, const sparc_blocks_t* blocks, const float* X, const float* X0, float* Z,
                                 sparc_stream_t stream) {
    if (m <= 0 || !blocks || !blocks->desc || !blocks->blk || !X || !Z || width % 32 || width < 32 || width > 128) return SPARC_ERR_ARG;
    BlocksArgs bl = {};
    bl.desc = blocks->desc; bl.ndesc = blocks->ndesc; bl.max_p = blocks->max_p;
    bl.blk = blocks->blk; bl.blkT = blocks->blk;
    bl.Xg = X; bl.X0 = X0; bl.out = Z; bl.W = width; bl.split = width;
    return sparc_blocks_launch(bl, X0 != nullptr, false, stream);
}
