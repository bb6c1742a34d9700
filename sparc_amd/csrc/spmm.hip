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
#include <cstdlib>
#include <vector>

#include "spmm.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // spmm_rows: one row per wave
constexpr int TILED_THREADS = 1024;
constexpr int TILED_WAVES = TILED_THREADS / 64;

// ------------------------------------------------------------------ shared epilogue
// A lane owns 4 consecutive columns of one row: `gcol` is their offset inside the row of the
// global operand (row stride wtot floats), `cl` their index inside the half (0..kp-1), `right`
// tells the A^T half from the A half, `half` is the lane distance between the two halves.
struct RowPre {  // what the epilogue reads of its own row, requested before the gathers
    float4 x0, rc0, rc1;
};

struct LapConst {  // per-lane constants of the Laplacian epilogue: rows of M1 / M2 and the 3x3 coupling
    float4 ma, mb, mc;
    float cf[9];
};

template <int EPI>
__device__ __forceinline__ LapConst lap_const(const SpmmArgs& a, const int kp, const int cl, const bool right) {
    LapConst c;
    c.ma = c.mb = c.mc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 9; ++i) c.cf[i] = 0.f;
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        const float* Mb = a.M12 + (right ? 3 * kp : 0) + cl;
        c.ma = *reinterpret_cast<const float4*>(Mb);
        c.mb = *reinterpret_cast<const float4*>(Mb + kp);
        c.mc = *reinterpret_cast<const float4*>(Mb + 2 * kp);
#pragma unroll
        for (int i = 0; i < 9; ++i) c.cf[i] = a.M12[6 * kp + i];
    }
    return c;
}

template <int EPI>
__device__ __forceinline__ RowPre row_preload(const SpmmArgs& a, const int64_t row, const int wtot, const int gcol) {
    RowPre r;
    r.x0 = r.rc0 = r.rc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_ADD_X0 || EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL)
        r.x0 = *reinterpret_cast<const float4*>(a.X0 + row * wtot + gcol);
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        r.rc0 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC);
        r.rc1 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC + 4);
    }
    return r;
}

// ROWS = rows per wave: 1 (the wave's owner lanes hold one row) or 4 (one row per 16-lane group).
template <int ROWS>
__device__ __forceinline__ double rows_sum(double v) {
    if (ROWS == 1) return wave_sum(v);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <bool DUAL, int EPI, int ROWS = 1>
__device__ __forceinline__ void row_epilogue(const SpmmArgs& a, const int64_t row, const int lane, const bool owner,
                                             const int half, const int kp, const int wtot, const int gcol,
                                             const int cl, const bool right, const RowPre& pre, const LapConst& lc,
                                             float4 acc, double& wloss) {
    const int64_t ro = row * wtot + gcol;
    if (EPI == EPI_STORE) {
        if (owner) *reinterpret_cast<float4*>(a.out + ro) = acc;
    } else if (EPI == EPI_ADD_X0) {
        if (owner) {
            const float4 x0 = pre.x0;
            *reinterpret_cast<float4*>(a.out + ro) = make_float4(acc.x + x0.x, acc.y + x0.y, acc.z + x0.z, acc.w + x0.w);
        }
    } else if (EPI == EPI_ADD_IDENT) {
        if (owner) {
            const int64_t cb = a.col0 + gcol;
            acc.x += (cb + 0 == row) ? 1.f : 0.f;
            acc.y += (cb + 1 == row) ? 1.f : 0.f;
            acc.z += (cb + 2 == row) ? 1.f : 0.f;
            acc.w += (cb + 3 == row) ? 1.f : 0.f;
            *reinterpret_cast<float4*>(a.out + ro) = acc;
        }
    } else if (EPI == EPI_PREP_ACC) {
        // acc holds S[row, col0 + gcol .. +3]; gather the row statistics of S
        double ss = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (owner) {
            const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t cj = a.col0 + gcol + q;
                if (cj < a.m && cj != row) {
                    const double sv = static_cast<double>(vals[q]);
                    ss += sv * sv;
                    s0 += sv;
                    s1 += sv * a.cvec[cj];
                    s2 += sv * a.c2vec[cj];
                }
            }
        }
        ss = rows_sum<ROWS>(ss); s0 = rows_sum<ROWS>(s0); s1 = rows_sum<ROWS>(s1); s2 = rows_sum<ROWS>(s2);
        if (owner && (ROWS == 1 ? lane == 0 : (lane & 15) == 0)) {
            double* r = a.racc + row * 8;
            r[0] += ss; r[1] += s0; r[2] += s1; r[3] += s2;
        }
    } else {  // EPI_LAP_SINGLE / EPI_LAP_DUAL
        // rowc: {+-1/rho, S_ii, d+cw, 1/sqrt(d)}, {r, r2, c, c2}; a negative 1/rho marks a row without
        // off-diagonal entries, whose chain result is exactly S_ii x and is dropped instead of cancelled
        const float4 rc0 = pre.rc0, rc1 = pre.rc1, x0 = pre.x0;
        const bool isolated = rc0.x < 0.f;
        const float irho = fabsf(rc0.x);
        const float* Cf = lc.cf;  // the 3x3 coupling, row-major
        // diagonal of U C V^T for this row
        const float ldiag = (Cf[0] + Cf[1] * rc1.z + Cf[2] * rc1.w) + rc1.x * (Cf[3] + Cf[4] * rc1.z + Cf[5] * rc1.w) +
                            rc1.y * (Cf[6] + Cf[7] * rc1.z + Cf[8] * rc1.w);
        const float u1 = right ? rc1.z : rc1.x, u2 = right ? rc1.w : rc1.y;
        const float4 ma = lc.ma, mb = lc.mb, mc = lc.mc;
        float4 t;
        t.x = (isolated ? 0.f : acc.x - rc0.y * x0.x) + ((ma.x + u1 * mb.x + u2 * mc.x) - ldiag * x0.x);
        t.y = (isolated ? 0.f : acc.y - rc0.y * x0.y) + ((ma.y + u1 * mb.y + u2 * mc.y) - ldiag * x0.y);
        t.z = (isolated ? 0.f : acc.z - rc0.y * x0.z) + ((ma.z + u1 * mb.z + u2 * mc.z) - ldiag * x0.z);
        t.w = (isolated ? 0.f : acc.w - rc0.y * x0.w) + ((ma.w + u1 * mb.w + u2 * mc.w) - ldiag * x0.w);
        float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DUAL) {  // W^T Yn sits `half` lanes to the right of W Yn
            wt.x = __shfl(t.x, lane + half);
            wt.y = __shfl(t.y, lane + half);
            wt.z = __shfl(t.z, lane + half);
            wt.w = __shfl(t.w, lane + half);
        }
        double dot = 0.0;
        if (owner && !right) {
            const float4 wy = make_float4(irho * t.x, irho * t.y, irho * t.z, irho * t.w);
            const float hd = 0.5f * rc0.z;
            const float ly0 = hd * x0.x - wy.x, ly1 = hd * x0.y - wy.y, ly2 = hd * x0.z - wy.z, ly3 = hd * x0.w - wy.w;
            dot = static_cast<double>(x0.x) * ly0 + static_cast<double>(x0.y) * ly1 +
                  static_cast<double>(x0.z) * ly2 + static_cast<double>(x0.w) * ly3;
            if (DUAL) {
                const float sc = a.inv_m * rc0.w;
                const float gy[4] = {(rc0.z * x0.x - wy.x - wt.x) * sc, (rc0.z * x0.y - wy.y - wt.y) * sc,
                                     (rc0.z * x0.z - wy.z - wt.z) * sc, (rc0.z * x0.w - wy.w - wt.w) * sc};
                float* gp = a.gradY + row * a.ldy + cl;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (cl + q < a.k) gp[q] = gy[q];
            }
        }
        wloss += dot;  // per lane; summed over the wave once, by the caller
    }
}

// ------------------------------------------------------------------ spmm_rows
// U gathers in flight: the U rows of X are requested before the first is consumed.
template <int LR, bool DUAL, int U>
__device__ __forceinline__ void gather_rows(float4& acc, const int c, const float v, const float vt, const int it,
                                            const int g, const int lr, const float* __restrict__ Xg) {
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

// One row, gathered from global memory by the whole wave.
template <int LR, bool DUAL, int EPI>
__device__ __forceinline__ void wave_row_global(const SpmmArgs& a, const int64_t row, const int lane, double& wloss) {
    constexpr int W = LR * 4;       // floats per row of X
    constexpr int G = 64 / LR;      // rows of X gathered per wave instruction
    constexpr int HALF = LR / 2;
    const int g = lane / LR, lr = lane % LR;
    const int64_t p0 = a.rowptr[row], p1 = a.rowptr[row + 1];
    const RowPre pre = row_preload<EPI>(a, row, W, lr * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p = p0; p < p1; p += 64) {
        const int cnt = static_cast<int>(min(static_cast<int64_t>(64), p1 - p));
        int c = 0;
        float v = 0.f, vt = 0.f;
        if (lane < cnt) {
            c = a.col[p + lane];
            v = a.val[p + lane];
            if (DUAL) vt = a.valT[p + lane];
        }
        // idle lanes point at a row that is read anyway and carry a zero weight
        const int c_first = __shfl(c, 0);
        if (lane >= cnt) c = c_first;
        const int iters = (cnt + G - 1) / G;  // wave-uniform; (iters - 1) * G + g <= 63
        int it = 0;
        if (G <= 8)
            for (; it + 8 <= iters; it += 8) gather_rows<LR, DUAL, 8>(acc, c, v, vt, it, g, lr, a.Xg);
        if (it + 4 <= iters) { gather_rows<LR, DUAL, 4>(acc, c, v, vt, it, g, lr, a.Xg); it += 4; }
        if (it + 2 <= iters) { gather_rows<LR, DUAL, 2>(acc, c, v, vt, it, g, lr, a.Xg); it += 2; }
        if (it < iters) gather_rows<LR, DUAL, 1>(acc, c, v, vt, it, g, lr, a.Xg);
    }
#pragma unroll
    for (int off = LR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    const bool right = DUAL && lr >= HALF;
    const int kp = DUAL ? W / 2 : W;
    const int cl = (DUAL ? (lr % HALF) : lr) * 4;
    const LapConst lc = lap_const<EPI>(a, kp, cl, right);
    row_epilogue<DUAL, EPI>(a, row, lane, g == 0, HALF, kp, W, lr * 4, cl, right, pre, lc, acc, wloss);
}

template <int LR, bool DUAL, int EPI>
__global__ __launch_bounds__(256) void spmm_rows(const SpmmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    double wloss = 0.0;
    for (int rr = wave; rr < ROWS_PER_BLOCK; rr += 4) {
        const int64_t row = blk * ROWS_PER_BLOCK + rr;
        if (row >= a.m) break;
        wave_row_global<LR, DUAL, EPI>(a, row, lane, wloss);
    }
    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        __shared__ double wl[4];
        wloss = wave_sum(wloss);
        if (lane == 0) wl[wave] = wloss;
        __syncthreads();
        if (threadIdx.x == 0) a.loss_part[blk] = (wl[0] + wl[1]) + (wl[2] + wl[3]);
    }
}

// ------------------------------------------------------------------ spmm_tiled
// One workgroup = (window of rows, column slab, part of the window's rows).  A wave works on FOUR
// rows at a time, one per 16-lane DPP row: lane (g, lr) owns 4 columns (16 bytes) of the slab of
// row g.  The 16 lanes of a row load 16 consecutive stored entries; step u broadcasts entry u
// inside each DPP row (v_mov_dpp row_newbcast) and gathers the operand row from the LDS window
// with one ds_read_b128.  Rows arrive sorted by length (rowinfo), so the four rows of a wave are
// equally long and row 0 of the wave bounds the trip count.  Rows with entries outside their
// window (rare in a cluster batch) are not tiled at all: workgroups whose descriptor carries
// slab < 0 gather them whole from global memory, so the hot loop has no second path.
template <int N>
__device__ __forceinline__ int row_bcast(const int x) {  // lane N of every 16-lane DPP row, to the whole row
    return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ float row_bcastf(const float x) {
    return __int_as_float(row_bcast<N>(__float_as_int(x)));
}

struct Chunk {  // 16 stored entries of one row, one per lane of the DPP row
    int addr;   // byte offset of the operand row inside the LDS window
    float v, vt;
};

template <bool DUAL>
__device__ __forceinline__ Chunk load_chunk(const SpmmArgs& a, const int lo, const int hi, const int base, const int lr,
                                            const int w0, const int own_addr) {
    Chunk ck;
    ck.addr = own_addr;  // idle lanes: a row of the window, zero weight
    ck.v = 0.f;
    ck.vt = 0.f;
    const int p = lo + base + lr;  // lr: position of the lane's entry inside the chunk
    if (p < hi) {
        ck.addr = (a.col[p] - w0) << 8;
        ck.v = a.val[p];
        if (DUAL) ck.vt = a.valT[p];
    }
    return ck;
}

// One entry of each of the wave's four rows.  The broadcasts are folded into the instructions that use
// them: the LDS address is one v_add_u32 with a DPP source (lane U of the row + this lane's column
// offset), and in the training pass the weight is two DPP moves into one register, val into the lanes
// of the A half (DPP banks 0-1) and valT into the lanes of the A^T half (banks 2-3): no select.
typedef float lds_vec4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const lds_vec4 lds_float4;

template <int U>
__device__ __forceinline__ int step_addr(const Chunk& ck, const int lane_base) {
    // lane U's row offset + this lane's column offset + the window's LDS base: an absolute LDS address.
    // Compiler-visible DPP (folded into one v_add_u32_dpp), so that the VALU-write -> DPP-read wait states
    // are inserted by the hazard recognizer.
    return __builtin_amdgcn_update_dpp(0, ck.addr, 0x150 + U, 0xf, 0xf, false) + lane_base;
}

template <bool DUAL, int U>
__device__ __forceinline__ float step_weight(const Chunk& ck) {
    if (DUAL) {  // val into DPP banks 0-1 (the A half), valT into banks 2-3 (the A^T half)
        int w = __builtin_amdgcn_update_dpp(0, __float_as_int(ck.v), 0x150 + U, 0xf, 0x3, false);
        w = __builtin_amdgcn_update_dpp(w, __float_as_int(ck.vt), 0x150 + U, 0xf, 0xc, false);
        return __int_as_float(w);
    }
    return row_bcastf<U>(ck.v);
}

__device__ __forceinline__ void step_fma(float4& acc, const float w, const lds_vec4 x) {
    acc.x = fmaf(w, x[0], acc.x);
    acc.y = fmaf(w, x[1], acc.y);
    acc.z = fmaf(w, x[2], acc.z);
    acc.w = fmaf(w, x[3], acc.w);
}

// Four entries of each of the wave's four rows.  The broadcasts are folded into the instructions that
// use them: the LDS address is one v_add_u32 with a DPP source, and in the training pass the weight is
// two DPP moves into one register, val into the lanes of the A half (DPP banks 0-1) and valT into the
// lanes of the A^T half (banks 2-3): no select.  The four LDS reads are requested together and the
// products alternate between two accumulators.
template <bool DUAL, int U>
__device__ __forceinline__ void tile_step4(float4& acc0, float4& acc1, const Chunk& ck, const int lane_base) {
    const int a0 = step_addr<U>(ck, lane_base), a1 = step_addr<U + 1>(ck, lane_base);
    const int a2 = step_addr<U + 2>(ck, lane_base), a3 = step_addr<U + 3>(ck, lane_base);
    const lds_vec4 x0 = *(lds_float4*)(unsigned long)(unsigned)a0;
    const lds_vec4 x1 = *(lds_float4*)(unsigned long)(unsigned)a1;
    const lds_vec4 x2 = *(lds_float4*)(unsigned long)(unsigned)a2;
    const lds_vec4 x3 = *(lds_float4*)(unsigned long)(unsigned)a3;
    const float w0 = step_weight<DUAL, U>(ck), w1 = step_weight<DUAL, U + 1>(ck);
    const float w2 = step_weight<DUAL, U + 2>(ck), w3 = step_weight<DUAL, U + 3>(ck);
    step_fma(acc0, w0, x0);
    step_fma(acc1, w1, x1);
    step_fma(acc0, w2, x2);
    step_fma(acc1, w3, x3);
}

// The same with two reads in flight and one accumulator: for the epilogues that leave few registers.
template <bool DUAL, int U>
__device__ __forceinline__ void tile_step2x2(float4& acc, const Chunk& ck, const int lane_base) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int a0 = h ? step_addr<U + 2>(ck, lane_base) : step_addr<U>(ck, lane_base);
        const int a1 = h ? step_addr<U + 3>(ck, lane_base) : step_addr<U + 1>(ck, lane_base);
        const lds_vec4 x0 = *(lds_float4*)(unsigned long)(unsigned)a0;
        const lds_vec4 x1 = *(lds_float4*)(unsigned long)(unsigned)a1;
        const float w0 = h ? step_weight<DUAL, U + 2>(ck) : step_weight<DUAL, U>(ck);
        const float w1 = h ? step_weight<DUAL, U + 3>(ck) : step_weight<DUAL, U + 1>(ck);
        step_fma(acc, w0, x0);
        step_fma(acc, w1, x1);
    }
}

// `steps` (1..16, wave-uniform, in an SGPR) entries of the chunk, four at a time (idle steps carry a
// zero weight); lane_base = this lane's column offset inside a window row + the window's LDS base
template <bool DUAL, bool WIDE>
__device__ __forceinline__ void tile_chunk(float4& acc, const Chunk& ck, const int steps, const int lane_base) {
    if (WIDE) {
        float4 acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (steps > 0) tile_step4<DUAL, 0>(acc, acc1, ck, lane_base);
        if (steps > 4) tile_step4<DUAL, 4>(acc, acc1, ck, lane_base);
        if (steps > 8) tile_step4<DUAL, 8>(acc, acc1, ck, lane_base);
        if (steps > 12) tile_step4<DUAL, 12>(acc, acc1, ck, lane_base);
        acc.x += acc1.x;
        acc.y += acc1.y;
        acc.z += acc1.z;
        acc.w += acc1.w;
    } else {
        if (steps > 0) tile_step2x2<DUAL, 0>(acc, ck, lane_base);
        if (steps > 4) tile_step2x2<DUAL, 4>(acc, ck, lane_base);
        if (steps > 8) tile_step2x2<DUAL, 8>(acc, ck, lane_base);
        if (steps > 12) tile_step2x2<DUAL, 12>(acc, ck, lane_base);
    }
}

template <bool DUAL, int EPI>
__global__ __launch_bounds__(TILED_THREADS) void spmm_tiled(const SpmmArgs a) {
    extern __shared__ float4 tile[];  // [window rows][16 lanes] float4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, lr = lane & 15;
    const int32_t* d = a.desc + static_cast<int64_t>(blockIdx.x) * 8;
    const int q_begin = d[0], q_end = d[1], w0 = d[2], wn = d[3] - d[2], slab = d[4];
    const int kp = a.kp, wtot = a.wtot;
    if (slab < 0) {
        // Rows with entries outside their window: this workgroup gathers them whole from global
        // memory (records [q_begin, q_end) of the row list, one row per wave at a time).
        double gl = 0.0;
        const int g_end = (a.dbg & 8) ? q_begin : q_end;
        for (int q = q_begin + wave; q < g_end; q += TILED_WAVES) {
            const int64_t row = a.rowinfo[12 * static_cast<int64_t>(q)];
            if (DUAL) {
                if (kp == 64) wave_row_global<32, true, EPI>(a, row, lane, gl);
                else wave_row_global<16, true, EPI>(a, row, lane, gl);
            } else {
                wave_row_global<16, false, EPI>(a, row, lane, gl);
            }
        }
        if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
            __shared__ double gw[TILED_WAVES];
            gl = wave_sum(gl);
            if (lane == 0) gw[wave] = gl;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (int w = 0; w < TILED_WAVES; ++w) t += gw[w];
                a.loss_part[blockIdx.x] = t;
            }
        }
        return;
    }
    // columns of this lane: dual = 8 lanes (32 floats) of the A half + 8 lanes of the A^T half,
    // single = 16 lanes (64 floats)
    const bool right = DUAL && lr >= 8;
    const int cl = DUAL ? slab * 32 + (lr & 7) * 4 : slab * 64 + lr * 4;
    const int gcol = (right ? kp : 0) + cl;

    // The records of this wave's first two slots are requested from global memory before anything else, so
    // that their entry chunks can be requested while the window is still being staged (every later slot
    // reads its record from the LDS copy made below).
    const int nq = q_end - q_begin;
    const int n_long = d[5];
    const int nslots = (a.dbg & 1) ? 0 : n_long + (nq - n_long + 3) / 4;
    auto slot_of = [&](const int round) { return round * TILED_WAVES + ((round & 1) ? TILED_WAVES - 1 - wave : wave); };
    auto record_index = [&](const int slot) { return slot < n_long ? slot : n_long + (slot - n_long) * 4 + g; };
    int4 early_ri[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int slot = slot_of(e), qi = record_index(slot);
        early_ri[e] = make_int4(-1, 0, 0, 0);
        if (slot < nslots && qi < nq)
            early_ri[e] = reinterpret_cast<const int4*>(a.rowinfo)[(static_cast<int64_t>(q_begin) + qi) * 3];
    }

    {   // stage the window with asynchronous global->LDS copies (no register round trip): one wave
        // instruction moves 64 x 16 bytes to LDS base + lane * 16, the source address is per lane
        const float* src = a.Xg + static_cast<int64_t>(w0) * wtot + gcol;
        const int n16 = (a.dbg & 4) ? 0 : wn * 16;
        for (int base = wave * 64; base < n16; base += TILED_THREADS) {
            const int i = base + lane;
            if (i < n16)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + static_cast<int64_t>(i >> 4) * wtot),
                    (__attribute__((address_space(3))) void*)(tile + base), 16, 0, 0);
        }
    }
    // the row records of this workgroup, in LDS: 3 x int4 per row =
    // {row, lo, hi, n_outside}, {col0, col1, val0, valT0}, {val1, valT1, 0, 0}
    __shared__ int4 info[SPARC_TILE_MAX_ROWS * 3];
    for (int i = threadIdx.x; i < nq * 3; i += TILED_THREADS)
        info[i] = reinterpret_cast<const int4*>(a.rowinfo)[static_cast<int64_t>(q_begin) * 3 + i];
    const char* tile_bytes = reinterpret_cast<const char*>(tile);
    constexpr bool WIDE = EPI != EPI_LAP_SINGLE && EPI != EPI_LAP_DUAL;   // the loss epilogues leave fewer registers
    // this lane's column offset inside a window row plus the window's LDS base address
    const int lr16 = lr * 16 + (int)(unsigned long)(__attribute__((address_space(3))) const char*)tile_bytes;
    const LapConst lc = lap_const<EPI>(a, kp, cl, right);
    // A wave owns slots wave, wave + 16, ... (4 rows each), sorted by work (longest first) and dealt in
    // serpentine order.  Slots: the first n_long records are long rows (more than SPARC_TILE_LONG_ROW
    // in-window entries), each worked on by the whole wave (DPP row g takes entries g, g + 4, ...; the four
    // partial sums are merged at the end); the other records are taken four at a time, one row per DPP row.
    // Up to two entries of a row may lie outside the window; their columns and values travel in the row
    // record and their operand rows come from global memory.
    struct Slot {
        int4 ri;
        Chunk c0, c1;
        RowPre pre;
        float ow0, ow1;
        float4 ox0, ox1;
    };
    // The entry chunks of a slot are requested two slots ahead: global-memory latency hides behind two
    // slots of LDS work (for the first two slots: behind the staging of the window).
    auto with_chunks = [&](const int4 ri, const int slot) {
        Slot s;
        const bool lng = slot < n_long;
        s.ri = ri;
        const int off = lng ? lr * 4 + g : lr;  // entry of this lane inside a chunk
        s.c0 = load_chunk<DUAL>(a, s.ri.y, s.ri.z, 0, off, w0, 0);
        s.c1 = load_chunk<DUAL>(a, s.ri.y, s.ri.z, lng ? 64 : 16, off, w0, 0);
        return s;
    };
    Slot cur = with_chunks(early_ri[0], slot_of(0)), nxt = with_chunks(early_ri[1], slot_of(1));
    __syncthreads();  // the window and the records have landed (the barrier waits for the LDS copies)

    // The epilogue operands and the outside operands of a slot are requested at the start of the slot's own
    // round and arrive behind its LDS gather (carrying them two slots ahead as well costs registers: spills
    // in the loss epilogues, and it measured slower for every epilogue).
    auto fetch_operands = [&](Slot& s, const int slot) {
        const bool lng = slot < n_long;
        const int qi = record_index(slot);
        s.ow0 = s.ow1 = 0.f;
        s.ox0 = s.ox1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.ri.x >= 0 && s.ri.w > 0 && (!lng || g == 0)) {
            const int4 o0 = info[qi * 3 + 1];
            const bool mirrored = right || a.rec_mirrored;   // which of the record's two values this lane multiplies by
            s.ow0 = __int_as_float(mirrored ? o0.w : o0.z);
            s.ox0 = *reinterpret_cast<const float4*>(a.Xg + static_cast<int64_t>(o0.x) * wtot + gcol);
            if (s.ri.w > 1) {
                const int4 o1 = info[qi * 3 + 2];
                s.ow1 = __int_as_float(mirrored ? o1.y : o1.x);
                s.ox1 = *reinterpret_cast<const float4*>(a.Xg + static_cast<int64_t>(o0.y) * wtot + gcol);
            }
        }
        s.pre = row_preload<EPI_STORE>(a, 0, 0, 0);
        if (s.ri.x >= 0) s.pre = row_preload<EPI>(a, s.ri.x, wtot, gcol);
    };
    auto fetch = [&](const int slot) {
        const int qi = record_index(slot);
        int4 ri = make_int4(-1, 0, 0, 0);
        if (slot < nslots && qi < nq) ri = info[qi * 3];
        return with_chunks(ri, slot);
    };

    double wloss = 0.0;
    for (int round = 0; round * TILED_WAVES < nslots; ++round) {  // wave-uniform
        const Slot nn = fetch(slot_of(round + 2));
        const bool lng = slot_of(round) < n_long;  // wave-uniform
        const int4 ri = cur.ri;
        // rows are sorted by length: DPP row 0 holds the longest of the four
        const int nmax = __builtin_amdgcn_readfirstlane(ri.z - ri.y);
        const int steps = (a.dbg & 2) ? 0 : (lng ? (nmax + 3) >> 2 : nmax);
        const int cstride = lng ? 64 : 16;  // entries per chunk
        const int off = lng ? lr * 4 + g : lr;
        fetch_operands(cur, slot_of(round));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (steps > 0) tile_chunk<DUAL, WIDE>(acc, cur.c0, min(steps, 16), lr16);
        if (steps > 16) tile_chunk<DUAL, WIDE>(acc, cur.c1, min(steps - 16, 16), lr16);
        for (int c = 2; c * 16 < steps; ++c) {
            const Chunk ck = load_chunk<DUAL>(a, ri.y, ri.z, c * cstride, off, w0, 0);
            tile_chunk<DUAL, WIDE>(acc, ck, min(steps - c * 16, 16), lr16);
        }
        // the (at most two) entries outside the window
        acc.x += fmaf(cur.ow1, cur.ox1.x, cur.ow0 * cur.ox0.x);
        acc.y += fmaf(cur.ow1, cur.ox1.y, cur.ow0 * cur.ox0.y);
        acc.z += fmaf(cur.ow1, cur.ox1.z, cur.ow0 * cur.ox0.z);
        acc.w += fmaf(cur.ow1, cur.ox1.w, cur.ow0 * cur.ox0.w);
        if (lng) {  // merge the four DPP rows
            acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16);
            acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
            acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
            acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        }
        row_epilogue<DUAL, EPI, 4>(a, ri.x, lane, ri.x >= 0 && (!lng || g == 0), 8, kp, wtot, gcol, cl, right, cur.pre,
                                   lc, acc, wloss);
        cur = nxt;
        nxt = nn;
    }

    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        __shared__ double wl[TILED_WAVES];
        wloss = wave_sum(wloss);
        if (lane == 0) wl[wave] = wloss;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < TILED_WAVES; ++w) t += wl[w];
            a.loss_part[blockIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------ small kernels of the Laplacian pass
// X = [Y/sqrt(d) | Y/(sqrt(d) rho)] (zero padded to KP columns per half) and the block
// partial column sums V^T Yn (3 rows) and U^T X2 (3 rows).
template <int KP, bool DUAL>
__global__ __launch_bounds__(256) void lap_scale(int64_t m, int k, int64_t ldy, const float* __restrict__ Y,
                                                 const float* __restrict__ rowc, float* __restrict__ X,
                                                 float* __restrict__ colpart, int64_t rows_per_block) {
    constexpr int RG = 256 / KP;
    constexpr int W = DUAL ? 2 * KP : KP;
    constexpr int NS = DUAL ? 6 : 3;
    const int col = threadIdx.x % KP, rg = threadIdx.x / KP;
    const int64_t base = blockIdx.x * rows_per_block;
    const int64_t end = min(m, base + rows_per_block);
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int64_t r = base + rg; r < end; r += RG) {
        const float4 rc0 = *reinterpret_cast<const float4*>(rowc + r * SPARC_ROWC);
        const float4 rc1 = *reinterpret_cast<const float4*>(rowc + r * SPARC_ROWC + 4);
        const float y = (col < k) ? Y[r * ldy + col] * rc0.w : 0.f;
        X[r * W + col] = y;
        s[0] += y;
        s[1] += rc1.z * y;
        s[2] += rc1.w * y;
        if (DUAL) {
            const float x2 = y * fabsf(rc0.x);
            X[r * W + KP + col] = x2;
            s[3] += x2;
            s[4] += rc1.x * x2;
            s[5] += rc1.y * x2;
        }
    }
    __shared__ float sm[RG][6][KP];
#pragma unroll
    for (int q = 0; q < NS; ++q) sm[rg][q][col] = s[q];
    __syncthreads();
    if (rg == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            float t = 0.f;
            for (int j = 0; j < RG; ++j) t += sm[j][q][col];
            colpart[(blockIdx.x * 6 + q) * KP + col] = t;
        }
    }
}

// M1 = C (V^T Yn), M2 = C^T (U^T X2): the 3 x KP matrices of the rank-3 correction.
__global__ void lap_colsum_finish(int KP, int nparts, bool dual, const float* __restrict__ colpart,
                                  const double* __restrict__ coupling, float* __restrict__ M12) {
    __shared__ double S[6][SPARC_MAX_K];
    const int q = threadIdx.x / KP, col = threadIdx.x % KP;
    const int ns = dual ? 6 : 3;
    if (q < ns) {
        double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // eight independent chains, fixed order
        int b = 0;
        for (; b + 8 <= nparts; b += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] += colpart[(static_cast<int64_t>(b + u) * 6 + q) * KP + col];
        }
        for (; b < nparts; ++b) t[0] += colpart[(static_cast<int64_t>(b) * 6 + q) * KP + col];
        S[q][col] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    __syncthreads();
    if (q < ns) {
        double t = 0.0;
        if (q < 3) {
            for (int b = 0; b < 3; ++b) t += coupling[q * 3 + b] * S[b][col];
        } else {
            for (int b = 0; b < 3; ++b) t += coupling[b * 3 + (q - 3)] * S[3 + b][col];
        }
        M12[q * KP + col] = static_cast<float>(t);
    }
    if (threadIdx.x < 9) M12[6 * KP + threadIdx.x] = static_cast<float>(coupling[threadIdx.x]);
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

// ------------------------------------------------------------------ launch helpers
// Optional timing of one kernel class with HIP events recorded on the launch stream.
struct Profiler {
    int epi = -1;                 // which epilogue class is timed
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;
    int64_t launches = 0;         // kernel launches covered by the pairs
    int depth = 0;                // > 0 inside a bracket: nested brackets do not record
} g_prof;

// One start/stop pair around `count` back-to-back launches of the timed class (the two inner passes of
// the Laplacian chain share a pair, which halves the share of the event overhead in the average).
struct Timed {
    bool on;
    hipStream_t s;
    int count;
    Timed(int epi, hipStream_t stream, int count_ = 1)
        : on(g_prof.depth == 0 && g_prof.epi == epi && g_prof.used + 2 <= g_prof.ev.size()), s(stream), count(count_) {
        ++g_prof.depth;
        if (on) (void)hipEventRecord(g_prof.ev[g_prof.used], s);
    }
    ~Timed() {
        --g_prof.depth;
        if (on) {
            (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
            g_prof.used += 2;
            g_prof.launches += count;
        }
    }
};

template <int LR, bool DUAL, int EPI>
int launch_rows(const SpmmArgs& a, hipStream_t stream) {
    const int64_t nb = (a.m + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (nb <= 0) return SPARC_OK;
    if (nb > 0x7fffffffLL) return SPARC_ERR_ARG;
    {
        Timed t(EPI, stream);
        hipLaunchKernelGGL((spmm_rows<LR, DUAL, EPI>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, stream, a);
    }
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

template <bool DUAL, int EPI>
int launch_tiled(const SpmmArgs& a, hipStream_t stream) {
    const size_t lds = static_cast<size_t>(a.tile_rows) * 16 * sizeof(float4);
    if (a.ndesc <= 0) return SPARC_OK;
    if (lds > SPARC_TILE_LDS_BYTES) return SPARC_ERR_ARG;
    static bool configured = false;  // per instantiation: allow the large dynamic LDS allocation
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_tiled<DUAL, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SPARC_TILE_LDS_BYTES) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        configured = true;
    }
    {
        SpmmArgs b = a;
        static const int dbg = [] {  // developer switch for kernel ablation timings, read once
            const char* e = getenv("SPARC_DBG");
            return e ? atoi(e) : 0;
        }();
        b.dbg = dbg;
        Timed t(EPI, stream);
        hipLaunchKernelGGL((spmm_tiled<DUAL, EPI>), dim3(static_cast<unsigned>(a.ndesc)), dim3(TILED_THREADS), lds, stream, b);
    }
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

template <int EPI>
int launch_single(int width, const SpmmArgs& a, hipStream_t s) {
    if (a.desc && width == 64) return launch_tiled<false, EPI>(a, s);
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
    if (a.desc && (kp == 32 || kp == 64)) return launch_tiled<true, EPI>(a, s);
    switch (kp) {
        case 16: return launch_rows<8, true, EPI>(a, s);
        case 32: return launch_rows<16, true, EPI>(a, s);
        case 64: return launch_rows<32, true, EPI>(a, s);
    }
    return SPARC_ERR_ARG;
}

int padded_k(int k) { return k <= 16 ? 16 : (k <= 32 ? 32 : 64); }

}  // namespace

int64_t sparc_rows_blocks(int64_t m) { return (m + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }

int sparc_spmm_launch(int epi, int width, const SpmmArgs& a, hipStream_t stream) {
    switch (epi) {
        case EPI_STORE: return launch_single<EPI_STORE>(width, a, stream);
        case EPI_ADD_X0: return launch_single<EPI_ADD_X0>(width, a, stream);
        case EPI_ADD_IDENT: return width == 64 ? launch_single<EPI_ADD_IDENT>(64, a, stream) : SPARC_ERR_ARG;
        case EPI_PREP_ACC: return width == 64 ? launch_single<EPI_PREP_ACC>(64, a, stream) : SPARC_ERR_ARG;
    }
    return SPARC_ERR_ARG;
}

extern "C" int sparc_profile_begin(int epilogue, int max_launches) {
    if (max_launches < 0 || epilogue < 0 || epilogue > EPI_PREP_ACC) return SPARC_ERR_ARG;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign(static_cast<size_t>(max_launches) * 2, nullptr);
    for (auto& e : g_prof.ev)
        if (hipEventCreate(&e) != hipSuccess) return SPARC_ERR_LAUNCH;
    g_prof.used = 0;
    g_prof.launches = 0;
    g_prof.depth = 0;
    g_prof.epi = epilogue;
    return SPARC_OK;
}

extern "C" int sparc_profile_end(double* total_ms, int64_t* launches) {
    double total = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return SPARC_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return SPARC_ERR_LAUNCH;
        total += ms;
    }
    n = g_prof.launches;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    g_prof.epi = -1;
    if (total_ms) *total_ms = total;
    if (launches) *launches = n;
    return SPARC_OK;
}

namespace {
// Checks the tiling arguments that can be checked on the host; fills the SpmmArgs fields.
int set_tiles(SpmmArgs& a, const int32_t* tiles, int ntiles, int tile_rows, int kp, int wtot) {
    a.kp = kp;
    a.wtot = wtot;
    a.desc = nullptr;
    a.rowinfo = nullptr;
    a.ndesc = 0;
    a.tile_rows = 0;
    if (!tiles) return SPARC_OK;
    if (ntiles <= 0 || tile_rows <= 0 || tile_rows > SPARC_TILE_MAX_ROWS) return SPARC_ERR_ARG;
    a.desc = tiles;
    a.rowinfo = tiles + static_cast<int64_t>(ntiles) * 8;  // the row records follow the descriptors
    a.ndesc = ntiles;
    a.tile_rows = tile_rows;
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_spmm(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                          int width, const float* X, const float* X0, float* Z, const int32_t* tiles, int ntiles,
                          int tile_rows, int val_is_mirrored, sparc_stream_t stream) {
    if (m < 0 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !val)) || !X || !Z) return SPARC_ERR_ARG;
    SpmmArgs a = {};
    a.m = m; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = val;
    a.Xg = X; a.X0 = X0; a.out = Z;
    a.rec_mirrored = val_is_mirrored != 0;
    if (tiles && width != 64) return SPARC_ERR_ARG;
    const int rc = set_tiles(a, tiles, ntiles, tile_rows, width, width);
    if (rc) return rc;
    return sparc_spmm_launch(X0 ? EPI_ADD_X0 : EPI_STORE, width, a, stream);
}

namespace {
constexpr int SCALE_MAX_BLOCKS = 256;
struct LapPlan {
    int kp, W;
    int64_t nscale, rows_per_block, nloss;
    size_t bytes;
};
LapPlan lap_plan(int64_t m, int k, int ntiles) {
    LapPlan p;
    p.kp = padded_k(k);
    p.W = 2 * p.kp;
    const int rg = 256 / p.kp;
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
    p.bytes = 3 * sparc_align_up(static_cast<size_t>(m) * p.W * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(SCALE_MAX_BLOCKS) * 6 * p.kp * sizeof(float)) +
              sparc_align_up((6 * SPARC_MAX_K + 16) * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(p.nloss > 0 ? p.nloss : 1) * sizeof(double));
    return p;
}
}  // namespace

extern "C" size_t sparc_lap_workspace_bytes(int64_t m, int k, int ntiles) {
    if (m < 0 || k < 1 || k > SPARC_MAX_K || ntiles < 0) return 0;
    return lap_plan(m, k, ntiles).bytes;
}

extern "C" int sparc_lap_fwd(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                             const float* valT, const float* rowc, const double* coupling, int k, int64_t ldy,
                             const float* Y, double* loss, float* gradY, const int32_t* tiles, int ntiles,
                             int tile_rows, void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m <= 0 || nnz < 0 || k < 1 || k > SPARC_MAX_K || ldy < k || !rowptr || !rowc || !coupling || !Y || !loss ||
        (nnz > 0 && (!colidx || !val)) || ntiles < 0)
        return SPARC_ERR_ARG;
    const LapPlan p = lap_plan(m, k, tiles ? ntiles : 0);
    WsCarver c(ws, ws_bytes);
    const bool dual = gradY != nullptr;
    const int W = dual ? p.W : p.kp;
    float* X = c.take<float>(static_cast<size_t>(m) * p.W);
    float* P1 = c.take<float>(static_cast<size_t>(m) * p.W);
    float* P2 = c.take<float>(static_cast<size_t>(m) * p.W);
    float* colpart = c.take<float>(static_cast<size_t>(SCALE_MAX_BLOCKS) * 6 * p.kp);
    float* M12 = c.take<float>(6 * SPARC_MAX_K + 16);
    double* loss_part = c.take<double>(static_cast<size_t>(p.nloss));
    if (!ws || !c.ok()) return SPARC_ERR_WS;

    SpmmArgs a = {};
    a.m = m; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = valT ? valT : val;
    a.rowc = rowc; a.M12 = M12; a.loss_part = loss_part; a.gradY = gradY; a.ldy = ldy; a.k = k;
    a.inv_m = static_cast<float>(1.0 / static_cast<double>(m));
    // the tiled kernel covers 64-float single rows and 32- or 64-float halves of dual rows
    const bool tiled = tiles && (dual ? p.kp >= 32 : p.kp == 64);
    int rcode = set_tiles(a, tiled ? tiles : nullptr, ntiles, tile_rows, p.kp, W);
    if (rcode) return rcode;
    const int64_t nparts = tiled ? ntiles : sparc_rows_blocks(m);

    const dim3 gs(static_cast<unsigned>(p.nscale)), bs(256);
#define SPARC_SCALE(KP)                                                                                              \
    if (dual)                                                                                                        \
        hipLaunchKernelGGL((lap_scale<KP, true>), gs, bs, 0, stream, m, k, ldy, Y, rowc, X, colpart, p.rows_per_block); \
    else                                                                                                             \
        hipLaunchKernelGGL((lap_scale<KP, false>), gs, bs, 0, stream, m, k, ldy, Y, rowc, X, colpart, p.rows_per_block);
    if (p.kp == 16) { SPARC_SCALE(16) } else if (p.kp == 32) { SPARC_SCALE(32) } else { SPARC_SCALE(64) }
#undef SPARC_SCALE
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL(lap_colsum_finish, dim3(1), dim3(6 * p.kp), 0, stream, p.kp, static_cast<int>(p.nscale), dual,
                       colpart, coupling, M12);
    SPARC_CHECK_LAUNCH();

    a.X0 = X;
    // Horner: chain(X) = A (X + A (X + A X))
    {
        Timed both(EPI_ADD_X0, stream, 2);   // the two inner passes, back to back
        a.Xg = X; a.out = P1;
        rcode = dual ? launch_dual<EPI_ADD_X0>(p.kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
        if (rcode) return rcode;
        a.Xg = P1; a.out = P2;
        rcode = dual ? launch_dual<EPI_ADD_X0>(p.kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
        if (rcode) return rcode;
    }
    a.Xg = P2; a.out = nullptr;
    rcode = dual ? launch_dual<EPI_LAP_DUAL>(p.kp, a, stream) : launch_single<EPI_LAP_SINGLE>(W, a, stream);
    if (rcode) return rcode;
    hipLaunchKernelGGL(loss_finish, dim3(1), dim3(256), 0, stream, nparts, loss_part, 1.0 / static_cast<double>(m), loss);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
