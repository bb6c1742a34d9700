// CSR row-gather product for cluster batches and the fused Laplacian/loss epilogue.
//
// One wavefront owns one row of the batch.  The row's column indices and values
// are streamed coalesced (one element per lane, 64 at a time) and handed to the
// consuming lanes with cross-lane reads; the dense operand X is gathered as
// whole rows, 16 bytes per lane, so that 64/LR rows of X arrive per wave
// instruction (LR = lanes that cover one row of X).  Partial sums of the 64/LR
// lane groups are combined with a butterfly at the end of the row.
//
// In "dual" mode X is [Yn | Yn/rho]: the left half is multiplied by A (val), the
// right half by A^T (valT, same pattern), so the forward product W Yn and the
// backward product W^T Yn share one pass over the index stream.
#include <vector>

#include "spmm.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // one row per wave

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

template <int LR, bool DUAL, int EPI>
__global__ __launch_bounds__(256) void spmm_rows(const SpmmArgs a) {
    constexpr int W = LR * 4;       // floats per row of X
    constexpr int G = 64 / LR;      // rows of X gathered per wave instruction
    constexpr int HALF = LR / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LR, lr = lane % LR;
    const int64_t blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    double wloss = 0.0;

    for (int rr = wave; rr < ROWS_PER_BLOCK; rr += 4) {
        const int64_t row = blk * ROWS_PER_BLOCK + rr;
        if (row >= a.m) break;
        const int64_t p0 = a.rowptr[row], p1 = a.rowptr[row + 1];
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
        const int64_t ro = row * W + lr * 4;

        if (EPI == EPI_STORE) {
            if (g == 0) *reinterpret_cast<float4*>(a.out + ro) = acc;
        } else if (EPI == EPI_ADD_X0) {
            if (g == 0) {
                const float4 x0 = *reinterpret_cast<const float4*>(a.X0 + ro);
                *reinterpret_cast<float4*>(a.out + ro) = make_float4(acc.x + x0.x, acc.y + x0.y, acc.z + x0.z, acc.w + x0.w);
            }
        } else if (EPI == EPI_ADD_IDENT) {
            if (g == 0) {
                const int64_t cb = a.col0 + lr * 4;
                acc.x += (cb + 0 == row) ? 1.f : 0.f;
                acc.y += (cb + 1 == row) ? 1.f : 0.f;
                acc.z += (cb + 2 == row) ? 1.f : 0.f;
                acc.w += (cb + 3 == row) ? 1.f : 0.f;
                *reinterpret_cast<float4*>(a.out + ro) = acc;
            }
        } else if (EPI == EPI_PREP_ACC) {
            // acc holds S[row, col0 + 4 lr .. +3]; gather the row statistics of S
            double ss = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, sd = 0.0;
            if (g == 0) {
                const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t cj = a.col0 + lr * 4 + q;
                    if (cj < a.m) {
                        const double sv = static_cast<double>(vals[q]);
                        if (cj == row) {
                            sd = sv;
                        } else {
                            ss += sv * sv;
                            s0 += sv;
                            s1 += sv * a.cvec[cj];
                            s2 += sv * a.c2vec[cj];
                        }
                    }
                }
            }
            ss = wave_sum(ss); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); sd = wave_sum(sd);
            if (lane == 0) {
                double* r = a.racc + row * 8;
                r[0] += ss; r[1] += s0; r[2] += s1; r[3] += s2; r[4] += sd;
            }
        } else {  // EPI_LAP_SINGLE / EPI_LAP_DUAL
            // rowc: {+-1/rho, S_ii, d+cw, 1/sqrt(d)}, {r, r2, c, c2}; a negative 1/rho marks a row without
            // off-diagonal entries, whose chain result is exactly S_ii x and is dropped instead of cancelled
            const float4 rc0 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC);
            const float4 rc1 = *reinterpret_cast<const float4*>(a.rowc + row * SPARC_ROWC + 4);
            const float4 x0 = *reinterpret_cast<const float4*>(a.X0 + ro);
            const bool isolated = rc0.x < 0.f;
            const float irho = fabsf(rc0.x);
            const bool right = DUAL && lr >= HALF;
            const int KP = DUAL ? W / 2 : W;
            const int cl = (DUAL ? (lr % HALF) : lr) * 4;
            const float* Mb = a.M12 + (right ? 3 * KP : 0) + cl;
            const float* Cf = a.M12 + 6 * KP;  // the 3x3 coupling, row-major
            // diagonal of U C V^T for this row
            const float ldiag = (Cf[0] + Cf[1] * rc1.z + Cf[2] * rc1.w) + rc1.x * (Cf[3] + Cf[4] * rc1.z + Cf[5] * rc1.w) +
                                rc1.y * (Cf[6] + Cf[7] * rc1.z + Cf[8] * rc1.w);
            const float u1 = right ? rc1.z : rc1.x, u2 = right ? rc1.w : rc1.y;
            const float4 ma = *reinterpret_cast<const float4*>(Mb);
            const float4 mb = *reinterpret_cast<const float4*>(Mb + KP);
            const float4 mc = *reinterpret_cast<const float4*>(Mb + 2 * KP);
            float4 t;
            t.x = (isolated ? 0.f : acc.x - rc0.y * x0.x) + ((ma.x + u1 * mb.x + u2 * mc.x) - ldiag * x0.x);
            t.y = (isolated ? 0.f : acc.y - rc0.y * x0.y) + ((ma.y + u1 * mb.y + u2 * mc.y) - ldiag * x0.y);
            t.z = (isolated ? 0.f : acc.z - rc0.y * x0.z) + ((ma.z + u1 * mb.z + u2 * mc.z) - ldiag * x0.z);
            t.w = (isolated ? 0.f : acc.w - rc0.y * x0.w) + ((ma.w + u1 * mb.w + u2 * mc.w) - ldiag * x0.w);
            float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (DUAL) {  // W^T Yn sits HALF lanes to the right of W Yn
                wt.x = __shfl(t.x, lane + HALF);
                wt.y = __shfl(t.y, lane + HALF);
                wt.z = __shfl(t.z, lane + HALF);
                wt.w = __shfl(t.w, lane + HALF);
            }
            double dot = 0.0;
            if (g == 0 && !right) {
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
            wloss += wave_sum(dot);
        }
    }

    if (EPI == EPI_LAP_SINGLE || EPI == EPI_LAP_DUAL) {
        __shared__ double wl[4];
        if (lane == 0) wl[wave] = wloss;
        __syncthreads();
        if (threadIdx.x == 0) a.loss_part[blk] = (wl[0] + wl[1]) + (wl[2] + wl[3]);
    }
}

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
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;  // four independent chains, fixed order
        int b = 0;
        for (; b + 4 <= nparts; b += 4) {
            t0 += colpart[(static_cast<int64_t>(b) * 6 + q) * KP + col];
            t1 += colpart[(static_cast<int64_t>(b + 1) * 6 + q) * KP + col];
            t2 += colpart[(static_cast<int64_t>(b + 2) * 6 + q) * KP + col];
            t3 += colpart[(static_cast<int64_t>(b + 3) * 6 + q) * KP + col];
        }
        for (; b < nparts; ++b) t0 += colpart[(static_cast<int64_t>(b) * 6 + q) * KP + col];
        S[q][col] = (t0 + t1) + (t2 + t3);
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

// Optional timing of one kernel class with HIP events recorded on the launch stream.
struct Profiler {
    int epi = -1;                 // which epilogue class is timed
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;
} g_prof;

template <int LR, bool DUAL, int EPI>
int launch_rows(const SpmmArgs& a, hipStream_t stream) {
    const int64_t nb = (a.m + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (nb <= 0) return SPARC_OK;
    if (nb > 0x7fffffffLL) return SPARC_ERR_ARG;
    const bool timed = g_prof.epi == EPI && g_prof.used + 2 <= g_prof.ev.size();
    if (timed) (void)hipEventRecord(g_prof.ev[g_prof.used], stream);
    hipLaunchKernelGGL((spmm_rows<LR, DUAL, EPI>), dim3(static_cast<unsigned>(nb)), dim3(256), 0, stream, a);
    if (timed) {
        (void)hipEventRecord(g_prof.ev[g_prof.used + 1], stream);
        g_prof.used += 2;
    }
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

template <int EPI>
int launch_single(int width, const SpmmArgs& a, hipStream_t s) {
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
        case EPI_ADD_IDENT: return width == 64 ? launch_rows<16, false, EPI_ADD_IDENT>(a, stream) : SPARC_ERR_ARG;
        case EPI_PREP_ACC: return width == 64 ? launch_rows<16, false, EPI_PREP_ACC>(a, stream) : SPARC_ERR_ARG;
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
        ++n;
    }
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    g_prof.epi = -1;
    if (total_ms) *total_ms = total;
    if (launches) *launches = n;
    return SPARC_OK;
}

extern "C" int sparc_spmm(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                          int width, const float* X, const float* X0, float* Z, sparc_stream_t stream) {
    if (m < 0 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !val)) || !X || !Z) return SPARC_ERR_ARG;
    SpmmArgs a = {};
    a.m = m; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = val;
    a.Xg = X; a.X0 = X0; a.out = Z;
    return sparc_spmm_launch(X0 ? EPI_ADD_X0 : EPI_STORE, width, a, stream);
}

namespace {
constexpr int SCALE_MAX_BLOCKS = 128;
struct LapPlan {
    int kp, W;
    int64_t nscale, rows_per_block, nloss;
    size_t bytes;
};
LapPlan lap_plan(int64_t m, int k) {
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
    p.bytes = 3 * sparc_align_up(static_cast<size_t>(m) * p.W * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(SCALE_MAX_BLOCKS) * 6 * p.kp * sizeof(float)) +
              sparc_align_up((6 * SPARC_MAX_K + 16) * sizeof(float)) +
              sparc_align_up(static_cast<size_t>(p.nloss > 0 ? p.nloss : 1) * sizeof(double));
    return p;
}
}  // namespace

extern "C" size_t sparc_lap_workspace_bytes(int64_t m, int k) {
    if (m < 0 || k < 1 || k > SPARC_MAX_K) return 0;
    return lap_plan(m, k).bytes;
}

extern "C" int sparc_lap_fwd(int64_t m, int64_t nnz, const int64_t* rowptr, const int32_t* colidx, const float* val,
                             const float* valT, const float* rowc, const double* coupling, int k, int64_t ldy,
                             const float* Y, double* loss, float* gradY, void* ws, size_t ws_bytes,
                             sparc_stream_t stream) {
    if (m <= 0 || nnz < 0 || k < 1 || k > SPARC_MAX_K || ldy < k || !rowptr || !rowc || !coupling || !Y || !loss ||
        (nnz > 0 && (!colidx || !val)))
        return SPARC_ERR_ARG;
    const LapPlan p = lap_plan(m, k);
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

    SpmmArgs a = {};
    a.m = m; a.rowptr = rowptr; a.col = colidx; a.val = val; a.valT = valT ? valT : val;
    a.rowc = rowc; a.M12 = M12; a.loss_part = loss_part; a.gradY = gradY; a.ldy = ldy; a.k = k;
    a.inv_m = static_cast<float>(1.0 / static_cast<double>(m));
    a.X0 = X;
    int rcode;
    // Horner: chain(X) = A (X + A (X + A X))
    a.Xg = X; a.out = P1;
    rcode = dual ? launch_dual<EPI_ADD_X0>(p.kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
    if (rcode) return rcode;
    a.Xg = P1; a.out = P2;
    rcode = dual ? launch_dual<EPI_ADD_X0>(p.kp, a, stream) : launch_single<EPI_ADD_X0>(W, a, stream);
    if (rcode) return rcode;
    a.Xg = P2; a.out = nullptr;
    rcode = dual ? launch_dual<EPI_LAP_DUAL>(p.kp, a, stream) : launch_single<EPI_LAP_SINGLE>(W, a, stream);
    if (rcode) return rcode;
    hipLaunchKernelGGL(loss_finish, dim3(1), dim3(256), 0, stream, p.nloss, loss_part, 1.0 / static_cast<double>(m), loss);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
