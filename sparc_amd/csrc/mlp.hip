// Backward helpers of the SpectralNet MLP (src/SpectralTrainer.py:28-39, autograd at :237):
// the activation derivative fused with the bias gradient, and the weight gradient of the
// narrow output layer.  The wide products stay library GEMMs.
#include "common.h"
#include <mutex>

namespace {

constexpr int ACT_RELU = 0;
constexpr int ACT_TANH = 1;
constexpr int COLSUM_LANES = 16;

typedef float v4f __attribute__((ext_vector_type(4)));

// 16 bytes that are read once: loaded past the caches' retention.  Used where whole 1024-column rows stream through
// (the activation derivative: 117 -> 107 us); the narrow-layer kernels, whose workgroups touch 256-byte row segments,
// lose with it (input gradient 49 -> 60 us), so they keep plain loads and stores.
__device__ __forceinline__ float4 stream_load4(const float* p) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <int ACT>
__device__ __forceinline__ float act_grad(float g, float h) {
    if (ACT == ACT_RELU) return h > 0.f ? g : 0.f;
    return g * (1.f - h * h);
}

// out = g * act'(h) and the column sums of out over the block's rows.  A row is covered by
// tpr threads (power of two), VEC columns each; 256 / tpr rows are in flight per iteration.
// partial is [gridDim.x][n].
template <int VEC, int ACT>
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(int64_t m, int n, const float* __restrict__ g, int64_t ldg,
                                                           const float* __restrict__ h, int64_t ldh,
                                                           float* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ partial, int tpr, int64_t rows_per_block) {
    __shared__ float red[256 * VEC];
    const int tx = threadIdx.x & (tpr - 1);
    const int ty = threadIdx.x / tpr;
    const int rpi = 256 / tpr;
    const int col = (blockIdx.y * tpr + tx) * VEC;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(m, r0 + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (col < n) {
#pragma unroll 4
        for (int64_t r = r0 + ty; r < r1; r += rpi) {
            if constexpr (VEC == 4) {
                const float4 gv = stream_load4(g + r * ldg + col);
                const float4 hv = stream_load4(h + r * ldh + col);
                float4 o;
                o.x = act_grad<ACT>(gv.x, hv.x);
                o.y = act_grad<ACT>(gv.y, hv.y);
                o.z = act_grad<ACT>(gv.z, hv.z);
                o.w = act_grad<ACT>(gv.w, hv.w);
                __builtin_nontemporal_store(v4f{o.x, o.y, o.z, o.w}, reinterpret_cast<v4f*>(out + r * ldo + col));
                acc[0] += o.x;
                acc[1 % VEC] += o.y;
                acc[2 % VEC] += o.z;
                acc[3 % VEC] += o.w;
            } else {
                const float o = act_grad<ACT>(g[r * ldg + col], h[r * ldh + col]);
                out[r * ldo + col] = o;
                acc[0] += o;
            }
        }
    }
    if (rpi > 1) {  // fold the rows in flight: fixed order
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = acc[v];
        __syncthreads();
        if (ty == 0) {
            for (int y = 1; y < rpi; ++y)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += red[(y * tpr + tx) * VEC + v];
        }
    }
    if (ty == 0 && col < n) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) partial[(int64_t)blockIdx.x * n + col + v] = acc[v];
    }
}

// db[c] = sum over the row blocks of partial[b][c]; 64 columns x 16 lanes per workgroup.  din > 0: the n columns are
// din + 1 rows of n / (din + 1) sums (row j < din: column j of a weight gradient [n'][din], row din: its bias gradient)
// and are written as the weight gradient followed by the bias gradient.
__global__ __launch_bounds__(64 * COLSUM_LANES) void colsum_finish(int nparts, int n, const float* __restrict__ partial,
                                                                    float* __restrict__ db, int din = 0) {
    __shared__ float red[COLSUM_LANES][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int b = lane;
        for (; b + COLSUM_LANES < nparts; b += 2 * COLSUM_LANES) {
            s0 += partial[(int64_t)b * n + c];
            s1 += partial[(int64_t)(b + COLSUM_LANES) * n + c];
        }
        if (b < nparts) s0 += partial[(int64_t)b * n + c];
    }
    red[lane][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (lane == 0 && c < n) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < COLSUM_LANES; ++l) s += red[l][threadIdx.x & 63];
        if (din > 0) {
            const int nn = n / (din + 1), j = c / nn, o = c % nn;
            db[j < din ? o * din + j : nn * din + o] = s;
        } else {
            db[c] = s;
        }
    }
}

struct ActPlan {
    int vec, tpr, coltiles, nparts;
    int64_t rows_per_block;
};

ActPlan act_plan(int64_t m, int n, bool aligned) {
    ActPlan p;
    p.vec = aligned ? 4 : 1;
    const int per_row = (n + p.vec - 1) / p.vec;
    p.tpr = 1;
    while (p.tpr < per_row && p.tpr < 256) p.tpr <<= 1;
    p.coltiles = (per_row + p.tpr - 1) / p.tpr;
    const int rpi = 256 / p.tpr;
    int64_t parts = 512 / p.coltiles;
    if (parts < 1) parts = 1;
    int64_t rpb = (m + parts - 1) / parts;
    rpb = (rpb + rpi - 1) / rpi * rpi;
    if (rpb < rpi) rpb = rpi;
    p.rows_per_block = rpb;
    p.nparts = (int)((m + rpb - 1) / rpb);
    return p;
}


// partial[chunk] = g[rows of chunk]^T h[rows of chunk][64-column block], f32 MFMA 16x16x4.
// Lane l = (jj = l & 15, kk = l >> 4).  Per step of four rows the lane reads h[row kk][4 jj .. 4 jj + 3]
// (16 bytes) and g[row kk][16 it + jj]; product tile (it, t) therefore holds the OUTPUT columns
// 4 j + t, j = 0..15, of the block: register r of lane l is dW[16 it + 4 kk + r][64 blk + 4 jj + t].
//
// FIRST (the backward of the MLP's first layer, whose input needs no gradient): the wide operand is
// formed on the fly as h * [hmask > 0] (incoming gradient times the ReLU derivative, never written
// out) and the narrow operand gets one more column of ones, so that row n_out of the result is the
// column sum of the masked operand: the bias gradient.
// FIRST == 2: the layer's activations are not read at all (half of the pass's traffic): when the forward
// was linear_narrow_in_fwd_kernel, a pre-activation is b + sum_k x[k] W[c][k] accumulated by fused
// multiply-adds in the order k = 0, 1, ..., so the same sequence here gives the same bits and the same sign.
// A lane owns four columns (their weights stay in registers); the row's x[k] sits in lane k of the lane's
// 16-lane group (the narrow operand's own loads) and is broadcast from there (hmask = W [n_in][n_out], wb = b).
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int K0, int KMAX>
__device__ __forceinline__ void preact_steps(v2f_t (&a)[2], const float xr, const v2f_t (&w)[2][KMAX], const int K) {
    if constexpr (K0 < KMAX) {
        if (K0 < K) {   // uniform
            const float xk = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xr), 0x150 + K0, 0xf, 0xf, false));
            a[0] = __builtin_elementwise_fma(v2f_t{xk, xk}, w[0][K0], a[0]);   // packed: two columns per instruction
            a[1] = __builtin_elementwise_fma(v2f_t{xk, xk}, w[1][K0], a[1]);
            preact_steps<K0 + 1, KMAX>(a, xr, w, K);
        }
    }
}

// KMAX (FIRST == 2): the layer's input columns rounded up to a multiple of four (registers for 4 x KMAX weights)
template <int NT, int FIRST, int KMAX = 4>
__global__ __launch_bounds__(256) void wgrad_small_kernel(int64_t m, int n_out, int n_in, const float* __restrict__ g,
                                                          int64_t ldg, const float* __restrict__ h, int64_t ldh,
                                                          const float* __restrict__ hmask, int64_t ldm,
                                                          const float* __restrict__ wb,
                                                          float* __restrict__ partial, int64_t rows_per_block) {
    extern __shared__ float sm[];  // [4 waves][NT * 16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jj = lane & 15, kk = lane >> 4;
    const int blk = blockIdx.x;
    const int64_t c0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t c1 = min(m, c0 + rows_per_block);
    const int64_t per_wave = ((c1 - c0 + 3) / 4 + 3) / 4 * 4;  // rows of one wave, a multiple of 4
    const int64_t w0 = c0 + wave * per_wave;
    const int64_t w1 = min(c1, w0 + per_wave);

    v4f acc[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][t] = v4f{0.f, 0.f, 0.f, 0.f};

    const float* hp = h + (int64_t)blk * 64 + 4 * jj;
    const float* mp = FIRST == 1 ? hmask + (int64_t)blk * 64 + 4 * jj : nullptr;
    v2f_t lw[2][KMAX], lb[2];   // FIRST == 2: the layer's weights and bias of this lane's four columns, in pairs
    if (FIRST == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t c = (int64_t)blk * 64 + 4 * jj + 2 * t;
            lb[t] = v2f_t{wb[c], wb[c + 1]};
#pragma unroll
            for (int k = 0; k < KMAX; ++k) lw[t][k] = k < n_out ? v2f_t{hmask[c * n_out + k], hmask[(c + 1) * n_out + k]} : v2f_t{0.f, 0.f};
        }
    }
    auto recomputed = [&](const float xr) {   // the pre-activations of this lane's columns in the row whose x is xr
        v2f_t a[2] = {lb[0], lb[1]};
        preact_steps<0, KMAX>(a, xr, lw, n_out);
        return make_float4(a[0][0], a[0][1], a[1][0], a[1][1]);
    };
    int gcol[NT];
    bool gok[NT], gone[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        gok[a] = 16 * a + jj < n_out;
        gone[a] = FIRST && 16 * a + jj == n_out;   // the column of ones
        gcol[a] = gok[a] ? 16 * a + jj : 0;
    }
    auto masked = [](float4 v, const float4 mk) {
        v.x = mk.x > 0.f ? v.x : 0.f;
        v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f;
        v.w = mk.w > 0.f ? v.w : 0.f;
        return v;
    };
    auto step = [&](const float4& hb, const float (&ga)[NT]) {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], hb.x, acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], hb.y, acc[a][1], 0, 0, 0);
            acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], hb.z, acc[a][2], 0, 0, 0);
            acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[a], hb.w, acc[a][3], 0, 0, 0);
        }
    };
    // whole groups of 16 rows: every load of the group is issued before its first product
    int64_t rb = w0;
    for (; rb + 16 <= w1; rb += 16) {
        float4 hb[4], hm[4];
        float ga[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + 4 * u + kk;
            hb[u] = *reinterpret_cast<const float4*>(hp + r * ldh);
            if (FIRST == 1) hm[u] = *reinterpret_cast<const float4*>(mp + r * ldm);
#pragma unroll
            for (int a = 0; a < NT; ++a) ga[u][a] = g[r * ldg + gcol[a]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (FIRST == 2) hm[u] = recomputed(ga[u][0]);   // (from the loaded values: lane k < n_out holds x[k])
#pragma unroll
            for (int a = 0; a < NT; ++a) ga[u][a] = gok[a] ? ga[u][a] : (gone[a] ? 1.f : 0.f);
            if (FIRST) hb[u] = masked(hb[u], hm[u]);
            step(hb[u], ga[u]);
        }
    }
    for (; rb < w1; rb += 4) {  // the chunk's last rows
        const int64_t r = rb + kk;
        const bool live = r < w1;
        const int64_t rc = live ? r : w1 - 1;
        float4 hb = *reinterpret_cast<const float4*>(hp + rc * ldh);
        if (FIRST == 1) hb = masked(hb, *reinterpret_cast<const float4*>(mp + rc * ldm));
        float ga[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) ga[a] = g[rc * ldg + gcol[a]];
        if (FIRST == 2) hb = masked(hb, recomputed(ga[0]));
        if (!live) hb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < NT; ++a) ga[a] = !live ? 0.f : (gok[a] ? ga[a] : (gone[a] ? 1.f : 0.f));
        step(hb, ga);
    }

    // the four waves' tiles meet in LDS and are added in wave order
    float* mine = sm + (size_t)wave * NT * 16 * 64;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 v = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
            *reinterpret_cast<float4*>(mine + (16 * a + 4 * kk + r) * 64 + 4 * jj) = v;
        }
    __syncthreads();
    float* out = partial + ((int64_t)blockIdx.y * NT * 16) * n_in + (int64_t)blk * 64;
    for (int e = threadIdx.x; e < NT * 16 * 64; e += 256) {
        const float s = ((sm[e] + sm[NT * 16 * 64 + e]) + sm[2 * NT * 16 * 64 + e]) + sm[3 * NT * 16 * 64 + e];
        out[(int64_t)(e >> 6) * n_in + (e & 63)] = s;
    }
}

// dW[i][j] = sum over the chunks of partial[chunk][i][j].  t_rows > 0 (the first layer's backward, n_out = t_rows + 1):
// rows 0 .. t_rows-1 are stored transposed, dW[j][i] as [n_in][t_rows] (the layout of the layer's weight), and row
// t_rows (the bias gradient) behind them.
__global__ __launch_bounds__(256) void wgrad_finish(int nchunks, int n_out, int n_out_pad, int n_in,
                                                    const float* __restrict__ partial, float* __restrict__ dW, int t_rows) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n_out * n_in) return;
    const int i = (int)(e / n_in), j = (int)(e % n_in);
    const int64_t stride = (int64_t)n_out_pad * n_in;
    const float* p = partial + (int64_t)i * n_in + j;
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // eight loads in flight, fixed order
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += p[(c + u) * stride];
    }
    for (; c < nchunks; ++c) t[0] += p[c * stride];
    const float sum = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    if (t_rows > 0) dW[i < t_rows ? (int64_t)j * t_rows + i : (int64_t)n_in * t_rows + j] = sum;
    else dW[e] = sum;
}

// out = act(x W^T + b) for a narrow output (n_out <= 64): the MLP's last layer.  W (n_out x K, zero
// padded to NT * 16 rows) sits in LDS with a 4-float row pad; a wave produces 16 rows at a time.
// Lane l = (i = l & 15, kk = l >> 4) reads a run of consecutive k of x's row i and the same k of W's
// row 16 a + i from LDS; every MFMA step uses one k of the run on both operands (the order of k inside
// the sum is free).  Register q of the result is out[row 4 kk + q][16 a + i].
//
// GRAM: the kernel also leaves, per workgroup, the float64 partial Gram matrix of the rows below gram_rows of its
// output (the orthonormalisation of a training step needs out[:gram_rows]^T out[:gram_rows] next, and the output
// tile of a wave sits in its registers in exactly the layout v_mfma_f64_16x16x4_f64 takes for both operands: step q
// multiplies the four rows 4 kk + q).  The host launches this form with at most one tile per wave, so that the k x k
// accumulators (8 NT^2 registers) never live beside the product loop's; the waves' partial matrices are added in four
// LDS images (over the weights, which nobody reads any more) in wave order and the workgroup writes their sum to
// slab[blockIdx.x][n_out][n_out].  Workgroups whose rows all lie at or above gram_rows write nothing.
template <int NT, int ACT, bool GRAM = false>
__global__ __launch_bounds__(GRAM && NT == 4 ? 512 : 1024)
void linear_narrow_fwd_kernel(int64_t m, int K, int n_out, const float* __restrict__ x, int64_t ldx,
                              const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ out,
                              int64_t ldo, int64_t gram_rows = 0, double* __restrict__ slab = nullptr) {
    extern __shared__ float Ws[];  // [NT * 16][K + 4]
    const int KS = K + 4;
    const int k4 = K >> 2;
    const int nw = blockDim.x >> 6;   // waves of the workgroup: 8, 12 or 16, whichever spreads the row tiles best
    for (int e = threadIdx.x; e < NT * 16 * k4; e += blockDim.x) {
        const int n = e / k4, c4 = e - n * k4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < n_out) v = *reinterpret_cast<const float4*>(W + (int64_t)n * K + 4 * c4);
        *reinterpret_cast<float4*>(Ws + n * KS + 4 * c4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int64_t ntiles = (m + 15) / 16;
    float bias[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) bias[a] = (16 * a + i < n_out) ? b[16 * a + i] : 0.f;
    typedef double gram_acc_t __attribute__((ext_vector_type(4)));
    float kept[NT][4];   // GRAM: the wave's output tile as it was stored
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) kept[a][q] = 0.f;
    // (GRAM: one pass at most, the host sizes the grid for it)
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < ntiles; tile += GRAM ? ntiles : (int64_t)gridDim.x * nw) {
        const int64_t r = min(tile * 16 + i, m - 1);
        // 64 consecutive k per group: lane (i, kk) reads x[row i][64 T + 16 kk .. + 15] (64 contiguous
        // bytes, a whole 256-byte line per row and wave instruction) and the same k of W's rows
        const float* xp = x + r * ldx + 16 * kk;
        const float* wp = Ws + i * KS + 16 * kk;
        v4f acc[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) acc[a] = v4f{0.f, 0.f, 0.f, 0.f};
        const int groups = K >> 6;
        v4f bufA[4], bufB[4];
        auto request = [&](v4f (&dst)[4], const int T) {   // T clamped: the last request is a harmless repeat
            const float* p = xp + 64 * min(T, groups - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[u] = *reinterpret_cast<const v4f*>(p + 4 * u);
            __builtin_amdgcn_sched_barrier(0);   // the requests stay ahead of the products that follow
        };
        auto products = [&](const v4f (&xa)[4], const int T) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 w[NT];
#pragma unroll
                for (int a = 0; a < NT; ++a) w[a] = *reinterpret_cast<const float4*>(wp + 16 * a * KS + 64 * T + 4 * u);
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][0], w[a].x, acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][1], w[a].y, acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][2], w[a].z, acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][3], w[a].w, acc[a], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // Two register buffers: a group's run arrives while the other group's products run.  The loop body
        // has no branch between a request and its use (the optimiser sinks a request into a conditional use).
        if (groups > 0) request(bufA, 0);
        int T = 0;
        for (; T + 2 <= groups; T += 2) {
            request(bufB, T + 1);
            products(bufA, T);
            request(bufA, T + 2);
            products(bufB, T + 1);
        }
        if (T < groups) products(bufA, T);
        for (int t = groups * 4 + 0; t < (K >> 4); ++t) {   // K not a multiple of 64: 16 k at a time, 4 per lane
            const float4 xr = *reinterpret_cast<const float4*>(x + r * ldx + 16 * t + 4 * kk);
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const float4 w = *reinterpret_cast<const float4*>(Ws + (16 * a + i) * KS + 16 * t + 4 * kk);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr.x, w.x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr.y, w.y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr.z, w.z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr.w, w.w, acc[a], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int c = 16 * a + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = tile * 16 + 4 * kk + q;
                if (row < m && c < n_out) {
                    float v = acc[a][q] + bias[a];
                    if (ACT == ACT_TANH) v = tanhf(v);
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
                    out[row * ldo + c] = v;
                    if (GRAM && row < gram_rows) kept[a][q] = v;
                }
            }
        }
    }
    if constexpr (GRAM) {
        if ((int64_t)blockIdx.x * nw * 16 >= gram_rows) return;   // (the whole workgroup)
        gram_acc_t g[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) g[a][c] = gram_acc_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double p[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) p[a] = static_cast<double>(kept[a][q]);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int c = 0; c < NT; ++c) g[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(p[a], p[c], g[a][c], 0, 0, 0);
        }
        constexpr int KP = NT * 16;
        double* img = reinterpret_cast<double*>(Ws);   // four images [KP][KP]
        __syncthreads();                               // every wave is done with the weights
        for (int round = 0; round * 4 < nw; ++round) {
            if ((wave >> 2) == round) {
                double* mine = img + (wave & 3) * KP * KP;
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int c = 0; c < NT; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int e = (a * 16 + kk + 4 * q) * KP + c * 16 + i;
                            if (round == 0) mine[e] = g[a][c][q]; else mine[e] += g[a][c][q];
                        }
            }
            __syncthreads();
        }
        double* dst = slab + (int64_t)blockIdx.x * n_out * n_out;
        for (int e = threadIdx.x; e < n_out * n_out; e += blockDim.x) {
            const int o = (e / n_out) * KP + e % n_out;
            dst[e] = ((img[o] + img[KP * KP + o]) + img[2 * KP * KP + o]) + img[3 * KP * KP + o];
        }
    }
}

// out = act(x W^T + b) for a narrow INPUT (K <= 16 columns) and a wide output: the MLP's first layer
// (src/SpectralTrainer.py:28-39 on 10 node features).  The result is all the traffic (m x N floats against m x K
// read), so the kernel is a store stream: a thread owns four consecutive output columns and keeps their 4 x K
// weights in registers; the workgroup's NIN_ROWS rows of x wait in LDS (zero padded to KMAX columns), where every
// lane reads the same address.  W is [N][K] row-major (torch's Linear.weight), K <= KMAX, N a multiple of 4.
constexpr int NIN_ROWS = 32;

template <int KMAX, int ACT>
__global__ __launch_bounds__(256) void linear_narrow_in_fwd_kernel(int64_t m, int K, int N, const float* __restrict__ x,
                                                                   int64_t ldx, const float* __restrict__ W,
                                                                   const float* __restrict__ b, float* __restrict__ out,
                                                                   int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float xs[NIN_ROWS * KMAX];
    const int64_t r0 = (int64_t)blockIdx.x * NIN_ROWS;
    const int rows = (int)min((int64_t)NIN_ROWS, m - r0);
    for (int e = threadIdx.x; e < NIN_ROWS * KMAX; e += 256) {
        const int r = e / KMAX, k = e - r * KMAX;
        xs[e] = (r < rows && k < K) ? x[(r0 + r) * ldx + k] : 0.f;
    }
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    const bool live = c < N;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f w[2][KMAX], bias[2];    // column pairs (c, c + 1) and (c + 2, c + 3): packed fp32 multiply-adds
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bias[t] = live ? v2f{b[c + 2 * t], b[c + 2 * t + 1]} : v2f{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            w[t][k] = (live && k < K) ? v2f{W[(int64_t)(c + 2 * t) * K + k], W[(int64_t)(c + 2 * t + 1) * K + k]} : v2f{0.f, 0.f};
    }
    __syncthreads();
    if (!live) return;
    float* op = out + r0 * ldo + c;
#pragma unroll 2
    for (int r = 0; r < rows; ++r) {
        v2f a[2] = {bias[0], bias[1]};
#pragma unroll
        for (int q = 0; q < KMAX / 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xs + r * KMAX + 4 * q);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = __builtin_elementwise_fma(v2f{xv.x, xv.x}, w[t][4 * q], a[t]);
                a[t] = __builtin_elementwise_fma(v2f{xv.y, xv.y}, w[t][4 * q + 1], a[t]);
                a[t] = __builtin_elementwise_fma(v2f{xv.z, xv.z}, w[t][4 * q + 2], a[t]);
                a[t] = __builtin_elementwise_fma(v2f{xv.w, xv.w}, w[t][4 * q + 3], a[t]);
            }
        }
        float o[4] = {a[0][0], a[0][1], a[1][0], a[1][1]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ACT == ACT_TANH) o[t] = tanhf(o[t]);
            if (ACT == ACT_RELU) o[t] = fmaxf(o[t], 0.f);
        }
        *reinterpret_cast<float4*>(op + r * ldo) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx = (g W) * [hprev > 0] and the column sums of dx, for a narrow g (n_out <= 64 columns): the input
// gradient of the MLP's last layer fused with the ReLU derivative and the bias gradient of the layer
// before it.  Workgroup = (64-column block of dx, chunk of rows); the block's slice of W lives in
// registers for the whole chunk.  Lane l = (j = l & 15, kk = l >> 4); product tile t holds the OUTPUT
// columns 4 j + t of the block, so register q of the four tiles is dx[row 4 kk + q][4 j .. 4 j + 3]: one
// 16-byte load of hprev and one 16-byte store per row.
template <int NT>
__global__ __launch_bounds__(256) void dgrad_narrow_relu_kernel(int64_t m, int n_out, int N, const float* __restrict__ g,
                                                                int64_t ldg, const float* __restrict__ W,
                                                                const float* __restrict__ hprev, int64_t ldh,
                                                                float* __restrict__ dx, int64_t ldx,
                                                                float* __restrict__ partial, int64_t rows_per_block) {
    __shared__ float red[4][16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int c0 = blockIdx.x * 64;
    // B operand: W[k][c0 + 4 j + t] for k = 4 s + kk, s = 0 .. NT * 4 - 1
    float wb[NT * 4][4];
#pragma unroll
    for (int s_ = 0; s_ < NT * 4; ++s_) {
        const int k = 4 * s_ + kk;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n_out) v = *reinterpret_cast<const float4*>(W + (int64_t)k * N + c0 + 4 * j);
        wb[s_][0] = v.x; wb[s_][1] = v.y; wb[s_][2] = v.z; wb[s_][3] = v.w;
    }
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t b1 = min(m, b0 + rows_per_block);
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r0 = b0 + 16 * wave; r0 < b1; r0 += 64) {
        const int64_t ra = min(r0 + j, m - 1);   // A operand row of this lane
        float ga[NT * 4];
#pragma unroll
        for (int s_ = 0; s_ < NT * 4; ++s_) {
            const int k = 4 * s_ + kk;
            ga[s_] = g[ra * ldg + min(k, n_out - 1)];
        }
        float4 hm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = min(r0 + 4 * kk + q, m - 1);
            hm[q] = *reinterpret_cast<const float4*>(hprev + row * ldh + c0 + 4 * j);
        }
        v4f acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < NT * 4; ++s_) {
            const float a = (4 * s_ + kk < n_out) ? ga[s_] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[s_][t], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = r0 + 4 * kk + q;
            if (row < b1) {
                float4 o;
                o.x = hm[q].x > 0.f ? acc[0][q] : 0.f;
                o.y = hm[q].y > 0.f ? acc[1][q] : 0.f;
                o.z = hm[q].z > 0.f ? acc[2][q] : 0.f;
                o.w = hm[q].w > 0.f ? acc[3][q] : 0.f;
                *reinterpret_cast<float4*>(dx + row * ldx + c0 + 4 * j) = o;
                csum[0] += o.x; csum[1] += o.y; csum[2] += o.z; csum[3] += o.w;
            }
        }
    }
    // column sums: the four row groups of a wave, then the four waves, in fixed order
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        csum[t] += __shfl_xor(csum[t], 16);
        csum[t] += __shfl_xor(csum[t], 32);
    }
    if (kk == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave][j][t] = csum[t];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int jj = threadIdx.x >> 2, t = threadIdx.x & 3;
        const float s = ((red[0][jj][t] + red[1][jj][t]) + red[2][jj][t]) + red[3][jj][t];
        partial[(int64_t)blockIdx.y * N + c0 + 4 * jj + t] = s;
    }
}

constexpr int WGRAD_CHUNKS = 128;     // most row chunks of one product
constexpr int WGRAD_CHUNK_ROWS = 256; // fewest rows of a chunk

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t sparc_act_bwd_bias_workspace_bytes(int64_t m, int n) {
    if (m < 0 || n <= 0) return 0;
    return sparc_align_up((size_t)512 * (size_t)n * sizeof(float)) + 256;
}

extern "C" int sparc_act_bwd_bias(int64_t m, int n, int act, const float* g, int64_t ldg, const float* h, int64_t ldh,
                                  float* gout, int64_t ldo, float* db, void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m < 0 || n <= 0 || (act != ACT_RELU && act != ACT_TANH) || !g || !h || !gout || !db || ldg < n || ldh < n || ldo < n)
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_act_bwd_bias_workspace_bytes(m, n) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0) return hipMemsetAsync(db, 0, (size_t)n * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    const bool aligned = n % 4 == 0 && ldg % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0 && aligned16(g) && aligned16(h) && aligned16(gout);
    const ActPlan p = act_plan(m, n, aligned);
    float* partial = static_cast<float*>(ws);
    dim3 grid(p.nparts, p.coltiles);
#define SPARC_ACT_LAUNCH(V, A)                                                                                         \
    hipLaunchKernelGGL((act_bwd_bias_kernel<V, A>), grid, dim3(256), 0, stream, m, n, g, ldg, h, ldh, gout, ldo, partial, \
                       p.tpr, p.rows_per_block)
    if (p.vec == 4) {
        if (act == ACT_RELU) SPARC_ACT_LAUNCH(4, ACT_RELU); else SPARC_ACT_LAUNCH(4, ACT_TANH);
    } else {
        if (act == ACT_RELU) SPARC_ACT_LAUNCH(1, ACT_RELU); else SPARC_ACT_LAUNCH(1, ACT_TANH);
    }
#undef SPARC_ACT_LAUNCH
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finish, dim3((n + 63) / 64), dim3(64 * COLSUM_LANES), 0, stream, p.nparts, n, partial, db, 0);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" size_t sparc_wgrad_small_workspace_bytes(int n_out, int n_in) {
    if (n_out <= 0 || n_out > SPARC_MAX_K || n_in <= 0) return 0;
    const int pad = (n_out + 15) / 16 * 16;
    return sparc_align_up((size_t)WGRAD_CHUNKS * pad * (size_t)n_in * sizeof(float)) + 256;
}

namespace {
int wgrad_launch(int first, int64_t m, int n_out, int n_rows_out, int n_in, const float* g, int64_t ldg, const float* h,
                 int64_t ldh, const float* hmask, int64_t ldm, const float* wb, float* dW, void* ws, sparc_stream_t stream) {
    const int nt = (n_rows_out + 15) / 16;
    int64_t rpb = (m + WGRAD_CHUNKS - 1) / WGRAD_CHUNKS;
    if (rpb < WGRAD_CHUNK_ROWS) rpb = WGRAD_CHUNK_ROWS;
    rpb = (rpb + 63) / 64 * 64;
    const int nchunks = (int)((m + rpb - 1) / rpb);
    float* partial = static_cast<float*>(ws);
    dim3 grid(n_in / 64, nchunks);
    const size_t lds = (size_t)4 * nt * 16 * 64 * sizeof(float);
#define SPARC_WGRAD(NT_, F_)                                                                                              \
    hipLaunchKernelGGL((wgrad_small_kernel<NT_, F_>), grid, dim3(256), lds, stream, m, n_out, n_in, g, ldg, h, ldh, hmask, \
                       ldm, wb, partial, rpb)
    if (first == 2) {   // at most 16 input columns: one tile with the column of ones, two for 16 columns
#define SPARC_WGRAD2(NT_, KM_)                                                                                             \
    hipLaunchKernelGGL((wgrad_small_kernel<NT_, 2, KM_>), grid, dim3(256), lds, stream, m, n_out, n_in, g, ldg, h, ldh, hmask, \
                       ldm, wb, partial, rpb)
        if (n_out <= 4) SPARC_WGRAD2(1, 4);
        else if (n_out <= 8) SPARC_WGRAD2(1, 8);
        else if (n_out <= 12) SPARC_WGRAD2(1, 12);
        else if (nt == 1) SPARC_WGRAD2(1, 16);
        else SPARC_WGRAD2(2, 16);
#undef SPARC_WGRAD2
    } else if (first) {
        switch (nt) {
            case 1: SPARC_WGRAD(1, 1); break;
            case 2: SPARC_WGRAD(2, 1); break;
            case 3: SPARC_WGRAD(3, 1); break;
            default: SPARC_WGRAD(4, 1); break;
        }
    } else {
        switch (nt) {
            case 1: SPARC_WGRAD(1, 0); break;
            case 2: SPARC_WGRAD(2, 0); break;
            case 3: SPARC_WGRAD(3, 0); break;
            default: SPARC_WGRAD(4, 0); break;
        }
    }
#undef SPARC_WGRAD
    SPARC_CHECK_LAUNCH();
    const int64_t total = (int64_t)n_rows_out * n_in;
    hipLaunchKernelGGL(wgrad_finish, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, nchunks, n_rows_out, nt * 16,
                       n_in, partial, dW, first ? n_out : 0);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_wgrad_small(int64_t m, int n_out, int n_in, const float* g, int64_t ldg, const float* h, int64_t ldh,
                                 float* dW, void* ws, size_t ws_bytes, sparc_stream_t stream) {
    if (m < 0 || n_out <= 0 || n_out > SPARC_MAX_K || n_in <= 0 || n_in % 64 != 0 || !g || !h || !dW || ldg < n_out ||
        ldh < n_in || ldh % 4 != 0 || !aligned16(h))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_wgrad_small_workspace_bytes(n_out, n_in) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0) return hipMemsetAsync(dW, 0, (size_t)n_out * n_in * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    return wgrad_launch(0, m, n_out, n_out, n_in, g, ldg, h, ldh, nullptr, 0, nullptr, dW, ws, stream);
}

extern "C" size_t sparc_first_layer_bwd_workspace_bytes(int din, int n) {
    return sparc_wgrad_small_workspace_bytes(din + 1, n);
}

extern "C" int sparc_first_layer_bwd(int64_t m, int din, int n, const float* x, int64_t ldx, const float* g, int64_t ldg,
                                     const float* h, int64_t ldh, float* dW_db, void* ws, size_t ws_bytes,
                                     sparc_stream_t stream) {
    if (m < 0 || din <= 0 || din + 1 > SPARC_MAX_K || n <= 0 || n % 64 != 0 || !x || !g || !h || !dW_db || ldx < din ||
        ldg < n || ldh < n || ldg % 4 != 0 || ldh % 4 != 0 || !aligned16(g) || !aligned16(h))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_first_layer_bwd_workspace_bytes(din, n) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0)
        return hipMemsetAsync(dW_db, 0, (size_t)(din + 1) * n * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    return wgrad_launch(1, m, din, din + 1, n, x, ldx, g, ldg, h, ldh, nullptr, dW_db, ws, stream);
}

extern "C" int sparc_first_layer_bwd_recompute(int64_t m, int din, int n, const float* x, int64_t ldx, const float* g, int64_t ldg,
                                               const float* W, const float* b, float* dW_db, void* ws, size_t ws_bytes,
                                               sparc_stream_t stream) {
    if (m < 0 || din <= 0 || din > 16 || n <= 0 || n % 64 != 0 || !x || !g || !W || !b || !dW_db || ldx < din || ldg < n ||
        ldg % 4 != 0 || !aligned16(g))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_first_layer_bwd_workspace_bytes(din, n) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0)
        return hipMemsetAsync(dW_db, 0, (size_t)(din + 1) * n * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    return wgrad_launch(2, m, din, din + 1, n, x, ldx, g, ldg, W, 0, b, dW_db, ws, stream);
}

namespace {
// Per device and per instantiation: allow the large dynamic LDS allocation (the weight matrix of the layer).
template <int NT, int ACT>
int configure_narrow_fwd() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SPARC_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_narrow_fwd_kernel<NT, ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        done[dev] = true;
    }
    return SPARC_OK;
}
}  // namespace

extern "C" int sparc_linear_narrow_fwd(int64_t m, int K, int n_out, int act, const float* x, int64_t ldx, const float* W,
                                       const float* b, float* out, int64_t ldo, sparc_stream_t stream) {
    if (m < 0 || K <= 0 || K % 16 != 0 || n_out <= 0 || n_out > SPARC_MAX_K || (act != ACT_RELU && act != ACT_TANH && act != 2) ||
        !x || !W || !b || !out || ldx < K || ldx % 4 != 0 || ldo < n_out || !aligned16(x) || !aligned16(W))
        return SPARC_ERR_ARG;
    const int nt = (n_out + 15) / 16;
    const size_t lds = (size_t)nt * 16 * (K + 4) * sizeof(float);
    if (lds > 150 * 1024) return SPARC_ERR_ARG;   // the weight matrix must fit the LDS of one CU
    if (m == 0) return SPARC_OK;
    const int64_t ntiles = (m + 15) / 16;
    // one workgroup per CU (the weights fill its LDS); the matrix pipe of a SIMD serves waves / 4 of its waves, each
    // with ceil(tiles / (256 * waves)) row tiles: take the wave count with the fewest tiles per SIMD
    int waves = 8;
    int64_t best = 0;
    for (int w = 8; w <= 16; w += 4) {
        const int64_t per_simd = ((ntiles + 256 * w - 1) / (256 * w)) * (w / 4);
        if (w == 8 || per_simd < best) { best = per_simd; waves = w; }
    }
    int64_t nb = (ntiles + waves - 1) / waves;
    if (nb > 256) nb = 256;
    dim3 grid((unsigned)nb), block(64 * waves);
#define SPARC_FWD(NT_, A_)                                                                                               \
    do {                                                                                                                 \
        if (configure_narrow_fwd<NT_, A_>() != SPARC_OK) return SPARC_ERR_LAUNCH;                                        \
        hipLaunchKernelGGL((linear_narrow_fwd_kernel<NT_, A_>), grid, block, lds, stream, m, K, n_out, x, ldx, W, b, out, ldo); \
    } while (0)
#define SPARC_FWD_ACT(NT_)                                    \
    do {                                                      \
        if (act == ACT_RELU) SPARC_FWD(NT_, ACT_RELU);        \
        else if (act == ACT_TANH) SPARC_FWD(NT_, ACT_TANH);   \
        else SPARC_FWD(NT_, 2);                               \
    } while (0)
    switch (nt) {
        case 1: SPARC_FWD_ACT(1); break;
        case 2: SPARC_FWD_ACT(2); break;
        case 3: SPARC_FWD_ACT(3); break;
        default: SPARC_FWD_ACT(4); break;
    }
#undef SPARC_FWD_ACT
#undef SPARC_FWD
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

namespace {
constexpr int FWD_GRAM_MAX_PARTS = 256;   // one workgroup per CU

template <int NT, int ACT>
int configure_narrow_fwd_gram() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SPARC_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_narrow_fwd_kernel<NT, ACT, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return SPARC_ERR_LAUNCH;
        done[dev] = true;
    }
    return SPARC_OK;
}
}  // namespace

extern "C" size_t sparc_linear_narrow_fwd_gram_workspace_bytes(int n_out) {
    if (n_out < 1 || n_out > SPARC_MAX_K) return 0;
    return sparc_align_up((size_t)FWD_GRAM_MAX_PARTS * n_out * n_out * sizeof(double));
}

extern "C" int sparc_linear_narrow_fwd_gram(int64_t m, int K, int n_out, int act, const float* x, int64_t ldx, const float* W,
                                            const float* b, float* out, int64_t ldo, int64_t gram_rows, double* G, void* ws,
                                            size_t ws_bytes, sparc_stream_t stream) {
    if (m <= 0 || K <= 0 || K % 16 != 0 || n_out <= 0 || n_out > SPARC_MAX_K || (act != ACT_RELU && act != ACT_TANH && act != 2) ||
        !x || !W || !b || !out || !G || ldx < K || ldx % 4 != 0 || ldo < n_out || !aligned16(x) || !aligned16(W) ||
        gram_rows < 1 || gram_rows > m)
        return SPARC_ERR_ARG;
    if (!ws || ws_bytes < sparc_linear_narrow_fwd_gram_workspace_bytes(n_out) || (reinterpret_cast<uintptr_t>(ws) & 255))
        return SPARC_ERR_WS;
    const int nt = (n_out + 15) / 16;
    const size_t lds_w = (size_t)nt * 16 * (K + 4) * sizeof(float);
    const size_t lds_g = (size_t)4 * nt * 16 * nt * 16 * sizeof(double);
    const size_t lds = lds_w > lds_g ? lds_w : lds_g;
    if (lds > 150 * 1024) return SPARC_ERR_ARG;
    // every wave takes one tile of 16 rows: the fewest waves per workgroup (8 at most for 64 output columns: their
    // accumulators need the registers of two waves per SIMD) that cover the rows with one workgroup per CU
    const int64_t ntiles = (m + 15) / 16;
    const int max_waves = nt == 4 ? 8 : 16;
    int waves = 8;
    while (waves < max_waves && (ntiles + waves - 1) / waves > FWD_GRAM_MAX_PARTS) waves += 4;
    const int64_t nb = (ntiles + waves - 1) / waves;
    if (nb > FWD_GRAM_MAX_PARTS) return SPARC_ERR_SIZE;
    const int nparts = (int)((gram_rows + 16 * waves - 1) / (16 * waves));
    double* slab = static_cast<double*>(ws);
    dim3 grid((unsigned)nb), block(64 * waves);
#define SPARC_FWDG(NT_, A_)                                                                                              \
    do {                                                                                                                 \
        if (configure_narrow_fwd_gram<NT_, A_>() != SPARC_OK) return SPARC_ERR_LAUNCH;                                   \
        hipLaunchKernelGGL((linear_narrow_fwd_kernel<NT_, A_, true>), grid, block, lds, stream, m, K, n_out, x, ldx, W, b, out, \
                           ldo, gram_rows, slab);                                                                        \
    } while (0)
#define SPARC_FWDG_ACT(NT_)                                    \
    do {                                                       \
        if (act == ACT_RELU) SPARC_FWDG(NT_, ACT_RELU);        \
        else if (act == ACT_TANH) SPARC_FWDG(NT_, ACT_TANH);   \
        else SPARC_FWDG(NT_, 2);                               \
    } while (0)
    switch (nt) {
        case 1: SPARC_FWDG_ACT(1); break;
        case 2: SPARC_FWDG_ACT(2); break;
        case 3: SPARC_FWDG_ACT(3); break;
        default: SPARC_FWDG_ACT(4); break;
    }
#undef SPARC_FWDG_ACT
#undef SPARC_FWDG
    SPARC_CHECK_LAUNCH();
    return sparc_gram_reduce(n_out, nparts, slab, G, stream);
}

extern "C" int sparc_linear_narrow_in_fwd(int64_t m, int K, int N, int act, const float* x, int64_t ldx, const float* W,
                                          const float* b, float* out, int64_t ldo, sparc_stream_t stream) {
    if (m < 0 || K <= 0 || K > 16 || N <= 0 || N % 4 != 0 || (act != ACT_RELU && act != ACT_TANH && act != 2) || !x || !W || !b ||
        !out || ldx < K || ldo < N || ldo % 4 != 0 || !aligned16(out))
        return SPARC_ERR_ARG;
    if (m == 0) return SPARC_OK;
    const int ny = (N + 1023) / 1024;
    dim3 grid((unsigned)((m + NIN_ROWS - 1) / NIN_ROWS), (unsigned)ny);
#define SPARC_NIN(KM_, A_)                                                                                            \
    hipLaunchKernelGGL((linear_narrow_in_fwd_kernel<KM_, A_>), grid, dim3(256), 0, stream, m, K, N, x, ldx, W, b, out, ldo)
#define SPARC_NIN_ACT(KM_)                                  \
    do {                                                    \
        if (act == ACT_RELU) SPARC_NIN(KM_, ACT_RELU);      \
        else if (act == ACT_TANH) SPARC_NIN(KM_, ACT_TANH); \
        else SPARC_NIN(KM_, 2);                             \
    } while (0)
    if (K <= 4) SPARC_NIN_ACT(4);
    else if (K <= 8) SPARC_NIN_ACT(8);
    else if (K <= 12) SPARC_NIN_ACT(12);
    else SPARC_NIN_ACT(16);
#undef SPARC_NIN_ACT
#undef SPARC_NIN
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

namespace {
constexpr int DGRAD_CHUNKS = 256;
int64_t dgrad_rows_per_block(int64_t m) {
    int64_t rpb = (m + DGRAD_CHUNKS - 1) / DGRAD_CHUNKS;
    rpb = (rpb + 63) / 64 * 64;
    return rpb < 64 ? 64 : rpb;
}
}  // namespace

// dx = (g W) * [hprev > 0] and the column sums of dx for a WIDE g (K columns, a multiple of 16): the input gradient of a
// hidden Linear layer (g [m][K], W [K][N], N a multiple of 128) fused with the ReLU derivative and the bias gradient of
// the layer before it, whose output hprev [m][N] is this layer's input.  As two steps (library GEMM, then
// act_bwd_bias) the m x N product is written, read back with hprev and written again; here it leaves the registers
// once, masked.  float32 in, float32 sums on v_mfma_f32_32x32x2_f32.
//
// Workgroup = 256 x 128 tile of dx, four wavefronts in a 2 x 2 arrangement of 128 x 64 (4 x 2 tiles of 32 x 32: 128
// accumulator registers), two workgroups per CU.  K is walked in chunks of 16.  Lane l = (i = l & 31, h = l >> 5) reads
// k = 8 c + 4 h + t of g's row i for c = 0, 1 with one 16-byte LDS read each (t = register of the read) and the same k
// of W's columns with 4-byte reads: matrix instruction (c, t) contracts the two k of its two lane halves (the order of
// k inside the sum is free).  Result register r of a 32 x 32 tile: row 8 (r >> 2) + 4 h + (r & 3), column i.
// Tiles are dealt so that the eight column tiles of a row tile run on one XCD one after the other: g's rows are read
// from memory once and served to the other seven from that XCD's L2.
//
// The chunks travel from global memory straight into LDS (global_load_lds, 16 bytes per lane, no register on the
// way) two chunks ahead of the products: three LDS buffers of 24 KB, one barrier per chunk.  A copy instruction fills
// 1 KB of LDS in lane order, so g's 64-byte rows cannot be padded; instead the four 16-byte pieces of row r are
// stored in the order piece ^ ((r >> 2) & 3), which puts the 16-byte reads of every 16-lane group on distinct banks.
// (With the chunk staged through registers and padded rows the same kernel takes 4 % longer: the wave then waits for
// its loads behind every chunk's products.)
//
// DP > 0 (the layer before is the network's first, on din <= DP inputs xin that need no gradient, weights w0 [N][din] and
// bias b0, its forward computed by linear_narrow_in_fwd_kernel): neither is hprev read nor dx stored;
// the weight and bias gradient of that first layer, dW[n][j] = sum over rows of dx[row][n] xin[row][j] and db[n] =
// sum over rows of dx[row][n], leave the registers as din + 1 column sums per tile instead (a lane owns a column, its
// registers are rows: the sums run down the registers): partial[row tile][din + 1][N].
constexpr int GW_BM = 256, GW_BN = 128, GW_BK = 16;
typedef float v16f __attribute__((ext_vector_type(16)));

template <int DP>
__global__ __launch_bounds__(256, 2) void dgrad_wide_relu_kernel(int64_t m, int K, int N, const float* __restrict__ g,
                                                                     int64_t ldg, const float* __restrict__ W,
                                                                     const float* __restrict__ hprev, int64_t ldh,
                                                                     float* __restrict__ dx, int64_t ldx,
                                                                     float* __restrict__ partial,
                                                                     const float* __restrict__ xin, int64_t ldxi, int din,
                                                                     const float* __restrict__ w0, const float* __restrict__ b0) {
    constexpr int A_FLOATS = GW_BM * GW_BK, B_FLOATS = GW_BK * GW_BN, BUF = A_FLOATS + B_FLOATS;
    __shared__ __attribute__((aligned(1024))) float lds[3 * BUF];
    const int ntn = N / GW_BN;
    const int64_t tile = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int64_t m0 = (tile / ntn) * GW_BM;
    const int n0 = (int)(tile % ntn) * GW_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int wr = (wave >> 1) * 128, wc = (wave & 1) * 64;

    // this lane's source of each of the wave's copy instructions (four for g: 16 rows each; two for W: two k each)
    const float* ga[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = 16 * (4 * wave + u) + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        ga[u] = g + min(m0 + row, m - 1) * ldg + 4 * piece;
    }
    const float* gb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) gb[u] = W + (int64_t)(2 * (2 * wave + u) + (lane >> 5)) * N + n0 + 4 * (lane & 31);
    // The copies are written as assembly: the compiler would otherwise make every LDS read wait for ALL copies in
    // flight (it cannot tell the buffer being read from the one being filled), which is the wait this kernel exists to
    // avoid.  The waits are placed by hand below: the loop contains no other vector memory operation.
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds;
    auto copy_chunk = [&](const int c) {
        const unsigned dst = lds_base + (unsigned)((c % 3) * BUF * 4);
        const int k0 = c * GW_BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* src = ga[u] + k0;
            const unsigned at = dst + (unsigned)((4 * wave + u) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(at), "v"(src) : "memory", "m0");
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* src = gb[u] + (int64_t)k0 * N;
            const unsigned at = dst + (unsigned)(A_FLOATS * 4 + (2 * wave + u) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(at), "v"(src) : "memory", "m0");
        }
    };

    v16f acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;

    const int nchunks = K / GW_BK;
    const int sw = (i >> 2) & 3;   // the piece order of this lane's rows
    copy_chunk(0);
    if (nchunks > 1) copy_chunk(1);
    for (int c = 0; c < nchunks; ++c) {
        // this wave's copies of chunk c have landed (those of chunk c + 1, six instructions, may still travel) ...
        if (c + 1 < nchunks) __builtin_amdgcn_s_waitcnt(0x0F76);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();   // ... and everybody else's; nobody reads the buffer of chunk c - 1 any more
        if (c + 2 < nchunks) copy_chunk(c + 2);
        const float* Ab = lds + (c % 3) * BUF + (wr + i) * GW_BK;
        const float* Bb = lds + (c % 3) * BUF + A_FLOATS + (4 * h) * GW_BN + wc + i;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float4 a[4];
            float b[2][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const float4*>(Ab + rt * 32 * GW_BK + 4 * ((2 * cc + h) ^ sw));
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t) b[ct][t] = Bb[(8 * cc + t) * GW_BN + 32 * ct];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const float av = t == 0 ? a[rt].x : t == 1 ? a[rt].y : t == 2 ? a[rt].z : a[rt].w;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[ct][t], acc[rt][ct], 0, 0, 0);
                    }
        }
    }

    float cs[2] = {0.f, 0.f};
    const bool full = m0 + GW_BM <= m;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f wsum[DP > 0 ? DP : 1];   // DP: the weighted column sums of this lane's two columns (one per column tile), as pairs
    if constexpr (DP > 0) {
        // The ReLU derivative is not read but recomputed: the first layer's forward (linear_narrow_in_fwd_kernel) forms a
        // pre-activation as b + sum_k x[k] W[c][k] by fused multiply-adds in the order k = 0, 1, ... (zero padded), so
        // the same sequence gives the same bits and the same sign; the epilogue then touches no memory but the tile's
        // rows of xin.  These wait in LDS, [256][DP], zero beyond din and beyond the last row.
        float* xs = lds;
        __syncthreads();   // every wave is done with the chunk buffers
        for (int e = tid; e < GW_BM * DP; e += 256) {
            const int row = e / DP, j = e % DP;
            xs[e] = (j < din && m0 + row < m) ? xin[(m0 + row) * ldxi + j] : 0.f;
        }
        v2f w2[DP], b2;   // the first layer's weights and bias of the lane's two columns
        {
            const int64_t c0 = n0 + wc + i, c1 = c0 + 32;
            b2 = v2f{b0[c0], b0[c1]};
#pragma unroll
            for (int k = 0; k < DP; ++k) w2[k] = k < din ? v2f{w0[c0 * din + k], w0[c1 * din + k]} : v2f{0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) wsum[j] = v2f{0.f, 0.f};
        v2f cs2 = {0.f, 0.f};
        const int rows_left = (int)min((int64_t)GW_BM, m - m0) - wr - 4 * h;   // local rows below this are real
        __syncthreads();
        // A row of xin per step (both lane halves read one address each).  Two rows make a basic block of their own
        // (a condition that is always true, but not to the compiler): in one block the compiler requests all 64 rows
        // before the first sum, on top of the 128 accumulators, and spills.
        const float* xq = xs + (wr + 4 * h) * DP;
        int go = 1;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int rq = 0; rq < 8; ++rq) {
                asm volatile("" : "+s"(go));
                if (go) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int r = 2 * rq + rr;
                        const int local = 32 * rt + 8 * (r >> 2) + (r & 3);
                        const float* xr = xq + local * DP;
                        float xv[DP];
#pragma unroll
                        for (int j4 = 0; j4 < DP; j4 += 4) {
                            const float4 t = *reinterpret_cast<const float4*>(xr + j4);
                            xv[j4] = t.x; xv[j4 + 1] = t.y; xv[j4 + 2] = t.z; xv[j4 + 3] = t.w;
                        }
                        v2f pre = b2;
#pragma unroll
                        for (int k = 0; k < DP; ++k) pre = __builtin_elementwise_fma(v2f{xv[k], xv[k]}, w2[k], pre);
                        const bool real = full || local < rows_left;
                        const v2f v = {(pre[0] > 0.f && real) ? acc[rt][0][r] : 0.f, (pre[1] > 0.f && real) ? acc[rt][1][r] : 0.f};
                        cs2 += v;
#pragma unroll
                        for (int j = 0; j < DP; ++j) wsum[j] = __builtin_elementwise_fma(v, v2f{xv[j], xv[j]}, wsum[j]);
                    }
                }
            }
        cs[0] = cs2[0];
        cs[1] = cs2[1];
    } else {
        constexpr int AHEAD = 3;
        float hv[AHEAD][16];
        auto request = [&](const int q) {
            const int rt = q >> 1, ct = q & 1;
            const int col = n0 + wc + 32 * ct + i;
            const int64_t row0 = m0 + wr + 32 * rt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + 8 * (r >> 2) + (r & 3);
                hv[q % AHEAD][r] = hprev[(full ? row : min(row, m - 1)) * ldh + col];
            }
        };
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) request(q);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rt = q >> 1, ct = q & 1;
            const int col = n0 + wc + 32 * ct + i;
            const int64_t row0 = m0 + wr + 32 * rt + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + 8 * (r >> 2) + (r & 3);
                const float v = hv[q % AHEAD][r] > 0.f ? acc[rt][ct][r] : 0.f;
                if (full || row < m) {
                    dx[row * ldx + col] = v;
                    cs[ct] += v;
                }
            }
            if (q + AHEAD < 8) request(q + AHEAD);
        }
    }
    __syncthreads();   // (the last chunk's buffer may be buffer 0: every wave is done with it; DP: everybody has read xs)
    if constexpr (DP > 0) {
        float* red = lds;   // [2 row halves][DP + 1][128]
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            cs[ct] += __shfl_xor(cs[ct], 32);
            if (h == 0) red[((wave >> 1) * (DP + 1) + DP) * GW_BN + wc + 32 * ct + i] = cs[ct];
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                const float t = wsum[j][ct] + __shfl_xor(wsum[j][ct], 32);
                if (h == 0) red[((wave >> 1) * (DP + 1) + j) * GW_BN + wc + 32 * ct + i] = t;
            }
        }
        __syncthreads();
        for (int e = tid; e < (din + 1) * GW_BN; e += 256) {
            const int j = e / GW_BN, c = e % GW_BN, js = j < din ? j : DP;
            partial[((tile / ntn) * (din + 1) + j) * N + n0 + c] = red[js * GW_BN + c] + red[((DP + 1) + js) * GW_BN + c];
        }
    } else {
        float* red = lds;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            cs[ct] += __shfl_xor(cs[ct], 32);
            if (h == 0) red[(wave >> 1) * GW_BN + wc + 32 * ct + i] = cs[ct];
        }
        __syncthreads();
        if (tid < GW_BN) partial[(tile / ntn) * N + n0 + tid] = red[tid] + red[GW_BN + tid];
    }
}

extern "C" size_t sparc_dgrad_narrow_relu_workspace_bytes(int64_t m, int N) {
    if (m < 0 || N <= 0) return 0;
    return sparc_align_up((size_t)DGRAD_CHUNKS * (size_t)N * sizeof(float)) + 256;
}

extern "C" int sparc_dgrad_narrow_relu(int64_t m, int n_out, int N, const float* g, int64_t ldg, const float* W,
                                       const float* hprev, int64_t ldh, float* dx, int64_t ldx, float* db_prev, void* ws,
                                       size_t ws_bytes, sparc_stream_t stream) {
    if (m < 0 || n_out <= 0 || n_out > SPARC_MAX_K || N <= 0 || N % 64 != 0 || !g || !W || !hprev || !dx || !db_prev ||
        ldg < n_out || ldh < N || ldx < N || ldh % 4 != 0 || ldx % 4 != 0 || !aligned16(W) || !aligned16(hprev) || !aligned16(dx))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_dgrad_narrow_relu_workspace_bytes(m, N) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0) return hipMemsetAsync(db_prev, 0, (size_t)N * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    const int64_t rpb = dgrad_rows_per_block(m);
    const int nparts = (int)((m + rpb - 1) / rpb);
    float* partial = static_cast<float*>(ws);
    dim3 grid(N / 64, nparts);
    switch ((n_out + 15) / 16) {
        case 1: hipLaunchKernelGGL((dgrad_narrow_relu_kernel<1>), grid, dim3(256), 0, stream, m, n_out, N, g, ldg, W, hprev, ldh, dx, ldx, partial, rpb); break;
        case 2: hipLaunchKernelGGL((dgrad_narrow_relu_kernel<2>), grid, dim3(256), 0, stream, m, n_out, N, g, ldg, W, hprev, ldh, dx, ldx, partial, rpb); break;
        case 3: hipLaunchKernelGGL((dgrad_narrow_relu_kernel<3>), grid, dim3(256), 0, stream, m, n_out, N, g, ldg, W, hprev, ldh, dx, ldx, partial, rpb); break;
        default: hipLaunchKernelGGL((dgrad_narrow_relu_kernel<4>), grid, dim3(256), 0, stream, m, n_out, N, g, ldg, W, hprev, ldh, dx, ldx, partial, rpb); break;
    }
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finish, dim3((N + 63) / 64), dim3(64 * COLSUM_LANES), 0, stream, nparts, N, partial, db_prev, 0);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" size_t sparc_dgrad_wide_relu_workspace_bytes(int64_t m, int N) {
    if (m < 0 || N <= 0) return 0;
    return sparc_align_up((size_t)((m + GW_BM - 1) / GW_BM) * (size_t)N * sizeof(float)) + 256;
}

extern "C" int sparc_dgrad_wide_relu(int64_t m, int K, int N, const float* g, int64_t ldg, const float* W, const float* hprev,
                                     int64_t ldh, float* dx, int64_t ldx, float* db_prev, void* ws, size_t ws_bytes,
                                     sparc_stream_t stream) {
    if (m < 0 || K <= 0 || K % GW_BK != 0 || N <= 0 || N % GW_BN != 0 || !g || !W || !hprev || !dx || !db_prev || ldg < K ||
        ldg % 4 != 0 || ldh < N || ldx < N || !aligned16(g) || !aligned16(W))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_dgrad_wide_relu_workspace_bytes(m, N) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0) return hipMemsetAsync(db_prev, 0, (size_t)N * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    const int64_t mt = (m + GW_BM - 1) / GW_BM, tiles = mt * (N / GW_BN);
    if (tiles > 0x7fffffffLL) return SPARC_ERR_SIZE;
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(dgrad_wide_relu_kernel<0>, dim3((unsigned)tiles), dim3(256), 0, stream, m, K, N, g, ldg, W, hprev, ldh, dx,
                       ldx, partial, static_cast<const float*>(nullptr), (int64_t)0, 0, static_cast<const float*>(nullptr),
                       static_cast<const float*>(nullptr));
    SPARC_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finish, dim3((N + 63) / 64), dim3(64 * COLSUM_LANES), 0, stream, (int)mt, N, partial, db_prev, 0);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}

extern "C" size_t sparc_dgrad_wide_first_workspace_bytes(int64_t m, int N, int din) {
    if (m < 0 || N <= 0 || din <= 0) return 0;
    return sparc_align_up((size_t)((m + GW_BM - 1) / GW_BM) * (size_t)(din + 1) * (size_t)N * sizeof(float)) + 256;
}

extern "C" int sparc_dgrad_wide_first(int64_t m, int K, int N, int din, const float* g, int64_t ldg, const float* W,
                                      const float* xin, int64_t ldxi, const float* W0, const float* b0, float* dW_db, void* ws,
                                      size_t ws_bytes, sparc_stream_t stream) {
    if (m < 0 || K <= 0 || K % GW_BK != 0 || N <= 0 || N % GW_BN != 0 || din <= 0 || din > 16 || !g || !W || !xin || !W0 || !b0 ||
        !dW_db || ldg < K || ldg % 4 != 0 || ldxi < din || !aligned16(g) || !aligned16(W))
        return SPARC_ERR_ARG;
    if (ws_bytes < sparc_dgrad_wide_first_workspace_bytes(m, N, din) || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return SPARC_ERR_WS;
    if (m == 0)
        return hipMemsetAsync(dW_db, 0, (size_t)(din + 1) * N * sizeof(float), stream) == hipSuccess ? SPARC_OK : SPARC_ERR_LAUNCH;
    const int64_t mt = (m + GW_BM - 1) / GW_BM, tiles = mt * (N / GW_BN);
    if (tiles > 0x7fffffffLL) return SPARC_ERR_SIZE;
    float* partial = static_cast<float*>(ws);
#define SPARC_GW_FIRST(DP_)                                                                                                  \
    hipLaunchKernelGGL(dgrad_wide_relu_kernel<DP_>, dim3((unsigned)tiles), dim3(256), 0, stream, m, K, N, g, ldg, W,             \
                       static_cast<const float*>(nullptr), (int64_t)0, static_cast<float*>(nullptr), (int64_t)0, partial, xin,   \
                       ldxi, din, W0, b0)
    if (din <= 4) SPARC_GW_FIRST(4);
    else if (din <= 8) SPARC_GW_FIRST(8);
    else if (din <= 12) SPARC_GW_FIRST(12);
    else SPARC_GW_FIRST(16);
#undef SPARC_GW_FIRST
    SPARC_CHECK_LAUNCH();
    const int cols = (din + 1) * N;
    hipLaunchKernelGGL(colsum_finish, dim3((cols + 63) / 64), dim3(64 * COLSUM_LANES), 0, stream, (int)mt, cols, partial, dW_db, din);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
