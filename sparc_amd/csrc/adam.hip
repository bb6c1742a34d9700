// One Adam update of every parameter tensor of a model in ONE launch that fills the chip.
//
// The reference steps torch.optim.Adam with its defaults (src/SpectralTrainer.py:188, :238; the autoencoder stage
// src/AETrainer.py likewise).  The MLP here has eight parameter tensors of 47 to a million floats; a multi-tensor
// kernel that deals one workgroup per 65 536 elements puts 25 workgroups on 256 CUs (46 us for 45 MB of state), so
// this one gives every 1024 elements a workgroup: a thread updates four consecutive elements with 16-byte accesses.
// Arithmetic of one element, in float32 with the step's scalars prepared in float64 on the host (the formulas of
// torch.optim.Adam, amsgrad = False, maximize = False):
//     g' = g + weight_decay p;   m = m + (1 - beta1) (g' - m);   v = beta2 v + (1 - beta2) g' g'
//     p = p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "common.h"

namespace {

constexpr int ADAM_BLOCK_ELEMS = 1024;   // elements per workgroup (256 threads x 4)

struct AdamArgs {
    sparc_adam_tensor_t t[SPARC_ADAM_MAX_TENSORS];
    int32_t first_block[SPARC_ADAM_MAX_TENSORS + 1];   // workgroups [first_block[i], first_block[i + 1]) belong to tensor i
    int n_tensors;
    float step_size, bc2_rsqrt, beta2, one_minus_beta1, one_minus_beta2, eps, weight_decay;
};

__device__ __forceinline__ void adam_one(float& p, const float g0, float& m, float& v, const AdamArgs& a) {
    const float g = a.weight_decay != 0.f ? fmaf(a.weight_decay, p, g0) : g0;
    m = fmaf(a.one_minus_beta1, g - m, m);
    v = fmaf(a.beta2, v, a.one_minus_beta2 * g * g);
    const float denom = sqrtf(v) * a.bc2_rsqrt + a.eps;
    p -= a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
    int ti = 0;   // the tensor of this workgroup (at most 32: a short scan of wave-uniform values)
    while (ti + 1 < a.n_tensors && static_cast<int>(blockIdx.x) >= a.first_block[ti + 1]) ++ti;
    const sparc_adam_tensor_t t = a.t[ti];
    const int64_t e = (static_cast<int64_t>(blockIdx.x) - a.first_block[ti]) * ADAM_BLOCK_ELEMS + threadIdx.x * 4;
    if (e >= t.n) return;
    const bool vec = e + 4 <= t.n && ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    if (vec) {
        float4 p = *reinterpret_cast<const float4*>(t.p + e), m = *reinterpret_cast<const float4*>(t.m + e);
        float4 v = *reinterpret_cast<const float4*>(t.v + e);
        const float4 g = *reinterpret_cast<const float4*>(t.g + e);
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        *reinterpret_cast<float4*>(t.p + e) = p;
        *reinterpret_cast<float4*>(t.m + e) = m;
        *reinterpret_cast<float4*>(t.v + e) = v;
    } else {
        for (int64_t i = e; i < min(e + 4, t.n); ++i) {
            float p = t.p[i], m = t.m[i], v = t.v[i];
            adam_one(p, t.g[i], m, v, a);
            t.p[i] = p; t.m[i] = m; t.v[i] = v;
        }
    }
}

}  // namespace

extern "C" int sparc_adam_step(int n_tensors, const sparc_adam_tensor_t* tensors, double lr, double beta1, double beta2,
                               double eps, double weight_decay, int64_t step, sparc_stream_t stream) {
    if (n_tensors < 0 || n_tensors > SPARC_ADAM_MAX_TENSORS || (n_tensors > 0 && !tensors) || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) ||
        !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return SPARC_ERR_ARG;
    AdamArgs a = {};
    int64_t blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const sparc_adam_tensor_t& t = tensors[i];
        if (t.n < 0 || (t.n > 0 && (!t.p || !t.g || !t.m || !t.v))) return SPARC_ERR_ARG;
        a.t[i] = t;
        a.first_block[i] = static_cast<int32_t>(blocks);
        blocks += (t.n + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS;
        if (blocks > 0x7fffffffLL) return SPARC_ERR_ARG;
    }
    a.first_block[n_tensors] = static_cast<int32_t>(blocks);
    a.n_tensors = n_tensors;
    if (blocks == 0) return SPARC_OK;
    const double bc1 = 1.0 - pow(beta1, static_cast<double>(step)), bc2 = 1.0 - pow(beta2, static_cast<double>(step));
    a.step_size = static_cast<float>(lr / bc1);
    a.bc2_rsqrt = static_cast<float>(1.0 / sqrt(bc2));
    a.beta2 = static_cast<float>(beta2);
    a.one_minus_beta1 = static_cast<float>(1.0 - beta1);
    a.one_minus_beta2 = static_cast<float>(1.0 - beta2);
    a.eps = static_cast<float>(eps);
    a.weight_decay = static_cast<float>(weight_decay);
    hipLaunchKernelGGL(adam_step_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, a);
    SPARC_CHECK_LAUNCH();
    return SPARC_OK;
}
