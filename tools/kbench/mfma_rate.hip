// Sustained rate of the float32 matrix instructions (operands in registers, nothing else in the loop):
//   mfma_rate <variant> <waves per SIMD: 1|2> <milliseconds>
// variant 0: v_mfma_f32_32x32x2_f32, 8 accumulators; 1: v_mfma_f32_16x16x4_f32, 32 accumulators;
//         2: v_mfma_f32_16x16x1_4b_f32 (four 16x16 blocks), 8 accumulators; 3: v_mfma_f32_32x32x1_2b_f32, 4 accumulators x 32 regs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v32f __attribute__((ext_vector_type(32)));

template <int V>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters, float a0, float b0) {
    const float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
    float sum = 0.f;
    if (V == 0) {
        v16f acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    } else if (V == 1) {
        v4f acc[32];
        for (int q = 0; q < 32; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 32; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 32; ++q) for (int r = 0; r < 4; ++r) sum += acc[q][r];
    } else if (V == 2) {
        v16f acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    } else {
        v32f acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 32; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 32; ++r) sum += acc[q][r];
    }
    if (sum == 123.f) out[0] = sum;
}

int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0, wps = argc > 2 ? atoi(argv[2]) : 2;
    const double ms_target = argc > 3 ? atof(argv[3]) : 2.0;
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    // flop per loop iteration and wave: V0 64 x 4096, V1 128 x 2048, V2 128 x 2048 (16x16x1 x 4 blocks), V3 32 x 4096 x ... (32x32x1 x 2 blocks = 4096)
    const double flop_iter = v == 0 ? 64 * 4096.0 : v == 1 ? 128 * 2048.0 : v == 2 ? 128 * 2048.0 : 32 * 4096.0;
    const int blocks = 256 * wps;   // 4 waves per block: one per SIMD
    int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        switch (v) {
            case 0: hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); break;
            case 1: hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); break;
            case 2: hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); break;
            default: hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); break;
        }
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) return 2;
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_iter * iters * blocks * 4 / (ms * 1e-3) / 1e12;
        printf("variant %d, %d waves/SIMD, %d iterations: %.3f ms, %.1f TFLOP/s\n", v, wps, iters, ms, tf);
        iters = (int)(iters * ms_target / ms) + 1;
    }
    return 0;
}
