// What would one launch for the three passes of the Laplacian chain save?  Three dependent launches of a kernel
// with the windowed kernel's shape (450 workgroups of 768 threads, 64 KB of LDS: two per CU, all resident at once)
// against ONE launch of the same work with two grid-wide barriers (an arrival counter in device memory, spun on
// with device-scope loads, sense-reversed by generation).  The body is a stand-in that touches its LDS window and
// writes a few words; what is measured is the fixed cost per pass: launch + ramp + drain against barrier.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ unsigned int g_arrived, g_generation;

__device__ void body(float* out, const float* in, int pass) {
    extern __shared__ float win[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) win[i] = in[(blockIdx.x * 16384 + i) & 0xfffff];
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) t += win[(i * 33) & 16383];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = t + pass;
}

__global__ __launch_bounds__(768) void one_pass(float* out, const float* in, int pass) { body(out, in, pass); }

__device__ void grid_barrier(unsigned int nblocks, int fenced) {
    // fenced: the real chain hands rows of the operand to other workgroups, so its stores must be visible device-wide
    // (a device-scope release writes the XCD's L2 back); unfenced: the cost of the meeting alone
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int gen = __hip_atomic_load(&g_generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&g_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nblocks) {
            __hip_atomic_store(&g_arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g_generation, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spin = 0;
            while (__hip_atomic_load(&g_generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (++spin > (1 << 22)) __builtin_trap();   // every workgroup is resident: cannot happen
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(768) void three_passes(float* a, float* b, int fenced) {
    body(b, a, 0);
    grid_barrier(gridDim.x, fenced);
    body(a, b, 1);
    grid_barrier(gridDim.x, fenced);
    body(b, a, 2);
}

int main() {
    const int nb = 450, nt = 768, lds = 65536, reps = 200;
    float *a, *b;
    CHECK(hipMalloc(&a, 4 << 20));
    CHECK(hipMalloc(&b, 4 << 20));
    CHECK(hipMemset(a, 0, 4 << 20));
    CHECK(hipFuncSetAttribute((const void*)one_pass, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)three_passes, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, three_passes, nt, lds));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("resident workgroups: %d per CU x %d CUs = %d (need %d)\n", per_cu, prop.multiProcessorCount, per_cu * prop.multiProcessorCount, nb);
    if (per_cu * prop.multiProcessorCount < nb) { printf("not all resident: no barrier run\n"); return 1; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) {
                    hipLaunchKernelGGL(one_pass, dim3(nb), dim3(nt), lds, 0, b, a, 0);
                    hipLaunchKernelGGL(one_pass, dim3(nb), dim3(nt), lds, 0, a, b, 1);
                    hipLaunchKernelGGL(one_pass, dim3(nb), dim3(nt), lds, 0, b, a, 2);
                } else {
                    hipLaunchKernelGGL(three_passes, dim3(nb), dim3(nt), lds, 0, a, b, mode == 1 ? 1 : 0);
                }
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (warm) printf("%s: %.2f us per chain of three passes\n", mode == 0 ? "three launches" : mode == 1 ? "one launch, two grid barriers with device-scope release/acquire" : "one launch, two grid barriers, no fences (the meeting alone)", 1e3 * ms / reps);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
