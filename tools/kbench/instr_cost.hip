// Throughput of the vector instructions of the windowed SpMM's inner loop (gfx950): every SIMD of the
// chip runs W waves that each issue independent copies of one instruction; reported as nanoseconds and
// (at the measured 2.4 GHz-class clock) cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o instr_cost instr_cost.hip && ./instr_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
constexpr int ITERS = 4000;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void bench(float* out, int dummy) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, b0 = 0.5f, b1 = 0.25f;
    int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a0, a3}, p3 = {a1, a2}, w = {b0, b1};
    // conflict-free LDS addresses: a DPP row reads one 16-lane-contiguous row, rows differ per DPP row
    const int lane = threadIdx.x & 63;
    const int addr64 = ((lane >> 4) * 37 % 128) * 128 + (lane & 15) * 8 + (dummy & 0x3f80);
    const int addr128 = ((lane >> 4) * 37 % 64) * 256 + (lane & 15) * 16 + (dummy & 0x3f00);
    const int addr96 = ((lane >> 4) * 37 % 64) * 192 + (lane & 15) * 12 + (dummy & 0x3f00);   // 12-byte lane stride: 4-byte aligned only
    f2 q0, q1, q2, q3; float4 r0, r1, r2, r3;
    typedef float f3 __attribute__((ext_vector_type(3)));
    f3 t0, t1, t2, t3;
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) { REP4(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (KIND == 1) { REP4(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w), "v"(w));) }
        if (KIND == 2) { REP4(asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(dummy));) }
        if (KIND == 3) { REP4(asm volatile("v_add_u32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(dummy), "v"(lane));) }
        if (KIND == 4) { REP4(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8192\n ds_read_b64 %2, %4 offset:16384\n ds_read_b64 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(addr64));) }
        if (KIND == 5) { REP4(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16384\n ds_read_b128 %2, %4 offset:32768\n ds_read_b128 %3, %4 offset:49152\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr128));) }
        if (KIND == 8) { REP4(asm volatile("ds_read_b96 %0, %4\n ds_read_b96 %1, %4 offset:12288\n ds_read_b96 %2, %4 offset:24576\n ds_read_b96 %3, %4 offset:36864\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(addr96));) }
        if (KIND == 6) { REP4(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(dummy));) }
        if (KIND == 7) {  // the SpMM step: 4 x (add_dpp, ds_read_b64, 2 mov_dpp, pk_fma)
            asm volatile(
                "v_add_u32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_u32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "s_nop 1\n"
                "ds_read_b64 %4, %0\n ds_read_b64 %5, %1\n ds_read_b64 %6, %2\n ds_read_b64 %7, %3\n"
                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr64), "v"(dummy));
            asm volatile(
                "v_mov_b32_dpp %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %2, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %0, %5 row_newbcast:0 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %1, %5 row_newbcast:1 row_mask:0xf bank_mask:0xc\n"
                "v_mov_b32_dpp %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %3, %5 row_newbcast:3 row_mask:0xf bank_mask:0xc\n"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(b0), "v"(b1));
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"
                "v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %3, %7, %1 op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 %0, %4, %8, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %5, %9, %1 op_sel_hi:[0,1,1]\n"
                : "+v"(p0), "+v"(p1) : "v"(w), "v"(p2), "v"(p3), "v"(w), "v"(q0), "v"(q1), "v"(q2), "v"(q3));
        }
    }
    if (KIND == 8) {  // the unaligned 12-byte reads must return lds[addr/4 .. +2] (lds[i] = i)
        const float want = (addr96 + 36864) / 4;
        if (t3[0] != want || t3[1] != want + 1 || t3[2] != want + 2) r0.x = -1e30f; else r0.x = 0.f;
        r1.x = r2.x = r3.x = 0.f; a0 = a1 = a2 = a3 = 0.f; p0 = p1 = p2 = p3 = f2{0.f, 0.f}; x0 = x1 = x2 = x3 = 0; q0 = q1 = q2 = q3 = f2{0.f, 0.f};
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1] + x0 + x1 + x2 + x3 +
                                                 q0[0] + q1[0] + q2[0] + q3[0] + r0.x + r1.x + r2.x + r3.x;
}

template <int KIND>
void run(const char* name, int per_iter) {
    float* out;
    hipMalloc(&out, 2048 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int threads = cfg == 0 ? 256 : cfg == 1 ? 512 : 1024;
        const int grid = cfg == 3 ? 512 : 256;
        const int waves_per_simd = threads / 256 * (grid / 256);
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(threads), 0, 0, out, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(threads), 0, 0, out, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = double(ITERS) * per_iter * waves_per_simd;
        printf("%-34s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (%.0f us)\n", name, waves_per_simd, ms * 1e6 / instr_per_simd, ms * 1e3);
    }
    if (KIND == 8) {
        std::vector<float> h(1024);
        hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (float v : h) bad += v != 0.f;
        printf("ds_read_b96 at 4-byte alignment: %s (%d of 1024 lanes differ)\n", bad ? "WRONG DATA" : "data correct", bad);
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 16);
    run<6>("v_mov_b32", 16);
    run<2>("v_mov_b32_dpp row_newbcast", 16);
    run<3>("v_add_u32_dpp row_newbcast", 16);
    run<4>("ds_read_b64 (4 then wait)", 16);
    run<5>("ds_read_b128 (4 then wait)", 16);
    run<8>("ds_read_b96, 12-B stride (4 then wait)", 16);
    run<7>("spmm group (20 instr)", 20);
    return 0;
}
