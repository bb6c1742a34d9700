// Finds the operand layout of v_smfmac_f32_16x16x64_bf16 (gfx950: A is 2:4 structured sparse, 8 stored bf16 per lane
// plus 2-bit positions in an index register; B dense, 16 bf16 per lane) by experiment: one stored A element per
// row at a time against a B whose every element carries its own (lane, element) code.
//   hipcc --offload-arch=gfx950 -O3 -o smfmac_probe smfmac_probe.hip && ./smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned short* A, const unsigned short* B, const int* idx, float* out) {
    const int l = threadIdx.x;
    bf16x8 a; bf16x16 b;
    unsigned short ar[8], br[16];
    for (int i = 0; i < 8; ++i) ar[i] = A[l * 8 + i];
    for (int i = 0; i < 16; ++i) br[i] = B[l * 16 + i];
    __builtin_memcpy(&a, ar, 16); __builtin_memcpy(&b, br, 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx[l], 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe32(const unsigned short* A, const unsigned short* B, const int* idx, float* out) {
    const int l = threadIdx.x;
    bf16x8 a; bf16x16 b;
    unsigned short ar[8], br[16];
    for (int i = 0; i < 8; ++i) ar[i] = A[l * 8 + i];
    for (int i = 0; i < 16; ++i) br[i] = B[l * 16 + i];
    __builtin_memcpy(&a, ar, 16); __builtin_memcpy(&b, br, 32);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, idx[l], 0, 0);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return static_cast<unsigned short>(u >> 16); }

int main() {
    unsigned short *dA, *dB; int* dI; float* dO;
    CK(hipMalloc(&dA, 64 * 8 * 2)); CK(hipMalloc(&dB, 64 * 16 * 2)); CK(hipMalloc(&dI, 64 * 4)); CK(hipMalloc(&dO, 64 * 4 * 4));
    std::vector<unsigned short> A(64 * 8), B(64 * 16);
    std::vector<int> I(64);
    std::vector<float> O(256), O2(256);
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < 8; ++s)
            for (int v = 0; v < 4; ++v) {
                std::fill(A.begin(), A.end(), 0);
                for (int l = 16 * g; l < 16 * g + 16; ++l) A[l * 8 + s] = bf16_of(1.f);
                for (int l = 0; l < 64; ++l) {
                    int w = 0;
                    for (int t = 0; t < 8; ++t) w |= ((t == s ? v : ((t & 1) ? 3 : 0))) << (2 * t);   // partners: positions 0 and 3
                    I[l] = w;
                }
                CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
                CK(hipMemcpy(dI, I.data(), I.size() * 4, hipMemcpyHostToDevice));
                for (int pass = 0; pass < 2; ++pass) {
                    for (int lb = 0; lb < 64; ++lb)
                        for (int e = 0; e < 16; ++e) {
                            const int code = lb * 16 + e;
                            B[lb * 16 + e] = bf16_of(pass == 0 ? float(code % 128 + 1) : float(code / 128 + 1));
                        }
                    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dI, dO);
                    CK(hipMemcpy(pass == 0 ? O.data() : O2.data(), dO, 1024, hipMemcpyDeviceToHost));
                }
                // which output entries are set, and which B element each one saw
                printf("A lanes %2d..%2d slot %d pos %d:", 16 * g, 16 * g + 15, s, v);
                int shown = 0;
                for (int l = 0; l < 64 && shown < 3; ++l)
                    for (int r = 0; r < 4 && shown < 3; ++r)
                        if (O[l * 4 + r] != 0.f) {
                            const int code = (int(O[l * 4 + r]) - 1) + 128 * (int(O2[l * 4 + r]) - 1);
                            printf("  out[lane %d reg %d] <- B[lane %d elem %d]", l, r, code / 16, code % 16);
                            ++shown;
                        }
                int nz = 0;
                for (float x : O) nz += x != 0.f;
                printf("  (%d outputs set)\n", nz);
            }
    // v_smfmac_f32_32x32x32_bf16: lane groups of 32
    printf("---- 32x32x32\n");
    float* dO32; CK(hipMalloc(&dO32, 64 * 16 * 4));
    std::vector<float> P(1024), P2(1024);
    for (int g = 0; g < 2; ++g)
        for (int s = 0; s < 8; ++s)
            for (int v = 0; v < 4; v += 3) {
                std::fill(A.begin(), A.end(), 0);
                for (int l = 32 * g; l < 32 * g + 32; ++l) A[l * 8 + s] = bf16_of(1.f);
                for (int l = 0; l < 64; ++l) {
                    int w = 0;
                    for (int t = 0; t < 8; ++t) w |= ((t == s ? v : ((t & 1) ? 3 : 0))) << (2 * t);
                    I[l] = w;
                }
                CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
                CK(hipMemcpy(dI, I.data(), I.size() * 4, hipMemcpyHostToDevice));
                for (int pass = 0; pass < 2; ++pass) {
                    for (int lb = 0; lb < 64; ++lb)
                        for (int e = 0; e < 16; ++e) {
                            const int code = lb * 16 + e;
                            B[lb * 16 + e] = bf16_of(pass == 0 ? float(code % 128 + 1) : float(code / 128 + 1));
                        }
                    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dI, dO32);
                    CK(hipMemcpy(pass == 0 ? P.data() : P2.data(), dO32, 4096, hipMemcpyDeviceToHost));
                }
                printf("A lanes %2d..%2d slot %d pos %d:", 32 * g, 32 * g + 31, s, v);
                int shown = 0;
                for (int l = 0; l < 64 && shown < 2; l += 33)
                    for (int r = 0; r < 16 && shown < 2; r += 5)
                        if (P[l * 16 + r] != 0.f) {
                            const int code = (int(P[l * 16 + r]) - 1) + 128 * (int(P2[l * 16 + r]) - 1);
                            printf("  out[lane %d reg %d] <- B[lane %d elem %d]", l, r, code / 16, code % 16);
                            ++shown;
                        }
                int nz = 0;
                for (float x : P) nz += x != 0.f;
                printf("  (%d outputs set)\n", nz);
            }
    return 0;
}
