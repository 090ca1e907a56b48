// Probe: is a chain of v_mfma_f32_16x16x4_f32 with the k's of each instruction mapped as {0,4,1,5},{2,6,3,7},{8,12,9,13},
// {10,14,11,15} per 16-k tile bit-identical to the chain of v_mfma_f32_32x32x2_f32 the GEMM kernels use
// (per tile: j = 0,1; i = 0..3: lane half lh contracts k = 4*lh + 8*j + i)?  Also compares with a scalar fmaf chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A[32][K], B[32][K] (Bt layout) -> C32[32][32] via 32x32x2 exactly like gemm_skinny
__global__ void k32(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kt = 0; kt < K / 16; ++kt) {
        const float* pa = A + l31 * K + kt * 16 + 4 * lh; const float* pb = B + l31 * K + kt * 16 + 4 * lh;
        f32x4 a0 = *(const f32x4*)pa, a1 = *(const f32x4*)(pa + 8), b0 = *(const f32x4*)pb, b1 = *(const f32x4*)(pb + 8);
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) { const int m = (r & 3) + 8 * (r >> 2) + 4 * lh; C[m * 32 + l31] = acc[r]; }
}
// rows 0..15 x cols 0..15 via 16x16x4 with the k map
__global__ void k16(const float* A, const float* B, float* C, int K, int variant) {
    const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
    const int map[4][4] = {{0, 4, 1, 5}, {2, 6, 3, 7}, {8, 12, 9, 13}, {10, 14, 11, 15}};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < K / 16; ++kt)
        for (int q = 0; q < 4; ++q) {
            int k = kt * 16 + map[q][kg];
            if (variant == 1) k = kt * 16 + map[q][3 - kg];      // reversed order inside the instruction
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k], B[r * K + k], acc, 0, 0, 0);
        }
    for (int q = 0; q < 4; ++q) C[(kg * 4 + q) * 16 + r] = acc[q];      // row = kg*4+q, col = r
}
int main() {
    const int K = 768;
    std::vector<float> A(32 * K), B(32 * K);
    unsigned s = 1234567u;
    for (auto& v : A) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) / 7919.0f; }
    for (auto& v : B) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) / 104729.0f; }
    float *dA, *dB, *dC32, *dC16, *dC16b;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC32, 32 * 32 * 4); hipMalloc(&dC16, 16 * 16 * 4); hipMalloc(&dC16b, 16 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC32, K);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC16, K, 0);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC16b, K, 1);
    std::vector<float> c32(32 * 32), c16(16 * 16), c16b(16 * 16);
    hipMemcpy(c32.data(), dC32, c32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c16.data(), dC16, c16.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c16b.data(), dC16b, c16b.size() * 4, hipMemcpyDeviceToHost);
    int same = 0, sameb = 0, samef = 0, samef32 = 0;
    const int order[16] = {0, 4, 1, 5, 2, 6, 3, 7, 8, 12, 9, 13, 10, 14, 11, 15};
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float f = 0.f;
            for (int kt = 0; kt < K / 16; ++kt) for (int j = 0; j < 16; ++j) { const int k = kt * 16 + order[j]; f = __builtin_fmaf(A[m * K + k], B[n * K + k], f); }
            same += !memcmp(&c32[m * 32 + n], &c16[m * 16 + n], 4);
            sameb += !memcmp(&c32[m * 32 + n], &c16b[m * 16 + n], 4);
            samef += !memcmp(&f, &c16[m * 16 + n], 4);
            samef32 += !memcmp(&f, &c32[m * 32 + n], 4);
        }
    printf("16x16x4(mapped) == 32x32x2 chain: %d / 256 ; reversed-in-instruction variant: %d / 256 ; host fmaf chain == 16x16x4: %d ; == 32x32x2: %d\n", same, sameb, samef, samef32);
    printf("sample %.9g %.9g\n", c32[0], c16[0]);
    return 0;
}
