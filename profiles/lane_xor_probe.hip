#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MASK> __device__ __forceinline__ int lane_xor_i(int x) {
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
    else if constexpr (MASK == 4) return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    else if constexpr (MASK == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);
    else if constexpr (MASK == 16) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((threadIdx.x & 16) ? r[0] : r[1]);
    } else {
        const u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((threadIdx.x & 32) ? r[0] : r[1]);
    }
}
__global__ void k(int* out) {
    const int x = threadIdx.x * 1000 + 7;
    out[0 * 64 + threadIdx.x] = lane_xor_i<1>(x) - __shfl_xor(x, 1, 64);
    out[1 * 64 + threadIdx.x] = lane_xor_i<2>(x) - __shfl_xor(x, 2, 64);
    out[2 * 64 + threadIdx.x] = lane_xor_i<4>(x) - __shfl_xor(x, 4, 64);
    out[3 * 64 + threadIdx.x] = lane_xor_i<8>(x) - __shfl_xor(x, 8, 64);
    out[4 * 64 + threadIdx.x] = lane_xor_i<16>(x) - __shfl_xor(x, 16, 64);
    out[5 * 64 + threadIdx.x] = lane_xor_i<32>(x) - __shfl_xor(x, 32, 64);
}
int main() {
    int* d; hipMalloc(&d, 6 * 64 * 4); hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    int h[6 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int m = 0; m < 6; ++m) { int bad = 0; for (int i = 0; i < 64; ++i) bad += h[m * 64 + i] != 0; printf("xor %d: %d lanes differ\n", 1 << m, bad); }
    return 0;
}
