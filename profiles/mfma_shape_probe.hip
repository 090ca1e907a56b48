// MFMA shape probe (round 5; not part of the library): the product stream of the split-bf16 GEMM (csrc/gemm_split.hip: eight
// waves per workgroup, one workgroup per CU, every wave 64 rows x 128 columns, six bf16 products per block and K step, fragments
// re-read from LDS every step) on v_mfma_f32_32x32x16_bf16 (SHAPE=32: what the library runs) and on v_mfma_f32_16x16x32_bf16
// (SHAPE=16) -- same FLOP, same LDS bytes per unit of K, same cycles on paper.  MI355X_MICROARCH.md ('DVFS give-back', item 7)
// reports that the chip holds a higher clock on the 16x16x32 shape when the bf16 pipe is power-limited; the library's kernel is
// (1.85-1.96 GHz with real operands).  This measures what the shape is worth for THIS product stream.
//   hipcc -O3 --offload-arch=gfx950 -DSHAPE=32 mfma_shape_probe.hip -o probe32 ; ./probe32 [launches] [ksteps]
// LDS holds random bf16 planes (the operands toggle the pipe as real data does; constant registers run at 2.38 GHz and say
// nothing); every K step reads its fragments at another offset.  Prints us per launch, TFLOP/s of bf16 products and the shader
// clock inside the loop (s_memtime / s_memrealtime of workgroup 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef SHAPE
#define SHAPE 32
#endif
#ifndef READS
#define READS 1          // 0: fragments stay in registers (constant operands)
#endif

constexpr int LDS_BYTES = 96 * 1024;       // the library's two tile buffers

__global__ __launch_bounds__(512, 1) void probe(const uint4* __restrict__ fill, float* __restrict__ out, int ksteps, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = fill[(i + 977 * blockIdx.x) % (LDS_BYTES / 16)];
    __syncthreads();
    // conflict-free 16-byte reads: lane -> its own 16 bytes of a 1-KB line; lines differ per fragment and per step
    const char* base = smem + lane * 16;
    auto frag = [&](int line) { return *reinterpret_cast<const bf16x8*>(base + (line % (LDS_BYTES / 1024)) * 1024); };
    unsigned long long t0 = 0, r0 = 0;
#if SHAPE == 32
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    bf16x8 fa[2][3], fb[4][3];
    for (int a = 0; a < 2; ++a) for (int p = 0; p < 3; ++p) fa[a][p] = frag(a * 3 + p + wave);
    for (int c = 0; c < 4; ++c) for (int p = 0; p < 3; ++p) fb[c][p] = frag(6 + c * 3 + p + wave);
    if (tid == 0 && blockIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int s = 0; s < 2 * ksteps; ++s) {            // K = 16 per step: 48 products, 18 fragment reads
#if READS
        const int o = 18 * s + 5 * wave;
        for (int a = 0; a < 2; ++a) for (int p = 0; p < 3; ++p) fa[a][p] = frag(o + a * 3 + p);
        for (int c = 0; c < 4; ++c) for (int p = 0; p < 3; ++p) fb[c][p] = frag(o + 6 + c * 3 + p);
#endif
        const int order[6][2] = {{1, 1}, {0, 2}, {0, 1}, {2, 0}, {1, 0}, {0, 0}};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][order[q][0]], fb[c][order[q][1]], acc[a][c], 0, 0, 0);
    }
    if (tid == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) sum += acc[a][c][r];
#else
    // (36 fragments of a K = 32 step do not fit beside 128 accumulator registers at two waves per SIMD: A's planes stay in
    // registers for the step, B's planes come one at a time -- b2 (a0), b1 (a1, a0), b0 (a2, a1, a0): 36 reads, 12 + 8 live)
    f32x4 acc[4][8];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.f;
    bf16x8 fa[4][3], fb[8];
    for (int a = 0; a < 4; ++a) for (int p = 0; p < 3; ++p) fa[a][p] = frag(a * 3 + p + wave);
    for (int c = 0; c < 8; ++c) fb[c] = frag(12 + c + wave);
    if (tid == 0 && blockIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#if READS == 2
    // PAIRED products on K = 16 tiles (what a 256x256 tile's LDS budget allows: K = 32 stages of three planes do not fit twice):
    // one 16x16x32 instruction contracts TWO of the six products of a 16-deep K tile -- lanes 0..31 feed product p, lanes 32..63
    // product p' (a0.b2 | a0.b1, a1.b1 | a1.b0, a2.b0 | a0.b0) -- three instructions per 16x16 block and K tile, each with
    // fragments of its own: 12 + 24 = 36 reads per K = 16 tile, twice the LDS bytes of the K = 32 form.
    for (int s = 0; s < 2 * ksteps; ++s) {
        const int o = 36 * s + 5 * wave;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            bf16x8 pa[4];
            for (int a = 0; a < 4; ++a) pa[a] = frag(o + pr * 12 + a);
            for (int c = 0; c < 8; ++c) fb[c] = frag(o + pr * 12 + 4 + c);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[a], fb[c], acc[a][c], 0, 0, 0);
        }
    }
#else
    for (int s = 0; s < ksteps; ++s) {                // K = 32 per step: 192 products, 36 fragment reads
        const int o = 36 * s + 5 * wave;
#if READS
        for (int a = 0; a < 4; ++a) for (int p = 0; p < 3; ++p) fa[a][p] = frag(o + a * 3 + p);
#endif
#pragma unroll
        for (int pb = 2; pb >= 0; --pb) {
#if READS
            for (int c = 0; c < 8; ++c) fb[c] = frag(o + 12 + pb * 8 + c);
#endif
#pragma unroll
            for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][pa], fb[c], acc[a][c], 0, 0, 0);
        }
    }
#endif
    if (tid == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float sum = 0.f;
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 8; ++c) for (int r = 0; r < 4; ++r) sum += acc[a][c][r];
#endif
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 400, ksteps = argc > 2 ? atoi(argv[2]) : 32;      // 32 steps of K = 32: K = 1024
    hipDeviceProp_t prop{};
    (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 2;       // two rounds of workgroups, as a 512-tile launch
    std::vector<unsigned> h(LDS_BYTES / 4);
    unsigned s = 12345u;
    for (auto& v : h) {         // random bf16 pairs of moderate magnitude (sign, exponent around 1, random mantissa)
        s = s * 1664525u + 1013904223u; const unsigned lo = 0x3f00u | ((s >> 9) & 0x80ffu);
        s = s * 1664525u + 1013904223u; const unsigned hi = 0x3f00u | ((s >> 9) & 0x80ffu);
        v = lo | (hi << 16);
    }
    uint4* fill; float* out; unsigned long long* clk;
    (void)hipMalloc(&fill, LDS_BYTES); (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&clk, 16);
    (void)hipMemcpy(fill, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), LDS_BYTES, 0, fill, out, ksteps, clk);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), LDS_BYTES, 0, fill, out, ksteps, clk);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    // per workgroup and K = 32: 8 waves x 192 products of 16x16x32 (or 96 of 32x32x16) = 8 x 64 x 128 x 32 x 2 x 6 FLOP
    const double flop = (double)grid * 8 * 64 * 128 * 32 * 2 * 6 * ksteps;
    printf("SHAPE %d READS %d: %d launches x %d workgroups, K = %d: %.1f us per launch, %.1f TFLOP/s of bf16 products, loop clock %.3f GHz (%llu cycles, %.1f cycles per K = 32 and wave pair)\n",
           SHAPE, READS, launches, grid, 32 * ksteps, 1e3 * ms / launches, flop / (1e-3 * ms / launches) / 1e12,
           (double)c[0] / (double)c[1] * 0.1, c[0], (double)c[0] / ksteps);
    return 0;
}
