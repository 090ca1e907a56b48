// K-loop probe (round 4; not part of the library): the steady-state loop of csrc/gemm.hip's 128x128 kernel on plain
// operands (no gather, no segments), with in-kernel stamps around prologue / loop / epilogue, at one or two workgroups per
// CU, and with parts of the loop switched off (timing-only builds) -- to find out what a SINGLE wave per SIMD loses.
//   hipcc -O3 --offload-arch=gfx950 -DOCC=2 -DABL=0 kloop_probe.hip -o kloop_probe
// ABL bits: 1 no global loads in the loop, 2 no ds_write, 4 no ds_read, 8 no barrier, 16 mask-style address math,
//           32 no memory instruction interleave (all memory ops ahead of the MFMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef OCC
#define OCC 2
#endif
#ifndef ABL
#define ABL 0
#endif

constexpr int BM = 128, BN = 128, BK = 16, LDW = BK + 4;
constexpr int TILE_FLOATS = 128 * LDW;
constexpr int LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;

__global__ __launch_bounds__(256, OCC) void probe(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                  int M, int N, int K, unsigned long long* stamps, const int* maskp, unsigned long long* samples, int stagger, unsigned* arrivals) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), r_entry = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    if (stagger > 0) {
        // every second workgroup to arrive on a CU waits `stagger` x 64 cycles: its fixed phases then fall into its partner's K loop
        __shared__ unsigned s_arr;
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
            const unsigned cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
            s_arr = atomicAdd(&arrivals[cu], 1u);
        }
        __syncthreads();
        if (s_arr & 1) for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int nbn = N / BN;
    const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int ntiles = K / BK;
    const int r0 = tid >> 2, kc = tid & 3;
    const float* ap[2]; const float* bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ap[i] = A + (long long)(m0 + r0 + 64 * i) * K + 4 * kc;
        bp[i] = Bt + (long long)(n0 + r0 + 64 * i) * K + 4 * kc;
    }
    // mask-style address math as in gemm.hip (three segments; here all three are the same rows, boundaries from memory)
    const int c0 = __builtin_amdgcn_readfirstlane(maskp[0]), c1 = __builtin_amdgcn_readfirstlane(maskp[1]);
    const long long d1_0 = maskp[2 + (tid & 1)], d1_1 = maskp[3], d2_0 = maskp[4], d2_1 = maskp[5];   // all zero at run time
    struct GTile { f32x4 a[2], b[2]; };
    auto load_tile_asm = [&](GTile& gt, int kt) {
        const float *pa0, *pa1, *pb0, *pb1;
        if (ABL & 16) {
            const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
            const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1) + ((int)m1 & c0) + ((int)m2 & c1);
            pa0 = (const float*)((const char*)ap[0] + (d1_0 & m1) + (d2_0 & m2) + (long long)ko * (BK * 4));
            pa1 = (const float*)((const char*)ap[1] + (d1_1 & m1) + (d2_1 & m2) + (long long)ko * (BK * 4));
            pb0 = bp[0] + kt * BK; pb1 = bp[1] + kt * BK;
        } else {
            pa0 = ap[0] + kt * BK; pa1 = ap[1] + kt * BK; pb0 = bp[0] + kt * BK; pb1 = bp[1] + kt * BK;
        }
        if (!(ABL & 1)) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[0]) : "v"(pa0));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[1]) : "v"(pa1));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[0]) : "v"(pb0));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[1]) : "v"(pb1));
        } else {
            asm volatile("" :: "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1));
        }
    };
    auto load_tile = [&](GTile& gt, int kt) {
        gt.a[0] = *reinterpret_cast<const f32x4*>(ap[0] + kt * BK); gt.a[1] = *reinterpret_cast<const f32x4*>(ap[1] + kt * BK);
        gt.b[0] = *reinterpret_cast<const f32x4*>(bp[0] + kt * BK); gt.b[1] = *reinterpret_cast<const f32x4*>(bp[1] + kt * BK);
    };
    auto store_tile = [&](const GTile& gt, int buf) {
        float* sa = smem + buf * 2 * TILE_FLOATS + r0 * LDW + 4 * kc;
        float* sb = sa + TILE_FLOATS;
        *reinterpret_cast<f32x4*>(sa) = gt.a[0]; *reinterpret_cast<f32x4*>(sa + 64 * LDW) = gt.a[1];
        *reinterpret_cast<f32x4*>(sb) = gt.b[0]; *reinterpret_cast<f32x4*>(sb + 64 * LDW) = gt.b[1];
    };
    const int a_off = (wave * 32 + l31) * LDW + 4 * lh;
    const int b_off = TILE_FLOATS + l31 * LDW + 4 * lh;
    struct Frag { f32x4 a[2]; f32x4 b[4][2]; };
    auto read_frags = [&](Frag& f, int buf) {
        const float* base = smem + buf * 2 * TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < 2; ++j) f.a[j] = *reinterpret_cast<const f32x4*>(base + a_off + 8 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) f.b[c][j] = *reinterpret_cast<const f32x4*>(base + b_off + c * 32 * LDW + 8 * j);
    };
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j][i], f.b[c][j][i], acc[c], 0, 0, 0);
    };
    Frag f0, f1;
    GTile g0, g1;
    load_tile(g0, 0); load_tile(g1, 1);
    store_tile(g0, 0); store_tile(g1, 1);
#if ABL & 32
#define SCHED_TILE
#else
#define SCHED_TILE                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 18; ++q_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
#endif
#define TILE_FULL(FC, FN, G, KT)                                                          \
    {                                                                                     \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(G.a[0]), "+v"(G.a[1]), "+v"(G.b[0]), "+v"(G.b[1]));  \
        if (!(ABL & 2)) store_tile(G, (KT) & 1);                                          \
        load_tile_asm(G, (KT) + 4);                                                       \
        if (!(ABL & 4)) read_frags(FN, ((KT) + 1) & 1);                                   \
        else asm volatile("" : "+v"(FN.a[0]), "+v"(FN.a[1]), "+v"(FN.b[0][0]), "+v"(FN.b[1][0]), "+v"(FN.b[2][0]), "+v"(FN.b[3][0])); \
        mma(FC);                                                                          \
        SCHED_TILE                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (!(ABL & 8)) __syncthreads();                                                  \
    }
    load_tile_asm(g0, 2); load_tile_asm(g1, 3);
    __syncthreads();
    read_frags(f0, 0);
    read_frags(f1, 1);
    __syncthreads();
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime(), r_loop0 = __builtin_amdgcn_s_memrealtime();
    int kt = 0;
    for (; kt + 5 < ntiles; kt += 2) {
#ifdef SAMPLES
        if ((kt & 7) == 0 && tid == 0 && (blockIdx.x & 63) == 0 && (kt >> 3) < 64) {
            unsigned long long* sp = samples + ((size_t)(blockIdx.x >> 6) * 64 + (kt >> 3)) * 2;
            sp[0] = __builtin_amdgcn_s_memtime(); sp[1] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        TILE_FULL(f0, f1, g0, kt)
        TILE_FULL(f1, f0, g1, kt + 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.a[0]), "+v"(g0.a[1]), "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g1.a[0]), "+v"(g1.a[1]), "+v"(g1.b[0]), "+v"(g1.b[1]));
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime(), r_loop1 = __builtin_amdgcn_s_memrealtime();
    const int loop_tiles = kt;
    // tail (plain): tiles kt .. ntiles-1; g0 / g1 hold tiles kt+2 / kt+3 (kt is even here)
#define TILE_STEP(FC, FN, G, KT)                                                          \
    {                                                                                     \
        if ((KT) + 2 < ntiles) store_tile(G, (KT) & 1);                                   \
        if ((KT) + 1 < ntiles) read_frags(FN, ((KT) + 1) & 1);                            \
        if ((KT) < ntiles) mma(FC);                                                       \
        __syncthreads();                                                                  \
    }
    for (; kt < ntiles; kt += 2) {
        TILE_STEP(f0, f1, g0, kt)
        TILE_STEP(f1, f0, g1, kt + 1)
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + c * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            C[(long long)m * N + n] = acc[c][r];
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        unsigned long long* s = stamps + (size_t)blockIdx.x * 8;
        s[0] = t_loop0 - t_entry; s[1] = t_loop1 - t_loop0; s[2] = t_end - t_loop1; s[3] = r_loop1 - r_loop0; s[4] = loop_tiles;
        s[5] = r_entry; s[6] = __builtin_amdgcn_s_memrealtime();
        { const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
          s[7] = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = 8192, N = 2048, K = argc > 1 ? atoi(argv[1]) : 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    float *dA, *dB, *dC; unsigned long long* dS; int* dMask;
    const int blocks = (M / BM) * (N / BN);
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dS, (size_t)blocks * 64)); CK(hipMalloc(&dMask, 64));
    unsigned long long* dSamp; CK(hipMalloc(&dSamp, 16 * 64 * 16)); CK(hipMemset(dSamp, 0, 16 * 64 * 16));
    const int iters = argc > 2 ? atoi(argv[2]) : 40;
    const int stagger = argc > 4 ? atoi(argv[4]) : 0;
    unsigned* dArr; CK(hipMalloc(&dArr, 4096 * 4)); CK(hipMemset(dArr, 0, 4096 * 4));
    int hmask[8] = {1 << 28, 1 << 29, 0, 0, 0, 0, 0, 0};
    CK(hipMemcpy(dMask, hmask, sizeof(hmask), hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const int lds = OCC == 1 ? 96 * 1024 : LDS_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds, 0, dA, dB, dC, M, N, K, dS, dMask, dSamp, stagger, dArr);
    CK(hipDeviceSynchronize());
    double maxerr = 0;
    if (!(ABL & 15)) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < 2000; ++t) {
            const int m = rand() % M, n = rand() % N;
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(s - hC[(size_t)m * N + n]));
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds, 0, dA, dB, dC, M, N, K, dS, dMask, dSamp, stagger, dArr);
    CK(hipEventRecord(e0));
    const int sync_every = argc > 3 ? atoi(argv[3]) : 0;     // a host round trip every so many launches (0: none)
    for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds, 0, dA, dB, dC, M, N, K, dS, dMask, dSamp, stagger, dArr);
        if (sync_every && i % sync_every == sync_every - 1) { int dummy; CK(hipMemcpy(&dummy, dMask, 4, hipMemcpyDeviceToHost)); }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    std::vector<unsigned long long> hs((size_t)blocks * 8);
    CK(hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> pro, loop, epi, clk;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int b = 0; b < blocks; ++b) {
        const unsigned long long* s = &hs[(size_t)b * 8];
        pro.push_back((double)s[0]); loop.push_back((double)s[1] / (double)s[4]); epi.push_back((double)s[2]);
        clk.push_back((double)s[1] / (double)s[3] * 0.1);    // cycles per 10 ns -> GHz
        rmin = std::min(rmin, s[5]); rmax = std::max(rmax, s[6]);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double cyc_tile = med(loop), ghz = med(clk);
    printf("OCC=%d ABL=%d K=%d iters=%d sync=%d stagger=%d: %.1f us/launch %.1f TFLOP/s | err %.2g | prologue %.0f cyc, loop %.0f cyc/tile (pipe %.1f%% busy), epilogue %.0f cyc, clock %.2f GHz, kernel span %.1f us\n",
           OCC, ABL, K, iters, sync_every, stagger, us, 2.0 * M * N * K / us / 1e6, maxerr, med(pro), cyc_tile, 100.0 * 2048.0 * OCC / cyc_tile, med(epi), ghz,
           (double)(rmax - rmin) * 0.01);
    if (argc > 5) {     // timeline of the last launch: when workgroups start and end, how many run at once, per-CU slot use
        std::vector<std::pair<double, int>> ev;
        std::vector<double> st, en;
        for (int b = 0; b < blocks; ++b) {
            const unsigned long long* s = &hs[(size_t)b * 8];
            st.push_back((double)(s[5] - rmin) * 0.01); en.push_back((double)(s[6] - rmin) * 0.01);
            ev.push_back({st.back(), +1}); ev.push_back({en.back(), -1});
        }
        std::sort(ev.begin(), ev.end());
        printf("  timeline (us: running workgroups):");
        int run = 0; double next = 0;
        for (auto& e : ev) { run += e.second; if (e.first >= next) { printf(" %.0f:%d", e.first, run); next = e.first + 10.0; } }
        printf("\n  starts by block id (us):");
        for (int b = 0; b < blocks; b += 64) printf(" %d:%.1f", b, st[b]);
        printf("\n  durations by block id (us):");
        for (int b = 0; b < blocks; b += 64) printf(" %d:%.1f", b, en[b] - st[b]);
        std::vector<int> percu(4096, 0);
        for (int b = 0; b < blocks; ++b) percu[hs[(size_t)b * 8 + 7] & 4095]++;
        int cus = 0, mn = 1 << 30, mx = 0;
        for (int c : percu) if (c) { ++cus; mn = std::min(mn, c); mx = std::max(mx, c); }
        printf("\n  %d CUs used, workgroups per CU min %d max %d\n", cus, mn, mx);
    }
#ifdef SAMPLES
    std::vector<unsigned long long> hp(16 * 64 * 2);
    CK(hipMemcpy(hp.data(), dSamp, hp.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < 16; b += 5) {
        printf("  block %4d clock per 8 tiles (GHz):", b * 64);
        for (int i = 0; i + 1 < 64 && hp[(b * 64 + i + 1) * 2]; ++i)
            printf(" %.2f", (double)(hp[(b * 64 + i + 1) * 2] - hp[(b * 64 + i) * 2]) / (double)(hp[(b * 64 + i + 1) * 2 + 1] - hp[(b * 64 + i) * 2 + 1]) * 0.1);
        printf("  | start at %.1f us\n", (double)(hp[(b * 64) * 2 + 1] - rmin) * 0.01);
    }
#endif
    return maxerr < 1e-2 ? 0 : 2;
}
