// fp32 MFMA GEMM for operands that lie K-MAJOR in memory:
//
//   C[M][N] (+)= sum_k A[k][m] * B[k][n]        A: [K][lda], m contiguous;  B: [K][ldb], n contiguous
//
// The weight gradients of the train step (keras_train.py:195, backward of seq2seq.py:237-390) are exactly this shape:
// dW[4W][kin] = dZ^T . X with dZ [rows][4W] and X [rows][kin] as the forward pass left them, the contraction running
// over the rows = time x batch (51 712 at BASELINE configs[3]).  gemm.hip wants both operands K-contiguous, which cost a
// transposition pass per operand plus the clearing of its padded scratch (rounds 1-2: transpose_kernel 3.9 %,
// fillBufferAligned 1.4 % of the step); here the tiles are taken as they lie.
//
// Tiling: 128 x 128 block tile, BK = 16, 256 threads = 4 waves x (32 rows x 128 columns), four
// v_mfma_f32_32x32x2_f32 accumulators per wave.  A k-row of a tile is 512 contiguous bytes: 32 lanes x 16 B per load
// instruction; the LDS image is [k][m] (ds_write_b128 of whole lines, conflict-free), and a lane's MFMA operand
// A[m = lane & 31][k = 2 j + (lane >> 5)] is ONE ds_read_b32 whose 32-lane halves read 32 consecutive words each
// (conflict-free without padding).  40 ds_read_b32 per 32 MFMAs and wave: 31 % of the LDS rate with eight waves per CU.
// Long K is split over blockIdx.z with float atomics into C (the sums are weight gradients: order-tolerant by nature,
// compared with the oracle to 2e-3 of their largest entry).
#include "common.h"

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CASV_TN_XCD_ORDER
#define CASV_TN_XCD_ORDER 1         // 0: tile index fastest (the order of rounds 3-4; A/B builds)
#endif
constexpr int TBM = 128, TBN = 128, TBK = 16;
constexpr int T_TILE = TBK * 128;                       // floats of one operand tile in LDS ([k][128])

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * T_TILE];     // 32 KB: two buffers x (A tile, B tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nbn = (g.N + TBN - 1) / TBN;
    // Workgroups are dealt round-robin over the 8 XCDs (private L2s) by their linear index.  K share fastest: the workgroups of ONE
    // K share -- every tile of the output over the same k-rows -- then sit on as few XCDs as the share count allows (one, when it
    // is a multiple of 8), so a k-row of an operand is fetched into ONE L2 and hit there by all the tiles that need it.  With the
    // tile index fastest every XCD held tiles of every K share: 41.6 GB per train step at the memory side for ~11 GB of operands
    // (profiles/r05_train_pmc.txt).  Placement changes traffic and speed only.
    const int nsplit = g.nsplit;
    const int tile = CASV_TN_XCD_ORDER ? (int)blockIdx.x / nsplit : (int)blockIdx.x % (gridDim.x / nsplit);
    const int zidx = CASV_TN_XCD_ORDER ? (int)blockIdx.x % nsplit : (int)blockIdx.x / (gridDim.x / nsplit);
    const int bn = tile % nbn, bm = tile / nbn;
    const int m0 = bm * TBM, n0 = bn * TBN;
    const int ktiles_all = (g.K + TBK - 1) / TBK;
    const int per = (ktiles_all + nsplit - 1) / nsplit;
    const int kt_begin = zidx * per;
    const int ntiles = ktiles_all - kt_begin < per ? (ktiles_all - kt_begin > 0 ? ktiles_all - kt_begin : 0) : per;
    if (ntiles <= 0) return;

    // staging role: k-rows sk and sk + 8 of a tile, 4 consecutive m (n) at column sc; columns past the operand's width read
    // column 0 instead (their products land in output elements that are never stored)
    const int sk = tid >> 5, sc = 4 * (tid & 31);
    const int am = (m0 + sc < g.M) ? m0 + sc : 0, bnn = (n0 + sc < g.N) ? n0 + sc : 0;
    const float* ap = g.A + am;
    const float* bp = g.B + bnn;
    struct GTile { f32x4 a[2], b[2]; };
    // whole tiles: plain loads, nothing touches the data until it goes to LDS two tiles later (a select on a loaded value
    // would make the compiler wait for the load where it is issued)
    auto load_tile = [&](GTile& gt, int kt_rel) {
        const long long k0 = (long long)(kt_begin + kt_rel) * TBK + sk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gt.a[i] = *reinterpret_cast<const f32x4*>(ap + (k0 + 8 * i) * g.lda);
            gt.b[i] = *reinterpret_cast<const f32x4*>(bp + (k0 + 8 * i) * g.ldb);
        }
    };
    // The same loads hidden from the compiler's wait bookkeeping (steady state only).  hipcc waits for a tile's loads with a
    // counter that is in order: in front of the LDS store of tile kt + 2 it also drains the loads of tile kt + 3 issued a
    // moment ago, so the prefetch is one tile deep -- less than the memory latency of operands that stream from HBM.  Issued
    // from asm statements the loads are invisible to that bookkeeping, and CASV_TN_FULL waits with a counted vmcnt(4): the four
    // loads of the tile it is about to store have landed, the four of the next tile stay in flight.
    // (running pointers: one 64-bit add per load and tile instead of a 64-bit multiply-add per load -- gemm.hip, round 4)
    const float* ra[2] = {nullptr, nullptr}; const float* rb[2] = {nullptr, nullptr};
    const long long astep = (long long)TBK * g.lda, bstep = (long long)TBK * g.ldb;
    auto run_set = [&](int kt_rel) {
        const long long k0 = (long long)(kt_begin + kt_rel) * TBK + sk;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ra[i] = ap + (k0 + 8 * i) * g.lda; rb[i] = bp + (k0 + 8 * i) * g.ldb; }
    };
    auto load_tile_asm = [&](GTile& gt, int kt_rel) {
        (void)kt_rel;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[i]) : "v"(ra[i]));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[i]) : "v"(rb[i]));
            ra[i] += astep; rb[i] += bstep;
        }
    };
    // the last tile of the K range may be partial: rows k >= K enter as zeros
    auto load_tile_tail = [&](GTile& gt, int kt_rel) {
        const int k0 = (kt_begin + kt_rel) * TBK + sk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + 8 * i;
            const bool in = k < g.K;
            const long long kk = in ? k : 0;
            const f32x4 va = *reinterpret_cast<const f32x4*>(ap + kk * g.lda);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(bp + kk * g.ldb);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            gt.a[i] = in ? va : zero; gt.b[i] = in ? vb : zero;
        }
    };
    // tiles of this workgroup that are whole: all but possibly the last one of the whole K range
    const int nwhole = ((kt_begin + ntiles) * TBK <= g.K) ? ntiles : ntiles - 1;
    auto load_any = [&](GTile& gt, int kt_rel) { if (kt_rel < nwhole) load_tile(gt, kt_rel); else load_tile_tail(gt, kt_rel); };
    // Column sums of A over k (the bias gradient db = sum_rows dZ of the same layer): the first column tile of every row
    // tile adds up the A values it stages anyway -- the separate pass over dZ (424 MB per layer at configs[3]) goes away.
    const bool do_colsum = g.colsum != nullptr && bn == 0;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&](const GTile& gt, int buf) {
        float* sa = smem + buf * 2 * T_TILE + sk * 128 + sc;
        if (do_colsum) csum += gt.a[0] + gt.a[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(sa + i * 8 * 128) = gt.a[i];
            *reinterpret_cast<f32x4*>(sa + T_TILE + i * 8 * 128) = gt.b[i];
        }
    };
    struct Frag { float a[8]; float b[4][8]; };
    const int a_off = lh * 128 + wave * 32 + l31, b_off = T_TILE + lh * 128 + l31;
    auto read_frags = [&](Frag& f, int buf) {
        const float* base = smem + buf * 2 * T_TILE;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.a[j] = base[a_off + j * 256];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) f.b[c][j] = base[b_off + j * 256 + c * 32];
    };
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j], f.b[c][j], acc[c], 0, 0, 0);
    };

    // The software pipeline of gemm.hip: while the 32 MFMAs of tile kt issue from F[kt & 1], F[(kt + 1) & 1] is read from
    // LDS[(kt + 1) & 1], tile kt + 2 goes from registers into LDS[kt & 1] (whose content sits in F[kt & 1]) and tile kt + 4 is
    // requested from memory; one barrier per tile.
    Frag f0, f1;
    GTile g0, g1;
    load_any(g0, 0);
    if (ntiles > 1) load_any(g1, 1);
    store_tile(g0, 0);
    if (ntiles > 1) store_tile(g1, 1);
#define CASV_TN_STEP(FC, FN, G, KT)                                                       \
    {                                                                                     \
        if ((KT) + 2 < ntiles) store_tile(G, (KT) & 1);                                   \
        if ((KT) + 4 < ntiles) load_any(G, (KT) + 4);                                     \
        if ((KT) + 1 < ntiles) read_frags(FN, ((KT) + 1) & 1);                            \
        if ((KT) < ntiles) mma(FC);                                                       \
        __syncthreads();                                                                  \
    }
#define CASV_TN_FULL(FC, FN, G, KT)                                                       \
    {                                                                                     \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(G.a[0]), "+v"(G.a[1]), "+v"(G.b[0]), "+v"(G.b[1]));  \
        store_tile(G, (KT) & 1);                                                          \
        load_tile_asm(G, (KT) + 4);                                                       \
        read_frags(FN, ((KT) + 1) & 1);                                                   \
        mma(FC);                                                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x320, 2, 0);                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
        __syncthreads();                                                                  \
    }
    int kt = 0;
    if (nwhole > 5) {
        // Steady state.  Tiles 2 and 3 are requested the hidden way already: a compiler-tracked load pending on ANY path into
        // the loop would put a full vmcnt(0) at the loop head, executed in every iteration.
        run_set(2);
        load_tile_asm(g0, 2); load_tile_asm(g1, 3);
        __syncthreads();
        read_frags(f0, 0);
        __syncthreads();
        for (; kt + 5 < nwhole; kt += 2) {          // the tiles requested here (kt + 4, kt + 5) are whole
            CASV_TN_FULL(f0, f1, g0, kt)
            CASV_TN_FULL(f1, f0, g1, kt + 1)
        }
        // loads issued by the asm statements are still in flight: they land before the compiler-scheduled rest touches them
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.a[0]), "+v"(g0.a[1]), "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g1.a[0]), "+v"(g1.a[1]), "+v"(g1.b[0]), "+v"(g1.b[1]));
    } else {
        if (ntiles > 2) load_any(g0, 2);
        if (ntiles > 3) load_any(g1, 3);
        __syncthreads();
        read_frags(f0, 0);
        __syncthreads();
    }
    for (; kt + 1 < ntiles; kt += 2) {
        CASV_TN_STEP(f0, f1, g0, kt)
        CASV_TN_STEP(f1, f0, g1, kt + 1)
    }
    if (kt < ntiles) CASV_TN_STEP(f0, f1, g0, kt)
#undef CASV_TN_STEP
#undef CASV_TN_FULL

    if (do_colsum) {            // (every wave is past the K loop's last barrier: the staging buffers are free)
        *reinterpret_cast<f32x4*>(smem + sk * 128 + sc) = csum;
        __syncthreads();
        if (tid < 128 && m0 + tid < g.Mstore) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += smem[q * 128 + tid];
            atomicAdd(g.colsum + m0 + tid, t);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + c * 32 + l31;
        if (n < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.Mstore) {
                    float* dst = g.C + (long long)m * g.ldc + n;
                    if (nsplit > 1) atomicAdd(dst, acc[c][r]);
                    else *dst = g.accumulate ? (*dst + acc[c][r]) : acc[c][r];
                }
            }
        }
    }
}

// `accumulate` = 0 with a split K needs a cleared C: the caller says so (out_zeroed) or gets a memset here.
void launch_gemm_tn(const TnArgs& g, hipStream_t stream) {
    const int nbm = (g.M + TBM - 1) / TBM, nbn = (g.N + TBN - 1) / TBN;
    const int tiles = nbm * nbn, ktiles = (g.K + TBK - 1) / TBK;
    // fill the chip (512 workgroup slots), keep >= 64 k-tiles per workgroup so that the atomic epilogue stays a small part
    int ks = (512 + tiles - 1) / tiles;
    if (ks > ktiles / 64) ks = ktiles / 64;
    if (ks < 1) ks = 1;
    if (ks > 1 && !g.accumulate && !g.out_zeroed) {
        if (g.ldc == g.N) (void)hipMemsetAsync(g.C, 0, (size_t)g.Mstore * g.N * sizeof(float), stream);
        else (void)hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.Mstore, stream);
    }
    TnArgs gg = g;
    gg.nsplit = ks;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * ks), dim3(256), 0, stream, gg);
}

}  // namespace casv
