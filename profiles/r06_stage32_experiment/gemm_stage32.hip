// fp32 MFMA GEMM for gfx950 with an optional fused LSTM-cell epilogue.
//
//   PLAIN:  C[M][N]  = A[M][K] . Bt[N][K]^T (+ bias)
//   LSTM:   z = [x | ctx | h][M][K] . Wt[4U][K]^T + b ; (h', c') = cell(z, c)     (U = units)
//
// The A operand is the concatenation of up to three row-gathered K-segments (layer input, attention
// context, recurrent state), so "x.K + h.R" of a Keras LSTMCell (seq2seq.py:272,337,344) is ONE
// contraction and the gate pre-activations never leave the accumulators.  Weight rows are stored
// gate-interleaved in blocks of 32 units ([unit/32][gate i,f,c,o][unit%32]) so that a 128-column block
// tile holds all four gates of 32 units and a wave's four 32x32 accumulator tiles hold, register for
// register, the i/f/c/o pre-activations of the same (row, unit).
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 waves, each wave 32 rows x 128 columns
// (4 x v_mfma_f32_32x32x2_f32 accumulators, exact fp32 = a k-ordered fmaf chain).  Operands are staged
// K-contiguous in LDS with a +4-float row pad (80-B rows: ds_read_b128 is conflict-free for its 16-lane
// groups) and double-buffered; each lane reads four consecutive k per ds_read_b128 and feeds them to
// four MFMAs (the k order inside a tile is permuted identically for A and B, which a sum allows).
// Small launches take the 32x128-tile variant of gemm_skinny.hip instead (plan_gemm; same values bit for bit).
#include "common.h"
#include <cstdlib>
#include <cstdio>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build (-DCASV_GEMM_PROF): thread 0 of every workgroup of the 128x128 LSTM kernel adds its shader-clock cycles
// in prologue / steady-state loop / epilogue, the loop's tile count and its wall-clock ticks (10 ns) to g_gemm_prof;
// gemm_prof_dump() prints and clears them (engine.hip calls it after a beamed decode).
#ifdef CASV_GEMM_PROF
__device__ unsigned long long g_gemm_prof[16 + 64];
#define CASV_STAMP(x) { asm volatile("" ::: "memory"); x = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); }
void gemm_prof_dump() {
    unsigned long long h[16 + 64];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_prof), sizeof(h));
    if (h[5]) {
        fprintf(stderr, "gemm_prof workgroup cycles, histogram in bins of 16384:");
        for (int i = 0; i < 64; ++i) if (h[16 + i]) fprintf(stderr, " %d:%llu", i, h[16 + i]);
        fprintf(stderr, "\n");
    }
    if (h[5]) fprintf(stderr, "gemm_prof prologue parts: setup %.0f, cell-state addresses %.0f, tiles 0/1 requested %.0f, landed+stored+tiles 2/3 requested+frags %.0f | tail tiles %.0f, epilogue %.0f cyc\n",
                      (double)h[8] / h[5], (double)h[9] / h[5], (double)h[10] / h[5], (double)h[11] / h[5], (double)h[12] / h[5], (double)h[13] / h[5]);
    if (h[5])
        fprintf(stderr, "gemm_prof: %llu workgroups: prologue %.0f cyc, loop %.1f cyc/tile over %.1f tiles, tail+epilogue %.0f cyc, "
                "whole workgroup %.0f cyc, loop clock %.3f GHz\n", h[5], (double)h[0] / h[5], (double)h[1] / (double)h[2], (double)h[2] / h[5],
                (double)h[3] / h[5], (double)h[6] / h[5], (double)h[1] / (double)h[4] * 0.1);
    unsigned long long z[16 + 64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z));
}
#endif

#ifndef CASV_PLAIN_WIDE
#define CASV_PLAIN_WIDE 1           // the PLAIN kernel's full-tile epilogue stores 16 bytes per lane through an in-quad transpose (0: A/B builds)
#endif

constexpr int BM = 128, BN = 128, BK = 16, LDW = BK + 4;   // 80-B LDS rows: ds_read_b128 conflict-free
constexpr int TILE_FLOATS = 128 * LDW;               // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_FLOATS * 4;
constexpr int CELL_LDS_BYTES = 128 * 128;            // LSTM epilogue: the tile's previous cell state, [half][row][16 floats]
constexpr int STASH_LDS_BYTES = 256 * 6 * 8;         // per thread: row pointers that wait for their turn outside the register file
// SPLIT variant (option split_bf16, an experiment -- see the comment in front of gemm_tile): an operand tile lies in LDS as three
// bf16 planes of 128 rows x 16 k (32-byte rows, the two 16-byte halves of a row swapped where bit 4 of the row is set: conflict-free
// for ds_read_b128's 16-lane groups without padding)
constexpr int SPLIT_PLANE_BYTES = 128 * 32;
constexpr int SPLIT_BUF_FLOATS = 6 * SPLIT_PLANE_BYTES / 4;     // A planes 0..2, B planes 0..2

// Software pipeline (per wave, so that ONE wave keeps its SIMD's matrix pipe busy and the two waves
// of a SIMD do not have to be out of phase to cover each other):
//   while the 32 MFMAs of tile kt run from fragment registers F[kt&1],
//     F[(kt+1)&1] <- LDS[(kt+1)&1]   (10 ds_read_b128; that buffer was filled during kt-1)
//     LDS[kt&1]   <- G               (tile kt+2, global loads issued during kt-1; the buffer's old
//                                     content, tile kt, already sits in F[kt&1])
//     G           <- global tile kt+3
//   one barrier per tile.
// KS = 2 (train step only): two 4-wave groups per workgroup take alternate K-tiles of the SAME output tile and add
// their accumulators through LDS before the epilogue -- a deterministic split-K that halves the critical path of the
// small-M recurrent GEMMs (M = 512 gives only 64 workgroups).  Inference keeps KS = 1 so that a row's sum order never
// depends on the batch it sits in.
// Tile (bm, bn) of job g by the calling workgroup; split-K part zidx of nsplit.
//
// SPLIT (KS = 1 only; option split_bf16, a measured experiment and NOT the default arithmetic): every fp32 operand value is taken
// apart into three bf16 values x = x0 + x1 + x2 (round to nearest, each remainder exact in fp32, so the sum is exact) while its tile
// is staged into LDS, and a K tile is contracted as six v_mfma_f32_32x32x16_bf16 products per 32x32 block with fp32 accumulation --
// a1.b1, a0.b2, a0.b1, a2.b0, a1.b0, a0.b0 (the order of gemm_split.hip's 256x256 tiles: an element's sum is the same instruction
// sequence in both, so the two tile shapes give the same bits); the three dropped products are below 2^-25 |a||b| -- at 6/16 of the matrix-pipe time of
// the fp32-input instruction.  The sums are fp32-accurate but NOT the k-ordered fmaf chain of the other kernels: results agree with
// them to rounding, not bit for bit.  Fragment registers roll instead of being double-buffered: a plane's fragments of tile kt + 1
// are read into the registers of tile kt as soon as their last product has issued (B planes 2 and 1 and A plane 0 behind the
// tile's barrier, B plane 0 at the head of the next tile), so one barrier per tile still separates every LDS buffer's reads from
// the stores that refill it.
template <int EPI, int KS, bool SPLIT = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bm, const int bn, const int zidx, const int nsplit, float* smem_all) {
#ifdef CASV_GEMM_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
    unsigned long long pt1 = pt0, pt2 = pt0, pr1 = 0, pr2 = 0, pa = pt0, pb = pt0, pc = pt0, pe = pt0; int ptiles = 0;
#endif
    const int grp = KS > 1 ? (threadIdx.x >> 8) : 0;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    static_assert(!SPLIT || KS == 1, "the bf16-split variant has no wave-group split-K");
    constexpr int BUF_FLOATS = SPLIT ? SPLIT_BUF_FLOATS : 2 * TILE_FLOATS;      // one LDS buffer: an A tile and a B tile
    float* smem = smem_all + grp * (2 * BUF_FLOATS);
    const int l31 = lane & 31, lh = lane >> 5;
    // wave-uniform by construction; say so (the value arrives through a vector load)
    const int step = __builtin_amdgcn_readfirstlane(g.step_ptr ? *g.step_ptr : g.step_imm);
    const int nbn = (g.N + BN - 1) / BN;
    const int m0 = bm * BM, n0 = bn * BN;
    if (m0 >= g.M || bm * nbn + bn >= ((g.M + BM - 1) / BM) * nbn) return;
    if (g.nact) {           // every wave looks at the same counts: a uniform exit ahead of the first barrier
        const int mlast = (m0 + BM < g.M ? m0 + BM : g.M) - 1;
        const int l0 = m0 / g.nact_group, l1 = mlast / g.nact_group;
        int alive = 0;
        for (int l = l0 + lane; l <= l1; l += 64) alive |= g.nact[l] > (l == l0 ? m0 - l0 * g.nact_group : 0);
        if (!__any(alive)) return;
    }

    const int r0 = tid >> 2, kc = tid & 3;     // staging: rows r0, r0+64; floats [4kc, 4kc+4)

    // Operand rows of the (up to three) K segments.  The row indices of every gathered segment -- and of the cell state, below --
    // are requested TOGETHER and without branches (a segment without an index array reads a dummy word): one memory round trip
    // for all of them instead of one per segment, each behind its own branch and full wait (round 4, in-kernel stamps: 7 800
    // cycles of set-up in front of the first tile request).
    const float* ap0[2]; const float* ap1[2]; const float* ap2[2];
    int tiles0 = 0, tiles1 = 0, tiles2 = 0;
    int mrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int m = m0 + r0 + 64 * i; mrow[i] = m < g.M ? m : g.M - 1; }
    const int* const no_rows = reinterpret_cast<const int*>(g.Bt);          // any readable word
    // (the segment descriptors as local copies, fetched from the kernel-argument segment in ONE batch of scalar loads: read
    // field by field behind the short-circuit conditions below they were two dozen dependent scalar round trips)
    const Seg sg0 = g.a[0], sg1 = g.a[1], sg2 = g.a[2], sgc = g.c_in;
    const int nseg = g.nseg;
    asm volatile("" :: "s"(sg0.base), "s"(sg0.rows), "s"(sg0.first_base), "s"(sg1.base), "s"(sg1.rows), "s"(sg1.first_base),
                 "s"(sg2.base), "s"(sg2.rows), "s"(sg2.first_base), "s"(sgc.base), "s"(sgc.rows), "s"(sgc.first_base));
    const Seg* const sgs[3] = {&sg0, &sg1, &sg2};
    int ridx[3][2];
    bool act[3], gat[3];
#pragma unroll
    for (int S = 0; S < 3; ++S) {
        const Seg& sg = *sgs[S];
        act[S] = nseg > S && !(sg.skip_first && step == 0 && !sg.first_base);
        gat[S] = act[S] && sg.rows && !(sg.first_base && step == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) ridx[S][i] = *(gat[S] ? sg.rows + mrow[i] : no_rows);
    }
    const bool cell_lane = EPI == EPI_LSTM && grp == 0 && !g.epi_plain;
    const bool cfirst = cell_lane && sgc.first_base && step == 0;
    const bool czero = cell_lane && sgc.skip_first && step == 0 && !cfirst;
    const bool cstage = cell_lane && !czero;
    int crow[2] = {0, 0};
    const float* cin = !cstage ? nullptr : cfirst ? sgc.first_base
        : sgc.base + (long long)(step * sgc.step_mul + sgc.step_add) * sgc.slot_stride;
    {
        const bool cgat = cstage && sgc.rows && !cfirst;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int ci = *(cgat ? sgc.rows + mrow[i] : no_rows); crow[i] = cgat ? ci : mrow[i]; }
    }
    auto cptr = [&](int i) { return cin + (long long)crow[i] * sgc.ld + bn * 32 + 4 * kc; };
#define CASV_SETUP_SEG(S, AP, TILES)                                                             \
    if (act[S]) {                                                                                \
        const Seg& sg = *sgs[S];                                                                 \
        const bool first = sg.first_base && step == 0;                                           \
        const float* base = first ? sg.first_base                                                \
            : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                          \
            const int rid = gat[S] ? ridx[S][i] : mrow[i];                                       \
            AP[i] = base + (long long)rid * sg.ld + 4 * kc;                                      \
        }                                                                                        \
        TILES = sg.width / BK;                                                                   \
    } else {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) AP[i] = nullptr;                           \
    }
    CASV_SETUP_SEG(0, ap0, tiles0)
    CASV_SETUP_SEG(1, ap1, tiles1)
    CASV_SETUP_SEG(2, ap2, tiles2)
#undef CASV_SETUP_SEG
    const int c0 = __builtin_amdgcn_readfirstlane(tiles0), c1 = __builtin_amdgcn_readfirstlane(tiles0 + tiles1);
    const int ntiles_all = __builtin_amdgcn_readfirstlane(tiles0 + tiles1 + tiles2);
    // split-K: this block contracts k-tiles [kt_begin, kt_begin + ntiles)
    const int per = (ntiles_all + nsplit - 1) / nsplit;
    const int kt_begin = zidx * per;
    const int nt_blk = ntiles_all - kt_begin < per ? (ntiles_all - kt_begin > 0 ? ntiles_all - kt_begin : 0) : per;
    // this wave group's share: tiles kt_begin + KS*i + grp
    const int ntiles = (nt_blk - grp + KS - 1) / KS, nt_min = nt_blk / KS, nt_max = (nt_blk + KS - 1) / KS;
    const int koff0 = sg0.koff, koff1 = sg1.koff, koff2 = sg2.koff;

    const float* bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int n = n0 + r0 + 64 * i; n = n < g.N ? n : g.N - 1;
        bp[i] = g.Bt + (long long)n * g.Ktot + 4 * kc;
    }

    // Segment selection without control flow: pointer = ap0 + mask1*(ap1-ap0) + mask2*(ap2-ap0) with wave-uniform
    // 0/-1 masks, so that the steady-state loop body stays ONE basic block (the scheduler interleaves only inside one).
    const long long d1_0 = (long long)((const char*)ap1[0] - (const char*)ap0[0]), d1_1 = (long long)((const char*)ap1[1] - (const char*)ap0[1]);
    const long long d2_0 = (long long)((const char*)ap2[0] - (const char*)ap0[0]), d2_1 = (long long)((const char*)ap2[1] - (const char*)ap0[1]);
    struct GTile { f32x4 a[2], b[2]; };
    auto load_tile = [&](GTile& gt, int kt_rel) {
        const int kt = kt_rel * KS + grp + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);                       // tile index inside its segment
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * BK;
        const char* pa0 = (const char*)ap0[0] + (d1_0 & m1) + (d2_0 & m2) + (long long)ko * (BK * 4);
        const char* pa1 = (const char*)ap0[1] + (d1_1 & m1) + (d2_1 & m2) + (long long)ko * (BK * 4);
        gt.a[0] = *reinterpret_cast<const f32x4*>(pa0); gt.a[1] = *reinterpret_cast<const f32x4*>(pa1);
        gt.b[0] = *reinterpret_cast<const f32x4*>(bp[0] + kb); gt.b[1] = *reinterpret_cast<const f32x4*>(bp[1] + kb);
    };
    // The first tiles of the prologue when they all lie in segment 0 (every launch of the decoder and encoder: the layer input
    // comes first and is at least four tiles wide): addressed without the segment masks, i.e. without waiting for the row
    // indices of the gathered segments behind it (the parents' h rows) -- one dependent memory round trip less in front of
    // the first MFMA of every workgroup.
    auto load_tile_seg0 = [&](GTile& gt, int kt) {
        gt.a[0] = *reinterpret_cast<const f32x4*>(ap0[0] + kt * BK); gt.a[1] = *reinterpret_cast<const f32x4*>(ap0[1] + kt * BK);
        gt.b[0] = *reinterpret_cast<const f32x4*>(bp[0] + koff0 + kt * BK); gt.b[1] = *reinterpret_cast<const f32x4*>(bp[1] + koff0 + kt * BK);
    };
    // The same four loads hidden from the compiler's wait bookkeeping (steady state only).  hipcc's s_waitcnt in front of
    // the LDS store of tile kt + 2 counts in order and drains the loads of tile kt + 3 with it (issued one tile earlier):
    // the prefetch was one tile deep for every second tile.  Issued from asm statements the loads are invisible to that
    // bookkeeping; CASV_TILE_FULL waits with a counted vmcnt(4) instead -- the four loads of the tile about to be stored
    // have landed, the four of the next tile stay in flight -- so every tile's operands have two tile times to arrive.
    auto issue_tile_asm = [&](GTile& gt, const char* pa0, const char* pa1, const float* pb0, const float* pb1) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[0]) : "v"(pa0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.a[1]) : "v"(pa1));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[0]) : "v"(pb0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gt.b[1]) : "v"(pb1));
    };
    auto load_tile_asm = [&](GTile& gt, int kt_rel) {
        const int kt = kt_rel * KS + grp + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * BK;
        const char* pa0 = (const char*)ap0[0] + (d1_0 & m1) + (d2_0 & m2) + (long long)ko * (BK * 4);
        const char* pa1 = (const char*)ap0[1] + (d1_1 & m1) + (d2_1 & m2) + (long long)ko * (BK * 4);
        issue_tile_asm(gt, pa0, pa1, bp[0] + kb, bp[1] + kb);
    };
    // Steady state of the KS = 1 kernels: RUNNING operand pointers, advanced by one K tile per request (four 64-bit adds), and
    // re-based where the request stream crosses into the next K segment -- a wave-uniform branch taken twice per K loop at most.
    // The mask arithmetic above costs ~24 vector and ~20 scalar instructions per tile; a wave that has the matrix pipe to itself
    // (its partner workgroup in prologue / epilogue: almost half of a decoder launch) issues them between its own MFMAs:
    // 2317 instead of 2189 cycles per tile in profiles/kloop_probe.hip (ABL = 16 against 0, one workgroup per CU).
    const char* ra0 = nullptr; const char* ra1 = nullptr; const float* rb0 = nullptr; const float* rb1 = nullptr;
    int rleft = 0, rseg = 0;
    // (the gathered segments' row pointers -- and the cell state's -- wait in LDS for their turn instead of in a dozen registers
    // through the whole loop; the B rows are K-contiguous: a scalar step re-bases them)
    const float** const rstash = reinterpret_cast<const float**>(smem_all + KS * (2 * BUF_FLOATS) + CELL_LDS_BYTES / 4) + 6 * tid;
    auto run_set = [&](int kt_rel) {                       // position the running pointers at tile kt_rel
        const int kt = kt_rel * KS + grp + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * BK;
        ra0 = (const char*)ap0[0] + (d1_0 & m1) + (d2_0 & m2) + (long long)ko * (BK * 4);
        ra1 = (const char*)ap0[1] + (d1_1 & m1) + (d2_1 & m2) + (long long)ko * (BK * 4);
        rb0 = bp[0] + kb; rb1 = bp[1] + kb;
        rseg = kt >= c1 ? 2 : kt >= c0 ? 1 : 0;
        rleft = (rseg == 0 ? c0 : rseg == 1 ? c1 : ntiles_all) - kt;
        rstash[0] = ap1[0]; rstash[1] = ap1[1]; rstash[2] = ap2[0]; rstash[3] = ap2[1];
        if (EPI == EPI_LSTM) { rstash[4] = cstage ? cptr(0) : bp[0]; rstash[5] = cstage ? cptr(1) : bp[0]; }   // (no cell state: any valid address)
    };
    auto run_advance = [&]() {
        ra0 += BK * 4; ra1 += BK * 4; rb0 += BK; rb1 += BK;
        if (--rleft == 0) {
            asm volatile("" ::: "memory");                  // (stays a branch: as selects it would be the mask arithmetic again)
            if (rseg == 0 && c1 > c0) {
                rseg = 1; rleft = c1 - c0;
                ra0 = (const char*)rstash[0]; ra1 = (const char*)rstash[1];
                rb0 += koff1 - koff0 - c0 * BK; rb1 += koff1 - koff0 - c0 * BK;
            } else if (rseg <= 1 && ntiles_all > c1) {
                const int kprev = rseg == 0 ? koff0 + c0 * BK : koff1 + (c1 - c0) * BK;
                rseg = 2; rleft = ntiles_all - c1;
                ra0 = (const char*)rstash[2]; ra1 = (const char*)rstash[3];
                rb0 += koff2 - kprev; rb1 += koff2 - kprev;
            } else {
                rleft = 1 << 30;                             // behind the last tile: nothing is requested from here any more
            }
        }
    };
    auto load_tile_run = [&](GTile& gt) { issue_tile_asm(gt, ra0, ra1, rb0, rb1); };    // (run_advance() follows behind the tile's MFMAs)
    auto load_tile_run_plain = [&](GTile& gt) {             // the same request, visible to the compiler (tail tiles)
        gt.a[0] = *reinterpret_cast<const f32x4*>(ra0); gt.a[1] = *reinterpret_cast<const f32x4*>(ra1);
        gt.b[0] = *reinterpret_cast<const f32x4*>(rb0); gt.b[1] = *reinterpret_cast<const f32x4*>(rb1);
        run_advance();
    };
    auto store_tile = [&](const GTile& gt, int buf) {
        float* sa = smem + buf * 2 * TILE_FLOATS + r0 * LDW + 4 * kc;
        float* sb = sa + TILE_FLOATS;
        *reinterpret_cast<f32x4*>(sa) = gt.a[0]; *reinterpret_cast<f32x4*>(sa + 64 * LDW) = gt.a[1];
        *reinterpret_cast<f32x4*>(sb) = gt.b[0]; *reinterpret_cast<f32x4*>(sb + 64 * LDW) = gt.b[1];
    };
    // SPLIT: x = x0 + x1 + x2 in bf16 (round to nearest; the remainders are exact), four k of a row at a time -> 8 bytes per plane
#ifndef CASV_ABLM
#define CASV_ABLM 0         // timing-only builds of the SPLIT variant (wrong results), a bit mask: 1 no split arithmetic, 2 no LDS stores, 4 no global loads, 8 no barrier, 16 no fragment reads
#endif
    auto split4 = [&](const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2, const bool isb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (CASV_ABLM & 1) {
                p0[h] = __float_as_uint(x[2 * h]); p1[h] = __float_as_uint(x[2 * h + 1]); p2[h] = p0[h] ^ p1[h];
                continue;
            }
            const f32x2 v = {x[2 * h], x[2 * h + 1]};
            const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const f32x2 r1 = v - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
            const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
            const f32x2 r2 = r1 - f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
            p0[h] = q0; p1[h] = q1; p2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        }
    };
    // (staging rows r0 and r0 + 64 share bit 4: one offset serves both)
    const int st_off = r0 * 32 + ((((kc >> 1) ^ (r0 >> 4)) & 1) * 16) + (kc & 1) * 8;
    char* const st_base = reinterpret_cast<char*>(smem) + st_off;
    // plane `plane` (A 0..2, B 3..5) of LDS buffer `buf`, row r0 + 64 i: the thread's 8 bytes
    auto store_plane = [&](const u32x2 v, int buf, int plane, int i) {
        *reinterpret_cast<u32x2*>(st_base + buf * (BUF_FLOATS * 4) + plane * SPLIT_PLANE_BYTES + i * 64 * 32) = v;
    };
    auto store_tile_split = [&](const GTile& gt, int buf) {          // all six planes of a K tile (prologue)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 p0, p1, p2;
            split4(gt.a[i], p0, p1, p2, false);
            store_plane(p0, buf, 0, i); store_plane(p1, buf, 1, i); store_plane(p2, buf, 2, i);
            split4(gt.b[i], p0, p1, p2, true);
            store_plane(p0, buf, 3, i); store_plane(p1, buf, 4, i); store_plane(p2, buf, 5, i);
        }
    };
    auto zero_tile_split = [&](int buf, int plane0, int nplanes) {   // a stage's missing second K tile: zeros in both operands
        for (int p = plane0; p < plane0 + nplanes; ++p) { store_plane(u32x2{0u, 0u}, buf, p, 0); store_plane(u32x2{0u, 0u}, buf, p, 1); }
    };
    // Fragments of the 32-deep instruction (gemm_split.hip): lane l holds row (l & 15) of its 16-row block and the stage's
    // k 8 (l >> 4) .. + 7 -- 16-byte piece (l >> 4) & 1 of K tile l >> 5, i.e. of LDS buffer l >> 5; pieces swapped in odd blocks
    const int fr_lane = (lane >> 5) * (BUF_FLOATS * 4) + (lane & 15) * 32;
    const char* const fr_e = reinterpret_cast<const char*>(smem) + fr_lane + ((lane >> 4) & 1) * 16;
    const char* const fr_o = reinterpret_cast<const char*>(smem) + fr_lane + (((lane >> 4) & 1) ^ 1) * 16;
    auto frag_a = [&](int plane, int rb) { return *reinterpret_cast<const bf16x8*>(((rb & 1) ? fr_o : fr_e) + plane * SPLIT_PLANE_BYTES + (wave * 32 + rb * 16) * 32); };
    auto frag_b = [&](int plane, int c) { return *reinterpret_cast<const bf16x8*>(((c & 1) ? fr_o : fr_e) + (3 + plane) * SPLIT_PLANE_BYTES + c * 16 * 32); };
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

#ifdef CASV_GEMM_PROF
    CASV_STAMP(pa)
#endif
    // LSTM: the previous cell state of the tile -- 128 rows x 32 units, 128 bytes per (gathered) row -- is fetched like an operand
    // tile: every thread 16-byte pieces of its two staging rows (chunks kc and kc + 4), parked in registers under the K loop and
    // turned into the accumulator layout through LDS behind it.  Round 4, in-kernel stamps of the c3 decode
    // (profiles/r04_gemm_stamps.txt): as 16 dependent 4-byte gathers per lane in the prologue the cell state took 26 000 cycles
    // (11 us) to land -- in front of the first MFMA, because no compiler-tracked load may be pending on a path into the loop --
    // and 23 000 when requested behind the loop instead; 4-byte-per-lane accesses are what the memory pipeline is slowest at.
    // cell-state area behind the tile buffers: [half j of the 128-byte row][row][16 floats], the image a wave's LDS-DMA writes
    // (lane L -> 16 bytes at base + 16 L: rows wave * 16 + L / 4, chunk L % 4)
    float* const cs = smem_all + KS * (2 * BUF_FLOATS);
    auto cell_dma = [&](int i, int j) {
        const float* src = rstash[4 + i] + 16 * j;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) float*)cs + (unsigned)((j * 128 + 64 * i + (tid >> 6) * 16) * 64));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
#ifdef CASV_GEMM_PROF
    CASV_STAMP(pb)
#endif
    // SPLIT: the stage pipeline of gemm_split.hip (two K tiles per stage, k 0..15 from buffer 0 and k 16..31 from buffer 1; six products
    // of v_mfma_f32_16x16x32_bf16 per 16x16 block: a0.b2, a0.b1 | X | a1.b1, a1.b0, a2.b0 | Y | a0.b0; a plane's place in LDS refilled in
    // place with the next stage's once every wave has read it: b2, a0, b1, a1 behind X, b0, a2 behind Y) on this kernel's 128x128 tile --
    // a wave 32 rows x 128 columns, operands through registers -- so that an element's sum is the same instruction sequence in both
    // kernels (same bits).  Written with conditions throughout (any tile count, ragged tiles); the compiler schedules it.
    const bool c_late = !SPLIT && EPI == EPI_LSTM && KS == 1 && nt_min > 5;
    if constexpr (SPLIT) {
        f32x4 acc16[2][8];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc16[rb][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bf16x8 fb[8], fP[2], fQ[2];
        GTile gt0, gt1;
        if (ntiles > 0) load_tile(gt0, 0);
        if (ntiles > 1) load_tile(gt1, 1);
        if (cstage) {           // the previous cell state into its LDS image (its area is not part of the K loop): requested with the first tiles
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>(cs + (j * 128 + r0 + 64 * i) * 16 + 4 * kc) = *reinterpret_cast<const f32x4*>(cptr(i) + 16 * j);
        }
        if (ntiles > 0) store_tile_split(gt0, 0);
        if (ntiles > 1) store_tile_split(gt1, 1); else if (ntiles > 0) zero_tile_split(1, 0, 6);
        if (ntiles > 2) load_tile(gt0, 2);
        if (ntiles > 3) load_tile(gt1, 3);
        __syncthreads();
        if (ntiles > 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) fb[c] = frag_b(2, c);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) fQ[rb] = frag_a(0, rb);
        }
#define CASV_SPLIT_PROD(PA, RB, C) acc16[RB][C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PA[RB], fb[C], acc16[RB][C], 0, 0, 0);
#define CASV_SPLIT_COUT(PA, ROLL) _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) { _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) CASV_SPLIT_PROD(PA, rb_, c_) ROLL }
#define CASV_SPLIT_ROUT(PA, ROLL) _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_) { _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) CASV_SPLIT_PROD(PA, rb_, c_) ROLL }
        for (int s = 0; 2 * s < ntiles; ++s) {
            const bool next = 2 * s + 2 < ntiles, next1 = 2 * s + 3 < ntiles;
            u32x2 ha[4], hb[4];                                                                 // a2', b0' of the next stage until window 2
            CASV_SPLIT_COUT(fQ, { fb[c_] = frag_b(1, c_); if (c_ < 2) fP[c_] = frag_a(1, c_); })   // p1 a0.b2
            CASV_SPLIT_COUT(fQ, {})                                                                // p2 a0.b1
            __syncthreads();                                                                    // X
            if (next) {                                                                         // window 1
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u32x2 p0, p1, p2;
                    split4(gt0.a[i], p0, p1, p2, false); store_plane(p0, 0, 0, i); store_plane(p1, 0, 1, i); ha[i] = p2;
                    split4(gt0.b[i], p0, p1, p2, true); store_plane(p1, 0, 4, i); store_plane(p2, 0, 5, i); hb[i] = p0;
                }
                if (next1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        u32x2 p0, p1, p2;
                        split4(gt1.a[i], p0, p1, p2, false); store_plane(p0, 1, 0, i); store_plane(p1, 1, 1, i); ha[2 + i] = p2;
                        split4(gt1.b[i], p0, p1, p2, true); store_plane(p1, 1, 4, i); store_plane(p2, 1, 5, i); hb[2 + i] = p0;
                    }
                } else { zero_tile_split(1, 0, 2); zero_tile_split(1, 4, 2); }
                if (2 * s + 4 < ntiles) load_tile(gt0, 2 * s + 4);
                if (2 * s + 5 < ntiles) load_tile(gt1, 2 * s + 5);
            }
            CASV_SPLIT_COUT(fP, { fb[c_] = frag_b(0, c_); })                                       // p3 a1.b1
            CASV_SPLIT_ROUT(fP, { fP[rb_] = frag_a(2, rb_); })                                     // p4 a1.b0
            CASV_SPLIT_ROUT(fP, {})                                                                // p5 a2.b0
            __syncthreads();                                                                    // Y
            if (next) {                                                                         // window 2
                store_plane(ha[0], 0, 2, 0); store_plane(ha[1], 0, 2, 1); store_plane(hb[0], 0, 3, 0); store_plane(hb[1], 0, 3, 1);
                if (next1) { store_plane(ha[2], 1, 2, 0); store_plane(ha[3], 1, 2, 1); store_plane(hb[2], 1, 3, 0); store_plane(hb[3], 1, 3, 1); }
                else zero_tile_split(1, 2, 2);
            }
            CASV_SPLIT_COUT(fQ, { if (next) { fb[c_] = frag_b(2, c_); if (c_ < 2) fP[c_] = frag_a(0, c_); } })   // p6 a0.b0
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) fQ[rb] = fP[rb];
        }
#undef CASV_SPLIT_COUT
#undef CASV_SPLIT_ROUT
#undef CASV_SPLIT_PROD
        // the accumulators as 32x32 blocks (the layout the epilogue is written for: lane l holds column l & 31 and the rows
        // 4 (l >> 5) + (r & 3) + 8 (r >> 2)) through a private 4.5-KB piece of the tile buffers: the four 16x16 blocks of a 32x32 one are
        // written column by column (a lane's four consecutive rows: one 16-byte store; 36-float columns: conflict-free), read back as
        // four 16-byte pieces per lane
        __syncthreads();
        {
            float* const cv = smem + wave * (32 * 36);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int bj = 0; bj < 2; ++bj)
                        *reinterpret_cast<f32x4*>(cv + (16 * bj + (lane & 15)) * 36 + 16 * bi + 4 * (lane >> 4)) = acc16[bi][2 * c + bj];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(cv + l31 * 36 + 4 * lh + 8 * m);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[c][4 * m + j] = v[j];
                }
            }
        }
    } else {
    Frag f0, f1;
    GTile g0, g1;
    // prologue: LDS[0] <- tile 0, LDS[1] <- tile 1, F0 <- LDS[0]; G0 <- tile 2, G1 <- tile 3
    if (KS == 1 && kt_begin == 0 && c0 >= 2 && ntiles > 1) { load_tile_seg0(g0, 0); load_tile_seg0(g1, 1); }
    else {
        if (ntiles > 0) load_tile(g0, 0);
        if (ntiles > 1) load_tile(g1, 1);
    }
#ifdef CASV_GEMM_PROF
    CASV_STAMP(pc)
#endif
    if (ntiles > 0) store_tile(g0, 0);
    if (ntiles > 1) store_tile(g1, 1);
    // Steady state (tiles kt+1..kt+4 exist, no conditionals): while the 32 MFMAs of tile kt issue from FC,
    //   LDS[kt&1] <- G (tile kt+2, requested two steps ago; the buffer's old content, tile kt, sits in FC)
    //   G <- global tile kt+4 ;  FN <- LDS[(kt+1)&1]
    // with the issue order pinned: one memory instruction + a few address ops behind each of the first MFMAs,
    // so a wave has no memory-only phase in which its SIMD partner's MFMA stream starves its instruction issue.
#define CASV_TILE_FULL_X(FC, FN, G, KT, NMEM, EXTRA_LOADS)                                \
    {                                                                                     \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(G.a[0]), "+v"(G.a[1]), "+v"(G.b[0]), "+v"(G.b[1]));  \
        store_tile(G, (KT) & 1);                                                          \
        if (KS == 1) load_tile_run(G); else load_tile_asm(G, (KT) + 4);                  \
        EXTRA_LOADS                                                                       \
        read_frags(FN, ((KT) + 1) & 1);                                                   \
        mma(FC);                                                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < NMEM; ++q_) {                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                            \
            __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 32 - NMEM, 0);                        \
        __builtin_amdgcn_sched_barrier(0);   /* keep all 32 MFMAs in front of the barrier: by then the LDS ops have landed */ \
        if (KS == 1) run_advance();          /* (its rare branch ends the tile's basic block: behind everything that interleaves) */ \
        __syncthreads();                                                                  \
    }
#define CASV_TILE_FULL(FC, FN, G, KT) CASV_TILE_FULL_X(FC, FN, G, KT, 18, )
#define CASV_TILE_STEP(FC, FN, G, KT)                                                     \
    {                                                                                     \
        if ((KT) + 2 < ntiles) store_tile(G, (KT) & 1);                                   \
        if ((KT) + 4 < ntiles) { if (KS == 1) load_tile_run_plain(G); else load_tile(G, (KT) + 4); }  \
        if ((KT) + 1 < ntiles) read_frags(FN, ((KT) + 1) & 1);                            \
        if ((KT) < ntiles) mma(FC);                                                       \
        __syncthreads();                                                                  \
    }
    int kt = 0;
    // (hidden transfers in flight across the tail below are older than anything the tail requests itself -- an odd tile count
    // requests one more tile there --, so the compiler's counted waits for its own loads stay correct: they only wait longer)
    if (nt_min > 5) {
        // Tiles 2 and 3 are requested the hidden way already: a compiler-tracked load pending on ANY path into the loop would
        // put a full vmcnt(0) at the loop head, executed in every iteration.  (The cell-state loads above are older than
        // every tile load, so the counted waits cover them too.)
        // (c0 >= 4: tiles 2 and 3 lie in the layer input too -- their addresses do not wait for the gathered segments' row indices)
        {
            const char *qa0, *qa1, *ra0, *ra1; const float *qb0, *qb1, *rb0, *rb1;
            if (KS == 1 && kt_begin == 0 && c0 >= 4) {      // (KS = 2: a wave group's tiles 2 and 3 are tiles 4 + grp and 6 + grp)
                qa0 = (const char*)(ap0[0] + 2 * BK); qa1 = (const char*)(ap0[1] + 2 * BK); qb0 = bp[0] + koff0 + 2 * BK; qb1 = bp[1] + koff0 + 2 * BK;
                ra0 = (const char*)(ap0[0] + 3 * BK); ra1 = (const char*)(ap0[1] + 3 * BK); rb0 = bp[0] + koff0 + 3 * BK; rb1 = bp[1] + koff0 + 3 * BK;
                asm volatile("" : "+v"(qa0), "+v"(ra0));           // (keeps the two paths apart: a select would wait for the indices)
            } else {
                auto addr = [&](int kt_rel, const char*& pa0, const char*& pa1, const float*& pb0, const float*& pb1) {
                    const int ktt = kt_rel * KS + grp + kt_begin;
                    const long long m1 = (ktt >= c0 && ktt < c1) ? -1LL : 0LL, m2 = (ktt >= c1) ? -1LL : 0LL;
                    const int ko = ktt - ((int)m1 & c0) - ((int)m2 & c1);
                    const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * BK;
                    pa0 = (const char*)ap0[0] + (d1_0 & m1) + (d2_0 & m2) + (long long)ko * (BK * 4);
                    pa1 = (const char*)ap0[1] + (d1_1 & m1) + (d2_1 & m2) + (long long)ko * (BK * 4);
                    pb0 = bp[0] + kb; pb1 = bp[1] + kb;
                };
                addr(2, qa0, qa1, qb0, qb1); addr(3, ra0, ra1, rb0, rb1);
            }
            issue_tile_asm(g0, qa0, qa1, qb0, qb1); issue_tile_asm(g1, ra0, ra1, rb0, rb1);
        }
        __syncthreads();
        read_frags(f0, 0);
        // The first steady-state step stores tile 2 into LDS buffer 0: every wave must have taken its fragments of tile 0
        // out of it first.  (Without this barrier only the latency of the tile-2 global loads kept a fast wave's store
        // behind a slow wave's read -- not enough once other kernels share the CU.)
        __syncthreads();
#ifdef CASV_GEMM_PROF
        pt1 = __builtin_amdgcn_s_memtime(); pr1 = __builtin_amdgcn_s_memrealtime();
#endif
        if (KS == 1) run_set(4);                    // the first steady-state tile requests tile 4
        for (; kt + 5 < nt_min; kt += 2) {          // both wave groups have all the tiles of the steady state
            CASV_TILE_FULL(f0, f1, g0, kt)
            CASV_TILE_FULL(f1, f0, g1, kt + 1)
        }
#ifdef CASV_GEMM_PROF
        pt2 = __builtin_amdgcn_s_memtime(); pr2 = __builtin_amdgcn_s_memrealtime(); ptiles = kt;
#endif
        // The cell state is requested behind the LAST tile loads, as LDS-DMA (no registers are held for it; hidden from the
        // compiler's wait bookkeeping like the tile loads): nothing is queued behind it that a counted wait looks at, so however
        // long the gathered rows take (rows of parents from many steps ago: HBM and page-table walks), no operand tile waits for
        // them; they land under the tail tiles.  The tile loads themselves land before the compiler-scheduled rest touches them.
        // (Every steady-state path of the LSTM kernel issues the four transfers -- jobs without a cell state from a valid dummy
        // address -- so that ONE wait statement with ONE count follows the loop: two statements on two paths would meet in copies of
        // registers whose loads are still in flight, csrc/check_asm_loads.py.)
        if (EPI == EPI_LSTM && KS == 1) {
            cell_dma(0, 0); cell_dma(0, 1); cell_dma(1, 0); cell_dma(1, 1);
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(g0.a[0]), "+v"(g0.a[1]), "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g1.a[0]), "+v"(g1.a[1]), "+v"(g1.b[0]), "+v"(g1.b[1]));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.a[0]), "+v"(g0.a[1]), "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g1.a[0]), "+v"(g1.a[1]), "+v"(g1.b[0]), "+v"(g1.b[1]));
        }
    } else {
        if (ntiles > 2) load_tile(g0, 2);
        if (ntiles > 3) load_tile(g1, 3);
        if (KS == 1 && ntiles > 4) run_set(4);
        __syncthreads();
        if (ntiles > 0) read_frags(f0, 0);
        __syncthreads();
    }
    if (cstage && !c_late) {        // (short K, odd tile counts, wave-group split-K: through registers, stored into the same image)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(cs + (j * 128 + r0 + 64 * i) * 16 + 4 * kc) = *reinterpret_cast<const f32x4*>(cptr(i) + 16 * j);
    }
    {
    for (; kt + 1 < nt_max; kt += 2) {          // same barrier count for both groups
        CASV_TILE_STEP(f0, f1, g0, kt)
        CASV_TILE_STEP(f1, f0, g1, kt + 1)
    }
    if (kt < nt_max) CASV_TILE_STEP(f0, f1, g0, kt)
    }
#undef CASV_TILE_STEP
#undef CASV_TILE_FULL

    }
#ifdef CASV_GEMM_PROF
    pe = __builtin_amdgcn_s_memtime();
#endif
    if (KS > 1) {       // acc(group 0) += acc(group 1), through LDS (all staging reads are behind the last barrier)
        float* red = smem_all + tid;
        if (grp == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(c * 16 + r) * 256] = acc[c][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] += red[(c * 16 + r) * 256];
    }

    // the cell state in the accumulator layout, out of its LDS image (every wave waits for its own transfers first -- also where
    // the image is not read: an LDS-DMA must not outlive the workgroup whose LDS it writes)
    if (c_late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float cpv[16];
    float bi = 0.f, bf_ = 0.f, bg = 0.f, bo = 0.f;       // LSTM: gate biases of this lane's unit (ONE branch; they arrive under the barrier)
    if (EPI == EPI_LSTM) {          // (uniform over the workgroup's remaining waves: group 1 of a KS = 2 launch has returned)
        if (!g.epi_plain) {
            if (g.bias) { bi = g.bias[n0 + l31]; bf_ = g.bias[n0 + 32 + l31]; bg = g.bias[n0 + 64 + l31]; bo = g.bias[n0 + 96 + l31]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cpv[r] = czero ? 0.0f : cs[((l31 >> 4) * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 16 + (l31 & 15)];
        }
    }

    // ---- epilogue ----
    if (EPI == EPI_PLAIN || g.epi_plain) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        if (nsplit == 1 && m0 + BM <= g.M && n0 + BN <= g.N) {
            // full tile, plain stores: no control flow between the 64 stores (a branch per element makes the compiler wait for the
            // element before -- see the LSTM epilogue below); C += : all 64 old values are requested first, then added and stored
            float* cb = cbase + (long long)(m0 + wave * 32 + 4 * lh) * g.out.ld + n0 + l31;
            float bcol[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bcol[c] = g.bias ? g.bias[n0 + c * 32 + l31] : 0.0f;
            if (g.accumulate) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] += cb[(long long)((r & 3) + 8 * (r >> 2)) * g.out.ld + c * 32];
            }
#if CASV_PLAIN_WIDE
            if (EPI == EPI_PLAIN && (g.out.ld & 3) == 0) {
                // 16-byte stores through an in-quad transpose: 16 store instructions per wave instead of 64, same bytes and addresses.
                // Pays where tiles are short (the train step's K = 512 contractions: c4 66.4-66.5 -> 66.0-66.3 ms per step); in the
                // LSTM kernel -- its cell epilogue, and the query job that rides in layer 1's launch -- it measured 0.3 % slower (c3).
                const int q = lane & 3;
                float* cq = cb + (long long)q * g.out.ld - q;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<QuadF4*>(cq + (long long)(8 * gq) * g.out.ld + c * 32) =
                            quad_transpose(acc[c][4 * gq] + bcol[c], acc[c][4 * gq + 1] + bcol[c], acc[c][4 * gq + 2] + bcol[c], acc[c][4 * gq + 3] + bcol[c], q);
            } else
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(long long)((r & 3) + 8 * (r >> 2)) * g.out.ld + c * 32] = acc[c][r] + bcol[c];
        } else
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c * 32 + l31;
            if (n < g.N) {
                const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < g.M) {
                        float* dst = cbase + (long long)m * g.out.ld + n;
                        if (nsplit > 1) atomicAdd(dst, acc[c][r] + (zidx == 0 ? b : 0.0f));
                        else *dst = g.accumulate ? (*dst + acc[c][r] + b) : (acc[c][r] + b);
                    }
                }
            }
        }
    } else {
        const int u = bn * 32 + l31;     // hidden unit of this lane
        const float* zin = g.zinit.base
            ? g.zinit.base + (long long)(step * g.zinit.step_mul + g.zinit.step_add) * g.zinit.slot_stride : nullptr;
        float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
        float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        float* gout = g.gates_out.base
            ? g.gates_out.base + (long long)(step * g.gates_out.step_mul + g.gates_out.step_add) * g.gates_out.slot_stride
            : nullptr;
        float* hout2 = g.out2.base ? g.out2.base + (long long)(step * g.out2.step_mul + g.out2.step_add) * g.out2.slot_stride : nullptr;
        if (zin) {      // precomputed input term (train step): all loads issue before the first store of the loop below
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float* zr = zin + (long long)(m < g.M ? m : g.M - 1) * g.zinit.ld + n0 + l31;
                acc[0][r] += zr[0]; acc[1][r] += zr[32]; acc[2][r] += zr[64]; acc[3][r] += zr[96];
            }
        }
        if (m0 + BM <= g.M && !hout2 && !gout) {
            // Full tile, inference outputs only: all cells first, then all stores, no control flow in between.  (With a branch per
            // row the compiler's wait bookkeeping put an s_waitcnt vmcnt(0) in front of every row's cell -- stores count in
            // vmcnt too, so each of the 16 rows waited for the stores of the row before: 16 serial store round trips, 44 000 cycles
            // of epilogue in the c3 decode where the arithmetic needs 7 000.)
            float hv[16], cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const LstmCellOut cell = lstm_cell(acc[0][r] + bi, acc[1][r] + bf_, acc[2][r] + bg, acc[3][r] + bo, cpv[r]);
                hv[r] = cell.h; cv[r] = cell.c;
            }
            float* cb = cout + (long long)(m0 + wave * 32 + 4 * lh) * g.c_out.ld + u;
            float* hb = hout + (long long)(m0 + wave * 32 + 4 * lh) * g.out.ld + u;
            // (16-byte stores from an in-quad transpose of the accumulators -- 8 store instructions per wave instead of 32 --
            // measured again in round 5, behind the round-4 epilogue: c3 271.0-271.3 ms without, 271.6-272.7 with.  Not store-issue bound.)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                cb[(long long)dm * g.c_out.ld] = cv[r];
                hb[(long long)dm * g.out.ld] = hv[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < g.M) {
                    const float cprev = cpv[r];
                    float zi = acc[0][r] + bi, zf = acc[1][r] + bf_, zg = acc[2][r] + bg, zo = acc[3][r] + bo;
                    const LstmCellOut cell = lstm_cell(zi, zf, zg, zo, cprev);
                    cout[(long long)m * g.c_out.ld + u] = cell.c;
                    hout[(long long)m * g.out.ld + u] = cell.h;
                    if (hout2) hout2[(long long)m * g.out2.ld + u] = cell.h;
                    if (gout) {
                        float* gr = gout + (long long)m * g.gates_out.ld + n0 + l31;
                        gr[0] = cell.i; gr[32] = cell.f; gr[64] = cell.g; gr[96] = cell.o;
                    }
                }
            }
        }
    }
#ifdef CASV_GEMM_PROF
#ifdef CASV_GEMM_PROF_PLAIN
    if (EPI == EPI_PLAIN && KS == 1 && threadIdx.x == 0 && ptiles > 0) {
#else
    if (EPI == EPI_LSTM && KS == 1 && threadIdx.x == 0 && ptiles > 0) {
#endif
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_gemm_prof[0], pt1 - pt0); atomicAdd(&g_gemm_prof[1], pt2 - pt1); atomicAdd(&g_gemm_prof[2], (unsigned long long)ptiles);
        atomicAdd(&g_gemm_prof[3], pt3 - pt2); atomicAdd(&g_gemm_prof[4], pr2 - pr1); atomicAdd(&g_gemm_prof[5], 1ull);
        atomicAdd(&g_gemm_prof[6], pt3 - pt0);
        { unsigned long long bin = (pt3 - pt0) >> 14; atomicAdd(&g_gemm_prof[16 + (bin < 63 ? bin : 63)], 1ull); }
        atomicAdd(&g_gemm_prof[8], pa - pt0); atomicAdd(&g_gemm_prof[9], pb - pa); atomicAdd(&g_gemm_prof[10], pc - pb); atomicAdd(&g_gemm_prof[11], pt1 - pc);
        atomicAdd(&g_gemm_prof[12], pe - pt2); atomicAdd(&g_gemm_prof[13], pt3 - pe);
    }
#endif
}

template <int EPI, int KS, bool SPLIT = false>
__global__ __launch_bounds__(256 * KS, 2) void gemm_kernel(const GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const GemmArgs& g = batch.g[blockIdx.y];
    const int nbn = (g.N + BN - 1) / BN;
    int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;
    if (g.xcd_rows > 0) {
        // Workgroups are dealt round-robin over the 8 XCDs (private L2s).  Give each XCD a compact block of the
        // tile grid so that an A row-panel / B column-panel is fetched into ONE L2 instead of all eight
        // (placement only changes speed, never results).
        const int xr = g.xcd_rows, xc = 8 / xr;
        const int nbm = (g.M + BM - 1) / BM;
        const int pr = nbm / xr, pc = nbn / xc;            // tiles per XCD along rows / columns (host checked divisibility)
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        bm = (xcd / xc) * pr + local / pc;
        bn = (xcd % xc) * pc + local % pc;
    }
    gemm_tile<EPI, KS, SPLIT>(g, bm, bn, blockIdx.z, gridDim.z, smem_all);
}

template <int EPI, int KS, bool SPLIT = false>
static void launch_one(const GemmBatch& bb, int blocks, int ksplit, hipStream_t stream) {
    // (CASV_DEBUG_LDS_PAD: measurement aid -- extra dynamic LDS, e.g. 16384 leaves room for ONE workgroup per CU)
    static const int lds_pad = [] { const char* e = getenv("CASV_DEBUG_LDS_PAD"); return e ? atoi(e) : 0; }();
    const int lds_bytes = (SPLIT ? 2 * SPLIT_BUF_FLOATS * 4 : GEMM_LDS_BYTES * KS) + CELL_LDS_BYTES + STASH_LDS_BYTES + lds_pad;
    // the attribute belongs to the (function, device) pair: a second model on another device needs its own
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI, KS, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_kernel<EPI, KS, SPLIT>), dim3(blocks, bb.count, ksplit), dim3(256 * KS), lds_bytes, stream, bb);
}

static int g_tile_mode = [] { const char* e = getenv("CASV_GEMM_TILE"); return e && e[0] >= '0' && e[0] <= '2' && !e[1] ? e[0] - '0' : -1; }();   // -1 by size, 0 = 128x128, 1 = 32x128, 2 = 64x128 where possible
void set_gemm_tile_mode(int mode) { g_tile_mode = mode; }
// Arithmetic of a GEMM launch: 0 = the fp32-input matrix instruction (k-ordered fmaf chain), 1 / 2 = bf16x3-split
// operands on the bf16 matrix instruction with fp32 accumulation (1: gemm_tile's SPLIT variant, 128x128 tiles; 2: 256x256 tiles,
// gemm_split.hip, where a job fills the chip that way, else as 1 -- both give the same bits).  WHICH of them a launch takes is
// decided by the C-ABI entry point that enqueues it (engine.h, arithmetic_of: the handle's "arithmetic" option; by default the beam
// search's decoder steps take 2 and everything else 0) and announced to the launcher for the calling thread (SplitScope).
// The process-wide override (option "split_bf16" / CASV_SPLIT_BF16 = 0, 1 or 2 in the environment when the library is loaded)
// puts EVERY launch of the decode path of every handle on one arithmetic: tests and A/B measurements.
static int g_split_override = [] { const char* e = getenv("CASV_SPLIT_BF16"); return e && e[0] >= '0' && e[0] <= '2' && !e[1] ? e[0] - '0' : -1; }();
static thread_local int t_split_bf16 = 0;
// (a captured step graph bakes in the arithmetic and the addresses of the pre-split weight images: both are part of its key through
// this counter -- engine.hip, StepRunner)
static long g_split_epoch = 0;
long gemm_split_epoch() { return g_split_epoch; }
void gemm_split_bump_epoch() { ++g_split_epoch; }
void set_gemm_split_override(int v) { v = v < -1 ? -1 : v > 2 ? 2 : v; if (v != g_split_override) ++g_split_epoch; g_split_override = v; }
int gemm_split_override() { return g_split_override; }
int gemm_split_enter(int mode) { const int prev = t_split_bf16; t_split_bf16 = mode < 0 ? 0 : mode > 2 ? 2 : mode; return prev; }
int gemm_split_bf16() { return t_split_bf16; }

static int count_ktiles(const GemmArgs& g) {
    int ktiles = 0;
    for (int i = 0; i < g.nseg; ++i) ktiles += g.a[i].width / BK;
    return ktiles;
}

// Tile shape and split-K factor of a launch.  Launches that would occupy at most half the CUs as 128x128 tiles (even
// after split-K) run as 32x128 tiles: 4x the workgroups, same values ...
struct GemmPlan { int blocks, sblocks, ksplit; bool skinny; int srows; };
static GemmPlan plan_gemm(int epi, const GemmBatch& b) {
    GemmPlan p{0, 0, 1, false, 32};
    for (int j = 0; j < b.count; ++j) {
        const int nbn = (b.g[j].N + BN - 1) / BN;
        const int nb = ((b.g[j].M + BM - 1) / BM) * nbn, ns = ((b.g[j].M + 31) / 32) * nbn;
        p.blocks = nb > p.blocks ? nb : p.blocks;
        p.sblocks = ns > p.sblocks ? ns : p.sblocks;
    }
    bool splittable = epi == EPI_PLAIN;         // split-K: every job of the batch must ask for it and share the shape
    for (int j = 0; j < b.count; ++j)
        if (b.g[j].step_ptr || b.g[j].ksplit == 0 || b.g[j].ksplit == 1 || b.g[j].ksplit != b.g[0].ksplit || b.g[j].Ktot != b.g[0].Ktot ||
            b.g[j].M != b.g[0].M || b.g[j].N != b.g[0].N) splittable = false;
    auto choose_ksplit = [&](int grid, int want) {
        if (!splittable) return 1;
        const int ktiles = count_ktiles(b.g[0]);
        int ks = b.g[0].ksplit;
        if (ks < 0) {                           // fill the chip, keep >= 16 k-tiles per workgroup
            ks = (want + grid - 1) / grid;
            if (ks > ktiles / 16) ks = ktiles / 16;
        }
        if (ks > ktiles) ks = ktiles;
        return ks < 1 ? 1 : ks;
    };
    p.ksplit = choose_ksplit(p.blocks * b.count, 512);
    // ... and so do launches of more than one but less than two 128x128 tiles per CU: half the CUs would carry two workgroups
    // and set the pace while the others idle (the encoder's anti-diagonals of three layers at 1024 lines: 384 tiles, 168 us;
    // as 1536 tiles of 32x128: 140 us)
    static const int ncu = [] { hipDeviceProp_t pr{}; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    const int grid = p.blocks * b.count * p.ksplit;
    p.skinny = g_tile_mode == 1 || (g_tile_mode < 0 && (grid <= 128 || (grid > ncu && grid < 2 * ncu)));
    if (p.skinny) p.ksplit = choose_ksplit(p.sblocks * b.count, 1024);
    // Split-K launches of few rows (the train step's per-time-step data GEMMs: 512 rows, K = 2048): 32x128 tiles with just enough
    // K splits for ONE round of workgroups.  As 128x128 tiles they needed eight splits to fill the chip, and the eight partial
    // tiles per output went through float atomics -- 16.8 MB per launch at the memory side's ~1.3 TB/s = 13 of the launch's
    // 36 us.  Measured on the train step of configs[3] (splits 2 / 3 / 4 / 6 / 8 of 32x128 tiles: 90.2 / 91.8 / 87.9 / 89.9 /
    // 94.2 ms per step; 128x128 tiles with four splits 103.2; before 94.1).
    if (g_tile_mode < 0 && splittable && b.g[0].ksplit < 0 && b.g[0].M <= 1024) {
        p.skinny = true;
        p.ksplit = choose_ksplit(p.sblocks * b.count, 512);
    }
    // 64 x 128 tiles (gemm_skinny.hip, two row blocks per workgroup) for launches without split-K that leave at most one 128x128
    // workgroup per CU, or go as 32-row tiles, while 64-row tiles still put two on every CU: the encoder at 1024 lines -- layer 1's
    // two directions (256 tiles of 128x128) and the anti-diagonals of the layers above (384)
    // (fused-LSTM launches only: the plain-epilogue launch it would also catch, the page call's logits -- 10240 x 640 x 512 --, runs
    // 67.6 us as 32-row tiles and 75.6 us as 64-row tiles)
    if ((epi == EPI_LSTM || g_tile_mode == 2) && !splittable && p.ksplit == 1) {
        int blocks64 = 0;
        for (int j = 0; j < b.count; ++j) blocks64 = std::max(blocks64, ((b.g[j].M + 63) / 64) * ((b.g[j].N + BN - 1) / BN));
        const bool fits = blocks64 * b.count >= 2 * ncu;
        if (g_tile_mode == 2 || (g_tile_mode < 0 && fits && (p.skinny || grid < 2 * ncu))) { p.skinny = true; p.srows = 64; }
    }
    return p;
}
bool gemm_is_skinny(int epi, const GemmBatch& b) { return plan_gemm(epi, b).skinny; }
static bool train_launch(const GemmBatch& b) {        // (the train step's fused-LSTM launches carry side outputs the decode path never has)
    for (int j = 0; j < b.count; ++j) if (b.g[j].zinit.base || b.g[j].gates_out.base || b.g[j].out2.base) return true;
    return false;
}
static bool splittable_launch(int epi, const GemmBatch& b) {        // (a launch whose jobs ask for split-K: never re-planned below)
    if (epi != EPI_PLAIN) return false;
    for (int j = 0; j < b.count; ++j) if (b.g[j].ksplit != 0 && b.g[j].ksplit != 1) return true;
    return false;
}

// Train step's big plain contractions (GemmArgs.ksplit = -1: the caller leaves the launch form to the launcher; M = 51 712 rows):
// a 128x128 tile grid that ends in a partial round of workgroups -- 1 616 tiles on 512 slots = 3.16 rounds -- spends the time of
// most of a round on its last few tiles (a workgroup alone on its CU still needs ~2/3 of the time two of them share).  The row
// blocks of that partial round go as a second launch of 32-row (64-row) tiles instead: four (two) times the workgroups for the same
// rows, every element contracted by the same k-ordered chain (gemm_skinny.hip: same bits, no split-K, nothing atomic).
// CASV_TAIL_CUT=0 switches it off (A/B measurements).
static bool cut_partial_round(int epi, const GemmBatch& b, const GemmPlan& plan, hipStream_t stream) {
    static const bool enabled = [] { const char* e = getenv("CASV_TAIL_CUT"); return !(e && e[0] == '0'); }();
    if (!enabled || epi != EPI_PLAIN || b.count != 1 || plan.ksplit != 1 || plan.skinny || g_tile_mode >= 0 || t_split_bf16) return false;
    const GemmArgs& g = b.g[0];
    if (g.ksplit != -1 || g.step_ptr || g.nact || g.Bimg) return false;
    for (int i = 0; i < g.nseg; ++i) if (g.a[i].rows || g.a[i].skip_first || g.a[i].first_base) return false;    // rows at base + m * ld only
    static const int ncu = [] { hipDeviceProp_t pr{}; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN, slots = 2 * ncu;
    const int tiles = nbm * nbn, rounds = tiles / slots, rem = tiles % slots;
    if (rounds < 1 || rem == 0 || rem * 4 > slots * 3) return false;         // (a last round that is three quarters full is left alone)
    const int main_bm = (rounds * slots) / nbn;
    if (main_bm < 1 || main_bm >= nbm) return false;
    GemmBatch bm{}, bt{};
    bm.count = bt.count = 1;
    GemmArgs& gm = bm.g[0]; GemmArgs& gt = bt.g[0];
    gm = g; gt = g;
    gm.M = main_bm * BM; gm.ksplit = 0;                                      // (0: one block per tile, and no second look at this function)
    gt.M = g.M - gm.M; gt.ksplit = 0;
    for (int i = 0; i < g.nseg; ++i) gt.a[i].base += (long long)gm.M * g.a[i].ld;
    gt.out.base += (long long)gm.M * g.out.ld;
    launch_gemm_batch(epi, bm, stream);
    const int tail_tiles = (nbm - main_bm) * nbn;
    launch_gemm_skinny(epi, bt, 1, tail_tiles * 4 <= 2 * slots ? 32 : 64, stream);
    return true;
}

// Split arithmetic, 256x256 tiles (gemm_split.hip: ONE workgroup per CU): a grid of more than one round of workgroups that ends in
// a partial round -- the page call's 10 240 rows x 2048 gate columns = 320 tiles on 256 CUs: the second round occupies a quarter
// of the chip for the time of a whole one -- keeps its whole rounds; the rows of the partial round (and a ragged last row block)
// go as 128x128 split tiles, which contract every element with the same instruction sequence (same bits: tests/test_gpu_split.py).
// The tail job addresses its rows through offset pointers (no kernel knows about the cut).  false: the job stays as it is.
static bool split256_cut_rows(int epi, const GemmArgs& g, GemmArgs& head, GemmArgs& tail) {
    static const bool enabled = [] { const char* e = getenv("CASV_SPLIT_TAIL_CUT"); return !(e && e[0] == '0'); }();
    static const int ncu = [] { hipDeviceProp_t pr{}; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    if (!enabled || g.N <= 0 || g.N % 256 || g.M < 256) return false;
    const int nbn = g.N / 256, nbm = g.M / 256, tiles = nbm * nbn;
    const int rounds = tiles / ncu, rem = tiles % ncu;
    int main_bm;
    if (rounds >= 1 && rem > 0 && rem * 4 <= ncu * 3) main_bm = (rounds * ncu) / nbn;      // (a last round that is three quarters full is left alone)
    else if (g.M % 256) main_bm = nbm;                                                      // ragged last row block only
    else return false;
    if (main_bm < 1 || (long long)main_bm * 256 >= g.M) return false;
    const int m_off = main_bm * 256;
    if (g.nact && (g.nact_group <= 0 || m_off % g.nact_group)) return false;
    head = g; tail = g;
    head.M = m_off; tail.M = g.M - m_off;
    auto shift = [&](Seg& sg) {
        if (!sg.base && !sg.first_base) return;
        if (sg.rows) sg.rows += m_off; else if (sg.base) sg.base += (long long)m_off * sg.ld;
        if (sg.first_base) sg.first_base += (long long)m_off * sg.ld;
    };
    for (int i = 0; i < g.nseg; ++i) shift(tail.a[i]);
    shift(tail.c_in);
    for (SlotPtr* sp : {&tail.out, &tail.c_out, &tail.zinit, &tail.gates_out, &tail.out2})
        if (sp->base) sp->base += (long long)m_off * sp->ld;
    if (g.nact) tail.nact += m_off / g.nact_group;
    (void)epi;
    return true;
}

void launch_gemm_batch(int epi, const GemmBatch& b, hipStream_t stream) {
    const GemmPlan plan = plan_gemm(epi, b);
    if (cut_partial_round(epi, b, plan, stream)) return;
    const int blocks = plan.blocks, ksplit = plan.ksplit;
    bool skinny = plan.skinny;
    // split-bf16 experiment: a launch that would go as 64- or 32-row tiles only because 128x128 tiles leave CUs idle (the
    // encoder's cells at 1024 lines: 256 / 384 tiles; the attention-query job) runs faster as 128x128 tiles on the bf16
    // instruction, one workgroup per CU, than as small tiles on the fp32-input one (c3: 190.9 -> 178 ms per batch)
    // ... and every other inference launch too, whatever tile shape was asked for: with the option on, all GEMM launches of the
    // decode path share ONE arithmetic (128x128 and 256x256 split tiles give the same bits), so that a row's result does not depend
    // on the batch it sits in -- the property the fp32-input kernels have among themselves.  (The train step's per-time-step launches
    // keep the fp32-input small tiles: its persistent recurrences, which they must equal, are fp32-input kernels.)
    if (t_split_bf16 && skinny && ksplit == 1 && !splittable_launch(epi, b) && !train_launch(b)) skinny = false;
    // XCD-aware tile order: minimise (A bytes x column-splits + B bytes x row-splits) over the 8 = xr * xc splits
    GemmBatch bb = b;
    for (int j = 0; j < bb.count; ++j) {
        GemmArgs& g = bb.g[j];
        const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
        g.xcd_rows = 0;
        if ((nbm * nbn) % 8 != 0 || nbm * nbn != blocks) continue;
        double best = 0; int best_xr = 0;
        for (int xr = 1; xr <= 8; xr *= 2) {
            const int xc = 8 / xr;
            if (nbm % xr || nbn % xc) continue;
            const double cost = (double)g.M * xc + (double)g.N * xr;     // both operands share K
            if (!best_xr || cost < best) { best = cost; best_xr = xr; }
        }
        g.xcd_rows = best_xr;
    }
    for (int j = 0; j < b.count && ksplit > 1; ++j) {
        const GemmArgs& g = b.g[j];
        if (g.accumulate || g.out_zeroed) continue;   // partial sums are added atomically: start from zero
        // the slot to clear is computed from the immediate step: split-K jobs never take their step from device memory
        // (plan_gemm keeps ksplit at 1 for them)
        float* cbase = g.out.base + (long long)(g.step_imm * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        if (g.out.ld == g.N) (void)hipMemsetAsync(cbase, 0, (size_t)g.M * g.N * sizeof(float), stream);
        else (void)hipMemset2DAsync(cbase, (size_t)g.out.ld * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, stream);
    }
    if (skinny) { launch_gemm_skinny(epi, bb, ksplit, plan.srows, stream); return; }
    // wave-group split-K (train step only, GemmArgs.kgroups): worth it while the grid leaves CUs idle
    bool two = true;
    for (int j = 0; j < b.count; ++j)
        if (b.g[j].kgroups != 2 || count_ktiles(b.g[j]) / ksplit < 16) two = false;
    if (blocks * b.count * ksplit > 256) two = false;
    if (t_split_bf16 && !two) {
        if (t_split_bf16 >= 2 && ksplit == 1) {
            // jobs that fill the chip as 256x256 tiles go to gemm_split.hip; the rest of the batch follows as a launch of its own
            GemmBatch big{}, rest{};
            for (int j = 0; j < b.count; ++j) {
                GemmArgs head = b.g[j], tail{};
                // (a job whose 256x256 grid ends in a partial round of workgroups, or in a ragged row block: whole rounds here, the
                // remaining rows as 128x128 split tiles -- the same bits -- in the launch of the rest)
                const bool cut = split256_cut_rows(epi, b.g[j], head, tail);
                if (gemm_split256_wants(epi, head)) {
                    big.g[big.count++] = head;
                    if (cut && rest.count < GEMM_MAX_JOBS) { rest.g[rest.count++] = tail; continue; }
                    if (cut) { big.g[big.count - 1] = b.g[j]; }      // (no room for the tail job: the job stays whole)
                    continue;
                }
                rest.g[rest.count] = b.g[j]; if (epi == EPI_PLAIN) rest.g[rest.count].epi_plain = 0; ++rest.count;
            }
            if (big.count && launch_gemm_split256(epi, big, stream)) {
                if (rest.count) launch_gemm_batch(epi, rest, stream);
                return;
            }
        }
        if (epi == EPI_LSTM) launch_one<EPI_LSTM, 1, true>(bb, blocks, ksplit, stream); else launch_one<EPI_PLAIN, 1, true>(bb, blocks, ksplit, stream);
        return;
    }
    if (epi == EPI_LSTM) { if (two) launch_one<EPI_LSTM, 2>(bb, blocks, ksplit, stream); else launch_one<EPI_LSTM, 1>(bb, blocks, ksplit, stream); }
    else { if (two) launch_one<EPI_PLAIN, 2>(bb, blocks, ksplit, stream); else launch_one<EPI_PLAIN, 1>(bb, blocks, ksplit, stream); }
}

void launch_gemm(int epi, const GemmArgs& g, hipStream_t stream) {
    GemmBatch b;
    b.g[0] = g;
    b.count = 1;
    launch_gemm_batch(epi, b, stream);
}

}  // namespace casv
