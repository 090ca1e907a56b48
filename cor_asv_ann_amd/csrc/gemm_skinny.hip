// Skinny-M variant of the fused GEMM (same GemmArgs, same results bit for bit as gemm.hip with KS = 1).
//
// A 128x128 tile grid gives ceil(M/128) * N/128 workgroups: at M = 512 (train step) that is 64 of them, at M = 256
// (greedy decode of a default batch) 32 -- on 256 CUs, each CU's matrix pipe then works through a whole 128x128xK tile
// while most of the chip idles.  Here a workgroup owns a 32 x 128 tile (4x as many workgroups) and its four waves
// own one 32x32 accumulator tile each -- wave w = gate w of the LSTM epilogue -- over the FULL K range, in the same
// k order and with the same MFMA sequence per element as the big kernel (so a row's value does not depend on which
// variant ran).
//
// Operands are staged through LDS in steps of 32 k (round 2).  The first version let every lane pull its fragments straight
// from global memory, 16 B out of each of 32 rows per instruction: a 128-B line was visited by four load instructions of two
// K tiles out of an L1 that the live lines did not fit, and the kernel ran at ~50 TFLOP/s whatever the MFMA chains did
// (profiles/r02_gemm_tile_trace.txt).  Now 8 threads take one whole line of a row at once (each line leaves the L2 exactly
// once per workgroup), two stages are double-buffered in LDS (+4-float row pad: ds_read_b128 conflict-free), and the stage
// after next is in flight in registers while the 16 MFMAs of the current one issue.  The LDS then brings the four gates of
// a (row, unit) into one lane for the cell epilogue.
#include "common.h"

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SBN = 128;
constexpr int SK2 = 32, SLD = SK2 + 4;                       // k per LDS stage, padded row stride (floats)

// RB = row blocks of 32 per workgroup: 32 x 128 tiles, or 64 x 128 (round 3: every wave carries two accumulators on one B
// fragment -- half the B traffic and LDS reads per FLOP; for launches whose 128x128 grid leaves one workgroup per CU or less
// while 64-row tiles still give every CU two: the encoder at 1024 lines).  Same k order and MFMA sequence per element.
#ifdef CASV_GEMM_PROF
__device__ unsigned long long g_skinny_prof[8];
void skinny_prof_dump() {
    unsigned long long h[8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_skinny_prof), sizeof(h));
    if (h[5]) fprintf(stderr, "skinny_prof (LSTM, 64x128): %llu workgroups: prologue %.0f cyc, stages %.1f cyc/stage over %.1f stages, epilogue %.0f cyc, workgroup %.0f cyc\n",
                      h[5], (double)h[0] / h[5], (double)h[1] / (double)h[2], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5]);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_prof), z, sizeof(z));
}
#endif

template <int EPI, int RB>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(const GemmBatch batch) {
#ifdef CASV_GEMM_PROF
    const unsigned long long sp0 = __builtin_amdgcn_s_memtime();
    unsigned long long sp1 = sp0, sp2 = sp0;
#endif
    constexpr int SBM = 32 * RB;
    constexpr int STAGE_FLOATS = (SBM + SBN) * SLD;          // A rows, then B rows
    __shared__ __attribute__((aligned(16))) float s_stage[2 * STAGE_FLOATS];      // 46 / 55 KB; the epilogue's gate exchange reuses it
    float (*s_gate)[16][64] = reinterpret_cast<float (*)[16][64]>(s_stage);
    const GemmArgs& g = batch.g[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int step = __builtin_amdgcn_readfirstlane(g.step_ptr ? *g.step_ptr : g.step_imm);
    const int nbn = (g.N + SBN - 1) / SBN;
    const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;      // consecutive workgroups (= different XCDs) take different B panels
    const int m0 = bm * SBM, n0 = bn * SBN;
    if (m0 >= g.M) return;
    if (g.nact) {           // every wave looks at the same counts: a uniform exit ahead of the first barrier
        const int mlast = (m0 + SBM < g.M ? m0 + SBM : g.M) - 1;
        const int l0 = m0 / g.nact_group, l1 = mlast / g.nact_group;
        int alive = 0;
        for (int l = l0 + lane; l <= l1; l += 64) alive |= g.nact[l] > (l == l0 ? m0 - l0 * g.nact_group : 0);
        if (!__any(alive)) return;
    }

    // staging role of this thread: A row tid >> 3 (of 32), B rows (tid >> 3) + 32 i, 16 bytes at k = 4 * (tid & 7) of every stage
    const int srow = tid >> 3, sk = 4 * (tid & 7);
    // (segment pointers of A row srow of row block 0; row block r adds its own row's offset: drow[r][segment])
    const float* ap0; const float* ap1; const float* ap2;
    int tiles0 = 0, tiles1 = 0, tiles2 = 0;                      // in stages of 32 k
    int mrow = m0 + srow; mrow = mrow < g.M ? mrow : g.M - 1;
    long long drow[RB][3];
#define CASV_SETUP_SEG(S, AP, TILES)                                                             \
    if (g.nseg > S && !(g.a[S].skip_first && step == 0 && !g.a[S].first_base)) {                 \
        const Seg& sg = g.a[S];                                                                  \
        const bool first = sg.first_base && step == 0;                                           \
        const float* base = first ? sg.first_base                                                \
            : sg.base + (long long)(step * sg.step_mul + sg.step_add) * sg.slot_stride;          \
        const int rid = (sg.rows && !first) ? sg.rows[mrow] : mrow;                              \
        AP = base + (long long)rid * sg.ld + sk;                                                 \
        TILES = sg.width / SK2;                                                                  \
        _Pragma("unroll") for (int r = 0; r < RB; ++r) {                                         \
            int mr = m0 + srow + 32 * r; mr = mr < g.M ? mr : g.M - 1;                           \
            const int rr = (sg.rows && !first) ? sg.rows[mr] : mr;                               \
            drow[r][S] = (long long)(rr - rid) * sg.ld * 4;                                      \
        }                                                                                        \
    } else {                                                                                     \
        AP = nullptr;                                                                            \
        _Pragma("unroll") for (int r = 0; r < RB; ++r) drow[r][S] = 0;                           \
    }
    CASV_SETUP_SEG(0, ap0, tiles0)
    CASV_SETUP_SEG(1, ap1, tiles1)
    CASV_SETUP_SEG(2, ap2, tiles2)
#undef CASV_SETUP_SEG
    const int c0 = __builtin_amdgcn_readfirstlane(tiles0), c1 = __builtin_amdgcn_readfirstlane(tiles0 + tiles1);
    const int ntiles_all = __builtin_amdgcn_readfirstlane(tiles0 + tiles1 + tiles2);
    const int nsplit = gridDim.z;
    const int per = (ntiles_all + nsplit - 1) / nsplit;
    const int kt_begin = blockIdx.z * per;
    const int ntiles = ntiles_all - kt_begin < per ? (ntiles_all - kt_begin > 0 ? ntiles_all - kt_begin : 0) : per;
    const int koff0 = g.a[0].koff, koff1 = g.a[1].koff, koff2 = g.a[2].koff;
    const long long d1 = (long long)((const char*)ap1 - (const char*)ap0), d2 = (long long)((const char*)ap2 - (const char*)ap0);

    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ncol = n0 + srow + 32 * i; ncol = ncol < g.N ? ncol : g.N - 1;
        bp[i] = g.Bt + (long long)ncol * g.Ktot + sk;
    }

    struct GStage { f32x4 a[RB], b[4]; };
    auto load_stage = [&](GStage& gs, int kt_rel) {
        const int kt = kt_rel + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * SK2;
        const char* pa = (const char*)ap0 + (d1 & m1) + (d2 & m2) + (long long)ko * (SK2 * 4);
        gs.a[0] = *reinterpret_cast<const f32x4*>(pa);
#pragma unroll
        for (int r = 1; r < RB; ++r)
            gs.a[r] = *reinterpret_cast<const f32x4*>(pa + ((drow[r][0] & ~(m1 | m2)) | (drow[r][1] & m1) | (drow[r][2] & m2)));
#pragma unroll
        for (int i = 0; i < 4; ++i) gs.b[i] = *reinterpret_cast<const f32x4*>(bp[i] + kb);
    };
    // The same loads hidden from the compiler's wait bookkeeping (steady state): two register sets alternate, a stage's 4 + RB
    // loads have TWO stage times to arrive, and the wait in front of its LDS store is a counted vmcnt(4 + RB) that leaves the next
    // stage's loads in flight (hipcc's own wait there is a vmcnt(0): round 2's loop kept one stage in registers and paid what
    // was left of the L2 latency after one stage of 16 MFMAs -- 1024 cycles -- in every iteration).
    auto load_stage_asm = [&](GStage& gs, int kt_rel) {
        const int kt = kt_rel + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * SK2;
        const char* pa = (const char*)ap0 + (d1 & m1) + (d2 & m2) + (long long)ko * (SK2 * 4);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.a[0]) : "v"(pa));
#pragma unroll
        for (int r = 1; r < RB; ++r) {
            const char* par = pa + ((drow[r][0] & ~(m1 | m2)) | (drow[r][1] & m1) | (drow[r][2] & m2));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.a[r]) : "v"(par));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* pb = bp[i] + kb;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.b[i]) : "v"(pb));
        }
    };
    // Steady state: RUNNING operand pointers, advanced by one stage per request and re-based where the request stream crosses into
    // the next K segment (a wave-uniform branch, twice per K loop at most) -- gemm.hip's scheme, round 4: the mask arithmetic above
    // is ~20 vector and ~25 scalar instructions per stage that a wave with the matrix pipe to itself issues between its own MFMAs.
    const char* ra = nullptr; long long rdl[RB]; const float* rbp[4]; int rleft = 0, rseg = 0;
    auto run_set = [&](int kt_rel) {
        const int kt = kt_rel + kt_begin;
        const long long m1 = (kt >= c0 && kt < c1) ? -1LL : 0LL, m2 = (kt >= c1) ? -1LL : 0LL;
        const int ko = kt - ((int)m1 & c0) - ((int)m2 & c1);
        const int kb = koff0 + ((int)m1 & (koff1 - koff0)) + ((int)m2 & (koff2 - koff0)) + ko * SK2;
        ra = (const char*)ap0 + (d1 & m1) + (d2 & m2) + (long long)ko * (SK2 * 4);
#pragma unroll
        for (int r = 0; r < RB; ++r) rdl[r] = (drow[r][0] & ~(m1 | m2)) | (drow[r][1] & m1) | (drow[r][2] & m2);
#pragma unroll
        for (int i = 0; i < 4; ++i) rbp[i] = bp[i] + kb;
        rseg = kt >= c1 ? 2 : kt >= c0 ? 1 : 0;
        rleft = (rseg == 0 ? c0 : rseg == 1 ? c1 : ntiles_all) - kt;
    };
    auto load_stage_run = [&](GStage& gs) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.a[0]) : "v"(ra));
#pragma unroll
        for (int r = 1; r < RB; ++r) {
            const char* par = ra + rdl[r];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.a[r]) : "v"(par));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.b[i]) : "v"(rbp[i]));
    };
    auto run_advance = [&]() {
        ra += SK2 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) rbp[i] += SK2;
        if (--rleft == 0) {
            asm volatile("" ::: "memory");                  // (stays a branch)
            if (rseg == 0 && c1 > c0) {
                rseg = 1; rleft = c1 - c0; ra = (const char*)ap1;
#pragma unroll
                for (int r = 0; r < RB; ++r) rdl[r] = drow[r][1];
#pragma unroll
                for (int i = 0; i < 4; ++i) rbp[i] += koff1 - koff0 - c0 * SK2;
            } else if (rseg <= 1 && ntiles_all > c1) {
                const int kprev = rseg == 0 ? koff0 + c0 * SK2 : koff1 + (c1 - c0) * SK2;
                rseg = 2; rleft = ntiles_all - c1; ra = (const char*)ap2;
#pragma unroll
                for (int r = 0; r < RB; ++r) rdl[r] = drow[r][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) rbp[i] += koff2 - kprev;
            } else {
                rleft = 1 << 30;
            }
        }
    };
    auto store_stage = [&](const GStage& gs, int buf) {
        float* sa = s_stage + buf * STAGE_FLOATS + srow * SLD + sk;
#pragma unroll
        for (int r = 0; r < RB; ++r) *reinterpret_cast<f32x4*>(sa + 32 * r * SLD) = gs.a[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sa + (SBM + 32 * i) * SLD) = gs.b[i];
    };
    f32x16 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;
    // fragments of one stage: two 16-k halves, in each lane half lh contracts k = 4 lh + 8 j + i in instruction (j, i) --
    // the k order of gemm.hip
    const int a_off = l31 * SLD + 4 * lh, b_off = (SBM + wave * 32 + l31) * SLD + 4 * lh;
    auto compute = [&](int buf) {
        const float* base = s_stage + buf * STAGE_FLOATS;
        f32x4 fa[RB][4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < RB; ++r) fa[r][q] = *reinterpret_cast<const f32x4*>(base + a_off + 32 * r * SLD + 8 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(base + b_off + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r][q][i], fb[q][i], acc[r], 0, 0, 0);
    };

    // LSTM: wave w finishes rows (r & 3, r >> 2 == w) of every row block.  Their previous cell state is requested BEHIND the
    // steady-state loop, under the tail stages (round 4: requested in front of the loop, these compiler-tracked loads were waited
    // for before the first MFMA -- no tracked load may be pending on a path into the loop; gemm.hip's stamps put that wait at
    // 26 000 cycles for its 16 gathers per lane).
    float cpv[RB][4];
    auto load_cell_state = [&]() {
        if (EPI == EPI_LSTM && !g.epi_plain) {
            const bool cfirst = g.c_in.first_base && step == 0;
            const bool czero = g.c_in.skip_first && step == 0 && !cfirst;
            const float* cin = cfirst ? g.c_in.first_base
                : g.c_in.base + (long long)(step * g.c_in.step_mul + g.c_in.step_add) * g.c_in.slot_stride;
            const int u = bn * 32 + l31;
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int m = m0 + 32 * r + q + 8 * wave + 4 * lh;
                    m = m < g.M ? m : g.M - 1;
                    cpv[r][q] = 0.0f;
                    if (!czero) {
                        const int rid = (g.c_in.rows && !cfirst) ? g.c_in.rows[m] : m;
                        cpv[r][q] = cin[(long long)rid * g.c_in.ld + u];
                    }
                }
        }
    };

    // ... and (train step) the precomputed input pre-activations x.K + b of its (row, unit) elements: requested here, under the
    // K loop, instead of as a dependent round trip between the K loop and the cell
    float zpre[RB][4][4];
    const bool has_zin = EPI == EPI_LSTM && !g.epi_plain && g.zinit.base != nullptr;
    if (has_zin) {
        const float* zin0 = g.zinit.base + (long long)(step * g.zinit.step_mul + g.zinit.step_add) * g.zinit.slot_stride;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 32 * r + q + 8 * wave + 4 * lh;
                const float* zr = zin0 + (long long)(m < g.M ? m : g.M - 1) * g.zinit.ld + n0 + l31;
                zpre[r][q][0] = zr[0]; zpre[r][q][1] = zr[32]; zpre[r][q][2] = zr[64]; zpre[r][q][3] = zr[96];
            }
    }

#ifdef CASV_GEMM_PROF
    sp1 = __builtin_amdgcn_s_memtime();
#endif
    if (ntiles > 0) {
        // stage j + 1 goes from registers into the LDS buffer whose readers passed the last barrier, stage j + 3 starts its way
        // from global memory into the same register set, stage j is contracted; one barrier per stage
        GStage g0, g1;                          // even / odd stages
        load_stage(g0, 0);
        store_stage(g0, 0);
        int kt = 0;
        // (the waits name every register of the stage: its loads must have landed before the LDS store reads them)
#define CASV_SK_WAIT(CNT, G)                                                                                       \
        if constexpr (RB == 1) asm volatile("s_waitcnt vmcnt(" CNT ")" : "+v"(G.a[0]), "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3])); \
        else asm volatile("s_waitcnt vmcnt(" CNT ")" : "+v"(G.a[0]), "+v"(G.a[RB - 1]), "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3]));
#define CASV_SK_FULL(G, J)                                                                                         \
        {                                                                                                          \
            if constexpr (RB == 1) { CASV_SK_WAIT("5", G) } else { CASV_SK_WAIT("6", G) }                          \
            store_stage(G, ((J) + 1) & 1);                                                                         \
            load_stage_run(G);                                                                                     \
            compute((J) & 1);                                                                                      \
            run_advance();                                                                                         \
            __syncthreads();                                                                                       \
        }
#define CASV_SK_STEP(G, J)                                                                                         \
        {                                                                                                          \
            if ((J) + 1 < ntiles) store_stage(G, ((J) + 1) & 1);                                                   \
            if ((J) + 3 < ntiles) load_stage(G, (J) + 3);                                                          \
            if ((J) < ntiles) compute((J) & 1);                                                                    \
            __syncthreads();                                                                                       \
        }
        if (ntiles > 4) {
            // (no compiler-tracked tile load may be pending on any path into the loop: it would put a vmcnt(0) at the loop head)
            load_stage_asm(g1, 1); load_stage_asm(g0, 2);
            run_set(3);                                     // the first steady-state stage requests stage 3
            __syncthreads();
            for (; kt + 4 < ntiles; kt += 2) {
                CASV_SK_FULL(g1, kt)
                CASV_SK_FULL(g0, kt + 1)
            }
            CASV_SK_WAIT("0", g0)
            CASV_SK_WAIT("0", g1)
            load_cell_state();
        } else {
            if (ntiles > 1) load_stage(g1, 1);
            if (ntiles > 2) load_stage(g0, 2);
            load_cell_state();
            __syncthreads();
        }
        for (; kt + 1 < ntiles; kt += 2) {
            CASV_SK_STEP(g1, kt)
            CASV_SK_STEP(g0, kt + 1)
        }
        if (kt < ntiles) CASV_SK_STEP(g1, kt)
#undef CASV_SK_STEP
#undef CASV_SK_FULL
#undef CASV_SK_WAIT
    } else {
        load_cell_state();
    }

#ifdef CASV_GEMM_PROF
    sp2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue ----
    if (EPI == EPI_PLAIN || g.epi_plain) {
        float* cbase = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        const int n = n0 + wave * 32 + l31;
        if (nsplit == 1 && !g.accumulate && m0 + SBM <= g.M && n0 + SBN <= g.N) {
            // full tile, plain stores: nothing between the stores that the compiler would wait at (gemm.hip's epilogue, round 4)
            const float b = g.bias ? g.bias[n] : 0.0f;
            float* cb = cbase + (long long)(m0 + 4 * lh) * g.out.ld + n;
            // (16-byte stores through an in-quad transpose, as in gemm.hip's plain epilogue: measured here too, round 5 -- no difference
            // for the logits launches of c3 or the page call)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(long long)(32 * rb + (r & 3) + 8 * (r >> 2)) * g.out.ld] = acc[rb][r] + b;
        } else if (n < g.N) {
            const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < g.M) {
                        float* dst = cbase + (long long)m * g.out.ld + n;
                        if (nsplit > 1) atomicAdd(dst, acc[rb][r] + (blockIdx.z == 0 ? b : 0.0f));
                        else *dst = g.accumulate ? (*dst + acc[rb][r] + b) : (acc[rb][r] + b);
                    }
                }
        }
    } else {
        const int u = bn * 32 + l31;
        const float bi = g.bias ? g.bias[n0 + l31] : 0.f, bf_ = g.bias ? g.bias[n0 + 32 + l31] : 0.f;
        const float bg = g.bias ? g.bias[n0 + 64 + l31] : 0.f, bo = g.bias ? g.bias[n0 + 96 + l31] : 0.f;
        float* cout = g.c_out.base + (long long)(step * g.c_out.step_mul + g.c_out.step_add) * g.c_out.slot_stride;
        float* hout = g.out.base + (long long)(step * g.out.step_mul + g.out.step_add) * g.out.slot_stride;
        float* gout = g.gates_out.base
            ? g.gates_out.base + (long long)(step * g.gates_out.step_mul + g.gates_out.step_add) * g.gates_out.slot_stride
            : nullptr;
        float* hout2 = g.out2.base ? g.out2.base + (long long)(step * g.out2.step_mul + g.out2.step_add) * g.out2.slot_stride : nullptr;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            // gate w of every (row, unit) of the row block -> LDS; then wave w takes accumulator rows r = 4w .. 4w+3 of all four gates
            if (rb) __syncthreads();                                          // (the exchange buffer is read out)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_gate[wave][r][lane] = acc[rb][r];  // (every wave is past the K loop's last barrier)
            __syncthreads();
            float z[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) z[q][c] = s_gate[c][4 * wave + q][lane];
            if (has_zin) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { z[q][0] += zpre[rb][q][0]; z[q][1] += zpre[rb][q][1]; z[q][2] += zpre[rb][q][2]; z[q][3] += zpre[rb][q][3]; }
            }
            if (m0 + 32 * RB <= g.M && !hout2 && !gout) {
                // full tile, inference outputs only: the four cells first, then the eight stores, no control flow in between (a branch
                // per row makes the compiler wait for the stores of the row before: gemm.hip's epilogue, round 4)
                float hv[4], cv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const LstmCellOut cell = lstm_cell(z[q][0] + bi, z[q][1] + bf_, z[q][2] + bg, z[q][3] + bo, cpv[rb][q]);
                    hv[q] = cell.h; cv[q] = cell.c;
                }
                const long long mb = m0 + 32 * rb + 8 * wave + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cout[(mb + q) * g.c_out.ld + u] = cv[q];
                    hout[(mb + q) * g.out.ld + u] = hv[q];
                }
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + 32 * rb + q + 8 * wave + 4 * lh;
                if (m < g.M) {
                    const LstmCellOut cell = lstm_cell(z[q][0] + bi, z[q][1] + bf_, z[q][2] + bg, z[q][3] + bo, cpv[rb][q]);
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
    if (EPI == EPI_LSTM && RB == 2 && threadIdx.x == 0) {
        const unsigned long long sp3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_skinny_prof[0], sp1 - sp0); atomicAdd(&g_skinny_prof[1], sp2 - sp1); atomicAdd(&g_skinny_prof[2], (unsigned long long)ntiles);
        atomicAdd(&g_skinny_prof[3], sp3 - sp2); atomicAdd(&g_skinny_prof[4], sp3 - sp0); atomicAdd(&g_skinny_prof[5], 1ull);
    }
#endif
}

// rows = 32 or 64 per tile (gemm.hip's plan)
void launch_gemm_skinny(int epi, const GemmBatch& b, int ksplit, int rows, hipStream_t stream) {
    int blocks = 0;
    for (int j = 0; j < b.count; ++j) {
        const int nb = ((b.g[j].M + rows - 1) / rows) * ((b.g[j].N + SBN - 1) / SBN);
        blocks = nb > blocks ? nb : blocks;
    }
    const dim3 grid(blocks, b.count, ksplit);
    if (rows == 64) {
        if (epi == EPI_LSTM) hipLaunchKernelGGL((gemm_skinny_kernel<EPI_LSTM, 2>), grid, dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((gemm_skinny_kernel<EPI_PLAIN, 2>), grid, dim3(256), 0, stream, b);
    } else {
        if (epi == EPI_LSTM) hipLaunchKernelGGL((gemm_skinny_kernel<EPI_LSTM, 1>), grid, dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((gemm_skinny_kernel<EPI_PLAIN, 1>), grid, dim3(256), 0, stream, b);
    }
}

}  // namespace casv
