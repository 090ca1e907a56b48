// One backward time step of an LSTM layer as ONE launch: the cell's pointwise backward (gate derivatives dZ, the running
// dL/dc) fused into the data GEMM dRec(t) = dZ(t) . Wr that carries the gradient to the previous step (oracle/train.py's
// lstm_backward; keras' LSTM gradient, seq2seq.py:268-272).  train.hip used to launch lstm_bwd_kernel and the split-K
// gemm_skinny kernel per step -- ~6.6 us + ~22 us + two launch gaps, 509 times per train step.
//
// Shape of the fusion.  The GEMM's K dimension is the gate axis in the packed order [unit group of 32][gate][unit], so a
// split-K share of whole unit groups needs exactly the cells of those units: workgroup (row block of 32, column tile of 128,
// K share) computes dZ for its 32 rows x 32 units x 4 gates one unit group ahead of the MFMAs that contract it, straight into
// the LDS image the A fragments are read from (four 32-k stages per group) -- the A operand never makes a round trip through
// memory.  The column tiles of a row block repeat that pointwise work (4x at N = 512: L2 hits); the tile bn == 0 also stores dZ
// (kept for the weight gradients) and dL/dc(t-1).  dL/dc ping-pongs between two buffers, because the other column tiles still
// read the old values while the first one writes.
//
// K loop: gemm_skinny.hip's (B stages of 32 k double-buffered in LDS, two register sets in flight, counted vmcnt waits), with
// the cells' inputs (11 x 16 B per thread and unit group) requested two unit groups ahead as untracked loads that ride in
// the same in-order queue: the counted waits leave them in flight too (vmcnt(15) where they are the youngest entries).
// Split-K shares meet in float atomics on the zeroed output, as before.
#include "common.h"
#include "train_kernels.h"
#include <math.h>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int FBM = 32, FBN = 128, FBK = 32;
constexpr int FBLD = FBK + 4;                   // B stage row stride
constexpr int FALD = 128 + 4;                   // A group row stride (one unit group = 128 k)
constexpr int FB_STAGE = FBN * FBLD, FA_GROUP = FBM * FALD;

struct PwIn { f32x4 a, mk, b, c, gi, gf, gg, go, cell, cp, dc; };
struct PwOut { f32x4 zi, zf, zg, zo, dc; };
}

__global__ __launch_bounds__(256, 2) void lstm_bwd_gemm_kernel(const BwdStepBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];          // (more than the 64 KB a kernel may claim statically)
    float* const s_a = s_dyn;                                              // 33.8 KB: two unit groups of dZ
    float* const s_b = s_dyn + 2 * FA_GROUP;                               // 36.9 KB: two stages of Wr
    const BwdStepJob& jb = batch.j[blockIdx.y];
    const LstmBwdArgs& p = jb.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = p.W, rows = p.rows, N = jb.N;
    const long long K = 4LL * W;
    const int nbn = (N + FBN - 1) / FBN;
    const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;
    const int m0 = bm * FBM, n0 = bn * FBN;
    if (m0 >= rows) return;
    const int nsplit = gridDim.z;
    const int gper = (W / 32) / nsplit;                 // unit groups of this K share (the launcher makes it exact)
    const int g_begin = blockIdx.z * gper;
    const int nst = 4 * gper;
    const bool writer = bn == 0;

    const int srow = tid >> 3, su = 4 * (tid & 7);
    const bool row_ok = m0 + srow < rows;
    const int mrow = row_ok ? m0 + srow : rows - 1;
    // ---- the cells' inputs: unit u = (g_begin + g) * 32 + su .. + 3 of row mrow ----
    const bool a_on = p.a != nullptr, m_on = p.mask_a != nullptr, b_on = p.b != nullptr, c_on = p.c != nullptr, cp_on = p.c_prev != nullptr;
    const float* const safe = p.cell + (long long)mrow * W;           // any readable row of W floats stands in for an absent input
    const int u0 = g_begin * 32 + su;
    const float* pa = (a_on ? p.a + (long long)mrow * p.lda : safe) + u0;
    const float* pm = (m_on ? p.mask_a : safe) + u0;
    const float* pb = (b_on ? p.b + (long long)mrow * p.ldb : safe) + u0;
    const float* pc = (c_on ? p.c + (long long)mrow * p.ldc : safe) + u0;
    const float* pg = p.gates + (long long)mrow * K + (long long)g_begin * 128 + su;
    const float* pcell = p.cell + (long long)mrow * W + u0;
    const float* pcp = (cp_on ? p.c_prev + (long long)mrow * p.ld_cprev : safe) + u0;
    const float* pdc = jb.dc_in + (long long)mrow * W + u0;
    float* qz = p.dz + (long long)mrow * K + (long long)g_begin * 128 + su;
    float* qdc = p.dc + (long long)mrow * W + u0;

#define CASV_LD16(DST, PTR, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF))
    auto load_in = [&](PwIn& in, int g) {              // 11 loads
        const float* xa = pa + g * 32; const float* xm = pm + g * 32; const float* xb = pb + g * 32; const float* xc = pc + g * 32;
        const float* xg = pg + g * 128; const float* xcell = pcell + g * 32; const float* xcp = pcp + g * 32; const float* xdc = pdc + g * 32;
        CASV_LD16(in.a, xa, 0); CASV_LD16(in.mk, xm, 0); CASV_LD16(in.b, xb, 0); CASV_LD16(in.c, xc, 0);
        CASV_LD16(in.gi, xg, 0); CASV_LD16(in.gf, xg, 128); CASV_LD16(in.gg, xg, 256); CASV_LD16(in.go, xg, 384);
        CASV_LD16(in.cell, xcell, 0); CASV_LD16(in.cp, xcp, 0); CASV_LD16(in.dc, xdc, 0);
    };
#define CASV_IN_REGS(IN) "+v"(IN.a), "+v"(IN.mk), "+v"(IN.b), "+v"(IN.c), "+v"(IN.gi), "+v"(IN.gf), "+v"(IN.gg), "+v"(IN.go), "+v"(IN.cell), "+v"(IN.cp), "+v"(IN.dc)
    // lstm_bwd_kernel's arithmetic (train_kernels.hip), four units at a time
    auto pointwise = [&](const PwIn& in, PwOut& o) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float dh = 0.f;
            if (a_on) dh += in.a[e] * (m_on ? in.mk[e] : 1.0f);
            if (b_on) dh += in.b[e];
            if (c_on) dh += in.c[e];
            const float ig = in.gi[e], fg = in.gf[e], gg = in.gg[e], og = in.go[e];
            const float cc = in.cell[e];
            const float cp = cp_on ? in.cp[e] : 0.0f;
            const float tc = tanhf(cc);
            const float dov = dh * tc;
            const float dct = dh * og * (1.0f - tc * tc) + in.dc[e];
            o.zi[e] = dct * gg * ig * (1.0f - ig);
            o.zf[e] = dct * cp * fg * (1.0f - fg);
            o.zg[e] = dct * ig * (1.0f - gg * gg);
            o.zo[e] = dov * og * (1.0f - og);
            o.dc[e] = dct * fg;
        }
    };
    auto store_group = [&](const PwOut& o, int g) {   // A image of unit group g: [row][gate * 32 + unit]
        float* sa = s_a + (g & 1) * FA_GROUP + srow * FALD + su;
        *reinterpret_cast<f32x4*>(sa) = o.zi; *reinterpret_cast<f32x4*>(sa + 32) = o.zf;
        *reinterpret_cast<f32x4*>(sa + 64) = o.zg; *reinterpret_cast<f32x4*>(sa + 96) = o.zo;
        if (writer && row_ok) {
            float* z = qz + g * 128;
            *reinterpret_cast<f32x4*>(z) = o.zi; *reinterpret_cast<f32x4*>(z + 32) = o.zf;
            *reinterpret_cast<f32x4*>(z + 64) = o.zg; *reinterpret_cast<f32x4*>(z + 96) = o.zo;
            *reinterpret_cast<f32x4*>(qdc + g * 32) = o.dc;
        }
    };

    // ---- B operand: rows n0 + srow + 32 i of Bt [N][K], 16 bytes at k = sk of every stage ----
    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ncol = n0 + srow + 32 * i; ncol = ncol < N ? ncol : N - 1;
        bp[i] = jb.Bt + (long long)ncol * K + (long long)g_begin * 128 + su;
    }
    struct BStage { f32x4 b[4]; };
    auto load_b = [&](BStage& gs, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* pbt = bp[i] + kt * FBK;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gs.b[i]) : "v"(pbt));
        }
    };
    auto store_b = [&](const BStage& gs, int buf) {
        float* sb = s_b + buf * FB_STAGE + srow * FBLD + su;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + 32 * i * FBLD) = gs.b[i];
    };
#define CASV_B_REGS(G) "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3])

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int a_off = l31 * FALD + 4 * lh, b_off = (wave * 32 + l31) * FBLD + 4 * lh;
    auto compute = [&](int J) {           // stage J: gate J & 3 of unit group J >> 2
        const float* abase = s_a + ((J >> 2) & 1) * FA_GROUP + (J & 3) * 32 + a_off;
        const float* bbase = s_b + (J & 1) * FB_STAGE + b_off;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(abase + 8 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(bbase + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][i], acc, 0, 0, 0);
    };

    // ---- prologue: group 0 computed, group 1's inputs and B stages 1, 2 in flight ----
    PwIn in;
    PwOut out;
    BStage g0, g1;
    load_in(in, 0);
    load_b(g0, 0);
    asm volatile("s_waitcnt vmcnt(0)" : CASV_IN_REGS(in));
    asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));
    pointwise(in, out);
    store_group(out, 0);
    store_b(g0, 0);
    // (what leaves next -- group 1's inputs, B stages 1 and 2 -- leaves inside the branch that consumes it: see below)
#define CASV_FB_START(IN1)                                                                                             \
    if (IN1) load_in(in, 1);      /* older than B(1), B(2): complete once the first counted wait has passed */          \
    load_b(g1, 1);                                                                                                     \
    load_b(g0, 2);                                                                                                     \
    __syncthreads();

    // One unit group = four stages.  Stage J: stage J + 1's B goes from registers into LDS, stage J + 3's B leaves for the same
    // register set, stage J is contracted.  In the third stage of group g the cells of group g + 1 are computed (their inputs
    // were requested a whole group earlier) and the inputs of group g + 2 leave -- AFTER that stage's B loads, so that they are
    // the youngest entries of the queue and the next two counted waits can leave them in flight.
    //   C0 / C3: loads that may stay in flight at the waits of the first / last stage (4 = the next B stage, + 11 = a group's inputs)
#define CASV_FB_STAGE(G, J, CNT)                                                                                       \
    {                                                                                                                  \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" : CASV_B_REGS(G));                                                    \
        store_b(G, ((J) + 1) & 1);                                                                                     \
    }
#define CASV_FB_GROUP(C0, C3, PW, IN)                                                                                  \
    {                                                                                                                  \
        const int J = 4 * g;                                                                                           \
        CASV_FB_STAGE(g1, J, C0)      load_b(g1, J + 3); compute(J);     __syncthreads();                              \
        CASV_FB_STAGE(g0, J + 1, 4)   load_b(g0, J + 4); compute(J + 1); __syncthreads();                              \
        CASV_FB_STAGE(g1, J + 2, 4)                                                                                    \
        if (PW) { asm volatile("" : CASV_IN_REGS(in)); pointwise(in, out); store_group(out, g + 1); }                  \
        load_b(g1, J + 5);                                                                                             \
        if (IN) load_in(in, g + 2);                                                                                    \
        compute(J + 2); __syncthreads();                                                                               \
        CASV_FB_STAGE(g0, J + 3, C3)  load_b(g0, J + 6); compute(J + 3); __syncthreads();                              \
    }
    // last group: nothing new is requested
#define CASV_FB_LAST()                                                                                                 \
    {                                                                                                                  \
        const int J = 4 * g;                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1));                                                          \
        asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));                                                          \
        store_b(g1, (J + 1) & 1); compute(J); __syncthreads();                                                         \
        store_b(g0, (J + 2) & 1);                                                                                      \
        load_b(g1, J + 3);                                                                                             \
        compute(J + 1); __syncthreads();                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1));                                                          \
        store_b(g1, (J + 3) & 1); compute(J + 2); __syncthreads();                                                     \
        compute(J + 3);                                                                                                \
    }
    // Every path from the first request to its last group is straight-line code (but for the loop), chosen BEFORE anything is
    // requested: a register with a load in flight must not meet a control-flow merge or a branch, where the compiler may
    // copy it -- before the data has arrived.
    int g = 0;
    if (gper >= 3) {
        CASV_FB_START(true)
        CASV_FB_GROUP(4, 15, true, true)
        for (g = 1; g + 2 < gper; ++g) CASV_FB_GROUP(15, 15, true, true)
        CASV_FB_GROUP(15, 4, true, false)              // g = gper - 2
        g = gper - 1;
        CASV_FB_LAST()
    } else if (gper == 2) {
        CASV_FB_START(true)
        CASV_FB_GROUP(4, 4, true, false)
        g = 1;
        CASV_FB_LAST()
    } else {
        CASV_FB_START(false)
        CASV_FB_LAST()
    }
#undef CASV_FB_START
#undef CASV_FB_LAST
#undef CASV_FB_GROUP
#undef CASV_FB_STAGE
#undef CASV_B_REGS
#undef CASV_IN_REGS
#undef CASV_LD16
    (void)nst;

    // ---- epilogue: the K shares meet in the zeroed output ----
    const int n = n0 + wave * 32 + l31;
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < rows) {
                float* dst = jb.out + (long long)m * jb.ld_out + n;
                if (nsplit > 1) atomicAdd(dst, acc[r]);
                else *dst = acc[r];
            }
        }
    }
}

// K shares: as many as keep ~2 workgroups per CU busy, whole unit groups each
void launch_lstm_bwd_gemm(const BwdStepBatch& b, hipStream_t stream) {
    if (b.count < 1) return;
    int blocks = 0;
    for (int j = 0; j < b.count; ++j) {
        const int nb = ((b.j[j].p.rows + FBM - 1) / FBM) * ((b.j[j].N + FBN - 1) / FBN);
        blocks = nb > blocks ? nb : blocks;
    }
    const int groups = b.j[0].p.W / 32;
    int ksplit = 1;
    for (int s = 2; s <= groups && s <= 8; ++s)
        if (groups % s == 0 && (long long)blocks * b.count * s <= 640) ksplit = s;
    constexpr size_t lds = (size_t)(2 * FA_GROUP + 2 * FB_STAGE) * sizeof(float);
    static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_gemm_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)lds_ok;          // (a refusal surfaces as the launch error the caller checks)
    hipLaunchKernelGGL(lstm_bwd_gemm_kernel, dim3(blocks, b.count, ksplit), dim3(256), lds, stream, b);
}

}  // namespace casv
