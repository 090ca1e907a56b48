// The attention cell's forward recurrence of the train step (seq2seq.py:343-349, attention.py:526-575, teacher-forced) as
// ONE launch.  train.hip walks it with four launches per time step -- attention query GEMM, attention rows, cell input rows,
// LSTM step GEMM: ~62 us of kernels and four launch boundaries, 102 times.  Here (row block of 32) x (unit group of 32)
// workgroups, one per CU, stay resident and all play the same three parts in every step, handing rows to the other 15 workgroups
// of their row block through memory (handoff.h):
//   H  behind h(t-1) of the row block: its 32 rows, all W/32 stages at once into registers, are contracted with the unit
//      group's recurrent weight panel (the h columns of Wr: 128 gate rows) AND, riding on the same A stages, with 32 rows of W_a:
//      the workgroup's 32 columns of the attention query h.W_a + b (every wave a k quarter of each stage, summed through LDS);
//   A  behind the row block's query: two of the row block's rows per workgroup, one wave each -- attention_row (row_kernels.h),
//      the per-step kernel's code; the context goes, masked, straight into the cell's input rows;
//   C  behind the row block's context: its 32 rows are contracted with the ctx columns of Wr on top of H's accumulators, then
//      the cell (x.Wx + b precomputed; cell state in registers for the whole sequence); h(t) goes out twice: time-major for the
//      loss and into the next step's input rows.
// The sums: gate pre-activations run over [h | ctx] (the per-step launch: [ctx | h]) and the query's k quarters meet in LDS
// (per-step: two split-K shares), so the results agree with the per-step path to fp32 rounding, not bit for bit.
#include "common.h"
#include "handoff.h"
#include "row_kernels.h"
#include "train_kernels.h"
#include <map>
#include <mutex>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TBM = 32, TBN = 128, TQN = 32;             // rows, gate columns, query columns of a workgroup
constexpr int TK2 = 32, TLD = TK2 + 4;
constexpr int TSTAGE = (TBM + TBN + TQN) * TLD;          // A rows, Wr rows, W_a rows of one stage
}

template <int NT>        // W / 32 = C / 32: unit groups per row block; stages per K half
__global__ __launch_bounds__(256, 1) void train_attention_cell_kernel(const TopRecArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_stage[2 * TSTAGE];      // 54 KB; the epilogues' exchanges reuse it
    __shared__ int s_ok;
    float (*s_gate)[16][64] = reinterpret_cast<float (*)[16][64]>(s_stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int W = NT * 32, C = W, KR = C + W;
    const int B = ra.B, U = ra.U;
    const int nrb = (B + TBM - 1) / TBM;
    const int ug = blockIdx.x % NT, rb = blockIdx.x / NT;
    const int m0 = rb * TBM, n0 = ug * TBN;
    unsigned* const cnt_h = ra.counters + (long long)(rb * 3 + 0) * 32;
    unsigned* const cnt_q = ra.counters + (long long)(rb * 3 + 1) * 32;
    unsigned* const cnt_c = ra.counters + (long long)(rb * 3 + 2) * 32;
    unsigned* const abort_w = ra.counters + (long long)nrb * 3 * 32;

    const int srow = tid >> 3, sk = 4 * (tid & 7);
    int mrow = m0 + srow; mrow = mrow < B ? mrow : B - 1;
    // weight rows staged by this thread: gate rows n0 + srow + 32 i of Wr [4W][KR], query row ug * 32 + srow of W_a^T [W][W]
    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bp[i] = ra.Wr + (long long)(n0 + srow + 32 * i) * KR + sk;
    const float* qp = ra.WaT + (long long)(ug * TQN + srow) * W + sk;

    struct BStage { f32x4 b[4], q; };
#define CASV_LOAD_BH(G, KT)  /* h half: columns C + k of Wr, and W_a */                                                  \
    {                                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(bp[0]), "n"((C + (KT) * TK2) * 4));  \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[1]) : "v"(bp[1]), "n"((C + (KT) * TK2) * 4));  \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[2]) : "v"(bp[2]), "n"((C + (KT) * TK2) * 4));  \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[3]) : "v"(bp[3]), "n"((C + (KT) * TK2) * 4));  \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.q) : "v"(qp), "n"((KT) * TK2 * 4));            \
    }
#define CASV_LOAD_BC(G, KT)  /* ctx half: columns k of Wr */                                                             \
    {                                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(bp[0]), "n"((KT) * TK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[1]) : "v"(bp[1]), "n"((KT) * TK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[2]) : "v"(bp[2]), "n"((KT) * TK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[3]) : "v"(bp[3]), "n"((KT) * TK2 * 4));      \
    }
#define CASV_LOAD_A(J)                                                                                                  \
    if constexpr ((J) < NT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(areg[(J) < NT ? (J) : 0]) : "v"(arow), "n"((J) * TK2 * 4));
#define CASV_B_REGS(G) "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3])
    auto store_a = [&](const f32x4& a, int buf) {
        *reinterpret_cast<f32x4*>(s_stage + buf * TSTAGE + srow * TLD + sk) = a;
    };
    auto store_b = [&](const BStage& gs, int buf, bool with_q) {
        float* sb = s_stage + buf * TSTAGE + (TBM + srow) * TLD + sk;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + 32 * i * TLD) = gs.b[i];
        if (with_q) *reinterpret_cast<f32x4*>(sb + TBN * TLD) = gs.q;
    };
    const int a_off = l31 * TLD + 4 * lh, b_off = (TBM + wave * 32 + l31) * TLD + 4 * lh, q_off = (TBM + TBN + l31) * TLD + 4 * lh;
    f32x16 acc, accq;
    auto compute = [&](int buf, bool with_q) {
        const float* base = s_stage + buf * TSTAGE;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(base + a_off + 8 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(base + b_off + 8 * q);
        }
        if (with_q) {       // this wave's k quarter of the stage for the query columns
            const f32x4 faq = *reinterpret_cast<const f32x4*>(base + a_off + 8 * wave);
            const f32x4 fq = *reinterpret_cast<const f32x4*>(base + q_off + 8 * wave);
#pragma unroll
            for (int i = 0; i < 4; ++i) accq = __builtin_amdgcn_mfma_f32_32x32x2f32(faq[i], fq[i], accq, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][i], acc, 0, 0, 0);
    };

    // cell state of this lane's four (row, unit) elements: rows m0 + q + 8 wave + 4 lh, unit ug * 32 + l31
    const int u = ug * 32 + l31;
    float cst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m = m0 + q + 8 * wave + 4 * lh; m = m < B ? m : B - 1;
        cst[q] = ra.c0 ? ra.c0[(long long)m * W + u] : 0.0f;
    }
    const float qbias = ra.bUW ? ra.bUW[ug * TQN + l31] : 0.0f;

    // One K half: A = 32 rows x W columns at `arow` (written by other workgroups a moment ago: all stages at once, past the L1),
    // B stages double-buffered in LDS, counted waits -- train_persist.hip's loop.  HALF 0: h columns + query rows, 1: ctx columns.
#define CASV_TOP_STAGE(G, J, HALF)                                                                                      \
        if constexpr ((J) < NT) {                                                                                       \
            if constexpr ((J) + 1 < NT) {                                                                               \
                if constexpr ((J) >= 2 && (J) + 2 < NT) {                                                               \
                    if constexpr (HALF == 0) asm volatile("s_waitcnt vmcnt(5)" : CASV_B_REGS(G), "+v"(G.q));            \
                    else asm volatile("s_waitcnt vmcnt(4)" : CASV_B_REGS(G));                                           \
                } else if constexpr ((J) >= 2) {                                                                        \
                    if constexpr (HALF == 0) asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(G), "+v"(G.q));            \
                    else asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(G));                                           \
                }                                                                                                       \
                store_a(areg[(J) + 1 < NT ? (J) + 1 : 0], ((J) + 1) & 1);                                               \
                store_b(G, ((J) + 1) & 1, HALF == 0);                                                                   \
            }                                                                                                           \
            if constexpr ((J) + 3 < NT) { if constexpr (HALF == 0) CASV_LOAD_BH(G, (J) + 3) else CASV_LOAD_BC(G, (J) + 3) } \
            compute((J) & 1, HALF == 0);                                                                                \
            __syncthreads();                                                                                            \
        }
#define CASV_TOP_HALF(HALF, COUNTER, TARGET, SKIPWAIT)                                                                  \
    {                                                                                                                   \
        BStage g0, g1;                                                                                                  \
        if constexpr (HALF == 0) CASV_LOAD_BH(g0, 0) else CASV_LOAD_BC(g0, 0)                                           \
        if constexpr (HALF == 0) asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0), "+v"(g0.q));                      \
        else asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));                                                      \
        store_b(g0, 0, HALF == 0);                                                                                      \
        if constexpr (NT > 1) { if constexpr (HALF == 0) CASV_LOAD_BH(g1, 1) else CASV_LOAD_BC(g1, 1) }                 \
        if constexpr (NT > 2) { if constexpr (HALF == 0) CASV_LOAD_BH(g0, 2) else CASV_LOAD_BC(g0, 2) }                 \
        if (!(SKIPWAIT)) {                                                                                              \
            if (!wait_deps(Dep{COUNTER, (unsigned)(TARGET)}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return; \
        }                                                                                                               \
        f32x4 areg[NT];                                                                                                 \
        CASV_LOAD_A(0) CASV_LOAD_A(1) CASV_LOAD_A(2) CASV_LOAD_A(3) CASV_LOAD_A(4) CASV_LOAD_A(5) CASV_LOAD_A(6) CASV_LOAD_A(7) \
        CASV_LOAD_A(8) CASV_LOAD_A(9) CASV_LOAD_A(10) CASV_LOAD_A(11) CASV_LOAD_A(12) CASV_LOAD_A(13) CASV_LOAD_A(14) CASV_LOAD_A(15) \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[j]));              \
        if constexpr (HALF == 0) {                                                                                      \
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0), "+v"(g0.q));                                           \
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1), "+v"(g1.q));                                           \
        } else {                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));                                                       \
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1));                                                       \
        }                                                                                                               \
        store_a(areg[0], 0);                                                                                            \
        __syncthreads();                                                                                                \
        CASV_TOP_STAGE(g1, 0, HALF) CASV_TOP_STAGE(g0, 1, HALF) CASV_TOP_STAGE(g1, 2, HALF) CASV_TOP_STAGE(g0, 3, HALF)   \
        CASV_TOP_STAGE(g1, 4, HALF) CASV_TOP_STAGE(g0, 5, HALF) CASV_TOP_STAGE(g1, 6, HALF) CASV_TOP_STAGE(g0, 7, HALF)   \
        CASV_TOP_STAGE(g1, 8, HALF) CASV_TOP_STAGE(g0, 9, HALF) CASV_TOP_STAGE(g1, 10, HALF) CASV_TOP_STAGE(g0, 11, HALF) \
        CASV_TOP_STAGE(g1, 12, HALF) CASV_TOP_STAGE(g0, 13, HALF) CASV_TOP_STAGE(g1, 14, HALF) CASV_TOP_STAGE(g0, 15, HALF) \
    }

    // Part A's rows.  A wave keeps its row for the whole sequence, so what the row can do ahead of its query it does (as in the
    // persistent decoder, persist.hip): the NEXT step's window is carried over from the weights just normalised (att_window_next:
    // the sum att_window reads back from memory, from registers), and the <= 11 rows of u and of the encoder outputs that window
    // attends are requested while the workgroup waits for the query.  At 16 unit groups a workgroup has two rows and four waves:
    // two waves share a row, each the tanh terms of half its columns -- wave 1 continues the per-lane sums of wave 0 in the
    // per-step kernel's order (column j = lane, then lane + 64), so the bits are attention_row's.
    constexpr int RPW = TBM / NT;                                  // rows of the row block per workgroup
    constexpr int WPR = RPW == 2 ? 2 : 1;                          // waves per row
    constexpr bool FAST_A = RPW * WPR == 4 && W / 4 == 64 * WPR;   // 16 unit groups: 2 rows x 2 waves, 128 float4 columns; 8: 4 rows x 1 wave, 64
    const int a_i = WPR == 2 ? (wave & 1) : wave, a_half = WPR == 2 ? (wave >> 1) : 0;
    const int a_r = m0 + ug * RPW + a_i;
    const bool a_live = FAST_A && a_r < B;
    const int a_rc = a_r < B ? a_r : B - 1;
    AttWin a_w{0, 0};
    const float* a_ub = nullptr; const float* a_eb = nullptr;
    if constexpr (FAST_A) {
        const AttnArgs& a0 = ra.att;
        a_w = att_window(a0, a_rc, 0, lane);
        const int ln = a0.line ? a0.line[a_rc] : a_rc / a0.rows_per_line;
        a_ub = a0.u + (long long)ln * a0.u_line;
        a_eb = a0.enc + (long long)ln * a0.enc_line;
    }

    for (int t = 0; t < U; ++t) {
        float* const rec = ra.RecIn + (long long)t * B * KR;            // the cell's input rows of this step: [ctx | h(t-1)]
        // x.Wx + b of this step's elements
        float zpre[4][4];
        {
            const float* zin0 = ra.Z + (long long)t * B * (4 * W);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q + 8 * wave + 4 * lh;
                const float* zr = zin0 + (long long)(m < B ? m : B - 1) * (4 * W) + n0 + l31;
                zpre[q][0] = zr[0]; zpre[q][1] = zr[32]; zpre[q][2] = zr[64]; zpre[q][3] = zr[96];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; accq[r] = 0.0f; }

        // =========== H: h(t-1) . [Wr_h | W_a] ===========
        {
            const float* arow = rec + (long long)mrow * KR + C + sk;
            CASV_TOP_HALF(0, cnt_h, NT * t, t == 0)
        }
        // the query columns: four k quarters -> LDS -> wave w sums rows 4 w .. 4 w + 3 of the accumulator layout; + bias
#pragma unroll
        for (int r = 0; r < 16; ++r) s_gate[wave][r][lane] = accq[r];
        __syncthreads();
        {
            float* wq = ra.WQ + (long long)t * B * W + ug * TQN + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = 4 * wave + q;                           // accumulator row index: matrix row (rr & 3) + 8 (rr >> 2) + 4 lh
                const float v = ((s_gate[0][rr][lane] + s_gate[1][rr][lane]) + (s_gate[2][rr][lane] + s_gate[3][rr][lane])) + qbias;
                const int m = m0 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (m < B) store_sc1(wq + (long long)m * W, v);
            }
        }
        publish(cnt_q);

        // =========== A: attention rows m0 + RPW ug ... ===========
        float4 uu[MAXWIN], xx[MAXWIN];
        if constexpr (FAST_A) {         // the window's rows: on their way while the workgroup waits for the row block's query
            const int Ta = ra.att.T;
#pragma unroll
            for (int i = 0; i < MAXWIN; ++i) {
                int sr = a_w.s_lo + i; sr = sr < Ta ? sr : Ta - 1; sr = sr < 0 ? 0 : sr;
                uu[i] = reinterpret_cast<const float4*>(a_ub + (long long)sr * ra.att.u_time)[lane + 64 * a_half];
                xx[i] = reinterpret_cast<const float4*>(a_eb + (long long)sr * ra.att.enc_time)[lane + 64 * a_half];
            }
        }
        if (!wait_deps(Dep{cnt_q, (unsigned)(NT * (t + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
        if constexpr (FAST_A) {
            AttnArgs a = ra.att;
            a.wq = ra.WQ + (long long)t * B * W;
            a.ctx = rec; a.ctx_ld = KR;
            a.step_imm = t; a.step_ptr = nullptr;
            a.win_out = ra.WIN + (long long)t * B;
            float e[MAXWIN];
#pragma unroll
            for (int i = 0; i < MAXWIN; ++i) e[i] = 0.0f;
            if constexpr (WPR == 1) {
                if (a_live) {
                    att_energy_sums<true, MAXWIN>(a, a_r, lane, a_w, [&](int sr) { return RegRows{uu, sr - a_w.s_lo}; }, [](int) { return true; }, e);
                    att_normalise(a, a_r, t, lane, a_w, e);
                    att_context<true, MAXWIN>(a, a_r, lane, a_w, [&](int sr) { return RegRows{xx, sr - a_w.s_lo}; }, e);
                }
            } else {
                float (*s_part)[MAXWIN][64] = reinterpret_cast<float (*)[MAXWIN][64]>(s_stage);      // [row][position][lane]: wave 0's sums
                float* s_sum = s_stage + RPW * MAXWIN * 64;                                           // [row][MAXWIN + 1]: the row's sums
                const int j = lane + 64 * a_half;
                const float4 q = load_sc1(a.wq + (long long)a_rc * W + 4 * j);
                const float4 v = reinterpret_cast<const float4*>(a.va)[j];
                float p[MAXWIN];
                if (a_half == 0) {
#pragma unroll
                    for (int i = 0; i < MAXWIN; ++i) {
                        p[i] = 0.0f;
                        p[i] += fast_tanh(q.x + uu[i].x) * v.x;
                        p[i] += fast_tanh(q.y + uu[i].y) * v.y;
                        p[i] += fast_tanh(q.z + uu[i].z) * v.z;
                        p[i] += fast_tanh(q.w + uu[i].w) * v.w;
                        s_part[a_i][i][lane] = p[i];
                    }
                } else {        // the tanh values first: they do not wait for wave 0
#pragma unroll
                    for (int i = 0; i < MAXWIN; ++i) {
                        uu[i].x = fast_tanh(q.x + uu[i].x); uu[i].y = fast_tanh(q.y + uu[i].y);
                        uu[i].z = fast_tanh(q.z + uu[i].z); uu[i].w = fast_tanh(q.w + uu[i].w);
                    }
                }
                __syncthreads();
                if (a_half == 1) {
#pragma unroll
                    for (int i = 0; i < MAXWIN; ++i) {
                        p[i] = s_part[a_i][i][lane];
                        p[i] += uu[i].x * v.x;
                        p[i] += uu[i].y * v.y;
                        p[i] += uu[i].z * v.z;
                        p[i] += uu[i].w * v.w;
                        p[i] = wave_sum(p[i]);
                    }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < MAXWIN; ++i) s_sum[a_i * (MAXWIN + 1) + i] = p[i];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < MAXWIN; ++i) e[i] = s_sum[a_i * (MAXWIN + 1) + i];
                if (a_live) {
                    att_normalise(a, a_r, t, lane, a_w, e, a_half == 1);
                    AttnArgs a2 = a;                        // this wave's half of the context columns
                    a2.C = C / 2; a2.ctx = a.ctx + a_half * (C / 2);
                    if (a.ctx_mask) a2.ctx_mask = a.ctx_mask + a_half * (C / 2);
                    att_context<true, MAXWIN>(a2, a_r, lane, a_w, [&](int sr) { return RegRows{xx, sr - a_w.s_lo}; }, e);
                }
            }
            publish(cnt_c);
            a_w = att_window_next(ra.att, a_w, e, lane);
        } else {
            for (int i = wave; i < RPW; i += 4) {
                const int r = m0 + ug * RPW + i;
                if (r < B) {
                    AttnArgs a = ra.att;
                    a.wq = ra.WQ + (long long)t * B * W;
                    a.ctx = rec; a.ctx_ld = KR;
                    a.step_imm = t; a.step_ptr = nullptr;
                    a.win_out = ra.WIN + (long long)t * B;
                    attention_row<true, 6>(a, r, t, lane);
                }
            }
            publish(cnt_c);
        }

        // =========== C: ctx(t) . Wr_c on top, then the cell ===========
        {
            const float* arow = rec + (long long)mrow * KR + sk;
            CASV_TOP_HALF(1, cnt_c, NT * (t + 1), false)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s_gate[wave][r][lane] = acc[r];
        __syncthreads();
        {
            float z[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) z[q][c] = s_gate[c][4 * wave + q][lane] + zpre[q][c];
            float* cout = ra.Cs + (long long)t * B * W;
            float* hout = ra.hs + (long long)t * B * W;
            float* gout = ra.Gt + (long long)t * B * (4 * W);
            float* hnext = t + 1 < U ? ra.RecIn + (long long)(t + 1) * B * KR + C : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q + 8 * wave + 4 * lh;
                if (m < B) {
                    const LstmCellOut cell = lstm_cell(z[q][0] + 0.f, z[q][1] + 0.f, z[q][2] + 0.f, z[q][3] + 0.f, cst[q]);
                    cst[q] = cell.c;
                    cout[(long long)m * W + u] = cell.c;
                    hout[(long long)m * W + u] = cell.h;
                    if (hnext) store_sc1(hnext + (long long)m * KR + u, cell.h);
                    float* gr = gout + (long long)m * (4 * W) + n0 + l31;
                    gr[0] = cell.i; gr[32] = cell.f; gr[64] = cell.g; gr[96] = cell.o;
                }
            }
        }
        publish(cnt_h);
    }
#undef CASV_TOP_HALF
#undef CASV_TOP_STAGE
#undef CASV_B_REGS
#undef CASV_LOAD_A
#undef CASV_LOAD_BC
#undef CASV_LOAD_BH
}

template <class K>
static int top_blocks_per_cu(K kernel) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    const void* f = reinterpret_cast<const void*>(kernel);
    auto it = cache.find({dev, f});
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, 0) != hipSuccess) n = 0;
    n = n > 1 ? 1 : (n < 0 ? 0 : n);          // planned for one workgroup per CU (its two idle waves in part A leave room)
    cache[{dev, f}] = n;
    return n;
}

size_t train_attention_cell_counter_bytes(int B) { return ((size_t)((B + TBM - 1) / TBM) * 3 * 32 + 32) * sizeof(unsigned); }

template <int NT> static int top_grid(const TopRecArgs& ra, int ncu) {
    const int grid = ((ra.B + TBM - 1) / TBM) * NT;
    return grid <= top_blocks_per_cu(train_attention_cell_kernel<NT>) * ncu ? grid : 0;
}
// Workgroups of the launch, or 0: no persistent form for this shape on this device (context as wide as the layer, i.e. depth >= 2;
// widths whose unit groups divide the 32 rows of a row block)
int train_attention_cell_grid(const TopRecArgs& ra, int ncu) {
    if (ra.W != ra.C || ra.B < 1 || ra.U < 1) return 0;
    switch (ra.W / 32 * (ra.W % 32 == 0)) {
        case 4: return top_grid<4>(ra, ncu);
        case 8: return top_grid<8>(ra, ncu);
        case 16: return top_grid<16>(ra, ncu);
        default: return 0;
    }
}
void launch_train_attention_cell(const TopRecArgs& ra, int grid, hipStream_t stream) {
    switch (ra.W / 32) {
        case 4: hipLaunchKernelGGL((train_attention_cell_kernel<4>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 8: hipLaunchKernelGGL((train_attention_cell_kernel<8>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 16: hipLaunchKernelGGL((train_attention_cell_kernel<16>), dim3(grid), dim3(256), 0, stream, ra); break;
        default: break;
    }
}

}  // namespace casv
