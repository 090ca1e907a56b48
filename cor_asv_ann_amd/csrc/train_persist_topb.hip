// The attention cell's backward recurrence of the train step as ONE launch -- the mirror of train_persist_top.hip.
//
// train.hip walks it with three launches per time step: cell backward + data GEMM (gemm_bwd.hip, 33 us), attention backward (27),
// query-path GEMM (11), ~81 us with their boundaries, 102 times.  Here (row block of 32) x (slot 0..W/32-1) workgroups, one per
// CU, stay resident and all play the same five parts in every step (handoff.h; dL/dc of a workgroup's cells in registers):
//   P  the cells of 32 rows x 32 units: dL/dh(t) = dG(t) + dRec(t+1)[h part] + dhatt(t+1)  ->  dZ(t);
//   G  two 32 x 128 tiles of dRec(t) = dZ(t) . Wr, one of the ctx half (needed by the attention backward of this step) and one
//      of the h half (needed by the cells of the next step), each over a quarter of the gate axis on the SAME registers of dZ rows,
//      added with float atomics as the per-step launches do;
//   A  the attention backward of two of the row block's samples, half a workgroup each (attn_bwd.h: the per-step kernel's
//      code): d_enc / du atomics, dva / dbv partial sums, dwq rows;
//   Q  32 columns of dhatt(t) = dwq(t) . W_a (every wave a k quarter of each stage, summed through LDS).
// Five counters per row block (dZ quarters, ctx tiles, h column tiles, dwq rows, dhatt columns).
#include "common.h"
#include "handoff.h"
#include "row_kernels.h"
#include "train_kernels.h"
#include "attn_bwd.h"
#include <math.h>
#include <map>
#include <mutex>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int VBM = 32, VBN = 128;
constexpr int VK2 = 32, VLD = VK2 + 4;
constexpr int VSTAGE = (VBM + VBN) * VLD;
struct VIn { f32x4 a, gi, gf, gg, go, cell, cp; };
}

template <int NT>        // W / 32 = C / 32 (16 at width 512): slots per row block; K stages per tile
__global__ __launch_bounds__(256, 1) void train_attention_cell_bwd_kernel(const TopBwdArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_stage[2 * VSTAGE];       // 46 KB; parts A and Q reuse it
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int W = NT * 32, C = W, KR = C + W;
    constexpr long long K = 4LL * W;
    const int B = ra.B, U = ra.U;
    const int nrb = (B + VBM - 1) / VBM;
    const int slot = blockIdx.x % NT, rb = blockIdx.x / NT;
    const int m0 = rb * VBM;
    // counters of this row block: 4 x dZ quarter, ctx tiles, 4 x h column tile, dwq rows, dhatt columns
    unsigned* const cbase = ra.counters + (long long)rb * 12 * 32;
    unsigned* const abort_w = ra.counters + (long long)nrb * 12 * 32;
    const int ug = slot, ct = slot >> 2, ks = slot & 3;
    unsigned* const z_mine = cbase + (ug / (NT / 4)) * 32;
    unsigned* const z_need = cbase + ks * 32;
    unsigned* const c_cnt = cbase + 4 * 32;
    unsigned* const h_mine = cbase + (5 + ct) * 32;
    unsigned* const h_need = cbase + (5 + (ug >> 2)) * 32;
    unsigned* const q_cnt = cbase + 9 * 32;
    unsigned* const d_cnt = cbase + 10 * 32;

    const int srow = tid >> 3, su = 4 * (tid & 7);
    const bool row_ok = m0 + srow < B;
    const int mrow = row_ok ? m0 + srow : B - 1;
    const int u0 = ug * 32 + su;

    // ---- operands of the tiles ----
    const float* bpc[4]; const float* bph[4];        // Wr^T rows of the ctx / h column tile, this K share
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bpc[i] = ra.WrT + (long long)(ct * VBN + srow + 32 * i) * K + (long long)ks * W + su;
        bph[i] = ra.WrT + (long long)(C + ct * VBN + srow + 32 * i) * K + (long long)ks * W + su;
    }
    const float* qp = ra.WaN + (long long)(ug * 32 + srow) * W + su;          // W_a rows of my 32 dhatt columns
    struct BStage { f32x4 b[4]; };
#define CASV_LOAD_B(G, BP, KT)                                                                                          \
    {                                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(BP[0]), "n"((KT) * VK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[1]) : "v"(BP[1]), "n"((KT) * VK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[2]) : "v"(BP[2]), "n"((KT) * VK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[3]) : "v"(BP[3]), "n"((KT) * VK2 * 4));      \
    }
#define CASV_LOAD_Q(G, KT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(qp), "n"((KT) * VK2 * 4));
#define CASV_LOAD_A(J)                                                                                                  \
    if constexpr ((J) < NT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(areg[(J) < NT ? (J) : 0]) : "v"(arow), "n"((J) * VK2 * 4));
#define CASV_LD16(DST, PTR, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF))
#define CASV_B_REGS(G) "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3])
    auto store_b = [&](const BStage& gs, int buf) {
        float* sb = s_stage + buf * VSTAGE + (VBM + srow) * VLD + su;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + 32 * i * VLD) = gs.b[i];
    };
    auto store_q = [&](const BStage& gs, int buf) {        // 32 rows of W_a in place of the first 32 B rows
        *reinterpret_cast<f32x4*>(s_stage + buf * VSTAGE + (VBM + srow) * VLD + su) = gs.b[0];
    };
    auto store_a = [&](const f32x4& a, int buf) {
        *reinterpret_cast<f32x4*>(s_stage + buf * VSTAGE + srow * VLD + su) = a;
    };
    const int a_off = l31 * VLD + 4 * lh, b_off = (VBM + wave * 32 + l31) * VLD + 4 * lh, q_off = (VBM + l31) * VLD + 4 * lh;
    f32x16 acc;
    auto compute = [&](int buf) {
        const float* base = s_stage + buf * VSTAGE;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(base + a_off + 8 * q);
            fb[q] = *reinterpret_cast<const f32x4*>(base + b_off + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i], fb[q][i], acc, 0, 0, 0);
    };
    auto compute_q = [&](int buf) {                         // this wave's k quarter of the stage against the 32 W_a rows
        const float* base = s_stage + buf * VSTAGE;
        const f32x4 fa = *reinterpret_cast<const f32x4*>(base + a_off + 8 * wave);
        const f32x4 fq = *reinterpret_cast<const f32x4*>(base + q_off + 8 * wave);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fq[i], acc, 0, 0, 0);
    };
    // one tile's K loop over the stages held in areg: B stages double-buffered in LDS, counted waits (train_persist_bwd.hip)
#define CASV_TILE_STAGE(G, BP, J)                                                                                       \
            if constexpr ((J) < NT) {                                                                                   \
                if constexpr ((J) + 1 < NT) {                                                                           \
                    if constexpr ((J) >= 2 && (J) + 2 < NT) asm volatile("s_waitcnt vmcnt(4)" : CASV_B_REGS(G));        \
                    else if constexpr ((J) >= 2) asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(G));                   \
                    store_a(areg[(J) + 1 < NT ? (J) + 1 : 0], ((J) + 1) & 1);                                           \
                    store_b(G, ((J) + 1) & 1);                                                                          \
                }                                                                                                       \
                if constexpr ((J) + 3 < NT) CASV_LOAD_B(G, BP, (J) + 3)                                                 \
                compute((J) & 1);                                                                                       \
                __syncthreads();                                                                                        \
            }
#define CASV_TILE_LOOP(BP)                                                                                              \
            CASV_TILE_STAGE(g1, BP, 0) CASV_TILE_STAGE(g0, BP, 1) CASV_TILE_STAGE(g1, BP, 2) CASV_TILE_STAGE(g0, BP, 3)   \
            CASV_TILE_STAGE(g1, BP, 4) CASV_TILE_STAGE(g0, BP, 5) CASV_TILE_STAGE(g1, BP, 6) CASV_TILE_STAGE(g0, BP, 7)   \
            CASV_TILE_STAGE(g1, BP, 8) CASV_TILE_STAGE(g0, BP, 9) CASV_TILE_STAGE(g1, BP, 10) CASV_TILE_STAGE(g0, BP, 11) \
            CASV_TILE_STAGE(g1, BP, 12) CASV_TILE_STAGE(g0, BP, 13) CASV_TILE_STAGE(g1, BP, 14) CASV_TILE_STAGE(g0, BP, 15)
#define CASV_Q_STAGE(G, J)                                                                                              \
            if constexpr ((J) < NT) {                                                                                   \
                if constexpr ((J) + 1 < NT) {                                                                           \
                    if constexpr ((J) >= 2 && (J) + 2 < NT) asm volatile("s_waitcnt vmcnt(1)" : "+v"(G.b[0]));          \
                    else if constexpr ((J) >= 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(G.b[0]));                     \
                    store_a(areg[(J) + 1 < NT ? (J) + 1 : 0], ((J) + 1) & 1);                                           \
                    store_q(G, ((J) + 1) & 1);                                                                          \
                }                                                                                                       \
                if constexpr ((J) + 3 < NT) CASV_LOAD_Q(G, (J) + 3)                                                     \
                compute_q((J) & 1);                                                                                     \
                __syncthreads();                                                                                        \
            }

    // ---- P: state and the inputs of the first step ----
    f32x4 dc;
#pragma unroll
    for (int e = 0; e < 4; ++e) dc[e] = 0.0f;
    auto load_pin = [&](VIn& in, int t) {
        const float* xg = ra.Gt + ((long long)t * B + mrow) * K + ug * 128 + su;
        const float* xc = ra.Cs + ((long long)t * B + mrow) * W + u0;
        const float* xa = ra.dG + ((long long)t * B + mrow) * W + u0;
        const float* xp = t > 0 ? ra.Cs + ((long long)(t - 1) * B + mrow) * W + u0 : ra.c0 + (long long)mrow * W + u0;
        CASV_LD16(in.a, xa, 0);
        CASV_LD16(in.gi, xg, 0); CASV_LD16(in.gf, xg, 128); CASV_LD16(in.gg, xg, 256); CASV_LD16(in.go, xg, 384);
        CASV_LD16(in.cell, xc, 0); CASV_LD16(in.cp, xp, 0);
    };
#define CASV_PIN_REGS(IN) "+v"(IN.a), "+v"(IN.gi), "+v"(IN.gf), "+v"(IN.gg), "+v"(IN.go), "+v"(IN.cell), "+v"(IN.cp)
    VIn in;
    load_pin(in, U - 1);
    asm volatile("s_waitcnt vmcnt(0)" : CASV_PIN_REGS(in));

    for (int i = 0; i < U; ++i) {
        const int t = U - 1 - i;
        // =========== P: dZ(t) of my cells ===========
        f32x4 bv, cv;
        if (i > 0) {
            if (!wait_deps(Dep{h_need, 4u * (unsigned)i}, Dep{d_cnt, (unsigned)(NT * i)}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            const float* xb = ra.dRec + ((long long)(t + 1) * B + mrow) * KR + C + u0;
            const float* xc = ra.dhatt + ((long long)(t + 1) * B + mrow) * W + u0;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(bv) : "v"(xb));
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(cv) : "v"(xc));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv), "+v"(cv));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = 0.0f; cv[e] = 0.0f; }
        }
        {
            f32x4 zi, zf, zg, zo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {         // lstm_bwd_kernel's arithmetic (train_kernels.hip): dh = a + b + c
                float dh = 0.f;
                dh += in.a[e];
                if (i > 0) { dh += bv[e]; dh += cv[e]; }
                const float ig = in.gi[e], fg = in.gf[e], gg = in.gg[e], og = in.go[e];
                const float cp = in.cp[e];
                const float tc = tanhf(in.cell[e]);
                const float dov = dh * tc;
                const float dct = dh * og * (1.0f - tc * tc) + dc[e];
                zi[e] = dct * gg * ig * (1.0f - ig);
                zf[e] = dct * cp * fg * (1.0f - fg);
                zg[e] = dct * ig * (1.0f - gg * gg);
                zo[e] = dov * og * (1.0f - og);
                dc[e] = dct * fg;
            }
            if (row_ok) {
                float* z = ra.dZ + ((long long)t * B + mrow) * K + ug * 128 + su;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(z), "v"(zi) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:128 sc1" :: "v"(z), "v"(zf) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:256 sc1" :: "v"(z), "v"(zg) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:384 sc1" :: "v"(z), "v"(zo) : "memory");
            }
        }
        publish(z_mine);

        // =========== G: the ctx tile, then the h tile, of dRec(t) on the same rows of dZ(t) ===========
        {
            BStage g0, g1;
            CASV_LOAD_B(g0, bpc, 0)
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));
            store_b(g0, 0);
            if constexpr (NT > 1) CASV_LOAD_B(g1, bpc, 1)
            if constexpr (NT > 2) CASV_LOAD_B(g0, bpc, 2)
            if (!wait_deps(Dep{z_need, (unsigned)((NT / 4) * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            const float* arow = ra.dZ + ((long long)t * B + mrow) * K + (long long)ks * W + su;
            f32x4 areg[NT];
            CASV_LOAD_A(0) CASV_LOAD_A(1) CASV_LOAD_A(2) CASV_LOAD_A(3) CASV_LOAD_A(4) CASV_LOAD_A(5) CASV_LOAD_A(6) CASV_LOAD_A(7)
            CASV_LOAD_A(8) CASV_LOAD_A(9) CASV_LOAD_A(10) CASV_LOAD_A(11) CASV_LOAD_A(12) CASV_LOAD_A(13) CASV_LOAD_A(14) CASV_LOAD_A(15)
            load_pin(in, t > 0 ? t - 1 : 0);      // what the next step's cells need that does not depend on this step
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[j]));
            asm volatile("s_waitcnt vmcnt(0)" : CASV_PIN_REGS(in));
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1));
            store_a(areg[0], 0);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            CASV_TILE_LOOP(bpc)
            {
                float* out = ra.dRec + (long long)t * B * KR + ct * VBN + wave * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < B) atomicAdd(out + (long long)m * KR, acc[r]);
                }
            }
            publish(c_cnt);
            // the h tile: same A registers, the other weight panel
            CASV_LOAD_B(g0, bph, 0)
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));
            store_b(g0, 0);
            if constexpr (NT > 1) CASV_LOAD_B(g1, bph, 1)
            if constexpr (NT > 2) CASV_LOAD_B(g0, bph, 2)
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g0));
            asm volatile("s_waitcnt vmcnt(0)" : CASV_B_REGS(g1));
            store_a(areg[0], 0);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            CASV_TILE_LOOP(bph)
            {
                float* out = ra.dRec + (long long)t * B * KR + C + ct * VBN + wave * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < B) atomicAdd(out + (long long)m * KR, acc[r]);
                }
            }
            publish(h_mine);
        }

        // =========== A: attention backward of samples m0 + 2 slot (+ 1), half a workgroup each ===========
        // (split_a: the workgroups of train_attention_cell_bwd_rows_kernel do it meanwhile, on the CU's other slot -- part A needs
        // the ctx tiles only, the h tile above and part A then overlap instead of following each other)
        if (!ra.split_a) {
        if (!wait_deps(Dep{c_cnt, (unsigned)(NT * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
        {
            constexpr int SPW = VBM / NT;                    // samples of the row block per workgroup (2 at 16 slots)
            constexpr int GROUPS = SPW >= 2 ? 2 : 1;         // groups working side by side
            const int nthr = 256 / GROUPS, grp = tid / nthr, gtid = tid % nthr;
            float* s_dx = s_stage + grp * 1024;
            float* s_da = s_stage + 2048 + grp * 64; float* s_ds = s_da + 16; float* s_av = s_da + 32;
            AttnBwdArgs p = ra.ab;
            p.dxh = ra.dRec + (long long)t * B * KR; p.ld_dxh = KR; p.ctx_off = 0;
            p.a = ra.Ast + (long long)(t + 1) * B * ra.ab.T; p.win = ra.WIN + (long long)t * B;
            p.wq = ra.WQ + (long long)t * B * W; p.dwq = ra.DWQ + (long long)t * B * W;
            p.ds_out = ra.DS ? ra.DS + (long long)t * B * 16 : nullptr;
            for (int s0 = 0; s0 < SPW; s0 += GROUPS) {
                const int b = m0 + slot * SPW + s0 + grp;
                if (ra.defer == 3) attention_bwd_sample<true, 3>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
                else if (ra.defer == 1) attention_bwd_sample<true, 1>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
                else attention_bwd_sample<true, 0>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
                __syncthreads();
            }
        }
        publish(q_cnt);
        }

        // =========== Q: my 32 columns of dhatt(t) = dwq(t) . W_a ===========
        {
            BStage g0, g1;
            CASV_LOAD_Q(g0, 0)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]));
            store_q(g0, 0);
            if constexpr (NT > 1) CASV_LOAD_Q(g1, 1)
            if constexpr (NT > 2) CASV_LOAD_Q(g0, 2)
            if (!wait_deps(Dep{q_cnt, (unsigned)(NT * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            const float* arow = ra.DWQ + ((long long)t * B + mrow) * W + su;
            f32x4 areg[NT];
            CASV_LOAD_A(0) CASV_LOAD_A(1) CASV_LOAD_A(2) CASV_LOAD_A(3) CASV_LOAD_A(4) CASV_LOAD_A(5) CASV_LOAD_A(6) CASV_LOAD_A(7)
            CASV_LOAD_A(8) CASV_LOAD_A(9) CASV_LOAD_A(10) CASV_LOAD_A(11) CASV_LOAD_A(12) CASV_LOAD_A(13) CASV_LOAD_A(14) CASV_LOAD_A(15)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[j]));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g1.b[0]));
            store_a(areg[0], 0);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            CASV_Q_STAGE(g1, 0) CASV_Q_STAGE(g0, 1) CASV_Q_STAGE(g1, 2) CASV_Q_STAGE(g0, 3)
            CASV_Q_STAGE(g1, 4) CASV_Q_STAGE(g0, 5) CASV_Q_STAGE(g1, 6) CASV_Q_STAGE(g0, 7)
            CASV_Q_STAGE(g1, 8) CASV_Q_STAGE(g0, 9) CASV_Q_STAGE(g1, 10) CASV_Q_STAGE(g0, 11)
            CASV_Q_STAGE(g1, 12) CASV_Q_STAGE(g0, 13) CASV_Q_STAGE(g1, 14) CASV_Q_STAGE(g0, 15)
            // four k quarters -> LDS -> wave w sums accumulator rows 4 w .. 4 w + 3
            float (*s_red)[16][64] = reinterpret_cast<float (*)[16][64]>(s_stage);
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
            __syncthreads();
            float* dh = ra.dhatt + (long long)t * B * W + ug * 32 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = 4 * wave + q;
                const float v = (s_red[0][rr][lane] + s_red[1][rr][lane]) + (s_red[2][rr][lane] + s_red[3][rr][lane]);
                const int m = m0 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (m < B) store_sc1(dh + (long long)m * W, v);
            }
        }
        publish(d_cnt);
    }
#undef CASV_PIN_REGS
#undef CASV_Q_STAGE
#undef CASV_TILE_LOOP
#undef CASV_TILE_STAGE
#undef CASV_B_REGS
#undef CASV_LD16
#undef CASV_LOAD_A
#undef CASV_LOAD_Q
#undef CASV_LOAD_B
    // dL/dc of the cell's initial state
    if (row_ok) *reinterpret_cast<f32x4*>(ra.dc_out + (long long)mrow * W + u0) = dc;
}

// Part A of the kernel above as a launch of its own (TopBwdArgs.split_a): same grid, same roles by workgroup index, same counters.
// Its workgroups wait for their row block's ctx tiles, work the attention backward of their samples and publish the dwq rows; the
// big kernel's workgroups go from their h tile straight to the wait for those rows.  Few registers, 9 KB of LDS: a workgroup of
// this kernel fits beside one of the big kernel on every CU.
template <int NT>
__global__ __launch_bounds__(256, 3) void train_attention_cell_bwd_rows_kernel(const TopBwdArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_scr[2 * 1024 + 2 * 64];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    constexpr int W = NT * 32, C = W, KR = C + W;
    const int B = ra.B, U = ra.U;
    const int nrb = (B + VBM - 1) / VBM;
    const int slot = blockIdx.x % NT, rb = blockIdx.x / NT;
    const int m0 = rb * VBM;
    unsigned* const cbase = ra.counters + (long long)rb * 12 * 32;
    unsigned* const abort_w = ra.counters + (long long)nrb * 12 * 32;
    unsigned* const c_cnt = cbase + 4 * 32;
    unsigned* const q_cnt = cbase + 9 * 32;
    constexpr int SPW = VBM / NT;                    // samples of the row block per workgroup (2 at 16 slots)
    constexpr int GROUPS = SPW >= 2 ? 2 : 1;         // groups working side by side
    const int nthr = 256 / GROUPS, grp = tid / nthr, gtid = tid % nthr;
    float* s_dx = s_scr + grp * 1024;
    float* s_da = s_scr + 2048 + grp * 64; float* s_ds = s_da + 16; float* s_av = s_da + 32;
    if constexpr (SPW == GROUPS) {
        // One sample per half workgroup and step (16 slots per row block).  Everything the forward pass kept -- the window, the
        // alignment values, the window's rows of the encoder outputs, the query, the rows of u -- is requested, and every tanh
        // taken, BEFORE the wait for the step's dL/dctx (this kernel's workgroups have nothing else to do meanwhile); behind the
        // wait: the dL/dctx row, eleven dot products, the score gradients, dwq from the tanh values in registers -- and the signal.
        // What nobody waits for inside the recurrence (du / d_enc atomics, the dva / dbv sums, the kept score gradients) follows
        // the signal.  The arithmetic is attention_bwd_sample's, sum for sum.
        constexpr int NTHR = 256 / GROUPS, NWV = NTHR / 64, NP = (MAXWIN + NWV - 1) / NWV, NE = C / 64, NJ = W / NTHR, NC = C / NTHR;
        const int lane = gtid & 63, gw = gtid >> 6;
        const int b = m0 + slot * SPW + grp;
        const bool active = b < B;
        const int bc = active ? b : B - 1;
        const AttnBwdArgs& ab = ra.ab;
        const int T = ab.T, defer = ra.defer;
        for (int i = 0; i < U; ++i) {
            const int t = U - 1 - i;
            // ---- ahead of the wait
            const int wv = ra.WIN[(long long)t * B + bc];
            const int s_lo = wv & 0xffff, cnt = wv >> 16;
            const float av = (gtid < 16 && gtid < cnt) ? ra.Ast[((long long)(t + 1) * B + bc) * T + s_lo + gtid] : 0.0f;
            float encv[NP][NE];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int pi = gw + k * NWV;
                const float* es = ab.enc + (long long)bc * ab.enc_line + (long long)(s_lo + (pi < cnt ? pi : 0)) * ab.enc_time;
#pragma unroll
                for (int e = 0; e < NE; ++e) encv[k][e] = es[lane + 64 * e];
            }
            float mcv[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) mcv[k] = ab.mcell ? ab.mcell[(long long)bc * ab.ld_mcell + ab.mc_off + gtid + NTHR * k] : 1.0f;
            float th[NJ][MAXWIN], vv[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = gtid + NTHR * jj;
                const float q = ra.WQ[((long long)t * B + bc) * W + j];
                vv[jj] = ab.va[j];
                const long long off0 = (long long)bc * ab.u_line + (long long)s_lo * ab.u_time + j;
                float uu[MAXWIN];
#pragma unroll
                for (int k = 0; k < MAXWIN; ++k) uu[k] = ab.u[off0 + (long long)(k < cnt ? k : 0) * ab.u_time];
#pragma unroll
                for (int k = 0; k < MAXWIN; ++k) th[jj][k] = fast_tanh(q + uu[k]);
            }
            if (!wait_deps(Dep{c_cnt, (unsigned)(NT * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            // ---- behind it
            const float* dx = ra.dRec + ((long long)t * B + bc) * KR;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c = gtid + NTHR * k;
                s_dx[c] = __hip_atomic_load(dx + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * mcv[k];
            }
            if (gtid < 16) s_av[gtid] = av;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NP; ++k) {          // da_s = dctx . enc_s
                const int pi = gw + k * NWV;
                if (pi < cnt) {
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < NE; ++e) part += s_dx[lane + 64 * e] * encv[k][e];
                    part = wave_sum(part);
                    if (lane == 0) s_da[pi] = part;
                }
            }
            __syncthreads();
            if (gtid < 16) {
                float dot = 0.f;
                for (int k = 0; k < cnt; ++k) dot += s_av[k] * s_da[k];
                s_ds[gtid] = gtid < cnt ? s_av[gtid] * (s_da[gtid] - dot) : 0.0f;          // dL/dscore
            }
            __syncthreads();
            float dvas[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                float dwq = 0.f, dva = 0.f;
#pragma unroll
                for (int k = 0; k < MAXWIN; ++k) {
                    const float ds = s_ds[k];
                    dva += ds * th[jj][k];
                    dwq += ds * vv[jj] * (1.0f - th[jj][k] * th[jj][k]);
                }
                dvas[jj] = dva;
                if (active) store_sc1(ra.DWQ + ((long long)t * B + b) * W + gtid + NTHR * jj, dwq);
            }
            publish(q_cnt);
            // ---- behind the signal
            if (active) {
                if (defer && gtid < 16) ra.DS[((long long)t * B + b) * 16 + gtid] = s_ds[gtid];
                if (!(defer & 1)) {                 // d enc_out[s] += a_s * dctx
                    float* de = ab.d_enc + (long long)b * ab.enc_line + (long long)s_lo * ab.enc_time;
                    for (int k = 0; k < cnt; ++k) {
                        const float a_k = s_av[k];
                        for (int c = gtid; c < C; c += NTHR) atomicAdd(de + (long long)k * ab.enc_time + c, a_k * s_dx[c]);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int j = gtid + NTHR * jj;
                    if (!(defer & 2)) {
                        const long long off0 = (long long)b * ab.u_line + (long long)s_lo * ab.u_time + j;
#pragma unroll
                        for (int k = 0; k < MAXWIN; ++k)
                            if (k < cnt) atomicAdd(ab.du + off0 + (long long)k * ab.u_time, s_ds[k] * vv[jj] * (1.0f - th[jj][k] * th[jj][k]));
                    }
                    ab.dva_part[(long long)b * W + j] += dvas[jj];
                }
                if (gtid == 0) {
                    float dbv = 0.f;
                    for (int k = 0; k < cnt; ++k) dbv += s_ds[k];
                    ab.dbv_part[b] += dbv;
                }
            }
        }
        return;
    }
    for (int i = 0; i < U; ++i) {
        const int t = U - 1 - i;
        if (!wait_deps(Dep{c_cnt, (unsigned)(NT * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
        AttnBwdArgs p = ra.ab;
        p.dxh = ra.dRec + (long long)t * B * KR; p.ld_dxh = KR; p.ctx_off = 0;
        p.a = ra.Ast + (long long)(t + 1) * B * ra.ab.T; p.win = ra.WIN + (long long)t * B;
        p.wq = ra.WQ + (long long)t * B * W; p.dwq = ra.DWQ + (long long)t * B * W;
        p.ds_out = ra.DS ? ra.DS + (long long)t * B * 16 : nullptr;
        for (int s0 = 0; s0 < SPW; s0 += GROUPS) {
            const int b = m0 + slot * SPW + s0 + grp;
            if (ra.defer == 3) attention_bwd_sample<true, 3>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
            else if (ra.defer == 1) attention_bwd_sample<true, 1>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
            else attention_bwd_sample<true, 0>(p, b < B ? b : B - 1, b < B && s0 + grp < SPW, gtid, nthr, s_dx, s_da, s_ds, s_av);
            __syncthreads();
        }
        publish(q_cnt);
    }
}

template <class K>
static int topb_blocks_per_cu(K kernel) {
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
    n = n > 1 ? 1 : (n < 0 ? 0 : n);
    cache[{dev, f}] = n;
    return n;
}

size_t train_attention_cell_bwd_counter_bytes(int B) { return ((size_t)((B + VBM - 1) / VBM) * 12 * 32 + 32) * sizeof(unsigned); }

template <int NT> static int topb_grid(const TopBwdArgs& ra, int ncu) {
    const int grid = ((ra.B + VBM - 1) / VBM) * NT;
    return grid <= topb_blocks_per_cu(train_attention_cell_bwd_kernel<NT>) * ncu ? grid : 0;
}
// Workgroups of the launch, or 0: no persistent form for this shape on this device (context as wide as the layer; whole column
// tiles of 128 units; unit groups that divide the 32 rows of a row block)
int train_attention_cell_bwd_grid(const TopBwdArgs& ra, int ncu) {
    if (ra.W != ra.C || ra.W % 128 || ra.B < 1 || ra.U < 1 || ra.ab.C > 1024) return 0;
    switch (ra.W / 32) {
        case 4: return topb_grid<4>(ra, ncu);
        case 8: return topb_grid<8>(ra, ncu);
        case 16: return topb_grid<16>(ra, ncu);
        default: return 0;
    }
}
void launch_train_attention_cell_bwd(const TopBwdArgs& ra, int grid, hipStream_t stream) {
    switch (ra.W / 32) {
        case 4: hipLaunchKernelGGL((train_attention_cell_bwd_kernel<4>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 8: hipLaunchKernelGGL((train_attention_cell_bwd_kernel<8>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 16: hipLaunchKernelGGL((train_attention_cell_bwd_kernel<16>), dim3(grid), dim3(256), 0, stream, ra); break;
        default: break;
    }
}

// Both launches resident at once, a workgroup of each on every CU, in whatever order the hardware takes them from their streams:
// the big kernel's registers admit one of its workgroups per CU; the small kernel asks for ROWS_LDS_PAD bytes of LDS it never touches,
// so that two of ITS workgroups do not fit a CU either (2 x 93 KB > 160 KB) while one fits beside the big kernel's (46 + 93 KB) --
// no CU can fill up with workgroups of one kind and lock the other kind out.  The runtime has no query for two kernels together:
// what is checked is each one's appetite (per SIMD one wave of each within the 512 registers, the two LDS images within the CU's
// LDS); if a device or a co-tenant proves it wrong the bounded waits give up and train.hip goes back to the single launch.
constexpr int ROWS_LDS_PAD = 84 * 1024;
template <int NT> static bool topb_rows_fit() {
    static const bool ok = [] {
        hipFuncAttributes a{}, b{};
        const void* fa = reinterpret_cast<const void*>(train_attention_cell_bwd_kernel<NT>);
        const void* fb = reinterpret_cast<const void*>(train_attention_cell_bwd_rows_kernel<NT>);
        if (hipFuncGetAttributes(&a, fa) != hipSuccess || hipFuncGetAttributes(&b, fb) != hipSuccess) return false;
        if (hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS_PAD) != hipSuccess) return false;
        // (numRegs counts the architectural registers; the big kernel also uses accumulation registers, 34-56 of them: 320 in all)
        const int big = 320, rows = ((b.numRegs + 7) / 8) * 8;
        const size_t lds_rows = b.sharedSizeBytes + ROWS_LDS_PAD;
        return b.numRegs > 0 && big + rows <= 512 && a.sharedSizeBytes + lds_rows <= 160 * 1024 && 2 * lds_rows > 160 * 1024;
    }();
    return ok;
}
bool train_attention_cell_bwd_rows_fit(const TopBwdArgs& ra) {
    switch (ra.W / 32) {
        case 4: return topb_rows_fit<4>();
        case 8: return topb_rows_fit<8>();
        case 16: return topb_rows_fit<16>();
        default: return false;
    }
}
void launch_train_attention_cell_bwd_rows(const TopBwdArgs& ra, int grid, hipStream_t stream) {
    switch (ra.W / 32) {
        case 4: hipLaunchKernelGGL((train_attention_cell_bwd_rows_kernel<4>), dim3(grid), dim3(256), ROWS_LDS_PAD, stream, ra); break;
        case 8: hipLaunchKernelGGL((train_attention_cell_bwd_rows_kernel<8>), dim3(grid), dim3(256), ROWS_LDS_PAD, stream, ra); break;
        case 16: hipLaunchKernelGGL((train_attention_cell_bwd_rows_kernel<16>), dim3(grid), dim3(256), ROWS_LDS_PAD, stream, ra); break;
        default: break;
    }
}

}  // namespace casv
