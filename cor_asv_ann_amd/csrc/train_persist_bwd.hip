// The backward recurrences of the train step (keras' LSTM gradient through time, oracle/train.py lstm_backward) as ONE launch
// per pair of layers -- the mirror of train_persist.hip.
//
// A backward time step is: the cells' pointwise backward (gate derivatives dZ(t) from dL/dh(t) = dOut(t) + dRec(t+1) and the
// running dL/dc), then the data GEMM dRec(t) = dZ(t) . Wr that carries the gradient one step further.  As launches that is
// ~33 us per step (gemm_bwd.hip: one fused launch), 407 times per train step for the plain layers.  Here 2 x 16 x 16 workgroups
// stay resident and play BOTH parts in every step:
//   P  workgroup (layer, row block, unit group) owns the cells of 32 rows x 32 units: their dL/dc lives in its registers for
//      the whole sequence; it waits for the four K shares of dRec(t+1) of its column tile, computes dZ(t) and hands it on;
//   G  the same workgroup as (layer, row block, column tile, K share) contracts dZ(t) of its 32 rows over its quarter of the gate
//      axis (= W columns: as many stages as the forward step) with its 128 x W panel of Wr^T, and adds the partial tile to
//      dRec(t) with float atomics (four shares per element, as the per-step launches do).
// Hand-offs as in handoff.h (write-through stores / memory-side atomics, drain, one counter per (row block, quarter) and per
// (row block, column tile)); everything a step needs that does not depend on the step before -- gate activations, cell states,
// dOut -- is requested together with the dZ rows of the step before it, so that P itself waits for one 16-byte load.
// Weight panels: slot s = (column tile, K share) of every row block and both layers sits on XCD s % 8, whose L2 holds them.
#include "common.h"
#include "handoff.h"
#include "train_kernels.h"
#include <math.h>
#include <map>
#include <mutex>

namespace casv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int QBM = 32, QBN = 128;
constexpr int QK2 = 32, QLD = QK2 + 4;
constexpr int QSTAGE = (QBM + QBN) * QLD;
struct PIn { f32x4 a, gi, gf, gg, go, cell, cp; };
}

template <int NT>        // W / 32: unit groups = slots per (layer, row block); K stages per step
__global__ __launch_bounds__(256, 2) void train_recurrence_bwd_kernel(const RecBwdArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_stage[2 * QSTAGE];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int W = NT * 32;
    constexpr long long K = 4LL * W;
    const int B = ra.B;
    const int nrb = (B + QBM - 1) / QBM;
    const int slot = blockIdx.x % NT, rest = blockIdx.x / NT;
    const int rb = rest % nrb, jb = rest / nrb;
    const RecBwdJob& job = ra.job[jb];
    const int len = job.len;
    const int m0 = rb * QBM;
    // counters of this (layer, row block): four "dZ quarter ready" lines, then up to four "dRec column tile ready" lines
    unsigned* const cbase = ra.counters + (long long)(jb * nrb + rb) * 8 * 32;
    unsigned* const abort_w = ra.counters + (long long)2 * nrb * 8 * 32;
    // P: unit group ug of the row block;  G: column tile ct, K share ks
    const int ug = slot, ct = slot >> 2, ks = slot & 3;
    unsigned* const z_mine = cbase + (ug / (NT / 4)) * 32;          // the quarter my dZ columns belong to
    unsigned* const z_need = cbase + ks * 32;
    unsigned* const r_mine = cbase + (4 + ct) * 32;
    unsigned* const r_need = cbase + (4 + (ug >> 2)) * 32;         // the column tile (128 units) my cells' dL/dh sits in

    const int srow = tid >> 3, su = 4 * (tid & 7);
    const bool row_ok = m0 + srow < B;
    const int mrow = row_ok ? m0 + srow : B - 1;
    const int u0 = ug * 32 + su;                                    // this thread's four cells: row mrow, units u0 .. u0 + 3

    // ---- G: operands ----
    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bp[i] = job.WrT + (long long)(ct * QBN + srow + 32 * i) * K + (long long)ks * W + su;
    struct BStage { f32x4 b[4]; };
#define CASV_LOAD_B(G, KT)                                                                                              \
    {                                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[0]) : "v"(bp[0]), "n"((KT) * QK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[1]) : "v"(bp[1]), "n"((KT) * QK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[2]) : "v"(bp[2]), "n"((KT) * QK2 * 4));      \
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(G.b[3]) : "v"(bp[3]), "n"((KT) * QK2 * 4));      \
    }
#define CASV_LOAD_A(J)                                                                                                  \
    if constexpr ((J) < NT) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(areg[(J) < NT ? (J) : 0]) : "v"(arow), "n"((J) * QK2 * 4));
#define CASV_LD16(DST, PTR, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF))
    auto store_b = [&](const BStage& gs, int buf) {
        float* sb = s_stage + buf * QSTAGE + (QBM + srow) * QLD + su;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sb + 32 * i * QLD) = gs.b[i];
    };
    auto store_a = [&](const f32x4& a, int buf) {
        *reinterpret_cast<f32x4*>(s_stage + buf * QSTAGE + srow * QLD + su) = a;
    };
    const int a_off = l31 * QLD + 4 * lh, b_off = (QBM + wave * 32 + l31) * QLD + 4 * lh;
    f32x16 acc;
    auto compute = [&](int buf) {
        const float* base = s_stage + buf * QSTAGE;
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

    // ---- P: state and the inputs of the first step ----
    const bool a_on = job.dOut != nullptr, cfin_on = job.dc_fin != nullptr;
    f32x4 dc, mk;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        dc[e] = cfin_on ? job.dc_fin[(long long)mrow * W + u0 + e] : 0.0f;
        mk[e] = job.mask ? job.mask[u0 + e] : 1.0f;
    }
    auto time_of = [&](int k) { return job.reverse ? len - 1 - k : k; };
    // (an absent input is read from a row that exists and not used)
    auto load_pin = [&](PIn& in, int k) {
        const int t = time_of(k);
        const float* xg = job.Gt + ((long long)t * B + mrow) * K + ug * 128 + su;
        const float* xc = job.Cs + ((long long)t * B + mrow) * W + u0;
        const float* xa = a_on ? job.dOut + ((long long)t * B + mrow) * job.ld_out + u0 : xc;
        const float* xp = k > 0 ? job.Cs + ((long long)time_of(k - 1) * B + mrow) * W + u0 : (job.c0 ? job.c0 + (long long)mrow * W + u0 : xc);
        CASV_LD16(in.a, xa, 0);
        CASV_LD16(in.gi, xg, 0); CASV_LD16(in.gf, xg, 128); CASV_LD16(in.gg, xg, 256); CASV_LD16(in.go, xg, 384);
        CASV_LD16(in.cell, xc, 0); CASV_LD16(in.cp, xp, 0);
    };
#define CASV_PIN_REGS(IN) "+v"(IN.a), "+v"(IN.gi), "+v"(IN.gf), "+v"(IN.gg), "+v"(IN.go), "+v"(IN.cell), "+v"(IN.cp)
    PIn in;
    load_pin(in, len - 1);
    asm volatile("s_waitcnt vmcnt(0)" : CASV_PIN_REGS(in));

    for (int i = 0; i < len; ++i) {
        const int k = len - 1 - i, t = time_of(k);
        // =========== P: dZ(t) of my cells ===========
        f32x4 bv;
        if (i > 0) {
            if (!wait_deps(Dep{r_need, 4u * (unsigned)i}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            const float* xb = job.dRec + ((long long)time_of(k + 1) * B + mrow) * W + u0;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(bv) : "v"(xb));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = job.dh_fin ? job.dh_fin[(long long)mrow * W + u0 + e] : 0.0f;
        }
        {
            const bool cp_on = k > 0 || job.c0 != nullptr;
            f32x4 zi, zf, zg, zo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {         // lstm_bwd_kernel's arithmetic (train_kernels.hip)
                float dh = 0.f;
                if (a_on) dh += in.a[e] * mk[e];
                if (i > 0 || job.dh_fin) dh += bv[e];
                const float ig = in.gi[e], fg = in.gf[e], gg = in.gg[e], og = in.go[e];
                const float cp = cp_on ? in.cp[e] : 0.0f;
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
                float* z = job.dZ + ((long long)t * B + mrow) * K + ug * 128 + su;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(z), "v"(zi) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:128 sc1" :: "v"(z), "v"(zf) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:256 sc1" :: "v"(z), "v"(zg) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:384 sc1" :: "v"(z), "v"(zo) : "memory");
            }
        }
        publish(z_mine);

        // =========== G: my share of dRec(t) ===========
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        {
            BStage g0, g1;
            CASV_LOAD_B(g0, 0)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g0.b[2]), "+v"(g0.b[3]));
            store_b(g0, 0);
            if constexpr (NT > 1) CASV_LOAD_B(g1, 1)
            if constexpr (NT > 2) CASV_LOAD_B(g0, 2)
            if (!wait_deps(Dep{z_need, (unsigned)((NT / 4) * (i + 1))}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
            const float* arow = job.dZ + ((long long)t * B + mrow) * K + (long long)ks * W + su;
            f32x4 areg[NT];
            CASV_LOAD_A(0) CASV_LOAD_A(1) CASV_LOAD_A(2) CASV_LOAD_A(3) CASV_LOAD_A(4) CASV_LOAD_A(5) CASV_LOAD_A(6) CASV_LOAD_A(7)
            CASV_LOAD_A(8) CASV_LOAD_A(9) CASV_LOAD_A(10) CASV_LOAD_A(11) CASV_LOAD_A(12) CASV_LOAD_A(13) CASV_LOAD_A(14) CASV_LOAD_A(15)
            // ... and what the next step's cells need that does not depend on this step (none left after the last)
            load_pin(in, k > 0 ? k - 1 : 0);
            // everything requested so far has arrived
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[j]));
            asm volatile("s_waitcnt vmcnt(0)" : CASV_PIN_REGS(in));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0.b[0]), "+v"(g0.b[1]), "+v"(g0.b[2]), "+v"(g0.b[3]),
                                                "+v"(g1.b[0]), "+v"(g1.b[1]), "+v"(g1.b[2]), "+v"(g1.b[3]));
            store_a(areg[0], 0);
            __syncthreads();
#define CASV_REC_STAGE(G, J)                                                                                            \
            if constexpr ((J) < NT) {                                                                                   \
                if constexpr ((J) + 1 < NT) {                                                                           \
                    if constexpr ((J) >= 2 && (J) + 2 < NT)                                                             \
                        asm volatile("s_waitcnt vmcnt(4)" : "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3]));    \
                    else if constexpr ((J) >= 2)                                                                        \
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(G.b[0]), "+v"(G.b[1]), "+v"(G.b[2]), "+v"(G.b[3]));    \
                    store_a(areg[(J) + 1 < NT ? (J) + 1 : 0], ((J) + 1) & 1);                                           \
                    store_b(G, ((J) + 1) & 1);                                                                          \
                }                                                                                                       \
                if constexpr ((J) + 3 < NT) CASV_LOAD_B(G, (J) + 3)                                                     \
                compute((J) & 1);                                                                                       \
                __syncthreads();                                                                                        \
            }
            CASV_REC_STAGE(g1, 0) CASV_REC_STAGE(g0, 1) CASV_REC_STAGE(g1, 2) CASV_REC_STAGE(g0, 3)
            CASV_REC_STAGE(g1, 4) CASV_REC_STAGE(g0, 5) CASV_REC_STAGE(g1, 6) CASV_REC_STAGE(g0, 7)
            CASV_REC_STAGE(g1, 8) CASV_REC_STAGE(g0, 9) CASV_REC_STAGE(g1, 10) CASV_REC_STAGE(g0, 11)
            CASV_REC_STAGE(g1, 12) CASV_REC_STAGE(g0, 13) CASV_REC_STAGE(g1, 14) CASV_REC_STAGE(g0, 15)
#undef CASV_REC_STAGE
        }
        {
            float* out = job.dRec + (long long)t * B * W + ct * QBN + wave * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < B) atomicAdd(out + (long long)m * W, acc[r]);
            }
        }
        publish(r_mine);
    }
#undef CASV_PIN_REGS
#undef CASV_LD16
#undef CASV_LOAD_A
#undef CASV_LOAD_B
    // dL/dc of the layer's initial state
    if (row_ok) *reinterpret_cast<f32x4*>(job.dc_out + (long long)mrow * W + u0) = dc;
}

template <class K>
static int recb_blocks_per_cu(K kernel) {
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
    n = n > 2 ? 2 : (n < 0 ? 0 : n);
    cache[{dev, f}] = n;
    return n;
}

size_t train_recurrence_bwd_counter_bytes(int B) { return ((size_t)2 * ((B + QBM - 1) / QBM) * 8 * 32 + 32) * sizeof(unsigned); }

template <int NT> static int recb_grid(const RecBwdArgs& ra, int ncu) {
    const int grid = ra.njobs * ((ra.B + QBM - 1) / QBM) * NT;
    return grid <= recb_blocks_per_cu(train_recurrence_bwd_kernel<NT>) * ncu ? grid : 0;
}
// Workgroups of the launch, or 0: no persistent form for this shape on this device (whole column tiles of 128 units only)
int train_recurrence_bwd_grid(const RecBwdArgs& ra, int ncu) {
    if (ra.W % 128 || ra.njobs < 1 || ra.njobs > 2 || ra.B < 1) return 0;
    switch (ra.W / 32) {
        case 4: return recb_grid<4>(ra, ncu);
        case 8: return recb_grid<8>(ra, ncu);
        case 12: return recb_grid<12>(ra, ncu);
        case 16: return recb_grid<16>(ra, ncu);
        default: return 0;
    }
}
void launch_train_recurrence_bwd(const RecBwdArgs& ra, int grid, hipStream_t stream) {
    switch (ra.W / 32) {
        case 4: hipLaunchKernelGGL((train_recurrence_bwd_kernel<4>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 8: hipLaunchKernelGGL((train_recurrence_bwd_kernel<8>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 12: hipLaunchKernelGGL((train_recurrence_bwd_kernel<12>), dim3(grid), dim3(256), 0, stream, ra); break;
        case 16: hipLaunchKernelGGL((train_recurrence_bwd_kernel<16>), dim3(grid), dim3(256), 0, stream, ra); break;
        default: break;
    }
}

}  // namespace casv
