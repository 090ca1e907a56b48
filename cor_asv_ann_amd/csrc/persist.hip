// Persistent decoder for small batches: ONE launch runs all S greedy decode steps (seq2seq.py:1215-1354) of up to a few
// hundred rows, where a step is a handful of short dependent GEMMs and the launch-per-kernel path is bound by launch and
// memory latency (BASELINE configs[1]: 256 rows, depth 2, width 256 -- 5 launches of ~20 us per character).
//
// Decomposition.  Rows are cut into blocks of 16, hidden units into groups of 16.  Workgroups (256 threads, all
// co-resident) have fixed roles for the whole launch:
//   LSTM  one tile = (layer, row block, unit group): 16 rows x 16 units x 4 gates, wave g = gate g, one
//         v_mfma_f32_16x16x4_f32 accumulator per wave over the FULL K range [x | ctx | h], then the cell (common.h) --
//         gate pre-activations meet in LDS, h', c' go to the slot of the next step;
//   ATT   four rows of a row block, one wave per row: the per-step kernel's functions (row_kernels.h); a wave keeps its row for
//         the launch, carries the next window over in registers and requests the window's rows ahead of its query;
//   PLAIN four 16x16 column tiles of the attention query h.W_a (for the NEXT step) or of the logits h.E^T.
// The softmax is not a phase of its own: every layer-1 tile recomputes max / sum / argmax of its 16 rows from the logits
// (a few KB) and turns logits into the fed-back distribution while it loads them as its A operand; the tile of unit group 0
// also emits the step's character and probability.
//
// Numerics are those of the per-step kernels, bit for bit: the 16x16x4 MFMA contracts four k per instruction as an
// ordered fmaf chain, and its k groups are fed in the order {0,4,1,5},{2,6,3,7},{8,12,9,13},{10,14,11,15} of every
// 16-k tile -- the order in which gemm.hip / gemm_skinny.hip's 32x32x2 chain visits them (measured identical on random
// data, and identical to a scalar fmaf chain on the host); cell, attention and softmax are the shared functions.  A row
// therefore decodes to the same characters and probabilities whichever path its batch takes (tested).
//
// Hand-offs (MI355X: 8 XCDs with private L2s, per-CU L1s): every buffer that crosses workgroups is indexed by the step
// (slot s + 1 holds the outputs of step s), so no address is rewritten and none is read before it was written.
// Producers store the payload write-through (sc1), drain (s_waitcnt vmcnt(0) in every storing wave), meet at the
// workgroup barrier, and ONE lane adds to the row block's monotonic counter (agent scope).  Consumers: one lane polls the
// counters it needs (relaxed, agent scope, s_sleep between polls, bounded in TIME), a barrier, and then EVERY load of the
// handed-off rows is a load that goes past the L2 (load_sc1) -- no agent-scope acquire, which would also empty this XCD's L2
// of the weights (MI355X_MICROARCH.md, "Valid forms": sc1 stores + drain + counter, sc1 loads behind the poll + barrier).
// Counters only grow (target = tiles per step x steps done) and are zeroed by a memset ahead of the launch.
// A wait that lasts longer than PERSIST_WAIT_TICKS sets the abort word; every other wait sees it and the launch drains.
// The grids are sized from the occupancy the runtime reports for these kernels (persist_*_blocks_per_cu), not from assumed
// register counts: every workgroup must be resident at once.
#include "common.h"
#include "row_kernels.h"
#include "handoff.h"
#include <math.h>
#include <map>
#include <mutex>
#include <utility>

namespace casv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build (-DCASV_PERSIST_PROF): the first workgroup of every role adds the wall-clock ticks (10 ns) it spends
// in each part of its tasks to pa.prof[role * 8 + part].
#ifdef CASV_PERSIST_PROF
#define PROF_T(var) const unsigned long long var = wall_clock64()
#define PROF_ADD(slot, t1, t0) do { if (threadIdx.x == 0 && prof_on) atomicAdd(const_cast<unsigned long long*>(pa.prof) + (slot), (unsigned long long)((t1) - (t0))); } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, t1, t0)
#endif

struct RowStat { float m, sum; int nan0; };   // softmax statistics of one row; nan0: mode 1 wrote NaN over p[0]

constexpr int PDB = 12;                         // weight tiles (16 k each) in flight per wave; a multiple of 3 (k_loop)

// The workgroup's 16 activation rows [x | ctx | h] are staged ONCE in LDS (they were written by other workgroups a moment
// ago, so every line is a miss in this XCD's L2: fetched one by one along the K loop they would cost a memory round trip per
// few tiles; together they cost one).  16 threads per row, 16 B per load.
// Inside every group of 16 k the rows are stored in MFMA-group order (position 4*kg + q = the k that k group kg contracts in
// instruction q, as the packed weights are): a lane's operands of one tile are ONE 16-B LDS read, no cross-lane traffic.
__device__ __forceinline__ int perm16(const int k) {       // natural k -> position inside its 16-group
    // positions of k = 0..15: 0,8,1,9, 4,12,5,13, 2,10,3,11, 6,14,7,15
    return ((k & 1) << 3) | (k & 4) | ((k & 8) >> 2) | ((k & 2) >> 1);
}
constexpr int STAGE_NB = 12;
struct RowSeg { const float* base; int ld, width; };      // rows `base + row * ld`, `width` floats wide (a multiple of 16)
// Up to three K segments of the 16 rows at once: chunk f (16 B) of the concatenated row belongs to thread f % 16 of the row,
// STAGE_NB chunks of a thread are in flight together, whichever segments they come from.
__device__ __forceinline__ void stage_rows(float* s_a, const int lda, const int koff, const RowSeg g0, const RowSeg g1,
                                           const RowSeg g2, const int rb, const int R, const int tid) {
    const int r = tid >> 4, c0 = tid & 15;
    int row = rb * 16 + r; row = row < R ? row : R - 1;
    const float* p0 = g0.base + (long long)row * g0.ld;
    const float* p1 = g1.width ? g1.base + (long long)row * g1.ld : p0;
    const float* p2 = g2.width ? g2.base + (long long)row * g2.ld : p0;
    const int b0 = g0.width >> 2, b1 = b0 + (g1.width >> 2), n4 = b1 + (g2.width >> 2);
    const int nmine = n4 > c0 ? (n4 - c0 + 15) >> 4 : 0;          // this thread's chunks f = c0 + 16 i
    float* dst = s_a + r * lda + koff + (c0 >> 2) * 16;           // chunk f sits in 16-group f >> 2 = (c0 >> 2) + 4 i
    const int k0 = 4 * (c0 & 3);
    const int q0 = perm16(k0), q1 = perm16(k0 + 1), q2 = perm16(k0 + 2), q3 = perm16(k0 + 3);
    for (int i0 = 0; i0 < nmine; i0 += STAGE_NB) {
        f32x4 v[STAGE_NB];
#pragma unroll
        for (int j = 0; j < STAGE_NB; ++j) {
            const int i = i0 + j < nmine ? i0 + j : nmine - 1;
            const int f = c0 + 16 * i;
            const float4 x = load_sc1(f < b0 ? p0 + 4 * f : f < b1 ? p1 + 4 * (f - b0) : p2 + 4 * (f - b1));
            v[j] = f32x4{x.x, x.y, x.z, x.w};
        }
#pragma unroll
        for (int j = 0; j < STAGE_NB; ++j) {
            if (i0 + j < nmine) {
                float* d = dst + (i0 + j) * 64;
                d[q0] = v[j][0]; d[q1] = v[j][1]; d[q2] = v[j][2]; d[q3] = v[j][3];
            }
        }
    }
}
__device__ __forceinline__ int permv(const int v) { return (v & ~15) | perm16(v & 15); }

// Weight tiles kt_begin .. kt_begin + PDB - 1 of this lane's packed row: issued ahead of the dependency wait (weights
// depend on nothing), so their latency hides behind it.
struct BRing { f32x4 t[PDB]; };
__device__ __forceinline__ void ring_start(BRing& ring, const float* __restrict__ b, const int kt_begin, const int nt) {
#pragma unroll
    for (int q = 0; q < PDB; ++q) {
        const int kt = kt_begin + q < nt ? kt_begin + q : nt - 1;
        ring.t[q] = *reinterpret_cast<const f32x4*>(b + (long long)kt * 16);
    }
}

// The K loop of one wave: acc += A[16 rows][K] . B[16 cols][K]^T in the k order of the module comment.  A from the staged
// rows (row stride lda), B from global memory through the ring.  Lane: row / column = lane & 15, k group = lane >> 4.
__device__ __forceinline__ f32x4 k_loop(const float* s_a, const int lda, const float* __restrict__ b, BRing& ring,
                                        const int kt_begin, const int nt, const int lane) {
    const int kg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (kt_begin >= nt) return acc;
    const float* arow = s_a + (lane & 15) * lda + 4 * kg;
    auto mma = [&](const f32x4 a, const f32x4 bq) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], bq[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], bq[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], bq[3], acc, 0, 0, 0);
    };
    // the A fragments run two tiles ahead of their MFMAs in three rotating registers (one ds_read_b128 per tile: issued right
    // before its use, its latency would be exposed once per tile)
    auto afrag = [&](const int t) { return *reinterpret_cast<const f32x4*>(arow + (t < nt ? t : nt - 1) * 16); };
    int kt = kt_begin;
    f32x4 a[3] = {afrag(kt), afrag(kt + 1), afrag(kt + 2)};
    for (; kt + PDB <= nt; kt += PDB) {
#pragma unroll
        for (int q = 0; q < PDB; ++q) {
            mma(a[q % 3], ring.t[q]);
            a[q % 3] = afrag(kt + q + 3);
            const int nx = kt + PDB + q < nt ? kt + PDB + q : nt - 1;          // past the end: a valid, unused re-load
            ring.t[q] = *reinterpret_cast<const f32x4*>(b + (long long)nx * 16);
        }
    }
    const int rest = nt - kt;
#pragma unroll
    for (int q = 0; q < PDB - 1; ++q)
        if (rest > q) { mma(a[q % 3], ring.t[q]); a[q % 3] = afrag(kt + q + 3); }
    return acc;
}

// Softmax statistics + greedy pick (the arithmetic of softmax_kernel, decode_kernels.hip) of FOUR rows by one wave, V <= 256:
// one row per quarter of the wave, lane j of a quarter owns the entries v = j, j + 16, ..., j + 240.  The one-row kernel sums a
// row as per-lane partial sums over v = l, l + 64, l + 128, l + 192 (lane l of 64, ascending) followed by butterflies over the
// partners 32, 16, 8, 4, 2, 1; the entries of leaves l = j, j + 16, j + 32, j + 48 sit in ONE lane here, so the four leaf sums
// and the first two butterfly levels -- (p[0] + p[2]) + (p[1] + p[3]), additions commute bit for bit -- are that lane's own
// arithmetic and only the partners 8, 4, 2, 1 cross lanes: the same additions in the same tree, the same bits.  Maximum, NaN count
// and pick do not depend on the order.  Against four rows one after another on all 64 lanes (before): a sixth of the butterfly
// steps, the four rows' transcendentals side by side.  FULL: V = Vp = 256, no entry is conditional.
template <class T, class Op> __device__ __forceinline__ T quarter_butterfly(T v, Op op) {
    v = op(v, lane_xor<8>(v)); v = op(v, lane_xor<4>(v)); v = op(v, lane_xor<2>(v)); v = op(v, lane_xor<1>(v));
    return v;
}
template <bool FULL>
__device__ __forceinline__ void row_stats_quarter(float* xrow, const int V, const int Vp, const int mode, const int lane,
                                                  const bool emit, int* out_idx, float* out_prob, int* nan_flag) {
    const int j = lane & 15;
    float* xp = xrow + perm16(j);                 // entry v = j + 16 i sits at position 16 i + perm16(j) of the staged row
    float xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = (FULL || j + 16 * i < V) ? xp[16 * i] : 0.0f;
    float m = -INFINITY, nan = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (FULL || j + 16 * i < V) { m = fmaxf(m, xv[i]); nan += (xv[i] != xv[i]) ? 1.0f : 0.0f; }
    m = quarter_butterfly(m, [](float a, float b) { return fmaxf(a, b); });
    nan = quarter_butterfly(nan, [](float a, float b) { return a + b; });
    if (nan > 0.0f) m = __builtin_nanf("");
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (FULL || j + 16 * i < V) xv[i] = expf(xv[i] - m);                  // xv now holds exp(x - m)
    float p[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        p[a] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (FULL || j + 16 * (a + 4 * k) < V) p[a] += xv[a + 4 * k];
    }
    float sum = (p[0] + p[2]) + (p[1] + p[3]);
    sum = quarter_butterfly(sum, [](float a, float b) { return a + b; });
    float best = -INFINITY, p0 = 0.0f;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int v = j + 16 * i;
        const float pv = (FULL || v < V) ? xv[i] / sum : 0.0f;
        if (v == 0) p0 = pv;
        if (v >= 1 && (FULL || v < V) && pv > best) { best = pv; bidx = v; }
        xv[i] = pv;                                                           // xv now holds the distribution
    }
    {   // wave_argmax's rule over the quarter: larger value wins, the lower index among equals
        auto step = [&](const float ob, const int oi) { if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; } };
        step(lane_xor<8>(best), lane_xor<8>(bidx)); step(lane_xor<4>(best), lane_xor<4>(bidx));
        step(lane_xor<2>(best), lane_xor<2>(bidx)); step(lane_xor<1>(best), lane_xor<1>(bidx));
    }
    int idx = bidx, nan0 = 0;
    float pr = best;
    if (bidx == 0x7fffffff) {                                                 // every candidate NaN: numpy raises here
        if (emit && j == 0 && nan_flag) atomicOr(nan_flag, 1);
        idx = 1; pr = __builtin_nanf("");
    } else if (mode == 1) {
        if (p0 >= best && p0 == p0) nan0 = 1;      // seq2seq.py:1334: NaN over index 0, stays in the feedback (p0: lane j = 0, the only one that uses it)
    }
    if (emit && j == 0) { *out_idx = idx; *out_prob = pr; }
    // the row becomes the fed-back distribution in place (zeros in the K padding beyond V)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int v = j + 16 * i;
        if (FULL || v < Vp) xp[16 * i] = (v == 0 && nan0) ? __builtin_nanf("") : xv[i];
    }
}

// the same for one row of any vocabulary size (values re-read from memory in every pass)
__device__ __forceinline__ RowStat row_stats(const float* x, const int V, const int Vp, const int mode, const int lane,
                                             const bool emit, int* out_idx, float* out_prob, int* nan_flag) {
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, x[permv(v)]);
    float anynan = 0.0f;
    for (int v = lane; v < V; v += 64) anynan += (x[permv(v)] != x[permv(v)]) ? 1.0f : 0.0f;
    m = wave_max(m);
    if (wave_sum(anynan) > 0.0f) m = __builtin_nanf("");
    float sum = 0.0f;
    for (int v = lane; v < V; v += 64) sum += expf(x[permv(v)] - m);
    sum = wave_sum(sum);
    float best = -INFINITY; int bidx = 0x7fffffff;
    float p0 = 0.0f;
    for (int v = lane; v < Vp; v += 64) {
        const float pv = v < V ? expf(x[permv(v)] - m) / sum : 0.0f;
        if (v == 0) p0 = pv;
        if (v >= 1 && v < V && pv > best) { best = pv; bidx = v; }
    }
    wave_argmax(best, bidx);
    p0 = __shfl(p0, 0, 64);
    RowStat st; st.m = m; st.sum = sum; st.nan0 = 0;
    int idx = bidx; float pr = best;
    if (bidx == 0x7fffffff) {
        if (emit && lane == 0 && nan_flag) atomicOr(nan_flag, 1);
        idx = 1; pr = __builtin_nanf("");
    } else if (mode == 1) {
        if (p0 >= best && p0 == p0) st.nan0 = 1;             // seq2seq.py:1334: NaN over index 0, stays in the feedback
    }
    if (emit && lane == 0) { *out_idx = idx; *out_prob = pr; }
    return st;
}

__global__ __launch_bounds__(256, 2) void persist_decode_kernel(const PersistArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float s_a[];         // 16 staged activation rows, stride pa.lda
    __shared__ float s_gate[4][16 * 16];
    __shared__ float s_m[16], s_sum[16];
    __shared__ float s_bias[64];                                        // the tile's bias values [gate][unit]
    __shared__ int s_nan0[16];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = pa.R, D = pa.D, W = pa.W, V = pa.V, Vp = pa.Vp, C = pa.C, S = pa.S, lda = pa.lda;
    const int NRB = (R + 15) / 16, NUG = W / 16;
    const int NQ4 = (W / 16 + 3) / 4, NL4 = (Vp / 16 + 3) / 4;         // query / logits tasks per row block (4 column tiles each)
    const int NCNT = D + 3;
    unsigned* const cnt = pa.counters;
    unsigned* const abort_w = pa.counters + (long long)NRB * NCNT * 32;
    auto counter = [&](int rb, int kind) { return cnt + ((long long)rb * NCNT + kind) * 32; };   // kind: 0..D-1 layers, D ctx, D+1 logits, D+2 query
    const long long RW = (long long)R * W;
    const int g = blockIdx.x;
    const int kg4 = 4 * (lane >> 4);
#ifdef CASV_PERSIST_PROF
    const bool prof_on = pa.prof && (g == 0 || g == pa.g_lstm || g == pa.g_lstm + pa.g_att);
    if (pa.prof && tid == 0) {          // where this workgroup runs: XCC_ID (register 20) and HW_ID (register 4)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pa.prof[32 + g] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    if (g < pa.g_lstm) {
        // ------------------------------------------------------------------ LSTM tiles
        // (a static s_setprio 1 / 3 for the tile waves -- ahead of the attention / plain workgroup that shares their SIMDs on 192 of the
        // 256 CUs -- measured in round 5: 8.25-8.30 ms per batch of configs[1] against 8.29-8.39 without, within the noise; not kept)
        const int NT = NRB * NUG;
#ifdef CASV_PERSIST_PROF
        unsigned long long prof_prev = wall_clock64();          // slot 5 / 13: from a task's publish to the next one's wait
        const unsigned long long prof_begin = prof_prev;
#endif
        for (int s = 0; s <= S; ++s) {
            for (int n = 1; n <= D; ++n) {
                if (s == S && n > 1) break;
                for (int t = g; t < NT; t += pa.g_lstm) {
                    const int rb = t / NUG, ug = t % NUG;
                    if (s == S && ug != 0) continue;               // after the last step: only the outputs of its logits
                    const bool top = n == D, first = n == 1;
                    const PersistLayer& L = pa.layer[n - 1];
                    const int wx = first ? Vp : W, nt = L.Kt / 16;
                    const int kt_begin = (first && s == 0) ? Vp / 16 : 0;        // step 0: the fed-back distribution is all zero
                    const float* b = L.w + ((long long)((ug * 4 + wave) * 16 + (lane & 15))) * L.Kt + kg4;
                    BRing ring;
                    if (s < S) ring_start(ring, b, kt_begin, nt);
                    Dep dx{nullptr, 0}, dc{nullptr, 0}, dh{nullptr, 0};
                    if (first) { if (s > 0) dx = Dep{counter(rb, D + 1), (unsigned)(s * NL4)}; }
                    else dx = Dep{counter(rb, n - 2), (unsigned)((s + 1) * NUG)};
                    if (s < S) {
                        if (top) dc = Dep{counter(rb, D), (unsigned)((s + 1) * 4)};
                        if (s > 0) dh = Dep{counter(rb, n - 1), (unsigned)(s * NUG)};
                    }
                    PROF_T(t0);
#ifdef CASV_PERSIST_PROF
                    PROF_ADD((n == 1 ? 0 : 8) + 5, t0, prof_prev);
#endif
                    if (!wait_deps(dx, dc, dh, abort_w, &s_ok)) return;
                    PROF_T(t1);
                    PROF_ADD((n == 1 ? 0 : 8) + 0, t1, t0);
                    // previous cell state of this thread's (row, unit)
                    const int erow = rb * 16 + (tid >> 4), eu = ug * 16 + (tid & 15);
                    const int erow_c = erow < R ? erow : R - 1;
                    float cprev = 0.0f;
                    if (s < S) cprev = pa.c[n - 1][(long long)s * RW + (long long)erow_c * W + eu];
                    // the tile's bias: requested ahead of the rows, parked in LDS behind them -- read from memory behind the gate
                    // exchange it was a cache round trip on the critical path of every cell
                    float bias_v = 0.0f;
                    if (tid < 64) bias_v = L.bias[(long long)ug * 64 + tid];
                    // the 16 rows of [x | ctx | h] -> LDS: x = the logits of step s-1 (layer 1; slot s) or the output of the layer
                    // below at this step (slot s+1), ctx of this step, h of step s-1 (slot s)
                    {
                        const RowSeg none{nullptr, 0, 0};
                        const RowSeg gx = first ? RowSeg{pa.logits + (long long)s * R * Vp, Vp, Vp} : RowSeg{pa.h[n - 2] + (long long)(s + 1) * RW, W, W};
                        const RowSeg gc = top ? RowSeg{pa.ctx + (long long)(s + 1) * R * C, C, C} : none;
                        const RowSeg gh{pa.h[n - 1] + (long long)s * RW, W, W};
                        if (s == S) stage_rows(s_a, lda, 0, gx, none, none, rb, R, tid);
                        else if (first && s == 0) { if (top) stage_rows(s_a, lda, wx, gc, gh, none, rb, R, tid); else stage_rows(s_a, lda, wx, gh, none, none, rb, R, tid); }
                        else if (top) stage_rows(s_a, lda, 0, gx, gc, gh, rb, R, tid);
                        else stage_rows(s_a, lda, 0, gx, gh, none, rb, R, tid);
                    }
                    if (tid < 64) s_bias[tid] = bias_v;
                    __syncthreads();
                    if (first && s > 0) {
                        // softmax statistics of the 16 rows from the logits of step s-1 (unit group 0 reports the character),
                        // then the rows become the fed-back distribution in place (softmax_kernel: expf(x - m) / sum)
                        const int rq = rb * 16 + wave * 4 + (lane >> 4);       // this quarter-wave's row
                        const int rqc = rq < R ? rq : R - 1;
                        float* const xq = s_a + (wave * 4 + (lane >> 4)) * lda;
                        const bool emit_q = ug == 0 && rq < R;
                        int* const oi_q = pa.out_idx + (long long)rqc * S + (s - 1);
                        float* const op_q = pa.out_prob + (long long)rqc * S + (s - 1);
                        if (V == 256 && Vp == 256) row_stats_quarter<true>(xq, V, Vp, pa.mode, lane, emit_q, oi_q, op_q, pa.nan_flag);
                        else if (V <= 256) row_stats_quarter<false>(xq, V, Vp, pa.mode, lane, emit_q, oi_q, op_q, pa.nan_flag);
                        else {
                            // large vocabularies: one row at a time on the whole wave, values re-read in every pass, then the rows are rewritten
                            RowStat q[4];
                            for (int i = 0; i < 4; ++i) {
                                const int rr = rb * 16 + wave * 4 + i, r = rr < R ? rr : R - 1;
                                q[i] = row_stats(s_a + (wave * 4 + i) * lda, V, Vp, pa.mode, lane, ug == 0 && rr < R,
                                                 pa.out_idx + (long long)r * S + (s - 1), pa.out_prob + (long long)r * S + (s - 1), pa.nan_flag);
                            }
                            if (lane == 0) {
                                for (int i = 0; i < 4; ++i) { s_m[wave * 4 + i] = q[i].m; s_sum[wave * 4 + i] = q[i].sum; s_nan0[wave * 4 + i] = q[i].nan0; }
                            }
                            __syncthreads();
                            const int r = tid >> 4;
                            const float m = s_m[r], sum = s_sum[r];
                            float* xrow = s_a + r * lda;
                            for (int v = tid & 15; v < Vp; v += 16) {
                                float pv = v < V ? expf(xrow[permv(v)] - m) / sum : 0.0f;
                                if (pa.mode == 1 && v == 0 && s_nan0[r]) pv = __builtin_nanf("");
                                xrow[permv(v)] = pv;
                            }
                        }
                        __syncthreads();
                    }
                    PROF_T(t2);
                    PROF_ADD((n == 1 ? 0 : 8) + 1, t2, t1);
                    if (s == S) continue;
                    const f32x4 acc = k_loop(s_a, lda, b, ring, kt_begin, nt, lane);
                    PROF_T(t3);
                    PROF_ADD((n == 1 ? 0 : 8) + 2, t3, t2);
                    // gate g of (row = 4*(lane>>4) + reg, unit = lane & 15) -> LDS; thread (row, unit) runs the cell
#pragma unroll
                    for (int q = 0; q < 4; ++q) s_gate[wave][((lane >> 4) * 4 + q) * 16 + (lane & 15)] = acc[q];
                    __syncthreads();
                    const int e = tid;
                    const float* bias = s_bias + (tid & 15);
                    const float zi = s_gate[0][e] + bias[0], zf = s_gate[1][e] + bias[16], zg = s_gate[2][e] + bias[32], zo = s_gate[3][e] + bias[48];
                    const LstmCellOut cell = lstm_cell(zi, zf, zg, zo, cprev);
                    if (erow < R) {
                        pa.c[n - 1][(long long)(s + 1) * RW + (long long)erow * W + eu] = cell.c;      // read back by this workgroup only
                        store_sc1(pa.h[n - 1] + (long long)(s + 1) * RW + (long long)erow * W + eu, cell.h);
                    }
                    PROF_T(t4);
                    PROF_ADD((n == 1 ? 0 : 8) + 3, t4, t3);
                    publish(counter(rb, n - 1));
                    PROF_T(t5);
                    PROF_ADD((n == 1 ? 0 : 8) + 4, t5, t4);
#ifdef CASV_PERSIST_PROF
                    prof_prev = t5;
#endif
                }
            }
        }
        return;
    }
    if (g < pa.g_lstm + pa.g_att) {
        // ------------------------------------------------------------------ attention rows
        const int ga = g - pa.g_lstm;
        // query -> attention -> context is the longer of the two chains between a step's top layer and the next one's (the other:
        // logits -> layer 1), so what a row can do ahead of its query it does: a wave keeps its row from step to step (one task
        // per workgroup), carries the NEXT step's window over from the weights it has just normalised (att_window_next: the sum
        // that att_window reads back from memory, from registers) and requests the <= 11 rows of u and of the encoder outputs that
        // window attends BEFORE it waits for the query.  Behind the wait: the query row (one round trip), tanh, sums, context.
        if (NRB * 4 <= pa.g_att && W <= 256 && C <= 256) {
            if (ga >= NRB * 4) return;
            const int rb = ga >> 2;
            const int r = rb * 16 + (ga & 3) * 4 + wave;
            const bool live = r < R;
            const int rc = live ? r : R - 1;
            AttnArgs a = pa.att;
            const int ln = a.line ? a.line[rc] : rc / a.rows_per_line;
            const float* ub = a.u + (long long)ln * a.u_line;
            const float* eb = a.enc + (long long)ln * a.enc_line;
            const int T = a.T, W4 = W >> 2, C4 = C >> 2;
            const int ju = lane < W4 ? lane : 0, jc = lane < C4 ? lane : 0;     // (lanes beyond a row's width read a valid address and use nothing)
            AttWin w = att_window(a, rc, 0, lane);                              // from the initial alignment
            for (int s = 0; s < S; ++s) {
                float4 uu[MAXWIN], xx[MAXWIN];
#pragma unroll
                for (int i = 0; i < MAXWIN; ++i) {
                    int sr = w.s_lo + i; sr = sr < T ? sr : T - 1; sr = sr < 0 ? 0 : sr;
                    uu[i] = reinterpret_cast<const float4*>(ub + (long long)sr * a.u_time)[ju];
                    xx[i] = reinterpret_cast<const float4*>(eb + (long long)sr * a.enc_time)[jc];
                }
                PROF_T(t0);
                if (!wait_deps(Dep{counter(rb, D + 2), (unsigned)((s + 1) * NQ4)}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
                PROF_T(t1);
                PROF_ADD(16, t1, t0);
                a.wq = pa.wq + (long long)s * RW;
                a.ctx = pa.ctx + (long long)(s + 1) * R * C;
                float e[MAXWIN];
#pragma unroll
                for (int i = 0; i < MAXWIN; ++i) e[i] = 0.0f;
                if (live) {
                    att_weights<true, MAXWIN>(a, r, s, lane, w, [&](int sr) { return RegRows{uu, sr - w.s_lo}; }, e);
                    att_context<true, MAXWIN>(a, r, lane, w, [&](int sr) { return RegRows{xx, sr - w.s_lo}; }, e);
                }
                PROF_T(t2);
                PROF_ADD(17, t2, t1);
                publish(counter(rb, D));
                PROF_T(t3);
                PROF_ADD(18, t3, t2);
                w = att_window_next(a, w, e, lane);
            }
            return;
        }
        for (int s = 0; s < S; ++s) {
            for (int t = ga; t < NRB * 4; t += pa.g_att) {
                const int rb = t >> 2, q = t & 3;
                PROF_T(t0);
                if (!wait_deps(Dep{counter(rb, D + 2), (unsigned)((s + 1) * NQ4)}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return;
                PROF_T(t1);
                PROF_ADD(16, t1, t0);
                const int r = rb * 16 + q * 4 + wave;
                if (r < R) {
                    AttnArgs a = pa.att;
                    a.wq = pa.wq + (long long)s * RW;
                    a.ctx = pa.ctx + (long long)(s + 1) * R * C;
                    attention_row<true>(a, r, s, lane);
                }
                PROF_T(t2);
                PROF_ADD(17, t2, t1);
                publish(counter(rb, D));
                PROF_T(t3);
                PROF_ADD(18, t3, t2);
            }
        }
        return;
    }
    {
        // ------------------------------------------------------------------ plain tiles: query of the next step, logits
        const int gp = g - pa.g_lstm - pa.g_att;
        const int per_rb = NQ4 + NL4;
        for (int s = -1; s < S; ++s) {                             // s = -1: the query of step 0 from the initial state
            for (int t = gp; t < NRB * per_rb; t += pa.g_plain) {
                const int rb = t / per_rb, k = t % per_rb;
                const bool query = k < NQ4;
                if (s < 0 && !query) continue;
                if (s == S - 1 && query) continue;                 // no step follows the last one
                // this wave's weight rows do not depend on anything: start them ahead of the wait
                const int ctb = (query ? k : k - NQ4) * 4 + wave;
                const int ctc = ctb < (query ? W / 16 : Vp / 16) ? ctb : 0;
                const float* b = (query ? pa.wa : pa.e) + ((long long)(ctc * 16 + (lane & 15))) * W + kg4;
                BRing ring;
                ring_start(ring, b, 0, W / 16);
                PROF_T(t0);
                if (s >= 0) { if (!wait_deps(Dep{counter(rb, D - 1), (unsigned)((s + 1) * NUG)}, Dep{nullptr, 0}, Dep{nullptr, 0}, abort_w, &s_ok)) return; }
                PROF_T(t1);
                PROF_ADD(24, t1, t0);
                const int ct = (query ? k : k - NQ4) * 4 + wave;               // this wave's 16-column tile
                const int nct = query ? W / 16 : Vp / 16;
                stage_rows(s_a, lda, 0, RowSeg{pa.h[D - 1] + (long long)(s + 1) * RW, W, W}, RowSeg{nullptr, 0, 0}, RowSeg{nullptr, 0, 0}, rb, R, tid);
                __syncthreads();
                if (ct < nct) {
                    const f32x4 acc = k_loop(s_a, lda, b, ring, 0, W / 16, lane);
                    const int col = ct * 16 + (lane & 15);
                    const float bias = query ? pa.bUW[col] : 0.0f;
                    float* out = query ? pa.wq + (long long)(s + 1) * RW : pa.logits + (long long)(s + 1) * R * Vp;
                    const int ld = query ? W : Vp;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int orow = rb * 16 + (lane >> 4) * 4 + q;
                        if (orow < R) store_sc1(out + (long long)orow * ld + col, acc[q] + bias);
                    }
                }
                PROF_T(t2);
                PROF_ADD(25, t2, t1);
                publish(counter(rb, query ? D + 2 : D + 1));
                PROF_T(t3);
                PROF_ADD(26, t3, t2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent encoder (seq2seq.py:237-314) for the same small batches: the BiLSTM layer and the stacked layers as tiles of
// 16 lines x 16 units that hand rows to each other through memory, one launch instead of one per (layer, time) cell.
//   phase A  the two directions of layer 1, time step by time step (a tile needs all unit groups of its own direction at the
//            previous step);
//   phase B  layers 2..D along anti-diagonals (cell (n, t) needs (n-1, t) and (n, t-1)); layer 2 at time t reads the backward
//            output of time t, i.e. all of phase A for t = 0 -- hence two phases, each walked in dependency order.
// The cell state of a tile never leaves its thread's register; outputs go to the same buffers, in the same k order and with
// the same cell function as the per-step launches: identical bits (tested).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PENC_MAXT = 8;        // tiles of one phase a workgroup may own

__global__ __launch_bounds__(256, 2) void persist_encode_kernel(const PersistEncArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float s_a[];
    __shared__ float s_gate[4][16 * 16];
    __shared__ float s_bias[64];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = pa.B, T = pa.T, D = pa.D, W = pa.W, lda = pa.lda;
    const int NRB = (B + 15) / 16, NUG = W / 16, NT = NRB * NUG;
    const int NCNT = D + 1;                                        // counters per row block: fw, bw, layers 2..D
    unsigned* const abort_w = pa.counters + (long long)NRB * NCNT * 32;
    auto counter = [&](int rb, int kind) { return pa.counters + ((long long)rb * NCNT + kind) * 32; };
    const int g = blockIdx.x, G = gridDim.x;
    const int kg4 = 4 * (lane >> 4);
    const int er = tid >> 4, ec = tid & 15;                        // this thread's (row, unit) of a tile in the cell epilogue
#ifdef CASV_PERSIST_PROF
    const bool prof_on = pa.prof && g == 0;
    int pslot = 0;                                                 // 0..4 phase A, 8..12 phase B (wait, stage, K loop, cell, publish)
    const unsigned long long tk0 = wall_clock64();
#endif

    // One cell: tile (kind, rb, ug) at time t, the `nth` step of its recurrence.  x rows: xbase + row * xld (width kx);
    // h rows of the previous step: hprev + row * hld.  Output h -> hout + row * hld (+ unit).
    auto cell = [&](const PersistLayer& L, int kx, const float* xbase, long long xld, const float* hprev, float* hout,
                    long long hld, int rb, int ug, bool first, float& creg, unsigned* done, Dep dx, Dep dh, Dep d3) -> bool {
        const int nt = first ? kx / 16 : L.Kt / 16;                 // zero initial state: the recurrent segment is skipped
        const float* b = L.w + ((long long)((ug * 4 + wave) * 16 + (lane & 15))) * L.Kt + kg4;
        BRing ring;
        ring_start(ring, b, 0, nt);
        PROF_T(t0);
        if (!wait_deps(dx, dh, d3, abort_w, &s_ok)) return false;
        PROF_T(t1);
        float bias_v = 0.0f;                                        // (ahead of the rows, parked in LDS behind them: see the decoder's tiles)
        if (tid < 64) bias_v = L.bias[(long long)ug * 64 + tid];
        stage_rows(s_a, lda, 0, RowSeg{xbase, (int)xld, kx}, RowSeg{hprev, (int)hld, first ? 0 : W}, RowSeg{nullptr, 0, 0}, rb, B, tid);
        if (tid < 64) s_bias[tid] = bias_v;
        __syncthreads();
        PROF_T(t2);
        const f32x4 acc = k_loop(s_a, lda, b, ring, 0, nt, lane);
        PROF_T(t3);
        PROF_ADD(pslot + 0, t1, t0); PROF_ADD(pslot + 1, t2, t1); PROF_ADD(pslot + 2, t3, t2);
#pragma unroll
        for (int q = 0; q < 4; ++q) s_gate[wave][((lane >> 4) * 4 + q) * 16 + (lane & 15)] = acc[q];
        __syncthreads();
        const float* bias = s_bias + ec;
        const float zi = s_gate[0][tid] + bias[0], zf = s_gate[1][tid] + bias[16], zg = s_gate[2][tid] + bias[32], zo = s_gate[3][tid] + bias[48];
        const LstmCellOut c = lstm_cell(zi, zf, zg, zo, first ? 0.0f : creg);
        creg = c.c;
        const int row = rb * 16 + er;
        if (row < B) store_sc1(hout + (long long)row * hld + ug * 16 + ec, c.h);
        PROF_T(t4);
        publish(done);
        PROF_T(t5);
        PROF_ADD(pslot + 3, t4, t3); PROF_ADD(pslot + 4, t5, t4);
        return true;
    };

    // ---- phase A: layer 1, forward and backward
    {
        float creg[PENC_MAXT];
#pragma unroll
        for (int i = 0; i < PENC_MAXT; ++i) creg[i] = 0.0f;
        const long long hld = (long long)T * 2 * W, xld = (long long)T * W;
        for (int st = 0; st < T; ++st) {
#pragma unroll
            for (int i = 0; i < PENC_MAXT; ++i) {
                const int tile = g + i * G;
                if (tile >= 2 * NT) break;
                const int dir = tile / NT, rb = (tile % NT) / NUG, ug = tile % NUG;
                const int t = dir == 0 ? st : T - 1 - st, tp = dir == 0 ? t - 1 : t + 1;
                float* H = pa.H1 + dir * W;
                const Dep dh = st > 0 ? Dep{counter(rb, dir), (unsigned)(st * NUG)} : Dep{nullptr, 0};
                if (!cell(pa.l1[dir], W, pa.x0 + (long long)t * W, xld, H + (long long)tp * 2 * W, H + (long long)t * 2 * W, hld, rb, ug,
                          st == 0, creg[i], counter(rb, dir), Dep{nullptr, 0}, dh, Dep{nullptr, 0})) return;
                if (st == T - 1) {                                  // final cell state of this direction
                    const int row = rb * 16 + er;
                    if (row < B) pa.cfin[(long long)(dir == 0 ? D : 0) * B * W + (long long)row * W + ug * 16 + ec] = creg[i];
                }
            }
        }
    }
#ifdef CASV_PERSIST_PROF
    pslot = 8;
    if (prof_on && tid == 0) atomicAdd(const_cast<unsigned long long*>(pa.prof) + 16, wall_clock64() - tk0);     // phase A in all
    const unsigned long long tk1 = wall_clock64();
#endif
    // ---- phase B: layers 2..D along anti-diagonals
    if (D >= 2) {
        float creg[PENC_MAXT];
#pragma unroll
        for (int i = 0; i < PENC_MAXT; ++i) creg[i] = 0.0f;
        for (int k = 0; k < T + D - 2; ++k) {
#pragma unroll
            for (int i = 0; i < PENC_MAXT; ++i) {
                const int tile = g + i * G;
                if (tile >= (D - 1) * NT) break;
                const int n = 2 + tile / NT, rb = (tile % NT) / NUG, ug = tile % NUG;
                const int t = k - (n - 2);
                if (t < 0 || t >= T) continue;
                const int win = n == 2 ? 2 * W : W;
                const float* xin = n == 2 ? pa.H1 : pa.Hn[n - 3];
                float* H = pa.Hn[n - 2];
                const long long xld = (long long)T * win, hld = (long long)T * W;
                Dep dx, dh = t > 0 ? Dep{counter(rb, n), (unsigned)(t * NUG)} : Dep{nullptr, 0};
                if (n == 2) dx = Dep{counter(rb, 1), (unsigned)((T - t) * NUG)};       // backward output of time t (the forward one came earlier)
                else dx = Dep{counter(rb, n - 1), (unsigned)((t + 1) * NUG)};
                const Dep dfw = n == 2 ? Dep{counter(rb, 0), (unsigned)((t + 1) * NUG)} : Dep{nullptr, 0};
                if (!cell(pa.ln[n - 2], win, xin + (long long)t * win, xld, H + (long long)(t - 1) * W, H + (long long)t * W, hld, rb, ug,
                          t == 0, creg[i], counter(rb, n), dx, dh, dfw)) return;
                if (t == T - 1) {
                    const int row = rb * 16 + er;
                    if (row < B) pa.cfin[(long long)(n - 1) * B * W + (long long)row * W + ug * 16 + ec] = creg[i];
                }
            }
        }
    }
#ifdef CASV_PERSIST_PROF
    if (prof_on && tid == 0) atomicAdd(const_cast<unsigned long long*>(pa.prof) + 17, wall_clock64() - tk1);
#endif
}

// Workgroups of `kernel` (256 threads, `lds` bytes of dynamic LDS) that one CU holds at once, as the runtime reports it for
// the code object that is actually loaded -- asked once per (device, LDS size).  0 = the query failed: no persistent launch.
template <class K>
static int blocks_per_cu(K kernel, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, lds});
    if (it != cache.end()) return it->second;
    int n = 0;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) n = 0;
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), 256, lds) != hipSuccess) n = 0;
    // The query counts registers and LDS; the workgroup's static LDS (gate exchange, flags: ~4.3 KB) is part of the kernel's
    // own figure.  Never plan for more than the two workgroups per CU the launch bounds promise.
    n = n > 2 ? 2 : (n < 0 ? 0 : n);
    cache[{dev, lds}] = n;
    return n;
}
int persist_encode_blocks_per_cu(size_t lds) { return lds > 150 * 1024 ? 0 : blocks_per_cu(persist_encode_kernel, lds); }
int persist_decode_blocks_per_cu(size_t lds) { return lds > 150 * 1024 ? 0 : blocks_per_cu(persist_decode_kernel, lds); }

size_t persist_enc_counter_bytes(int B, int D) {
    const size_t nrb = (B + 15) / 16;
    return (nrb * (D + 1) * 32 + 32) * sizeof(unsigned);
}

int launch_persist_encode(const PersistEncArgs& pa, int grid, hipStream_t stream) {
    const size_t lds = (size_t)16 * pa.lda * sizeof(float);
    if (lds > 150 * 1024) return -1;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&persist_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(persist_encode_kernel, dim3(grid), dim3(256), lds, stream, pa);
    return 0;
}

size_t persist_counter_bytes(int R, int D) {
    const size_t nrb = (R + 15) / 16;
    return (nrb * (D + 3) * 32 + 32) * sizeof(unsigned);
}

size_t persist_lds_bytes(const PersistArgs& pa) { return (size_t)16 * pa.lda * sizeof(float); }

int launch_persist_decode(const PersistArgs& pa, hipStream_t stream) {
    const size_t lds = persist_lds_bytes(pa);
    if (lds > 150 * 1024) return -1;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&persist_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
    }
    hipLaunchKernelGGL(persist_decode_kernel, dim3(pa.g_lstm + pa.g_att + pa.g_plain), dim3(256), lds, stream, pa);
    return 0;
}

}  // namespace casv
