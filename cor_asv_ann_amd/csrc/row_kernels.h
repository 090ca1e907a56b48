// Row-level device functions shared by the per-step kernels (decode_kernels.hip) and the persistent decoder
// (persist.hip): one 64-lane wave owns one decoder row, all reductions are shuffle butterflies with a fixed order, so a
// row's result depends neither on the batch it sits in nor on which kernel ran it.
#pragma once
#include "common.h"
#include <math.h>

namespace casv {

// write-through store (global_store ... sc1): payload handed to another workgroup inside a launch
__device__ __forceinline__ void store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and its read side: 16 B past this XCD's L2 (global_load ... sc1, as two 8-B loads).  Payload that is only ever read
// this way needs no cache invalidation on the consumer's side -- an invalidation would also evict the weights from the L2.
__device__ __forceinline__ float4 load_sc1(const float* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}

__device__ __forceinline__ float wave_sum(float v) { return wave_butterfly(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ double wave_sum_d(double v) { return wave_butterfly(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); }); }
// argmax butterfly of the softmax kernels: larger value wins, the lower index among equals
__device__ __forceinline__ void wave_argmax(float& best, int& bidx) {
    auto step = [&](const float ob, const int oi) { if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; } };
    step(lane_xor<32>(best), lane_xor<32>(bidx)); step(lane_xor<16>(best), lane_xor<16>(bidx)); step(lane_xor<8>(best), lane_xor<8>(bidx));
    step(lane_xor<4>(best), lane_xor<4>(bidx)); step(lane_xor<2>(best), lane_xor<2>(bidx)); step(lane_xor<1>(best), lane_xor<1>(bidx));
}

// ---------------------------------------------------------------------------------------------
// attention.py:526-575.  Only the <= 2*window+1 positions that survive the window mask are evaluated
// (the reference evaluates all T and multiplies by a 0/1 mask, attention.py:540-569: same values).
//   t' = sum_s a_prev[s]*s + 1 (float64 accumulate, rounded once to fp32 -- see oracle/model.py)
//   keep s with |t' - s| <= window ; e[s] = exp(tanh(wq + u[s]).v_a + b_v) ; a' = e / sum e
//   ctx = sum_s a'[s] * enc[s]
// ---------------------------------------------------------------------------------------------
constexpr int MAXWIN = 11;
constexpr int ATT_ROWS = 8;      // rows (waves) per workgroup: the N=8 hypotheses of one line share u/enc rows in L1

// One decoder row r at `step` (reads alignment slot `step` or the parent's, writes slot step + 1), in three parts so that
// a workgroup can stage the attended rows between them (attention_line_kernel); attention_row chains them on global memory.
struct AttWin { int s_lo, cnt; };

// (1) t' = sum_s a_prev[s] * s + 1 and the window |t' - s| <= window_width (attention.py:553-567).  acc = this lane's share of the
// sum: a_prev[s] * s over s = lane, lane + 64, ... in ascending order, in float64.
__device__ __forceinline__ AttWin att_window_of(const AttnArgs& a, const double acc) {
    const int T = a.T;
    const float tp = (float)(wave_sum_d(acc) + 1.0);
    const float win = (float)a.window;
    int s_lo = 0, s_hi = -1;                    // empty unless t' is a number
    if (tp == tp && fabsf(tp) < 1.0e9f) {
        int lo = (int)floorf(tp - win) - 1, hi = (int)floorf(tp + win) + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi > T - 1 ? T - 1 : hi;
        s_lo = T; s_hi = -1;
        for (int s = lo; s <= hi; ++s)
            if (fabsf(tp - (float)s) <= win) { s_lo = s < s_lo ? s : s_lo; s_hi = s; }
    }
    return AttWin{s_lo, s_hi - s_lo + 1};       // cnt <= MAXWIN
}
__device__ __forceinline__ AttWin att_window(const AttnArgs& a, const int r, const int step, const int lane) {
    const int T = a.T;
    const float* ap = a.a_base + (a.prev ? (long long)a.prev[r] : (long long)step * a.R + r) * T;
    double acc = 0.0;
    for (int s = lane; s < T; s += 64) acc += (double)ap[s] * (double)s;
    return att_window_of(a, acc);
}
// ... of the NEXT step, from the weights e[] that att_weights has just normalised over the window w: the alignment row it wrote
// (e[i] at s_lo + i, zero elsewhere, NaN everywhere for an empty window) enters the sum from registers instead of from memory --
// the same products in the same order (a wave that keeps its row from step to step: the persistent decoder).
__device__ __forceinline__ AttWin att_window_next(const AttnArgs& a, const AttWin w, const float (&e)[MAXWIN], const int lane) {
    const int T = a.T;
    double acc = 0.0;
    for (int s = lane; s < T; s += 64) {
        float v = 0.0f;
        if (w.cnt <= 0) v = __builtin_nanf("");
#pragma unroll
        for (int i = 0; i < MAXWIN; ++i)
            if (i < w.cnt && s == w.s_lo + i) v = e[i];
        acc += (double)v * (double)s;
    }
    return att_window_of(a, acc);
}
// A window's rows held in registers (this lane's float4 of each), in the place of the row pointers att_weights / att_context index:
// urow(s_lo + i)[j] -> the register of window position i.
struct RegRows {
    const float4* regs; int i;
    __device__ __forceinline__ float4 operator[](int) const { return regs[i]; }
};

// (2) energies exp(tanh(wq + u[s]) . v_a + b_v) over the window, normalised; writes the alignment row and the per-row
// by-products.  urow(s) -> the row u[line][s] as float4s (global memory, or the workgroup's staged copy).
// HANDOFF: the query row was written by another workgroup of the same launch (read past the L2).
// WB = window rows requested together (MAXWIN: all of them, fewest round trips; less: fewer registers, more waves per SIMD).
// The sums per window position are independent of one another: the same values either way.
// In two halves, so that several waves can share the positions of one row (persist.hip): att_energy_sums leaves in e[i], for the
// window positions i that sel(i) picks, tanh(wq + u[s_lo + i]) . v_a summed over the row (every lane holds the sum; the other
// positions keep what they held); att_normalise turns the eleven sums into the row's weights and -- `write` -- stores the alignment
// row and the by-products.  att_weights = the two in a row over all positions.
template <bool HANDOFF, int WB, class URow, class Sel>
__device__ __forceinline__ void att_energy_sums(const AttnArgs& a, const int r, const int lane, const AttWin w, URow urow, Sel sel,
                                                float (&e)[MAXWIN]) {
    const int W = a.W;
    const int s_lo = w.s_lo;
    const float4* wq4 = reinterpret_cast<const float4*>(a.wq + (long long)(a.wq_rows ? a.wq_rows[r] : r) * W);
    const float4* va4 = reinterpret_cast<const float4*>(a.va);
    const int W4 = W >> 2;
    // The kernel is latency-bound (one decoder row per wave slot), so all window rows are requested
    // together: loads are unconditional on clamped row indices, positions past the window get weight 0.
#pragma unroll
    for (int i = 0; i < MAXWIN; ++i) if (sel(i)) e[i] = 0.0f;
#pragma unroll
    for (int i0 = 0; i0 < MAXWIN; i0 += WB) {
        for (int j = lane; j < W4; j += 64) {
            const float4 q = HANDOFF ? load_sc1(reinterpret_cast<const float*>(wq4 + j)) : wq4[j], v = va4[j];
            float4 uu[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) if (i0 + i < MAXWIN && sel(i0 + i)) uu[i] = urow(s_lo + i0 + i)[j];
#pragma unroll
            for (int i = 0; i < WB; ++i) if (i0 + i < MAXWIN && sel(i0 + i)) {
                e[i0 + i] += fast_tanh(q.x + uu[i].x) * v.x;
                e[i0 + i] += fast_tanh(q.y + uu[i].y) * v.y;
                e[i0 + i] += fast_tanh(q.z + uu[i].z) * v.z;
                e[i0 + i] += fast_tanh(q.w + uu[i].w) * v.w;
            }
        }
        if (WB < MAXWIN) __builtin_amdgcn_sched_barrier(0);            // keep the batches apart: that is where the registers go
    }
#pragma unroll
    for (int i = 0; i < MAXWIN; ++i) if (sel(i)) e[i] = wave_sum(e[i]);
}
__device__ __forceinline__ void att_normalise(const AttnArgs& a, const int r, const int step, const int lane, const AttWin w,
                                              float (&e)[MAXWIN], const bool write = true) {
    const int T = a.T;
    const int s_lo = w.s_lo, cnt = w.cnt;
    float* aout = const_cast<float*>(a.a_base) + ((long long)(step + 1) * a.R + r) * T;
    const float bv = a.bv[0];
    float denom = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXWIN; ++i) {
        const float sc = e[i] + bv;
        e[i] = i < cnt ? expf(sc) : 0.0f;
        denom += e[i];
    }
    const float nanv = __builtin_nanf("");
    float amax = 0.0f;
    double pos = 0.0;
#pragma unroll
    for (int i = 0; i < MAXWIN; ++i) {
        if (i < cnt) {
            e[i] = e[i] / denom;
            amax = fmaxf(amax, e[i]);
            pos += (double)e[i] * (double)(s_lo + i);
        }
    }
    if (!write) return;
    for (int s = lane; s < T; s += 64) {
        float v = 0.0f;
        if (cnt <= 0) v = nanv;                 // 0/0 everywhere, as in the reference
#pragma unroll
        for (int i = 0; i < MAXWIN; ++i)
            if (i < cnt && s == s_lo + i) v = e[i];
        aout[s] = v;
    }
    if (lane == 0) {
        if (a.apos) a.apos[r] = cnt <= 0 ? (double)nanv : pos;
        if (a.amax1) a.amax1[r] = (amax == 1.0f) ? 1 : 0;
        if (a.win_out) a.win_out[r] = cnt > 0 ? (s_lo | (cnt << 16)) : 0;
        if (a.win_store) a.win_store[(long long)(step + 1) * a.R + r] = cnt > 0 ? (s_lo | (cnt << 16)) : -1;
    }
}
template <bool HANDOFF, int WB, class URow>
__device__ __forceinline__ void att_weights(const AttnArgs& a, const int r, const int step, const int lane, const AttWin w,
                                            URow urow, float (&e)[MAXWIN]) {
    att_energy_sums<HANDOFF, WB>(a, r, lane, w, urow, [](int) { return true; }, e);
    att_normalise(a, r, step, lane, w, e);
}

// (3) context = sum_s a'[s] * enc[s].  erow(s) -> the row enc[line][s] as float4s.  HANDOFF: the context vector goes out
// with write-through stores (another workgroup of the same launch consumes it).
template <bool HANDOFF, int WB, class ERow>
__device__ __forceinline__ void att_context(const AttnArgs& a, const int r, const int lane, const AttWin w, ERow erow,
                                            const float (&e)[MAXWIN]) {
    const int C = a.C, s_lo = w.s_lo, cnt = w.cnt;
    const float nanv = __builtin_nanf("");
    float4* ctx4 = reinterpret_cast<float4*>(a.ctx + (long long)r * (a.ctx_ld ? a.ctx_ld : C));
    const float4* mk4 = a.ctx_mask ? reinterpret_cast<const float4*>(a.ctx_mask + (long long)r * a.ctx_mask_ld) : nullptr;
    const int C4 = C >> 2;
    for (int c = lane; c < C4; c += 64) {
        float4 v = cnt <= 0 ? make_float4(nanv, nanv, nanv, nanv) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i0 = 0; i0 < MAXWIN; i0 += WB) {
            float4 x[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) if (i0 + i < MAXWIN) x[i] = erow(s_lo + i0 + i)[c];
#pragma unroll
            for (int i = 0; i < WB; ++i)
                if (i0 + i < MAXWIN && i0 + i < cnt) { v.x += e[i0 + i] * x[i].x; v.y += e[i0 + i] * x[i].y; v.z += e[i0 + i] * x[i].z; v.w += e[i0 + i] * x[i].w; }
            if (WB < MAXWIN) __builtin_amdgcn_sched_barrier(0);
        }
        if (mk4) { const float4 k = mk4[c]; v.x *= k.x; v.y *= k.y; v.z *= k.z; v.w *= k.w; }
        if (HANDOFF) {
            float* dst = reinterpret_cast<float*>(ctx4 + c);
            store_sc1(dst, v.x); store_sc1(dst + 1, v.y); store_sc1(dst + 2, v.z); store_sc1(dst + 3, v.w);
        } else {
            ctx4[c] = v;
        }
    }
}

template <bool HANDOFF, int WB = MAXWIN>
__device__ __forceinline__ void attention_row(const AttnArgs& a, const int r, const int step, const int lane) {
    const int ln = a.line ? a.line[r] : r / a.rows_per_line;
    const int T = a.T;
    const AttWin w = att_window(a, r, step, lane);
    const float* ub = a.u + (long long)ln * a.u_line;
    const float* eb = a.enc + (long long)ln * a.enc_line;
    float e[MAXWIN];
    att_weights<HANDOFF, WB>(a, r, step, lane, w, [&](int s) { s = s < T ? s : T - 1; return reinterpret_cast<const float4*>(ub + (long long)s * a.u_time); }, e);
    att_context<HANDOFF, WB>(a, r, lane, w, [&](int s) { s = s < T ? s : T - 1; return reinterpret_cast<const float4*>(eb + (long long)s * a.enc_time); }, e);
}

}  // namespace casv
