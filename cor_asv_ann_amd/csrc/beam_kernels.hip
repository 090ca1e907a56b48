// Device-resident best-first beam search: decode_sequence_beam + Node of seq2seq.py:1356-1608, for
// all lines of a batch at once.  One 256-thread workgroup owns one line: its priority queue
// (next_beam), its list of finished hypotheses (final_beam) and its trie of nodes live in HBM and
// never visit the host.  Per search iteration the workgroup
//   A  turns the step's output rows into child nodes (rejection candidate seq2seq.py:1457-1470,
//      beam threshold :1472-1480, successive-reset feedback :1515-1520),
//   B  merges them into the queue in `insort_left` order and caps it (:1529-1532),
//   C  pops the next <= N hypotheses (:1399-1420), files finished ones, evaluates the stop
//      conditions and builds the next step's input rows and parent indices.
// Order of nodes = (pro_cost descending, creation order ascending): `insort_left` + `pop()` take,
// among equal pro_cost, the node inserted first.
#include "common.h"
#include <math.h>
#include <stdio.h>

namespace casv {


constexpr int BEAM_SORT_CAP = 4096;    // new keys sorted in LDS at once; more are sorted in runs of this size and merged by rank

__device__ __forceinline__ bool before(double ka, int ia, double kb, int ib) {
    if (ia == 0x7fffffff) return false;
    if (ib == 0x7fffffff) return true;
    return ka > kb || (ka == kb && ia < ib);
}
// candidate order inside one row: score descending, ties towards the higher index
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia > ib);
}

__global__ void beam_init_kernel(const BeamState s, const BeamParams p) {
    const int line = blockIdx.x;
    const int N = p.N, Vp = (s.V + 31) & ~31;
    const long long nb = (long long)line * s.node_cap;
    if (threadIdx.x == 0) {
        s.n_parent[nb] = -1; s.n_chr[nb] = -1; s.n_prob[nb] = 0.f; s.n_cum[nb] = 0.0; s.n_len[nb] = 1;
        s.n_exp[nb] = line * N; s.n_k[nb] = 0; s.n_rejpos[nb] = -1; s.n_pos[nb] = 0.0; s.n_is1[nb] = 0;
        s.n_count[line] = 1;
        s.q_n[line] = 0; s.q_n[s.B + line] = 0;      // [0..B) count, [B..2B) head offset
        s.f_n[line] = 0; s.f_total[line] = 0;
        s.nact[line] = 1; s.line_done[line] = 0; s.line_steps[line] = 0;
        s.beam_node[line * N] = 0;
        if (line == 0) { s.active_lines[0] = s.B; s.active_lines[1] = 0; s.active_lines[2] = 0; s.active_lines[3] = 0; }
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) s.prev[line * N + i] = line * N;
    for (int i = threadIdx.x; i < N * Vp; i += blockDim.x) s.p_in[(long long)line * N * Vp + i] = 0.f;
}
void launch_beam_init(const BeamState& s, const BeamParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(beam_init_kernel, dim3(s.B), dim3(256), 0, stream, s, p);
}

// Diagnostic build (-DCASV_BEAM_PROF): the workgroup of line 0 adds the wall-clock ticks (10 ns) of each phase of its step
// to g_beam_prof; beam_prof_dump() prints and clears them.
#ifdef CASV_BEAM_PROF
__device__ unsigned long long g_beam_prof[8];
__device__ unsigned g_beam_wg[3 * 4096];          // per line: start and end tick of its workgroup in the last launch, new keys
#define BPROF(slot) do { __syncthreads(); if (line == 0 && tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(g_beam_prof + (slot), now - bt); bt = now; } } while (0)
void beam_prof_dump(int steps) {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_beam_prof), sizeof h) != hipSuccess) return;
    fprintf(stderr, "beam prof us/step: A1 %.2f  offsets %.2f  A2 %.2f  sort %.2f  merge %.2f  pop %.2f  inputs %.2f\n",
            h[0] * 0.01 / steps, h[1] * 0.01 / steps, h[2] * 0.01 / steps, h[3] * 0.01 / steps, h[4] * 0.01 / steps, h[5] * 0.01 / steps, h[6] * 0.01 / steps);
    for (auto& v : h) v = 0;
    {
        static unsigned w[3 * 4096];
        if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_beam_wg), sizeof w) == hipSuccess) {
            unsigned t0 = ~0u, t1 = 0; int n = 0;
            for (int i = 0; i < 4096; ++i) if (w[3 * i + 1]) { t0 = w[3 * i] < t0 ? w[3 * i] : t0; t1 = w[3 * i + 1] > t1 ? w[3 * i + 1] : t1; ++n; }
            if (n) {
                double sum = 0; unsigned dmax = 0, smax = 0; int knew = 0;
                for (int i = 0; i < 4096; ++i) if (w[3 * i + 1]) { const unsigned d = w[3 * i + 1] - w[3 * i]; sum += d; if (d > dmax) { dmax = d; knew = (int)w[3 * i + 2]; } if (w[3 * i] - t0 > smax) smax = w[3 * i] - t0; }
                fprintf(stderr, "beam prof last launch: %d workgroups, span %.2f us, mean duration %.2f, max %.2f (new keys %d), latest start +%.2f us\n",
                        n, (t1 - t0) * 0.01, sum / n * 0.01, dmax * 0.01, knew, smax * 0.01);
            }
        }
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_beam_prof), h, sizeof h);
}
#else
#define BPROF(slot)
#endif

// Phase A1 for expansion row i of `line` (one wave): the row's softmax (when the step hands over logits), the rejection
// candidate (seq2seq.py:1457-1470) written over its score, the beam width from the relative threshold (:1472-1480), the
// ranks of index 0 and of the rejection index => number of children.  The row's final scores stay in `vals`
// (lane + 64 k; -inf beyond V) and in the score store.
template <int VPL>
__device__ __forceinline__ RowRec expand_row_scores(const BeamState& s, const BeamParams& p, const int line, const int i, const int step,
                                                    const int lane, float (&vals)[VPL]) {
    const int N = p.N, V = s.V, Vp = (V + 31) & ~31, T = s.T, R = s.R;
    const long long nbase = (long long)line * s.node_cap;
    const int r = line * N + i;
    const long long exp = (long long)(step + 1) * R + r;
    float* sc = const_cast<float*>(s.p_base) + exp * Vp;
    const int node = s.beam_node[r];
    const int plen = s.n_len[nbase + node];
    const double ppos = s.n_pos[nbase + node];
    const int pis1 = s.n_is1[nbase + node];
    const double pos = s.apos[r];
    double mis = 0.0;
    int srcpos = 0;
    if (plen > 1) {
        mis = fabs(pos - ppos - 1.0);
        if (pis1) srcpos = (int)ppos + 1;
        else srcpos = (pos == pos && fabs(pos) < 1e9) ? (int)rint(pos) : -1;
    }
    int rej = -1;
    if (p.rejection != 0.0 && (mis < 0.1 || pis1) && srcpos >= 0 && srcpos < T)
        rej = s.src_rej[line * T + srcpos];           // -1: input row all zero (np.any false)
    bool anynan = false;
    if (s.logits) {
        // the row's softmax (decode_kernels.hip softmax_kernel: per-lane partial results over v = lane, lane + 64, ... in
        // ascending order, the same butterflies, expf(x - m) / sum), written to the score store and kept in registers
        const float* x = s.logits + (long long)r * Vp;
        float m = -INFINITY, nn = 0.0f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int v = lane + 64 * k;
            vals[k] = v < V ? x[v] : 0.0f;
            if (v < V) { m = fmaxf(m, vals[k]); nn += (vals[k] != vals[k]) ? 1.0f : 0.0f; }
        }
        m = wave_butterfly(m, [](float a, float b) { return fmaxf(a, b); });
        if (wave_butterfly(nn, [](float a, float b) { return a + b; }) > 0.0f) m = __builtin_nanf("");
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) if (lane + 64 * k < V) sum += expf(vals[k] - m);
        sum = wave_butterfly(sum, [](float a, float b) { return a + b; });
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int v = lane + 64 * k;
            const float pv = v < V ? expf(vals[k] - m) / sum : 0.0f;
            if (v < Vp) sc[v] = pv;                      // columns V..Vp-1: exact zeros (K padding of the next step's GEMM)
            vals[k] = v < V ? pv : -INFINITY;
            if (v < V && pv != pv) anynan = true;
        }
    } else {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int v = lane + 64 * k;
            vals[k] = v < V ? sc[v] : -INFINITY;
            if (v < V && vals[k] != vals[k]) anynan = true;
        }
    }
    anynan = __any(anynan);
    if (rej >= 0) {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int v = lane + 64 * k;
            if (v == rej && (double)vals[k] < p.rejection) { vals[k] = (float)p.rejection; sc[v] = vals[k]; }
        }
    }
    float hi = -INFINITY;
#pragma unroll
    for (int k = 0; k < VPL; ++k) hi = fmaxf(hi, vals[k]);
    hi = wave_butterfly(hi, [](float a, float b) { return fmaxf(a, b); });
    const double thr = (double)hi * p.threshold_in;
    float v0 = __shfl(vals[0], 0, 64);                  // score of index 0
    float vr = 0.f;
    if (rej >= 0) {
#pragma unroll
        for (int k = 0; k < VPL; ++k) if (lane + 64 * k == rej) vr = vals[k];
        vr = wave_butterfly(vr, [](float a, float b) { return fmaxf(a, b); });   // scores are >= 0
    }
    int cnt = 0, rank0 = 0, rankr = 0;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int v = lane + 64 * k;
        if (v < V) {
            cnt += ((double)vals[k] >= thr) ? 1 : 0;
            rank0 += better(vals[k], v, v0, 0) ? 1 : 0;
            if (rej >= 0) rankr += better(vals[k], v, vr, rej) ? 1 : 0;
        }
    }
    {
        auto add = [](int a, int b) { return a + b; };
        cnt = wave_butterfly(cnt, add); rank0 = wave_butterfly(rank0, add); rankr = wave_butterfly(rankr, add);
    }
    rank0 += 1; rankr += 1;
    const int beampos = cnt < p.width_in ? cnt : p.width_in;
    int count = beampos - (rank0 <= beampos ? 1 : 0);          // '' never becomes a node (s2s:1505)
    const int rejlate = (rej > 0 && rankr > beampos) ? 1 : 0;  // `if rej_idx:` is false for 0 (s2s:1498)
    count += rejlate;
    if (anynan) count = 0;
    return RowRec{count, beampos, rej, srcpos, anynan ? 1 : 0, rejlate};
}

// The children of one expansion row in creation order (seq2seq.py:1482-1529), from its final scores `vals`: the beampos best
// (score descending, ties towards the higher index), then the rejection candidate if it lies beyond them; index 0 ('') is
// skipped.  emit(k, index, score, is_rejection) is called by every lane with the same arguments.  Returns the children made.
template <int VPL, class Emit>
__device__ __forceinline__ int select_children(const float (&vals)[VPL], const int V, const int beampos, const int rejlate, int rej,
                                               const int lane, Emit emit) {
    unsigned long long taken = 0;
    int created = 0;
    const int total = beampos + rejlate;
    for (int ps = 1; ps <= total; ++ps) {
        float bv = -INFINITY; int bi = -1;
        if (ps <= beampos) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const int v = lane + 64 * k;
                if (v < V && !((taken >> k) & 1ull) && (bi < 0 || better(vals[k], v, bv, bi))) { bv = vals[k]; bi = v; }
            }
            {
                auto step = [&](const float ov, const int oi) { if (oi >= 0 && (bi < 0 || better(ov, oi, bv, bi))) { bv = ov; bi = oi; } };
                step(lane_xor<32>(bv), lane_xor<32>(bi)); step(lane_xor<16>(bv), lane_xor<16>(bi)); step(lane_xor<8>(bv), lane_xor<8>(bi));
                step(lane_xor<4>(bv), lane_xor<4>(bi)); step(lane_xor<2>(bv), lane_xor<2>(bi)); step(lane_xor<1>(bv), lane_xor<1>(bi));
            }
#pragma unroll
            for (int k = 0; k < VPL; ++k) if (lane + 64 * k == bi) taken |= 1ull << k;
        } else {                                   // the rejection candidate beyond the beam width
            bi = rej; bv = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) if (lane + 64 * k == rej) bv = vals[k];
            bv = wave_butterfly(bv, [](float a, float b) { return fmaxf(a, b); });
        }
        bool isrej = false;
        if (rej >= 0 && bi == rej) { isrej = true; rej = -1; }
        if (bi == 0) continue;
        emit(created, bi, bv, isrej);
        ++created;
    }
    return created;
}

// Wide beams (N >= 64 hypotheses per line: the reference's default batch_size = 256) on few lines: phase A -- everything that is
// per ROW (softmax, rejection, child selection) -- runs as a grid of its own over all rows of all lines, one wave per row
// (beam_expand_kernel), and leaves per row a record and the list of its children; the per-line kernel (SPLIT) then only numbers
// the children, writes their node records (one thread per child), sorts, merges and pops.  With one workgroup per line doing
// all of it, a page of 40 lines x 256 hypotheses kept 40 CUs busy for 0.45 ms per step, most of it walking rows.
template <int VPL>
__global__ __launch_bounds__(512) void beam_expand_kernel(const BeamState s, const BeamParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 8 + wave;
    if (r >= s.R) return;
    const int N = p.N, line = r / N, i = r - line * N;
    const int step = s.step_ptr ? *s.step_ptr : s.step_imm;
    if (s.line_done[line] || i >= s.nact[line] || step + 1 >= s.S) return;
    const int CMAX = (p.width_in < s.V ? p.width_in : s.V) + 1;
    float vals[VPL];
    RowRec rc = expand_row_scores<VPL>(s, p, line, i, step, lane, vals);
    int rej_k = -1;
    if (rc.count > 0) {
        short* ci = s.cand_idx + (long long)r * CMAX;
        float* cv = s.cand_val + (long long)r * CMAX;
        select_children<VPL>(vals, s.V, rc.beampos, rc.rejlate, rc.rej, lane, [&](const int k, const int bi, const float bv, const bool isrej) {
            if (lane == 0) { ci[k] = (short)bi; cv[k] = bv; }
            if (isrej) rej_k = k;
        });
    }
    rc.rej = rej_k;                        // from here on: WHICH child is the rejection candidate (-1: none)
    if (lane == 0) s.rowrec[r] = rc;
}

// VPL = vocabulary entries per lane (V <= 64 * VPL); NWV = waves per workgroup (one wave expands one hypothesis row);
// SPLIT: phase A has run as beam_expand_kernel
template <int VPL, int NWV, bool SPLIT>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(VPL <= 4 && NWV <= 8 ? 8 : (NWV > 8 ? 4 : 1), 8))) void beam_step_kernel(const BeamState s, const BeamParams p) {
    constexpr int NT = 64 * NWV;
    // dynamic LDS: [sort_cap new keys (f64)] [q_stage old keys (f64)] [pop_cap head keys (f64)] [sort_cap new ids] [q_stage old ids]
    // [7 x (N+1) row records] [pop_cap head ids] [pop_cap head characters]
    extern __shared__ __attribute__((aligned(16))) unsigned char beam_smem[];
    __shared__ int sh_nnew, sh_nb, sh_done;

    const int line = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step = s.step_ptr ? *s.step_ptr : s.step_imm;
    if (s.line_done[line]) return;
    const int N = p.N, V = s.V, Vp = (V + 31) & ~31, T = s.T, R = s.R;
    const int CMAX = (p.width_in < V ? p.width_in : V) + 1;      // children per expansion: <= min(width_in, V) + the late rejection
    const int cap = p.sort_cap;                    // new keys the LDS holds (a power of two)
    const int q_stage = p.q_stage;                 // entries of the old queue staged in LDS (0: search in HBM)
    // The head of the merged queue -- the entries the pop loop of phase C walks, N plus as many finished hypotheses as it meets on
    // the way -- is kept in LDS while the merge writes it: one thread walks it serially, and at N = 256 (the reference's default
    // batch_size) a walk through global memory was two dependent round trips per entry.
    const int pop_cap = p.pop_cap;
    double* s_key = reinterpret_cast<double*>(beam_smem);
    double* o_key = s_key + cap;
    double* pop_key = o_key + q_stage;
    int* s_id = reinterpret_cast<int*>(pop_key + pop_cap);
    int* o_id = s_id + cap;
    int* r_count = o_id + q_stage;
    int* r_off = r_count + (N + 1);
    int* r_beampos = r_off + (N + 1);
    int* r_rej = r_beampos + (N + 1);
    int* r_srcpos = r_rej + (N + 1);
    int* r_nan = r_srcpos + (N + 1);
    int* r_rejlate = r_nan + (N + 1);
    int* pop_id = r_rejlate + (N + 1);
    int* pop_chr = pop_id + pop_cap;
    const int nact = s.nact[line];
    const long long nbase = (long long)line * s.node_cap;
    if (tid == 0) s.line_steps[line] = step + 1;
    if (step + 1 >= s.S) return;       // children of the last iteration are never popped (s2s:1398)

#ifdef CASV_BEAM_PROF
    unsigned long long bt = wall_clock64();
    if (tid == 0 && line < 4096) { g_beam_wg[3 * line] = (unsigned)bt; g_beam_wg[3 * line + 1] = 0; }
#endif
    // ---------------- A1: per row, rejection overwrite + child count ----------------
    if (!SPLIT) {
        for (int i = wave; i < nact; i += NWV) {
            float vals[VPL];
            const RowRec rc = expand_row_scores<VPL>(s, p, line, i, step, lane, vals);
            if (lane == 0) {
                r_count[i] = rc.count; r_beampos[i] = rc.beampos; r_rej[i] = rc.rej; r_srcpos[i] = rc.srcpos;
                r_nan[i] = rc.nan; r_rejlate[i] = rc.rejlate;
            }
        }
    } else {
        for (int i = tid; i < nact; i += NT) {
            const RowRec rc = s.rowrec[line * N + i];
            r_count[i] = rc.count; r_rej[i] = rc.rej; r_srcpos[i] = rc.srcpos;      // r_rej: the child that is the rejection
        }
    }
    __syncthreads();
    BPROF(0);
    if (tid == 0) {
        int o = 0;
        for (int i = 0; i < nact; ++i) { r_off[i] = o; o += r_count[i]; }
        r_off[nact] = o;
        sh_nnew = o;
        if (o > s.active_lines[1]) atomicMax(s.active_lines + 1, o);      // statistic: most new keys of any line and step
    }
    __syncthreads();
    const int nnew = sh_nnew;
    const int id0 = s.n_count[line];
    // More new keys than the LDS holds (wide beams: N * (beam_width_in + 1) can reach 256 * 51 with the reference's
    // settings): they go to a scratch array in HBM, are sorted in runs of `cap` and merged by rank (phase B).
    const bool big = nnew > cap;
    const long long gbase = (long long)line * s.g_cap;
    double* g0k = s.g_key + 2 * gbase; int* g0i = s.g_id + 2 * gbase;          // unsorted, then sorted runs
    double* g1k = g0k + s.g_cap; int* g1i = g0i + s.g_cap;                     // the merged order

    BPROF(1);
    // ---------------- A2: iterative selection, node records, keys ----------------
    auto new_node = [&](const int i, const int k, const int bi, const float bv, const bool isrej, const int node, const int plen,
                        const double pcum, const double pos, const int is1, const long long exp) {
        const int slot = r_off[i] + k;
        const int id = id0 + slot;
        const float cost = -logf(bv);
        const double cum = pcum + (double)cost;
        const int len = plen + 1;
        const long long g = nbase + id;
        const int srcpos = r_srcpos[i];
        s.n_parent[g] = node; s.n_chr[g] = bi; s.n_prob[g] = bv; s.n_cum[g] = cum; s.n_len[g] = len;
        s.n_exp[g] = (int)exp; s.n_k[g] = k;
        s.n_rejpos[g] = isrej ? srcpos : -1;
        s.n_pos[g] = isrej ? (double)srcpos : pos;
        s.n_is1[g] = isrej ? 1 : is1;
        s.created[exp * CMAX + k] = (short)bi;
        const double key = -(cum + p.cost0 * fabs((double)(len - T)));
        if (big) { g0k[slot] = key; g0i[slot] = id; }
        else { s_key[slot] = key; s_id[slot] = id; }
    };
    if (!SPLIT) {
        for (int i = wave; i < nact; i += NWV) {
            if (r_count[i] == 0) continue;
            const int r = line * N + i;
            const long long exp = (long long)(step + 1) * R + r;
            const float* sc = s.p_base + exp * Vp;
            const int node = s.beam_node[r];
            const int plen = s.n_len[nbase + node];
            const double pcum = s.n_cum[nbase + node];
            const double pos = s.apos[r];
            const int is1 = s.amax1[r];
            float vals[VPL];
#pragma unroll
            for (int k = 0; k < VPL; ++k) { const int v = lane + 64 * k; vals[k] = v < V ? sc[v] : -INFINITY; }
            select_children<VPL>(vals, V, r_beampos[i], r_rejlate[i], r_rej[i], lane, [&](const int k, const int bi, const float bv, const bool isrej) {
                if (lane == 0) new_node(i, k, bi, bv, isrej, node, plen, pcum, pos, is1, exp);
            });
        }
    } else {
        // the children were chosen by beam_expand_kernel: one thread per child writes its node record
        for (int c = tid; c < nnew; c += NT) {
            int lo = 0, hi = nact;                          // row i with r_off[i] <= c < r_off[i + 1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (r_off[mid] <= c) lo = mid; else hi = mid; }
            const int i = lo, k = c - r_off[i];
            const int r = line * N + i;
            const long long exp = (long long)(step + 1) * R + r;
            const int node = s.beam_node[r];
            new_node(i, k, (int)s.cand_idx[(long long)r * CMAX + k], s.cand_val[(long long)r * CMAX + k], k == r_rej[i], node,
                     s.n_len[nbase + node], s.n_cum[nbase + node], s.apos[r], s.amax1[r], exp);
        }
    }
    __syncthreads();
    if (tid == 0) s.n_count[line] = id0 + nnew;

    BPROF(2);
    // ---------------- B: sort the new nodes, merge with the queue, cap ----------------
    auto bitonic = [&](int npow) {          // s_key/s_id[0..npow) best first
        for (int k = 2; k <= npow; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npow; i += NT) {
                    const int l = i ^ j;
                    if (l > i) {
                        const bool up = (i & k) == 0;
                        const double ka = s_key[i], kb = s_key[l];
                        const int ia = s_id[i], ib = s_id[l];
                        const bool sw = up ? before(kb, ib, ka, ia) : before(ka, ia, kb, ib);
                        if (sw) { s_key[i] = kb; s_id[i] = ib; s_key[l] = ka; s_id[l] = ia; }
                    }
                }
                __syncthreads();
            }
        }
    };
    if (!big) {
        int npow = 1;
        while (npow < nnew) npow <<= 1;
        for (int i = nnew + tid; i < npow; i += NT) { s_key[i] = 0.0; s_id[i] = 0x7fffffff; }
        __syncthreads();
        bitonic(npow);
    } else {
        const int nruns = (nnew + cap - 1) / cap;
        for (int run = 0; run < nruns; ++run) {
            const int base = run * cap, nr = nnew - base < cap ? nnew - base : cap;
            int npow = 1;
            while (npow < nr) npow <<= 1;
            for (int i = tid; i < npow; i += NT) {
                s_key[i] = i < nr ? g0k[base + i] : 0.0; s_id[i] = i < nr ? g0i[base + i] : 0x7fffffff;
            }
            __syncthreads();
            bitonic(npow);
            for (int i = tid; i < nr; i += NT) { g0k[base + i] = s_key[i]; g0i[base + i] = s_id[i]; }
            __syncthreads();
        }
        // position in the merged order = position in the own run + rank in every other run (ids are unique: no ties)
        for (int j = tid; j < nnew; j += NT) {
            const double k = g0k[j]; const int id = g0i[j];
            const int own = j / cap;
            int pos = j - own * cap;
            for (int run = 0; run < nruns; ++run) {
                if (run == own) continue;
                const int base = run * cap;
                int lo = 0, hi = nnew - base < cap ? nnew - base : cap;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (before(g0k[base + mid], g0i[base + mid], k, id)) lo = mid + 1; else hi = mid; }
                pos += lo;
            }
            g1k[pos] = k; g1i[pos] = id;
        }
        __syncthreads();
    }
    BPROF(3);
    const int par = step & 1;
    const long long qstride = (long long)s.B * s.q_cap;
    const int qn_old = s.q_n[line], qhead = s.q_n[s.B + line];
    const double* okey = s.q_key + par * qstride + (long long)line * s.q_cap + qhead;
    const int* oid = s.q_id + par * qstride + (long long)line * s.q_cap + qhead;
    double* nkey = s.q_key + (par ^ 1) * qstride + (long long)line * s.q_cap;
    int* nid = s.q_id + (par ^ 1) * qstride + (long long)line * s.q_cap;
    const int qcap = s.q_cap;                     // max_batches * batch_size (s2s:1531)
    const bool staged = qn_old <= q_stage;
    if (staged) {                                 // one coalesced pass instead of log2(n) dependent HBM probes
        for (int i = tid; i < qn_old; i += NT) { o_key[i] = okey[i]; o_id[i] = oid[i]; }
        __syncthreads();
    }
    // merge of the sorted new keys (in LDS, or in HBM for the wide case) with the old queue, by rank
    auto merge = [&](const double* nk, const int* ni, const double* ok, const int* oi) {
        for (int i = tid; i < qn_old; i += NT) {      // old element i moves behind the new ones before it
            const double k = ok[i]; const int id = oi[i];
            int lo = 0, hi = nnew;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (before(nk[mid], ni[mid], k, id)) lo = mid + 1; else hi = mid; }
            const int pos = i + lo;
            if (pos < qcap) { nkey[pos] = k; nid[pos] = id; }
            if (pos < pop_cap) { pop_key[pos] = k; pop_id[pos] = id; }
        }
        for (int j = tid; j < nnew; j += NT) {
            const double k = nk[j]; const int id = ni[j];
            int lo = 0, hi = qn_old;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (before(ok[mid], oi[mid], k, id)) lo = mid + 1; else hi = mid; }
            const int pos = j + lo;
            if (pos < qcap) { nkey[pos] = k; nid[pos] = id; }
            if (pos < pop_cap) { pop_key[pos] = k; pop_id[pos] = id; }
        }
    };
    if (!big) { if (staged) merge(s_key, s_id, o_key, o_id); else merge(s_key, s_id, okey, oid); }
    else { if (staged) merge(g1k, g1i, o_key, o_id); else merge(g1k, g1i, okey, oid); }
    __threadfence_block();
    __syncthreads();
    const int qn = (qn_old + nnew) < qcap ? (qn_old + nnew) : qcap;

    BPROF(4);
    // ---------------- C: pop the next beam ----------------
    const int pre = qn < pop_cap ? qn : pop_cap;
    for (int i = tid; i < pre; i += NT) pop_chr[i] = s.n_chr[nbase + pop_id[i]];
    __syncthreads();
    // Wide beams (SPLIT: N >= 64 hypotheses per line): the walk is done by all threads -- every head entry is flagged finished /
    // unfinished, exclusive prefix counts (wave ballots + one pass over the waves' totals) give every unfinished entry its slot in the
    // next beam and the walk's end (the entry behind the N-th unfinished one); only the FINISHED entries in front of that end -- a
    // handful at most -- are filed one by one, in order, as the serial walk files them.  Round 4: one thread walking N + 64 entries
    // took 40-45 us of the page call's 207-us step.  (Not taken when the head held in LDS does not reach the walk's end: the serial
    // walk below goes on into the queue in HBM.)
    bool walked = false;
    if (SPLIT && pop_cap <= NT) {
        __shared__ int w_nf[NWV], w_fin[NWV];
        __shared__ int sh_hend, sh_nfin;
        __shared__ double sh_b0;
        int* finl = r_off;                                   // (the row records of phase A are dead: r_off and r_beampos, 2 N + 2 >= N + 64 ints)
        const int h = tid;
        const bool in = h < pre;
        const int chr = in ? pop_chr[h] : 0;
        const bool fin = in && chr == p.eos, unf = in && chr != p.eos;
        const unsigned long long bu = __ballot(unf), bf = __ballot(fin);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (lane == 0) { w_nf[wave] = __popcll(bu); w_fin[wave] = __popcll(bf); }
        if (tid == 0) { sh_hend = -1; sh_nfin = 0; }
        __syncthreads();
        int nf_x = __popcll(bu & below), fin_x = __popcll(bf & below), nf_total = 0;
        for (int w = 0; w < NWV; ++w) { if (w < wave) { nf_x += w_nf[w]; fin_x += w_fin[w]; } nf_total += w_nf[w]; }
        if (nf_total >= N || pre == qn) {                    // (uniform) the walk ends inside the head
            walked = true;
            const int nb = nf_total < N ? nf_total : N;
            if (unf && nf_x < N) {
                const int id = pop_id[h];
                s.beam_node[line * N + nf_x] = id;
                r_count[nf_x] = id;                          // r_count is free now: ids of the popped nodes
                if (nf_x == 0) sh_b0 = pop_key[h];
                if (nf_x == N - 1) sh_hend = h + 1;
            }
            __syncthreads();
            const int hend = sh_hend >= 0 ? sh_hend : qn;    // fewer than N unfinished entries: the whole queue was walked
            if (fin && h < hend) { finl[fin_x] = h; atomicMax(&sh_nfin, fin_x + 1); }
            __syncthreads();
            if (tid == 0) {
                int fn = s.f_n[line], ftot = s.f_total[line];
                double* fkey = s.f_key + (long long)line * s.f_cap;
                int* fid = s.f_id + (long long)line * s.f_cap;
                for (int f = 0; f < sh_nfin; ++f) {          // '\n': finished hypothesis -> final_beam (s2s:1402)
                    const int hh = finl[f];
                    const int id = pop_id[hh];
                    const double key = pop_key[hh];
                    ++ftot;
                    int ppos = fn;
                    while (ppos > 0 && before(key, id, fkey[ppos - 1], fid[ppos - 1])) --ppos;
                    if (ppos < s.f_cap) {
                        const int last = fn < s.f_cap ? fn : s.f_cap - 1;
                        for (int q = last; q > ppos; --q) { fkey[q] = fkey[q - 1]; fid[q] = fid[q - 1]; }
                        fkey[ppos] = key; fid[ppos] = id;
                        if (fn < s.f_cap) ++fn;
                    }
                }
                int done = 0;
                if (nb == 0) done = 1;                                                 // s2s:1416
                else if (ftot > p.width_out && (fn > 0 ? fkey[0] : 0.0) > sh_b0) done = 1;   // s2s:1418-1420
                if (ftot != s.f_total[line]) { s.f_n[line] = fn; s.f_total[line] = ftot; }
                s.q_n[line] = qn - hend; s.q_n[s.B + line] = hend;
                sh_nb = nb; sh_done = done;
                if (done) { s.line_done[line] = 1; s.nact[line] = 0; atomicSub(s.active_lines, 1); }
                else s.nact[line] = nb;
            }
        }
    }
    if (!walked && tid == 0) {
        int nb = 0, h = 0;
        int fn = s.f_n[line], ftot = s.f_total[line];
        double* fkey = s.f_key + (long long)line * s.f_cap;
        int* fid = s.f_id + (long long)line * s.f_cap;
        double b0 = 0.0;
        while (h < qn && nb < N) {
            const int id = h < pre ? pop_id[h] : nid[h];
            const int chr = h < pre ? pop_chr[h] : s.n_chr[nbase + id];
            const double key = h < pre ? pop_key[h] : nkey[h];
            if (chr == p.eos) {                    // '\n': finished hypothesis -> final_beam (s2s:1402)
                ++ftot;
                int ppos = fn;
                while (ppos > 0 && before(key, id, fkey[ppos - 1], fid[ppos - 1])) --ppos;
                if (ppos < s.f_cap) {
                    const int last = fn < s.f_cap ? fn : s.f_cap - 1;
                    for (int q = last; q > ppos; --q) { fkey[q] = fkey[q - 1]; fid[q] = fid[q - 1]; }
                    fkey[ppos] = key; fid[ppos] = id;
                    if (fn < s.f_cap) ++fn;
                }
            } else {
                if (nb == 0) b0 = key;
                s.beam_node[line * N + nb] = id;
                r_count[nb] = id;                       // r_count is free now: ids of the popped nodes
                ++nb;
            }
            ++h;
        }
        int done = 0;
        if (nb == 0) done = 1;                                             // s2s:1416
        else if (ftot > p.width_out && (fn > 0 ? fkey[0] : 0.0) > b0) done = 1;   // s2s:1418-1420
        if (ftot != s.f_total[line]) { s.f_n[line] = fn; s.f_total[line] = ftot; }
        s.q_n[line] = qn - h; s.q_n[s.B + line] = h;
        sh_nb = nb; sh_done = done;
        if (done) { s.line_done[line] = 1; s.nact[line] = 0; atomicSub(s.active_lines, 1); }
        else s.nact[line] = nb;
    }
    __syncthreads();
    if (sh_done) return;
    BPROF(5);
    const int nb = sh_nb;
    // next step's inputs: fed-back scores with the better siblings reset (s2s:1515-1520); one wave per row.  Wide beams: as a grid
    // over all rows of all lines behind this kernel (beam_inputs_kernel) -- one workgroup copying its line's 256 rows took 55 us.
    if (!SPLIT)
    for (int j = wave; j < N; j += NWV) {
        const int r = line * N + j;
        float* pin = s.p_in + (long long)r * Vp;
        if (j < nb) {
            const int id = r_count[j];
            const int exp = s.n_exp[nbase + id];
            const int k = s.n_k[nbase + id];
            const float* src = s.p_base + (long long)exp * Vp;
            const short* cr = s.created + (long long)exp * CMAX;
            for (int v = lane; v < Vp; v += 64) pin[v] = src[v];
            for (int q = lane; q < k; q += 64) pin[(int)cr[q]] = 0.f;       // behind the copy, in program order of one wave
            if (lane == 0) s.prev[r] = exp;
        } else {
            for (int v = lane; v < Vp; v += 64) pin[v] = 0.f;
            if (lane == 0) s.prev[r] = line * N;
        }
    }
    BPROF(6);
#ifdef CASV_BEAM_PROF
    if (tid == 0 && line < 4096) { g_beam_wg[3 * line + 1] = (unsigned)wall_clock64(); g_beam_wg[3 * line + 2] = (unsigned)nnew; }
#endif
}
// Wide beams: the next step's input rows (the last part of beam_step_kernel) as a grid of one wave per row.
__global__ __launch_bounds__(512) void beam_inputs_kernel(const BeamState s, const BeamParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 8 + wave;
    if (r >= s.R) return;
    const int N = p.N, line = r / N, j = r - line * N;
    const int step = s.step_ptr ? *s.step_ptr : s.step_imm;
    if (s.line_done[line] || step + 1 >= s.S) return;
    const int Vp = (s.V + 31) & ~31;
    const int CMAX = (p.width_in < s.V ? p.width_in : s.V) + 1;
    const long long nbase = (long long)line * s.node_cap;
    float* pin = s.p_in + (long long)r * Vp;
    if (j < s.nact[line]) {
        const int id = s.beam_node[line * N + j];
        const int exp = s.n_exp[nbase + id];
        const int k = s.n_k[nbase + id];
        const float* src = s.p_base + (long long)exp * Vp;
        const short* cr = s.created + (long long)exp * CMAX;
        for (int v = lane; v < Vp; v += 64) pin[v] = src[v];
        for (int q = lane; q < k; q += 64) pin[(int)cr[q]] = 0.f;       // behind the copy, in program order of one wave
        if (lane == 0) s.prev[r] = exp;
    } else {
        for (int v = lane; v < Vp; v += 64) pin[v] = 0.f;
        if (lane == 0) s.prev[r] = line * N;
    }
}

// LDS plan of one launch: sort capacity (new keys held at once), staged entries of the old queue, bytes.
size_t beam_lds_bytes(int N, int width_in, int V, int q_cap, int* sort_cap, int* q_stage, int* pop_cap) {
    const int cm = (width_in < V ? width_in : V) + 1;
    *pop_cap = N + 64;                                   // the N hypotheses popped plus finished ones met on the way (more: from HBM)
    const size_t rows = (size_t)7 * (N + 1) * 4 + (size_t)(*pop_cap) * 16;
    int cap = 1;
    while (cap < N * cm && cap < BEAM_SORT_CAP) cap <<= 1;
    while (cap > 256 && rows + (size_t)cap * 12 > 56 * 1024) cap >>= 1;
    const size_t base = rows + (size_t)cap * 12;
    *sort_cap = cap;
    *q_stage = ((size_t)q_cap * 12 + base <= 56 * 1024) ? q_cap : 0;
    return base + (size_t)(*q_stage) * 12 + 16;
}
void launch_beam_step(const BeamState& s, const BeamParams& p, hipStream_t stream) {
    int q_stage = 0, sort_cap = 0, pop_cap = 0;
    const size_t lds = beam_lds_bytes(p.N, p.width_in, s.V, s.q_cap, &sort_cap, &q_stage, &pop_cap);
    BeamParams pp = p;
    pp.q_stage = q_stage; pp.sort_cap = sort_cap; pp.pop_cap = pop_cap;
    const int vpl = (s.V + 63) / 64;
    const bool wide = p.N >= 8;          // eight waves: one hypothesis row per wave at the default N = 8
    // Wide beams (the reference's default batch_size = 256 hypotheses per step, seq2seq.py:111,1414) on few lines (a page of
    // the OCR-D processor is ~40): one workgroup per line leaves most of the chip idle and walks 256 rows with 8 waves --
    // sixteen waves per line halve the row loops (expansion, selection, next inputs) and the sort / merge passes.
    const bool huge = p.N >= 64;
    // sixteen waves per line and phase A as a grid of its own (beam_expand_kernel) when the host has given it buffers
    const bool split = huge && s.rowrec != nullptr;
#define CASV_BEAM_LAUNCH(VPL_, NWV_) hipLaunchKernelGGL((beam_step_kernel<VPL_, NWV_, false>), dim3(s.B), dim3(64 * NWV_), lds, stream, s, pp)
#define CASV_BEAM_SPLIT(VPL_) do { hipLaunchKernelGGL((beam_expand_kernel<VPL_>), dim3((s.R + 7) / 8), dim3(512), 0, stream, s, pp); \
                                   hipLaunchKernelGGL((beam_step_kernel<VPL_, 16, true>), dim3(s.B), dim3(1024), lds, stream, s, pp); \
                                   hipLaunchKernelGGL(beam_inputs_kernel, dim3((s.R + 7) / 8), dim3(512), 0, stream, s, pp); } while (0)
    if (vpl <= 4) { if (split) CASV_BEAM_SPLIT(4); else if (huge) CASV_BEAM_LAUNCH(4, 16); else if (wide) CASV_BEAM_LAUNCH(4, 8); else CASV_BEAM_LAUNCH(4, 4); }
    else if (vpl <= 8) { if (split) CASV_BEAM_SPLIT(8); else if (huge) CASV_BEAM_LAUNCH(8, 16); else if (wide) CASV_BEAM_LAUNCH(8, 8); else CASV_BEAM_LAUNCH(8, 4); }
    else if (vpl <= 16) { if (split) CASV_BEAM_SPLIT(16); else if (huge) CASV_BEAM_LAUNCH(16, 16); else if (wide) CASV_BEAM_LAUNCH(16, 8); else CASV_BEAM_LAUNCH(16, 4); }
    else if (vpl <= 32) { if (wide) CASV_BEAM_LAUNCH(32, 8); else CASV_BEAM_LAUNCH(32, 4); }
    else { if (wide) CASV_BEAM_LAUNCH(64, 8); else CASV_BEAM_LAUNCH(64, 4); }
#undef CASV_BEAM_SPLIT
#undef CASV_BEAM_LAUNCH
}

// Results, best first (seq2seq.py:1538-1544): walk the trie from each finished node to the root.
__global__ void beam_extract_kernel(const BeamState s, const BeamParams p, const BeamOut o) {
    extern __shared__ int chain[];              // S node ids
    const int line = blockIdx.x / p.max_results, k = blockIdx.x % p.max_results;
    const int tid = threadIdx.x;
    const long long nbase = (long long)line * s.node_cap;
    const int fn = s.f_n[line];
    const long long ob = (long long)blockIdx.x;
    if (tid == 0 && k == 0) { o.n_found[line] = s.f_total[line]; o.n_steps[line] = s.line_steps[line]; }
    if (k >= fn) { if (tid == 0) { o.len[ob] = 0; o.score[ob] = 0.0; } return; }
    const int fnode = s.f_id[(long long)line * s.f_cap + k];
    const int len = s.n_len[nbase + fnode] - 1;
    if (tid == 0) {
        int cur = fnode;
        for (int j = len - 1; j >= 0; --j) { if (j < s.S) chain[j] = cur; cur = s.n_parent[nbase + cur]; }
        o.len[ob] = len;
        o.score[ob] = s.n_cum[nbase + fnode] / (double)len;
    }
    __syncthreads();
    for (int j = tid; j < len && j < s.S; j += blockDim.x) {
        const int nd = chain[j];
        o.idx[ob * s.S + j] = s.n_chr[nbase + nd];
        o.prob[ob * s.S + j] = s.n_prob[nbase + nd];
        o.rejpos[ob * s.S + j] = s.n_rejpos[nbase + nd];
    }
    if (o.align) {
        for (int j = 0; j < len && j < s.S; ++j) {
            const int nd = chain[j];
            const int rp = s.n_rejpos[nbase + nd];
            const float* src = o.a_base + (long long)s.n_exp[nbase + nd] * s.T;
            float* dst = o.align + (ob * s.S + j) * s.T;
            for (int t = tid; t < s.T; t += blockDim.x) dst[t] = rp >= 0 ? (t == rp ? 1.0f : 0.0f) : src[t];
        }
    }
}
// The same walk for the window form of the alignments: a rejection step is the one-hot row at its source position
// (seq2seq.py:1495), every other step the window the attention kernel recorded for the node's expansion.
__global__ void beam_extract_sparse_kernel(const BeamState s, const BeamParams p, const BeamOut o, const SparseAlignOut sp) {
    extern __shared__ int chain[];
    const int line = blockIdx.x / p.max_results, k = blockIdx.x % p.max_results;
    const int tid = threadIdx.x;
    const long long nbase = (long long)line * s.node_cap;
    const long long ob = (long long)blockIdx.x;
    if (k >= s.f_n[line]) return;
    const int fnode = s.f_id[(long long)line * s.f_cap + k];
    const int len = s.n_len[nbase + fnode] - 1;
    if (tid == 0) {
        int cur = fnode;
        for (int j = len - 1; j >= 0; --j) { if (j < s.S) chain[j] = cur; cur = s.n_parent[nbase + cur]; }
    }
    __syncthreads();
    for (int j = tid; j < len && j < s.S; j += blockDim.x) {
        const int nd = chain[j];
        const int rp = s.n_rejpos[nbase + nd];
        const long long exp = s.n_exp[nbase + nd];
        float* w = sp.w + (ob * s.S + j) * sp.K;
        int lo;
        if (rp >= 0) {
            lo = rp;
            for (int q = 0; q < sp.K; ++q) w[q] = q == 0 ? 1.0f : 0.0f;
        } else {
            const int win = sp.win_store[exp];
            if (win < 0) {
                lo = -1;
                for (int q = 0; q < sp.K; ++q) w[q] = __builtin_nanf("");
            } else {
                lo = win & 0xffff;
                const int cnt = win >> 16;
                for (int q = 0; q < sp.K; ++q) w[q] = (q < cnt && lo + q < s.T) ? o.a_base[exp * s.T + lo + q] : 0.0f;
            }
        }
        sp.lo[ob * s.S + j] = lo;
    }
}
void launch_beam_extract_sparse(const BeamState& s, const BeamParams& p, const BeamOut& o, const SparseAlignOut& sp, hipStream_t stream) {
    hipLaunchKernelGGL(beam_extract_sparse_kernel, dim3(s.B * p.max_results), dim3(256), (size_t)s.S * 4, stream, s, p, o, sp);
}

void launch_beam_extract(const BeamState& s, const BeamParams& p, const BeamOut& o, hipStream_t stream) {
    hipLaunchKernelGGL(beam_extract_kernel, dim3(s.B * p.max_results), dim3(256), (size_t)s.S * 4, stream, s, p, o);
}

}  // namespace casv
