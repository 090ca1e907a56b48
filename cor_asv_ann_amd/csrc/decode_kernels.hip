// HBM/L2-bound kernels of the decoder step: local monotonic additive attention (window + context),
// tied-projection softmax with greedy bookkeeping, sparse embedding of the encoder input, and the
// little helpers that move per-step state.  One 64-lane wave owns one decoder row throughout, so all
// row reductions are DPP/shuffle butterflies with a fixed order (results do not depend on batch size).
#include "common.h"
#include <algorithm>
#include "row_kernels.h"
#include <math.h>

namespace casv {

// The window rows are requested in two batches of 6: 75 registers instead of 111, six waves per SIMD instead of four -- the
// kernel is bound by its chain of dependent memory latencies, so rows in flight count for more than the round trip the second
// batch adds (c3: 8.4 -> 7.9 ms per batch; batches of 4, 3, 2 with up to eight waves: 8.5, 9.4, 8.6 ms).
constexpr int ATT_WB = 6;
__global__ __launch_bounds__(64 * ATT_ROWS) __attribute__((amdgpu_waves_per_eu(6, 6))) void attention_kernel(const AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * ATT_ROWS + wave;
    const int nrows = a.nrows ? *a.nrows : a.R;
    if (r >= nrows) return;
    if (a.nact && r % a.nact_group >= a.nact[r / a.nact_group]) return;
    const int step = a.step_ptr ? *a.step_ptr : a.step_imm;
    attention_row<false, ATT_WB>(a, r, step, lane);
}

void launch_attention(const AttnArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(attention_kernel, dim3((a.R + ATT_ROWS - 1) / ATT_ROWS), dim3(64 * ATT_ROWS), 0, stream, a);
}

// ---------------------------------------------------------------------------------------------
// softmax(h . E^T) (seq2seq.py:379) + the greedy pick of seq2seq.py:1250 / :1329-1338.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_kernel(const SoftmaxArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= a.R) return;
    if (a.nact && r % a.nact_group >= a.nact[r / a.nact_group]) return;
    const int step = a.step_ptr ? *a.step_ptr : a.step_imm;
    const int V = a.V, Vp = (V + 31) & ~31;
    const float* x = a.logits + (long long)r * Vp;
    float* p = a.p_base + ((long long)(step + 1) * a.R + r) * Vp;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
    // fmaxf drops NaN; a NaN logit row must stay NaN like numpy's max does
    float anynan = 0.0f;
    for (int v = lane; v < V; v += 64) anynan += (x[v] != x[v]) ? 1.0f : 0.0f;
    m = wave_max(m);
    if (wave_sum(anynan) > 0.0f) m = __builtin_nanf("");
    float sum = 0.0f;
    for (int v = lane; v < V; v += 64) sum += expf(x[v] - m);
    sum = wave_sum(sum);
    float best = -INFINITY; int bidx = 0x7fffffff;          // over v >= 1
    float p0 = 0.0f;
    for (int v = lane; v < Vp; v += 64) {
        // columns V..Vp-1 are padding of the K dimension of the next step's GEMM: they must be exact zeros
        // (their weight rows are zero, but 0 * stale-NaN would poison the row)
        const float pv = v < V ? expf(x[v] - m) / sum : 0.0f;
        p[v] = pv;
        if (v == 0) p0 = pv;
        if (v >= 1 && v < V && pv > best) { best = pv; bidx = v; }
    }
    if (a.mode < 0) return;
    wave_argmax(best, bidx);
    p0 = __shfl(p0, 0, 64);
    if (lane == 0) {
        int idx = bidx; float pr = best;
        if (bidx == 0x7fffffff) {                 // every candidate NaN: numpy raises here
            if (a.nan_flag) atomicOr(a.nan_flag, 1);
            idx = 1; pr = __builtin_nanf("");
        } else if (a.mode == 1) {
            // np.nanargmax over all V: index 0 wins only if strictly greater than everything after it
            if (p0 >= best && p0 == p0) p[0] = __builtin_nanf("");   // s2s:1334, stays in the feedback
        }
        a.out_idx[(long long)r * a.S + step] = idx;
        a.out_prob[(long long)r * a.S + step] = pr;
    }
}

void launch_softmax(const SoftmaxArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(softmax_kernel, dim3((a.R + 3) / 4), dim3(256), 0, stream, a);
}

// ---------------------------------------------------------------------------------------------
// char_input_projection on the encoder side (seq2seq.py:243-244) for sparse input rows:
// x0[row] = sum_a val[row][a] * E[idx[row][a]]   (one-hot: a single exact row copy)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void embed_sparse_kernel(const float* __restrict__ E, const int* __restrict__ idx,
                                                           const float* __restrict__ val, float* __restrict__ x0,
                                                           int rows, int A, int V, int W) {
    const int row = blockIdx.x;
    if (row >= rows) return;
    for (int w = threadIdx.x * 4; w < W; w += 128 * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < A; ++k) {
            const int i = idx[(long long)row * A + k];
            if (i < 0 || i >= V) continue;
            const float c = val[(long long)row * A + k];
            const float4 ev = *reinterpret_cast<const float4*>(E + (long long)i * W + w);
            acc.x += c * ev.x; acc.y += c * ev.y; acc.z += c * ev.z; acc.w += c * ev.w;
        }
        *reinterpret_cast<float4*>(x0 + (long long)row * W + w) = acc;
    }
}
void launch_embed_sparse(const float* E, const int* idx, const float* val, float* x0, int rows, int A,
                         int V, int W, hipStream_t stream) {
    hipLaunchKernelGGL(embed_sparse_kernel, dim3(rows), dim3(128), 0, stream, E, idx, val, x0, rows, A, V, W);
}

// (lo, K weights) of decode step s of line b from the alignment store (slot s+1, row b) -- decode_batch_greedy's
// per-step alignments (seq2seq.py:1249,1262) without the T-wide rows
__global__ void greedy_extract_sparse_kernel(const float* a_base, const int* win_store, int B, int S, int T, SparseAlignOut sp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (b, s)
    if (i >= B * S) return;
    const int b = i / S, s = i % S;
    const long long slot = (long long)(s + 1) * B + b;
    const int win = win_store[slot];
    float* w = sp.w + (long long)i * sp.K;
    if (win < 0) {
        sp.lo[i] = -1;
        for (int k = 0; k < sp.K; ++k) w[k] = __builtin_nanf("");
        return;
    }
    const int lo = win & 0xffff, cnt = win >> 16;
    sp.lo[i] = lo;
    for (int k = 0; k < sp.K; ++k) w[k] = (k < cnt && lo + k < T) ? a_base[slot * T + lo + k] : 0.0f;
}
void launch_greedy_extract_sparse(const float* a_base, const int* win_store, int B, int S, int T, const SparseAlignOut& sp,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(greedy_extract_sparse_kernel, dim3((B * S + 255) / 256), dim3(256), 0, stream, a_base, win_store, B, S, T, sp);
}

// ---------------------------------------------------------------------------------------------
// Result records of the lines of the last decode call, packed where the results already lie (what
// crosses the GPUs is one all-gather of fixed-width records).  Record of a line = 2S + 4 int32:
// [0,S) characters, [S,2S) probabilities (bit patterns), length, score (float64, 2 words), 1 -- the layout of
// cor_asv_ann_amd/sharding.py::pack_records.  A line without a finished hypothesis falls back to its input
// (correct_lines, seq2seq.py:826-836: the input characters with probability 1, score 0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pack_records_kernel(const RecordSrc r, int* __restrict__ rec) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= r.B) return;
    const int S = r.S, W = 2 * S + 4;
    int* out = rec + (long long)j * W;
    const long long row = (long long)j * r.row_mul;
    int len = 0;
    double score = 0.0;
    bool fallback = false;
    if (r.len) {                                  // beam: the search's best result, or the fallback
        len = r.len[row];
        if (len > 0) score = r.score[row]; else fallback = true;
    } else {                                      // batched greedy (seq2seq.py:1254-1263): up to the first end-of-line
        int any = 0;
        for (int k = lane; k < r.T * r.A; k += 64) any |= (r.src_idx[(long long)j * r.T * r.A + k] >= 0 && r.src_val[(long long)j * r.T * r.A + k] != 0.0f);
        if (__any(any)) {
            int first = S;
            for (int s0 = 0; s0 < S && first == S; s0 += 64) {
                const int s = s0 + lane;
                const unsigned long long hit = __ballot(s < S && r.idx[row * S + s] == r.eos);
                if (hit) first = s0 + __ffsll((long long)hit) - 1;
            }
            len = first < S ? first + 1 : S;
            double acc = 0.0;
            for (int s = lane; s < len; s += 64) acc += -log((double)r.prob[row * S + s]);
            acc = wave_sum_d(acc);
            score = acc / (double)(len > 0 ? len : 1);
        }
    }
    if (fallback) {
        // The input line itself (seq2seq.py:826-836): every position up to the last one that carries a symbol, in order -- an
        // unmapped character inside the line keeps its place (index 0, as the host packs it) instead of shortening the line;
        // of several alternatives at a position (confusion-network input: the slots are sorted by vocabulary index) the one with
        // the highest confidence, which is the first alternative the reference takes.
        int last = -1;
        for (int t = lane; t < r.T; t += 64) {
            bool any = false;
            for (int a = 0; a < r.A; ++a) any |= r.src_idx[((long long)j * r.T + t) * r.A + a] >= 0;
            if (any) last = t;
        }
        last = (int)wave_max((float)last);
        len = last + 1 < S ? last + 1 : S;
        for (int s = lane; s < S; s += 64) {
            int c = 0;
            if (s < len) {
                float best = -1.0f;
                for (int a = 0; a < r.A; ++a) {
                    const int ci = r.src_idx[((long long)j * r.T + s) * r.A + a];
                    const float cv = r.src_val ? r.src_val[((long long)j * r.T + s) * r.A + a] : 1.0f;
                    if (ci >= 0 && cv > best) { best = cv; c = ci; }
                }
            }
            out[s] = c > 0 ? c : 0;
            out[S + s] = s < len ? __float_as_int(1.0f) : 0;
        }
    } else {
        for (int s = lane; s < S; s += 64) {
            out[s] = s < len ? r.idx[row * S + s] : 0;
            out[S + s] = s < len ? __float_as_int(r.prob[row * S + s]) : 0;
        }
    }
    if (lane == 0) {
        out[2 * S] = len;
        const long long bits = __double_as_longlong(score);
        out[2 * S + 1] = (int)(unsigned)(bits & 0xffffffffLL); out[2 * S + 2] = (int)(unsigned)((unsigned long long)bits >> 32);
        out[2 * S + 3] = 1;
    }
}
void launch_pack_records(const RecordSrc& r, int* rec, hipStream_t stream) {
    hipLaunchKernelGGL(pack_records_kernel, dim3(r.B), dim3(64), 0, stream, r, rec);
}

__global__ void advance_step_kernel(int* step_ptr) { *step_ptr += 1; }
void launch_advance_step(int* step_ptr, hipStream_t stream) {
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, stream, step_ptr);
}

// A list of small fills and row copies as ONE launch (the set-up in front of a decode: zeroed slots, flags and counters, the
// encoder's final states spread over the hypothesis rows -- a dozen launches of ~5 us each otherwise; blockIdx.y = the operation).
__global__ void small_ops_kernel(const SmallOps ops) {
    const SmallOp& o = ops.op[blockIdx.y];
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (!o.src) {
        unsigned* d = reinterpret_cast<unsigned*>(o.dst);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < o.n; i += stride) d[i] = o.fill;
    } else {
        const unsigned* sp = reinterpret_cast<const unsigned*>(o.src);
        unsigned* d = reinterpret_cast<unsigned*>(o.dst);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < o.n; i += stride) {
            const long long r = i / o.width; const int w = (int)(i - r * o.width);
            d[r * o.row_mul * o.dst_ld + w] = sp[r * o.src_ld + w];
        }
    }
}
bool launch_small_ops(const SmallOps& ops, hipStream_t stream) {
    if (ops.dropped) return false;      // (a set-up launch that silently left out a clear or a copy would hand stale memory to what follows)
    if (ops.count < 1) return true;
    long long most = 0;
    for (int i = 0; i < ops.count; ++i) most = ops.op[i].n > most ? ops.op[i].n : most;
    const int blocks = (int)((most + 4 * 256 - 1) / (4 * 256));           // ~four words per thread for the largest operation
    hipLaunchKernelGGL(small_ops_kernel, dim3(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks, ops.count), dim3(256), 0, stream, ops);
    return true;
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(dst)[i]; const float4 b = reinterpret_cast<const float4*>(src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(dst)[i] = a;
    }
}
void launch_add_inplace(float* dst, const float* src, long long n, hipStream_t stream) {
    const long long n4 = n / 4;                  // (callers pass multiples of the hidden width: a multiple of 32)
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 4096);
    if (n4 > 0) hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, stream, dst, src, n4);
}
__global__ void tanh_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = fast_tanh(src[i]);
}
void launch_tanh(const float* src, float* dst, long long n, hipStream_t stream) {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    if (n > 0) hipLaunchKernelGGL(tanh_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n);
}

__global__ void cross_sum_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = reinterpret_cast<const float2*>(src)[i];
        const float sum = v.x + v.y;
        reinterpret_cast<float2*>(dst)[i] = make_float2(sum, sum);
    }
}
void launch_cross_sum(const float* src, float* dst, long long n, hipStream_t stream) {
    const long long n2 = n / 2;
    const int blocks = (int)std::min<long long>((n2 + 255) / 256, 4096);
    if (n2 > 0) hipLaunchKernelGGL(cross_sum_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n2);
}

__global__ void scatter_rows_kernel(const float* src, int src_ld, float* dst, int dst_ld, int rows, int width,
                                    int dst_row_mul) {
    const int i = blockIdx.x;
    for (int w = threadIdx.x; w < width; w += blockDim.x)
        dst[(long long)i * dst_row_mul * dst_ld + w] = src[(long long)i * src_ld + w];
}
void launch_scatter_rows(const float* src, int src_ld, float* dst, int dst_ld, int rows, int width,
                         int dst_row_mul, hipStream_t stream) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows), dim3(128), 0, stream, src, src_ld, dst, dst_ld, rows, width,
                       dst_row_mul);
}

}  // namespace casv
