// Memory-bound kernels of the train step (kt:195 `train_on_batch`): everything of forward / backward /
// update that is not a contraction.  All sequence tensors of the train step are TIME-MAJOR,
// [t][b][feature], so that one time step is a contiguous [B][F] slab (a GEMM operand) and a whole
// sequence is a [T*B][F] matrix (operand of the batched weight-gradient GEMMs).
#include "train_kernels.h"
#include "attn_bwd.h"
#include <math.h>

namespace casv {

__device__ __forceinline__ float wsum(float v) { return wave_butterfly(v, [](float a, float b) { return a + b; }); }

// ---- transpose: dst[c][r] = src[r][c] (32x32 tiles through LDS) ----
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, long long ld_src,
                                 float* __restrict__ dst, long long ld_dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * ld_src + c] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * ld_dst + r] = tile[tx][i];
    }
}
void launch_transpose(const float* src, int rows, int cols, long long ld_src, float* dst, long long ld_dst, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, src, rows, cols, ld_src, dst, ld_dst);
}

// ---- embedding rows, time-major output: out[(t*B+b)][:] = sum_a val * E[idx[b][t][a]] ----
__global__ void embed_tm_kernel(const float* __restrict__ E, const int* __restrict__ idx, const float* __restrict__ val,
                                float* __restrict__ out, int B, int T, int A, int V, int W) {
    const int t = blockIdx.x / B, b = blockIdx.x % B;
    const long long in = ((long long)b * T + t) * A;
    for (int w = threadIdx.x * 4; w < W; w += blockDim.x * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < A; ++k) {
            const int i = idx[in + k];
            if (i < 0 || i >= V) continue;
            const float c = val ? val[in + k] : 1.0f;
            const float4 e = *reinterpret_cast<const float4*>(E + (long long)i * W + w);
            acc.x += c * e.x; acc.y += c * e.y; acc.z += c * e.z; acc.w += c * e.w;
        }
        *reinterpret_cast<float4*>(out + ((long long)t * B + b) * W + w) = acc;
    }
}
void launch_embed_tm(const float* E, const int* idx, const float* val, float* out, int B, int T, int A, int V, int W, hipStream_t st) {
    hipLaunchKernelGGL(embed_tm_kernel, dim3(B * T), dim3(128), 0, st, E, idx, val, out, B, T, A, V, W);
}

// dE[idx] += val * dX[(t*B+b)]   (float atomics: rows of frequent characters collide)
__global__ void embed_scatter_kernel(float* __restrict__ dE, const int* __restrict__ idx, const float* __restrict__ val,
                                     const float* __restrict__ dX, long long ld_dx, int B, int T, int A, int V, int W) {
    const int t = blockIdx.x / B, b = blockIdx.x % B;
    const long long in = ((long long)b * T + t) * A;
    const float* src = dX + ((long long)t * B + b) * ld_dx;
    for (int k = 0; k < A; ++k) {
        const int i = idx[in + k];
        if (i < 0 || i >= V) continue;
        const float c = val ? val[in + k] : 1.0f;
        for (int w = threadIdx.x; w < W; w += blockDim.x) atomicAdd(dE + (long long)i * W + w, c * src[w]);
    }
}
void launch_embed_scatter(float* dE, const int* idx, const float* val, const float* dX, long long ld_dx, int B, int T, int A,
                          int V, int W, hipStream_t st) {
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(B * T), dim3(128), 0, st, dE, idx, val, dX, ld_dx, B, T, A, V, W);
}

// ---- out[r][f] = in[r][f] * mask[f] (+ add[r][f]) : time-constant dropout masks (seq2seq.py:298,367) ----
__global__ void mul_mask_kernel(const float* __restrict__ in, long long ld_in, const float* __restrict__ mask,
                                float* __restrict__ out, long long ld_out, long long rows, int F) {
    const long long n = rows * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / F; const int f = (int)(i % F);
        out[r * ld_out + f] = in[r * ld_in + f] * (mask ? mask[f] : 1.0f);
    }
}
void launch_mul_mask(const float* in, long long ld_in, const float* mask, float* out, long long ld_out, long long rows, int F, hipStream_t st) {
    const long long n = rows * F;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(mul_mask_kernel, dim3(blocks), dim3(256), 0, st, in, ld_in, mask, out, ld_out, rows, F);
}

__global__ void add_mul_mask_kernel(const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
                                    const float* __restrict__ mask, float* __restrict__ out, long long ld_out, long long rows, int F) {
    const long long n = rows * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / F; const int f = (int)(i % F);
        out[r * ld_out + f] = (a[r * lda + f] + b[r * ldb + f]) * (mask ? mask[f] : 1.0f);
    }
}
void launch_add_mul_mask(const float* a, long long lda, const float* b, long long ldb, const float* mask, float* out, long long ld_out,
                         long long rows, int F, hipStream_t st) {
    const long long n = rows * F;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(add_mul_mask_kernel, dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, mask, out, ld_out, rows, F);
}
__global__ void tanh_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dy[i] *= 1.0f - y[i] * y[i];
}
void launch_tanh_bwd(float* dy, const float* y, long long n, hipStream_t st) {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, y, n);
}

// ---- out[(t*B+b)][f] = in[(t*B+b)][f] * mask[b][f]: per-sample, time-constant mask ----
__global__ void mul_rowmask_kernel(const float* __restrict__ in, long long ld_in, const float* __restrict__ mask, long long ld_mask,
                                   float* __restrict__ out, long long ld_out, long long rows, int B, int F) {
    const long long n = rows * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / F; const int f = (int)(i % F);
        out[r * ld_out + f] = in[r * ld_in + f] * (mask ? mask[(r % B) * ld_mask + f] : 1.0f);
    }
}
void launch_mul_rowmask(const float* in, long long ld_in, const float* mask, long long ld_mask, float* out, long long ld_out,
                        long long rows, int B, int F, hipStream_t st) {
    const long long n = rows * F;
    hipLaunchKernelGGL(mul_rowmask_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st, in, ld_in, mask,
                       ld_mask, out, ld_out, rows, B, F);
}

// ---- softmax + weighted categorical cross-entropy (Keras), dlogits in place ----
// One wave per row, a workgroup's waves walk the rows with the grid's stride; the loss terms are summed per wave and per
// workgroup first (float64) and reach the accumulator as ONE atomic per workgroup: 52 224 adds to a single address -- one per
// row -- were 0.63 ms of the step (profiles/r02_train_kernel_stats.csv: softmax_ce_kernel), every one serialised behind the
// others at the memory side.
__global__ __launch_bounds__(256) void softmax_ce_kernel(float* __restrict__ logits, const int* __restrict__ target,
                                                         const float* __restrict__ weight, long long rows, int B, int U, int V,
                                                         int Vp, float inv_count, double* __restrict__ loss, int want_grad) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mine = 0.0;
    for (long long r = blockIdx.x * 4LL + wave; r < rows; r += 4LL * gridDim.x) {
        float* x = logits + r * Vp;
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
        m = wave_butterfly(m, [](float a, float b) { return fmaxf(a, b); });
        float sum = 0.f;
        for (int v = lane; v < V; v += 64) sum += expf(x[v] - m);
        sum = wsum(sum);
        const long long src = (r % B) * U + r / B;        // row r = t*B + b  <-  (b, t) of the caller's (B,U) arrays
        const int tg = target[src];
        const float wgt = weight[src];
        float pt = 0.f;
        if (tg >= 0 && tg < V) pt = expf(x[tg] - m) / sum;
        const bool ok = tg >= 0 && pt > 1e-7f && pt < 1.0f - 1e-7f;      // tf.clip_by_value passes gradient inside only
        if (tg >= 0) {
            const float pc = fminf(fmaxf(pt, 1e-7f), 1.0f - 1e-7f);
            mine += (double)(-logf(pc) * wgt * inv_count);               // (the same value in every lane)
        }
        if (want_grad) {
            const float sc = ok ? wgt * inv_count : 0.0f;
            for (int v = lane; v < Vp; v += 64) {
                float gvl = 0.f;
                if (v < V) gvl = (expf(x[v] - m) / sum - (v == tg ? 1.0f : 0.0f)) * sc;
                x[v] = gvl;
            }
        }
    }
    if (lane == 0) part[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s4 = part[0] + part[1] + part[2] + part[3];
        if (s4 != 0.0) atomicAdd(loss, s4);
    }
}
void launch_softmax_ce(float* logits, const int* target, const float* weight, int B, int U, int V, int Vp, float inv_count,
                       double* loss, int want_grad, hipStream_t st) {
    const long long rows = (long long)B * U;
    const long long wgs = std::min<long long>((rows + 3) / 4, 2048);
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)wgs), dim3(256), 0, st, logits, target, weight, rows, B, U, V,
                       Vp, inv_count, loss, want_grad);
}

// ---- LSTM cell backward, pointwise part ----
// dh = a*mask_a + b + c ; gates/dz in the interleaved column order of the fused weight
__global__ void lstm_bwd_kernel(const LstmBwdBatch batch) {
    const LstmBwdArgs& p = batch.a[blockIdx.y];
    const int W = p.W;
    const long long n = (long long)p.rows * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / W; const int u = (int)(i % W);
        float dh = 0.f;
        if (p.a) dh += p.a[r * p.lda + u] * (p.mask_a ? p.mask_a[u] : 1.0f);
        if (p.b) dh += p.b[r * p.ldb + u];
        if (p.c) dh += p.c[r * p.ldc + u];
        const long long gi = r * 4 * W + (u >> 5) * 128 + (u & 31);
        const float ig = p.gates[gi], fg = p.gates[gi + 32], gg = p.gates[gi + 64], og = p.gates[gi + 96];
        const float cc = p.cell[r * W + u];
        const float cp = p.c_prev ? p.c_prev[r * p.ld_cprev + u] : 0.0f;
        const float tc = tanhf(cc);
        const float dov = dh * tc;
        const float dct = dh * og * (1.0f - tc * tc) + p.dc[r * W + u];
        p.dz[gi] = dct * gg * ig * (1.0f - ig);
        p.dz[gi + 32] = dct * cp * fg * (1.0f - fg);
        p.dz[gi + 64] = dct * ig * (1.0f - gg * gg);
        p.dz[gi + 96] = dov * og * (1.0f - og);
        p.dc[r * W + u] = dct * fg;
    }
}
void launch_lstm_bwd_batch(const LstmBwdBatch& b, hipStream_t st) {
    if (b.count < 1) return;
    long long n = 0;
    for (int j = 0; j < b.count; ++j) n = std::max(n, (long long)b.a[j].rows * b.a[j].W);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096), b.count), dim3(256), 0, st, b);
}
void launch_lstm_bwd(const LstmBwdArgs& p, hipStream_t st) {
    LstmBwdBatch b{};
    b.a[0] = p; b.count = 1;
    launch_lstm_bwd_batch(b, st);
}

// ---- attention backward for one decoder time step (oracle/train.py; no gradient through the window
// mask nor through the previous alignment, attention.py:567) ----
// One workgroup per sample.  Everything that is read is independent of what is written, so the loads batch; the
// read-modify-writes of d_enc / du go out as fire-and-forget float atomics (distinct addresses: no contention).
// dva / dbv are kept as per-sample partial sums across the steps and reduced once after the loop.
__global__ __launch_bounds__(256) void attention_bwd_kernel(const AttnBwdArgs p) {
    __shared__ float s_dx[2048];
    __shared__ float s_da[16], s_ds[16], s_av[16];
    attention_bwd_sample<false>(p, blockIdx.x, true, threadIdx.x, 256, s_dx, s_da, s_ds, s_av);
}
// ---- the deferred sums of the attention backward (attn_bwd.h, DEFER): one wave per (sample, 64 columns), its share of the
// sample's [T][64] sums in LDS, one pass over the steps (loads of four steps in flight together) ----
__global__ __launch_bounds__(64) void attention_deferred_enc_kernel(const AttnDeferArgs p) {
    extern __shared__ float s_acc[];                          // [T][64]
    const int lane = threadIdx.x, c = blockIdx.x * 64 + lane, b = blockIdx.y;
    const int T = p.T, B = p.B;
    for (int s = 0; s < T; ++s) s_acc[s * 64 + lane] = 0.0f;
    const float mk = p.mcell ? p.mcell[(long long)b * p.ld_mcell + p.mc_off + c] : 1.0f;
    constexpr int NB = 4;
    for (int t0 = 0; t0 < p.U; t0 += NB) {
        float dx[NB], av[NB]; int wv[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + j < p.U ? t0 + j : p.U - 1;
            wv[j] = p.WIN[(long long)t * B + b];
            dx[j] = p.dRec[((long long)t * B + b) * p.ld_drec + c];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = t0 + j < p.U ? t0 + j : p.U - 1;
            const int s_lo = wv[j] & 0xffff, cnt = wv[j] >> 16;
            av[j] = lane < cnt ? p.Ast[((long long)(t + 1) * B + b) * T + s_lo + lane] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (t0 + j < p.U) {
                const int s_lo = wv[j] & 0xffff, cnt = wv[j] >> 16;
                const float d = dx[j] * mk;
                for (int i = 0; i < cnt; ++i) s_acc[(s_lo + i) * 64 + lane] += __shfl(av[j], i, 64) * d;
            }
        }
    }
    for (int s = 0; s < T; ++s) {
        float* o = p.d_enc + ((long long)s * B + b) * p.C + c;
        *o += s_acc[s * 64 + lane];
    }
}
__global__ __launch_bounds__(64) void attention_deferred_u_kernel(const AttnDeferArgs p) {
    extern __shared__ float s_acc[];                          // [T][64]
    const int lane = threadIdx.x, j = blockIdx.x * 64 + lane, b = blockIdx.y;
    const int T = p.T, B = p.B, W = p.W;
    for (int s = 0; s < T; ++s) s_acc[s * 64 + lane] = 0.0f;
    const float v = p.va[j];
    const float* ub = p.u + (long long)b * W + j;             // u[s][b][j] = ub[s * B * W]
    for (int t = 0; t < p.U; ++t) {
        const int wv = p.WIN[(long long)t * B + b];
        const float q = p.WQ[((long long)t * B + b) * W + j];
        const float ds = lane < 16 ? p.DS[((long long)t * B + b) * 16 + lane] : 0.0f;
        const int s_lo = wv & 0xffff, cnt = wv >> 16;
        float uu[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) uu[i] = ub[(long long)(s_lo + (i < cnt ? i : 0)) * B * W];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const float th = fast_tanh(q + uu[i]);
            const float dpre = __shfl(ds, i, 64) * v * (1.0f - th * th);
            if (i < cnt) s_acc[(s_lo + i) * 64 + lane] += dpre;
        }
    }
    for (int s = 0; s < T; ++s) {
        float* o = p.du + ((long long)s * B + b) * W + j;
        *o += s_acc[s * 64 + lane];
    }
}
void launch_attention_deferred(const AttnDeferArgs& p, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_deferred_enc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_DEFER_MAX_T * 64 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_deferred_u_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_DEFER_MAX_T * 64 * 4);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (p.what & 1) hipLaunchKernelGGL(attention_deferred_enc_kernel, dim3(p.C / 64, p.B), dim3(64), (size_t)p.T * 64 * 4, st, p);
    if (p.what & 2) hipLaunchKernelGGL(attention_deferred_u_kernel, dim3(p.W / 64, p.B), dim3(64), (size_t)p.T * 64 * 4, st, p);
}

void launch_attention_bwd(const AttnBwdArgs& p, hipStream_t st) {
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(p.B), dim3(256), 0, st, p);
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] += x[i];
}
void launch_axpy(float* y, const float* x, long long n, hipStream_t st) {
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, st, y, x, n);
}

// ---- out[c] += sum_r in[r][c] ----
__global__ void colsum_kernel(const float* __restrict__ in, long long rows, int cols, long long ld, float* __restrict__ out,
                              int rows_per_block) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s = 0.f;
    if (c < cols)
        for (long long r = r0 + rl; r < r1; r += 4) s += in[r * ld + c];
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
void launch_colsum(const float* in, long long rows, int cols, long long ld, float* out, hipStream_t st) {
    const int rpb = 512;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, (unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, in, rows, cols, ld, out, rpb);
}

// ---- embedding regulariser (seq2seq.py:530-553): value and gradient ----
//   1 * sum_w (E[0][w] - stopgrad(mean_{v>=1} E[v][w]))^2  +  0.01 * sum_v (1 - |E[v]|^2)^2
// Two small grids instead of one workgroup walking the whole table (0.54 ms of the step): columns in blocks of 64 (four row
// groups per block meet in LDS), rows one wave each.  The gradient is added in place: every contribution to dE of this step
// has been queued on the same stream before.
__global__ __launch_bounds__(256) void reg_cols_kernel(const float* __restrict__ E, float* __restrict__ dE, int V, int W,
                                                       double* __restrict__ loss, int want_grad) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6, w = blockIdx.x * 64 + c;
    float s = 0.f;
    if (w < W) for (int v = 1 + g; v < V; v += 4) s += E[(long long)v * W + w];
    part[g][c] = s;
    __syncthreads();
    float sq = 0.f;
    if (g == 0 && w < W) {
        const float mr = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) / (float)(V - 1);
        const float d = E[w] - mr;
        sq = d * d;
        if (want_grad) dE[w] += 2.0f * d;
    }
    if (g == 0) {
        sq = wsum(sq);
        if (c == 0) atomicAdd(loss, (double)sq);
    }
}
__global__ __launch_bounds__(256) void reg_rows_kernel(const float* __restrict__ E, float* __restrict__ dE, int V, int W,
                                                       double* __restrict__ loss, int want_grad) {
    const int lane = threadIdx.x & 63, v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const float* row = E + (long long)v * W;
    float n = 0.f;
    for (int w = lane; w < W; w += 64) n += row[w] * row[w];
    n = wsum(n);
    if (want_grad) {
        const float k = -0.04f * (1.0f - n);
        for (int w = lane; w < W; w += 64) dE[(long long)v * W + w] += k * row[w];
    }
    if (lane == 0) atomicAdd(loss, (double)(0.01f * (1.0f - n) * (1.0f - n)));
}
void launch_reg(const float* E, float* dE, int V, int W, double* loss, int want_grad, hipStream_t st) {
    hipLaunchKernelGGL(reg_cols_kernel, dim3((W + 63) / 64), dim3(256), 0, st, E, dE, V, W, loss, want_grad);
    hipLaunchKernelGGL(reg_rows_kernel, dim3((V + 3) / 4), dim3(256), 0, st, E, dE, V, W, loss, want_grad);
}

// ---- global gradient norm and Adam (Keras Adam(clipnorm)) ----
__global__ void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ acc) {
    __shared__ double red[256];
    double s = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = g[i]; s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}
void launch_sumsq(const float* g, long long n, double* acc, hipStream_t st) {
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, st, g, n, acc);
}

__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, const double* __restrict__ normsq, float clipnorm, float lr_t, float b1, float b2, float eps) {
    const float norm = (float)sqrt(*normsq);
    const float scale = (clipnorm > 0.f && norm >= clipnorm) ? clipnorm / norm : 1.0f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
void launch_adam(float* w, const float* g, float* m, float* v, long long n, const double* normsq, float clipnorm, float lr_t,
                 float b1, float b2, float eps, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, st, w, g, m, v, n,
                       normsq, clipnorm, lr_t, b1, b2, eps);
}

// ---- the same over a list of tensors, one launch each ----
bool multi_add(MultiTensor& mt, float* w, float* g, float* m, float* v, long long n, int max_blocks) {
    if (mt.count >= MULTI_MAX || n < 1) return n < 1;
    const int k = mt.count++;
    mt.w[k] = w; mt.g[k] = g; mt.m[k] = m; mt.v[k] = v; mt.n[k] = n;
    if (k == 0) mt.first_block[0] = 0;
    mt.first_block[k + 1] = mt.first_block[k] + (int)std::min<long long>((n + 255) / 256, max_blocks);
    return true;
}
__device__ __forceinline__ int multi_find(const MultiTensor& mt, const int block) {      // (uniform per workgroup: scalar code)
    int k = 0;
    while (k + 1 < mt.count && block >= mt.first_block[k + 1]) ++k;
    return k;
}
__global__ void sumsq_multi_kernel(const MultiTensor mt, double* __restrict__ acc) {
    __shared__ double red[256];
    const int k = multi_find(mt, blockIdx.x);
    const float* __restrict__ g = mt.g[k];
    const long long n = mt.n[k], stride = (long long)(mt.first_block[k + 1] - mt.first_block[k]) * blockDim.x;
    double s = 0.0;
    for (long long i = (long long)(blockIdx.x - mt.first_block[k]) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = g[i]; s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(acc, red[0]);
}
void launch_sumsq_multi(const MultiTensor& mt, double* acc, hipStream_t st) {
    if (mt.count < 1) return;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)mt.first_block[mt.count]), dim3(256), 0, st, mt, acc);
}
__global__ void adam_multi_kernel(const MultiTensor mt, const double* __restrict__ normsq, float clipnorm, float lr_t, float b1, float b2,
                                  float eps) {
    const int k = multi_find(mt, blockIdx.x);
    float* __restrict__ w = mt.w[k]; const float* __restrict__ g = mt.g[k]; float* __restrict__ m = mt.m[k]; float* __restrict__ v = mt.v[k];
    const long long n = mt.n[k], stride = (long long)(mt.first_block[k + 1] - mt.first_block[k]) * blockDim.x;
    const float norm = (float)sqrt(*normsq);
    const float scale = (clipnorm > 0.f && norm >= clipnorm) ? clipnorm / norm : 1.0f;
    for (long long i = (long long)(blockIdx.x - mt.first_block[k]) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
void launch_adam_multi(const MultiTensor& mt, const double* normsq, float clipnorm, float lr_t, float b1, float b2, float eps, hipStream_t st) {
    if (mt.count < 1) return;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)mt.first_block[mt.count]), dim3(256), 0, st, mt, normsq, clipnorm, lr_t, b1, b2, eps);
}

}  // namespace casv
