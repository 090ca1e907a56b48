// Kernels of the train step that are not contractions (train_kernels.hip).
#pragma once
#include "common.h"
#include <algorithm>

namespace casv {

void launch_transpose(const float* src, int rows, int cols, long long ld_src, float* dst, long long ld_dst, hipStream_t st);
void launch_embed_tm(const float* E, const int* idx, const float* val, float* out, int B, int T, int A, int V, int W, hipStream_t st);
void launch_embed_scatter(float* dE, const int* idx, const float* val, const float* dX, long long ld_dx, int B, int T, int A,
                          int V, int W, hipStream_t st);
void launch_mul_mask(const float* in, long long ld_in, const float* mask, float* out, long long ld_out, long long rows, int F, hipStream_t st);
// residual_connections (seq2seq.py:284-291,359-360): out[r][f] = (a[r][f] + b[r][f]) * mask[f] (mask nullptr: times one)
void launch_add_mul_mask(const float* a, long long lda, const float* b, long long ldb, const float* mask, float* out, long long ld_out,
                         long long rows, int F, hipStream_t st);
// bridge_dense backward (seq2seq.py:299-301): dy[i] *= 1 - y[i]^2
void launch_tanh_bwd(float* dy, const float* y, long long n, hipStream_t st);
void launch_mul_rowmask(const float* in, long long ld_in, const float* mask, long long ld_mask, float* out, long long ld_out,
                        long long rows, int B, int F, hipStream_t st);
void launch_softmax_ce(float* logits, const int* target, const float* weight, int B, int U, int V, int Vp, float inv_count,
                       double* loss, int want_grad, hipStream_t st);

// The forward recurrence of up to two independent plain LSTM layers over all their time steps as ONE launch (train_persist.hip).
struct RecJob {
    const float* Wr;                 // [4W][W] recurrent weights, rows gate-interleaved in groups of 32 units (train.hip)
    const float* Z;                  // [len][B][4W]  x.Wx + b of every step
    float* hs; long long hs_ld;      // [len][B][hs_ld] outputs
    float* Cs; float* Gt;            // [len][B][W] cell states, [len][B][4W] gate activations (kept for the backward pass)
    const float* h0; const float* c0;   // [B][W] initial state or nullptr (zeros)
    int len, reverse;
    float* om; long long om_ld; const float* omask;   // optional second copy of the outputs, times a per-unit mask [W] (the next layer's dropped-out input) or nullptr
    float* zero;                     // optional [len][B][W]: cleared tile by tile along the way (the backward pass's dL/dh accumulator of this layer)
};
struct RecArgs { RecJob job[2]; int njobs, B, W; unsigned* counters; int fault; };   // fault: test of the give-up path (one workgroup leaves early)
size_t train_recurrence_counter_bytes(int B);
int train_recurrence_grid(const RecArgs& ra, int ncu);      // 0: no persistent form for this shape on this device
void launch_train_recurrence(const RecArgs& ra, int grid, hipStream_t stream);

// ... and the backward recurrence of up to two plain layers (train_persist_bwd.hip): cell backward + data GEMM of every step.
struct RecBwdJob {
    const float* WrT;                                         // [W][4W] recurrent weights, transposed (the data GEMM's operand)
    const float* dOut; long long ld_out; const float* mask;   // [len][B][ld_out] gradient w.r.t. the outputs (x feature mask [W]) or nullptr
    const float* dh_fin; const float* dc_fin;                 // [B][W] gradient w.r.t. the final state or nullptr
    const float* Gt; const float* Cs; const float* c0;        // what the forward pass kept; initial cell state or nullptr
    float* dZ;                                                // [len][B][4W] out: gate derivatives (over the forward pass's Z)
    float* dRec;                                              // [len][B][W] out, zeroed: dL/dh of the step before (slot of the first = dL/dh0)
    float* dc_out;                                            // [B][W] out: dL/dc0
    int len, reverse;
};
struct RecBwdArgs { RecBwdJob job[2]; int njobs, B, W; unsigned* counters; };
size_t train_recurrence_bwd_counter_bytes(int B);
int train_recurrence_bwd_grid(const RecBwdArgs& ra, int ncu);
void launch_train_recurrence_bwd(const RecBwdArgs& ra, int grid, hipStream_t stream);

// ... and the attention cell's forward recurrence (train_persist_top.hip): query, attention rows, cell input rows and LSTM step of
// every time step in one launch.
struct TopRecArgs {
    const float* Wr;             // [4W][C + W] the cell's recurrent-side weights, columns [ctx | h], rows gate-interleaved
    const float* WaT;            // [W][W] attention query weights (row = query column), bias bUW [W]
    const float* bUW;
    const float* Z;              // [U][B][4W]  y.Wx + b of every step
    float* RecIn;                // [U][B][C + W] the cell's input rows [ctx * mask | h(t-1)]; slot 0's h part = the initial state
    float* WQ;                   // [U][B][W] attention queries (kept for the backward pass)
    float* hs; float* Cs; float* Gt;   // [U][B][W], [U][B][W], [U][B][4W]
    const float* c0;             // [B][W] initial cell state
    int* WIN;                    // [U][B] attention windows (kept for the backward pass)
    AttnArgs att;                // u, enc, v_a, b_v, alignment store, strides, context mask (wq / ctx / step / win_out are set per step)
    int B, U, W, C;
    unsigned* counters;
};
size_t train_attention_cell_counter_bytes(int B);
int train_attention_cell_grid(const TopRecArgs& ra, int ncu);
void launch_train_attention_cell(const TopRecArgs& ra, int grid, hipStream_t stream);

struct LstmBwdArgs {
    const float* a; long long lda; const float* mask_a;     // gradient from the layer above (x dropout mask)
    const float* b; long long ldb;                           // recurrent gradient
    const float* c; long long ldc;                           // extra (attention query path)
    const float* gates; const float* cell;                   // [rows][4W] interleaved, [rows][W]
    const float* c_prev; long long ld_cprev;                 // nullptr = zero state
    float* dc;                                               // [rows][W] in: dL/dc_t, out: dL/dc_{t-1}
    float* dz;                                               // [rows][4W] interleaved
    int rows, W;
};
struct LstmBwdBatch { LstmBwdArgs a[2]; int count; };     // independent layers walking backwards in lockstep: one launch
void launch_lstm_bwd(const LstmBwdArgs& p, hipStream_t st);
void launch_lstm_bwd_batch(const LstmBwdBatch& b, hipStream_t st);

// ... and the same with the data GEMM out[rows][N] (+)= dz[rows][4W] . Bt[N][4W]^T of that step behind it, as one launch
// (gemm_bwd.hip).  p.dc is written (dL/dc of the previous step), dc_in is read: two different buffers.  `out` is zeroed.
struct BwdStepJob { LstmBwdArgs p; const float* dc_in; const float* Bt; float* out; long long ld_out; int N; };
struct BwdStepBatch { BwdStepJob j[2]; int count; };
void launch_lstm_bwd_gemm(const BwdStepBatch& b, hipStream_t st);

struct AttnBwdArgs {
    const float* dxh; long long ld_dxh; int ctx_off;         // dL/dx of the cell input; context part at ctx_off
    const float* mcell; long long ld_mcell; int mc_off;      // per-sample input mask (context part at mc_off) or nullptr
    const float* a; const int* win;                          // alignment rows [B][T] of this step, window (s_lo | cnt<<16)
    const float* wq; const float* va;
    const float* u; long long u_line, u_time;
    const float* enc; long long enc_line, enc_time;
    float* d_enc; float* du; float* dwq;
    float* dva_part; float* dbv_part;                        // [B][W], [B]: per-sample sums over the steps
    int B, T, W, C;
    float* ds_out;                                           // deferred form (attn_bwd.h): dL/dscore row of this step, [B][16]
};
void launch_attention_bwd(const AttnBwdArgs& p, hipStream_t st);
// The deferred sums of the persistent attention-cell backward: d_enc += sum_t a_t (x) dctx_t, du += sum_t dpre_t, per sample in LDS
// (T <= ATTN_DEFER_MAX_T positions).  dctx_t = dRec[t][b][0:C] (row stride ld_drec) x the sample's input mask.
constexpr int ATTN_DEFER_MAX_T = 288;
struct AttnDeferArgs {
    const float* dRec; long long ld_drec;                    // [U][B][ld_drec]
    const float* mcell; long long ld_mcell; int mc_off;
    const float* Ast; const int* WIN; const float* DS;       // [U+1][B][T] (row t + 1 belongs to step t), [U][B], [U][B][16]
    const float* WQ; const float* va; const float* u;        // [U][B][W], [W], [T][B][W]
    float* d_enc; float* du;                                 // [T][B][C], [T][B][W]
    int B, U, T, W, C;
    int what;                                                // bits: 1 = d_enc, 2 = du
};
void launch_attention_deferred(const AttnDeferArgs& p, hipStream_t st);

// ... and the attention cell's backward recurrence (train_persist_topb.hip): cell backward, data GEMM, attention backward and
// query-path GEMM of every time step in one launch.
struct TopBwdArgs {
    const float* WrT;            // [C + W][4W] the cell's recurrent-side weights transposed (rows: ctx columns, then h columns)
    const float* WaN;            // [W][W] attention query weights as the query-path GEMM takes them
    const float* dG;             // [U][B][W] gradient w.r.t. the cell's outputs
    const float* Gt; const float* Cs; const float* c0;     // what the forward pass kept; initial cell state
    float* dZ;                   // [U][B][4W] out (over the forward pass's Z)
    float* dRec;                 // [U][B][C + W] out, zeroed: gradient w.r.t. the cell's input rows [ctx | h(t-1)]
    float* dhatt;                // [U][B][W] out: gradient w.r.t. h(t-1) through the attention query
    float* DWQ;                  // [U][B][W] out: gradient w.r.t. the attention queries
    float* DS;                   // [U][B][16] out: dL/dscore rows (what is summed behind the recurrence is summed from them) or nullptr
    int defer;                   // attn_bwd.h DEFER bits: 1 = d_enc, 2 = du summed behind the recurrence instead of by atomics inside it
    const float* WQ; const float* Ast; const int* WIN;     // [U][B][W], [U+1][B][T], [U][B] kept by the forward pass
    float* dc_out;               // [B][W] out: dL/dc0
    AttnBwdArgs ab;              // mask, v_a, u, enc, d_enc, du, dva / dbv partial sums, sizes (dxh / a / win / wq / dwq are set per step)
    int B, U, W, C;
    unsigned* counters;
    int split_a;                 // part A (the attention backward of the row block's samples) runs as a launch of its own beside this one
};
size_t train_attention_cell_bwd_counter_bytes(int B);
int train_attention_cell_bwd_grid(const TopBwdArgs& ra, int ncu);
void launch_train_attention_cell_bwd(const TopBwdArgs& ra, int grid, hipStream_t stream);
// ... part A as its own launch (same grid, same counters), to be resident TOGETHER with the launch above (split_a = 1) on another
// stream: the two hand rows to each other as the parts of one launch do.  _fits: both fit a CU at once on this device.
bool train_attention_cell_bwd_rows_fit(const TopBwdArgs& ra);
void launch_train_attention_cell_bwd_rows(const TopBwdArgs& ra, int grid, hipStream_t stream);

void launch_axpy(float* y, const float* x, long long n, hipStream_t st);
void launch_colsum(const float* in, long long rows, int cols, long long ld, float* out, hipStream_t st);
void launch_reg(const float* E, float* dE, int V, int W, double* loss, int want_grad, hipStream_t st);
void launch_sumsq(const float* g, long long n, double* acc, hipStream_t st);
// ... over a list of tensors in ONE launch each (the train step has 30-odd parameter tensors: one launch per tensor is mostly
// launch boundaries).  first_block[k] .. first_block[k + 1] are tensor k's workgroups of the launch.
constexpr int MULTI_MAX = 48;
struct MultiTensor { float* w[MULTI_MAX]; float* g[MULTI_MAX]; float* m[MULTI_MAX]; float* v[MULTI_MAX]; long long n[MULTI_MAX]; int first_block[MULTI_MAX + 1]; int count; };
bool multi_add(MultiTensor& mt, float* w, float* g, float* m, float* v, long long n, int max_blocks);      // false: the list is full
void launch_sumsq_multi(const MultiTensor& mt, double* acc, hipStream_t st);
void launch_adam_multi(const MultiTensor& mt, const double* normsq, float clipnorm, float lr_t, float b1, float b2, float eps, hipStream_t st);
void launch_adam(float* w, const float* g, float* m, float* v, long long n, const double* normsq, float clipnorm, float lr_t,
                 float b1, float b2, float eps, hipStream_t st);

}  // namespace casv
