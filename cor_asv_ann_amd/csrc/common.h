// Shared declarations of the HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace casv {

// tanh on the transcendental units (v_exp_f32 + v_rcp_f32): |error| <= ~2e-7 absolute.  The energies
// need 11*W tanh per decoder row; libm's tanhf made this kernel VALU-bound at 5x the time.
// Value of lane (id ^ MASK) of a fully active 64-lane wave, without the LDS crossbar round trip of ds_bpermute (what
// __shfl_xor compiles to): DPP for partners inside a row of 16 lanes, the gfx950 row-swap instructions across rows.  Same
// partner, same value: butterflies built from these reduce in exactly the order of the __shfl_xor loops they replace
// (profiles/lane_xor_probe.hip checks every mask against __shfl_xor on the device).
template <int MASK> __device__ __forceinline__ int lane_xor(const int x) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "one bit");
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);                 // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4)                                                                      // i^7 then i^3
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    else if constexpr (MASK == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);            // row_ror:8
    else if constexpr (MASK == 16) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((__lane_id() & 16) ? r[0] : r[1]);
    } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((__lane_id() & 32) ? r[0] : r[1]);
    }
}
template <int MASK> __device__ __forceinline__ float lane_xor(const float x) { return __int_as_float(lane_xor<MASK>(__float_as_int(x))); }
template <int MASK> __device__ __forceinline__ double lane_xor(const double x) {
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)lane_xor<MASK>((int)(unsigned)b), hi = (unsigned)lane_xor<MASK>((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// butterfly over the wave, partners 32, 16, 8, 4, 2, 1 in this order: v = op(v, partner's v)
template <class T, class Op> __device__ __forceinline__ T wave_butterfly(T v, Op op) {
    v = op(v, lane_xor<32>(v)); v = op(v, lane_xor<16>(v)); v = op(v, lane_xor<8>(v));
    v = op(v, lane_xor<4>(v)); v = op(v, lane_xor<2>(v)); v = op(v, lane_xor<1>(v));
    return v;
}

// 4 x 4 transpose inside every quad of lanes: the caller's lane q (= lane & 3) holds column q of rows 0..3 in (v0, v1, v2, v3) and
// gets row q's four columns back -- two exchange stages (partner q ^ 1, then q ^ 2), each swapping the off-diagonal blocks.  A GEMM
// epilogue whose lanes hold one column of several rows stores 16 bytes per lane with it (gemm.hip, gemm_skinny.hip: plain epilogues).
typedef float QuadF4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ QuadF4 quad_transpose(float v0, float v1, float v2, float v3, const int q) {
    const bool b0 = q & 1, b1 = q & 2;
    const float y01 = lane_xor<1>(b0 ? v0 : v1), y23 = lane_xor<1>(b0 ? v2 : v3);
    if (b0) { v0 = y01; v2 = y23; } else { v1 = y01; v3 = y23; }
    const float z0 = lane_xor<2>(b1 ? v0 : v2), z1 = lane_xor<2>(b1 ? v1 : v3);
    if (b1) { v0 = z0; v1 = z1; } else { v2 = z0; v3 = z1; }
    return QuadF4{v0, v1, v2, v3};
}

#ifdef CASV_ACCURATE_ACT       // measurement aid (profiles/r04_split_bf16.txt, section 9): libm's functions in place of the exp2/rcp forms
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#else
__device__ __forceinline__ float fast_tanh(float x) {
    // both branches are evaluated and one is selected: a dozen straight-line instructions instead of a divergent branch per
    // call (the attention rows make 44 calls per loop iteration); the selected value is the one the branch computed
    const float ax = fabsf(x);
    const float x2 = x * x;         // odd Taylor region: avoids the cancellation of 1 - 2/(1+e^2x)
    const float small = x * (1.0f + x2 * (-0.333333333f + x2 * (0.133333333f + x2 * (-0.0539682540f + x2 * 0.0218694885f))));
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681472f));
    return ax < 0.25f ? small : copysignf(t, x);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.44269504088896340736f));
}
#endif

// The LSTM cell on gate pre-activations (Keras gate order i, f, c~, o; recurrent_activation = sigmoid,
// seq2seq.py:268-272).  ONE definition for every GEMM variant, so that a row's result does not depend on which
// variant the launcher picked for the batch it sits in.
struct LstmCellOut { float i, f, g, o, c, h; };
__device__ __forceinline__ LstmCellOut lstm_cell(float zi, float zf, float zg, float zo, float cprev) {
    LstmCellOut r;
    r.i = fast_sigmoid(zi);
    r.f = fast_sigmoid(zf);
    r.g = fast_tanh(zg);
    r.o = fast_sigmoid(zo);
    r.c = __builtin_fmaf(r.f, cprev, r.i * r.g);
    r.h = r.o * fast_tanh(r.c);
    return r;
}


// One K-segment of a GEMM's A operand: rows of `width` floats (a multiple of 32) taken from
//   base + slot * slot_stride + row_id * ld,   slot = step * step_mul + step_add,
// where row_id = rows ? rows[m] : m.  `skip_first` drops the segment (treated as zeros) at
// step 0 -- the zero initial LSTM state of the encoder.
struct Seg {
    const float* base;
    const int* rows;
    long long slot_stride;
    int step_mul, step_add;
    int ld;
    int width;
    int skip_first;
    int koff;   // offset of this segment in the fused weight's K dimension
    const float* first_base;   // if set: at step 0 the rows come from first_base + m * ld (initial LSTM state)
};

// Output / state pointer that moves with the step the same way.
struct SlotPtr {
    float* base;
    long long slot_stride;
    int step_mul, step_add;
    int ld;
};

struct GemmArgs {
    Seg a[3];
    int nseg;
    const float* Bt;      // [N][Ktot], K contiguous (LSTM: rows in gate-interleaved order)
    const float* bias;    // [N] (same order) or nullptr
    int M, N, Ktot;
    SlotPtr out;          // PLAIN: C[M][N]; LSTM: h'[M][N/4]
    // LSTM epilogue only
    Seg c_in;             // previous cell state rows (width = units), skip_first = zero state
    SlotPtr c_out;
    SlotPtr zinit;        // LSTM: pre-activation term added to the contraction, [M][4U] interleaved (x.K + b of all steps, precomputed)
    SlotPtr gates_out;    // training: activated gates i,f,g,o, [M][4U] in the interleaved column order (or null)
    SlotPtr out2;         // training: a second copy of h (the next step's cell input rows [ctx | h]) (or null)
    const int* nact;      // beam decode: live rows per line (0 = search finished); a tile without a live row is skipped --
    int nact_group;       // rows per line                                   its rows are never read
    int epi_plain;        // job of an EPI_LSTM launch that takes the PLAIN epilogue (lets independent GEMMs of both kinds share a launch)
    int accumulate;       // PLAIN: C += A.B^T instead of C = (weight-gradient sums)
    int out_zeroed;       // PLAIN with split-K: the caller has already cleared the output (no memset per launch)
    int kgroups;          // 2 = allow the two-wave-group split-K variant (train step; changes the summation order)
    int xcd_rows;         // XCD-aware tile order: the 8 XCDs form an xcd_rows x (8/xcd_rows) grid over the tile grid (0 = off)
    int b_static;         // Bt changes only at casv_commit_weights (inference weights): the split-bf16 experiment may keep a pre-split image of it
    const void* Bimg;     // set by launch_gemm_split256: that image (gemm_split.hip), or nullptr
    int ksplit;           // PLAIN: 0/1 = one block per tile; n > 1 = K split over n blocks (float atomics into C);
                          // -1 = let the launcher choose (train step only: sums become order-dependent)
    // step source
    int step_imm;
    const int* step_ptr;
};

enum { EPI_PLAIN = 0, EPI_LSTM = 1 };

// Up to four independent GEMMs in one launch (blockIdx.y selects the job): the two directions of the
// bidirectional encoder layer, or the (layer, time) cells on one anti-diagonal of the stacked encoder.
constexpr int GEMM_MAX_JOBS = 4;
struct GemmBatch {
    GemmArgs g[GEMM_MAX_JOBS];
    int count;
};

// C[M][N] (+)= sum_k A[k][m] * B[k][n] for operands that lie K-major (gemm_tn.hip: weight gradients of the train step).
// lda / ldb / M / N multiples of 4, 16-byte aligned bases; rows m >= Mstore are computed but not stored (padded operand).
struct TnArgs {
    const float* A; long long lda; const float* B; long long ldb; float* C; long long ldc;
    int M, Mstore, N, K;
    int accumulate;      // C += (gradient sums) instead of C =
    int out_zeroed;      // C = with a split K: the caller has cleared C already
    float* colsum;       // optional [Mstore]: += sum_k A[k][m] (the bias gradient that goes with a weight gradient)
    int nsplit;          // set by launch_gemm_tn: K shares of the launch
};
void launch_gemm_tn(const TnArgs& g, hipStream_t stream);
bool launch_gemm_tn_split(const TnArgs& g, hipStream_t stream);      // the same contraction on bf16x3-split operands (gemm_tn_split.hip); false: no such form for the shape
void launch_gemm(int epi, const GemmArgs& g, hipStream_t stream);
void launch_gemm_batch(int epi, const GemmBatch& b, hipStream_t stream);
void launch_gemm_skinny(int epi, const GemmBatch& b, int ksplit, int rows, hipStream_t stream);
// optional vendor path for plain contractions C[M][N] (+)= A[M][K] . Bt[N][K]^T (+ bias) (vendor_gemm.hip); false = not taken
bool vendor_gemm_nt(const float* A, long long lda, long long M, int K, const float* Bt, int N, const float* bias, float* C, long long ldc,
                    int accumulate, hipStream_t stream);
void vendor_gemm_release(hipStream_t stream);         // frees the vendor path's workspace of a stream (model destruction)
bool gemm_is_skinny(int epi, const GemmBatch& b);     // which tile shape launch_gemm_batch will pick
// tile shape of the GEMM launches: -1 = by size (default), 0 = always 128x128, 1 = always 32x128 (same results)
void set_gemm_tile_mode(int mode);
// arithmetic of the GEMM launches (gemm.hip): 0 = fp32-input matrix instruction, 1 / 2 = bf16x3-split operands (three bf16 per fp32
// value, six products, fp32 accumulation) on the bf16 matrix instruction -- fp32-accurate sums, another summation order
void set_gemm_split_override(int v);       // process-wide: -1 none (every entry point's own choice), 0 / 1 / 2 = all launches of the decode path
int gemm_split_override();
int gemm_split_enter(int mode);            // arithmetic of the launches the calling thread enqueues from here on; returns the previous one
struct SplitScope {                        // ... for the lifetime of a scope (the C-ABI entry points, engine.h: arithmetic_of)
    int prev;
    explicit SplitScope(int mode) : prev(gemm_split_enter(mode)) {}
    ~SplitScope() { gemm_split_enter(prev); }
    SplitScope(const SplitScope&) = delete; SplitScope& operator=(const SplitScope&) = delete;
};
bool gemm_split256_wants(int epi, const GemmArgs& g);
#ifdef CASV_S2_CLOCK
void s2_clock_dump();
#endif
void gemm_split_prepare(const float* Bt, int N, int K, hipStream_t stream);    // makes the image of a weight buffer now (ahead of a graph recording)
void gemm_split_invalidate(const float* Bt);     // drops the pre-split image of a weight buffer (call where it changes or is released); nullptr: all
bool launch_gemm_split256(int epi, const GemmBatch& b, hipStream_t stream);     // false: not launched (the caller takes another path)
int gemm_split_bf16();                     // the calling thread's current arithmetic (SplitScope)
long gemm_split_epoch();                   // changes whenever the option or a pre-split weight image does (key of captured step graphs)
void gemm_split_bump_epoch();

// ---- small kernels (decode_kernels.hip) ----
struct AttnArgs {
    const float* wq;        // [R][W]  h_d . W_a + b_UW
    const int* wq_rows;     // optional [R]: row r's query is row wq_rows[r] of `wq` (beam search: the query of the parent expansion, computed once per expansion)
    const float* u;         // [B][T][W]
    const float* enc;       // [B][T][C]
    const float* va;        // [W]
    const float* bv;        // [1]
    const float* a_base;    // alignment store [(S+1)*R][T]
    const int* prev;        // [R] global row id of the parent expansion
    const int* line;        // [R] or nullptr (then line = r / rows_per_line)
    int rows_per_line;
    float* ctx;             // [R][C] (row stride ctx_ld if set)
    long long ctx_ld;       // 0 = C; train step: the context goes straight into the cell's input rows [ctx | h]
    const float* ctx_mask; long long ctx_mask_ld;   // optional per-row keep-mask of the context (train step: dropout on the cell input)
    int R, T, W, C, window;
    int step_imm; const int* step_ptr;   // output slot = step + 1
    double* apos;           // [R] sum_s a'[s]*s
    int* amax1;             // [R] max(a') == 1.0
    const int* nrows;       // optional device row count (rows >= *nrows are skipped)
    const int* nact;        // optional live rows per line: row r is skipped unless r % nact_group < nact[r / nact_group]
    int nact_group;
    long long u_line, u_time, enc_line, enc_time;   // element strides of u / enc by line and by position
    int* win_out;           // optional [R]: window of this step, s_lo | cnt << 16 (train step backward)
    int* win_store;         // optional [(S+1)*R]: the same per output slot (decode: sparse form of the alignment store)
};
void launch_attention(const AttnArgs& a, hipStream_t stream);

struct SoftmaxArgs {
    const float* logits;    // [R][V]
    float* p_base;          // score store [(S+1)*R][V], output slot = step + 1
    int R, V;
    int step_imm; const int* step_ptr;
    // greedy bookkeeping (mode < 0: none)
    int mode;               // 0: argmax over 1..V-1; 1: argmax over all V with index-0 NaN write-back
    int* out_idx;           // [R][S]
    float* out_prob;        // [R][S]
    int S;
    int* nan_flag;          // set when a row is all NaN (numpy would raise)
    const int* nact;        // optional live rows per line: row r is skipped unless r % nact_group < nact[r / nact_group]
    int nact_group;
};
void launch_softmax(const SoftmaxArgs& a, hipStream_t stream);

void launch_embed_sparse(const float* E, const int* idx, const float* val, float* x0,
                         int rows, int A, int V, int W, hipStream_t stream);
void launch_advance_step(int* step_ptr, hipStream_t stream);
// elementwise helpers of the optional topologies (seq2seq.py:284-301): dst[i] += src[i]; dst[i] = tanh(src[i]) (n floats, 16-byte aligned)
void launch_add_inplace(float* dst, const float* src, long long n, hipStream_t stream);
void launch_tanh(const float* src, float* dst, long long n, hipStream_t stream);
// deep_bidirectional_encoder's "cross sum" (seq2seq.py:246-259, as computed): dst[2k] = dst[2k+1] = src[2k] + src[2k+1] (n floats, n even)
void launch_cross_sum(const float* src, float* dst, long long n, hipStream_t stream);
// Source of the result records of one decode call (pack_records_kernel): row j * row_mul of idx / prob [rows][S];
// beam: len / score per row (len == 0: no finished hypothesis -> the input line src_idx [B][T][A], slot 0);
// batched greedy: len == nullptr, the line ends at its first `eos`, empty input lines (src_idx / src_val) give empty records.
struct RecordSrc {
    const int* idx; const float* prob; const int* len; const double* score;
    const int* src_idx; const float* src_val;
    int B, S, T, A, row_mul, eos;
};
void launch_pack_records(const RecordSrc& r, int* rec, hipStream_t stream);
void launch_scatter_rows(const float* src, int src_ld, float* dst, int dst_ld, int rows, int width,
                         int dst_row_mul, hipStream_t stream);
// Small fills and row copies batched into one launch (decode_kernels.hip): op = fill `n` 4-byte words at dst with `fill`
// (src == nullptr), or copy rows: word w < width of source row r (stride src_ld) -> destination row r * row_mul (stride dst_ld),
// n = rows * width.
struct SmallOp { void* dst; const void* src; long long n; long long src_ld, dst_ld; int width, row_mul; unsigned fill; };
constexpr int SMALL_OPS_MAX = 28;
static_assert(SMALL_OPS_MAX >= 10 + 2 * 8 + 2, "the greedy decode's set-up launch takes 10 + 2 * depth operations (depth <= 8) and the encoder's 1 + depth");
struct SmallOps {
    SmallOp op[SMALL_OPS_MAX]; int count;
    int dropped;            // operations that did not fit (sticky): launch_small_ops then launches NOTHING and says so
    bool fill(void* dst, size_t bytes, unsigned word = 0u) {
        if (count >= SMALL_OPS_MAX || (bytes & 3)) { ++dropped; return false; }
        op[count++] = SmallOp{dst, nullptr, (long long)(bytes / 4), 0, 0, 1, 1, word}; return true;
    }
    bool rows(const float* src, long long src_ld, float* dst, long long dst_ld, int nrows, int width, int row_mul) {
        if (count >= SMALL_OPS_MAX) { ++dropped; return false; }
        op[count++] = SmallOp{dst, src, (long long)nrows * width, src_ld, dst_ld, width, row_mul, 0u}; return true;
    }
};
bool launch_small_ops(const SmallOps& ops, hipStream_t stream);    // false: an operation had been dropped (nothing launched)

// ---- beam search (beam_kernels.hip) ----
struct BeamParams {
    int N, width_in, width_out, max_results;
    double threshold_in, rejection, cost0;
    int q_stage;   // set by launch_beam_step: entries of the old queue staged in LDS
    int sort_cap;  // set by launch_beam_step: new keys sorted in LDS at once
    int pop_cap;   // set by launch_beam_step: entries at the head of the merged queue kept in LDS for the pop loop
    int eos;       // vocabulary index of the end-of-line character
};

// What phase A of the beam step finds out about one expansion row (beam_kernels.hip)
struct RowRec { int count, beampos, rej, srcpos, nan, rejlate; };

struct BeamState {
    // per line
    int B, T, V, S, R;          // R = B*N
    int node_cap, q_cap, f_cap;
    // node pool [B][node_cap]
    int* n_parent; int* n_chr; float* n_prob; double* n_cum; int* n_len; int* n_exp;
    int* n_k; int* n_rejpos; double* n_pos; int* n_is1;
    int* n_count;               // [B]
    // per expansion row [(S+1)*R]
    short* created;             // [..][beam_width_in + 1] vocabulary indices of the children, in creation order
    // queue: two buffers [2][B][q_cap] of (key, node id) sorted best-first (step parity selects the live one);
    // q_n[0..B) = entries, q_n[B..2B) = offset of the first entry after the last pop
    double* q_key; int* q_id; int* q_n;
    // scratch for steps that create more keys than the LDS sorts at once: [B][2][g_cap]
    double* g_key; int* g_id; int g_cap;
    // finals
    double* f_key; int* f_id; int* f_n; int* f_total;
    // current beam
    int* beam_node;             // [R]
    int* nact;                  // [B] active rows of the current step
    int* line_done;             // [B]
    int* line_steps;            // [B]
    int* active_lines;          // [0] lines still searching, [1] statistic: most new keys of any line in any step
    // step io
    int* prev;                  // [R]
    float* p_in;                // [R][V]
    const float* p_base;        // score store
    const float* logits;        // [R][Vp] of this step: the kernel turns them into the step's row of the score store itself
                                // (the arithmetic of softmax_kernel); nullptr: a softmax launch has done that already
    const double* apos; const int* amax1;
    const int* src_rej;         // [B][T]
    const int* step_ptr;        // step number in device memory (graph replay), or nullptr: step_imm
    int step_imm;
    // wide beams (N >= 64): phase A as its own grid -- per row its record and the list of its children in creation order
    RowRec* rowrec;             // [R] or nullptr (then the per-line kernel does phase A itself)
    short* cand_idx; float* cand_val;   // [R][min(beam_width_in, V) + 1]
};
void launch_beam_init(const BeamState& s, const BeamParams& p, hipStream_t stream);
void launch_beam_step(const BeamState& s, const BeamParams& p, hipStream_t stream);
#ifdef CASV_GEMM_PROF
void gemm_prof_dump();               // diagnostic build: in-kernel cycle stamps of the 128x128 LSTM GEMM (gemm.hip)
void skinny_prof_dump();             // ... and of the 64x128 LSTM launches (gemm_skinny.hip)
#endif
#ifdef CASV_BEAM_PROF
void beam_prof_dump(int steps);      // diagnostic build: phase times of the beam step kernel (beam_kernels.hip)
#endif
struct BeamOut {
    int* idx; float* prob; int* len; double* score; int* rejpos; float* align; int* n_found; int* n_steps;
    const float* a_base;
};
void launch_beam_extract(const BeamState& s, const BeamParams& p, const BeamOut& o, hipStream_t stream);
// Window form of the alignments: (lo, K weights) per result row and step; lo = -1 marks an all-NaN row.
struct SparseAlignOut { int* lo; float* w; int K; const int* win_store; };
void launch_beam_extract_sparse(const BeamState& s, const BeamParams& p, const BeamOut& o, const SparseAlignOut& sp, hipStream_t stream);
void launch_greedy_extract_sparse(const float* a_base, const int* win_store, int B, int S, int T, const SparseAlignOut& sp, hipStream_t stream);

// ---- persistent small-batch decoder (persist.hip) ----
struct PersistLayer {
    const float* w;        // [W/16 unit groups][4 gates][16 units][Kt], K of every 16-tile in MFMA-group order
    const float* bias;     // [W/16][4][16]
    int Kt;
};
struct PersistArgs {
    int R, D, W, V, Vp, C, T, S, mode;
    int lda;                               // row stride of the staged activation rows in LDS: widest K of any task + 4
    PersistLayer layer[8];
    const float* wa; const float* bUW;     // attention query weights [W][W] (packed K order), bias
    const float* e;                        // tied output projection [Vp][W] (packed K order; rows >= V are zero)
    float* h[8]; float* c[8];              // state stores [(S+1)*R][W], slot s+1 = outputs of step s
    float* ctx;                            // [(S+1)*R][C]
    float* wq;                             // [(S+1)*R][W]
    float* logits;                         // [(S+1)*R][Vp]
    AttnArgs att;                          // u, enc, v_a, b_v, alignment store, window store (wq / ctx are set per step)
    int* out_idx; float* out_prob; int* nan_flag;
    unsigned* counters;                    // persist_counter_bytes(); zeroed ahead of the launch
    int g_lstm, g_att, g_plain;            // workgroups per role
    unsigned long long* prof;              // diagnostic build only (CASV_PERSIST_PROF): 32 tick sums, see persist.hip
};
// persistent encoder: x0 [B][T][W] embedded input, H1 [B][T][2W] layer-1 outputs (fw | bw), Hn[n-2] [B][T][W] outputs of layer
// n >= 2, cfin [(D+1)][B][W] final cell states (slot 0: backward direction of layer 1, slot n-1: layer n, slot D: forward)
struct PersistEncArgs {
    int B, T, D, W, lda;
    PersistLayer l1[2];            // forward, backward
    PersistLayer ln[7];            // layers 2..D
    const float* x0; float* H1; float* Hn[7]; float* cfin;
    unsigned* counters;            // persist_enc_counter_bytes(); zeroed ahead of the launch
    unsigned long long* prof;      // diagnostic build only (CASV_PERSIST_PROF): tick sums of workgroup 0, see persist.hip
};
size_t persist_enc_counter_bytes(int B, int D);
// workgroups per CU the runtime admits for the persistent kernels at a given dynamic-LDS size (capped at 2; 0: none)
int persist_encode_blocks_per_cu(size_t lds_bytes);
int persist_decode_blocks_per_cu(size_t lds_bytes);
int launch_persist_encode(const PersistEncArgs& pa, int grid, hipStream_t stream);
size_t persist_counter_bytes(int R, int D);
size_t persist_lds_bytes(const PersistArgs& pa);
int launch_persist_decode(const PersistArgs& pa, hipStream_t stream);   // -1: the staged rows do not fit the LDS

}  // namespace casv
