// Shared host-side declarations of the engine (model handle, device buffers, launch helpers).
// Host side of the C ABI (include/cor_asv_ann_hip.h): weight repacking, device buffers, and the launch
// sequences of the encoder (seq2seq.py:237-314), the decoder step (seq2seq.py:416-480) and the
// greedy / beam decode loops (seq2seq.py:1215-1544).  All device memory and the HIP stream belong to
// the handle; callers pass plain host pointers.
#pragma once
#include "common.h"
#include "../../include/cor_asv_ann_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

using namespace casv;

// Limits of the boundary (documented in include/cor_asv_ann_hip.h)
constexpr int CASV_MAX_T = 4096;        // characters per line (positions of the encoder); decode steps S <= 2 * CASV_MAX_T
constexpr int CASV_MAX_BEAM_N = 1024;   // hypotheses per line and step

inline thread_local char g_err[512] = "";
inline int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(e_ == hipErrorOutOfMemory ? CASV_ERR_NOMEM : CASV_ERR_HIP, "%s failed: %s (%s:%d)", #x, \
                hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

inline long g_devbuf_generation = 0;    // bumped by every (re)allocation: captured hipGraphs hold raw pointers
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        ++g_devbuf_generation;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(CASV_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        // CASV_POISON=1 (tests): fresh buffers hold 0xFF bytes (NaN as float, -1 as int) so that any read of memory
        // the kernels have not written shows up in the results instead of depending on what the allocator returns
        static const bool poison = getenv("CASV_POISON") && getenv("CASV_POISON")[0] == '1';
        if (poison) (void)hipMemset(p, 0xFF, bytes);
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); ++g_devbuf_generation; } p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum ProfClass { PC_LSTM = 0, PC_GEMM, PC_ATTN, PC_SOFTMAX, PC_BEAM, PC_EMBED, PC_LSTM_SMALL, PC_PERSIST, PC_COUNT };
// "lstm_gemm" = the 128x128-tile fused LSTM GEMM (the dominant kernel); "lstm_gemm_small" = its 32x128-tile variant
// "persist" = the persistent small-batch decoder (all steps of a greedy decode in one launch)
inline const char* kProfNames[PC_COUNT] = {"lstm_gemm", "gemm", "attention", "softmax", "beam", "embed", "lstm_gemm_small", "persist"};

struct Prof {
    bool on = false;
    bool only_lstm = false;          // level 2: events only around the dominant kernel (cheaper inside a timed region)
    int sample = 1; long long seen = 0;   // level 3: ... and only around every `sample`-th of its launches (an event pair keeps
                                     // the next launch from overlapping the kernel's tail: ~8 us each, 2 % of a decode step)
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Rec { hipEvent_t a, b; int cls; };
    std::vector<Rec> recs;
    double flops[PC_COUNT] = {0}, bytes[PC_COUNT] = {0};
    long long launches[PC_COUNT] = {0};
    double ms[PC_COUNT] = {0};
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void reset() { used = 0; recs.clear(); for (int i = 0; i < PC_COUNT; ++i) { flops[i] = bytes[i] = ms[i] = 0; launches[i] = 0; } }
    void collect() {
        for (auto& r : recs) { float t = 0; if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms[r.cls] += t; }
        recs.clear(); used = 0;
    }
};

struct LstmW {
    DevBuf wt, bias; int kin = 0;     // packed [4W][kin + W], gate-interleaved
    DevBuf pw, pbias;                 // decoder layers: the same values in the persistent decoder's layout (persist.hip)
};

struct TrainState;

struct casv_model {
    casv_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_inputs = nullptr;                       // marks the point where host input buffers have been consumed
    int W = 0, V = 0, Vp = 0, C = 0, D = 0;
    std::map<std::string, std::vector<float>> host;      // Keras-layout tensors
    std::map<std::string, size_t> expect;                // name -> element count
    bool committed = false;
    // packed device weights
    DevBuf E, WaT, bUW, va, bv, UT;
    DevBuf WaP, EP;                                       // query / output projection in the persistent decoder's K order
    DevBuf p_ctx, p_wq, p_logits, p_counters;             // persistent decoder: slot-indexed hand-off buffers, counters
    DevBuf p_enc_counters, d_flags;                       // persistent encoder's counters; [0] its give-up word set aside, [1] the decoder's, [2] the NaN flag
    bool enc_check_pending = false;                       // the persistent encoder's give-up word has not been looked at yet (engine.hip, settle_encoder)
    char* pin_in = nullptr; size_t pin_in_cap = 0;        // pinned staging of casv_encode's inputs (reused behind ev_inputs)
    char* pin_out = nullptr; size_t pin_out_cap = 0;      // pinned staging of the greedy decode's results
    size_t pin_limit = (size_t)64 << 20;                  // larger inputs / results bypass the pinned staging (option "pin_limit_mb")
    int persist_mode = -1;                                // -1 by size, 0 never, 1 always (greedy decode of small batches)
    int persist_skip = 0, persist_penalty = 0; bool persist_told = false;   // back-off after a persistent launch gave up waiting
    int ncu = 0;
    LstmW enc_fw, enc_bw;
    std::vector<LstmW> enc_dfw, enc_dbw;                  // deep_bidirectional_encoder: the two directions of layer n >= 2 at index n
    std::vector<DevBuf> br_hT, br_hb, br_cT, br_cb;        // bridge_dense: Dense kernels transposed [W][W] and biases of layer n at index n
    DevBuf br_tmp;
    std::vector<LstmW> enc, dec;                         // enc[n] for layer n>=2 at index n; dec[n] n=1..D
    // encoder session
    int B = 0, T = 0, A = 0;
    bool encoded = false;                                 // inputs (casv_encode) or explicit outputs (casv_set_encoder_outputs) are on the device
    int enc_arith = -1;                                   // arithmetic the encoder outputs on the device were computed in; -1: not computed yet (ensure_encoded)
    bool enc_explicit = false;                            // the outputs were handed in (casv_set_encoder_outputs): only u = attention_dense(enc_out) is the library's
    DevBuf d_idx, d_val, d_srcrej, x0, H1, Ha, Hb, Hc, cfin, hfin, u;
    float* enc_out = nullptr;
    DevBuf a0; bool has_a0 = false;                       // initial alignment handed in with casv_set_encoder_outputs
    // decode session
    int R = 0, S = 0;
    std::vector<DevBuf> st_h, st_c;
    DevBuf st_a, st_p, ctx, wq, logits, prev, pin, apos, amax1, d_step, d_line, d_nan;
    DevBuf st_q;                                          // beam search: the attention query of every expansion, [(S+1)*R][W] (engine.hip, launch_step)
    DevBuf o_idx, o_prob, o_align, st_win, sp_lo, sp_w;
    // what the last decode call left on the device (casv_get_alignments_sparse): 0 nothing, 1 greedy, 2 beam
    int last_decode = 0, last_S = 0, last_rows = 0, last_mode = 0; unsigned long long last_signature = 0;
    BeamState last_beam{}; BeamParams last_beam_params{};
    // beam
    DevBuf b_parent, b_chr, b_prob, b_cum, b_len, b_exp, b_k, b_rejpos, b_pos, b_is1, b_count, b_created;
    DevBuf b_gkey, b_gid, b_qkey, b_qid, b_qn, b_fkey, b_fid, b_fn, b_ftotal, b_beamnode, b_nact, b_done, b_steps, b_active;
    DevBuf bo_idx, bo_prob, bo_len, bo_score, bo_rej, bo_align, bo_found, bo_nsteps;
    DevBuf b_rowrec, b_candidx, b_candval;                // wide beams: phase A's per-row results (beam_expand_kernel)
    // training session (train.hip)
    TrainState* train = nullptr;
    // options
    int eos = 1;                                          // vocabulary index of '\n' (seq2seq.py:1255,1344,1402)
    bool use_graph = false;
    int arith = -1;                                       // option "arithmetic": -1 by entry point (arithmetic_of), 0 fp32-input chain, 1 / 2 split-bf16 everywhere
    bool vendor_gemm = false;                             // calibration only: the train step's plain whole-sequence contractions through hipBLASLt (vendor_gemm.hip)
    bool fused_backward = true;                           // train step: cell backward fused into the step's data GEMM (gemm_bwd.hip)
    const int* skip_nact = nullptr;                       // beam decode: live rows per line, handed to the step's kernels when
    int skip_group = 0;                                   // skipping can pay (wide beams, or a line has finished); rows per line
    // the captured step graph of the last decode configuration (option "graph"): kept across calls, rebuilt when the
    // configuration or any device buffer changes
    hipGraph_t step_graph = nullptr; hipGraphExec_t step_exec = nullptr; std::string step_graph_key;
    void* comm = nullptr; int comm_rank = 0, comm_world = 1; DevBuf comm_send, comm_recv;     // RCCL communicator (comm.hip)
    DevBuf rec; int rec_rows = 0, rec_S = 0;              // result records of this rank's lines, packed on the device (casv_records_*)
    int* pin_active = nullptr; hipEvent_t ev_active[2] = {nullptr, nullptr};   // beam decode: unfinished-line count, read one chunk behind
    int stat_beam[3] = {0, 0, 0};                         // last beam decode: most new hypotheses of one line in one step; rows stepped
                                                          // and distinct parent expansions among them (N <= 16 only)
    Prof prof;

    void prof_begin(int cls, double fl, double by, hipEvent_t& a) {
        a = nullptr;
        if (!prof.on || (prof.only_lstm && cls != PC_LSTM && cls != PC_PERSIST)) return;
        if (prof.sample > 1 && (prof.seen++ % prof.sample)) return;
        a = prof.get(); (void)hipEventRecord(a, stream);
        prof.flops[cls] += fl; prof.bytes[cls] += by; prof.launches[cls] += 1;
    }
    void prof_end(int cls, hipEvent_t a) {
        if (!a) return;
        hipEvent_t b = prof.get(); (void)hipEventRecord(b, stream);
        prof.recs.push_back({a, b, cls});
    }
};


// Which arithmetic the GEMM launches of a C-ABI call take (gemm.hip).  The rule depends on NOTHING but the
// entry point -- never on the batch:
//   ENTRY_SEARCH  casv_decode_beam (R = lines x hypotheses rows per step, the GEMM-bound bulk of the path): bf16x3-split operands on
//                 the bf16 matrix instruction (2) -- its decoder steps AND the encoder pass whose outputs it consumes (casv_encode only
//                 stages the input; the encoder runs for the first entry point that needs its outputs, in that entry point's
//                 arithmetic, and again if a later one needs the other: engine.hip, ensure_encoded);
//   ENTRY_CHAIN   the greedy decodes, the explicit decoder step, casv_get_encoder_outputs: the fp32-input instruction's k-ordered
//                 chain (0) -- the arithmetic the persistent small-batch kernels are built on;
//   ENTRY_TRAIN   casv_train_step: 2 -- the whole-sequence contractions that have a split form (input projections of all time
//                 steps, their data gradients, logits: gemm_split.hip / gemm.hip's SPLIT tiles; the K-major weight gradients:
//                 gemm_tn_split.hip) take it; the persistent recurrences and the per-time-step launches that must equal them are
//                 fp32-input kernels.
// So a line's bits are a function of (weights, line, entry point) only: they do not change with the batch it sits in, the tile
// shape, the launch form (persistent or per step), what was decoded from the same encoding before, or the shard of a multi-GPU job.
// A handle's "arithmetic" option (0 / 1 / 2) or the process-wide override put all of them on one arithmetic.
enum { ENTRY_CHAIN = 0, ENTRY_SEARCH = 1, ENTRY_TRAIN = 2 };
inline int arithmetic_of(const casv_model* m, int entry) {
    const int o = gemm_split_override();
    const int a = o >= 0 ? o : m->arith;
    return a >= 0 ? a : (entry == ENTRY_CHAIN ? 0 : 2);
}

inline int upload(DevBuf& b, const std::vector<float>& v) {
    if (int rc = b.ensure(v.size() * sizeof(float))) return rc;
    HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}


inline Seg mkseg(const float* base, int ld, int width, int koff, const int* rows = nullptr,
                 long long slot_stride = 0, int mul = 0, int add = 0, int skip_first = 0) {
    Seg s{}; s.base = base; s.rows = rows; s.slot_stride = slot_stride; s.step_mul = mul; s.step_add = add;
    s.ld = ld; s.width = width; s.skip_first = skip_first; s.koff = koff; return s;
}
inline SlotPtr mkslot(float* base, int ld, long long slot_stride = 0, int mul = 0, int add = 0) {
    SlotPtr s{}; s.base = base; s.slot_stride = slot_stride; s.step_mul = mul; s.step_add = add; s.ld = ld; return s;
}

inline void run_gemm_batch(casv_model* m, int epi, GemmBatch& b) {
    hipEvent_t a{};
    const int cls = epi == EPI_LSTM ? (m->prof.on && gemm_is_skinny(epi, b) ? PC_LSTM_SMALL : PC_LSTM) : PC_GEMM;
    double fl = 0, by = 0;
    for (int j = 0; j < b.count; ++j) {
        const GemmArgs& g = b.g[j];
        int kact = 0;
        for (int i = 0; i < g.nseg; ++i) kact += g.a[i].width;
        fl += 2.0 * g.M * (double)g.N * kact;
        by += 4.0 * ((double)g.M * kact + (double)g.N * kact + (double)g.M * g.N);
    }
    m->prof_begin(cls, fl, by, a);
    launch_gemm_batch(epi, b, m->stream);
    m->prof_end(cls, a);
}

inline void run_gemm(casv_model* m, int epi, GemmArgs& g) {
    GemmBatch b;
    b.g[0] = g;
    b.count = 1;
    run_gemm_batch(m, epi, b);
}


int casv_train_release(casv_model* m);
extern "C" int casv_comm_destroy(casv_model* m);
