/*
 * cor_asv_ann_hip.h -- C ABI of the MI355X-native hot path of cor-asv-ann.
 *
 * The reference has no FFI: its hot path sits behind the Python class
 * ocrd_cor_asv_ann/lib/seq2seq.py:13 `Sequence2Sequence`, which crosses into the
 * Keras/TensorFlow runtime at `predict_on_batch` / `train_on_batch`.  This header is the
 * boundary a maintainer binds (ctypes) in place of those crossings.
 * Each entry point names the reference call site(s) it replaces.
 *
 * Conventions: every function returns 0 on success and a negative casv_status otherwise;
 * the message is available from casv_last_error() (thread-local).  The caller owns all
 * host buffers (C-contiguous, float32 / int32 / float64 as declared); the library owns
 * all device memory and its HIP stream.
 *
 * Limits (CASV_ERR_ARG beyond them; the reference itself has none but host memory): depth 1..8, width a
 * multiple of 32, vocabulary 2..4096, line length T <= 4096 positions (decode steps S <= 2T <= 8192),
 * hypotheses per line and step (batch_size) <= 1024, beam_width_in >= 1 (values above the vocabulary size act like
 * the vocabulary size), at most 64 results per line, and S * batch_size * (min(beam_width_in, V) + 1) < 2^31
 * hypotheses per line.  A search keeps every expansion's state on the device: casv_decode_beam needs about
 * 2T * batch_size * ((8 * depth + 4) * width + 4 * (V + T) + 60 * (beam_width_in + 1)) bytes per line
 * (CASV_ERR_NOMEM if the device cannot hold them -- decode fewer lines per call; the Python facade does that).  A handle is bound to one HIP device and is not
 * thread-safe (the reference runs single-threaded: wrapper/transcode.py:46 max_workers=1).
 * No C++ exceptions and no callbacks cross this boundary.
 */
#ifndef COR_ASV_ANN_HIP_H
#define COR_ASV_ANN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CASV_OK = 0,
    CASV_ERR_ARG = -1,      /* bad argument / unsupported topology */
    CASV_ERR_STATE = -2,    /* call order (e.g. decode before encode) */
    CASV_ERR_HIP = -3,      /* HIP runtime failure */
    CASV_ERR_NOMEM = -4,
    CASV_ERR_NAN = -5       /* numpy would have raised "All-NaN slice" (seq2seq.py:1329,1335) */
} casv_status;

/* Topology of seq2seq.py:108-157.  residual_connections (seq2seq.py:284-291, 359-360), bridge_dense (seq2seq.py:299-301) and
 * deep_bidirectional_encoder (seq2seq.py:246-281) are built: encoder layers >= 3 hand on LSTM output + input sequence; the final
 * h / c of every encoder layer pass through Dense(width, tanh) layers "bridge<n>_h_K|b", "bridge<n>_c_K|b" ((W,W) kernel, (W) bias)
 * on their way to the decoder; with a deep bidirectional encoder EVERY encoder layer n is a BiLSTM ("enc<n>_fw_K|R|b",
 * "enc<n>_bw_K|R|b", kernels (2W,4W) for n >= 2) that reads the reference Lambda's "cross sum" of the layer below -- features 2k and
 * 2k+1 of [fw | bw] both replaced by their sum -- and hands on its backward final state; the attended width is then 2 * width for
 * every depth (att_U (2W,W), top decoder kernel (3W,4W)).  The train step adds the training graph's decoder sums (layers >= 2 and
 * the projection's input) -- which the reference's inference decoder does not have (seq2seq.py:421-436), restated as it is.
 * lm_loss/lm_predict and stateful must be 0 (both published models use the default topology, wrapper/ocrd-tool.json:61-74). */
typedef struct {
    int32_t depth;          /* seq2seq.py:117 */
    int32_t width;          /* seq2seq.py:115; must be a multiple of 32 (any other width: pad with dead units, as
                             * cor_asv_ann_amd/engine.py does -- all-zero weights keep a unit at h = c = 0, exactly) */
    int32_t voc_size;       /* seq2seq.py:123 */
    int32_t window_width;   /* attention.py:515, seq2seq.py:347 (5) */
    int32_t residual_connections, deep_bidirectional_encoder, bridge_dense, lm, stateful;
} casv_config;

/* Beam parameters of seq2seq.py:159-169 plus the constants of :1389-1397. */
typedef struct {
    int32_t batch_size;          /* N hypotheses per line per step (seq2seq.py:1414) */
    int32_t beam_width_in;       /* seq2seq.py:164 */
    int32_t beam_width_out;      /* seq2seq.py:169 */
    int32_t max_results;         /* how many finished hypotheses to return per line (>=1) */
    double  beam_threshold_in;   /* seq2seq.py:167 */
    double  rejection_threshold; /* seq2seq.py:162 */
    double  cost0;               /* seq2seq.py:1394 (3.0) */
} casv_beam_params;

typedef struct casv_model casv_model;

const char* casv_last_error(void);
int casv_device_count(void);
const char* casv_version(void);

/* Sequence2Sequence.configure() (seq2seq.py:190-489): allocate the weight set on a device. */
int casv_model_create(const casv_config* cfg, int device_id, casv_model** out);
void casv_model_destroy(casv_model* m);

/* Keras set_weights/get_weights per tensor (seq2seq.py:1172, 509-523).  Names and shapes are
 * those of SURVEY.md A.2: "E" (V,W); "enc1_fw_K|R|b", "enc1_bw_K|R|b"; "enc<n>_K|R|b";
 * "att_U" (C,W); "dec<n>_K|R|b"; "att_Wa" (W,W), "att_va" (W), "att_bUW" (W), "att_bv" (1); with bridge_dense also
 * "bridge<n>_h_K" (W,W), "bridge<n>_h_b" (W), "bridge<n>_c_K", "bridge<n>_c_b" for n = 1..depth (Keras Dense: y = x.K + b).
 * Layout is Keras': kernels (in,4W) with gate blocks i,f,c,o; row-major float32. */
int casv_set_weight(casv_model* m, const char* name, const float* data, int64_t count);
int casv_get_weight(casv_model* m, const char* name, float* out, int64_t capacity);
/* _resync_decoder() (seq2seq.py:526-528): repack the named tensors into the kernel layouts. */
int casv_commit_weights(casv_model* m);

/* encoder_model.predict_on_batch (seq2seq.py:403-406, 811, 1231).
 * Input is the sparse form of vectorize_lines' (B,T,V) array (seq2seq.py:1059-1093): per
 * position up to A (index, value) pairs, index < 0 = empty slot; a position with no pairs
 * is true-zero padding.  src_rej[b*T+t] = argmax of that input row or -1 if it is all zero
 * (what seq2seq.py:1458-1462 reads back during the beam search).
 * Leaves enc_out, u = attention_dense(enc_out) and the initial decoder states on the device. */
int casv_encode(casv_model* m, int32_t B, int32_t T, int32_t A,
                const int32_t* idx, const float* val, const int32_t* src_rej);
/* Copy the encoder outputs back (parity tests; mirrors the list encoder_model returns):
 * enc_out (B,T,C); states (2*depth, B, W) ordered h1,c1,...,hd,cd.  Either may be NULL. */
int casv_get_encoder_outputs(casv_model* m, float* enc_out, float* states);

/* The `encoder_outputs=` argument of decode_sequence_greedy / decode_sequence_beam (seq2seq.py:1305-1308,1382-1386) and
 * the `attention_input` + state inputs of decoder_model (seq2seq.py:470-473): install encoder outputs computed elsewhere
 * instead of running the encoder.  enc_out (B,T,C); states (2*depth, B, W) ordered h1,c1,...,hd,cd; a0 (B,T) initial
 * alignment or NULL for zeros (what the encoder returns, seq2seq.py:307-309); src_rej as in casv_encode or NULL. */
int casv_set_encoder_outputs(casv_model* m, int32_t B, int32_t T, const float* enc_out, const float* states,
                             const float* a0, const int32_t* src_rej);

/* One decoder_model.predict_on_batch (seq2seq.py:477-480, 1245, 1321, 1428) on explicit
 * inputs, for parity tests: R rows, `line[r]` selects the encoded line each row attends to.
 * p_in (R,V); states_in (2*depth, R, W) + a_in (R,T)  ->  probs (R,V), states_out, a_out. */
int casv_decoder_step(casv_model* m, int32_t R, const int32_t* line,
                      const float* p_in, const float* states_in, const float* a_in,
                      float* probs, float* states_out, float* a_out);

/* decode_batch_greedy (seq2seq.py:1215-1286; mode 0: argmax without index 0, S = 2T steps for
 * all lines) and decode_sequence_greedy for every line at once (seq2seq.py:1288-1354; mode 1:
 * argmax over all V with the index-0 NaN write-back, a line's steps after its '\n' are not
 * reported).  out_idx/out_prob (B,S); out_len (B) = reported steps per line (mode 0: S);
 * out_align (B,S,T) or NULL. */
int casv_decode_greedy(casv_model* m, int32_t mode, int32_t S,
                       int32_t* out_idx, float* out_prob, int32_t* out_len, float* out_align);

/* decode_sequence_beam (seq2seq.py:1356-1544) for all encoded lines at once.
 * For result k < max_results of line b: out_idx/out_prob ((B*max_results), S) characters and
 * probabilities, out_len length (incl. the final '\n'), out_score = cum_cost/(length-1)
 * (seq2seq.py:1543), out_rej ((B*max_results), S) = source position if that step was a
 * rejection candidate (one-hot alignment row, seq2seq.py:1495) else -1, out_align
 * ((B*max_results), S, T) or NULL.  n_found (B) = finished hypotheses per line (0 = the
 * generator would raise StopIteration, seq2seq.py:826); n_steps (B) = search iterations run. */
int casv_decode_beam(casv_model* m, const casv_beam_params* p, int32_t S,
                     int32_t* out_idx, float* out_prob, int32_t* out_len, double* out_score,
                     int32_t* out_rej, float* out_align, int32_t* n_found, int32_t* n_steps);

/* The soft alignments of the LAST casv_decode_greedy / casv_decode_beam call in window form, for the post-decode
 * re-alignment of wrapper/transcode.py:126,279-349 (which skips entries <= 1/V anyway, transcode.py:316): the attention
 * of attention.py:553-571 is zero outside a window of <= 2*window_width+1 positions, so step s of result row n is
 * out_lo[n*S+s] = first position of the window (-1: the row is all NaN, the window fell off the line) and
 * out_w[(n*S+s)*K + k] = weight at position lo+k (zeros beyond the window; a rejection step of the beam is the one-hot
 * row at its source position, seq2seq.py:1495).  Rows and S as in the decode call (greedy: B rows; beam: B*max_results;
 * steps beyond a result's length are zero).  12 instead of T floats per step cross PCIe.  K >= 2*window_width+1. */
int casv_get_alignments_sparse(casv_model* m, int32_t K, int32_t* out_lo, float* out_w);

/* Host-side (no device call): the Viterbi re-alignment of wrapper/transcode.py:279-349 (`_alignment2path`) on the window form
 * of ONE line's alignments (n_rows steps: lo (n_rows) first position or -1 = all-NaN row, w (n_rows, K) weights), for the
 * first i_max input and j_max output positions, visiting cells with a score above min_score (the wrapper passes 1/voc_size,
 * transcode.py:127).  path (i_max + 1): output position of every input position on the path, -1 where the path does not pass
 * (path[i_max] = j_max, path[0] = 0, as the reference's dictionary); *dist = sum of 1 - score along the path.  Same float32
 * forward scores, tie rules and border behaviour (numpy's index -1) as the reference. */
int casv_realign_path(int32_t n_rows, int32_t T, int32_t K, const int32_t* lo, const float* w, int32_t i_max,
                      int32_t j_max, float min_score, int32_t* path, double* dist);

/* Adam(clipnorm) of seq2seq.py:496 (Keras defaults: lr 1e-3, beta 0.9/0.999, epsilon 1e-7, clipnorm 5). */
typedef struct {
    float lr, beta1, beta2, epsilon, clipnorm;
} casv_adam_params;

/* Start a training session: device master copies of the handle's weights plus zeroed Adam moments
 * (encoder_decoder_model.compile, seq2seq.py:494-497).  frozen_csv: comma-separated tensor-name prefixes whose
 * tensors are not trained (layer.trainable = False after load_transfer_weights, seq2seq.py:1206-1211) or NULL. */
int casv_train_begin(casv_model* m, const casv_adam_params* p, const char* frozen_csv);
/* One model.train_on_batch (mode 1, keras_train.py:195), model.test_on_batch (mode 0, keras_train.py:407: no
 * regulariser, no update) or loss + gradients without update (mode 2, parity tests).
 * enc_idx/enc_val: encoder input as in casv_encode, (B,T,A); dec_in / dec_out: (B,U) character indices of the
 * one-hot decoder input and target rows of vectorize_lines (seq2seq.py:1095-1106), -1 = true-zero row;
 * weights (B,U) temporal sample weights (seq2seq.py:1111-1112).  Dropout enters as explicit keep-masks already
 * scaled by 1/(1-rate), or NULL for none: mask_enc = 2W + (depth-1)*W floats (seq2seq.py:293-298; depth*2W with a deep
 * bidirectional encoder), mask_dec =
 * (depth-1)*W floats (seq2seq.py:363-367), mask_cell = (B, W+C) (LSTMCell(dropout), seq2seq.py:345).
 * loss = weighted categorical cross-entropy (+ embedding regulariser in the train phase); grad_norm = global
 * L2 norm of the gradients before clipping. */
int casv_train_step(casv_model* m, int32_t mode, int32_t B, int32_t T, int32_t U, int32_t A,
                    const int32_t* enc_idx, const float* enc_val, const int32_t* dec_in, const int32_t* dec_out,
                    const float* weights, const float* mask_enc, const float* mask_dec, const float* mask_cell,
                    double* loss, double* grad_norm);
/* Gradient of the last step for one tensor, in Keras layout (parity tests). */
int casv_train_get_gradient(casv_model* m, const char* name, float* out, int64_t capacity);
/* Make casv_get_weight see the current training weights (ModelCheckpoint, EarlyStopping restore; seq2seq.py:619-622). */
int casv_train_sync_weights(casv_model* m);
/* End the session: the trained weights replace the handle's weights and are repacked for inference
 * (_resync_decoder after training, seq2seq.py:645). */
int casv_train_end(casv_model* m);

/* Multi-GPU: lines are independent (seq2seq.py:113 stateful=False), so one process per GPU decodes a
 * contiguous shard of the lines with its own handle and NO data-path collective; what crosses the GPUs is one all-gather
 * of fixed-width result records per batch -- RCCL over xGMI.  The reference has no counterpart (single process,
 * wrapper/transcode.py:46).  casv_comm_unique_id: rank 0 draws the 128-byte RCCL id and hands it to the other ranks by
 * whatever channel the host program has (file, socket, MPI, torch store); casv_comm_init: every rank joins with it, on its
 * handle's device; casv_comm_all_gather: `send` (bytes_per_rank bytes, host) of every rank -> `recv` (world * bytes_per_rank
 * bytes, host, rank order), staged through device memory and gathered on the handle's stream; casv_comm_all_reduce_max: the
 * maximum of one double over the ranks (step timing; also a barrier).  RCCL is opened at the first of these calls (dlopen),
 * it is not a link dependency of the library. */
int casv_comm_unique_id(void* out128);
int casv_comm_init(casv_model* m, int32_t rank, int32_t world, const void* unique_id128);
int casv_comm_all_gather(casv_model* m, const void* send, void* recv, int64_t bytes_per_rank);
int casv_comm_all_reduce_max(casv_model* m, double* value);
int casv_comm_destroy(casv_model* m);

/* Result records packed on the device, so that the gather needs no host-side packing and no host-to-device copy: one record
 * per line = 2S+4 int32 words -- S character indices, S probabilities (bit patterns), length, score (float64, 2 words), 1 --
 * the layout of cor_asv_ann_amd/sharding.py.  casv_records_reset: a zeroed buffer of `rows` records of S steps on the handle's
 * device; casv_records_append: the best result of every line of the LAST casv_decode_beam (max_results rows per line: the
 * first) or casv_decode_greedy (mode 0) call goes to records [row_offset, row_offset + B) -- a line without a finished
 * hypothesis gets what correct_lines falls back to (its input characters, probability 1, score 0; seq2seq.py:826-836), an
 * empty padding line an empty record; casv_records_read: the buffer to the host; casv_records_device_ptr: its device address
 * (after the handle's stream has drained) for a host program that runs the collective itself (torch.distributed);
 * casv_comm_all_gather_records: RCCL all-gather of every rank's buffer, result (world * rows records, rank order) to the host. */
int casv_records_reset(casv_model* m, int32_t rows, int32_t S);
int casv_records_append(casv_model* m, int32_t row_offset);
int casv_records_read(casv_model* m, int32_t* out);
int casv_records_device_ptr(casv_model* m, void** ptr, int64_t* bytes);
int casv_comm_all_gather_records(casv_model* m, int32_t* recv);

/* Measurement support for bench.py: per-kernel HIP-event timing on the library's stream.
 * casv_profile(m, 1) starts recording for all kernel classes, casv_profile(m, 2) only for "lstm_gemm" (fewer
 * event records inside a timed region), casv_profile(m, 3) only for every 13th launch of it (an event pair keeps the next
 * launch from overlapping the kernel's tail, ~8 us each: level 2 costs a beamed decode 2 %, level 3 0.2 %),
 * casv_profile(m, 0) stops; casv_profile_read returns, for kernel class `name`
 * ("lstm_gemm", "lstm_gemm_small", "gemm", "attention", "softmax", "beam", "embed", "persist"), the number of launches, their
 * summed duration (ms) and their summed algorithmic FLOPs and bytes. */
int casv_profile(casv_model* m, int32_t enable);
int casv_profile_read(casv_model* m, const char* name, int64_t* launches, double* total_ms,
                      double* flops, double* bytes);
/* Measurement aid: average duration (ms) of `iters` isolated launches of the GEMM kernel on random
 * operands; lstm=1 selects the fused LSTM-cell epilogue (N = 4*units), gather=1 a permuted row index. */
int casv_debug_gemm(casv_model* m, int32_t lstm, int32_t M, int32_t N, int32_t K, int32_t gather,
                    int32_t iters, double* ms_per_launch);
/* Test support: ONE plain contraction C (M,N) = A (M,K) . Bt (N,K)^T (+ bias (N) or NULL) on the caller's operands through the
 * launcher every GEMM of the path goes through -- with whatever tile shape ("tile") and arithmetic ("arithmetic" / "split_bf16";
 * by entry point: 0) the options select; K a multiple of 32.  flags: 1 = the launcher may split K over workgroups and 2 = over the two wave groups of a
 * workgroup (the forms the train step's contractions take: sums in another order), 4 = Bt counts as a weight (the split-bf16
 * arithmetic keeps a pre-split image of it for the call), 8 = the operands lie K-MAJOR -- A is (K,M), Bt is (K,N), C = A^T . Bt, no
 * bias: the train step's weight gradients (csrc/gemm_tn.hip; csrc/gemm_tn_split.hip under the split arithmetic where M and N are
 * multiples of 256).  tests/test_gpu_gemm.py compares the result with a float64 product. */
int casv_debug_contract(casv_model* m, int32_t flags, int32_t M, int32_t N, int32_t K, const float* A, const float* Bt,
                        const float* bias, float* C);
/* Options: "graph" = replay the decode step through a captured hipGraph (1) or launch kernels eagerly (0);
 * "persistent" = greedy decoding through the persistent decoder (all steps in ONE launch, workgroups hand rows to each
 * other through memory: small batches, where a step is too short for a launch per kernel): -1 by batch size (default:
 * up to 512 lines), 0 never, 1 always -- the results are the same bit for bit; the same option governs the train step's
 * recurrences (every pair of plain layers walks its sequence in ONE launch forward and ONE backward unless 0).  (A process started
 * with CASV_FAULT_INJECTION=1 -- the test suite -- also accepts 2 = as -1, and one workgroup of the first forward recurrence leaves
 * without handing on: the give-up path, the step is redone with per-step launches.  Not available otherwise.);
 * "fused_backward" = 1 (default): a backward time step without a persistent form is ONE launch (cell backward inside the data
 * GEMM), 0: two;
 * "vendor_gemm" = 0 (default): every contraction runs in this library's own kernels; 1 = calibration: the train step's plain
 * whole-sequence contractions (input projections, their data gradients) go through hipBLASLt where it can be loaded at run time
 * (bench.py reports that time beside the own-kernel figure; inference never uses it);
 * "pin_limit_mb" (default 64) = inputs of casv_encode / results of casv_decode_greedy up to this size travel through a pinned staging
 * buffer of the handle (the call returns without waiting for its copies); larger ones are copied from / to the caller's arrays directly;
 * "eos" = vocabulary index of the end-of-line character '\n' (default 1: '' and '\n' sort first, seq2seq.py:580);
 * "tile" (process-wide; alias "skinny") = tile shape of the GEMM launches: -1 by size (default), 0 always 128x128,
 * 1 always 32x128, 2 = 64x128 wherever there is no split-K -- a measurement/test switch, the values computed are the same bit
 * for bit;
 * "arithmetic" (per handle; default -1) = which matrix instruction the handle's GEMM launches run on.  Operands, accumulators and
 * results are float32 either way.  0 = the fp32-input instruction (v_mfma_f32_32x32x2_f32): every sum is ONE k-ordered fmaf
 * chain, the arithmetic of all launches up to round 5 and of the persistent small-batch kernels.  1 / 2 = every fp32 operand value
 * is taken apart into three bf16 values (round to nearest, exact sum) and six products per term are contracted on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (1: as 128x128 tiles, csrc/gemm.hip; 2: as 256x256 tiles where a job fills the
 * chip that way, csrc/gemm_split.hip -- the same bits from both): all 24 mantissa bits take part, measured error against float64
 * <= 1.1 x the fp32 chain's (tests/test_gpu_gemm.py), at 6/16 of the matrix-pipe time -- but another summation order, so results
 * agree with arithmetic 0 (and with the oracle, within the tolerances of tests/) to rounding, not bit for bit.
 * -1 = BY ENTRY POINT (csrc/engine.h, arithmetic_of): the decoder steps of casv_decode_beam take 2 (R = lines x hypotheses rows
 * per step: the GEMM-bound bulk of the path); casv_encode / casv_set_encoder_outputs, casv_decode_greedy, casv_decoder_step and
 * casv_train_step take 0.  The choice never looks at the batch: a line's bits are a function of (weights, line, entry point) --
 * not of the batch it is decoded in, the tile shape, the launch form (persistent or per step) or the GPU of a sharded job
 * (tests/test_gpu_arithmetic.py).  With 1 / 2 the persistent small-batch kernels (fp32-input kernels) are not used.
 * "split_bf16" (process-wide; default -1 = none; also CASV_SPLIT_BF16 = 0 / 1 / 2 in the environment when the library is loaded):
 * override of every handle's "arithmetic" for all launches of the decode path (tests, A/B measurements). */
int casv_set_option(casv_model* m, const char* key, int64_t value);
/* Statistics of the last call (tests): "beam_max_new_keys" = most child hypotheses one line created in one search
 * iteration of the last casv_decode_beam; "beam_sort_capacity" = how many of them are sorted in LDS at once (more are
 * sorted in runs and merged by rank). */
int casv_get_stat(casv_model* m, const char* key, int64_t* value);
int casv_synchronize(casv_model* m);

#ifdef __cplusplus
}
#endif
#endif
