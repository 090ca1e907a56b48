// Teacher-forced train step on the device: kt:195 `model.train_on_batch` / kt:407 `test_on_batch` of the
// graph `encoder_decoder_model` (seq2seq.py:237-390) compiled with weighted categorical cross-entropy and
// Adam(clipnorm=5) (seq2seq.py:494-497), plus the embedding regulariser (seq2seq.py:530-553).
//
// Layout: every sequence tensor is time-major [t][b][f].  Per LSTM layer the parameters are split into
//   Wx [4W][kx]  input part   -> ONE GEMM over all time steps gives the input pre-activations (+ bias)
//   Wr [4W][kr]  recurrent part (attention cell: [context | h]) -> one fused LSTM-cell GEMM per step
// rows in the gate-interleaved order of gemm.hip.  Backward per step = pointwise cell backward + one
// data GEMM through the recurrent part; input gradients and all weight gradients are big GEMMs over
// the whole sequence (operands transposed by a bandwidth-bound kernel).  Adam and the global-norm clip
// run on these packed tensors; Keras layouts exist only at set/get time.
#include "engine.h"
#include "train_kernels.h"

namespace {

struct TTensor { std::string name; DevBuf w, g, m, v; size_t n = 0; bool frozen = false; };

struct TLayer {
    std::string name;
    int kx = 0, kr = 0, len = 0;
    bool reverse = false;
    int iwx = -1, iwr = -1, ib = -1;
    DevBuf wxT, wrT;                       // derived: [kx][4W], [kr][4W]
    DevBuf Hown, Cs, Gt, Z, dRec;
    bool drec_cleared = false;             // this step's forward recurrence has cleared dRec (train_persist.hip: RecJob.zero)
    float* hs = nullptr; long long hs_ld = 0;
};

}  // namespace

struct TrainState {
    casv_adam_params ap{};
    long step = 0;
    std::vector<TTensor> tens;
    std::vector<TLayer> layers;            // enc1_fw, enc1_bw, enc2.., dec1..decD
    int iE = -1, iUT = -1, iWaT = -1, ibUW = -1, iva = -1, ibv = -1;
    DevBuf ETp, WaN, UaN;                  // derived: E^T padded [W][Vp], W_a [W][W], U_a [C][W]
    // bridge_dense (seq2seq.py:299-301): per encoder layer n (index n - 1) and state s (0 = h, 1 = c) the Dense kernel transposed
    // [W out][W in] (the B operand of the forward product), its bias, and -- derived -- the kernel as Keras holds it [W in][W out]
    // (the B operand of the data gradient); the bridged states, and the tanh' products of the backward pass
    struct Bridge { int ikt = -1, ib = -1; DevBuf kn; };
    std::vector<Bridge> bridge;            // [2 * (n - 1) + s]
    DevBuf hbr, cbr, brtmp, Ytop;          // [D][B][W] bridged final states; scratch [B][W]; residual_connections: top output + its input [U*B][W]
    int B = 0, T = 0, U = 0, A = 0;
    DevBuf e_idx, e_val, d_in, d_out, d_w, m_enc, m_dec, m_cell;
    DevBuf X0, H1, u, Y0, Ym, WQ, Ast, WIN, RecIn, prev, logits, dG, d_enc, du, DWQ, DSrows, dhatt, dfin, dcbuf, dcbuf2, HP, dX0, dXtop, dXl, dYl, dOin, dvaP, dbvP;
    std::vector<DevBuf> O, DO;             // masked layer outputs (encoder O[n], decoder DO[n])
    std::vector<DevBuf> XD;                // deep_bidirectional_encoder: layer n's input = the cross sum of O[n-1] (seq2seq.py:246-259)
    DevBuf loss, normsq;
    DevBuf dcalt;                          // second dL/dc buffers of the fused backward steps (two layers)
    DevBuf rec_cnt; int rec_launches = 0, rec_checked = 0, rec_skip = 0, rec_penalty = 0;   // persistent recurrences: counters per launch, back-off
    const unsigned* rec_abort[16] = {nullptr};    // ... and where each launch leaves its "gave up" word
    // The attention cell's backward recurrence as TWO launches side by side (train_persist_topb.hip, split_a): the second stream and
    // the events that tie it into the step; split_off: a step gave up with the two launches in flight -- one launch from then on.
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool split_off = false; int split_launch = -1;
    int find(const std::string& n) const { for (size_t i = 0; i < tens.size(); ++i) if (tens[i].name == n) return (int)i; return -1; }
    float* W_(int i) { return tens[i].w.as<float>(); }
    float* G_(int i) { return tens[i].g.as<float>(); }
};

static int gate_row(int W, int u, int g) { return (u / 32) * 128 + g * 32 + (u % 32); }

// Keras (K (kin,4W), R (W,4W), b) -> Wx [4W][kx], Wr [4W][kr], bias [4W]; top: K rows [W, W+C) join the recurrent part
static void pack_train_lstm(int W, int kx, int kctx, const std::vector<float>& K, const std::vector<float>& R,
                            const std::vector<float>& b, std::vector<float>& wx, std::vector<float>& wr, std::vector<float>& bias) {
    const int kr = kctx + W;
    wx.assign((size_t)4 * W * kx, 0.f); wr.assign((size_t)4 * W * kr, 0.f); bias.assign(4 * W, 0.f);
    for (int u = 0; u < W; ++u)
        for (int g = 0; g < 4; ++g) {
            const int n = gate_row(W, u, g), col = g * W + u;
            for (int k = 0; k < kx; ++k) wx[(size_t)n * kx + k] = K[(size_t)k * 4 * W + col];
            for (int k = 0; k < kctx; ++k) wr[(size_t)n * kr + k] = K[(size_t)(kx + k) * 4 * W + col];
            for (int k = 0; k < W; ++k) wr[(size_t)n * kr + kctx + k] = R[(size_t)k * 4 * W + col];
            bias[n] = b[col];
        }
}
static void unpack_train_lstm(int W, int kx, int kctx, const std::vector<float>& wx, const std::vector<float>& wr,
                              const std::vector<float>& bias, std::vector<float>& K, std::vector<float>& R, std::vector<float>& b) {
    const int kr = kctx + W;
    K.assign((size_t)(kx + kctx) * 4 * W, 0.f); R.assign((size_t)W * 4 * W, 0.f); b.assign(4 * W, 0.f);
    for (int u = 0; u < W; ++u)
        for (int g = 0; g < 4; ++g) {
            const int n = gate_row(W, u, g), col = g * W + u;
            for (int k = 0; k < kx; ++k) K[(size_t)k * 4 * W + col] = wx[(size_t)n * kx + k];
            for (int k = 0; k < kctx; ++k) K[(size_t)(kx + k) * 4 * W + col] = wr[(size_t)n * kr + k];
            for (int k = 0; k < W; ++k) R[(size_t)k * 4 * W + col] = wr[(size_t)n * kr + kctx + k];
            b[col] = bias[n];
        }
}

static int add_tensor(TrainState* ts, const std::string& name, const std::vector<float>& host, bool frozen) {
    ts->tens.emplace_back();
    TTensor& t = ts->tens.back();
    t.name = name; t.n = host.size(); t.frozen = frozen;
    const size_t bytes = host.size() * 4;
    if (t.w.ensure(bytes) || t.g.ensure(bytes) || t.m.ensure(bytes) || t.v.ensure(bytes)) return -1;
    if (hipMemcpy(t.w.p, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
    (void)hipMemset(t.m.p, 0, bytes); (void)hipMemset(t.v.p, 0, bytes); (void)hipMemset(t.g.p, 0, bytes);
    return (int)ts->tens.size() - 1;
}

static bool is_frozen(const std::string& name, const std::string& csv) {
    size_t pos = 0;
    while (pos < csv.size()) {
        size_t e = csv.find(',', pos);
        if (e == std::string::npos) e = csv.size();
        const std::string p = csv.substr(pos, e - pos);
        if (!p.empty() && name.compare(0, p.size(), p) == 0) return true;
        pos = e + 1;
    }
    return false;
}

int casv_train_release(casv_model* m) {
    if (!m || !m->train) return 0;
    TrainState* ts = m->train;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    if (ts->side) { (void)hipStreamSynchronize(ts->side); (void)hipStreamDestroy(ts->side); if (ts->ev_fork) (void)hipEventDestroy(ts->ev_fork); if (ts->ev_join) (void)hipEventDestroy(ts->ev_join); ts->side = nullptr; }
    for (auto& t : ts->tens) { t.w.release(); t.g.release(); t.m.release(); t.v.release(); }
    for (auto& l : ts->layers) { l.wxT.release(); l.wrT.release(); l.Hown.release(); l.Cs.release(); l.Gt.release(); l.Z.release(); l.dRec.release(); }
    for (auto& b : ts->bridge) b.kn.release();
    for (DevBuf* b : {&ts->hbr, &ts->cbr, &ts->brtmp, &ts->Ytop}) b->release();
    DevBuf* bufs[] = {&ts->ETp, &ts->WaN, &ts->UaN, &ts->e_idx, &ts->e_val, &ts->d_in, &ts->d_out, &ts->d_w, &ts->m_enc, &ts->m_dec,
        &ts->m_cell, &ts->X0, &ts->H1, &ts->u, &ts->Y0, &ts->Ym, &ts->WQ, &ts->Ast, &ts->WIN, &ts->RecIn, &ts->prev,
        &ts->logits, &ts->dG, &ts->d_enc, &ts->du, &ts->DWQ, &ts->DSrows, &ts->dhatt, &ts->dfin, &ts->dcbuf, &ts->HP, &ts->dX0, &ts->dXtop, &ts->dXl, &ts->dYl, &ts->dOin, &ts->dcbuf2, &ts->dvaP, &ts->dbvP,
        &ts->loss, &ts->normsq, &ts->rec_cnt, &ts->dcalt};
    for (DevBuf* b : bufs) b->release();
    for (auto& b : ts->O) b.release();
    for (auto& b : ts->DO) b.release();
    for (auto& b : ts->XD) b.release();
    delete ts;
    m->train = nullptr;
    return 0;
}

// Refresh the layouts derived from the master tensors (after set-up and after every update).
static void refresh_derived(casv_model* m) {
    TrainState* ts = m->train;
    const int W = m->W, V = m->V, Vp = m->Vp, C = m->C;
    for (auto& l : ts->layers) {
        launch_transpose(ts->W_(l.iwx), 4 * W, l.kx, l.kx, l.wxT.as<float>(), 4 * W, m->stream);
        launch_transpose(ts->W_(l.iwr), 4 * W, l.kr, l.kr, l.wrT.as<float>(), 4 * W, m->stream);
    }
    launch_transpose(ts->W_(ts->iE), V, W, W, ts->ETp.as<float>(), Vp, m->stream);
    launch_transpose(ts->W_(ts->iWaT), W, W, W, ts->WaN.as<float>(), W, m->stream);
    launch_transpose(ts->W_(ts->iUT), W, C, C, ts->UaN.as<float>(), W, m->stream);
    for (auto& b : ts->bridge) launch_transpose(ts->W_(b.ikt), W, W, W, b.kn.as<float>(), W, m->stream);
}

extern "C" int casv_train_begin(casv_model* m, const casv_adam_params* ap, const char* frozen_csv) {
    if (!m || !ap) return fail(CASV_ERR_ARG, "null argument");
    if (m->W > 1024) return fail(CASV_ERR_ARG, "training supports width <= 1024");
    HIPCHK(hipSetDevice(m->device));
    for (auto& kv : m->expect)
        if (!m->host.count(kv.first)) return fail(CASV_ERR_STATE, "weight '%s' has not been set", kv.first.c_str());
    casv_train_release(m);
    TrainState* ts = new TrainState();
    m->train = ts;
    ts->ap = *ap;
    const int W = m->W, D = m->D, C = m->C, Vp = m->Vp;
    const std::string fz = frozen_csv ? frozen_csv : "";
    auto add_lstm = [&](const std::string& prefix, int kx, int kctx, bool reverse) -> int {
        std::vector<float> wx, wr, bias;
        pack_train_lstm(W, kx, kctx, m->host[prefix + "_K"], m->host[prefix + "_R"], m->host[prefix + "_b"], wx, wr, bias);
        TLayer l;
        l.name = prefix; l.kx = kx; l.kr = kctx + W; l.reverse = reverse;
        const bool frozen = is_frozen(prefix + "_", fz);
        l.iwx = add_tensor(ts, prefix + "_Wx", wx, frozen);
        l.iwr = add_tensor(ts, prefix + "_Wr", wr, frozen);
        l.ib = add_tensor(ts, prefix + "_b", bias, frozen);
        if (l.iwx < 0 || l.iwr < 0 || l.ib < 0) return -1;
        if (l.wxT.ensure((size_t)kx * 4 * W * 4) || l.wrT.ensure((size_t)l.kr * 4 * W * 4)) return -1;
        ts->layers.push_back(std::move(l));
        return 0;
    };
    ts->iE = add_tensor(ts, "E", m->host["E"], false);
    int rc = add_lstm("enc1_fw", W, 0, false) | add_lstm("enc1_bw", W, 0, true);
    const bool deep_b = m->cfg.deep_bidirectional_encoder != 0;
    for (int n = 2; n <= D && !deep_b; ++n) rc |= add_lstm("enc" + std::to_string(n), n == 2 ? 2 * W : W, 0, false);
    for (int n = 2; n <= D && deep_b; ++n)      // every layer bidirectional, 2W-wide inputs (seq2seq.py:273-276): [.. encN_fw, encN_bw ..]
        rc |= add_lstm("enc" + std::to_string(n) + "_fw", 2 * W, 0, false) | add_lstm("enc" + std::to_string(n) + "_bw", 2 * W, 0, true);
    for (int n = 1; n < D; ++n) rc |= add_lstm("dec" + std::to_string(n), W, 0, false);
    rc |= add_lstm("dec" + std::to_string(D), W, C, false);
    std::vector<float> ut((size_t)W * C), wat((size_t)W * W);
    const auto& Uk = m->host["att_U"]; const auto& Wa = m->host["att_Wa"];
    for (int j = 0; j < W; ++j) for (int c = 0; c < C; ++c) ut[(size_t)j * C + c] = Uk[(size_t)c * W + j];
    for (int j = 0; j < W; ++j) for (int k = 0; k < W; ++k) wat[(size_t)j * W + k] = Wa[(size_t)k * W + j];
    ts->iUT = add_tensor(ts, "att_UT", ut, false);
    ts->iWaT = add_tensor(ts, "att_WaT", wat, false);
    ts->ibUW = add_tensor(ts, "att_bUW", m->host["att_bUW"], false);
    ts->iva = add_tensor(ts, "att_va", m->host["att_va"], false);
    ts->ibv = add_tensor(ts, "att_bv", m->host["att_bv"], false);
    if (m->cfg.bridge_dense)
        for (int n = 1; n <= D; ++n)
            for (int s = 0; s < 2; ++s) {
                const std::string b = "bridge" + std::to_string(n) + (s ? "_c" : "_h");
                const auto& K = m->host[b + "_K"];
                std::vector<float> kt((size_t)W * W);
                for (int j = 0; j < W; ++j) for (int k = 0; k < W; ++k) kt[(size_t)j * W + k] = K[(size_t)k * W + j];
                TrainState::Bridge br;
                const bool frozen = is_frozen(b + "_", fz);
                br.ikt = add_tensor(ts, b + "_KT", kt, frozen);
                br.ib = add_tensor(ts, b + "_b", m->host[b + "_b"], frozen);
                if (br.ikt < 0 || br.ib < 0 || br.kn.ensure((size_t)W * W * 4)) rc |= -1;
                ts->bridge.push_back(std::move(br));
            }
    if (rc || ts->iE < 0 || ts->iUT < 0 || ts->iWaT < 0 || ts->ibUW < 0 || ts->iva < 0 || ts->ibv < 0) {
        casv_train_release(m);
        return fail(CASV_ERR_NOMEM, "could not allocate the training tensors");
    }
    if (ts->ETp.ensure((size_t)W * Vp * 4) || ts->WaN.ensure((size_t)W * W * 4) || ts->UaN.ensure((size_t)C * W * 4) ||
        ts->loss.ensure(16) || ts->normsq.ensure(16)) { casv_train_release(m); return fail(CASV_ERR_NOMEM, "out of memory"); }
    HIPCHK(hipMemset(ts->ETp.p, 0, (size_t)W * Vp * 4));
    ts->O.resize(D + 1); ts->DO.resize(D + 1); ts->XD.resize(D + 1);
    refresh_derived(m);
    HIPCHK(hipStreamSynchronize(m->stream));
    return CASV_OK;
}

static GemmArgs plain_gemm(const float* A, long long lda, int M, int K, const float* Bt, int N, const float* bias, float* C_, long long ldc,
                           int accumulate = 0) {
    GemmArgs g{};
    g.nseg = 1; g.a[0] = mkseg(A, (int)lda, K, 0);
    g.Bt = Bt; g.bias = bias; g.M = M; g.N = N; g.Ktot = K;
    g.out = mkslot(C_, (int)ldc); g.accumulate = accumulate; g.ksplit = -1; g.kgroups = 2;
    return g;
}

// A plain contraction over the whole sequence (thousands of rows, nothing fused): through the vendor library when that is switched
// on and has a solution for the shape (vendor_gemm.hip), else gemm.hip.
static void run_plain(casv_model* m, GemmArgs& g) {
#ifdef CASV_GEMM_PROF_PLAIN
    {   // diagnostic build: stamps of every plain whole-sequence contraction, one line per launch (M, N, K)
        (void)hipStreamSynchronize(m->stream);
        casv::gemm_prof_dump();
        fprintf(stderr, "plain gemm M=%d N=%d K=%d accumulate=%d\n", g.M, g.N, g.Ktot, g.accumulate);
    }
#endif
    if (m->vendor_gemm && g.nseg == 1 && !g.a[0].rows && !g.step_ptr && g.M >= 4096 && g.a[0].width == g.Ktot && g.a[0].koff == 0) {
        hipEvent_t ev{};
        m->prof_begin(PC_GEMM, 2.0 * g.M * (double)g.N * g.Ktot, 4.0 * ((double)g.M * g.Ktot + (double)g.N * g.Ktot + (double)g.M * g.N), ev);
        const bool done = vendor_gemm_nt(g.a[0].base, g.a[0].ld, g.M, g.Ktot, g.Bt, g.N, g.bias, g.out.base, g.out.ld, g.accumulate, m->stream);
        m->prof_end(PC_GEMM, ev);
        if (done) return;
    }
    run_gemm(m, EPI_PLAIN, g);
}

// C[M][N] += A^T . B for operands as the forward / backward passes left them: A [K][lda], B [K][ldb] (gemm_tn.hip)
static void run_gemm_tn(casv_model* m, const float* A, long long lda, int M, int Mstore, const float* B, long long ldb, int N, long long K,
                        float* C, long long ldc, float* colsum = nullptr) {
    TnArgs g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.Mstore = Mstore; g.N = N; g.K = (int)K; g.accumulate = 1;
    g.colsum = colsum;
    hipEvent_t ev{};
    m->prof_begin(PC_GEMM, 2.0 * Mstore * (double)N * (double)K, 4.0 * ((double)K * (M + N) + (double)Mstore * N), ev);
    // (the step's arithmetic, engine.h ENTRY_TRAIN: the big weight gradients on bf16x3-split operands where the shape has that form)
    if (!(gemm_split_bf16() && launch_gemm_tn_split(g, m->stream))) launch_gemm_tn(g, m->stream);
    m->prof_end(PC_GEMM, ev);
}

// One recurrent step of a layer as a GEMM job; k = processing index (time t = k, or len-1-k when reversed).
static GemmArgs layer_step_job(casv_model* m, TLayer& l, int k, const float* h0, const float* c0, const float* rec_in /*top cell*/) {
    TrainState* ts = m->train;
    const int W = m->W, B = ts->B;
    const int mul = l.reverse ? -1 : 1, add_t = l.reverse ? l.len - 1 : 0, add_p = l.reverse ? l.len : -1;
    GemmArgs g{};
    g.nseg = 1;
    if (rec_in) g.a[0] = mkseg(rec_in, l.kr, l.kr, 0, nullptr, (long long)B * l.kr, 1, 0);
    else {
        g.a[0] = mkseg(l.hs, (int)l.hs_ld, W, 0, nullptr, (long long)B * l.hs_ld, mul, add_p, 1);
        g.a[0].first_base = h0;
    }
    g.Bt = ts->W_(l.iwr); g.bias = nullptr; g.M = B; g.N = 4 * W; g.Ktot = l.kr;
    g.zinit = mkslot(l.Z.as<float>(), 4 * W, (long long)B * 4 * W, mul, add_t);
    g.out = mkslot(l.hs, (int)l.hs_ld, (long long)B * l.hs_ld, mul, add_t);
    g.c_in = mkseg(l.Cs.as<float>(), W, W, 0, nullptr, (long long)B * W, mul, add_p, 1);
    g.c_in.first_base = c0;
    g.c_out = mkslot(l.Cs.as<float>(), W, (long long)B * W, mul, add_t);
    g.gates_out = mkslot(l.Gt.as<float>(), 4 * W, (long long)B * 4 * W, mul, add_t);
    g.step_imm = k; g.step_ptr = nullptr; g.kgroups = 2;
    return g;
}

static int time_of(const TLayer& l, int k) { return l.reverse ? l.len - 1 - k : k; }

// Z = x . Wx^T + b for every time step (one GEMM)
static void layer_input_gemm(casv_model* m, TLayer& l, const float* x, long long ldx) {
    TrainState* ts = m->train;
    GemmArgs g = plain_gemm(x, ldx, l.len * ts->B, l.kx, ts->W_(l.iwx), 4 * m->W, ts->W_(l.ib), l.Z.as<float>(), 4 * m->W);
    run_plain(m, g);
}

// weight gradients of one layer from dZ (in l.Z), its inputs x and its recurrent-side inputs rec [len*B][kr]: both
// contractions run over the rows = time x batch, on the operands as they lie
static int layer_weight_grads(casv_model* m, TLayer& l, const float* x, long long ldx, const float* rec, long long ldrec) {
    TrainState* ts = m->train;
    const int W = m->W;
    const long long rows = (long long)l.len * ts->B;
    if (ts->tens[l.iwx].frozen) return 0;
    // (the bias gradient = column sums of dZ rides in the first of the two launches)
    run_gemm_tn(m, l.Z.as<float>(), 4 * W, 4 * W, 4 * W, x, ldx, l.kx, rows, ts->G_(l.iwx), l.kx, ts->G_(l.ib));
    run_gemm_tn(m, l.Z.as<float>(), 4 * W, 4 * W, 4 * W, rec, ldrec, l.kr, rows, ts->G_(l.iwr), l.kr);
    return 0;
}

// Backward of plain LSTM layers.  dOut (+mask) = gradient w.r.t. the layer's output sequence; dh_fin/dc_fin w.r.t. its
// final state; h0/c0 its initial state (nullptr = zero).  Leaves dL/dh0 in dRec slot(first step) and dL/dc0 in dc.
struct LayerBwd {
    TLayer* l;
    const float* dOut; long long ld_out; const float* mask;
    const float* dh_fin; const float* dc_fin; const float* h0; const float* c0; float* dc;
    const float* x; long long ldx; float* dX; long long ld_dx; int dx_accumulate;
};

// pointwise part of processing index k (gate derivatives -> dZ)
static LstmBwdArgs layer_backward_pointwise(casv_model* m, const LayerBwd& a, int k) {
    TrainState* ts = m->train;
    TLayer& l = *a.l;
    const int W = m->W, B = ts->B;
    const int t = time_of(l, k);
    LstmBwdArgs p{};
    p.a = a.dOut ? a.dOut + (long long)t * B * a.ld_out : nullptr; p.lda = a.ld_out; p.mask_a = a.mask;
    if (k < l.len - 1) { p.b = l.dRec.as<float>() + (long long)time_of(l, k + 1) * B * W; p.ldb = W; }
    else if (a.dh_fin) { p.b = a.dh_fin; p.ldb = W; }
    p.gates = l.Gt.as<float>() + (long long)t * B * 4 * W;
    p.cell = l.Cs.as<float>() + (long long)t * B * W;
    if (k > 0) { p.c_prev = l.Cs.as<float>() + (long long)time_of(l, k - 1) * B * W; p.ld_cprev = W; }
    else { p.c_prev = a.c0; p.ld_cprev = W; }
    p.dc = a.dc; p.dz = l.Z.as<float>() + (long long)t * B * 4 * W; p.rows = B; p.W = W;
    return p;
}
// ... and the data GEMM that carries dZ to the previous step
static GemmArgs layer_backward_gemm(casv_model* m, const LayerBwd& a, int k) {
    TrainState* ts = m->train;
    TLayer& l = *a.l;
    const int W = m->W, B = ts->B;
    const int t = time_of(l, k);
    GemmArgs g = plain_gemm(l.Z.as<float>() + (long long)t * B * 4 * W, 4 * W, B, 4 * W, l.wrT.as<float>(), l.kr, nullptr,
                            l.dRec.as<float>() + (long long)t * B * W, W);
    g.out_zeroed = 1;           // casv_train_step clears dRec once per step
    return g;
}

static int layer_backward_finish(casv_model* m, const LayerBwd& a) {
    TrainState* ts = m->train;
    TLayer& l = *a.l;
    const int W = m->W, B = ts->B;
    const long long rows = (long long)l.len * B;
    if (a.dX) {
        GemmArgs g = plain_gemm(l.Z.as<float>(), 4 * W, (int)rows, 4 * W, l.wxT.as<float>(), l.kx, nullptr, a.dX, a.ld_dx, a.dx_accumulate);
        run_plain(m, g);
    }
    // The recurrent-side input of a step is the h of the previously processed step (h0 / zero at the first): the layer's own
    // outputs, one time step apart from dZ -- contracted as they lie (gemm_tn takes a row offset and a row stride; the first step's
    // B rows against h0 as a launch of their own), not copied into a shifted image first (105 MB per layer at configs[3]).
    if (ts->tens[l.iwx].frozen) return 0;
    const float* dZ = l.Z.as<float>();
    // (the bias gradient = column sums of dZ rides in the launch over the inputs)
    run_gemm_tn(m, dZ, 4 * W, 4 * W, 4 * W, a.x, a.ldx, l.kx, rows, ts->G_(l.iwx), l.kx, ts->G_(l.ib));
    const long long rest = (long long)(l.len - 1) * B;         // rows whose previous step is a row of hs
    if (rest > 0) {
        // forward layer: dZ[1..] with H[0..len-2]; reversed layer: dZ[0..len-2] with H[1..]
        const float* dz = dZ + (l.reverse ? 0 : (long long)B * 4 * W);
        const float* hp = l.hs + (l.reverse ? (long long)B * l.hs_ld : 0);
        run_gemm_tn(m, dz, 4 * W, 4 * W, 4 * W, hp, l.hs_ld, W, rest, ts->G_(l.iwr), l.kr);
    }
    if (a.h0) run_gemm_tn(m, dZ + (long long)time_of(l, 0) * B * 4 * W, 4 * W, 4 * W, 4 * W, a.h0, W, W, B, ts->G_(l.iwr), l.kr);
    return 0;
}

// Up to two independent layers walk their sequences backwards in lockstep: one pointwise launch each and ONE data-GEMM
// launch per step for both (the per-step launches are latency-bound, so pairing them is nearly free).
static int layers_backward(casv_model* m, const LayerBwd* a, int count) {
    TrainState* ts = m->train;
    const int W = m->W, B = ts->B;
    int maxlen = 0;
    for (int j = 0; j < count; ++j) {
        if (a[j].dc_fin) HIPCHK(hipMemcpyAsync(a[j].dc, a[j].dc_fin, (size_t)B * W * 4, hipMemcpyDeviceToDevice, m->stream));
        else HIPCHK(hipMemsetAsync(a[j].dc, 0, (size_t)B * W * 4, m->stream));
        maxlen = std::max(maxlen, a[j].l->len);
    }
    bool persistent = false;
    if (m->persist_mode != 0 && ts->rec_skip == 0 && ts->rec_launches < 16 && m->ncu >= 64) {
        RecBwdArgs ra{};
        ra.njobs = count; ra.B = B; ra.W = W;
        bool plain = true;
        for (int j = 0; j < count; ++j) {
            TLayer& l = *a[j].l;
            plain = plain && l.kr == W;
            ra.job[j] = RecBwdJob{l.wrT.as<float>(), a[j].dOut, a[j].ld_out, a[j].mask, a[j].dh_fin, a[j].dc_fin, l.Gt.as<float>(), l.Cs.as<float>(), a[j].c0,
                                  l.Z.as<float>(), l.dRec.as<float>(), a[j].dc, l.len, l.reverse ? 1 : 0};
        }
        const size_t cb = train_recurrence_bwd_counter_bytes(B);
        ra.counters = reinterpret_cast<unsigned*>(static_cast<char*>(ts->rec_cnt.p) + cb * ts->rec_launches);
        const int grid = plain ? train_recurrence_bwd_grid(ra, m->ncu) : 0;
        if (grid) {
            hipEvent_t ev{};
            double flops = 0;
            for (int j = 0; j < count; ++j) flops += 2.0 * B * 4.0 * W * W * a[j].l->len;
            m->prof_begin(PC_PERSIST, flops, 0.0, ev);
            launch_train_recurrence_bwd(ra, grid, m->stream);
            m->prof_end(PC_PERSIST, ev);
            ts->rec_abort[ts->rec_launches++] = ra.counters + (cb / sizeof(unsigned) - 32);
            persistent = true;
        }
    }
    if (persistent) {
    } else if (m->fused_backward) {
        // one launch per step for both layers: the cells' backward inside the data GEMM (gemm_bwd.hip); dL/dc ping-pongs
        float* dcb[2][2];
        int done[2] = {0, 0};
        for (int j = 0; j < count; ++j) { dcb[j][0] = a[j].dc; dcb[j][1] = ts->dcalt.as<float>() + (size_t)j * B * W; }
        for (int i = 0; i < maxlen; ++i) {
            BwdStepBatch fb{};
            double flops = 0, bytes = 0;
            for (int j = 0; j < count; ++j) {
                const int k = a[j].l->len - 1 - i;
                if (k < 0) continue;
                BwdStepJob& q = fb.j[fb.count++];
                q.p = layer_backward_pointwise(m, a[j], k);
                q.dc_in = dcb[j][done[j] & 1]; q.p.dc = dcb[j][(done[j] + 1) & 1];
                ++done[j];
                const GemmArgs g = layer_backward_gemm(m, a[j], k);
                q.Bt = g.Bt; q.out = g.out.base; q.ld_out = g.out.ld; q.N = g.N;
                flops += 2.0 * B * (double)g.N * 4.0 * W; bytes += 4.0 * ((double)B * 4 * W + (double)g.N * 4 * W + (double)B * g.N);
            }
            hipEvent_t ev{};
            m->prof_begin(PC_GEMM, flops, bytes, ev);
            launch_lstm_bwd_gemm(fb, m->stream);
            m->prof_end(PC_GEMM, ev);
        }
        for (int j = 0; j < count; ++j)
            if (done[j] & 1) HIPCHK(hipMemcpyAsync(a[j].dc, dcb[j][1], (size_t)B * W * 4, hipMemcpyDeviceToDevice, m->stream));
    } else
    for (int i = 0; i < maxlen; ++i) {
        GemmBatch b{};
        LstmBwdBatch pw{};
        for (int j = 0; j < count; ++j) {
            const int k = a[j].l->len - 1 - i;
            if (k < 0) continue;
            pw.a[pw.count++] = layer_backward_pointwise(m, a[j], k);
            b.g[b.count++] = layer_backward_gemm(m, a[j], k);
        }
        launch_lstm_bwd_batch(pw, m->stream);
        run_gemm_batch(m, EPI_PLAIN, b);
    }
    for (int j = 0; j < count; ++j)
        if (int rc = layer_backward_finish(m, a[j])) return rc;
    return 0;
}

// Forward recurrence of up to two independent plain layers: ONE persistent launch (train_persist.hip) where the shape has
// one and the device is ours, else one launch per step for both (the two are interchangeable bit for bit).
// om / om_ld / omask: the layer's outputs once more, times the next layer's per-unit dropout mask (nullptr: times one) -- written by
// the persistent recurrence itself; *masked tells the caller whether it was (the per-step launches leave it to launch_mul_mask).
struct LayerFwd { TLayer* l; const float* h0; const float* c0; float* om = nullptr; long long om_ld = 0; const float* omask = nullptr; };
static int layers_forward(casv_model* m, const LayerFwd* a, int count, bool* masked = nullptr) {
    if (masked) *masked = false;
    TrainState* ts = m->train;
    const int W = m->W, B = ts->B;
    int maxlen = 0;
    for (int j = 0; j < count; ++j) maxlen = std::max(maxlen, a[j].l->len);
    if (m->persist_mode != 0 && ts->rec_skip == 0 && ts->rec_launches < 16 && m->ncu >= 64) {
        RecArgs ra{};
        ra.njobs = count; ra.B = B; ra.W = W;
        for (int j = 0; j < count; ++j) {
            TLayer& l = *a[j].l;
            ra.job[j] = RecJob{ts->W_(l.iwr), l.Z.as<float>(), l.hs, l.hs_ld, l.Cs.as<float>(), l.Gt.as<float>(), a[j].h0, a[j].c0, l.len,
                               l.reverse ? 1 : 0, masked ? a[j].om : nullptr, a[j].om_ld, a[j].omask, l.kr == W ? l.dRec.as<float>() : nullptr};
        }
        const size_t cb = train_recurrence_bwd_counter_bytes(B);        // (one slot size for both kinds of launch)
        ra.counters = reinterpret_cast<unsigned*>(static_cast<char*>(ts->rec_cnt.p) + cb * ts->rec_launches);
        ra.fault = m->persist_mode == 2;       // (test of the give-up path)
        if (const int grid = train_recurrence_grid(ra, m->ncu)) {
            hipEvent_t ev{};
            double flops = 0;
            for (int j = 0; j < count; ++j) flops += 2.0 * B * 4.0 * W * W * a[j].l->len;
            m->prof_begin(PC_PERSIST, flops, 0.0, ev);
            launch_train_recurrence(ra, grid, m->stream);
            m->prof_end(PC_PERSIST, ev);
            ts->rec_abort[ts->rec_launches++] = ra.counters + (train_recurrence_counter_bytes(B) / sizeof(unsigned) - 32);
            if (masked) *masked = true;
            for (int j = 0; j < count; ++j) a[j].l->drec_cleared = a[j].l->kr == W;
            return 0;
        }
    }
    for (int k = 0; k < maxlen; ++k) {
        GemmBatch b{};
        for (int j = 0; j < count; ++j)
            if (k < a[j].l->len) b.g[b.count++] = layer_step_job(m, *a[j].l, k, a[j].h0, a[j].c0, nullptr);
        run_gemm_batch(m, EPI_LSTM, b);
    }
    return 0;
}

extern "C" int casv_train_step(casv_model* m, int32_t mode, int32_t B, int32_t T, int32_t U, int32_t A,
                               const int32_t* enc_idx, const float* enc_val, const int32_t* dec_in, const int32_t* dec_out,
                               const float* weights, const float* mask_enc, const float* mask_dec, const float* mask_cell,
                               double* loss_out, double* norm_out) {
    if (!m || !enc_idx || !dec_in || !dec_out || !weights || !loss_out) return fail(CASV_ERR_ARG, "null argument");
    if (!m->train) return fail(CASV_ERR_STATE, "casv_train_begin must run first");
    if (B < 1 || T < 1 || U < 1 || A < 1) return fail(CASV_ERR_ARG, "bad shape");
    if (mode < 0 || mode > 2) return fail(CASV_ERR_ARG, "mode must be 0 (evaluate), 1 (train) or 2 (gradients only)");
    HIPCHK(hipSetDevice(m->device));
    SplitScope arithmetic(arithmetic_of(m, ENTRY_TRAIN));       // (engine.h: the whole-sequence contractions with a split form take the bf16x3-split arithmetic)
    TrainState* ts = m->train;
    hipStream_t st = m->stream;
    const int W = m->W, V = m->V, Vp = m->Vp, C = m->C, D = m->D;
    const long long TB = (long long)T * B, UB = (long long)U * B;
    ts->B = B; ts->T = T; ts->U = U; ts->A = A;
    m->encoded = false;                    // the final-state buffers are shared with the inference session
    const bool training = mode != 0;

    // ---- buffers ----
#define ENS(buf, bytes) if (int rc_ = (buf).ensure(bytes)) return rc_;
    ENS(ts->e_idx, TB * A * 4) ENS(ts->e_val, TB * A * 4) ENS(ts->d_in, UB * 4) ENS(ts->d_out, UB * 4) ENS(ts->d_w, UB * 4)
    const bool deep = m->cfg.deep_bidirectional_encoder != 0 && D >= 2;
    ENS(ts->m_enc, (size_t)std::max(D + 1, 2 * D) * W * 4) ENS(ts->m_dec, (size_t)D * W * 4) ENS(ts->m_cell, (size_t)B * (W + C) * 4)
    ENS(ts->X0, TB * W * 4) ENS(ts->H1, TB * 2 * W * 4) ENS(ts->u, TB * W * 4) ENS(ts->Y0, UB * W * 4) ENS(ts->Ym, UB * W * 4)
    ENS(ts->WQ, UB * W * 4) ENS(ts->Ast, (size_t)(U + 1) * B * T * 4) ENS(ts->WIN, UB * 4)
    ENS(ts->RecIn, UB * (C + W) * 4) ENS(ts->prev, (size_t)B * 4) ENS(ts->logits, UB * Vp * 4)
    ENS(ts->dG, UB * W * 4) ENS(ts->d_enc, TB * C * 4) ENS(ts->du, TB * W * 4) ENS(ts->DWQ, UB * W * 4) ENS(ts->DSrows, UB * 16 * 4) ENS(ts->dhatt, UB * W * 4)
    ENS(ts->dfin, (size_t)2 * D * B * W * 4) ENS(ts->dcbuf, (size_t)B * W * 4)
    ENS(ts->dX0, TB * W * 4) ENS(ts->dXtop, UB * W * 4) ENS(ts->dXl, TB * 2 * W * 4) ENS(ts->dYl, UB * W * 4) ENS(ts->dOin, TB * 2 * W * 4)
    ENS(ts->dcbuf2, (size_t)B * W * 4) ENS(ts->dvaP, (size_t)B * W * 4) ENS(ts->dbvP, (size_t)B * 4)
    for (auto& l : ts->layers) {
        const bool enc = l.name.compare(0, 3, "enc") == 0;
        l.len = enc ? T : U;
        const long long rows = (long long)l.len * B;
        ENS(l.Cs, rows * W * 4) ENS(l.Gt, rows * 4 * W * 4) ENS(l.Z, rows * 4 * W * 4) ENS(l.dRec, rows * l.kr * 4)
        if (l.name == "enc1_fw") { l.hs = ts->H1.as<float>(); l.hs_ld = 2 * W; }
        else if (l.name == "enc1_bw") { l.hs = ts->H1.as<float>() + W; l.hs_ld = 2 * W; }
        else if (deep && enc && l.name.size() > 3 && l.name.compare(l.name.size() - 3, 3, "_fw") == 0) { ENS(l.Hown, rows * 2 * W * 4) l.hs = l.Hown.as<float>(); l.hs_ld = 2 * W; }
        else if (deep && enc) { l.hs = (&l - 1)->hs + W; l.hs_ld = 2 * W; }        // (a backward direction: the second half of its forward partner's rows)
        else { ENS(l.Hown, rows * W * 4) l.hs = l.Hown.as<float>(); l.hs_ld = W; }
    }
    ENS(ts->rec_cnt, 16 * train_recurrence_bwd_counter_bytes(B)) ENS(ts->dcalt, (size_t)2 * B * W * 4)
    const bool residual = m->cfg.residual_connections != 0, bridged = m->cfg.bridge_dense != 0;
    if (bridged) { ENS(ts->hbr, (size_t)D * B * W * 4) ENS(ts->cbr, (size_t)D * B * W * 4) ENS(ts->brtmp, (size_t)B * W * 4) }
    if (residual && D >= 2) ENS(ts->Ytop, UB * W * 4)
    for (int n = 1; n <= D; ++n) ENS(ts->O[n], TB * ((n == 1 || deep) ? 2 * W : W) * 4)
    for (int n = 2; n <= D && deep; ++n) ENS(ts->XD[n], TB * 2 * W * 4)
    for (int n = 1; n < D; ++n) ENS(ts->DO[n], UB * W * 4)
#undef ENS
    // ---- inputs ----
    HIPCHK(hipMemcpyAsync(ts->e_idx.p, enc_idx, TB * A * 4, hipMemcpyHostToDevice, st));
    if (enc_val) HIPCHK(hipMemcpyAsync(ts->e_val.p, enc_val, TB * A * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ts->d_in.p, dec_in, UB * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ts->d_out.p, dec_out, UB * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ts->d_w.p, weights, UB * 4, hipMemcpyHostToDevice, st));
    // masks: enc = 2W + (D-1)*W floats, dec = (D-1)*W floats, cell = B*(W+C)
    const float* menc = nullptr; const float* mdec = nullptr; const float* mcell = nullptr;
    if (mask_enc) { HIPCHK(hipMemcpyAsync(ts->m_enc.p, mask_enc, (size_t)(deep ? 2 * D : D + 1) * W * 4, hipMemcpyHostToDevice, st)); menc = ts->m_enc.as<float>(); }
    if (mask_dec && D > 1) { HIPCHK(hipMemcpyAsync(ts->m_dec.p, mask_dec, (size_t)(D - 1) * W * 4, hipMemcpyHostToDevice, st)); mdec = ts->m_dec.as<float>(); }
    if (mask_cell) { HIPCHK(hipMemcpyAsync(ts->m_cell.p, mask_cell, (size_t)B * (W + C) * 4, hipMemcpyHostToDevice, st)); mcell = ts->m_cell.as<float>(); }
    auto menc_n = [&](int n) { return menc ? menc + (n == 1 ? 0 : deep ? (n - 1) * 2 * W : 2 * W + (n - 2) * W) : nullptr; };   // layer n (1-based)
    auto mdec_n = [&](int n) { return mdec ? mdec + (n - 1) * W : nullptr; };
    long cnt = 0;
    for (long long i = 0; i < UB; ++i) cnt += weights[i] != 0.f;
    const float inv_count = 1.0f / (float)std::max(cnt, 1L);
    HIPCHK(hipMemsetAsync(ts->loss.p, 0, 16, st));
    HIPCHK(hipMemsetAsync(ts->normsq.p, 0, 16, st));
    HIPCHK(hipMemsetAsync(ts->rec_cnt.p, 0, 16 * train_recurrence_bwd_counter_bytes(B), st));
    ts->rec_launches = ts->rec_checked = 0; ts->split_launch = -1;
    for (auto& l : ts->layers) l.drec_cleared = false;
    if (ts->rec_skip > 0) --ts->rec_skip;
    if (training) for (auto& t : ts->tens) HIPCHK(hipMemsetAsync(t.g.p, 0, t.n * 4, st));

    TLayer* Lfw = &ts->layers[0]; TLayer* Lbw = &ts->layers[1];
    auto enc_layer = [&](int n) -> TLayer& { return ts->layers[n]; };            // n >= 2 -> index n (not with a deep bidirectional encoder)
    auto enc_dir = [&](int n, int dir) -> TLayer& { return ts->layers[2 * (n - 1) + dir]; };      // deep: layer n = 1..D, direction 0 fw / 1 bw
    auto dec_layer = [&](int n) -> TLayer& { return ts->layers[(deep ? 2 * D - 1 : D) + n]; };    // n = 1..D
    float* hfin = m->hfin.as<float>(); float* cfin = m->cfin.as<float>();
    if (int rc = m->hfin.ensure((size_t)D * B * W * 4)) return rc;
    if (int rc = m->cfin.ensure((size_t)(D + 1) * B * W * 4)) return rc;
    hfin = m->hfin.as<float>(); cfin = m->cfin.as<float>();

    // ================= forward: encoder =================
    launch_embed_tm(ts->W_(ts->iE), ts->e_idx.as<int>(), enc_val ? ts->e_val.as<float>() : nullptr, ts->X0.as<float>(), B, T, A, V, W, st);
    layer_input_gemm(m, *Lfw, ts->X0.as<float>(), W);
    layer_input_gemm(m, *Lbw, ts->X0.as<float>(), W);
    bool masked1 = false;
    {
        const LayerFwd f[2] = {{Lfw, nullptr, nullptr, ts->O[1].as<float>(), 2 * W, menc_n(1)},
                               {Lbw, nullptr, nullptr, ts->O[1].as<float>() + W, 2 * W, menc_n(1) ? menc_n(1) + W : nullptr}};
        if (int rc = layers_forward(m, f, 2, &masked1)) return rc;
    }
    // final states handed to the decoder: layer 1 = backward direction after t = 0 (seq2seq.py:280)
    HIPCHK(hipMemcpy2DAsync(hfin, (size_t)W * 4, Lbw->hs, (size_t)2 * W * 4, (size_t)W * 4, B, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(cfin, Lbw->Cs.p, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
    // bridge_dense (seq2seq.py:299-301): the decoder starts from tanh(state . K + b) of every encoder layer's final h and c
    const float* h0base = bridged ? ts->hbr.as<float>() : hfin; const float* c0base = bridged ? ts->cbr.as<float>() : cfin;
    auto bridge_forward = [&](int n) {              // encoder layer n (1-based): hfin / cfin slot n - 1 -> hbr / cbr slot n - 1
        if (!bridged) return;
        for (int s_ = 0; s_ < 2; ++s_) {
            const TrainState::Bridge& br = ts->bridge[2 * (n - 1) + s_];
            const float* src = (s_ ? cfin : hfin) + (size_t)(n - 1) * B * W;
            float* dst = (s_ ? ts->cbr.as<float>() : ts->hbr.as<float>()) + (size_t)(n - 1) * B * W;
            GemmArgs g = plain_gemm(src, W, B, W, ts->W_(br.ikt), W, ts->W_(br.ib), ts->brtmp.as<float>(), W);
            g.ksplit = 0; g.kgroups = 0;            // (one k-ordered chain per element, no atomics into an uncleared buffer)
            run_gemm(m, EPI_PLAIN, g);
            launch_tanh(ts->brtmp.as<float>(), dst, (long long)B * W, st);
        }
    };
    bridge_forward(1);
    if (!masked1) launch_mul_mask(ts->H1.as<float>(), 2 * W, menc_n(1), ts->O[1].as<float>(), 2 * W, TB, 2 * W, st);
    launch_embed_tm(ts->W_(ts->iE), ts->d_in.as<int>(), nullptr, ts->Y0.as<float>(), B, U, 1, V, W, st);
    // Encoder layer n and decoder layer n-1 depend only on encoder layer n-1 / decoder layer n-2, so the two
    // recurrences advance in lockstep, one launch per step for both.
    const float* y = ts->Y0.as<float>();
    // deep_bidirectional_encoder (seq2seq.py:246-281): every encoder layer n >= 2 is a BiLSTM on the cross sum of the (dropped-out) layer
    // below; its two directions are the pair of a launch, the decoder layer n - 1 walks alone
    for (int n = 2; n <= D && deep; ++n) {
        TLayer& lf = enc_dir(n, 0); TLayer& lb = enc_dir(n, 1); TLayer& ld = dec_layer(n - 1);
        launch_cross_sum(ts->O[n - 1].as<float>(), ts->XD[n].as<float>(), TB * 2 * W, st);
        layer_input_gemm(m, lf, ts->XD[n].as<float>(), 2 * W);
        layer_input_gemm(m, lb, ts->XD[n].as<float>(), 2 * W);
        layer_input_gemm(m, ld, y, W);
        bool maskede = false, maskedd = false;
        {
            const LayerFwd f[2] = {{&lf, nullptr, nullptr, ts->O[n].as<float>(), 2 * W, menc_n(n)},
                                   {&lb, nullptr, nullptr, ts->O[n].as<float>() + W, 2 * W, menc_n(n) ? menc_n(n) + W : nullptr}};
            if (int rc = layers_forward(m, f, 2, &maskede)) return rc;
        }
        HIPCHK(hipMemcpy2DAsync(hfin + (size_t)(n - 1) * B * W, (size_t)W * 4, lb.hs, (size_t)2 * W * 4, (size_t)W * 4, B, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(cfin + (size_t)(n - 1) * B * W, lb.Cs.p, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
        bridge_forward(n);
        if (!maskede) launch_mul_mask(lf.hs, 2 * W, menc_n(n), ts->O[n].as<float>(), 2 * W, TB, 2 * W, st);
        const float* h0 = h0base + (size_t)(n - 2) * B * W; const float* c0 = c0base + (size_t)(n - 2) * B * W;
        const bool res_n = residual && n >= 3;
        {
            const LayerFwd f[1] = {{&ld, h0, c0, res_n ? nullptr : ts->DO[n - 1].as<float>(), W, mdec_n(n - 1)}};
            if (int rc = layers_forward(m, f, 1, &maskedd)) return rc;
        }
        if (res_n) launch_add_mul_mask(ld.hs, W, y, W, mdec_n(n - 1), ts->DO[n - 1].as<float>(), W, UB, W, st);
        else if (!maskedd) launch_mul_mask(ld.hs, W, mdec_n(n - 1), ts->DO[n - 1].as<float>(), W, UB, W, st);
        y = ts->DO[n - 1].as<float>();
    }
    for (int n = 2; n <= D && !deep; ++n) {
        TLayer& le = enc_layer(n);
        TLayer& ld = dec_layer(n - 1);
        layer_input_gemm(m, le, ts->O[n - 1].as<float>(), le.kx);
        layer_input_gemm(m, ld, y, W);
        const float* h0 = h0base + (size_t)(n - 2) * B * W; const float* c0 = c0base + (size_t)(n - 2) * B * W;
        // residual_connections: encoder layer n >= 3 / decoder layer n - 1 >= 2 hand on LSTM output + input sequence (seq2seq.py:284-291,
        // 359-360) -- the sum and the next layer's dropout mask in one pass behind the recurrences (which then write no masked copy)
        const bool res_n = residual && n >= 3;
        bool maskedn = false;
        {
            const LayerFwd f[2] = {{&le, nullptr, nullptr, res_n ? nullptr : ts->O[n].as<float>(), W, menc_n(n)},
                                   {&ld, h0, c0, res_n ? nullptr : ts->DO[n - 1].as<float>(), W, mdec_n(n - 1)}};
            if (int rc = layers_forward(m, f, 2, &maskedn)) return rc;
        }
        HIPCHK(hipMemcpyAsync(hfin + (size_t)(n - 1) * B * W, le.hs + (long long)(T - 1) * B * W, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(cfin + (size_t)(n - 1) * B * W, le.Cs.as<float>() + (long long)(T - 1) * B * W, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
        bridge_forward(n);
        if (res_n) {
            launch_add_mul_mask(le.hs, W, ts->O[n - 1].as<float>(), W, menc_n(n), ts->O[n].as<float>(), W, TB, W, st);
            launch_add_mul_mask(ld.hs, W, y, W, mdec_n(n - 1), ts->DO[n - 1].as<float>(), W, UB, W, st);
        } else if (!maskedn) {
            launch_mul_mask(le.hs, W, menc_n(n), ts->O[n].as<float>(), W, TB, W, st);
            launch_mul_mask(ld.hs, W, mdec_n(n - 1), ts->DO[n - 1].as<float>(), W, UB, W, st);
        }
        y = ts->DO[n - 1].as<float>();
    }
    const float* enc_out = ts->O[D].as<float>();
    { GemmArgs g = plain_gemm(enc_out, C, (int)TB, C, ts->W_(ts->iUT), W, nullptr, ts->u.as<float>(), W); run_plain(m, g); }

    // ================= forward: attention cell =================
    TLayer& top = dec_layer(D);
    launch_mul_rowmask(y, W, mcell, W + C, ts->Ym.as<float>(), W, UB, B, W, st);
    layer_input_gemm(m, top, ts->Ym.as<float>(), W);
    const float* h0t = h0base + (size_t)(D - 1) * B * W; const float* c0t = c0base + (size_t)(D - 1) * B * W;
    HIPCHK(hipMemsetAsync(ts->Ast.p, 0, (size_t)B * T * 4, st));
    // The cell's input rows [ctx * mask | h(t-1)] (LSTMCell(dropout) masks the cell input [y | ctx] per sample, seq2seq.py:345; the y
    // part is masked where it is precomputed) are filled where their parts are produced: the attention rows write the masked
    // context straight into them, the cell of the step before stores its h a second time.
    HIPCHK(hipMemcpy2DAsync(ts->RecIn.as<float>() + C, (size_t)(C + W) * 4, h0t, (size_t)W * 4, (size_t)W * 4, B, hipMemcpyDeviceToDevice, st));
    AttnArgs att{};         // what every step's attention rows share
    att.u = ts->u.as<float>(); att.enc = enc_out; att.va = ts->W_(ts->iva); att.bv = ts->W_(ts->ibv);
    att.a_base = ts->Ast.as<float>(); att.prev = nullptr; att.line = nullptr; att.rows_per_line = 1;
    att.ctx_ld = C + W; att.ctx_mask = mcell ? mcell + W : nullptr; att.ctx_mask_ld = W + C;
    att.R = B; att.T = T; att.W = W; att.C = C; att.window = m->cfg.window_width;
    att.apos = nullptr; att.amax1 = nullptr; att.nrows = nullptr;
    att.u_line = W; att.u_time = (long long)B * W; att.enc_line = C; att.enc_time = (long long)B * C;
    bool top_persistent = false;
    if (m->persist_mode != 0 && ts->rec_skip == 0 && ts->rec_launches < 16 && m->ncu >= 64) {
        // ONE launch for the whole recurrence of the cell (train_persist_top.hip)
        TopRecArgs ra{};
        ra.Wr = ts->W_(top.iwr); ra.WaT = ts->W_(ts->iWaT); ra.bUW = ts->W_(ts->ibUW); ra.Z = top.Z.as<float>();
        ra.RecIn = ts->RecIn.as<float>(); ra.WQ = ts->WQ.as<float>(); ra.hs = top.hs; ra.Cs = top.Cs.as<float>(); ra.Gt = top.Gt.as<float>();
        ra.c0 = c0t; ra.WIN = ts->WIN.as<int>(); ra.att = att; ra.B = B; ra.U = U; ra.W = W; ra.C = C;
        const size_t cb = train_recurrence_bwd_counter_bytes(B);
        ra.counters = reinterpret_cast<unsigned*>(static_cast<char*>(ts->rec_cnt.p) + cb * ts->rec_launches);
        if (const int grid = top.hs_ld == W ? train_attention_cell_grid(ra, m->ncu) : 0) {
            hipEvent_t ev{};
            m->prof_begin(PC_PERSIST, 2.0 * B * U * ((double)4 * W * (C + W) + (double)W * W), 0.0, ev);
            launch_train_attention_cell(ra, grid, st);
            m->prof_end(PC_PERSIST, ev);
            ts->rec_abort[ts->rec_launches++] = ra.counters + (train_attention_cell_counter_bytes(B) / sizeof(unsigned) - 32);
            top_persistent = true;
        }
    }
    if (!top_persistent) HIPCHK(hipMemsetAsync(ts->WQ.p, 0, UB * W * 4, st));      // (the per-step launches' split-K partial sums land here; the persistent kernel stores its queries)
    for (int t = 0; t < U && !top_persistent; ++t) {
        const float* hprev = t == 0 ? h0t : top.hs + (long long)(t - 1) * B * W;
        float* wq = ts->WQ.as<float>() + (long long)t * B * W;
        { GemmArgs g = plain_gemm(hprev, W, B, W, ts->W_(ts->iWaT), W, ts->W_(ts->ibUW), wq, W); g.out_zeroed = 1; run_gemm(m, EPI_PLAIN, g); }
        AttnArgs a = att;
        float* rec = ts->RecIn.as<float>() + (long long)t * B * (C + W);
        a.wq = wq; a.ctx = rec; a.step_imm = t; a.step_ptr = nullptr;
        a.win_out = ts->WIN.as<int>() + (long long)t * B;
        launch_attention(a, st);
        GemmArgs g = layer_step_job(m, top, t, nullptr, c0t, ts->RecIn.as<float>());
        g.c_in.skip_first = 0;
        if (t + 1 < U) g.out2 = mkslot(rec + (long long)B * (C + W) + C, C + W);
        run_gemm(m, EPI_LSTM, g);
    }
    // ================= loss =================
    // (residual_connections: the projection reads the cell's outputs PLUS the cell's input sequence, seq2seq.py:359-360 at the top layer)
    const bool res_top = residual && D >= 2;
    const float* proj_in = top.hs; long long proj_ld = W;
    if (res_top) {
        launch_add_mul_mask(top.hs, top.hs_ld, y, W, nullptr, ts->Ytop.as<float>(), W, UB, W, st);
        proj_in = ts->Ytop.as<float>();
    }
    { GemmArgs g = plain_gemm(proj_in, proj_ld, (int)UB, W, ts->W_(ts->iE), V, nullptr, ts->logits.as<float>(), Vp); run_plain(m, g); }
    launch_softmax_ce(ts->logits.as<float>(), ts->d_out.as<int>(), ts->d_w.as<float>(), B, U, V, Vp, inv_count, ts->loss.as<double>(),
                      training ? 1 : 0, st);
    // Did every persistent recurrence so far run to its end?  (A launch gives up when its workgroups wait too long for each
    // other -- a GPU shared with another process: handoff.h.)  Nothing has been updated yet: start over with per-step launches,
    // and keep to them for the next 16, 32, ... steps.  Asked after the forward pass and again in front of the update.
    auto recurrences_gave_up = [&](bool& any) -> int {
        any = false;
        if (ts->rec_launches == ts->rec_checked) return 0;
        unsigned gave_up[16] = {0};
        for (int i = ts->rec_checked; i < ts->rec_launches; ++i)
            HIPCHK(hipMemcpyAsync(&gave_up[i], ts->rec_abort[i], 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int i = ts->rec_checked; i < ts->rec_launches; ++i) {
            any |= gave_up[i] != 0;
            if (gave_up[i] && i == ts->split_launch && !ts->split_off) {        // (the two launches were not resident together: one launch from now on)
                ts->split_off = true;
                fprintf(stderr, "cor_asv_ann_hip: the two launches of the attention cell's backward were not resident together (serialised "
                                "dispatch, or the GPU is shared): ONE launch from now on\n");
            }
        }
        ts->rec_checked = ts->rec_launches;
        if (any) {
            ts->rec_penalty = ts->rec_penalty ? std::min(2 * ts->rec_penalty, 1 << 20) : 16;
            ts->rec_skip = ts->rec_penalty + 1;
            fprintf(stderr, "cor_asv_ann_hip: a persistent recurrence of the train step gave up waiting (is the GPU shared?); "
                            "per-step launches for the next %d steps\n", ts->rec_penalty);
        }
        return 0;
    };
    {
        bool any = false;
        if (int rc = recurrences_gave_up(any)) return rc;
        if (any) return casv_train_step(m, mode, B, T, U, A, enc_idx, enc_val, dec_in, dec_out, weights, mask_enc, mask_dec, mask_cell, loss_out, norm_out);
    }
    if (!training) {                       // K.in_train_phase: the regulariser counts only in the train phase
        HIPCHK(hipMemcpyAsync(loss_out, ts->loss.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (norm_out) *norm_out = 0.0;
        return CASV_OK;
    }

    // ================= backward =================
    float* dlog = ts->logits.as<float>();
    // tied projection: dE += dlogits^T . G ; dG = dlogits . E
    {
        run_gemm_tn(m, dlog, Vp, Vp, V, proj_in, proj_ld, W, UB, ts->G_(ts->iE), W);      // (the padding columns of dlogits are zero)
        GemmArgs g2 = plain_gemm(dlog, Vp, (int)UB, Vp, ts->ETp.as<float>(), W, nullptr, ts->dG.as<float>(), W);
        run_plain(m, g2);
    }
    HIPCHK(hipMemsetAsync(ts->d_enc.p, 0, TB * C * 4, st));
    HIPCHK(hipMemsetAsync(ts->du.p, 0, TB * W * 4, st));
    HIPCHK(hipMemsetAsync(ts->dvaP.p, 0, (size_t)B * W * 4, st));
    for (auto& l : ts->layers) if (!l.drec_cleared) HIPCHK(hipMemsetAsync(l.dRec.p, 0, (size_t)l.len * B * l.kr * 4, st));
    HIPCHK(hipMemsetAsync(ts->dbvP.p, 0, (size_t)B * 4, st));
    float* dfin = ts->dfin.as<float>();          // [n-1][0|1][B][W]
    auto dfin_h = [&](int n) { return dfin + (size_t)(2 * (n - 1)) * B * W; };
    auto dfin_c = [&](int n) { return dfin + (size_t)(2 * (n - 1) + 1) * B * W; };
    // ---- attention cell (top decoder layer) ----
    {
        float* dc = dfin_c(D);
        HIPCHK(hipMemsetAsync(dc, 0, (size_t)B * W * 4, st));
        const int kr = C + W;
        bool topb_persistent = false;
        if (m->persist_mode != 0 && ts->rec_skip == 0 && ts->rec_launches < 16 && m->ncu >= 64) {
            // ONE launch for the whole backward recurrence of the cell (train_persist_topb.hip)
            TopBwdArgs ra{};
            ra.WrT = top.wrT.as<float>(); ra.WaN = ts->WaN.as<float>(); ra.dG = ts->dG.as<float>();
            ra.Gt = top.Gt.as<float>(); ra.Cs = top.Cs.as<float>(); ra.c0 = c0t; ra.dZ = top.Z.as<float>(); ra.dRec = top.dRec.as<float>();
            ra.dhatt = ts->dhatt.as<float>(); ra.DWQ = ts->DWQ.as<float>(); ra.WQ = ts->WQ.as<float>(); ra.Ast = ts->Ast.as<float>();
            ra.WIN = ts->WIN.as<int>(); ra.dc_out = dc; ra.B = B; ra.U = U; ra.W = W; ra.C = C;
            AttnBwdArgs& ab = ra.ab;
            ab.mcell = mcell; ab.ld_mcell = W + C; ab.mc_off = W; ab.va = ts->W_(ts->iva);
            ab.u = ts->u.as<float>(); ab.u_line = W; ab.u_time = (long long)B * W;
            ab.enc = enc_out; ab.enc_line = C; ab.enc_time = (long long)B * C;
            ab.d_enc = ts->d_enc.as<float>(); ab.du = ts->du.as<float>();
            ab.dva_part = ts->dvaP.as<float>(); ab.dbv_part = ts->dbvP.as<float>(); ab.B = B; ab.T = T; ab.W = W; ab.C = C;
            const size_t cb = train_recurrence_bwd_counter_bytes(B);
            ra.counters = reinterpret_cast<unsigned*>(static_cast<char*>(ts->rec_cnt.p) + cb * ts->rec_launches);
            // d_enc / du summed behind the recurrence instead of by float atomics inside it (attn_bwd.h, DEFER)
            static const int defer_opt = [] { const char* e = getenv("CASV_ATTN_DEFER"); const int v = e ? atoi(e) : 1; return v == 3 ? 3 : v == 0 ? 0 : 1; }();       // (the recurrence has forms 0, 1 and 3 only)
            const int defer = (T <= ATTN_DEFER_MAX_T && W % 64 == 0 && C % 64 == 0) ? defer_opt : 0;
            ra.DS = defer ? ts->DSrows.as<float>() : nullptr; ra.defer = defer;
            if (const int grid = train_attention_cell_bwd_grid(ra, m->ncu)) {
                hipEvent_t ev{};
                m->prof_begin(PC_PERSIST, 2.0 * B * U * ((double)4 * W * (C + W) + (double)W * W), 0.0, ev);
                // the attention backward of the samples as a launch of its own on a second stream, resident beside the big one: it
                // then runs under the h tiles instead of behind them (CASV_TOPB_SPLIT=0: one launch)
                // Two launches that hand rows to each other must be resident TOGETHER: where the runtime serialises kernel dispatch
                // (rocprofv3 --pmc sets ROCPROF_COUNTERS; AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING) the side launch would never see
                // the main one's counters -- one launch there, said once on stderr (profiles taken that way describe that form).
                static const bool split_opt = [] {
                    const char* e = getenv("CASV_TOPB_SPLIT");
                    if (e && e[0] == '0') return false;
                    auto on = [](const char* name) { const char* v = getenv(name); return v && v[0] && !(v[0] == '0' && !v[1]); };
                    const char* why = getenv("ROCPROF_COUNTERS") ? "ROCPROF_COUNTERS (counter collection)" : on("AMD_SERIALIZE_KERNEL") ? "AMD_SERIALIZE_KERNEL" :
                                      on("HIP_LAUNCH_BLOCKING") ? "HIP_LAUNCH_BLOCKING" : nullptr;
                    if (why && !(e && e[0] == '1')) {       // (CASV_TOPB_SPLIT=1 insists)
                        fprintf(stderr, "cor_asv_ann_hip: kernel dispatch is serialised (%s): the attention cell's backward runs as ONE launch "
                                        "(the two-launch form needs both resident together)\n", why);
                        return false;
                    }
                    return true;
                }();
                ra.split_a = split_opt && !ts->split_off && train_attention_cell_bwd_rows_fit(ra) ? 1 : 0;
                if (ra.split_a && !ts->side) {          // (no second stream to be had: one launch, as before)
                    if (hipStreamCreateWithFlags(&ts->side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ts->side = nullptr; }
                    else if (hipEventCreateWithFlags(&ts->ev_fork, hipEventDisableTiming) != hipSuccess ||
                             hipEventCreateWithFlags(&ts->ev_join, hipEventDisableTiming) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipStreamDestroy(ts->side); ts->side = nullptr;
                    }
                    if (!ts->side) { ts->split_off = true; ra.split_a = 0; }
                }
                if (ra.split_a) {
                    HIPCHK(hipEventRecord(ts->ev_fork, st));
                    HIPCHK(hipStreamWaitEvent(ts->side, ts->ev_fork, 0));
                    launch_train_attention_cell_bwd_rows(ra, grid, ts->side);
                    HIPCHK(hipEventRecord(ts->ev_join, ts->side));
                    ts->split_launch = ts->rec_launches;
                }
                launch_train_attention_cell_bwd(ra, grid, st);
                if (ra.split_a) HIPCHK(hipStreamWaitEvent(st, ts->ev_join, 0));
                if (defer) {
                    AttnDeferArgs da{};
                    da.dRec = ra.dRec; da.ld_drec = kr; da.mcell = mcell; da.ld_mcell = W + C; da.mc_off = W;
                    da.Ast = ra.Ast; da.WIN = ra.WIN; da.DS = ra.DS; da.WQ = ra.WQ; da.va = ab.va; da.u = ab.u;
                    da.d_enc = ab.d_enc; da.du = ab.du; da.B = B; da.U = U; da.T = T; da.W = W; da.C = C; da.what = defer;
                    launch_attention_deferred(da, st);
                }
                m->prof_end(PC_PERSIST, ev);
                ts->rec_abort[ts->rec_launches++] = ra.counters + (train_attention_cell_bwd_counter_bytes(B) / sizeof(unsigned) - 32);
                topb_persistent = true;
            }
        }
        if (!topb_persistent) HIPCHK(hipMemsetAsync(ts->dhatt.p, 0, UB * W * 4, st));     // (split-K outputs of the per-step GEMMs; the persistent kernel stores them)
        for (int t = U - 1; t >= 0 && !topb_persistent; --t) {
            LstmBwdArgs p{};
            p.a = ts->dG.as<float>() + (long long)t * B * W; p.lda = W;
            if (t < U - 1) {
                p.b = top.dRec.as<float>() + (long long)(t + 1) * B * kr + C; p.ldb = kr;
                p.c = ts->dhatt.as<float>() + (long long)(t + 1) * B * W; p.ldc = W;
            }
            p.gates = top.Gt.as<float>() + (long long)t * B * 4 * W;
            p.cell = top.Cs.as<float>() + (long long)t * B * W;
            p.c_prev = t > 0 ? top.Cs.as<float>() + (long long)(t - 1) * B * W : c0t; p.ld_cprev = W;
            p.dc = dc; p.dz = top.Z.as<float>() + (long long)t * B * 4 * W; p.rows = B; p.W = W;
            float* drec = top.dRec.as<float>() + (long long)t * B * kr;
            if (m->fused_backward) {
                BwdStepBatch fb{};
                fb.count = 1;
                BwdStepJob& q = fb.j[0];
                q.p = p;
                q.dc_in = ((U - 1 - t) & 1) ? ts->dcalt.as<float>() : dc; q.p.dc = ((U - 1 - t) & 1) ? dc : ts->dcalt.as<float>();
                q.Bt = top.wrT.as<float>(); q.out = drec; q.ld_out = kr; q.N = kr;
                hipEvent_t ev{};
                m->prof_begin(PC_GEMM, 2.0 * B * (double)kr * 4.0 * W, 4.0 * ((double)B * 4 * W + (double)kr * 4 * W + (double)B * kr), ev);
                launch_lstm_bwd_gemm(fb, st);
                m->prof_end(PC_GEMM, ev);
            } else {
                launch_lstm_bwd(p, st);
                GemmArgs g = plain_gemm(p.dz, 4 * W, B, 4 * W, top.wrT.as<float>(), kr, nullptr, drec, kr); g.out_zeroed = 1; run_gemm(m, EPI_PLAIN, g);
            }
            AttnBwdArgs ab{};
            ab.dxh = drec; ab.ld_dxh = kr; ab.ctx_off = 0; ab.mcell = mcell; ab.ld_mcell = W + C; ab.mc_off = W;
            ab.a = ts->Ast.as<float>() + (long long)(t + 1) * B * T; ab.win = ts->WIN.as<int>() + (long long)t * B;
            ab.wq = ts->WQ.as<float>() + (long long)t * B * W; ab.va = ts->W_(ts->iva);
            ab.u = ts->u.as<float>(); ab.u_line = W; ab.u_time = (long long)B * W;
            ab.enc = enc_out; ab.enc_line = C; ab.enc_time = (long long)B * C;
            ab.d_enc = ts->d_enc.as<float>(); ab.du = ts->du.as<float>(); ab.dwq = ts->DWQ.as<float>() + (long long)t * B * W;
            ab.dva_part = ts->dvaP.as<float>(); ab.dbv_part = ts->dbvP.as<float>(); ab.B = B; ab.T = T; ab.W = W; ab.C = C;
            launch_attention_bwd(ab, st);
            GemmArgs g = plain_gemm(ab.dwq, W, B, W, ts->WaN.as<float>(), W, nullptr, ts->dhatt.as<float>() + (long long)t * B * W, W);
            g.out_zeroed = 1;
            run_gemm(m, EPI_PLAIN, g);
        }
        if (!topb_persistent && m->fused_backward && (U & 1)) HIPCHK(hipMemcpyAsync(dc, ts->dcalt.p, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));   // dL/dc0 of the cell
        launch_colsum(ts->dvaP.as<float>(), B, W, W, ts->G_(ts->iva), st);
        launch_colsum(ts->dbvP.as<float>(), B, 1, 1, ts->G_(ts->ibv), st);
        // dL/dh0 of the cell = recurrent part of step 0 + the query path of step 0
        launch_mul_mask(top.dRec.as<float>() + C, kr, nullptr, dfin_h(D), W, B, W, st);
        launch_axpy(dfin_h(D), ts->dhatt.as<float>(), (long long)B * W, st);       // slot of step 0
        // y-part gradient for all steps, weight grads
        GemmArgs g = plain_gemm(top.Z.as<float>(), 4 * W, (int)UB, 4 * W, top.wxT.as<float>(), W, nullptr, ts->dXtop.as<float>(), W);
        run_plain(m, g);
        launch_mul_rowmask(ts->dXtop.as<float>(), W, mcell, W + C, ts->dXtop.as<float>(), W, UB, B, W, st);
        if (res_top) launch_axpy(ts->dXtop.as<float>(), ts->dG.as<float>(), UB * W, st);       // the sum's other branch: dL/d(cell input sequence) += dL/d(projection input)
        if (int rc = layer_weight_grads(m, top, ts->Ym.as<float>(), W, ts->RecIn.as<float>(), kr)) return rc;
        // attention parameters: dWaT = DWQ^T . Hprev ; dbUW = colsum(DWQ) ; u path
        if (!ts->tens[ts->iWaT].frozen) {
            // the h_prev of every step are columns C.. of the cell's recurrent-side inputs
            run_gemm_tn(m, ts->DWQ.as<float>(), W, W, W, ts->RecIn.as<float>() + C, kr, W, UB, ts->G_(ts->iWaT), W, ts->G_(ts->ibUW));
        }
        {
            run_gemm_tn(m, ts->du.as<float>(), W, W, W, enc_out, C, C, TB, ts->G_(ts->iUT), C);
            GemmArgs gd = plain_gemm(ts->du.as<float>(), W, (int)TB, W, ts->UaN.as<float>(), C, nullptr, ts->d_enc.as<float>(), C, 1);
            run_plain(m, gd);
        }
    }
    // bridge_dense backward: dfin_h(n) / dfin_c(n) arrive as gradients w.r.t. the BRIDGED states; through tanh' and the Dense layer
    // they become the gradients w.r.t. encoder layer n's own final states (in place), and leave the Dense layers' gradients
    auto bridge_backward = [&](int n) {
        if (!bridged) return;
        for (int s_ = 0; s_ < 2; ++s_) {
            const TrainState::Bridge& br = ts->bridge[2 * (n - 1) + s_];
            float* d = s_ ? dfin_c(n) : dfin_h(n);
            const float* raw = (s_ ? cfin : hfin) + (size_t)(n - 1) * B * W;
            const float* post = (s_ ? ts->cbr.as<float>() : ts->hbr.as<float>()) + (size_t)(n - 1) * B * W;
            launch_tanh_bwd(d, post, (long long)B * W, st);
            if (!ts->tens[br.ikt].frozen) run_gemm_tn(m, d, W, W, W, raw, W, W, B, ts->G_(br.ikt), W, ts->G_(br.ib));
            GemmArgs g = plain_gemm(d, W, B, W, br.kn.as<float>(), W, nullptr, ts->brtmp.as<float>(), W);
            g.ksplit = 0; g.kgroups = 0;
            run_gemm(m, EPI_PLAIN, g);
            (void)hipMemcpyAsync(d, ts->brtmp.p, (size_t)B * W * 4, hipMemcpyDeviceToDevice, st);
        }
    };
    bridge_backward(D);
    // ---- decoder layer n with encoder layer n+1 (the mirror of the forward pairing) ----
    const float* dy = ts->dXtop.as<float>();        // gradient w.r.t. DO[D-1] (or Y0 when D == 1)
    const float* dO = ts->d_enc.as<float>();        // gradient w.r.t. O[D]
    long long ld_dO = C;
    // (a pair's input gradients are the next pair's output gradients: two buffers per chain taking turns, nothing is copied)
    float* dy_bufs[2] = {ts->dXtop.as<float>(), ts->dYl.as<float>()}; int dy_cur = 0;
    float* do_bufs[2] = {ts->dXl.as<float>(), ts->dOin.as<float>()}; int do_out = 0;
    for (int n = D - 1; n >= 1 && deep; --n) {
        TLayer& ld = dec_layer(n);
        TLayer& lf = enc_dir(n + 1, 0); TLayer& lb = enc_dir(n + 1, 1);
        const float* xin = n == 1 ? ts->Y0.as<float>() : ts->DO[n - 1].as<float>();
        const bool res_d = residual && n >= 2;
        if (res_d) launch_mul_mask(dy, W, mdec_n(n), dy_bufs[dy_cur ^ 1], W, UB, W, st);
        {
            LayerBwd one[1] = {{&ld, dy, W, mdec_n(n), nullptr, nullptr, h0base + (size_t)(n - 1) * B * W, c0base + (size_t)(n - 1) * B * W, dfin_c(n),
                                xin, W, dy_bufs[dy_cur ^ 1], W, res_d ? 1 : 0}};
            if (int rc = layers_backward(m, one, 1)) return rc;
        }
        HIPCHK(hipMemcpyAsync(dfin_h(n), ld.dRec.as<float>(), (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
        bridge_backward(n);
        dy_cur ^= 1; dy = dy_bufs[dy_cur];
        // the BiLSTM n + 1: forward direction takes columns [0,W) of dO, backward direction [W,2W); both add into the gradient of their
        // shared input, whose cross sum (the Lambda is its own adjoint) is the gradient w.r.t. O[n]
        float* dXn = do_bufs[0]; float* dOn = do_bufs[1];
        LayerBwd pair[2] = {
            {&lf, dO, ld_dO, menc_n(n + 1), nullptr, nullptr, nullptr, nullptr, ts->dcbuf.as<float>(),
             ts->XD[n + 1].as<float>(), 2 * W, dXn, 2 * W, 0},
            {&lb, dO + W, ld_dO, menc_n(n + 1) ? menc_n(n + 1) + W : nullptr, dfin_h(n + 1), dfin_c(n + 1), nullptr, nullptr, ts->dcbuf2.as<float>(),
             ts->XD[n + 1].as<float>(), 2 * W, dXn, 2 * W, 1}};
        if (int rc = layers_backward(m, pair, 2)) return rc;
        launch_cross_sum(dXn, dOn, TB * 2 * W, st);
        dO = dOn; ld_dO = 2 * W;
    }
    for (int n = D - 1; n >= 1 && !deep; --n) {
        TLayer& ld = dec_layer(n);
        TLayer& le = enc_layer(n + 1);
        const float* xin = n == 1 ? ts->Y0.as<float>() : ts->DO[n - 1].as<float>();
        // residual_connections: decoder layer n >= 2 / encoder layer n + 1 >= 3 pass their output gradient (times the mask) straight on to
        // their input as well: the input-gradient buffers start from it and the layers' data gradients are added
        const bool res_d = residual && n >= 2, res_e = residual && n + 1 >= 3;
        if (res_d) launch_mul_mask(dy, W, mdec_n(n), dy_bufs[dy_cur ^ 1], W, UB, W, st);
        if (res_e) launch_mul_mask(dO, ld_dO, menc_n(n + 1), do_bufs[do_out], le.kx, TB, W, st);
        LayerBwd pair[2] = {
            {&ld, dy, W, mdec_n(n), nullptr, nullptr, h0base + (size_t)(n - 1) * B * W, c0base + (size_t)(n - 1) * B * W, dfin_c(n),
             xin, W, dy_bufs[dy_cur ^ 1], W, res_d ? 1 : 0},
            {&le, dO, ld_dO, menc_n(n + 1), dfin_h(n + 1), dfin_c(n + 1), nullptr, nullptr, ts->dcbuf.as<float>(),
             ts->O[n].as<float>(), le.kx, do_bufs[do_out], le.kx, res_e ? 1 : 0}};
        if (int rc = layers_backward(m, pair, 2)) return rc;
        // dL/dh0, dL/dc0 of the decoder layer go to the encoder layer of the same index
        HIPCHK(hipMemcpyAsync(dfin_h(n), ld.dRec.as<float>(), (size_t)B * W * 4, hipMemcpyDeviceToDevice, st));
        bridge_backward(n);
        dy_cur ^= 1; dy = dy_bufs[dy_cur];
        dO = do_bufs[do_out]; ld_dO = le.kx; do_out ^= 1;
    }
    launch_embed_scatter(ts->G_(ts->iE), ts->d_in.as<int>(), nullptr, dy, W, B, U, 1, V, W, st);

    // ---- encoder layer 1: forward direction takes columns [0,W) of dO1, backward direction [W,2W) ----
    {
        LayerBwd pair[2] = {
            {Lfw, dO, ld_dO, menc_n(1), nullptr, nullptr, nullptr, nullptr, ts->dcbuf.as<float>(),
             ts->X0.as<float>(), W, ts->dX0.as<float>(), W, 0},
            {Lbw, dO + W, ld_dO, menc_n(1) ? menc_n(1) + W : nullptr, dfin_h(1), dfin_c(1), nullptr, nullptr, ts->dcbuf2.as<float>(),
             ts->X0.as<float>(), W, ts->dX0.as<float>(), W, 1}};
        if (int rc = layers_backward(m, pair, 2)) return rc;
    }
    launch_embed_scatter(ts->G_(ts->iE), ts->e_idx.as<int>(), enc_val ? ts->e_val.as<float>() : nullptr, ts->dX0.as<float>(), W, B, T, A, V, W, st);

    {
        bool any = false;
        if (int rc = recurrences_gave_up(any)) return rc;
        if (any) return casv_train_step(m, mode, B, T, U, A, enc_idx, enc_val, dec_in, dec_out, weights, mask_enc, mask_dec, mask_cell, loss_out, norm_out);
        if (ts->rec_launches) ts->rec_penalty = 0;
    }
    // ---- regulariser, clip, update ----
    launch_reg(ts->W_(ts->iE), ts->G_(ts->iE), V, W, ts->loss.as<double>(), 1, st);
    // (norm and update over all parameter tensors as one launch each; lists of MULTI_MAX tensors at a time)
    auto over_tensors = [&](int max_blocks, auto&& launch) {
        MultiTensor mt{};
        for (auto& t : ts->tens) {
            if (t.frozen) continue;
            if (mt.count == MULTI_MAX) { launch(mt); mt = MultiTensor{}; }
            multi_add(mt, t.w.as<float>(), t.g.as<float>(), t.m.as<float>(), t.v.as<float>(), (long long)t.n, max_blocks);
        }
        launch(mt);
    };
    // (few workgroups per tensor: every one of them ends in an atomic add on the ONE sum -- ~12 ns each, one after the other)
    over_tensors(64, [&](const MultiTensor& mt) { launch_sumsq_multi(mt, ts->normsq.as<double>(), st); });
    if (mode == 1) {
        ts->step += 1;
        const double b1 = ts->ap.beta1, b2 = ts->ap.beta2;
        const float lr_t = (float)(ts->ap.lr * sqrt(1.0 - pow(b2, (double)ts->step)) / (1.0 - pow(b1, (double)ts->step)));
        over_tensors(2048, [&](const MultiTensor& mt) {
            launch_adam_multi(mt, ts->normsq.as<double>(), ts->ap.clipnorm, lr_t, (float)b1, (float)b2, ts->ap.epsilon, st);
        });
        refresh_derived(m);
    }
    double nsq = 0.0;
    HIPCHK(hipMemcpyAsync(loss_out, ts->loss.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nsq, ts->normsq.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (norm_out) *norm_out = sqrt(nsq);
    if (m->prof.on) m->prof.collect();
    return CASV_OK;
}

// Keras-layout view of the master weights (which = 0) or of the last gradients (which = 1).
static int keras_view(casv_model* m, int which, std::map<std::string, std::vector<float>>& out) {
    TrainState* ts = m->train;
    const int W = m->W, C = m->C, D = m->D;
    HIPCHK(hipStreamSynchronize(m->stream));
    auto fetch = [&](int i, std::vector<float>& v) -> int {
        v.resize(ts->tens[i].n);
        HIPCHK(hipMemcpy(v.data(), which ? ts->tens[i].g.p : ts->tens[i].w.p, v.size() * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    if (int rc = fetch(ts->iE, out["E"])) return rc;
    for (auto& l : ts->layers) {
        std::vector<float> wx, wr, b;
        if (int rc = fetch(l.iwx, wx)) return rc;
        if (int rc = fetch(l.iwr, wr)) return rc;
        if (int rc = fetch(l.ib, b)) return rc;
        unpack_train_lstm(W, l.kx, l.kr - W, wx, wr, b, out[l.name + "_K"], out[l.name + "_R"], out[l.name + "_b"]);
    }
    std::vector<float> ut, wat;
    if (int rc = fetch(ts->iUT, ut)) return rc;
    if (int rc = fetch(ts->iWaT, wat)) return rc;
    auto& Uk = out["att_U"]; auto& Wa = out["att_Wa"];
    Uk.resize((size_t)C * W); Wa.resize((size_t)W * W);
    for (int j = 0; j < W; ++j) for (int c = 0; c < C; ++c) Uk[(size_t)c * W + j] = ut[(size_t)j * C + c];
    for (int j = 0; j < W; ++j) for (int k = 0; k < W; ++k) Wa[(size_t)k * W + j] = wat[(size_t)j * W + k];
    if (int rc = fetch(ts->ibUW, out["att_bUW"])) return rc;
    if (int rc = fetch(ts->iva, out["att_va"])) return rc;
    if (int rc = fetch(ts->ibv, out["att_bv"])) return rc;
    for (size_t i = 0; i < ts->bridge.size(); ++i) {
        const std::string b = "bridge" + std::to_string(i / 2 + 1) + (i % 2 ? "_c" : "_h");
        std::vector<float> kt;
        if (int rc = fetch(ts->bridge[i].ikt, kt)) return rc;
        auto& K = out[b + "_K"];
        K.resize((size_t)W * W);
        for (int j = 0; j < W; ++j) for (int k = 0; k < W; ++k) K[(size_t)k * W + j] = kt[(size_t)j * W + k];
        if (int rc = fetch(ts->bridge[i].ib, out[b + "_b"])) return rc;
    }
    (void)D;
    return 0;
}

extern "C" int casv_train_get_gradient(casv_model* m, const char* name, float* out, int64_t capacity) {
    if (!m || !name || !out) return fail(CASV_ERR_ARG, "null argument");
    if (!m->train) return fail(CASV_ERR_STATE, "no training session");
    HIPCHK(hipSetDevice(m->device));
    std::map<std::string, std::vector<float>> view;
    if (int rc = keras_view(m, 1, view)) return rc;
    auto it = view.find(name);
    if (it == view.end()) return fail(CASV_ERR_ARG, "unknown tensor '%s'", name);
    if ((size_t)capacity < it->second.size()) return fail(CASV_ERR_ARG, "buffer too small");
    memcpy(out, it->second.data(), it->second.size() * 4);
    return CASV_OK;
}

// Copy the current master weights into the handle's Keras-layout tensors without ending the session
// (ModelCheckpoint / EarlyStopping(restore_best_weights), seq2seq.py:619-622).
extern "C" int casv_train_sync_weights(casv_model* m) {
    if (!m) return fail(CASV_ERR_ARG, "null argument");
    if (!m->train) return fail(CASV_ERR_STATE, "no training session");
    HIPCHK(hipSetDevice(m->device));
    std::map<std::string, std::vector<float>> view;
    if (int rc = keras_view(m, 0, view)) return rc;
    for (auto& kv : view) m->host[kv.first] = kv.second;
    return CASV_OK;
}

// Bring the trained weights back into the handle (Keras layout) and repack them for inference
// (the reference's _resync_decoder after training, seq2seq.py:645).
extern "C" int casv_train_end(casv_model* m) {
    if (!m) return fail(CASV_ERR_ARG, "null argument");
    if (!m->train) return fail(CASV_ERR_STATE, "no training session");
    HIPCHK(hipSetDevice(m->device));
    std::map<std::string, std::vector<float>> view;
    if (int rc = keras_view(m, 0, view)) return rc;
    for (auto& kv : view) m->host[kv.first] = kv.second;
    casv_train_release(m);
    return casv_commit_weights(m);
}
