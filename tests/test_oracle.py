"""The oracle pinned by what is available (the reference holds no golden
vectors and cannot be run here): analytic known answers, an independent implementation of the
LSTM stacks (torch CPU), fp32-vs-fp64 self-agreement, and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import ModelConfig, make_weights, make_lines, make_vocabulary, vectorize_lines
from oracle.decode import (OracleModel, Node, decode_batch_greedy, decode_sequence_greedy,
                           decode_sequence_beam, correct_lines)
from oracle.model import lstm_step, encode, decoder_step, attention


def test_lstm_step_known_answer():
    # 1 unit, hand-computed: z = x*K + h*R + b with gate order i,f,c,o
    K = np.array([[0.5, -0.25, 1.0, 2.0]])
    R = np.array([[0.1, 0.2, -0.3, 0.4]])
    b = np.array([0.0, 1.0, 0.0, -1.0])
    x, h, c = np.array([[2.0]]), np.array([[0.5]]), np.array([[-1.0]])
    z = np.array([1.05, 0.6, 1.85, 3.2])
    sig = lambda v: 1 / (1 + np.exp(-v))
    c2 = sig(z[1]) * -1.0 + sig(z[0]) * np.tanh(z[2])
    h2 = sig(z[3]) * np.tanh(c2)
    hh, cc = lstm_step(x, h, c, K, R, b)
    assert np.allclose(hh, h2, atol=1e-12) and np.allclose(cc, c2, atol=1e-12)


def _torch_lstm(K, R, b, bidirectional=False, Kb=None, Rb=None, bb=None):
    nin, W = K.shape[0], R.shape[0]
    m = torch.nn.LSTM(nin, W, batch_first=True, bidirectional=bidirectional).double()

    def put(Km, Rm, bv, sfx):
        # torch gate order i,f,g,o == Keras i,f,c,o; torch weights are (4W, in)
        getattr(m, 'weight_ih_l0' + sfx).data = torch.tensor(Km.T.copy())
        getattr(m, 'weight_hh_l0' + sfx).data = torch.tensor(Rm.T.copy())
        getattr(m, 'bias_ih_l0' + sfx).data = torch.tensor(bv.copy())
        getattr(m, 'bias_hh_l0' + sfx).data = torch.zeros(4 * W, dtype=torch.float64)
    put(K, R, b, '')
    if bidirectional:
        put(Kb, Rb, bb, '_reverse')
    return m


def test_encoder_matches_torch_lstm_stack():
    cfg = ModelConfig(depth=3, width=32, voc_size=40)
    w = make_weights(cfg, dtype=np.float64, emb_scale=8.0)
    rng = np.random.default_rng(0)
    for k in w:
        if k.endswith('_b'):
            w[k] = rng.normal(size=w[k].shape)      # non-trivial biases
    lines, idx = make_lines(5, 17, 3, voc_size=40)
    x = np.eye(40)[idx]
    out = encode(cfg, w, x)
    with torch.no_grad():
        x0 = torch.tensor(x @ w['E'])
        l1 = _torch_lstm(w['enc1_fw_K'], w['enc1_fw_R'], w['enc1_fw_b'], True,
                         w['enc1_bw_K'], w['enc1_bw_R'], w['enc1_bw_b'])
        y1, (h1, c1) = l1(x0)
        l2 = _torch_lstm(w['enc2_K'], w['enc2_R'], w['enc2_b'])
        y2, (h2, c2) = l2(y1)
        l3 = _torch_lstm(w['enc3_K'], w['enc3_R'], w['enc3_b'])
        y3, (h3, c3) = l3(y2)
    assert np.allclose(out[0], y3.numpy(), atol=1e-10)
    # decoder layer 1 starts from the BACKWARD final state (seq2seq.py:280-281)
    assert np.allclose(out[1], h1[1].numpy(), atol=1e-10) and np.allclose(out[2], c1[1].numpy(), atol=1e-10)
    assert np.allclose(out[3], h2[0].numpy(), atol=1e-10) and np.allclose(out[6], c3[0].numpy(), atol=1e-10)
    assert out[7].shape == (5, 18) and not out[7].any()


def test_encoder_with_residual_connections_and_bridge_matches_torch():
    """The optional topologies of seq2seq.py:284-301 against an independent implementation (torch LSTM stack, float64): from the third
    layer on a layer's output sequence is LSTM output + input sequence (the second layer's 2W-wide input is not summed), the states
    handed to the decoder are the LSTM's own final states -- through Dense(width, tanh) layers when bridge_dense is set."""
    cfg = ModelConfig(depth=4, width=32, voc_size=40, residual_connections=True, bridge_dense=True)
    w = make_weights(cfg, dtype=np.float64, emb_scale=8.0)
    rng = np.random.default_rng(1)
    for k in w:
        if k.endswith('_b'):
            w[k] = rng.normal(size=w[k].shape) * 0.3
    lines, idx = make_lines(5, 17, 3, voc_size=40)
    x = np.eye(40)[idx]
    out = encode(cfg, w, x)
    with torch.no_grad():
        x0 = torch.tensor(x @ w['E'])
        l1 = _torch_lstm(w['enc1_fw_K'], w['enc1_fw_R'], w['enc1_fw_b'], True,
                         w['enc1_bw_K'], w['enc1_bw_R'], w['enc1_bw_b'])
        y1, (h1, c1) = l1(x0)
        y2, (h2, c2) = _torch_lstm(w['enc2_K'], w['enc2_R'], w['enc2_b'])(y1)
        o3, (h3, c3) = _torch_lstm(w['enc3_K'], w['enc3_R'], w['enc3_b'])(y2)
        y3 = o3 + y2
        o4, (h4, c4) = _torch_lstm(w['enc4_K'], w['enc4_R'], w['enc4_b'])(y3)
        y4 = o4 + y3
    assert np.allclose(out[0], y4.numpy(), atol=1e-10)
    br = lambda n, part, v: np.tanh(v.numpy() @ w['bridge%d_%s_K' % (n, part)] + w['bridge%d_%s_b' % (n, part)])
    assert np.allclose(out[1], br(1, 'h', h1[1]), atol=1e-10) and np.allclose(out[2], br(1, 'c', c1[1]), atol=1e-10)
    assert np.allclose(out[5], br(3, 'h', h3[0]), atol=1e-10) and np.allclose(out[8], br(4, 'c', c4[0]), atol=1e-10)
    # the default topology's tensors are the same with and without the flags (the bridge tensors are drawn last)
    w0 = make_weights(ModelConfig(depth=4, width=32, voc_size=40), dtype=np.float64, emb_scale=8.0)
    w1 = make_weights(cfg, dtype=np.float64, emb_scale=8.0)
    assert all(np.array_equal(w0[k], w1[k]) for k in w0) and len(w1) == len(w0) + 16


def test_attention_window_edges():
    cfg = ModelConfig(depth=2, width=32, voc_size=16)
    w = make_weights(cfg, dtype=np.float64)
    T = 30
    rng = np.random.default_rng(1)
    enc = rng.normal(size=(1, T, 32)); u = enc @ w['att_U']; h = rng.normal(size=(1, 32))
    # first step: a_prev = 0 -> t' = 1 -> s in [0, 6] (|t'-s| == 5 is inside)
    _, a = attention(cfg, w, h, np.zeros((1, T)), enc, u)
    assert np.nonzero(a[0])[0].tolist() == list(range(0, 7)) and abs(a.sum() - 1) < 1e-12
    # one-hot at 10 -> t' = 11 -> s in [6, 16]: 11 positions
    a_prev = np.zeros((1, T)); a_prev[0, 10] = 1
    ctx, a = attention(cfg, w, h, a_prev, enc, u)
    assert np.nonzero(a[0])[0].tolist() == list(range(6, 17))
    assert np.allclose(ctx, (a[0][:, None] * enc[0]).sum(0))
    # real-valued t' = 11.5 -> s in [7, 16]: 10 positions
    a_prev = np.zeros((1, T)); a_prev[0, 10] = a_prev[0, 11] = 0.5
    _, a = attention(cfg, w, h, a_prev, enc, u)
    assert np.nonzero(a[0])[0].tolist() == list(range(7, 17))
    # window past the end of the line -> 0/0 = NaN (reference behaviour)
    a_prev = np.zeros((1, T)); a_prev[0, 29] = 1.0; a_prev[0, 28] = 0.3
    with np.errstate(invalid='ignore'):
        _, a = attention(cfg, w, h, a_prev, enc, u)
    assert np.isnan(a).all()
    # exactly one position left -> weight exactly 1.0
    a_prev = np.zeros((1, T)); a_prev[0, 29] = 1.0; a_prev[0, 4] = 1.0    # t' = 34
    _, a = attention(cfg, w, h, a_prev, enc, u)
    assert a[0, 29] == 1.0 and np.count_nonzero(a) == 1


def test_vectorize_lines_layouts():
    cfg = ModelConfig(depth=1, width=32, voc_size=8)
    c_i = {'': 0, '\n': 1, 'a': 2, 'b': 3, 'c': 4, 'd': 5, 'e': 6, 'f': 7}
    m = OracleModel(cfg, {}, mapping=(c_i, {i: c for c, i in c_i.items()}))
    enc, din, dout, wts = vectorize_lines(m, ['ab\n', 'c\n'], ['ab\n', 'cd\n'])
    assert enc.dtype == np.uint32 and enc.shape == (2, 3, 8)
    assert enc[0].argmax(1).tolist() == [2, 3, 1] and enc[1].sum() == 2 and not enc[1, 2].any()
    assert din.shape == (2, 4, 8) and not din[:, 0].any() and din[0, 1, 2] == 1
    assert dout[0].argmax(1).tolist()[:3] == [2, 3, 1] and wts.tolist() == [[1, 1, 1, 0], [1, 1, 1, 0]]
    # unknown and GAP characters fall to index 0; only the former is reported
    enc, _, _, _ = vectorize_lines(m, ['x\a\n'], [''])
    assert enc[0, 0, 0] == 1 and enc[0, 1, 0] == 1 and len(m.errors) == 1
    # probability line
    enc, _, _, _ = vectorize_lines(m, ['ab\n'], [''], [[0.9, 0.5, 1.0]])
    assert enc.dtype == np.float32 and enc[0, 0, 2] == np.float32(0.9) and enc[0, 1, 3] == 0.5
    # confusion network: alternatives of different length are zero padded to the longest
    confmat = [[[('a', 0.6), ('bc', 0.4)], [('\n', 1.0)]]]
    enc, _, _, _ = vectorize_lines(m, confmat, [''], confmat)
    assert enc.shape == (1, 3, 8)
    assert enc[0, 0, 2] == np.float32(0.6) and enc[0, 0, 3] == np.float32(0.4)
    assert enc[0, 1, 4] == np.float32(0.4) and enc[0, 1].sum() == np.float32(0.4) and enc[0, 2, 1] == 1


def test_node_order_and_procost():
    root = Node(state=None, value='', scores=None, cost=0.0, length0=10, cost0=3.0)
    a = Node(state=None, value='a', scores=None, cost=np.float32(0.5), parent=root)
    b = Node(state=None, value='b', scores=None, cost=np.float32(0.25), parent=a)
    assert root.pro_cost() == -27.0 and a.pro_cost() == -(0.5 + 3.0 * 8)
    assert b.length == 3 and b.length0 == 10 and b.cost0 == 3.0 and abs(b.cum_cost - 0.75) < 1e-7
    assert b > a > root and str(b) == 'ab'


@pytest.mark.parametrize('d,W,V,L,es', [(1, 64, 40, 12, 8.0), (2, 64, 40, 12, 8.0), (3, 32, 24, 9, 16.0)])
def test_fp32_fp64_agree(d, W, V, L, es):
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    lines, _ = make_lines(6, L, 11, voc_size=V)
    res = {}
    for dt in (np.float32, np.float64):
        m = OracleModel(cfg, make_weights(cfg, dtype=dt, emb_scale=es), batch_size=4)
        enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
        res[dt] = decode_batch_greedy(m, enc_in, return_indexes=True)
    assert (res[np.float32][5] == res[np.float64][5]).all()
    assert np.allclose(res[np.float32][3], res[np.float64][3], rtol=1e-4)


def test_greedy_variants_consistent():
    """decode_sequence_greedy on one line equals decode_batch_greedy on a batch of that one line as
    long as index 0 never wins (the only difference of the two loops besides the early exit)."""
    cfg = ModelConfig(depth=2, width=64, voc_size=40)
    m = OracleModel(cfg, make_weights(cfg, emb_scale=8.0))
    lines, _ = make_lines(3, 10, 5, voc_size=40)
    for line in lines:
        enc_in, _, _, _ = vectorize_lines(m, [line], [[]])
        batch = decode_batch_greedy(m, enc_in)
        text, probs, score, aligns = decode_sequence_greedy(m, enc_in[0])
        assert batch[1][0] == text and np.allclose(batch[2][0], probs) and abs(batch[3][0] - score) < 1e-6


def test_beam_follows_source_when_flat():
    """With a flat output distribution only the rejection candidate (source character at
    probability rejection_threshold) passes the beam threshold: the search reproduces the input at
    cost -ln(0.3) per character (seq2seq.py:1457-1470)."""
    cfg = ModelConfig(depth=1, width=32, voc_size=40)
    m = OracleModel(cfg, make_weights(cfg, emb_scale=1.0), batch_size=4)
    lines, _ = make_lines(2, 9, 5, voc_size=40)
    out, probs, scores, aligns = correct_lines(m, lines, fast=False, greedy=False)
    assert out == lines
    assert np.allclose(scores, -np.log(np.float32(0.3)), atol=1e-6)
    assert all(np.max(a) == 1.0 for a in aligns[0])          # one-hot rejection alignments
    assert correct_lines(m, [], fast=False, greedy=False) == ([], [], [], [])
    with pytest.raises(AssertionError):
        correct_lines(m, lines, fast=True, greedy=False)


def test_golden_fixtures_match_oracle(golden_dir):
    """The committed fixtures are exactly what the oracle produces today (they were generated by
    tests/golden/make_golden.py from the oracle; the reference cannot be run here)."""
    from tests.golden.make_golden import CASES, run_case
    for name in CASES:
        path = os.path.join(golden_dir, name + '.npz')
        assert os.path.exists(path), 'missing fixture %s (run python tests/golden/make_golden.py)' % path
        with np.load(path) as f:
            want = {k: f[k] for k in f.files}
        got = run_case(name)
        for k, v in want.items():
            if v.dtype.kind in 'iuU':
                assert np.array_equal(got[k], v), (name, k)
            else:
                assert np.allclose(got[k], v, rtol=2e-5, atol=1e-6, equal_nan=True), (name, k)


def test_attention_cell_known_answer():
    """attention.py:526-575 by hand: W_a = 0 and b_UW = 0 make the query vanish, u[s] = (c_s, 0), v_a = (1, 0), b_v = 0.5
    give the energy exp(tanh(c_s) + 0.5); the previous alignment sits on position 3, so t' = 4 and positions 0..9 of the
    12 survive the |t' - s| <= 5 mask; the context is the weighted mean of the attended rows."""
    import math
    cfg = ModelConfig(depth=2, width=2, voc_size=4)
    T = 12
    c = [0.1 * s - 0.4 for s in range(T)]
    w = {'att_Wa': np.zeros((2, 2)), 'att_bUW': np.zeros(2), 'att_va': np.array([1.0, 0.0]), 'att_bv': np.array([0.5])}
    u = np.array([[[cs, 0.0] for cs in c]])
    enc = np.array([[[float(s), 1.0 - s] for s in range(T)]])
    a_prev = np.zeros((1, T)); a_prev[0, 3] = 1.0
    ctx, a = attention(cfg, w, np.array([[7.0, -3.0]]), a_prev, enc, u)
    e = [math.exp(math.tanh(c[s]) + 0.5) if abs(4.0 - s) <= 5 else 0.0 for s in range(T)]
    tot = sum(e)
    assert e[10] == 0.0 and e[9] > 0.0 and e[0] > 0.0
    assert np.allclose(a[0], [x / tot for x in e], atol=1e-15)
    assert np.allclose(ctx[0], [sum(e[s] / tot * s for s in range(T)), sum(e[s] / tot * (1.0 - s) for s in range(T))], atol=1e-13)
    # a query that is not zero shifts every energy's argument by the same h.W_a + b_UW
    w2 = dict(w, att_Wa=np.array([[0.2, 0.0], [0.0, 0.0]]), att_bUW=np.array([0.05, 0.0]))
    _, a2 = attention(cfg, w2, np.array([[1.5, 9.0]]), a_prev, enc, u)
    e2 = [math.exp(math.tanh(c[s] + 0.2 * 1.5 + 0.05) + 0.5) if abs(4.0 - s) <= 5 else 0.0 for s in range(T)]
    assert np.allclose(a2[0], [x / sum(e2) for x in e2], atol=1e-15)


def test_tied_projection_and_softmax_known_answer():
    """seq2seq.py:379 `softmax(h . E^T)` with the SAME matrix that embeds the input (seq2seq.py:239-243), through a
    whole decoder step: all kernels zero, so the top cell's output is o * tanh(i * g) with gates set by the bias alone,
    the attention is uniform over its window, and the probabilities are softmax(E h) by hand."""
    import math
    cfg = ModelConfig(depth=1, width=2, voc_size=3)
    W, C, V, T = 2, 4, 3, 5
    E = np.array([[1.0, 0.0], [0.0, 2.0], [1.0, 1.0]])
    sig = lambda v: 1 / (1 + math.exp(-v))
    bias = np.array([0.3, -0.2, 50.0, 50.0, 0.7, 0.1, 2.0, -1.0])        # i (2), f (2), c~ (2), o (2)
    w = {'E': E, 'att_U': np.zeros((C, W)), 'att_Wa': np.zeros((W, W)), 'att_bUW': np.zeros(W), 'att_va': np.zeros(W),
         'att_bv': np.zeros(1), 'dec1_K': np.zeros((W + C, 4 * W)), 'dec1_R': np.zeros((W, 4 * W)), 'dec1_b': bias}
    enc = np.arange(T * C, dtype=np.float64).reshape(1, T, C)
    states = [np.zeros((1, W)), np.zeros((1, W)), np.zeros((1, T))]        # h, c = 0: the forget gate does not matter
    p, new = decoder_step(cfg, w, np.array([[0.2, 0.5, 0.3]]), enc, states)
    h = [sig(bias[6 + k]) * math.tanh(sig(bias[k]) * math.tanh(bias[4 + k])) for k in range(W)]
    logits = [E[v, 0] * h[0] + E[v, 1] * h[1] for v in range(V)]
    z = sum(math.exp(x) for x in logits)
    assert np.allclose(new[0][0], h, atol=1e-15)
    assert np.allclose(p[0], [math.exp(x) / z for x in logits], atol=1e-15) and abs(p.sum() - 1) < 1e-15
    assert np.allclose(new[2][0], [0.2] * 5, atol=1e-15)                   # zero energies: uniform over the window s <= 6


def test_oracle_matches_keras_goldens(golden_dir):
    """Reference-generated fixtures (tests/golden/make_keras_goldens.py, run where Keras 2.3 / TF 1.15 exist) pin the
    oracle to the reference itself.  None can be produced in this pipeline: until someone drops them into
    tests/golden/keras/, parity stays "unpinned" and this test says so."""
    import glob
    from tests.golden.make_golden import CASES, NTENS
    files = sorted(glob.glob(os.path.join(os.environ.get('CASV_GOLDEN_DIR', os.path.join(golden_dir, 'keras')), '*.npz')))
    if not files:
        pytest.skip('no reference-generated fixtures under tests/golden/keras (parity unpinned: see DESIGN.md section 3)')
    for path in files:
        name = os.path.splitext(os.path.basename(path))[0]
        d, W, V, B, L, seed, es, N = CASES[name]
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        m = OracleModel(cfg, make_weights(cfg, emb_scale=es), batch_size=N)
        with np.load(path) as f:
            g = {k: f[k] for k in f.files}
        lines, _ = make_lines(B, L, seed, voc_size=V)
        enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
        enc = m.encode(enc_in)
        assert np.allclose(enc[0][:NTENS], g['enc_out'], rtol=2e-4, atol=2e-6), name
        assert np.allclose(np.stack(enc[1:-1])[:, :NTENS], g['enc_states'], rtol=2e-4, atol=2e-6), name
        p, states = np.zeros((B, V), np.float32), enc[1:]
        for s in range(3):
            p, states = m.step(p, enc[0], states)
            assert np.allclose(p[:NTENS], g['step%d_probs' % s], rtol=2e-4, atol=2e-6), (name, s)
            assert np.allclose(states[-1][:NTENS], g['step%d_align' % s], rtol=2e-4, atol=2e-6), (name, s)
        got = decode_batch_greedy(m, enc_in, return_indexes=True)
        for j in range(B):
            n = int(g['greedy_len'][j])
            assert np.array_equal(got[5][j, :n], g['greedy_idx'][j, :n]), (name, j)
            r = next(decode_sequence_beam(m, source_seq=enc_in[j]), ('', None, 0.0, None))
            assert r[0] == str(g['beam_text'][j]) and abs(r[2] - g['beam_score'][j]) < 1e-4, (name, j)


def test_keras_golden_hook_reads_the_schema_its_generator_writes(golden_dir, tmp_path, monkeypatch):
    """`test_oracle_matches_keras_goldens` has never seen a file (none can be produced here).  Feed it one in the schema
    of tests/golden/make_keras_goldens.py::run_case -- produced by the ORACLE standing in for the reference, with the
    generator's conventions (greedy characters only up to the end of the line + `greedy_len`, the decoder's
    (B,1,V)-shaped outputs squeezed the way the generator squeezes them) -- so that the hook works the first time a
    reference-generated file appears under tests/golden/keras/ or $CASV_GOLDEN_DIR.  Pins nothing about Keras."""
    from tests.golden.make_golden import CASES, NTENS
    name = 'd2_w64_v96'
    d, W, V, B, L, seed, es, N = CASES[name]
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    m = OracleModel(cfg, make_weights(cfg, emb_scale=es), batch_size=N)
    lines, idx = make_lines(B, L, seed, voc_size=V)
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
    out = {'idx': idx.astype(np.int32)}
    enc = m.encode(enc_in)
    out['enc_out'] = enc[0][:NTENS]
    out['enc_states'] = np.stack(enc[1:-1])[:, :NTENS]
    p, states = np.zeros((B, V), np.float32), enc[1:]
    for s in range(3):
        p, states = m.step(p, enc[0], states)
        out['step%d_probs' % s] = p[:, None, :][:NTENS, -1]                 # generator: res[0] is (B,1,V)
        out['step%d_states' % s] = np.stack(states[:-1])[:, :NTENS]
        out['step%d_align' % s] = states[-1][:NTENS]
    g = decode_batch_greedy(m, enc_in, return_indexes=True)
    c_i = m.mapping[0]
    gi = np.zeros((B, 2 * (L + 1)), np.int16)
    for j, text in enumerate(g[1]):
        gi[j, :len(text)] = [c_i[c] for c in text]
    out['greedy_idx'] = gi
    out['greedy_len'] = np.array([len(t) for t in g[1]], np.int32)
    out['greedy_scores'] = np.asarray(g[3], np.float64)
    texts, scores = [], []
    for j in range(B):
        r = next(decode_sequence_beam(m, source_seq=enc_in[j]), ('', None, 0.0, None))
        texts.append(r[0]); scores.append(r[2])
    out['beam_text'] = np.array(texts)
    out['beam_score'] = np.asarray(scores, np.float64)
    np.savez_compressed(str(tmp_path / (name + '.npz')), **out)
    monkeypatch.setenv('CASV_GOLDEN_DIR', str(tmp_path))
    test_oracle_matches_keras_goldens(golden_dir)                           # must not skip and must not fail
    # ... and a file that disagrees is reported, not skipped over
    out['step1_probs'] = out['step1_probs'] * 1.01
    np.savez_compressed(str(tmp_path / (name + '.npz')), **out)
    with pytest.raises(AssertionError):
        test_oracle_matches_keras_goldens(golden_dir)


def test_deep_bidirectional_encoder_matches_torch():
    """deep_bidirectional_encoder (seq2seq.py:246-281) against torch BiLSTMs in float64: every layer bidirectional, fed the Lambda's
    "cross sum" of the layer below -- x + pairwise-reversed x, i.e. both features 2k and 2k+1 become x[2k] + x[2k+1] --, the backward
    final states handed on, enc_out 2W wide."""
    from oracle.model import cross_sum
    cfg = ModelConfig(depth=3, width=32, voc_size=40, deep_bidirectional_encoder=True)
    assert cfg.ctx_width == 64
    w = make_weights(cfg, dtype=np.float64, emb_scale=8.0)
    assert w['enc3_bw_K'].shape == (64, 128) and w['dec3_K'].shape == (32 + 64, 128) and w['att_U'].shape == (64, 32)
    x = np.arange(12.0).reshape(1, 2, 6)
    assert np.array_equal(cross_sum(x)[0, 0], [1, 1, 5, 5, 9, 9])
    lines, idx = make_lines(5, 17, 3, voc_size=40)
    xin = np.eye(40)[idx]
    out = encode(cfg, w, xin)
    with torch.no_grad():
        y = torch.tensor(xin @ w['E'])
        states = []
        for n in (1, 2, 3):
            if n > 1:
                y = y + y.reshape(y.shape[:-1] + (y.shape[-1] // 2, 2)).flip(-1).reshape(y.shape)
            layer = _torch_lstm(w['enc%d_fw_K' % n], w['enc%d_fw_R' % n], w['enc%d_fw_b' % n], True,
                                w['enc%d_bw_K' % n], w['enc%d_bw_R' % n], w['enc%d_bw_b' % n])
            y, (h, c) = layer(y)
            states += [h[1].numpy(), c[1].numpy()]
    assert out[0].shape == (5, 18, 64) and np.allclose(out[0], y.numpy(), atol=1e-10)
    for got, want in zip(out[1:7], states):
        assert np.allclose(got, want, atol=1e-10)
