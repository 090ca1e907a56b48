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
