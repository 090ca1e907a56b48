"""Train step (kt:195 train_on_batch / kt:407 test_on_batch) on the device against the oracle, and the
facade's train() end to end on a small copy task."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
from oracle.decode import OracleModel, correct_lines
from oracle.train import forward_backward, adam_step


def _idx(a):
    return np.where(a.any(axis=2), a.argmax(axis=2), -1).astype(np.int32)


@pytest.mark.parametrize('d,W,V,B,L,es,with_masks', [(1, 32, 40, 4, 9, 3.0, False), (2, 32, 40, 4, 9, 3.0, True),
                                                     (3, 64, 96, 8, 12, 6.0, True), (4, 64, 96, 6, 10, 8.0, False),
                                                     # widths that are no multiple of 32 (dead-unit padding, engine.py)
                                                     (1, 20, 24, 3, 7, 3.0, True), (2, 50, 40, 4, 9, 4.0, True),
                                                     # three / five unit groups per K share of the fused backward step
                                                     (2, 96, 40, 5, 8, 4.0, True), (3, 160, 40, 3, 6, 4.0, False),
                                                     # whole column tiles of 128 units: the backward recurrences are persistent too
                                                     (3, 128, 40, 37, 7, 4.0, True), (2, 256, 48, 5, 6, 4.0, False)])
@pytest.mark.parametrize('path', ['fused', 'stepwise'])
def test_train_step_matches_oracle(d, W, V, B, L, es, with_masks, path):
    """fused: the recurrences as one launch per pair of layers (train_persist.hip; backward: train_persist_bwd.hip at widths of
    whole 128-unit column tiles, else the cell's backward inside each step's data GEMM, gemm_bwd.hip); stepwise: one launch per
    time step and per operation."""
    _train_step_case(d, W, V, B, L, es, with_masks, path, {})


@pytest.mark.parametrize('flags', [dict(residual_connections=True), dict(bridge_dense=True),
                                   dict(residual_connections=True, bridge_dense=True)])
@pytest.mark.parametrize('d,W,V,B,L,es,with_masks', [(4, 64, 96, 6, 10, 8.0, True), (3, 128, 40, 37, 7, 4.0, True), (2, 50, 40, 4, 9, 4.0, True),
                                                     (1, 32, 40, 4, 9, 3.0, False), (5, 96, 40, 5, 8, 4.0, False)])
@pytest.mark.parametrize('path', ['fused', 'stepwise'])
def test_train_step_with_optional_topologies_matches_oracle(d, W, V, B, L, es, with_masks, path, flags):
    """residual_connections (seq2seq.py:284-291: encoder layers >= 3; :359-360: decoder layers >= 2 and the projection's input) and
    bridge_dense (:299-301) in the train step: loss, every gradient (the Dense layers' included), test_on_batch and three Adam
    updates against the oracle, whose backward is checked against torch autograd with the same flags (tests/test_oracle_train.py)."""
    _train_step_case(d, W, V, B, L, es, with_masks, path, flags)


@pytest.mark.parametrize('flags', [dict(deep_bidirectional_encoder=True),
                                   dict(deep_bidirectional_encoder=True, residual_connections=True, bridge_dense=True)])
@pytest.mark.parametrize('d,W,V,B,L,es,with_masks', [(4, 64, 96, 6, 10, 8.0, True), (3, 128, 40, 37, 7, 4.0, True), (2, 96, 40, 5, 8, 4.0, True),
                                                     (5, 96, 40, 5, 8, 4.0, False)])
@pytest.mark.parametrize('path', ['fused', 'stepwise'])
def test_train_step_with_a_deep_bidirectional_encoder_matches_oracle(d, W, V, B, L, es, with_masks, path, flags):
    """deep_bidirectional_encoder (seq2seq.py:246-281) in the train step: every encoder layer a BiLSTM on the cross sum of the layer
    below (2W-wide masks, inputs and attended sequence), alone and with the other two optional topologies."""
    _train_step_case(d, W, V, B, L, es, with_masks, path, flags)


def _train_step_case(d, W, V, B, L, es, with_masks, path, flags):
    from cor_asv_ann_amd.engine import HipEngine
    cfg = ModelConfig(depth=d, width=W, voc_size=V, **flags)
    w = make_weights(cfg, emb_scale=es)
    rng = np.random.default_rng(4)
    for k in w:
        if k.endswith('_b') or k in ('att_bUW', 'att_bv'):
            w[k] = (w[k] + rng.normal(size=w[k].shape) * 0.2).astype(np.float32)
    om = OracleModel(cfg, w)
    src, sidx = make_lines(B, L, 1, voc_size=V)
    tgt, _ = make_lines(B, L, 2, voc_size=V)
    tgt[1] = tgt[1][:L // 2] + '\n'                         # ragged targets: padded steps have weight 0
    enc_in, dec_in, dec_out, wts = vectorize_lines(om, src, tgt)
    C = cfg.ctx_width
    masks = None
    if with_masks:
        keep = lambda shape: ((rng.random(shape) > 0.2) / 0.8).astype(np.float32)
        masks = {'enc': [keep(2 * W if (n == 0 or cfg.deep_bidirectional_encoder) else W) for n in range(d)], 'dec': [keep(W) for _ in range(d - 1)],
                 'cell': keep((B, W + C))}
    loss, grads, aux = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, masks)
    eng = HipEngine(d, W, V, **flags)
    eng.set_weights(w)
    eng.set_option('persistent', -1 if path == 'fused' else 0)
    eng.set_option('fused_backward', 1 if path == 'fused' else 0)
    eng.train_begin()
    gl, gn = eng.train_step(sidx, None, _idx(dec_in), _idx(dec_out), wts, masks, mode=2)
    onorm = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    assert abs(gl - loss) < 2e-5 * abs(loss) and abs(gn - onorm) < 1e-4 * onorm
    gg = eng.train_gradients()
    for k in grads:
        scale = max(np.abs(grads[k]).max(), 1e-6 * onorm)
        assert np.abs(gg[k] - grads[k]).max() < 2e-3 * scale + 1e-7, k
    # test_on_batch: no regulariser, no dropout
    el, _ = eng.train_step(sidx, None, _idx(dec_in), _idx(dec_out), wts, None, mode=0)
    _, _, a0 = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, None, want_grads=False)
    assert abs(el - a0['loss_ce']) < 2e-5 * abs(a0['loss_ce'])
    # three updates with Adam(clipnorm=5)
    st = {'t': 0, 'm': {}, 'v': {}}
    w2 = {k: v.copy() for k, v in w.items()}
    for _ in range(3):
        lo, g2, _ = forward_backward(cfg, w2, enc_in, dec_in, dec_out, wts, masks)
        adam_step(w2, g2, st)
        lg, _ = eng.train_step(sidx, None, _idx(dec_in), _idx(dec_out), wts, masks, mode=1)
        assert abs(lg - lo) < 5e-5 * abs(lo)
    eng.train_end()
    wg = eng.get_weights()
    for k in w2:
        assert np.abs(wg[k] - w2[k]).max() < 5e-6, k
    eng.close()


@pytest.fixture(scope='module')
def copy_model(tmp_path_factory):
    """A small model trained on a copy task through the facade's train() (cor-asv-ann-train's path), shared by the tests
    below: the functional, well-conditioned fixture for beam search and evaluation."""
    import os
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    tmp_path = tmp_path_factory.mktemp('copytask')
    cwd = os.getcwd()
    os.chdir(tmp_path)                                       # checkpoints go to the CWD like the reference's
    try:
        rng = np.random.default_rng(0)
        alphabet = 'abcdefghij '
        lines = [''.join(rng.choice(list(alphabet), size=rng.integers(5, 12))) for _ in range(1600)]
        (tmp_path / 'train.tsv').write_text(''.join('%s\t%s\n' % (l, l) for l in lines))
        s2s = Sequence2Sequence()
        s2s.depth, s2s.width, s2s.batch_size, s2s.epochs, s2s.dropout = 2, 64, 32, 50, 0.0
        s2s._rng = np.random.default_rng(1)
        s2s.configure()
        s2s.train([str(tmp_path / 'train.tsv')])
    finally:
        os.chdir(cwd)
    return s2s, lines, tmp_path


def test_facade_train_copy_task(copy_model):
    """cor-asv-ann-train's path: map_files -> epochs of train_on_batch -> validation -> early stopping ->
    trained weights usable by correct_lines; afterwards the trained model is the functional fixture for the
    beam search (non-degenerate, well-conditioned distributions): GPU == oracle on it."""
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    s2s, lines, tmp_path = copy_model
    assert s2s.status == 2
    hist = s2s.history
    assert min(h['val_loss'] for h in hist) < 0.85 * hist[0]['val_loss'], hist
    test = [l + '\n' for l in lines[:8]]
    s2s.batch_size = 4
    got = s2s.correct_lines(test, fast=False, greedy=False)
    cfg = ModelConfig(depth=2, width=64, voc_size=s2s.voc_size)
    om = OracleModel(cfg, s2s.get_weights(), mapping=s2s.mapping, batch_size=4)
    want = correct_lines(om, test, fast=False, greedy=False)
    assert got[0] == want[0]
    assert np.allclose(got[2], want[2], atol=1e-4)
    fast = s2s.correct_lines(test, fast=True, greedy=True)
    wfast = correct_lines(om, test, fast=True, greedy=True)
    assert fast[0] == wfast[0]
    s2s.save(str(tmp_path / 'm.h5'))                         # the reference's container
    other = Sequence2Sequence()
    other.load_config(str(tmp_path / 'm.h5')); other.configure(); other.load_weights(str(tmp_path / 'm.h5'))
    other.batch_size = 4
    assert other.correct_lines(test, fast=True, greedy=True)[0] == fast[0]
    # per-epoch checkpoints in the reference's naming and container (seq2seq.py:621-622); the best one is what train() kept
    import glob
    from cor_asv_ann_amd import hdf5
    ckpts = sorted(glob.glob(str(tmp_path / 'model.ckpt.weights-*.h5')))
    assert len(ckpts) == len([h for h in hist if np.isfinite(h['val_loss'])]) and all(hdf5.is_hdf5(c) for c in ckpts)
    best = min(range(len(hist)), key=lambda i: hist[i]['val_loss'])
    third = Sequence2Sequence()
    third.load_config(ckpts[best]); third.configure(); third.load_weights(ckpts[best])
    for k, v in s2s.get_weights().items():
        assert np.array_equal(third.get_weights()[k], v), k


def test_evaluate_runs_through_the_device(copy_model, caplog):
    """f4: `evaluate()` (seq2seq.py:651-754) un-stubbed -- greedy and beamed decoding on the HIP path, this package's metrics
    on top.  Expected figures: the same metrics over the ORACLE's decoded strings and scores for the same lines (OCR
    column: the source lines themselves)."""
    import logging
    import math
    from cor_asv_ann_amd.metrics import Alignment, Edits, splitwords
    s2s, lines, tmp_path = copy_model
    rng = np.random.default_rng(7)
    pairs = []
    for text in lines[100:111]:                              # 11 lines: the last batch of 4 is partly padding
        chars = list(text)
        for k in rng.choice(len(chars), size=max(1, len(chars) // 6), replace=False):
            chars[k] = str(rng.choice(list('abcdefghij')))   # OCR errors the model may or may not repair
        pairs.append((''.join(chars), text))
    tsv = tmp_path / 'eval.tsv'
    tsv.write_text(''.join('%s\t%s\n' % p for p in pairs))
    s2s.batch_size = 4
    s2s.logger = logging.getLogger('evalgpu')
    with caplog.at_level(logging.INFO, logger='evalgpu'):
        s2s.evaluate([str(tsv)], fast=False)
    text = '\n'.join(r.getMessage() for r in caplog.records)

    cfg = ModelConfig(depth=2, width=64, voc_size=s2s.voc_size)
    om = OracleModel(cfg, s2s.get_weights(), mapping=s2s.mapping, batch_size=4)
    src = [a + '\n' for a, _ in pairs]
    tgt = [b + '\n' for _, b in pairs]
    greedy, beamed = ([], []), ([], [])
    for b0 in range(0, len(src), 4):                         # the batches gen_lines forms (the last one padded with '')
        batch = src[b0:b0 + 4] + [''] * (4 - len(src[b0:b0 + 4]))
        for res, kw in ((greedy, dict(fast=False, greedy=True)), (beamed, dict(fast=False, greedy=False))):
            out = correct_lines(om, batch, **kw)
            res[0].extend(out[0][:len(src[b0:b0 + 4])]); res[1].extend(out[2][:len(src[b0:b0 + 4])])
    cols = {'OCR:   ': (src, None), 'greedy:': greedy, 'beamed:': beamed}
    for label, (outs, scores) in cols.items():
        c, w = Edits(s2s.logger), Edits(s2s.logger)
        ca, wa = Alignment(0, logger=s2s.logger), Alignment(0, logger=s2s.logger)
        for o, t in zip(outs, tgt):
            c.add(*ca.get_adjusted_distance(o, t, normalization='historic_latin', gtlevel=1), o, t)
            w.add(*wa.get_adjusted_distance(splitwords(o), splitwords(t), normalization='historic_latin', gtlevel=1), o, t)
        assert 'CER %s %.3f±%.3f' % (label, c.mean, math.sqrt(c.varia)) in text, (label, text)
        assert 'WER %s %.3f±%.3f' % (label, w.mean, math.sqrt(w.varia)) in text, (label, text)
        if scores is not None:
            ppl = math.exp(sum(scores) / max(c.length, 1))
            got = float(text.split('ppl %s ' % label.strip())[1].split()[0])
            assert abs(got - ppl) < 2e-3 * ppl, (label, got, ppl)
    assert 'finished ' in text


def test_frozen_layers_do_not_train():
    """After a transfer from a shallower model the hidden layers taken over stay fixed (`trainable = False`,
    seq2seq.py:1206-1211): they receive no update and do not count in the clipping norm; everything else trains as
    the oracle does with their gradients left out."""
    from cor_asv_ann_amd.engine import HipEngine
    d, W, V, B, L = 2, 32, 40, 4, 9
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    w = make_weights(cfg, emb_scale=3.0)
    om = OracleModel(cfg, w)
    src, sidx = make_lines(B, L, 1, voc_size=V)
    tgt, _ = make_lines(B, L, 2, voc_size=V)
    enc_in, dec_in, dec_out, wts = vectorize_lines(om, src, tgt)
    frozen = ('enc1_', 'dec1_')
    is_frozen = lambda k: k.startswith(frozen)
    eng = HipEngine(d, W, V)
    eng.set_weights(w)
    eng.train_begin(frozen=frozen)
    _, grads, _ = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, None)
    want_norm = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for k, g in grads.items() if not is_frozen(k)))
    _, gn = eng.train_step(sidx, None, _idx(dec_in), _idx(dec_out), wts, None, mode=2)
    assert abs(gn - want_norm) < 1e-4 * want_norm
    st = {'t': 0, 'm': {}, 'v': {}}
    w2 = {k: v.copy() for k, v in w.items()}
    for _ in range(2):
        _, g2, _ = forward_backward(cfg, w2, enc_in, dec_in, dec_out, wts, None)
        adam_step(w2, g2, st, frozen=frozen)
        eng.train_step(sidx, None, _idx(dec_in), _idx(dec_out), wts, None, mode=1)
    eng.train_end()
    got = eng.get_weights()
    for k in w:
        if is_frozen(k):
            assert np.array_equal(got[k], w[k]), k
        else:
            assert not np.array_equal(got[k], w[k]), k
            assert np.abs(got[k] - w2[k]).max() < 5e-6, k
    eng.close()


def test_train_step_with_confidence_inputs():
    """Encoder inputs that are not one-hot (probability lines, confusion networks: several weighted alternatives per
    position, seq2seq.py:1067-1093): the embedding is a weighted sum of rows, and its gradient scatters back with the
    same weights."""
    from cor_asv_ann_amd.engine import HipEngine
    d, W, V, B, L = 2, 32, 40, 4, 8
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    w = make_weights(cfg, emb_scale=3.0)
    om = OracleModel(cfg, w)
    rng = np.random.default_rng(8)
    src, sidx = make_lines(B, L, 1, voc_size=V)
    tgt, _ = make_lines(B, L, 2, voc_size=V)
    _, dec_in, dec_out, wts = vectorize_lines(om, src, tgt)
    T, A = L + 1, 3
    idx = np.full((B, T, A), -1, np.int32)
    val = np.zeros((B, T, A), np.float32)
    idx[:, :, 0] = sidx
    val[:, :, 0] = rng.uniform(0.3, 1.0, (B, T))
    alt = rng.random((B, T)) < 0.5
    idx[:, :, 1] = np.where(alt, rng.integers(2, V, (B, T)), -1)
    val[:, :, 1] = np.where(alt, rng.uniform(0.05, 0.5, (B, T)), 0.0)
    idx[0, 2, 2] = 5; val[0, 2, 2] = 0.1
    idx[1, 3, :] = -1; val[1, 3, :] = 0.0                     # an all-zero position inside a line
    enc_in = np.zeros((B, T, V), np.float32)
    for b in range(B):
        for t in range(T):
            for a in range(A):
                if idx[b, t, a] >= 0:
                    enc_in[b, t, idx[b, t, a]] += val[b, t, a]
    loss, grads, _ = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, None)
    eng = HipEngine(d, W, V)
    eng.set_weights(w)
    eng.train_begin()
    gl, gn = eng.train_step(idx, val, _idx(dec_in), _idx(dec_out), wts, None, mode=2)
    onorm = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    assert abs(gl - loss) < 2e-5 * abs(loss) and abs(gn - onorm) < 1e-4 * onorm
    gg = eng.train_gradients()
    for k in grads:
        scale = max(np.abs(grads[k]).max(), 1e-6 * onorm)
        assert np.abs(gg[k] - grads[k]).max() < 2e-3 * scale + 1e-7, k
    eng.train_end()
    eng.close()


def test_c4_full_size_step_equals_oracle(golden_dir):
    """BASELINE configs[3] at full size (depth 4, width 512, V 256, 512 lines x 100 characters, dropout masks): the
    shapes at which the launcher picks the 128x128 two-wave-group split-K recurrent GEMMs, the atomics split-K and the
    whole-sequence weight-gradient GEMMs.  Expected values come from the oracle through a committed fixture
    (tests/golden/make_c4_golden.py; the numpy oracle needs minutes for this step)."""
    import os
    from cor_asv_ann_amd.engine import HipEngine
    from tests.golden.make_c4_golden import DEPTH, WIDTH, VOC, c4_inputs, sample_positions
    with np.load(os.path.join(golden_dir, 'c4_train_step.npz')) as f:
        g = {k: f[k] for k in f.files}
    cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
    w = make_weights(cfg, emb_scale=4.0)
    sidx, dec_in, dec_out, wts, masks = c4_inputs()
    eng = HipEngine(DEPTH, WIDTH, VOC)
    eng.set_weights(w)
    eng.train_begin()
    loss, norm = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=2)
    assert abs(loss - float(g['loss'])) < 2e-5 * abs(float(g['loss'])), (loss, float(g['loss']))
    assert abs(norm - float(g['grad_norm'])) < 1e-4 * float(g['grad_norm']), (norm, float(g['grad_norm']))
    grads = eng.train_gradients()
    onorm = float(g['grad_norm'])
    for k, got in grads.items():
        flat = got.ravel()
        want = g['sample/' + k]
        scale = max(float(g['max/' + k]), 1e-6 * onorm)
        assert np.abs(flat[sample_positions(k, flat.size)] - want).max() < 2e-3 * scale + 1e-7, k
        tnorm = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        assert abs(tnorm - float(g['norm/' + k])) < 1e-3 * float(g['norm/' + k]) + 1e-6 * onorm, k
    # a second evaluation of the same step: float atomics may reorder sums, nothing else may move
    loss2, norm2 = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=2)
    assert abs(loss2 - loss) < 1e-6 * abs(loss) and abs(norm2 - norm) < 1e-5 * norm
    # ... and with one launch per time step instead of the persistent launches: the recurrences of the plain layers give the same
    # bits either way; the attention cell's persistent form sums its gate pre-activations over [h | ctx] instead of [ctx | h] and
    # the attention query in four k quarters instead of two split-K shares: the same forward pass to fp32 rounding
    eng.set_option('persistent', 0)
    loss3, norm3 = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=2)
    assert abs(loss3 - loss) < 1e-6 * abs(loss) and abs(norm3 - norm) < 1e-5 * norm
    # ... and with the cell's backward as a launch of its own in front of every backward step's data GEMM
    eng.set_option('fused_backward', 0)
    loss4, norm4 = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=2)
    assert abs(loss4 - loss) < 1e-6 * abs(loss) and abs(norm4 - norm) < 2e-5 * norm
    grads4 = eng.train_gradients()
    for k, got in grads.items():
        scale = max(float(g['max/' + k]), 1e-6 * onorm)
        assert np.abs(grads4[k] - got).max() < 1e-4 * scale + 1e-7, k
    eng.set_option('fused_backward', 1)
    # ... and with this library's own GEMM kernel for the plain whole-sequence contractions instead of the vendor's (hipBLASLt,
    # loaded at run time where present: input projections of all time steps and their data gradients)
    eng.set_option('persistent', -1)
    eng.set_option('vendor_gemm', 0)
    loss5, norm5 = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=2)
    assert abs(loss5 - loss) < 1e-6 * abs(loss) and abs(norm5 - norm) < 2e-5 * norm
    grads5 = eng.train_gradients()
    for k, got in grads.items():
        scale = max(float(g['max/' + k]), 1e-6 * onorm)
        assert np.abs(grads5[k] - got).max() < 1e-4 * scale + 1e-7, k
    eng.set_option('vendor_gemm', 1)
    eval_a, _ = eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=0)
    eng.set_option('persistent', -1)
    eval_b, _ = eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=0)
    assert abs(eval_a - eval_b) < 1e-6 * abs(eval_a)
    eng.train_end()
    eng.close()


def test_fused_backward_step_against_the_host(tmp_path):
    """One fused backward step (cell backward + data GEMM, gemm_bwd.hip) on random data against a double-precision host
    computation of the same step: every path of the kernel's software pipeline (1, 2, 3 and more unit groups per K share,
    partial row blocks and column tiles, split-K atomics at the train step's full size)."""
    import os, re, shutil, subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'bwd_step_check')
    subprocess.check_call([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-I', os.path.join(root, 'include'),
                           os.path.join(root, 'tests', 'native', 'bwd_step_check.hip'),
                           os.path.join(root, 'cor_asv_ann_amd', 'csrc', 'gemm_bwd.hip'), '-o', exe])
    for rows, width, n in [(4, 32, 32), (4, 32, 96), (40, 64, 64), (33, 96, 96), (7, 160, 160), (100, 256, 256),
                           (512, 512, 512), (512, 512, 1024)]:
        out = subprocess.run([exe, str(rows), str(width), str(n)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        m = re.search(r'dz err (\S+)\s+dc err (\S+)\s+out err (\S+) \(scale (\S+)\)', out.stdout)
        assert m, out.stdout
        dz, dc, err, scale = map(float, m.groups())
        assert dz < 2e-6 and dc < 2e-6 and err < 2e-6 * max(scale, 1.0) * (4 * width) ** 0.5, out.stdout


def test_persistent_recurrences_under_uneven_load():
    """The hand-offs inside the persistent forward / backward recurrences must not depend on timing: train steps of random
    shapes while a second model handle keeps the CUs busy with beamed decodes on another stream, each step evaluated with the
    persistent launches and with one launch per time step, and twice.  Loss and gradient norm agree up to the order of float
    atomic sums (at these sizes the whole-sequence input GEMMs of the forward pass split K three or four ways, so even two
    identical evaluations may differ in the last bit of a row's loss; at configs[3]'s size they do not and
    test_c4_full_size_step_equals_oracle compares the forward passes exactly) -- a stale or early read would show as garbage.
    A persistent launch that cannot become resident beside the foreign kernels gives up and the step is redone stepwise: the
    results must not show which of the two happened."""
    import threading
    import time
    from cor_asv_ann_amd.engine import HipEngine
    stop = []

    def background():
        cfg = ModelConfig(depth=2, width=256, voc_size=64)
        e = HipEngine(2, 256, 64)
        e.set_weights(make_weights(cfg, emb_scale=32.0))
        _, bidx = make_lines(96, 30, 1, voc_size=64)
        while not stop:
            e.encode(bidx)
            e.decode_beam(batch_size=8)
        e.close()

    th = threading.Thread(target=background)
    th.start()
    try:
        rng = np.random.default_rng(11)
        t0, cases = time.time(), 0
        while time.time() - t0 < 10.0 or cases < 8:
            d = int(rng.integers(2, 5)); W = int(rng.choice([128, 256, 512])); V = int(rng.choice([40, 96]))
            B = int(rng.integers(1, 300)); L = int(rng.integers(3, 24))
            cfg = ModelConfig(depth=d, width=W, voc_size=V)
            w = make_weights(cfg, seed=int(rng.integers(1, 1 << 30)), emb_scale=float(rng.choice([3., 8.])))
            _, sidx = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
            _, tidx = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
            U = L + 2
            dec_in = np.full((B, U), -1, np.int32); dec_out = np.full((B, U), -1, np.int32)
            dec_in[:, 1:L + 2] = tidx; dec_out[:, :L + 1] = tidx
            wts = (dec_out >= 0).astype(np.float32)
            eng = HipEngine(d, W, V)
            eng.set_weights(w)
            eng.train_begin()
            # two different batches take turns, so that a value left over from the evaluation before is a wrong value
            _, sidx2 = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
            seen = {}
            for rep in range(2):
                for p in (-1, 0):
                    eng.set_option('persistent', p)
                    for which, src in enumerate((sidx, sidx2)):
                        loss, norm = eng.train_step(src, None, dec_in, dec_out, wts, None, mode=2)
                        assert np.isfinite(loss) and np.isfinite(norm)
                        if (p, which) in seen:
                            assert abs(loss - seen[p, which][0]) < 1e-8 * abs(loss) and abs(norm - seen[p, which][1]) < 2e-5 * norm, \
                                ('not reproducible', p, d, W, V, B, L)
                        seen[p, which] = (loss, norm)
            for which in (0, 1):
                assert abs(seen[-1, which][0] - seen[0, which][0]) < 1e-8 * abs(seen[0, which][0]), ('persistent != stepwise (forward)', d, W, V, B, L)
                assert abs(seen[-1, which][1] - seen[0, which][1]) < 2e-5 * seen[0, which][1], ('persistent != stepwise (gradients)', d, W, V, B, L)
            assert seen[0, 0][0] != seen[0, 1][0]
            eng.train_end()
            eng.close()
            cases += 1
    finally:
        stop.append(1)
        th.join()


_ONE_STEP = '''
import sys, json
import numpy as np
sys.path.insert(0, %r)
from oracle import ModelConfig, make_weights, make_lines
from cor_asv_ann_amd.engine import HipEngine
d, W, V, B, L = 2, 512, 40, 70, 9
cfg = ModelConfig(depth=d, width=W, voc_size=V)
w = make_weights(cfg, seed=5, emb_scale=4.0)
_, sidx = make_lines(B, L, 21, voc_size=V)
_, tidx = make_lines(B, L, 22, voc_size=V)
U = L + 2
dec_in = np.full((B, U), -1, np.int32); dec_out = np.full((B, U), -1, np.int32)
dec_in[:, 1:L + 2] = tidx; dec_out[:, :L + 1] = tidx
wts = (dec_out >= 0).astype(np.float32)
eng = HipEngine(d, W, V); eng.set_weights(w); eng.train_begin()
out = [eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=2) for _ in range(2)]
eng.train_end(); eng.close()
print(json.dumps(out))
'''


def test_attention_backward_beside_the_recurrence_or_inside_it():
    """The attention cell's backward recurrence runs as two launches side by side (the recurrence and, on a second stream, the
    attention backward of its samples: train_persist_topb.hip, split_a) -- or, after a give-up or with CASV_TOPB_SPLIT=0, as the one
    launch it was: the same loss and gradient norm either way (sums in the same order; the float atomics of d_enc / du may differ in
    the last bits of the norm)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for split in ('1', '0'):
        env = dict(os.environ, CASV_TOPB_SPLIT=split)
        out = subprocess.run([sys.executable, '-c', _ONE_STEP % root], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        res[split] = json.loads(out.stdout.strip().splitlines()[-1])
    for (l1, n1), (l0, n0) in zip(res['1'], res['0']):
        assert np.isfinite(l1) and abs(l1 - l0) < 1e-8 * abs(l0), (l1, l0)
        assert abs(n1 - n0) < 2e-5 * n0, (n1, n0)


def test_two_train_sessions_share_the_gpu(capfd):
    """Two model handles run train steps at the same time, each with persistent recurrences that want every CU: whichever way the
    hardware interleaves their workgroups -- one launch after the other, or both partly resident, waiting for workgroups that cannot
    start (then both give up after their bounded wait and redo the step with per-step launches) -- every step must return what
    the same step returns alone."""
    import threading
    from cor_asv_ann_amd.engine import HipEngine
    d, W, V, B, L = 2, 512, 64, 512, 16
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    w = make_weights(cfg, emb_scale=4.0)
    jobs = []
    for k in range(2):
        _, sidx = make_lines(B, L, 31 + k, voc_size=V)
        _, tidx = make_lines(B, L, 41 + k, voc_size=V)
        U = L + 2
        dec_in = np.full((B, U), -1, np.int32); dec_out = np.full((B, U), -1, np.int32)
        dec_in[:, 1:L + 2] = tidx; dec_out[:, :L + 1] = tidx
        jobs.append((sidx, dec_in, dec_out, (dec_out >= 0).astype(np.float32)))
    engs = [HipEngine(d, W, V) for _ in range(2)]
    for e in engs:
        e.set_weights(w)
        e.train_begin()
    alone = [engs[k].train_step(jobs[k][0], None, jobs[k][1], jobs[k][2], jobs[k][3], None, mode=2) for k in range(2)]
    got = [[], []]
    start = threading.Barrier(2)

    def work(k):
        start.wait()
        for _ in range(8):
            got[k].append(engs[k].train_step(jobs[k][0], None, jobs[k][1], jobs[k][2], jobs[k][3], None, mode=2))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        for loss, norm in got[k]:
            assert abs(loss - alone[k][0]) < 1e-8 * abs(alone[k][0]) and abs(norm - alone[k][1]) < 2e-5 * alone[k][1]
    for e in engs:
        e.train_end()
        e.close()
    print(capfd.readouterr().err[-400:])


def test_a_recurrence_that_loses_a_workgroup_gives_up_and_the_step_falls_back(capfd):
    """What a GPU shared with another process's persistent kernel can do to a persistent recurrence: a workgroup its peers wait for
    does not run.  Option "persistent" = 2 makes one workgroup of the forward recurrences leave without handing on; the others
    give up after their bounded wait (50 ms), the launch drains, the host sees the abort word before anything is updated and
    redoes the step with per-step launches: same results, a message, and no second attempt for the next steps."""
    from cor_asv_ann_amd.engine import HipEngine
    d, W, V, B, L = 2, 256, 64, 70, 5
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    w = make_weights(cfg, emb_scale=4.0)
    _, sidx = make_lines(B, L, 3, voc_size=V)
    _, tidx = make_lines(B, L, 4, voc_size=V)
    U = L + 2
    dec_in = np.full((B, U), -1, np.int32); dec_out = np.full((B, U), -1, np.int32)
    dec_in[:, 1:L + 2] = tidx; dec_out[:, :L + 1] = tidx
    wts = (dec_out >= 0).astype(np.float32)
    eng = HipEngine(d, W, V)
    eng.set_weights(w)
    eng.train_begin()
    eng.set_option('persistent', 0)
    want = eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=2)
    capfd.readouterr()
    eng.set_option('persistent', 2)
    got = eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=2)
    assert 'gave up waiting' in capfd.readouterr().err
    assert abs(got[0] - want[0]) < 1e-8 * abs(want[0]) and abs(got[1] - want[1]) < 2e-5 * want[1]
    again = eng.train_step(sidx, None, dec_in, dec_out, wts, None, mode=2)         # backed off: no launch, no message
    assert 'gave up waiting' not in capfd.readouterr().err
    assert abs(again[0] - want[0]) < 1e-8 * abs(want[0]) and abs(again[1] - want[1]) < 2e-5 * want[1]
    eng.train_end()
    eng.close()


def test_attention_sums_behind_the_recurrence_equal_the_atomics(tmp_path):
    """The persistent attention-cell backward with d_enc (CASV_ATTN_DEFER=1, the default) or d_enc and du (3) summed behind the
    recurrence gives the gradients of the variant that adds them by float atomics inside it (0): every tensor to 1e-4 of its
    largest entry.  The switch is read once per process, so each variant runs in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
from oracle.decode import OracleModel
from cor_asv_ann_amd.engine import HipEngine
cfg = ModelConfig(depth=2, width=128, voc_size=64)
w = make_weights(cfg, emb_scale=3.0)
om = OracleModel(cfg, w)
src, _ = make_lines(96, 23, 5, voc_size=64)
tgt, _ = make_lines(96, 19, 6, voc_size=64)
enc_in, dec_in, dec_out, wts = vectorize_lines(om, src, tgt)
idx_of = lambda a: np.where(a.any(axis=2), a.argmax(axis=2), -1).astype(np.int32)
eng = HipEngine(2, 128, 64)
eng.set_weights(w)
eng.train_begin()
loss, norm = eng.train_step(idx_of(enc_in), None, idx_of(dec_in), idx_of(dec_out), wts, None, mode=2)
g = eng.train_gradients()
np.savez(sys.argv[1], loss=loss, norm=norm, **g)
''' % root
    out = {}
    for mode in ('0', '1', '3'):
        path = str(tmp_path / ('g%s.npz' % mode))
        env = dict(os.environ, CASV_ATTN_DEFER=mode)
        p = subprocess.run([sys.executable, '-c', script, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        with np.load(path) as f:
            out[mode] = {k: f[k] for k in f.files}
    ref = out['0']
    for mode in ('1', '3'):
        got = out[mode]
        assert abs(float(got['loss']) - float(ref['loss'])) < 1e-6 * abs(float(ref['loss']))
        assert abs(float(got['norm']) - float(ref['norm'])) < 2e-5 * float(ref['norm'])
        for k in ref:
            if k in ('loss', 'norm'):
                continue
            assert np.abs(got[k] - ref[k]).max() < 1e-4 * max(float(np.abs(ref[k]).max()), 1e-6 * float(ref['norm'])) + 1e-7, (mode, k)
