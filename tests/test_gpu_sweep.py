"""Randomised sweep of model shapes and beam parameters: the device search against the oracle, with device memory
poisoned at allocation (CASV_POISON=1) so that reads of unwritten memory cannot hide behind zero-filled pages.
A mismatch counts only where the oracle itself is well-conditioned (its fp32 and fp64 runs agree on the line)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ.setdefault('CASV_POISON', '1')      # read once by the library, at its first allocation

from oracle import ModelConfig, make_weights, make_lines
from oracle.decode import OracleModel, correct_lines


@pytest.mark.parametrize('seed,ncase,tile', [(1, 40, -1), (7, 40, 0), (11, 30, 2)])
def test_random_beam_configurations(seed, ncase, tile):
    """tile: -1 = the launcher's choice (32x128 tiles at these sizes), 0 = 128x128 tiles forced, 2 = 64x128, so that every GEMM
    kernels see the odd shapes of the sweep."""
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    rng = np.random.default_rng(seed)
    bad, total, excluded, skipped_cases = [], 0, 0, 0
    for case in range(ncase):
        d = int(rng.integers(1, 4)); W = int(rng.choice([32, 64, 96])); V = int(rng.choice([24, 64, 100, 257]))
        B = int(rng.integers(1, 7)); L = int(rng.integers(3, 20)); es = float(rng.choice([2., 6., 10., 14., 20.]))
        kw = dict(batch_size=int(rng.choice([1, 2, 3, 4, 8, 16])), beam_width_in=int(rng.choice([3, 15, 50])),
                  beam_threshold_in=float(rng.choice([0.05, 0.2, 0.6])),
                  rejection_threshold=float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.9])), beam_width_out=int(rng.choice([1, 4, 16])))
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        wseed = int(rng.integers(1, 10 ** 6))
        lines, _ = make_lines(B, L, wseed, voc_size=V)
        if B > 1 and rng.random() < 0.5:
            lines[0] = lines[0][:max(1, L // 2)] + '\n'          # ragged batch
        if rng.random() < 0.3:
            lines[-1] = '中' + lines[-1][1:]                  # unmapped character -> index 0
        res = {}
        for dt in (np.float32, np.float64):
            om = OracleModel(cfg, make_weights(cfg, seed=wseed, dtype=dt, emb_scale=es), **kw)
            try:
                res[dt] = correct_lines(om, lines, fast=False, greedy=False)
            except (IndexError, ValueError):      # the reference raises here; nothing to compare
                res[dt] = None
        if res[np.float32] is None:
            skipped_cases += 1
            continue
        w32 = make_weights(cfg, seed=wseed, emb_scale=es)
        s2s = Sequence2Sequence()
        s2s.depth, s2s.width = d, W
        s2s.mapping, s2s.voc_size = OracleModel(cfg, w32).mapping, V
        for k, v in kw.items():
            setattr(s2s, k, v)
        s2s.configure(); s2s.set_weights(w32); s2s.status = 2
        s2s._require_engine().set_option('tile', tile)         # process-wide
        try:
            got = s2s.correct_lines(lines, fast=False, greedy=False)
        finally:
            s2s.engine.set_option('tile', -1)
        want = res[np.float32]
        for j in range(B):
            total += 1
            ok = got[0][j] == want[0][j] and abs(got[2][j] - want[2][j]) < 1e-4
            conditioned = res[np.float64] is not None and res[np.float64][0][j] == want[0][j]
            excluded += not conditioned
            if not ok and conditioned:
                bad.append((case, j, dict(d=d, W=W, V=V, B=B, L=L, es=es, seed=wseed, **kw)))
        s2s.engine.close()
    # lines on which the oracle's own fp32 and fp64 searches disagree pin nothing: say how many there were and bound them
    print('sweep seed %d: %d lines compared, %d of them excluded as ill-conditioned (oracle fp32 != fp64), %d cases skipped '
          '(the reference raises)' % (seed, total, excluded, skipped_cases))
    assert total > 50 and not bad, bad
    assert excluded <= 0.10 * total, 'too many lines excluded for fp32/fp64 disagreement: %d of %d' % (excluded, total)
    # ... and cases in which the reference itself raises (the oracle does the same) compare nothing: a sweep must not pass on them
    assert skipped_cases <= 0.25 * ncase, 'too many cases skipped because the reference raises: %d of %d' % (skipped_cases, ncase)


def test_random_train_steps():
    """Loss and every gradient tensor of the device train step against the oracle over random shapes (odd batch
    sizes, ragged sources/targets, with and without dropout masks)."""
    from cor_asv_ann_amd.engine import HipEngine
    from oracle import vectorize_lines
    from oracle.train import forward_backward
    rng = np.random.default_rng(3)
    idx_of = lambda a: np.where(a.any(axis=2), a.argmax(axis=2), -1).astype(np.int32)
    for case in range(12):
        d = int(rng.integers(1, 5)); W = int(rng.choice([32, 64])); V = int(rng.choice([20, 50, 100])); B = int(rng.integers(1, 8))
        L = int(rng.integers(2, 14)); Lt = int(rng.integers(2, 14)); es = float(rng.choice([1., 3., 6.]))
        masks_on = rng.random() < 0.6
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        w = make_weights(cfg, seed=int(rng.integers(1, 10 ** 6)), emb_scale=es)
        for k in w:
            if k.endswith('_b') or k in ('att_bUW', 'att_bv'):
                w[k] = (w[k] + rng.normal(size=w[k].shape) * 0.2).astype(np.float32)
        om = OracleModel(cfg, w)
        src, _ = make_lines(B, L, int(rng.integers(1, 10 ** 6)), voc_size=V)
        tgt, _ = make_lines(B, Lt, int(rng.integers(1, 10 ** 6)), voc_size=V)
        if B > 1:
            tgt[0] = tgt[0][:max(1, Lt // 2)] + '\n'
            src[-1] = src[-1][:max(1, L // 2)] + '\n'
        enc_in, dec_in, dec_out, wts = vectorize_lines(om, src, tgt)
        C = cfg.ctx_width
        masks = None
        if masks_on:
            keep = lambda shape: ((rng.random(shape) > 0.2) / 0.8).astype(np.float32)
            masks = {'enc': [keep(2 * W if n == 0 else W) for n in range(d)], 'dec': [keep(W) for _ in range(d - 1)],
                     'cell': keep((B, W + C))}
        loss, grads, _ = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, masks)
        eng = HipEngine(d, W, V)
        eng.set_weights(w)
        eng.train_begin()
        gl, gn = eng.train_step(idx_of(enc_in), None, idx_of(dec_in), idx_of(dec_out), wts, masks, mode=2)
        gg = eng.train_gradients()
        onorm = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
        assert abs(gl - loss) < 3e-5 * abs(loss) and abs(gn - onorm) < 1e-4 * onorm, (case, gl, loss)
        for k in grads:
            if k == 'att_bv':
                continue                           # analytically zero (sum of softmax-Jacobian rows)
            assert np.abs(gg[k] - grads[k]).max() < 2e-3 * max(np.abs(grads[k]).max(), 1e-6 * onorm), (case, k)
        eng.train_end()
        eng.close()


def test_random_greedy_inputs():
    """Fast (batched greedy) and per-line greedy decoding over random shapes and the three input forms
    (plain, probability lines, confusion networks with alternatives of different length)."""
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    rng = np.random.default_rng(11)
    checked = 0
    for case in range(30):
        d = int(rng.integers(1, 4)); W = int(rng.choice([32, 64])); V = int(rng.choice([24, 64, 100, 257]))
        B = int(rng.integers(1, 7)); L = int(rng.integers(3, 16)); es = float(rng.choice([2., 6., 10.]))
        kind = str(rng.choice(['plain', 'prob', 'confmat']))
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        wseed = int(rng.integers(1, 10 ** 6))
        lines, _ = make_lines(B, L, wseed, voc_size=V)
        if B > 1:
            lines[0] = lines[0][:max(1, L // 2)] + '\n'
        if kind == 'plain':
            inp, conf = lines, None
        elif kind == 'prob':
            inp, conf = lines, [list(rng.uniform(0.3, 1.0, len(line)).astype(np.float32)) for line in lines]
        else:
            conf = []
            for line in lines:
                chunks = []
                for k, ch in enumerate(line):
                    if ch != '\n' and rng.random() < 0.4:
                        p = float(rng.uniform(0.5, 0.9))
                        alt = line[(k + 1) % (len(line) - 1)]
                        chunks.append([(ch, p), (alt + ch if rng.random() < 0.5 else alt, 1 - p)])
                    else:
                        chunks.append([(ch, 1.0)])
                conf.append(chunks)
            inp = conf
        res = {}
        for dt in (np.float32, np.float64):
            om = OracleModel(cfg, make_weights(cfg, seed=wseed, dtype=dt, emb_scale=es))
            res[dt] = correct_lines(om, inp, conf, fast=True, greedy=True)
        w32 = make_weights(cfg, seed=wseed, emb_scale=es)
        s2s = Sequence2Sequence()
        s2s.depth, s2s.width = d, W
        s2s.mapping, s2s.voc_size = OracleModel(cfg, w32).mapping, V
        s2s.configure(); s2s.set_weights(w32); s2s.status = 2
        got = s2s.correct_lines(inp, conf, fast=True, greedy=True)
        for j in range(B):
            if res[np.float64][0][j] != res[np.float32][0][j]:
                continue                                  # ill-conditioned line
            checked += 1
            assert got[0][j] == res[np.float32][0][j], (case, j, kind)
            assert abs(got[2][j] - res[np.float32][2][j]) < 1e-4
            assert np.allclose(got[1][j], res[np.float32][1][j], rtol=2e-4, atol=2e-6)
        s2s.engine.close()
    assert checked > 40


def test_random_per_line_greedy():
    """`correct_lines(fast=False, greedy=True)` = decode_sequence_greedy per line (seq2seq.py:1288-1354): argmax over all
    V with the index-0 NaN write-back, stop at end-of-line.  Short lines, tiny vocabularies (index 0 wins often) and
    padding lines; where the reference's np.nanargmax raises, so must the facade; what a row computes after its line
    has ended must not matter."""
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    rng = np.random.default_rng(23)
    checked = raised = 0
    for case in range(40):
        d = int(rng.integers(1, 4)); W = int(rng.choice([32, 64])); V = int(rng.choice([5, 8, 16, 64]))
        B = int(rng.integers(1, 6)); L = int(rng.integers(1, 9)); es = float(rng.choice([4., 10., 16.]))
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        wseed = int(rng.integers(1, 10 ** 6))
        lines, _ = make_lines(B, L, wseed, voc_size=V)
        if B > 1:
            lines[int(rng.integers(0, B))] = ''                              # padding line of a partial batch
        if B > 2:
            lines[0] = lines[0][:1] + '\n' if lines[0] else lines[0]         # one-character line in a longer batch
        res = {}
        for dt in (np.float32, np.float64):
            om = OracleModel(cfg, make_weights(cfg, seed=wseed, dtype=dt, emb_scale=es))
            try:
                res[dt] = correct_lines(om, lines, fast=False, greedy=True)
            except ValueError:
                res[dt] = None
        w32 = make_weights(cfg, seed=wseed, emb_scale=es)
        s2s = Sequence2Sequence()
        s2s.depth, s2s.width = d, W
        s2s.mapping, s2s.voc_size = OracleModel(cfg, w32).mapping, V
        s2s.configure(); s2s.set_weights(w32); s2s.status = 2
        conditioned = (res[np.float32] is None) == (res[np.float64] is None) and (
            res[np.float32] is None or res[np.float32][0] == res[np.float64][0])
        try:
            got = s2s.correct_lines(lines, fast=False, greedy=True)
        except ValueError:
            got = None
        if conditioned:
            assert (got is None) == (res[np.float32] is None), (case, d, W, V, lines)
            if got is None:
                raised += 1
            else:
                checked += 1
                assert got[0] == res[np.float32][0], (case, lines)
                for j in range(B):
                    assert np.allclose(got[1][j], res[np.float32][1][j], rtol=2e-4, atol=2e-6)
        if s2s.engine is not None:
            s2s.engine.close()
    assert checked > 15 and checked + raised > 25, (checked, raised)
