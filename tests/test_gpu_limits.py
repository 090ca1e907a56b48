"""The boundary's envelope: parameter values the reference accepts (its own test runs fixed_beam_width=50 with the
default batch_size=256, /root/reference/tests/test_all.py:56-57 -> wrapper/transcode.py:64-66) must decode, not raise:
wide beams whose new hypotheses no longer fit the LDS sort, beam widths beyond one wave, long lines, large
vocabularies.  Every case is compared with the oracle; a mismatch counts where the oracle's own fp32 and fp64 runs agree."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ.setdefault('CASV_POISON', '1')

from oracle import ModelConfig, make_weights, make_lines
from oracle.decode import OracleModel, correct_lines


def _facade(cfg, weights, mapping, **kw):
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    s2s = Sequence2Sequence()
    s2s.depth, s2s.width = cfg.depth, cfg.width
    s2s.mapping, s2s.voc_size = mapping, cfg.voc_size
    for k, v in kw.items():
        setattr(s2s, k, v)
    s2s.configure()
    s2s.set_weights(weights)
    s2s.status = 2
    return s2s


def _compare_beam(cfg, es, lines, wseed=20250614, **kw):
    """-> (lines checked, lines where the oracle is conditioned, mismatches on conditioned lines, facade)"""
    res = {}
    for dt in (np.float32, np.float64):
        om = OracleModel(cfg, make_weights(cfg, seed=wseed, dtype=dt, emb_scale=es), **kw)
        res[dt] = correct_lines(om, lines, fast=False, greedy=False)
    w32 = make_weights(cfg, seed=wseed, emb_scale=es)
    s2s = _facade(cfg, w32, OracleModel(cfg, w32).mapping, **kw)
    got = s2s.correct_lines(lines, fast=False, greedy=False)
    want = res[np.float32]
    bad, cond = [], 0
    for j in range(len(lines)):
        conditioned = res[np.float64][0][j] == want[0][j]
        cond += conditioned
        ok = got[0][j] == want[0][j] and abs(got[2][j] - want[2][j]) < 1e-4
        if conditioned and not ok:
            bad.append((j, got[0][j], want[0][j], got[2][j], want[2][j]))
    return len(lines), cond, bad, s2s


@pytest.mark.parametrize('es,thr', [(12.0, 0.2), (3.0, 0.01)])
def test_reference_default_beam_n256_width50(es, thr):
    """batch_size = 256 hypotheses per step, beam_width_in = 50, rejection 0.1: up to 256 * 51 = 13 056 new hypotheses
    per line and step.  The flat model (es = 3, threshold 0.01) really creates that many: more than the 4096 keys the
    LDS sorts at once, so the sort runs in several passes merged by rank."""
    cfg = ModelConfig(depth=2, width=64, voc_size=96)
    lines, _ = make_lines(3, 9, 77, voc_size=96)
    lines[1] = lines[1][:5] + '\n'
    n, cond, bad, s2s = _compare_beam(cfg, es, lines, batch_size=256, beam_width_in=50, beam_threshold_in=thr,
                                      rejection_threshold=0.1)
    most = s2s.engine.stat('beam_max_new_keys')
    if thr < 0.1:
        assert most > s2s.engine.stat('beam_sort_capacity'), most      # the multi-pass sort ran
    assert cond >= 2 and not bad, (cond, bad, most)
    s2s.engine.close()


def test_trained_model_with_the_reference_test_settings(tmp_path, monkeypatch):
    """A trained (well-conditioned) copy-task model decoded with the settings of the reference's own test
    (batch_size 256, fixed_beam_width 50): GPU == oracle, line for line."""
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    alphabet = 'abcdefghij '
    lines = [''.join(rng.choice(list(alphabet), size=rng.integers(5, 12))) for _ in range(1600)]
    (tmp_path / 'train.tsv').write_text(''.join('%s\t%s\n' % (l, l) for l in lines))
    s2s = Sequence2Sequence()
    s2s.depth, s2s.width, s2s.batch_size, s2s.epochs, s2s.dropout = 2, 64, 32, 30, 0.0
    s2s._rng = np.random.default_rng(1)
    s2s.configure()
    s2s.train([str(tmp_path / 'train.tsv')])
    test = [l + '\n' for l in lines[:6]]
    test[2] = test[2][:3] + 'jj' + test[2][5:]                 # a line the model has a reason to correct
    s2s.batch_size, s2s.beam_width_in, s2s.rejection_threshold = 256, 50, 0.1
    got = s2s.correct_lines(test, fast=False, greedy=False)
    cfg = ModelConfig(depth=2, width=64, voc_size=s2s.voc_size)
    om = OracleModel(cfg, s2s.get_weights(), mapping=s2s.mapping, batch_size=256, beam_width_in=50, rejection_threshold=0.1)
    want = correct_lines(om, test, fast=False, greedy=False)
    assert got[0] == want[0]
    assert np.allclose(got[2], want[2], atol=1e-4)
    for j in range(len(test)):
        assert np.allclose(got[1][j], want[1][j], rtol=2e-4, atol=2e-6)
    s2s.engine.close()


@pytest.mark.parametrize('V,width_in', [(96, 100), (257, 100), (257, 200)])
def test_beam_width_in_beyond_one_wave(V, width_in):
    """beam_width_in > 63 (more children per expansion than a wave has lanes), also beyond the vocabulary size."""
    cfg = ModelConfig(depth=1, width=32, voc_size=V)
    lines, _ = make_lines(4, 7, 5, voc_size=V)
    n, cond, bad, s2s = _compare_beam(cfg, 4.0, lines, batch_size=4, beam_width_in=width_in, beam_threshold_in=0.001,
                                      rejection_threshold=0.3)
    assert s2s.engine.stat('beam_max_new_keys') > 4 * 64
    assert cond >= 2 and not bad, (cond, bad)
    s2s.engine.close()


@pytest.mark.parametrize('V', [1500, 3000])
def test_large_vocabulary(V):
    """V > 1024 (more than 16 vocabulary entries per lane in the beam kernel): greedy and beamed decoding."""
    cfg = ModelConfig(depth=1, width=32, voc_size=V)
    lines, _ = make_lines(3, 8, 9, voc_size=V)
    n, cond, bad, s2s = _compare_beam(cfg, 16.0, lines, batch_size=4)
    assert cond >= 2 and not bad, (cond, bad)
    om = OracleModel(cfg, make_weights(cfg, emb_scale=16.0))
    want = correct_lines(om, lines, fast=True, greedy=True)
    got = s2s.correct_lines(lines, fast=True, greedy=True)
    assert got[0] == want[0] and np.allclose(got[2], want[2], atol=1e-4)
    s2s.engine.close()


def test_long_lines():
    """T = 701 positions (S = 1402 decode steps): batched greedy, per-line greedy and a small beam."""
    V = 24
    cfg = ModelConfig(depth=1, width=32, voc_size=V)
    lines, _ = make_lines(2, 700, 3, voc_size=V)
    lines[1] = lines[1][:300] + '\n'
    w = make_weights(cfg, emb_scale=10.0)
    om = OracleModel(cfg, w, batch_size=2)
    s2s = _facade(cfg, w, om.mapping, batch_size=2)
    want = correct_lines(om, lines, fast=True, greedy=True)
    got = s2s.correct_lines(lines, fast=True, greedy=True)
    om64 = OracleModel(cfg, make_weights(cfg, dtype=np.float64, emb_scale=10.0), batch_size=2)
    want64 = correct_lines(om64, lines, fast=True, greedy=True)
    for j in range(2):
        # 1402 chaotic steps: compare up to the first character on which the oracle's own precisions part ways
        n = next((k for k, (a, b) in enumerate(zip(want[0][j], want64[0][j])) if a != b), min(len(want[0][j]), len(want64[0][j])))
        assert n > 50 and got[0][j][:n] == want[0][j][:n], (j, n)
    wb = correct_lines(om, lines, fast=False, greedy=False)
    gb = s2s.correct_lines(lines, fast=False, greedy=False)
    wb64 = correct_lines(om64, lines, fast=False, greedy=False)
    for j in range(2):
        if wb[0][j] == wb64[0][j]:
            assert gb[0][j] == wb[0][j] and abs(gb[2][j] - wb[2][j]) < 1e-4
    s2s.engine.close()


def test_limits_are_reported_not_crashed():
    from cor_asv_ann_amd.engine import HipEngine
    from cor_asv_ann_amd._native import NativeError
    with pytest.raises(NativeError):
        HipEngine(1, 32, 5000)                     # vocabulary beyond 4096
    cfg = ModelConfig(depth=1, width=32, voc_size=24)
    eng = HipEngine(1, 32, 24)
    eng.set_weights(make_weights(cfg))
    eng.encode(np.full((1, 4), 3, np.int32))
    with pytest.raises(NativeError):
        eng.decode_beam(batch_size=2000)           # hypotheses per step beyond 1024
    with pytest.raises(NativeError):
        eng.encode(np.full((1, 5000), 3, np.int32))    # line longer than 4096
    eng.close()
