"""The arithmetic policy of the device path (csrc/engine.h `arithmetic_of`), on a real MI355X.

Which matrix instruction a GEMM launch runs on is decided by the C-ABI ENTRY POINT that enqueues it and by nothing else:
the beam search (casv_decode_beam: its decoder steps and the encoder pass it consumes) contracts bf16x3-split fp32 operands on the
bf16 matrix instruction with fp32 accumulation; the greedy decodes, casv_get_encoder_outputs and the explicit decoder step (and the
encoder pass THEY consume) run the fp32-input instruction's k-ordered chain.
The invariant that makes the path shard is the one the fp32-only library had: a line's bits are a function
of (weights, line, entry point) -- not of the batch it sits in, the tile shape its rows land in (256x256 / 128x128 split tiles,
64- / 32-row fp32 tiles), the launch form of its encoder (persistent for <= 512 lines, per step above) or the rank that decodes it.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ModelConfig, make_weights


def _fixture(golden_dir):
    with np.load(os.path.join(golden_dir, 'c3_beam_short.npz')) as f:
        return f['idx'], float(f['meta'][6])


def _engine(cfg, weights, arithmetic=None):
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
    eng.set_weights(weights)
    if arithmetic is not None:
        eng.set_option('arithmetic', arithmetic)
    return eng


BEAM_KEYS = ('idx', 'len', 'n_found', 'n_steps', 'score', 'prob')


def test_default_arithmetic_is_chosen_by_the_entry_point(golden_dir):
    """configs[2]'s shape (depth 4, width 512, 1024 lines, N = 8) on three handles: default, fp32-input everywhere, split everywhere.
    The default handle's encoder outputs (casv_get_encoder_outputs) and greedy decode are the fp32 handle's bit for bit; its beam
    search -- encoder pass included: casv_encode only stages the input, the encoder runs for the entry point that consumes it -- is the
    split handle's bit for bit, takes the decisions of the fp32 handle's and differs from it in the sixth digit of the scores (it IS
    the other arithmetic).  And what a search returns does not depend on what was done with the same encoding before it."""
    idx, emb = _fixture(golden_dir)
    cfg = ModelConfig(depth=4, width=512, voc_size=256)
    weights = make_weights(cfg, emb_scale=emb)
    auto, fp32, split = _engine(cfg, weights), _engine(cfg, weights, 0), _engine(cfg, weights, 2)
    out = {}
    for name, eng in (('auto', auto), ('fp32', fp32), ('split', split)):
        eng.encode(idx)
        enc, states = eng.encoder_outputs()
        gi, gp, _, _ = eng.decode_greedy()
        out[name] = dict(enc=enc, states=np.stack(states), gi=gi, gp=gp, beam=eng.decode_beam(batch_size=8))
    a, f, s = out['auto'], out['fp32'], out['split']
    for k in ('enc', 'states', 'gi', 'gp'):
        assert np.array_equal(a[k], f[k], equal_nan=True), k
    assert not np.array_equal(a['enc'], s['enc'])                      # (the split handle's encoder is the other arithmetic)
    assert np.allclose(a['enc'], s['enc'], rtol=2e-4, atol=2e-6)
    for k in BEAM_KEYS:
        assert np.array_equal(a['beam'][k], s['beam'][k], equal_nan=True), k
    for k in ('idx', 'len', 'n_found', 'n_steps'):
        assert np.array_equal(a['beam'][k], f['beam'][k]), k
    assert np.allclose(a['beam']['score'], f['beam']['score'], rtol=0, atol=1e-5)
    assert not np.array_equal(a['beam']['score'], f['beam']['score'])
    # the search first, straight after casv_encode (above it came third): the same bits; and the greedy decode behind it as well
    auto.encode(idx)
    first = auto.decode_beam(batch_size=8)
    gi, gp, _, _ = auto.decode_greedy()
    for k in BEAM_KEYS:
        assert np.array_equal(a['beam'][k], first[k], equal_nan=True), k
    assert np.array_equal(gi, a['gi']) and np.array_equal(gp.view(np.int32), a['gp'].view(np.int32))
    for eng in (auto, fp32, split):
        eng.close()


def test_explicit_encoder_outputs_follow_the_consumer_too(golden_dir):
    """casv_set_encoder_outputs (the `encoder_outputs=` argument of decode_sequence_greedy / _beam, seq2seq.py:1305,1382): the one
    product the library adds to handed-in outputs, u = attention_dense(enc_out), is computed in the consuming entry point's
    arithmetic -- a search on the outputs of an fp32 encoder pass differs from the default search (whose encoder pass is split) only
    by rounding, equals itself whatever ran in between, and a greedy decode on them equals the plain greedy decode bit for bit."""
    idx, emb = _fixture(golden_dir)
    idx = idx[:96]
    cfg = ModelConfig(depth=4, width=512, voc_size=256)
    eng = _engine(cfg, make_weights(cfg, emb_scale=emb))
    eng.encode(idx)
    enc, states = eng.encoder_outputs()
    gi, gp, _, _ = eng.decode_greedy()
    beam = eng.decode_beam(batch_size=8)
    i3 = idx[:, :, None]
    eng.set_encoder_outputs(enc, states, src_rej=eng.source_rejection(i3, np.ones(i3.shape, np.float32)))
    b1 = eng.decode_beam(batch_size=8)
    gi2, gp2, _, _ = eng.decode_greedy()
    b2 = eng.decode_beam(batch_size=8)
    assert np.array_equal(gi, gi2) and np.array_equal(gp.view(np.int32), gp2.view(np.int32))
    for k in BEAM_KEYS:
        assert np.array_equal(b1[k], b2[k], equal_nan=True), k
    for k in ('idx', 'len', 'n_found', 'n_steps'):
        assert np.array_equal(b1[k], beam[k]), k
    assert np.allclose(b1['score'], beam['score'], rtol=0, atol=1e-5)
    eng.close()


def test_default_results_do_not_depend_on_the_batch_at_full_width(golden_dir):
    """The default policy at configs[2]'s shape: 1024 lines (greedy: per-step encoder on 64-row fp32 tiles; search: 256x256 split
    tiles), their first 600 (another tile grid), 11 from the middle (greedy: persistent encoder and decoder; search: 128x128 split
    tiles) and one line alone: every line's search result -- characters, probabilities, score, step count -- and greedy result is the
    same bits in all of them."""
    idx, emb = _fixture(golden_dir)
    cfg = ModelConfig(depth=4, width=512, voc_size=256)
    eng = _engine(cfg, make_weights(cfg, emb_scale=emb))
    eng.encode(idx)
    gi, gp, _, _ = eng.decode_greedy()
    full = eng.decode_beam(batch_size=8)
    for lo, hi in ((0, 600), (500, 511), (777, 778)):
        eng.encode(idx[lo:hi])
        gi2, gp2, _, _ = eng.decode_greedy()
        part = eng.decode_beam(batch_size=8)
        assert np.array_equal(gi[lo:hi], gi2) and np.array_equal(gp[lo:hi].view(np.int32), gp2.view(np.int32)), (lo, hi)
        for k in BEAM_KEYS:
            assert np.array_equal(full[k][lo:hi], part[k], equal_nan=True), (k, lo, hi)
    eng.close()


def test_facade_attribute_selects_the_arithmetic():
    """`Sequence2Sequence.arithmetic` ('auto' | 'fp32' | 'split') reaches the handle: same strings from all three on a
    well-conditioned batch, and 'auto' gives the scores of neither extreme's beam search... but of 'fp32' for greedy."""
    from oracle import make_lines
    from oracle.decode import OracleModel
    from tests.test_gpu_parity import _facade
    cfg = ModelConfig(depth=2, width=128, voc_size=64)
    weights = make_weights(cfg, emb_scale=16.0)
    om = OracleModel(cfg, weights, batch_size=4)
    lines, _ = make_lines(40, 20, 5, voc_size=64)
    res = {}
    for arith in ('auto', 'fp32', 'split'):
        s2s = _facade(cfg, weights, om.mapping, N=4)
        s2s.arithmetic = arith
        res[arith] = (s2s.correct_lines(lines, fast=True, greedy=True), s2s.correct_lines(lines, fast=False, greedy=False))
        s2s.engine.close()
    for arith in ('fp32', 'split'):
        assert res['auto'][0][0] == res[arith][0][0] and res['auto'][1][0] == res[arith][1][0]
        assert np.allclose(res['auto'][1][2], res[arith][1][2], atol=1e-5)
    assert res['auto'][0][2] == res['fp32'][0][2]                        # greedy scores: the fp32-input arithmetic's, bit for bit
    s2s = _facade(cfg, weights, om.mapping, N=4)
    s2s.arithmetic = 'bf16'
    with pytest.raises(KeyError):
        s2s.correct_lines(lines[:2], fast=True, greedy=True)
    s2s.engine.close()
