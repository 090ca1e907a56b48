"""The persistent small-batch decoder (csrc/persist.hip: all greedy steps in one launch, hand-offs between workgroups
through write-through stores and monotonic counters) against the per-step kernels -- bit for bit -- and against the
oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ.setdefault('CASV_POISON', '1')

from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
from oracle.decode import OracleModel, decode_batch_greedy, correct_lines


def _engine(cfg, weights):
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
    eng.set_weights(weights)
    return eng


@pytest.mark.parametrize('d,W,V,B,L,es', [(1, 32, 24, 3, 9, 8.0), (2, 64, 96, 16, 14, 12.0), (2, 256, 256, 40, 30, 64.0),
                                          (3, 96, 100, 21, 12, 10.0), (4, 128, 257, 70, 10, 24.0), (2, 32, 640, 5, 6, 10.0),
                                          # more than 256 lines: the attention rows' workgroups take the slots the other roles leave
                                          (2, 64, 50, 300, 8, 6.0)])
def test_persistent_equals_per_step_kernels_bit_for_bit(d, W, V, B, L, es):
    """Characters, probabilities, lengths and every alignment row: identical bits on both paths, in both greedy modes
    (mode 1 = per-line greedy with the NaN write-back; it may raise on both paths alike)."""
    from cor_asv_ann_amd._native import NativeError
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    eng = _engine(cfg, make_weights(cfg, emb_scale=es))
    _, idx = make_lines(B, L, 5, voc_size=V)
    if B > 2:
        idx[1, L // 2:] = -1                       # ragged batch: zero rows behind a shorter line
    encs = {}
    for persistent in (0, 1):          # the persistent ENCODER (one launch for all layers and time steps) against the per-step one
        eng.set_option('persistent', persistent)
        eng.encode(idx)
        encs[persistent] = eng.encoder_outputs()
    assert np.array_equal(encs[0][0], encs[1][0]), 'encoder outputs'
    for a, b in zip(encs[0][1], encs[1][1]):
        assert np.array_equal(a, b), 'encoder final states'
    for mode in (0, 1):
        out = {}
        for persistent in (0, 1):
            eng.set_option('persistent', persistent)
            eng.encode(idx)
            try:
                out[persistent] = eng.decode_greedy(mode=mode, want_align=True)
            except NativeError as err:
                out[persistent] = err.code
        eng.set_option('persistent', -1)
        if isinstance(out[0], int) or isinstance(out[1], int):
            assert out[0] == out[1], (mode, out[0] if isinstance(out[0], int) else 'ok', out[1] if isinstance(out[1], int) else 'ok')
            continue
        for k, name in enumerate(('idx', 'prob', 'len', 'align')):
            a, b = out[0][k], out[1][k]
            if mode == 1 and name != 'len':        # rows keep stepping after their line has ended; only the reported part counts
                for j in range(B):
                    n = int(out[0][2][j])
                    assert np.array_equal(a[j, :n], b[j, :n], equal_nan=True), (mode, name, j)
            else:
                assert np.array_equal(a, b, equal_nan=True), (mode, name)
    eng.close()


def test_persistent_path_is_the_default_for_small_batches_and_matches_the_oracle():
    cfg = ModelConfig(depth=2, width=64, voc_size=96)
    weights = make_weights(cfg, emb_scale=12.0)
    om = OracleModel(cfg, weights)
    lines, idx = make_lines(19, 17, 3, voc_size=96)
    enc_in, _, _, _ = vectorize_lines(om, lines, [[] for _ in lines])
    want = decode_batch_greedy(om, enc_in, return_indexes=True)
    eng = _engine(cfg, weights)
    eng.encode(idx)
    gi, gp, _, ga = eng.decode_greedy(mode=0, want_align=True)
    assert np.array_equal(gi, want[5])
    for j in range(len(lines)):
        n = len(want[2][j])
        assert np.allclose(gp[j, :n], want[2][j], rtol=2e-4, atol=2e-6)
        assert np.allclose(ga[j, :n], np.asarray(want[4][j]), atol=1e-5)
    lo, w = eng.alignments_sparse(len(lines), gi.shape[1])               # window store filled by the persistent path too
    assert (lo >= 0).all() and np.allclose(w.sum(axis=2), 1.0, atol=1e-5)
    eng.close()


def test_c2_full_size_on_the_persistent_path():
    """BASELINE configs[1] (depth 2, width 256, 256 lines of 100 characters): the persistent and the per-step path are
    compared with each other over all 202 steps of all lines.  (The comparison of this configuration with the ORACLE is
    tests/test_gpu_parity.py::test_c2_full_size_properties, which runs on the path the library picks by itself -- the
    persistent one.)"""
    cfg = ModelConfig(depth=2, width=256, voc_size=256)
    weights = make_weights(cfg, emb_scale=64.0)
    lines, idx = make_lines(256, 100, 102)
    eng = _engine(cfg, weights)
    res = {}
    for persistent in (0, 1):
        eng.set_option('persistent', persistent)
        eng.encode(idx)
        res[persistent] = eng.decode_greedy(mode=0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    eng.close()


def test_handoffs_under_uneven_load():
    """The hand-offs between workgroups must not depend on timing: random shapes decoded on the persistent path while a second
    model handle keeps other CUs busy with beamed decodes on another stream, every result compared bit for bit with the
    per-step kernels and with a repetition of itself (a stale or early read would show as a difference).  The long version of
    this (thousands of cases, idle and loaded: all identical) is scratch-level; here a few seconds of it."""
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
        rng = np.random.default_rng(5)
        t0, cases = time.time(), 0
        while time.time() - t0 < 8.0 or cases < 20:
            d = int(rng.integers(1, 4)); W = int(rng.choice([32, 64, 128, 256])); V = int(rng.choice([24, 64, 100, 256]))
            B = int(rng.integers(1, 200)); L = int(rng.integers(2, 40))
            cfg = ModelConfig(depth=d, width=W, voc_size=V)
            eng = _engine(cfg, make_weights(cfg, seed=int(rng.integers(1, 1 << 30)), emb_scale=float(rng.choice([8., 24., 64.]))))
            _, idx = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
            out = {}
            for rep in range(2):
                for p in (0, 1):
                    eng.set_option('persistent', p)
                    eng.encode(idx)
                    enc = eng.encoder_outputs()
                    gi, gp, _, ga = eng.decode_greedy(mode=0, want_align=True)
                    cur = (enc[0], np.stack(enc[1]), gi, gp, ga)
                    if p in out:
                        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[p], cur)), ('not reproducible', p, d, W, V, B, L)
                    out[p] = cur
                assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[0], out[1])), ('persistent != per-step', d, W, V, B, L)
            eng.close()
            cases += 1
    finally:
        stop.append(1)
        th.join()


def test_two_handles_decoding_persistently_at_once():
    """Two model handles on two host threads, both on the persistent encoder + decoder (each launch wants most of the chip's
    workgroup slots and its workgroups wait for each other): the launches are ordered on the DEVICE (an event between the
    handles' streams -- no host call waits inside the process-wide lock any more), none gives up, and every result equals the
    handle's own results when it runs alone."""
    import threading
    from cor_asv_ann_amd.engine import HipEngine
    cfg = ModelConfig(depth=2, width=128, voc_size=64)
    weights = make_weights(cfg, emb_scale=24.0)
    batches = [make_lines(64 + 16 * k, 20 + k, 900 + k, voc_size=64)[1] for k in range(4)]
    alone = []
    eng = _engine(cfg, weights)
    eng.set_option('persistent', 1)
    for idx in batches:
        eng.encode(idx)
        alone.append(eng.decode_greedy(mode=0)[:2])
    eng.close()
    errors = []

    def worker(order):
        try:
            e = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
            e.set_weights(weights)
            e.set_option('persistent', 1)
            for rep in range(6):
                for k in order:
                    e.encode(batches[k])
                    gi, gp = e.decode_greedy(mode=0)[:2]
                    if not (np.array_equal(gi, alone[k][0]) and np.array_equal(gp, alone[k][1])):
                        errors.append((order, rep, k))
            e.close()
        except Exception as err:            # reported by the main thread
            errors.append(repr(err))

    threads = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3], [3, 2, 1, 0])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
