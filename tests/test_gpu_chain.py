"""Layers 2..D of a beamed decoder step as ONE launch (csrc/gemm.hip gemm_chain_kernel: tiles pulled from per-XCD queues,
a tile waits for the row block of the layer below) against one launch per layer -- bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ.setdefault('CASV_POISON', '1')

from oracle import ModelConfig, make_weights, make_lines


def _engine(cfg, weights):
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
    eng.set_weights(weights)
    return eng


@pytest.mark.parametrize('d,W,V,B,N,L,es,fits', [(3, 128, 64, 32, 8, 12, 24.0, True), (4, 128, 100, 64, 8, 10, 32.0, True),
                                                 (4, 256, 256, 64, 16, 8, 64.0, True), (3, 64, 48, 128, 8, 9, 16.0, True),
                                                 (4, 128, 100, 40, 8, 10, 32.0, False), (2, 128, 64, 32, 8, 8, 24.0, False)])
def test_chained_layers_equal_separate_launches(d, W, V, B, N, L, es, fits):
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    eng = _engine(cfg, make_weights(cfg, emb_scale=es))
    _, idx = make_lines(B, L, 11, voc_size=V)
    out = {}
    for chain in (0, 1, 1):            # twice chained: the counter sets alternate between launches and are reused across calls
        eng.set_option('chain', chain)
        eng.encode(idx)
        cur = eng.decode_beam(batch_size=N, max_results=2, want_align=True)
        if chain in out:
            prev = out[chain]
            for k in ('idx', 'prob', 'len', 'score', 'rej', 'align', 'n_found', 'n_steps'):
                assert np.array_equal(prev[k], cur[k], equal_nan=True), ('repeat', k)
        out[chain] = cur
    eng.set_option('chain', 0)
    for k in ('idx', 'prob', 'len', 'score', 'rej', 'align', 'n_found', 'n_steps'):
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    # a tile grid that does not split over the 8 XCDs (or a single upper layer) is launched layer by layer
    assert (eng.stat('chained_launches') > 0) == fits
