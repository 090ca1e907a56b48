"""The split-bf16 arithmetic on EVERY GEMM launch of the decode path (process-wide override `split_bf16`; csrc/gemm.hip SPLIT
variant = mode 1, csrc/gemm_split.hip = mode 2) against the oracle, on a real MI355X.  (By default only the beam search's decoder
steps take it -- tests/test_gpu_arithmetic.py; here the encoder, the greedy decodes and the explicit decoder step take it too.)

With the override on, the fused LSTM GEMM takes every fp32 operand value apart into three bf16 values and contracts six
products per K tile on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The sums are fp32-accurate but not the k-ordered
fmaf chain of the fp32-input kernels, so the comparisons that can hold are the ones with the ORACLE (same tolerances as
tests/test_gpu_parity.py: indices / strings / counts exact, probabilities and states rtol 2e-4 + atol 2e-6, scores 1e-4);
the tests that compare two kernels of this library bit for bit (tile shapes, batch sizes, persistent kernels) cannot hold
between a launch that takes the split path and one that does not, and are not repeated here -- the last test states that
difference instead: same decisions, scores to 1e-5.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ModelConfig, make_weights
from tests.golden.make_golden import CASES
from tests import test_gpu_parity as parity


@pytest.fixture
def split_option():
    """-> set(mode, tile=-1): switches the process-wide options (mode 0 / 1 / 2 = every launch on that arithmetic); both are back at
    their defaults (-1 = no override) when the test ends."""
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(1, 32, 8)

    def set_(mode, tile=-1):
        eng.set_option('split_bf16', mode)
        eng.set_option('tile', tile)
    try:
        yield set_
    finally:
        eng.set_option('split_bf16', -1)
        eng.set_option('tile', -1)
        eng.close()


@pytest.mark.parametrize('mode', [1, 2])
def test_c3_short_lines_equal_the_oracle_with_split_operands(mode, split_option, golden_dir):
    """BASELINE configs[2]'s shape end to end (depth 4, width 512, 1024 lines, N = 8) against the committed oracle fixture:
    every line's string, length, n_found, n_steps exact, scores 1e-4, probabilities rtol 2e-4 -- unchanged test body."""
    split_option(mode)
    parity.test_c3_short_lines_equal_the_oracle(golden_dir)


@pytest.mark.parametrize('mode', [1, 2])
def test_c3_bench_batch_with_split_operands(mode, split_option, golden_dir):
    """bench.py's own configs[2] batch (1024 lines x 100 characters) against the oracle's fp32 and fp64 searches of all its
    lines: within the same factor of the oracle's own fp32 noise as the fp32-input kernels -- unchanged test body."""
    split_option(mode)
    parity.test_c3_bench_batch_agrees_with_the_oracle_like_its_own_fp64_run(golden_dir)


@pytest.mark.parametrize('mode', [1, 2])
def test_full_width_decoder_steps_with_split_operands(mode, split_option):
    """Three teacher-forced decoder steps at R = 8192 rows / depth 4 / width 512 against the oracle -- unchanged test body."""
    split_option(mode)
    parity.test_decoder_step_at_full_width_rows()


@pytest.mark.parametrize('name', list(CASES))
def test_golden_fixtures_with_split_operands(name, split_option, golden_dir):
    """The golden fixtures (encoder outputs, teacher-forced steps, exact greedy index matrix, beam results) with every GEMM
    launch forced onto 128x128 tiles and those tiles onto the split path (ragged tiles, all three K segments, the encoder's
    zero initial state) -- unchanged test body."""
    split_option(1, tile=0)
    parity.test_golden(name, golden_dir)


@pytest.mark.parametrize('kind,mode_', [('plain', 'beam'), ('prob', 'fast'), ('confmat', 'greedy')])
def test_correct_lines_with_split_operands(kind, mode_, split_option):
    split_option(1, tile=0)
    parity.test_correct_lines_equals_oracle(kind, mode_)


def test_random_beam_configurations_with_split_operands(split_option):
    """The randomised sweep of model shapes and beam parameters against the oracle (tests/test_gpu_sweep.py: 40 cases, poisoned
    device memory, ragged batches, unmapped characters) with every GEMM launch on 128x128 split tiles -- unchanged test body."""
    from tests import test_gpu_sweep as sweep
    split_option(1)
    sweep.test_random_beam_configurations(7, 40, 0)


def test_split_and_fp32_kernels_take_the_same_decisions(split_option, golden_dir):
    """What changes between the two arithmetics on configs[2]'s shape: nothing that is decided (strings, lengths, step counts
    of all 1024 lines), the scores in the sixth digit."""
    from cor_asv_ann_amd.engine import HipEngine
    with np.load(os.path.join(golden_dir, 'c3_beam_short.npz')) as f:
        idx, meta = f['idx'], f['meta']
    cfg = ModelConfig(depth=4, width=512, voc_size=256)
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
    eng.set_weights(make_weights(cfg, emb_scale=float(meta[6])))
    outs = []
    for mode in (0, 1, 2):
        split_option(mode)
        eng.encode(idx)
        outs.append(eng.decode_beam(batch_size=8))
    eng.close()
    for other in outs[1:]:
        for k in ('idx', 'len', 'n_found', 'n_steps'):
            assert np.array_equal(outs[0][k], other[k]), k
        assert np.allclose(outs[0]['score'], other['score'], rtol=0, atol=1e-5)
        assert not np.array_equal(outs[0]['score'], other['score'])       # (it IS another arithmetic: the switch did something)
    # the two tile shapes of the split arithmetic contract every element with the same instruction sequence: the same bits
    for k in ('idx', 'len', 'n_found', 'n_steps', 'score', 'prob'):
        assert np.array_equal(outs[1][k], outs[2][k], equal_nan=True), k


@pytest.mark.parametrize('B,N', [(40, 256), (103, 100)])
def test_partial_round_of_a_wide_search_goes_as_small_split_tiles_with_the_same_bits(B, N, split_option):
    """The OCR-D page call's shape (depth 2, width 512, 40 lines x 256 hypotheses = 10 240 rows per step: 320 tiles of 256x256 on
    256 CUs) and a ragged one (103 x 100 = 10 300 rows): under mode 2 the rows of the partial round go as 128x128 split tiles
    (gemm.hip, split256_cut_rows; the gathered state rows, the cell state, the outputs and the live-row counts of the tail job
    through offset pointers) -- the whole search returns the bits of mode 1, where every launch is 128x128 tiles."""
    from cor_asv_ann_amd.engine import HipEngine
    from oracle import make_lines
    cfg = ModelConfig(depth=2, width=512, voc_size=96)
    _, idx = make_lines(B, 14, 31, voc_size=96)
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size)
    eng.set_weights(make_weights(cfg, emb_scale=64.0))
    outs = []
    for mode in (1, 2):
        split_option(mode)
        eng.encode(idx)
        outs.append(eng.decode_beam(batch_size=N, beam_width_in=15, rejection_threshold=0.5))
    eng.close()
    assert outs[0]['n_steps'].max() > 3
    for k in ('idx', 'len', 'n_found', 'n_steps', 'score', 'prob'):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k


def test_split_kernels_do_not_depend_on_what_else_runs_on_the_gpu(split_option, golden_dir):
    """Two model handles decode the configs[2]-shaped fixture at the same time on two streams (the 256x256 split kernel wants the
    whole LDS of a CU and hands tiles over through LDS-DMA: foreign workgroups in between must change nothing), and a third run
    follows alone: all three results are the same bits."""
    import threading
    from cor_asv_ann_amd.engine import HipEngine
    with np.load(os.path.join(golden_dir, 'c3_beam_short.npz')) as f:
        idx, meta = f['idx'], f['meta']
    cfg = ModelConfig(depth=4, width=512, voc_size=256)
    weights = make_weights(cfg, emb_scale=float(meta[6]))
    split_option(2)
    engines = [HipEngine(cfg.depth, cfg.width, cfg.voc_size) for _ in range(2)]
    for e in engines:
        e.set_weights(weights)
    results, errors = [None, None], []

    def run(k):
        try:
            out = []
            for _ in range(3):
                engines[k].encode(idx)
                out.append(engines[k].decode_beam(batch_size=8))
            results[k] = out
        except Exception as err:           # noqa: BLE001 -- reported by the main thread
            errors.append(err)
    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    engines[0].encode(idx)
    alone = engines[0].decode_beam(batch_size=8)
    for e in engines:
        e.close()
    for out in results:
        for res in out:
            for k in ('idx', 'len', 'n_found', 'n_steps', 'score', 'prob'):
                assert np.array_equal(res[k], alone[k], equal_nan=True), k
