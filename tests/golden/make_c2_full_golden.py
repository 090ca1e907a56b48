"""Generates tests/golden/c2_greedy_full.npz: the oracle's batched greedy decode (decode_batch_greedy, seq2seq.py:1215-1286)
of bench.py's own BASELINE configs[1] workload -- depth 2, width 256, V 256, its 256 lines of 100 characters (seed 102) and
its weights (emb_scale 128) -- in float32 AND in float64: the index picked at every one of the 2T = 202 steps of every line,
and its probability.

With the bench's peaky weights the greedy recurrence is chaotic (the full softmax is fed back, seq2seq.py:1252): the oracle's
own fp32 and fp64 runs pick another character after 29 steps at the median (10 at the earliest) and agree to the end on 3 of
the 256 lines.  What a line pins is therefore its prefix up to that step; the fixture holds both runs so that
tests/test_gpu_parity.py::test_c2_bench_batch_agrees_with_the_oracle_like_its_own_fp64_run can hold the device to the
oracle's own noise.  Like the other fixtures this pins the ORACLE (the reference's Keras is not runnable here).

    python tests/golden/make_c2_full_golden.py          (~30 s)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, WIDTH, VOC, LINES, LENGTH, LINE_SEED, EMB_SCALE = 2, 256, 256, 256, 100, 102, 128.0      # bench.py: WORKLOADS['c2']


def run(dtype):
    from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
    from oracle.decode import OracleModel, decode_batch_greedy
    cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
    m = OracleModel(cfg, make_weights(cfg, dtype=dtype, emb_scale=EMB_SCALE), batch_size=LINES)
    lines, idx = make_lines(LINES, LENGTH, LINE_SEED, voc_size=VOC)
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
    g = decode_batch_greedy(m, enc_in, return_indexes='probs')
    return idx, g[1], np.asarray(g[3], np.float64), g[5], g[6]


def main():
    idx, text32, score32, i32, p32 = run(np.float32)
    _, text64, score64, i64, p64 = run(np.float64)
    S = i32.shape[1]
    first = np.array([np.nonzero(i32[j] != i64[j])[0][0] if (i32[j] != i64[j]).any() else S for j in range(LINES)])
    out = {'idx': idx.astype(np.int32),
           'greedy_idx': i32.astype(np.int16), 'greedy_prob': p32.astype(np.float32), 'greedy_text': np.array(text32), 'greedy_score': score32,
           'greedy_idx64': i64.astype(np.int16), 'greedy_prob64': p64.astype(np.float64), 'greedy_text64': np.array(text64), 'greedy_score64': score64,
           'meta': np.asarray([DEPTH, WIDTH, VOC, LINES, LENGTH, int(EMB_SCALE), LINE_SEED], np.int64)}
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'c2_greedy_full.npz'), **out)
    print('c2_greedy_full.npz: %d lines x %d steps; oracle fp32 and fp64 first differ at step min %d / 10th percentile %d / median %d, '
          'never on %d lines; equal strings on %d lines' % (LINES, S, first.min(), int(np.percentile(first, 10)), int(np.median(first)),
                                                           int((first == S).sum()), sum(a == b for a, b in zip(text32, text64))))


if __name__ == '__main__':
    main()
