"""Generates tests/golden/page_beam.npz: the oracle's beamed decode of `bench.py --workload page` -- the OCR-D processor's call
(wrapper/transcode.py:110-115 with the defaults of wrapper/ocrd-tool.json): depth 2, width 512, V 640, one page of 40
confusion-network lines x 60 positions (seed 106), batch_size = 256 hypotheses per step, fixed beam width 15, relative 0.2,
rejection threshold 0.5, the bench's weights (emb_scale 128) -- in float32 AND in float64, every line.

A wide search among near-ties is ill-conditioned: the oracle's own fp32 and fp64 runs return another string on about a third of
the lines.  The fixture holds both runs so that tests/test_gpu_parity.py::test_page_call_agrees_with_the_oracle_like_its_own_fp64_run
can hold the device to a small factor of the oracle's own disagreement (the recipe of make_c3_full_golden.py).  Like the other
fixtures this pins the ORACLE (the reference's Keras is not runnable here).

    python tests/golden/make_page_golden.py [workers]        (8 workers: a few minutes)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# bench.py: WORKLOADS['page']
DEPTH, WIDTH, VOC, LINES, LENGTH, LINE_SEED, EMB_SCALE = 2, 512, 640, 40, 60, 106, 128.0
BEAM_N, WIDTH_IN, THRESHOLD_IN, REJECTION = 256, 15, 0.2, 0.5

_models = {}


def _model(dtype):
    from oracle import ModelConfig, make_weights
    from oracle.decode import OracleModel
    key = np.dtype(dtype).name
    if key not in _models:
        cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
        _models[key] = OracleModel(cfg, make_weights(cfg, dtype=dtype, emb_scale=EMB_SCALE), batch_size=BEAM_N,
                                   rejection_threshold=REJECTION, beam_width_in=WIDTH_IN, beam_threshold_in=THRESHOLD_IN)
    return _models[key]


def page_lines():
    from cor_asv_ann_amd.synthetic import make_confmat_lines
    return make_confmat_lines(LINES, LENGTH, LINE_SEED, voc_size=VOC)


def search(dtype, j):
    """Line j of the page, decoded as correct_lines does it (seq2seq.py:803-823): the whole page vectorised and encoded as one
    batch (padded to its longest line), then the line's own search."""
    from oracle import vectorize_lines
    from oracle.decode import decode_sequence_beam
    m = _model(dtype)
    lines = page_lines()
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines], lines)
    enc = m.encode(enc_in)
    st = {}
    try:
        r = next(decode_sequence_beam(m, source_seq=enc_in[j], encoder_outputs=[e[j:j + 1] for e in enc], stats=st))
        return r[0], float(r[2]), np.asarray(r[1], np.float64), st['finals'], st['steps']
    except StopIteration:
        return '', 0.0, np.zeros(0), st['finals'], st['steps']
    except IndexError:              # source_seq[source_pos] beyond the line (reference quirk 6)
        return None


def work(j):
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    return j, search(np.float32, j), search(np.float64, j)


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    import multiprocessing as mp
    t0 = time.time()
    res = [None] * LINES
    with mp.Pool(workers) as pool:
        for n, (j, a, b) in enumerate(pool.imap_unordered(work, range(LINES))):
            res[j] = (a, b)
            print('%d of %d lines  (%.0f s)' % (n + 1, LINES, time.time() - t0), flush=True)
    S = max([len(r[2]) for pair in res for r in pair if r is not None] + [1])

    def pack(k):
        pr = np.zeros((LINES, S), np.float32)
        for j, pair in enumerate(res):
            if pair[k] is not None:
                pr[j, :len(pair[k][2])] = pair[k][2]
        return {'text': np.array([p[k][0] if p[k] is not None else '' for p in res]),
                'score': np.asarray([p[k][1] if p[k] is not None else 0.0 for p in res], np.float64),
                'probs': pr,
                'found': np.asarray([p[k][3] if p[k] is not None else -1 for p in res], np.int32),
                'steps': np.asarray([p[k][4] if p[k] is not None else -1 for p in res], np.int32)}
    o32, o64 = pack(0), pack(1)
    out = {'beam_text': o32['text'], 'beam_score': o32['score'], 'beam_probs': o32['probs'], 'beam_found': o32['found'], 'beam_steps': o32['steps'],
           'beam_text64': o64['text'], 'beam_score64': o64['score'], 'beam_probs64': o64['probs'], 'beam_found64': o64['found'], 'beam_steps64': o64['steps'],
           'meta': np.asarray([DEPTH, WIDTH, VOC, LINES, LENGTH, BEAM_N, int(EMB_SCALE), LINE_SEED, WIDTH_IN], np.int64),
           'params': np.asarray([THRESHOLD_IN, REJECTION], np.float64)}
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'page_beam.npz'), **out)
    same = sum(1 for a, b in res if a is not None and b is not None and a[0] == b[0])
    print('page_beam.npz: %d lines; oracle fp32 and fp64 return the same string on %d of them, the same (string, finished hypotheses, '
          'iterations) on %d; lines with a finished hypothesis %d  (%.0f s)'
          % (LINES, same, sum(1 for a, b in res if a is not None and b is not None and (a[0], a[3], a[4]) == (b[0], b[3], b[4])),
             int((o32['found'] > 0).sum()), time.time() - t0))


if __name__ == '__main__':
    main()
