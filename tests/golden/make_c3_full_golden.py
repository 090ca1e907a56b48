"""Generates tests/golden/c3_beam_full.npz: the oracle's beamed decode of the FIRST lines of bench.py's own configs[2] batch
(depth 4, width 512, V 256, N = 8 hypotheses per step, the bench's weights and its 1024 x 100-character lines, seed 103) --
the workload the headline metric is quoted on, at its full line length.

On 100-character lines with the bench's peaky weights the oracle's fp32 and fp64 searches part ways on many lines (a near-tie
early in a 200-step search changes everything after it: DESIGN.md section 3), so a line pins something only where the two
agree: the script runs both on each of the first `count` lines of the batch, records the fp32 results and a flag per line
(same string, same numbers of finished hypotheses and search iterations, scores within 1e-5), and prints how many lines
carry the flag.  tests/test_gpu_parity.py::test_c3_bench_batch_equals_the_oracle decodes the WHOLE 1024-line batch on the
device and compares the flagged lines.  Like the other fixtures this pins the ORACLE (the reference's Keras is not runnable here).

    python tests/golden/make_c3_full_golden.py [workers] [count]        (8 workers, 128 lines: ~20 min on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden import make_c3_golden as short          # the same model and search code (DEPTH, WIDTH, VOC, BEAM_N, EMB_SCALE)

LINES, LENGTH, LINE_SEED = 1024, 100, 103                  # bench.py: LINES, LENGTH, LINE_SEED


def work(j):
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    from oracle import make_lines
    lines, _ = make_lines(LINES, LENGTH, LINE_SEED, voc_size=short.VOC)
    return j, short.search(np.float32, [lines[j]])[0], short.search(np.float64, [lines[j]])[0]


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    import multiprocessing as mp
    from oracle import make_lines
    lines, idx = make_lines(LINES, LENGTH, LINE_SEED, voc_size=short.VOC)
    t0 = time.time()
    res = [None] * count
    with mp.Pool(workers) as pool:
        for n, (j, a, b) in enumerate(pool.imap_unordered(work, range(count))):
            res[j] = (a, b)
            if (n + 1) % 8 == 0:
                print('%d of %d lines  (%.0f s)' % (n + 1, count, time.time() - t0), flush=True)
    why = {'string': 0, 'counts': 0, 'score': 0, 'index_error': 0}
    ok = np.zeros(count, np.int32)
    for j, (a, b) in enumerate(res):
        if a is None or b is None:
            why['index_error'] += 1
        elif a[0] != b[0]:
            why['string'] += 1
        elif a[3] != b[3] or a[4] != b[4]:
            why['counts'] += 1
        elif abs(a[1] - b[1]) > 1e-5:
            why['score'] += 1
        else:
            ok[j] = 1
    S = max(len(r[2]) for pair in res for r in pair if r is not None)

    def pack(k):        # the results of one precision for all lines
        pr = np.zeros((count, S), np.float32)
        for j, pair in enumerate(res):
            if pair[k] is not None:
                pr[j, :len(pair[k][2])] = pair[k][2]
        return {'text': np.array([p[k][0] if p[k] is not None else '' for p in res]),
                'score': np.asarray([p[k][1] if p[k] is not None else 0.0 for p in res], np.float64),
                'probs': pr,
                'found': np.asarray([p[k][3] if p[k] is not None else -1 for p in res], np.int32),
                'steps': np.asarray([p[k][4] if p[k] is not None else -1 for p in res], np.int32)}
    o32, o64 = pack(0), pack(1)
    out = {
        'idx': idx.astype(np.int32),                                        # the whole batch, as bench.py builds it
        'conditioned': ok,
        'beam_text': o32['text'], 'beam_score': o32['score'], 'beam_probs': o32['probs'], 'beam_found': o32['found'], 'beam_steps': o32['steps'],
        # the fp64 search of every line: how far the fp32 oracle itself is from exact arithmetic on this workload
        'beam_text64': o64['text'], 'beam_score64': o64['score'], 'beam_probs64': o64['probs'], 'beam_found64': o64['found'], 'beam_steps64': o64['steps'],
        'meta': np.asarray([short.DEPTH, short.WIDTH, short.VOC, LINES, LENGTH, short.BEAM_N, int(short.EMB_SCALE), count], np.int64),
    }
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'c3_beam_full.npz'), **out)
    print('c3_beam_full.npz: %d of the first %d lines of the bench batch are well-conditioned (oracle fp32 == fp64); the others: %s; '
          'conditioned lines with a finished hypothesis %d, search iterations min/median/max %d/%d/%d  (%.0f s)'
          % (int(ok.sum()), count, why, int(((out['beam_found'] > 0) & (ok > 0)).sum()), out['beam_steps'][ok > 0].min(),
             int(np.median(out['beam_steps'][ok > 0])), out['beam_steps'][ok > 0].max(), time.time() - t0))


if __name__ == '__main__':
    main()
