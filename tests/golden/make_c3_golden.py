"""Generates tests/golden/c3_beam_short.npz: the oracle's beamed decode at BASELINE configs[2]'s model shape and batch
(depth 4, width 512, V 256, 1024 lines, N = 8 hypotheses per step) on SHORT lines (20 characters), where the oracle's
own fp32 and fp64 searches agree on every line -- on 100-character lines with peaky weights they do not (DESIGN.md
section 3), which is why the full-size test could only check properties.  (python tests/golden/make_c3_golden.py [workers])

Candidate lines are drawn from one seeded stream; a candidate is kept iff the fp32 and the fp64 search return the same
string, the same numbers of finished hypotheses and of search iterations, and scores within 1e-5 -- the first 1024 kept
candidates are the fixture (the script prints how many were drawn and why the others were dropped).  Like make_golden.py
this pins the ORACLE (the reference's Keras is not runnable here);
tests/test_gpu_parity.py::test_c3_short_lines_equal_the_oracle compares the device with it, all 1024 lines.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, WIDTH, VOC, LINES, LENGTH, BEAM_N = 4, 512, 256, 1024, 20, 8
EMB_SCALE = 128.0          # bench.py's scale: the search runs all N rows
LINE_SEED = 1103
CHUNK = 64                 # candidates per work item

_models = {}


def _model(dtype):
    from oracle import ModelConfig, make_weights
    from oracle.decode import OracleModel
    key = np.dtype(dtype).name
    if key not in _models:
        cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
        _models[key] = OracleModel(cfg, make_weights(cfg, dtype=dtype, emb_scale=EMB_SCALE), batch_size=BEAM_N)
    return _models[key]


def candidate_lines(first, count):
    """Candidates first .. first+count-1 of the seeded stream (every chunk reproducible by itself)."""
    from oracle import make_lines
    lines, idx = [], []
    for c in range(first // CHUNK, (first + count + CHUNK - 1) // CHUNK):
        l, i = make_lines(CHUNK, LENGTH, LINE_SEED + c, voc_size=VOC)
        lines += l
        idx.append(i)
    off = first % CHUNK
    return lines[off:off + count], np.concatenate(idx)[off:off + count]


def search(dtype, lines):
    from oracle import vectorize_lines
    from oracle.decode import decode_sequence_beam
    m = _model(dtype)
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
    enc = m.encode(enc_in)
    out = []
    for j in range(len(lines)):
        st = {}
        try:
            r = next(decode_sequence_beam(m, source_seq=enc_in[j], encoder_outputs=[e[j:j + 1] for e in enc], stats=st))
            out.append((r[0], float(r[2]), np.asarray(r[1], np.float64), st['finals'], st['steps']))
        except StopIteration:
            out.append(('', 0.0, np.zeros(0), st['finals'], st['steps']))
        except IndexError:          # source_seq[source_pos] beyond the line (reference quirk 6): not a fixture line
            out.append(None)
    return out


def work(first):
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    lines, idx = candidate_lines(first, CHUNK)
    r32, r64 = search(np.float32, lines), search(np.float64, lines)
    return first, lines, idx, r32, r64


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else LINES
    import multiprocessing as mp
    t0 = time.time()
    kept, dropped = [], {'string': 0, 'counts': 0, 'score': 0, 'index_error': 0}
    drawn = 0
    with mp.Pool(workers) as pool:
        first = 0
        while len(kept) < limit:
            batch = [first + CHUNK * i for i in range(workers)]
            first += CHUNK * workers
            for f, lines, idx, r32, r64 in pool.imap(work, batch):
                for j in range(len(lines)):
                    drawn += 1
                    a, b = r32[j], r64[j]
                    if a is None or b is None:
                        dropped['index_error'] += 1
                    elif a[0] != b[0]:
                        dropped['string'] += 1
                    elif a[3] != b[3] or a[4] != b[4]:
                        dropped['counts'] += 1
                    elif abs(a[1] - b[1]) > 1e-5:
                        dropped['score'] += 1
                    else:
                        kept.append((lines[j], idx[j], a))
            print('drawn %d kept %d dropped %s  (%.0f s)' % (drawn, len(kept), dropped, time.time() - t0), flush=True)
    kept = kept[:limit]
    S = max(len(k[2][2]) for k in kept)
    probs = np.zeros((len(kept), S), np.float32)
    for i, k in enumerate(kept):
        probs[i, :len(k[2][2])] = k[2][2]
    out = {
        'idx': np.stack([k[1] for k in kept]).astype(np.int32),
        'beam_text': np.array([k[2][0] for k in kept]),
        'beam_score': np.asarray([k[2][1] for k in kept], np.float64),
        'beam_probs': probs,
        'beam_found': np.asarray([k[2][3] for k in kept], np.int32),
        'beam_steps': np.asarray([k[2][4] for k in kept], np.int32),
        'meta': np.asarray([DEPTH, WIDTH, VOC, len(kept), LENGTH, BEAM_N, int(EMB_SCALE), drawn], np.int64),
    }
    here = os.path.dirname(os.path.abspath(__file__))
    name = 'c3_beam_short.npz' if limit == LINES else 'c3_beam_short_%d.npz' % limit
    np.savez_compressed(os.path.join(here, name), **out)
    found = out['beam_found']
    print('%s: %d lines kept of %d drawn (fp32 == fp64 on every kept line); dropped %s; lines with a finished hypothesis %d, '
          'search iterations min/median/max %d/%d/%d, output != input on %d lines'
          % (name, len(kept), drawn, dropped, int((found > 0).sum()), out['beam_steps'].min(), int(np.median(out['beam_steps'])),
             out['beam_steps'].max(), sum(1 for k in kept if k[2][0] != k[0])))


if __name__ == '__main__':
    main()
