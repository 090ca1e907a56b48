"""Where a c3 batch spends its wall time outside the kernels: host stages of correct_lines (beamed), timed around the engine calls."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from bench import make_model
from cor_asv_ann_amd.synthetic import make_lines
s2s, cfg, weights = make_model(0, 4, 512, 8, 128.0)
eng = s2s._require_engine()
lines, _ = make_lines(1024, 100, 1003, voc_size=cfg.voc_size)
s2s.correct_lines(lines, fast=False, greedy=False, alignments=False)
t = time.perf_counter
acc = {}
N = 5
for it in range(N):
    t0 = t(); idx, val, _ = s2s._sparse_lines(lines, None); t1 = t()
    eng.encode(idx, val); t2 = t()
    res = eng.decode_beam(max_results=1, want_align=False, **s2s._beam_kwargs()); t3 = t()
    texts, _ = s2s._texts(res['idx'], res['len']); t4 = t()
    out = []
    for k in range(len(lines)):
        n = int(res['len'][k])
        out.append((texts[k], res['prob'][k, :n].tolist(), float(res['score'][k]), []))
    t5 = t()
    for k, v in (('vectorize', t1 - t0), ('encode', t2 - t1), ('decode_beam', t3 - t2), ('texts', t4 - t3), ('lists', t5 - t4), ('total', t5 - t0)):
        acc[k] = acc.get(k, 0) + v
print({k: round(v / N * 1e3, 2) for k, v in acc.items()})
t0 = t()
for it in range(N): s2s.correct_lines(lines, fast=False, greedy=False, alignments=False)
print('correct_lines ms', (t() - t0) / N * 1e3)
