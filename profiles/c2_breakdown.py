"""Where a c2 batch (BASELINE configs[1]: 256 lines, depth 2, width 256, greedy) spends its wall time: host stages of
correct_lines(fast=True) timed around the engine calls.  With a library built with EXTRA=-DCASV_PERSIST_PROF the persistent
decoder also prints its per-step phase times (and, with CASV_PERSIST_PLACEMENT=1, which roles share a CU)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from bench import make_model
from cor_asv_ann_amd.synthetic import make_lines
s2s, cfg, weights = make_model(0, 2, 256, 1, 4.0)
eng = s2s._require_engine()
lines, _ = make_lines(256, 100, 102, voc_size=cfg.voc_size)
for it in range(3):
    s2s.correct_lines(lines, fast=True, alignments=False)
def t(): return time.perf_counter()
acc = {}
N = 20
for it in range(N):
    t0 = t(); idx, val, _ = s2s._sparse_lines(lines, None); t1 = t()
    eng.encode(idx, val); t2 = t()
    gi, gp, _, ga = eng.decode_greedy(mode=0, want_align=False); t3 = t()
    nonpad = ((idx >= 0) & (val != 0)).any(axis=(1, 2))
    res = s2s._greedy_results(gi, gp, ga, nonpad); t4 = t()
    for k, v in (('vectorize', t1 - t0), ('encode', t2 - t1), ('decode', t3 - t2), ('results', t4 - t3), ('total', t4 - t0)):
        acc[k] = acc.get(k, 0) + v
print({k: round(v / N * 1e3, 3) for k, v in acc.items()})
t0 = t()
for it in range(N): s2s.correct_lines(lines, fast=True, alignments=False)
print('correct_lines ms', (t() - t0) / N * 1e3)
