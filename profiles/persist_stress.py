"""Stress of the persistent kernels' hand-offs: many decodes of random shapes, each compared bit for bit with the per-step
kernels, while a second model handle keeps the GPU busy with beamed decodes on another stream (uneven load on the CUs)."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '/root/repo')
from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_lines
from cor_asv_ann_amd.engine import HipEngine

stop = False
def background():
    cfg = ModelConfig(depth=2, width=256, voc_size=64)
    e = HipEngine(2, 256, 64); e.set_weights(make_weights(cfg, emb_scale=32.0))
    _, idx = make_lines(96, 30, 1, voc_size=64)
    n = 0
    while not stop:
        e.encode(idx); e.decode_beam(batch_size=8); n += 1
    print('background beam decodes:', n, flush=True)
    e.close()

load = len(sys.argv) > 1 and sys.argv[1] == 'load'
th = threading.Thread(target=background) if load else None
if th: th.start()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); n = bad = 0
while time.time() - t0 < float(sys.argv[3]) if len(sys.argv) > 3 else 60.0:
    d = int(rng.integers(1, 4)); W = int(rng.choice([32, 64, 128, 256])); V = int(rng.choice([24, 64, 100, 256]))
    B = int(rng.integers(1, 200)); L = int(rng.integers(2, 40)); es = float(rng.choice([8., 24., 64.]))
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    eng = HipEngine(d, W, V); eng.set_weights(make_weights(cfg, seed=int(rng.integers(1, 1 << 30)), emb_scale=es))
    _, idx = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
    out = {}
    for rep in range(3):
        for p in (0, 1):
            eng.set_option('persistent', p)
            eng.encode(idx)
            enc = eng.encoder_outputs()
            gi, gp, gl, ga = eng.decode_greedy(mode=0, want_align=True)
            cur = (enc[0], np.stack(enc[1]), gi, gp, ga)
            if p in out:
                same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[p], cur))
                if not same: bad += 1; print('NOT REPRODUCIBLE', p, d, W, V, B, L, flush=True)
            out[p] = cur
        if not all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[0], out[1])):
            bad += 1; print('MISMATCH persistent vs per-step', d, W, V, B, L, flush=True)
        n += 1
    eng.close()
stop = True
if th: th.join()
print('cases', n, 'bad', bad, 'load' if load else 'idle', flush=True)
sys.exit(1 if bad else 0)
