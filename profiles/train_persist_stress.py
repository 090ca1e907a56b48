"""Stress of the train step's persistent recurrences (train_persist*.hip): random shapes, two batches taking turns, each step
evaluated with the persistent launches and with per-step launches, idle and beside another model handle's beamed decodes on
a second stream.  Prints one line per phase; a stale or early read between workgroups shows as a mismatch.
    python profiles/train_persist_stress.py [seconds per phase]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_lines
from cor_asv_ann_amd.engine import HipEngine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
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


def phase(name, seed, loaded):
    th = None
    if loaded:
        del stop[:]
        th = threading.Thread(target=background); th.start()
    rng = np.random.default_rng(seed)
    t0, cases, steps, bad = time.time(), 0, 0, []
    worst_l, worst_n = 0.0, 0.0
    try:
        while time.time() - t0 < budget:
            d = int(rng.integers(2, 5)); W = int(rng.choice([128, 256, 512])); V = int(rng.choice([40, 96, 256]))
            B = int(rng.integers(1, 520)); L = int(rng.integers(2, 40))
            cfg = ModelConfig(depth=d, width=W, voc_size=V)
            w = make_weights(cfg, seed=int(rng.integers(1, 1 << 30)), emb_scale=float(rng.choice([3., 8.])))
            srcs = [make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)[1] for _ in range(2)]
            _, tidx = make_lines(B, L, int(rng.integers(1, 1 << 30)), voc_size=V)
            U = L + 2
            dec_in = np.full((B, U), -1, np.int32); dec_out = np.full((B, U), -1, np.int32)
            dec_in[:, 1:L + 2] = tidx; dec_out[:, :L + 1] = tidx
            wts = (dec_out >= 0).astype(np.float32)
            eng = HipEngine(d, W, V); eng.set_weights(w); eng.train_begin()
            seen = {}
            for rep in range(2):
                for p in (-1, 0):
                    eng.set_option('persistent', p)
                    for which in (0, 1):
                        loss, norm = eng.train_step(srcs[which], None, dec_in, dec_out, wts, None, mode=2)
                        steps += 1
                        if (p, which) in seen:
                            worst_l = max(worst_l, abs(loss - seen[p, which][0]) / abs(loss)); worst_n = max(worst_n, abs(norm - seen[p, which][1]) / norm)
                        seen[p, which] = (loss, norm)
            for which in (0, 1):
                dl = abs(seen[-1, which][0] - seen[0, which][0]) / abs(seen[0, which][0]); dn = abs(seen[-1, which][1] - seen[0, which][1]) / seen[0, which][1]
                worst_l = max(worst_l, dl); worst_n = max(worst_n, dn)
                if not (dl < 1e-6 and dn < 2e-5 and np.isfinite(seen[-1, which][0])):
                    bad.append((d, W, V, B, L, dl, dn))
            eng.train_end(); eng.close()
            cases += 1
    finally:
        if th is not None:
            stop.append(1); th.join()
    print('%-8s %4d shapes, %5d train steps: %d mismatches; largest relative difference of the loss %.2e, of the gradient norm %.2e (persistent vs per-step, and repeats)%s'
          % (name, cases, steps, len(bad), worst_l, worst_n, '' if not bad else '  ' + repr(bad[:3])), flush=True)


phase('idle', 101, False)
phase('loaded', 202, True)
