"""Generates tests/golden/c4_train_step.npz: the oracle's train step at BASELINE configs[3] full size
(depth 4, width 512, V 256, 512 lines x 100 characters, targets = sources with 5 % substitutions, dropout 0.2
as explicit masks) -- loss, global gradient norm, per-tensor gradient norms and a seeded sample of every gradient
tensor.  The numpy oracle needs several minutes for this step on a few cores, hence a fixture instead of a live run
(python tests/golden/make_c4_golden.py).  Like make_golden.py this pins the ORACLE (the reference's Keras is not
runnable here); tests/test_gpu_train.py::test_c4_full_size_step_equals_oracle compares the device with it.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, WIDTH, VOC, B, LENGTH = 4, 512, 256, 512, 100
NSAMPLE = 512


def c4_inputs(batch=B):
    """The batch of `bench.py --workload c4` (same seeds): index arrays, weights, dropout masks."""
    from cor_asv_ann_amd.synthetic import make_lines
    _, sidx = make_lines(batch, LENGTH, 104, voc_size=VOC)
    rng = np.random.default_rng(1104)
    tidx = sidx.copy()
    sub = rng.random(tidx.shape) < 0.05
    sub[:, -1] = False
    tidx[sub] = rng.integers(2, VOC, size=int(sub.sum()))
    U = LENGTH + 2
    dec_in = np.full((batch, U), -1, np.int32)
    dec_out = np.full((batch, U), -1, np.int32)
    dec_in[:, 1:LENGTH + 2] = tidx
    dec_out[:, :LENGTH + 1] = tidx
    wts = (dec_out >= 0).astype(np.float32)
    keep = lambda shape: ((rng.random(shape) >= 0.2) / 0.8).astype(np.float32)
    masks = {'enc': [keep(2 * WIDTH if n == 0 else WIDTH) for n in range(DEPTH)], 'dec': [keep(WIDTH) for _ in range(DEPTH - 1)],
             'cell': keep((batch, 2 * WIDTH))}
    return sidx, dec_in, dec_out, wts, masks


def sample_positions(name, size):
    rng = np.random.default_rng(abs(hash_name(name)) % (2 ** 31))
    return rng.integers(0, size, size=min(NSAMPLE, size))


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h


def one_hot(idx, V):
    out = np.zeros(idx.shape + (V,), np.float32)
    b, t = np.nonzero(idx >= 0)
    out[b, t, idx[b, t]] = 1.0
    return out


def main():
    from oracle import ModelConfig, make_weights
    from oracle.train import forward_backward
    cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
    w = make_weights(cfg, emb_scale=4.0)
    sidx, dec_in, dec_out, wts, masks = c4_inputs()
    t0 = time.time()
    loss, grads, _ = forward_backward(cfg, w, one_hot(sidx, VOC), one_hot(dec_in, VOC), one_hot(dec_out, VOC), wts, masks)
    print('oracle train step: %.1f s, loss %.6f' % (time.time() - t0, loss))
    out = {'loss': np.float64(loss),
           'grad_norm': np.float64(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))}
    for k, g in grads.items():
        flat = np.asarray(g, np.float32).ravel()
        out['norm/' + k] = np.float64(np.sqrt(float((flat.astype(np.float64) ** 2).sum())))
        out['max/' + k] = np.float64(np.abs(flat).max())
        out['sample/' + k] = flat[sample_positions(k, flat.size)]
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c4_train_step.npz'), **out)
    print('grad norm %.6f' % out['grad_norm'])


if __name__ == '__main__':
    main()
