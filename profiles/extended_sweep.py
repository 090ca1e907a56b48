"""One-off robustness check: tests/test_gpu_sweep.py's random sweeps under seeds the test suite does not use.
    python profiles/extended_sweep.py beam SEED...       the beam-configuration sweep (40 cases per seed and tile shape)
    python profiles/extended_sweep.py other OFFSET...    the greedy / per-line greedy / train-step sweeps with every generator
                                                         seed shifted by OFFSET
Prints the mismatching well-conditioned cases, if any."""
import sys
sys.path.insert(0, '.')
import numpy as np
import tests.test_gpu_sweep as sw

mode, args = sys.argv[1], [int(a) for a in sys.argv[2:]]
if mode == 'beam':
    for seed in args or [11]:
        for tile in (-1, 0):
            try:
                sw.test_random_beam_configurations(seed, 40, tile)
                print('seed %d tile %d: ok' % (seed, tile), flush=True)
            except AssertionError as err:
                print('seed %d tile %d: MISMATCH %s' % (seed, tile, str(err)[:2000]), flush=True)
else:
    real = np.random.default_rng
    for off in args or [100]:
        sw.np.random.default_rng = lambda seed=None, _off=off: real(None if seed is None else seed + _off)
        for name in ('test_random_greedy_inputs', 'test_random_per_line_greedy', 'test_random_train_steps'):
            try:
                getattr(sw, name)()
                print('offset %d %s: ok' % (off, name), flush=True)
            except AssertionError as err:
                print('offset %d %s: MISMATCH %s' % (off, name, str(err)[:2000]), flush=True)
    sw.np.random.default_rng = real
