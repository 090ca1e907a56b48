"""One-off robustness check: tests/test_gpu_sweep.py's random beam-configuration sweep under seeds the test suite does not use
(python profiles/extended_sweep.py SEED...).  Prints the mismatching well-conditioned cases, if any."""
import sys
sys.path.insert(0, '.')
from tests.test_gpu_sweep import test_random_beam_configurations as sweep
for seed in [int(a) for a in sys.argv[1:]] or [11]:
    for tile in (-1, 0):
        try:
            sweep(seed, 40, tile)
            print('seed %d tile %d: ok' % (seed, tile), flush=True)
        except AssertionError as err:
            print('seed %d tile %d: MISMATCH %s' % (seed, tile, str(err)[:2000]), flush=True)
