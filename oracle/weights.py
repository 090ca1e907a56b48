"""Synthetic vocabulary, weights and lines: defined once in `cor_asv_ann_amd/synthetic.py` (so that `bench.py` can
build its workload without importing the oracle) and re-exported here under the names the tests use."""
from cor_asv_ann_amd.synthetic import *  # noqa: F401,F403
from cor_asv_ann_amd.synthetic import WEIGHT_SEED, ModelConfig, make_vocabulary, weight_names, make_weights, make_lines  # noqa: F401
