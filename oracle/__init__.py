"""CPU oracle for the cor-asv-ann hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch numpy restatement of the reference's
encoder / attention-decoder forward, its three decode loops and its train step
(``ocrd_cor_asv_ann/lib/seq2seq.py:190-489,1215-1608``,
``ocrd_cor_asv_ann/lib/attention.py:526-575``), written from the reference's
source text plus the Keras 2.3 / TF 1.15 layer semantics recorded in
SURVEY.md appendix A.

PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or
fixtures for this path (its tests assert only file existence and substrings,
``tests/test_all.py:21-104``), and Keras/TensorFlow are not installed here, so
the reference itself cannot be run to pin this restatement.  It is pinned only
by analytic known-answer tests, an independent cross-check of the LSTM stacks
and gradients against torch-CPU, and fp32-vs-fp64 self agreement
(tests/test_oracle_*.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``cor_asv_ann_amd``) never
does: it fails loudly when the HIP library is missing.
"""
from .weights import ModelConfig, make_vocabulary, make_weights, make_lines, weight_names  # noqa: F401
from .model import encode, decoder_step, lstm_step  # noqa: F401
from .decode import (Node, decode_batch_greedy, decode_sequence_greedy,  # noqa: F401
                     decode_sequence_beam, correct_lines, vectorize_lines)
