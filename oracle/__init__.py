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

Keras / TensorFlow semantics this restatement relies on.  They live in the reference's un-vendored dependencies
(``requirements.txt:3-6``: tensorflow-gpu 1.15, keras 2.3), so each is a DECISION written down with the Keras 2.3.1
symbol it restates; ``tests/golden/make_keras_goldens.py`` is the script that would check all of them at once against
the reference itself (it needs that environment; nothing in this pipeline can run it).

====================================  =======================================================================  ==================
decision                              Keras 2.3.1 symbol restated                                              where here
====================================  =======================================================================  ==================
gate blocks i, f, c, o of width W;    ``keras.layers.recurrent.LSTMCell.call``: ``z0..z3`` slices of           ``model.lstm_step``
``z = (x.K + h.R) + b``               ``K.dot(inputs, kernel) + K.dot(h_tm1, recurrent_kernel)``, then
                                      ``K.bias_add`` (``implementation=2``, the default of LSTM / LSTMCell
                                      in 2.3; implementation 1 sums per gate ``(x.K_g + b_g) + h.R_g``:
                                      same value up to fp32 rounding order)
recurrent activation = logistic       ``recurrent_activation='sigmoid'`` passed explicitly (seq2seq.py:271)    ``model._sigmoid``
                                      -- not Keras' default ``hard_sigmoid``
``LSTMCell(dropout=d)`` multiplies    ``LSTMCell.call`` under implementation 2: ``inputs *= dp_mask[0]``       ``train.forward_backward``
the cell input by ONE mask            (four masks are drawn, ``_generate_dropout_mask(count=4)``, only the     (``masks['cell']``)
                                      first is used; implementation 1 would use one per gate)
``Dropout(noise_shape=(1, F))``       ``K.dropout(inputs, rate, noise_shape)``: mask broadcast over batch      ``masks['enc'|'dec']``
                                      and time, scaled by 1 / (1 - rate); identity at inference
Bidirectional: backward outputs       ``keras.layers.wrappers.Bidirectional.call``: ``K.reverse(y_rev, 1)``,   ``model._run_lstm``
re-reversed, states fw then bw        ``merge_mode='concat'``, ``states = y[1:] + y_rev[1:]``
window mask ``|t' - s| <= 5``         ``K.relu(x, max_value=5, threshold=5)`` (tensorflow_backend.relu):       ``model.attention``
                                      ``x * cast(x > threshold)`` then ``clip(0, max_value)``: zero iff
                                      ``x <= 5``; ``K.equal(.., 0)`` of that is the mask (attention.py:562-567)
``t'`` accumulated in float64         ``K.sum(a * arange(T))`` is a fp32 reduction of unspecified order in     ``model.attention``
                                      TF; here float64 and rounded once, so that no order is privileged
softmax without masking               ``K.softmax`` = ``tf.nn.softmax`` over the last axis                     ``model.decoder_step``
loss = sum(ce * w) / count(w != 0)    ``training_utils.weighted_masked_objective``: ``score *= weights``;      ``train.forward_backward``
                                      ``score /= K.mean(K.cast(K.not_equal(weights, 0), floatx))``;
                                      ``K.mean(score)`` (sample_weight_mode='temporal', seq2seq.py:494-497)
ce = -sum(y * log(clip(p / sum p)))   ``tensorflow_backend.categorical_crossentropy``: normalise, then         ``train.forward_backward``
                                      ``tf.clip_by_value(output, 1e-7, 1 - 1e-7)``; the clip passes no
                                      gradient outside the interval
global-norm clip then Adam            ``Optimizer.get_gradients``: ``norm = sqrt(sum(sum(g^2)))``,             ``train.adam_step``
                                      ``clip_norm(g, clipnorm, norm)`` = ``g * c / norm`` where
                                      ``norm >= c``; ``Adam.get_updates``: ``lr_t = lr * sqrt(1 - b2^t) /
                                      (1 - b1^t)``, ``p -= lr_t * m / (sqrt(v) + epsilon)``, epsilon 1e-7
``predict_on_batch`` broadcasts a     no batch-size check between inputs in ``Model.predict_on_batch``         ``decode.decode_sequence_beam``
(1, T, C) attended input              (seq2seq.py:1428-1429 relies on it)
candidate order of equal scores       ``np.argsort`` (quicksort, unstable) at seq2seq.py:1473: ties have no    ``decode`` module docstring
                                      defined order; here towards the higher index
====================================  =======================================================================  ==================

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``cor_asv_ann_amd``) never
does: it fails loudly when the HIP library is missing.
"""
from .weights import ModelConfig, make_vocabulary, make_weights, make_lines, weight_names  # noqa: F401
from .model import encode, decoder_step, lstm_step  # noqa: F401
from .decode import (Node, decode_batch_greedy, decode_sequence_greedy,  # noqa: F401
                     decode_sequence_beam, correct_lines, vectorize_lines)
