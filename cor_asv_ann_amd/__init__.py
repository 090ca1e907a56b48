"""MI355X-native hot path of cor-asv-ann: encoder / attention-decoder forward, greedy and beamed
decoding and the train step as HIP kernels behind the `Sequence2Sequence` API
(ocrd_cor_asv_ann/lib/seq2seq.py:13)."""
GAP = '\a'  # seq2seq.py:11
