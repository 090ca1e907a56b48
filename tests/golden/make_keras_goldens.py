"""The hook that turns "parity unpinned" into "pinned": run the REFERENCE (Keras 2.3 / TF 1.15) on the seeded cases of
tests/golden/make_golden.py and dump its tensors in the same .npz schema.

This script cannot run in the build pipeline (no tensorflow / keras / h5py there, and the reference never travels to
the GPU box).  It is for any machine that has the reference's own environment:

    pip install 'tensorflow-gpu==1.15.*' 'keras==2.3.*' 'h5py<3' ocrd_cor_asv_ann      # requirements.txt:3-6 of the reference
    python tests/golden/make_keras_goldens.py            # all cases, or: ... c1_d1_w128_peaky d2_w64_v96

Per case it
  1. draws the synthetic weights and lines exactly as make_golden.py does (cor_asv_ann_amd/synthetic.py),
  2. writes them into the reference's model container with THIS repo's dependency-free writer
     (cor_asv_ann_amd/keras_h5.py: Keras `save_weights` HDF5 layout + `config` group) -- so the run also checks that
     Keras itself reads what we write, which no test here can,
  3. loads that file through the reference's own `load_config / configure / load_weights` (scripts/proc.py:52-55),
  4. calls `encoder_model.predict_on_batch`, three teacher-forced `decoder_model.predict_on_batch` steps,
     `decode_batch_greedy` and `decode_sequence_beam` (seq2seq.py:403-406, 477-480, 1215-1286, 1356-1544),
  5. stores the results as tests/golden/keras/<case>.npz with the keys of make_golden.py.

`tests/test_oracle.py::test_oracle_matches_keras_goldens` compares the oracle with every file found there (and says
that it found none otherwise); `tests/test_gpu_parity.py::test_golden` can be pointed at the same directory with
CASV_GOLDEN_DIR.  Until such files exist, DESIGN.md and oracle/__init__.py say "parity unpinned".
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden.make_golden import CASES, NTENS          # noqa: E402  (case table shared with the oracle fixtures)


def reference_model(cfg, weights, mapping, beam_n):
    """A reference `Sequence2Sequence` holding the given tensors, loaded through its own file format."""
    from ocrd_cor_asv_ann.lib.seq2seq import Sequence2Sequence          # the reference (not importable in the pipeline)
    from cor_asv_ann_amd import keras_h5
    config = {'width': np.array(cfg.width), 'depth': np.array(cfg.depth), 'stateful': np.array(False),
              'residual_connections': np.array(False), 'deep_bidirectional_encoder': np.array(False),
              'bridge_dense': np.array(False),
              'mapping': np.fromiter((ord(mapping[1][i]) if mapping[1][i] else 0 for i in range(cfg.voc_size)), dtype=np.uint32)}
    fd, path = tempfile.mkstemp(suffix='.h5')
    os.close(fd)
    keras_h5.write_model(path, config, weights)
    s2s = Sequence2Sequence(progbars=False)
    s2s.load_config(path)
    s2s.configure()
    s2s.load_weights(path)
    s2s.batch_size = beam_n                       # also the number of hypotheses per step (seq2seq.py:1414)
    os.remove(path)
    return s2s


def run_case(name):
    from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_lines, make_vocabulary
    d, W, V, B, L, seed, es, N = CASES[name]
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    weights = make_weights(cfg, emb_scale=es)
    mapping = make_vocabulary(V)
    s2s = reference_model(cfg, weights, mapping, N)
    lines, idx = make_lines(B, L, seed, voc_size=V)
    enc_in, _, _, _ = s2s.vectorize_lines(lines, lines)
    out = {'idx': idx.astype(np.int32)}
    enc = s2s.encoder_model.predict_on_batch(enc_in)               # [enc_out, h1, c1, ..., hd, cd, a0]
    out['enc_out'] = enc[0][:NTENS]
    out['enc_states'] = np.stack(enc[1:-1])[:, :NTENS]
    p = np.zeros((B, 1, V), np.float32)
    states = list(enc[1:])
    for s in range(3):                                             # teacher forcing with the reference's own outputs
        res = s2s.decoder_model.predict_on_batch([p, enc[0]] + states)
        p, states = res[0], list(res[1:])
        out['step%d_probs' % s] = p[:NTENS, -1]
        out['step%d_states' % s] = np.stack(states[:-1])[:, :NTENS]
        out['step%d_align' % s] = states[-1][:NTENS]
    _, g_lines, g_probs, g_scores, _ = s2s.decode_batch_greedy(enc_in)
    c_i = mapping[0]
    gi = np.zeros((B, 2 * (L + 1)), np.int16)
    for j, text in enumerate(g_lines):
        gi[j, :len(text)] = [c_i[c] for c in text]                 # characters up to the end-of-line; the oracle fixture
    out['greedy_idx'] = gi                                         # keeps all 2T steps: compare the prefix
    out['greedy_len'] = np.array([len(t) for t in g_lines], np.int32)
    out['greedy_scores'] = np.asarray(g_scores, np.float64)
    texts, scores = [], []
    for j in range(B):
        try:
            r = next(s2s.decode_sequence_beam(source_seq=enc_in[j]))
            texts.append(r[0]); scores.append(r[2])
        except StopIteration:
            texts.append(''); scores.append(0.0)
    out['beam_text'] = np.array(texts)
    out['beam_score'] = np.asarray(scores, np.float64)
    return out


def main():
    names = sys.argv[1:] or list(CASES)
    os.makedirs(os.path.join(HERE, 'keras'), exist_ok=True)
    for name in names:
        out = run_case(name)
        np.savez_compressed(os.path.join(HERE, 'keras', name + '.npz'), **out)
        print('%-20s written (reference run)' % name)


if __name__ == '__main__':
    main()
