"""Generates tests/golden/*.npz from the oracle (python tests/golden/make_golden.py).

The reference (Keras 2.3 / TF 1.15) is not installed in this pipeline and holds no golden vectors, so
these fixtures pin the ORACLE's outputs -- they let the GPU tests run without recomputing the oracle
and make any later change of the oracle visible.  Each case: seeded synthetic weights + lines
(SURVEY.md section 8d), every tensor of one encode + three decoder steps for a few lines, the full greedy
index matrix, and the beam top-1 per line.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ModelConfig, make_weights, make_lines, vectorize_lines   # noqa: E402
from oracle.decode import OracleModel, decode_batch_greedy, decode_sequence_beam   # noqa: E402

# name -> (depth, width, voc, lines, length, line seed, emb_scale, beam N)
CASES = {
    'c1_d1_w128_flat': (1, 128, 256, 32, 40, 101, 4.0, 8),      # BASELINE configs[0], survey weights
    'c1_d1_w128_peaky': (1, 128, 256, 32, 40, 101, 14.0, 8),
    'd2_w128_v64': (2, 128, 64, 8, 20, 7, 16.0, 4),
    'd4_w128_v256': (4, 128, 256, 8, 30, 101, 64.0, 8),
    'd2_w64_v96': (2, 64, 96, 6, 15, 9, 12.0, 4),               # V not a power of two
}
NTENS = 4   # lines whose intermediate tensors are stored


def run_case(name):
    d, W, V, B, L, seed, es, N = CASES[name]
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    weights = make_weights(cfg, emb_scale=es)
    m = OracleModel(cfg, weights, batch_size=N)
    lines, idx = make_lines(B, L, seed, voc_size=V)
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
    out = {'idx': idx.astype(np.int32)}
    enc = m.encode(enc_in)
    out['enc_out'] = enc[0][:NTENS]
    out['enc_states'] = np.stack(enc[1:-1])[:, :NTENS]
    p = np.zeros((B, V), np.float32)
    states = enc[1:]
    for s in range(3):
        p, states = m.step(p, enc[0], states)
        out['step%d_probs' % s] = p[:NTENS]
        out['step%d_states' % s] = np.stack(states[:-1])[:, :NTENS]
        out['step%d_align' % s] = states[-1][:NTENS]
    g = decode_batch_greedy(m, enc_in, return_indexes=True)
    out['greedy_idx'] = g[5].astype(np.int16)
    out['greedy_scores'] = np.asarray(g[3], np.float64)
    margins = []
    beam_txt, beam_score, beam_found, beam_steps = [], [], [], []
    for j in range(B):
        st = {}
        try:
            r = next(decode_sequence_beam(m, source_seq=enc_in[j], encoder_outputs=[e[j:j + 1] for e in enc], stats=st))
            beam_txt.append(r[0]); beam_score.append(r[2])
        except StopIteration:
            beam_txt.append(''); beam_score.append(0.0)
        beam_found.append(st['finals']); beam_steps.append(st['steps'])
    out['beam_text'] = np.array(beam_txt)
    out['beam_score'] = np.asarray(beam_score, np.float64)
    out['beam_found'] = np.asarray(beam_found, np.int32)
    out['beam_steps'] = np.asarray(beam_steps, np.int32)
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        out = run_case(name)
        np.savez_compressed(os.path.join(here, name + '.npz'), **out)
        d, W, V, B, L, seed, es, N = CASES[name]
        # conditioning report: fp64 run of the same case must take the same decisions
        cfg = ModelConfig(depth=d, width=W, voc_size=V)
        m64 = OracleModel(cfg, make_weights(cfg, dtype=np.float64, emb_scale=es), batch_size=N)
        lines, _ = make_lines(B, L, seed, voc_size=V)
        enc_in, _, _, _ = vectorize_lines(m64, lines, [[] for _ in lines])
        g64 = decode_batch_greedy(m64, enc_in, return_indexes=True)
        agree = (g64[5] == out['greedy_idx']).mean()
        print('%-20s greedy fp32==fp64: %.4f  beam found %s steps %s' %
              (name, agree, out['beam_found'].tolist(), out['beam_steps'].tolist()))


if __name__ == '__main__':
    main()
