"""Input vectorisation and the three decode loops, restated -- test infrastructure.

``vectorize_lines``          <- seq2seq.py:1020-1119
``decode_batch_greedy``      <- seq2seq.py:1215-1286
``decode_sequence_greedy``   <- seq2seq.py:1288-1354
``decode_sequence_beam``     <- seq2seq.py:1356-1544, ``Node`` <- seq2seq.py:1546-1608
``correct_lines``            <- seq2seq.py:782-842

Scalar-arithmetic spec (the reference ran on NumPy 1.x, where np.float32 scalars combined with
Python floats promote to float64): per-character cost = float32 -log(p) of the float32
probability; cumulative costs and pro_cost accumulate in float64; the beam threshold
`highest * beam_threshold_in` and the rejection comparison are evaluated in float64.
Ties in the candidate order are broken towards the higher index (stable ascending argsort,
iterated in reverse).
"""
from bisect import insort_left
from dataclasses import dataclass, field
import numpy as np

from .model import encode, decoder_step
from .weights import ModelConfig, make_vocabulary

GAP = '\a'   # seq2seq.py:11


@dataclass
class OracleModel:
    cfg: ModelConfig
    weights: dict
    mapping: tuple = None
    batch_size: int = 256               # s2s:111 (also beam N, s2s:1414)
    rejection_threshold: float = 0.3    # s2s:162
    beam_width_in: int = 15             # s2s:164
    beam_threshold_in: float = 0.2      # s2s:167
    beam_width_out: int = 16            # s2s:169
    recompute_u: bool = False           # True = reference dataflow (u inside every step, s2s:459-460)
    errors: list = field(default_factory=list)

    def __post_init__(self):
        if self.mapping is None:
            self.mapping = make_vocabulary(self.cfg.voc_size)

    @property
    def voc_size(self):
        return self.cfg.voc_size

    def encode(self, x):
        return encode(self.cfg, self.weights, x)

    def step(self, p_in, enc_out, states, u=None):
        if u is None and not self.recompute_u:
            u = enc_out @ self.weights['att_U']
        return decoder_step(self.cfg, self.weights, p_in, enc_out, states, u=u)


def vectorize_lines(m, encoder_input_sequences, decoder_input_sequences, encoder_conf_sequences=None):
    """seq2seq.py:1020-1119.  Returns (enc_in (B,T,V), dec_in (B,Tt+1,V), dec_out (B,Tt+1,V),
    weights (B,Tt+1)); plain text gives uint32 one-hot rows, confidences give float32 rows."""
    V = m.voc_size
    c_i = m.mapping[0]
    max_enc = max(map(len, encoder_input_sequences))
    max_dec = max(map(len, decoder_input_sequences))
    assert len(encoder_input_sequences) == len(decoder_input_sequences)
    B = len(encoder_input_sequences)
    with_confmat = False
    if encoder_conf_sequences:
        assert len(encoder_conf_sequences) == len(encoder_input_sequences)
        if type(encoder_conf_sequences[0][0]) is list:
            with_confmat = True
            max_enc = max(sum(max(len(x[0]) for x in chunk) if chunk else 0 for chunk in seq)
                          for seq in encoder_conf_sequences)
            encoder_input_sequences = encoder_conf_sequences
    enc = np.zeros((B, max_enc, V), dtype=np.float32 if encoder_conf_sequences else np.uint32)
    dec_in = np.zeros((B, max_dec + 1, V), dtype=np.uint32)
    dec_out = np.zeros((B, max_dec + 1, V), dtype=np.uint32)

    def lookup(char, what, i):
        if char not in c_i:
            if char != GAP:
                m.errors.append('unmapped character "%s" at %s sequence %d' % (char, what, i))
            return 0
        return c_i[char]

    for i, (enc_seq, dec_seq) in enumerate(zip(encoder_input_sequences, decoder_input_sequences)):
        if with_confmat:
            j = 0
            for chunk in enc_seq:
                max_chars = max(len(x[0]) for x in chunk) if chunk else 0
                for chars, conf in chunk:
                    for k, char in enumerate(chars):
                        enc[i, j + k, lookup(char, 'encoder input', i)] = conf
                j += max_chars
        else:
            for j, char in enumerate(enc_seq):
                idx = lookup(char, 'encoder input', i)
                enc[i, j, idx] = 1
                if encoder_conf_sequences:
                    enc[i, j, idx] = encoder_conf_sequences[i][j]
        for j, char in enumerate(dec_seq):
            idx = lookup(char, 'decoder input', i)
            dec_in[i, j + 1, idx] = 1
            dec_out[i, j, idx] = 1
    weights = np.ones(dec_out.shape[:-1], dtype=np.float32)
    weights[np.all(dec_out == 0, axis=2)] = 0.
    return enc, dec_in, dec_out, weights


def decode_batch_greedy(m, encoder_input_data, return_indexes=False):
    """seq2seq.py:1215-1286: 2T fixed steps, argmax without index 0, full softmax fed back.
    return_indexes: also the (B, 2T) matrix of the indices picked at every step (the reference keeps them only up to a line's
    newline); 'probs': and the matrix of their probabilities."""
    V = m.voc_size
    i_c = m.mapping[1]
    enc = m.encode(encoder_input_data)
    enc_out, states = enc[0], enc[1:]
    u = None if m.recompute_u else enc_out @ m.weights['att_U']
    B, T = encoder_input_data.shape[:2]
    dec_in = np.zeros((B, V), dtype=np.uint32)
    dec_out_data = np.zeros((B, T * 2, V), dtype=np.uint32)
    seqs = [''] * B
    probs = [[] for _ in range(B)]
    scores_acc = [0.] * B
    aligns = [[] for _ in range(B)]
    all_idx = np.zeros((B, 2 * T), np.int64)
    all_prob = np.zeros((B, 2 * T), np.float64)
    nonpad = [bool(np.any(encoder_input_data[j])) for j in range(B)]
    for i in range(T * 2):
        dec_out_data[:, i] = dec_in          # uint32 truncation quirk
        scores, states = m.step(dec_in, enc_out, states, u=u)
        alignment = states[-1]
        indexes = np.nanargmax(scores[:, 1:], axis=1) + 1     # s2s:1250
        all_idx[:, i] = indexes
        all_prob[:, i] = scores[np.arange(B), indexes]
        dec_in = scores                                        # s2s:1252
        with np.errstate(divide='ignore'):
            logscores = -np.log(scores)
        for j, idx in enumerate(indexes):
            if seqs[j].endswith('\n') or not nonpad[j]:
                continue
            seqs[j] += i_c[int(idx)]
            probs[j].append(scores[j, idx])
            scores_acc[j] += logscores[j, idx]
            aligns[j].append(alignment[j])
    for j in range(B):
        if seqs[j]:
            scores_acc[j] /= len(seqs[j])
    if return_indexes == 'probs':
        return dec_out_data, seqs, probs, scores_acc, aligns, all_idx, all_prob
    if return_indexes:
        return dec_out_data, seqs, probs, scores_acc, aligns, all_idx
    return dec_out_data, seqs, probs, scores_acc, aligns


def decode_sequence_greedy(m, source_seq=None, encoder_outputs=None):
    """seq2seq.py:1288-1354: one line, argmax over all V with the index-0 NaN write-back quirk."""
    i_c = m.mapping[1]
    if encoder_outputs is None:
        encoder_outputs = m.encode(np.expand_dims(source_seq, axis=0))
    attended, states = encoder_outputs[0], list(encoder_outputs[1:])
    u = None if m.recompute_u else attended @ m.weights['att_U']
    target = np.zeros((1, m.voc_size), dtype=np.uint32)
    text, dprobs, dscore, aligns = '', [], 0, []
    for i in range(attended.shape[1] * 2):
        scores, new_states = m.step(target, attended, states, u=u)
        idx = np.nanargmax(scores[0])
        prob = scores[0, idx]
        score = -np.log(prob)
        char = i_c[int(idx)]
        if char == '':
            scores[0, idx] = np.nan                # s2s:1334: NaN stays in the fed-back vector
            idx = np.nanargmax(scores[0])
            prob = scores[0, idx]
            score = -np.log(prob)
            char = i_c[int(idx)]
        text += char
        dprobs.append(prob)
        dscore += score
        aligns.append(new_states[-1][0])
        if char == '\n':
            break
        target = scores
        states = list(new_states)
    return text, dprobs, dscore / len(text), aligns


class Node(object):
    """One hypothesis in the character trie (seq2seq.py:1546-1608)."""
    def __init__(self, state, value, scores, cost, parent=None, prob=1.0, alignment=None,
                 length0=None, cost0=None):
        self._sequence = None
        self.value = value
        self.parent = parent
        self.state = state
        self.cum_cost = (parent.cum_cost + float(cost)) if parent else float(cost)   # float64 sum
        self.length = 1 if parent is None else parent.length + 1
        self.length0 = length0 or (parent.length0 if parent else 1)
        self.cost0 = cost0 or (parent.cost0 if parent else 0)
        self.prob = prob
        self.scores = scores
        if alignment is None:
            self.alignment = parent.alignment if parent else []
        else:
            self.alignment = alignment

    def to_sequence(self):
        if not self._sequence:
            self._sequence = []
            cur = self
            while cur:
                self._sequence.insert(0, cur)
                cur = cur.parent
        return self._sequence

    def __str__(self):
        return ''.join(n.value for n in self.to_sequence()[1:])

    def pro_cost(self):
        return - (self.cum_cost + self.cost0 * abs(self.length - self.length0))   # s2s:1595

    def __lt__(self, other): return self.pro_cost() < other.pro_cost()
    def __le__(self, other): return self.pro_cost() <= other.pro_cost()
    def __eq__(self, other): return self.pro_cost() == other.pro_cost()
    def __ne__(self, other): return self.pro_cost() != other.pro_cost()
    def __gt__(self, other): return self.pro_cost() > other.pro_cost()
    def __ge__(self, other): return self.pro_cost() >= other.pro_cost()
    __hash__ = object.__hash__


def decode_sequence_beam(m, source_seq=None, encoder_outputs=None, stats=None):
    """seq2seq.py:1356-1544: best-first search with N = batch_size hypotheses per step;
    generator of (text, probs, cum_cost/(length-1), alignments), best first."""
    V = m.voc_size
    i_c = m.mapping[1]
    if encoder_outputs is None:
        encoder_outputs = m.encode(np.expand_dims(source_seq, axis=0))
    attended = encoder_outputs[0]
    T = attended.shape[1]
    u = None if m.recompute_u else attended @ m.weights['att_U']
    states_values = list(encoder_outputs[1:])
    next_beam = [Node(state=states_values, value='', scores=np.zeros(V), prob=[], cost=0.0,
                      alignment=[], length0=T, cost0=3.0)]
    final_beam = []
    max_batches = T * 2
    steps_run = 0
    for l in range(max_batches):
        beam = []
        while next_beam:
            node = next_beam.pop()
            if node.value == '\n':
                insort_left(final_beam, node)
            else:
                beam.append(node)
            if len(beam) >= m.batch_size:
                break
        if not beam:
            break
        if (len(final_beam) > m.beam_width_out and
                final_beam[-1].pro_cost() > beam[0].pro_cost()):
            break
        steps_run += 1
        target = np.vstack([node.scores for node in beam])
        states_val = [np.vstack([node.state[layer] for node in beam])
                      for layer in range(len(beam[0].state))]
        scores_output, states_output = m.step(target, attended, states_val, u=u)
        for i, node in enumerate(beam):
            states = [layer[i:i + 1] for layer in states_output]
            scores = scores_output[i]
            alignment = states[-1][0]
            misalignment = 0.0
            if node.length > 1:
                prev_alignment = node.alignment
                prev_source_pos = float(np.matmul(np.asarray(prev_alignment, np.float64), np.arange(T)))
                source_pos = float(np.matmul(alignment.astype(np.float64), np.arange(T)))
                misalignment = abs(source_pos - prev_source_pos - 1)
                if np.max(prev_alignment) == 1.0:
                    source_pos = int(prev_source_pos) + 1
                else:
                    source_pos = int(round(source_pos))     # round-half-even like ndarray.round
            else:
                source_pos = 0
            source_scores = source_seq[source_pos]          # IndexError quirk A.9 (6) kept
            if (m.rejection_threshold
                    and (misalignment < 0.1 or (len(node.alignment) and np.max(node.alignment) == 1.0))
                    and np.any(source_scores)):
                rej_idx = int(np.nanargmax(source_scores))
                if float(scores[rej_idx]) < m.rejection_threshold:
                    scores[rej_idx] = m.rejection_threshold
            else:
                rej_idx = None
            scores_order = np.argsort(scores, kind='stable')
            highest = scores[scores_order[-1]]
            beampos = V - int(np.searchsorted(scores[scores_order].astype(np.float64),
                                              float(highest) * m.beam_threshold_in))
            beampos = min(beampos, m.beam_width_in)
            pos = 0
            for idx in reversed(scores_order):
                idx = int(idx)
                pos += 1
                score = scores[idx]
                with np.errstate(divide='ignore'):
                    logscore = -np.log(score)
                alignment1 = alignment
                if rej_idx is not None and idx == rej_idx:
                    alignment1 = np.eye(T, dtype=alignment.dtype)[source_pos]
                    rej_idx = None
                elif pos > beampos:
                    if rej_idx:
                        continue
                    else:
                        break
                value = i_c[idx]
                if np.isnan(logscore) or value == '':
                    continue
                scores1 = np.copy(scores)
                scores[idx] = 0
                insort_left(next_beam, Node(parent=node, state=states, value=value, scores=scores1,
                                            prob=score, cost=logscore, alignment=alignment1))
        if len(next_beam) > max_batches * m.batch_size:
            next_beam = next_beam[-max_batches * m.batch_size:]
    if stats is not None:
        stats['steps'] = steps_run
        stats['finals'] = len(final_beam)
        stats['left'] = len(next_beam)
    while final_beam:
        node = final_beam.pop()
        nodes = node.to_sequence()[1:]
        yield (''.join(n.value for n in nodes),
               [n.prob for n in nodes],
               node.cum_cost / (node.length - 1),
               [n.alignment for n in nodes])


def correct_lines(m, lines, conf=None, fast=True, greedy=True):
    """seq2seq.py:782-842."""
    assert not fast or greedy, "cannot decode in fast mode with beam search enabled"
    if not lines:
        return [], [], [], []
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines], conf)
    if fast:
        _, out_lines, out_probs, out_scores, aligns = decode_batch_greedy(m, enc_in)
        return out_lines, out_probs, out_scores, aligns
    enc = m.encode(enc_in)
    out_lines, out_probs, out_scores, aligns = [], [], [], []
    for j, input_line in enumerate(lines):
        if not input_line:
            line, probs, score, alignment = '', [], 0, []
        elif greedy:
            line, probs, score, alignment = decode_sequence_greedy(
                m, encoder_outputs=[e[j:j + 1] for e in enc])
        else:
            try:
                line, probs, score, alignment = next(decode_sequence_beam(
                    m, source_seq=enc_in[j], encoder_outputs=[e[j:j + 1] for e in enc]))
            except StopIteration:
                if isinstance(input_line[0], tuple):
                    line = ''.join(chunk[0] for chunk in input_line)
                if isinstance(input_line[0], list):
                    line = ''.join(chunk[0][0] if chunk else '' for chunk in input_line)
                else:
                    line = input_line
                probs = [1.0] * len(line)
                score = 0
                alignment = np.eye(len(line)).tolist()
        line = line.replace(GAP, '')
        out_lines.append(line)
        out_probs.append(probs)
        out_scores.append(score)
        aligns.append(alignment)
    return out_lines, out_probs, out_scores, aligns
