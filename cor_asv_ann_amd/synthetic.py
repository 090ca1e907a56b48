"""Synthetic vocabulary, weights and lines: the seeded inputs of `bench.py`, of the golden
fixtures and of the tests (the oracle re-exports them; no arithmetic of the path lives here).

Weight inventory and draw order follow the reference's layer creation order
(``seq2seq.py:239-243`` char embedding, ``:265-283`` encoder LSTMs, ``:313``
attention_dense, ``:329-350`` decoder LSTMs, ``attention.py:598-609`` attention
cell weights created before the wrapped cell's).
"""
from dataclasses import dataclass
import numpy as np

WEIGHT_SEED = 20250614


@dataclass(frozen=True)
class ModelConfig:
    depth: int = 2
    width: int = 512
    voc_size: int = 256
    window: int = 5          # attention.py:515, seq2seq.py:347
    residual_connections: bool = False       # seq2seq.py:125,284-291,359-360
    bridge_dense: bool = False               # seq2seq.py:132,299-301
    deep_bidirectional_encoder: bool = False # seq2seq.py:128,246-281: every encoder layer bidirectional, its input the "cross sum" of the layer below

    @property
    def ctx_width(self):
        # attended width: the top encoder layer is a BiLSTM only when depth == 1 -- or always, with deep_bidirectional_encoder
        # (seq2seq.py:273-295)
        return 2 * self.width if (self.depth == 1 or self.deep_bidirectional_encoder) else self.width


def make_vocabulary(voc_size=256):
    """idx 0 = '' (unknown, seq2seq.py:122), idx 1 = '\\n', then printable code points
    in sorted order (seq2seq.py:580-585 sorts the character set)."""
    chars = ['', '\n']
    cp = 0x20
    while len(chars) < voc_size:
        if cp == 0x7f:
            cp = 0xa1
        chars.append(chr(cp))
        cp += 1
    c_i = {c: i for i, c in enumerate(chars)}
    i_c = {i: c for i, c in enumerate(chars)}
    return c_i, i_c


def weight_names(cfg):
    """Ordered (name, shape) list; the order is the draw order for synthetic weights."""
    d, W, V, C = cfg.depth, cfg.width, cfg.voc_size, cfg.ctx_width
    names = [('E', (V, W))]
    for direction in ('fw', 'bw'):
        names += [('enc1_%s_K' % direction, (W, 4 * W)),
                  ('enc1_%s_R' % direction, (W, 4 * W)),
                  ('enc1_%s_b' % direction, (4 * W,))]
    for n in range(2, d + 1):
        if getattr(cfg, 'deep_bidirectional_encoder', False):
            for direction in ('fw', 'bw'):
                names += [('enc%d_%s_K' % (n, direction), (2 * W, 4 * W)), ('enc%d_%s_R' % (n, direction), (W, 4 * W)),
                          ('enc%d_%s_b' % (n, direction), (4 * W,))]
            continue
        nin = 2 * W if n == 2 else W
        names += [('enc%d_K' % n, (nin, 4 * W)), ('enc%d_R' % n, (W, 4 * W)), ('enc%d_b' % n, (4 * W,))]
    names += [('att_U', (C, W))]
    for n in range(1, d):
        names += [('dec%d_K' % n, (W, 4 * W)), ('dec%d_R' % n, (W, 4 * W)), ('dec%d_b' % n, (4 * W,))]
    names += [('att_Wa', (W, W)), ('att_va', (W,)), ('att_bUW', (W,)), ('att_bv', (1,)),
              ('dec%d_K' % d, (W + C, 4 * W)), ('dec%d_R' % d, (W, 4 * W)), ('dec%d_b' % d, (4 * W,))]
    if getattr(cfg, 'bridge_dense', False):
        # Dense(width, activation='tanh') on the final h and c of every encoder layer (seq2seq.py:299-301: 'bridge_h_<n>', 'bridge_c_<n>');
        # listed (and drawn) LAST so that the default topology's tensors are the same with and without the flag
        for n in range(1, d + 1):
            for s in ('h', 'c'):
                names += [('bridge%d_%s_K' % (n, s), (W, W)), ('bridge%d_%s_b' % (n, s), (W,))]
    return names


def make_weights(cfg, seed=WEIGHT_SEED, dtype=np.float32, emb_scale=4.0):
    """Glorot-uniform matrices, zero biases with unit forget block, attention biases 0,
    embedding ~ N(0, (emb_scale/sqrt(W))^2) (the reference's
    N(0, 0.001^2) init gives a flat softmax on which argmax parity would test noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = cfg.width
    out = {}
    for name, shape in weight_names(cfg):
        if name == 'E':
            w = rng.standard_normal(shape) * (emb_scale / np.sqrt(W))
        elif name == 'att_va':
            lim = np.sqrt(6.0 / (W + 1))          # Keras shape (W, 1)
            w = rng.uniform(-lim, lim, shape)
        elif name.startswith('bridge') and name.endswith('_b'):
            w = rng.uniform(-0.1, 0.1, shape)     # (Keras: zeros; drawn here so that a dropped bias shows in the parity tests)
        elif name.endswith('_b'):
            w = np.zeros(shape)
            w[W:2 * W] = 1.0                      # unit_forget_bias, gate order i,f,c,o
        elif name in ('att_bUW', 'att_bv'):
            w = np.zeros(shape)
        else:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-lim, lim, shape)
        out[name] = np.ascontiguousarray(w, dtype=dtype)
    return out


def make_lines(n, length, seed, voc_size=256):
    """n lines of `length` i.i.d. uniform characters from idx 2..V-1, each ending in '\\n'.
    Returns (list of str, int32 index array (n, length+1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    _, i_c = make_vocabulary(voc_size)
    idx = rng.integers(2, voc_size, size=(n, length))
    idx = np.concatenate([idx, np.ones((n, 1), dtype=idx.dtype)], axis=1).astype(np.int32)
    lines = [''.join(i_c[int(i)] for i in row) for row in idx]
    return lines, idx


def make_confmat_lines(n, length, seed, voc_size=256):
    """n synthetic confusion-network lines as the OCR-D wrapper hands them to `correct_lines(lines, conf=lines)`
    (wrapper/transcode.py:236-277,110-115): a line is a list of chunks, a chunk a list of (characters, confidence)
    alternatives, best first -- here single characters, 1 to 3 alternatives per position (60 % / 28 % / 12 %), the
    confidences of a position summing to at most 1; the last chunk is the end of line.  `length` positions + '\n'."""
    rng = np.random.Generator(np.random.PCG64(seed))
    _, i_c = make_vocabulary(voc_size)
    lines = []
    for _ in range(n):
        line = []
        for _ in range(length):
            k = int(rng.choice([1, 2, 3], p=[0.60, 0.28, 0.12]))
            chars = rng.choice(np.arange(2, voc_size), size=k, replace=False)
            conf = np.sort(rng.dirichlet(np.ones(k) * 0.7) * rng.uniform(0.7, 1.0))[::-1]
            line.append([(i_c[int(c)], float(p)) for c, p in zip(chars, conf)])
        line.append([('\n', 1.0)])
        lines.append(line)
    return lines
