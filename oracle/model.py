"""Encoder and one decoder step, restated in numpy -- test infrastructure.

Follows the Keras graph of ``seq2seq.py:237-314`` (encoder), ``:416-480``
(inference decoder) and ``attention.py:526-575`` (attention cell), with the
Keras layer semantics of SURVEY.md appendix A.1 (gate order i,f,c,o; logistic
recurrent activation; no masking of padded positions).
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_step(x, h, c, K, R, b, xK=None):
    """One Keras LSTMCell step (recurrent_activation='sigmoid', seq2seq.py:271,336,346).
    z = (x.K + h.R) + b; blocks i,f,c,o."""
    W = h.shape[1]
    z = (x @ K if xK is None else xK) + h @ R
    z = z + b
    i = _sigmoid(z[:, :W])
    f = _sigmoid(z[:, W:2 * W])
    g = np.tanh(z[:, 2 * W:3 * W])
    o = _sigmoid(z[:, 3 * W:])
    c2 = f * c + i * g
    h2 = o * np.tanh(c2)
    return h2, c2


def _run_lstm(x_seq, K, R, b, reverse=False):
    """Full-sequence LSTM from zero state; returns (outputs (B,T,W), h_final, c_final).
    With reverse=True this is Keras' go_backwards copy with its output re-reversed
    (Bidirectional, seq2seq.py:274-276): out[:, t] is the state after reading t..T-1 backwards."""
    B, T, _ = x_seq.shape
    W = R.shape[0]
    dt = x_seq.dtype
    h = np.zeros((B, W), dt)
    c = np.zeros((B, W), dt)
    out = np.empty((B, T, W), dt)
    xK = x_seq @ K
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h, c = lstm_step(None, h, c, K, R, b, xK=xK[:, t])
        out[:, t] = h
    return out, h, c


def bridge(cfg, w, n, h, c):
    """bridge_dense (seq2seq.py:299-301): the final states of encoder layer n pass through Dense(width, activation='tanh') layers
    'bridge_h_<n>' / 'bridge_c_<n>' on their way to decoder layer n."""
    if not getattr(cfg, 'bridge_dense', False):
        return h, c
    return (np.tanh(h @ w['bridge%d_h_K' % n] + w['bridge%d_h_b' % n]).astype(h.dtype),
            np.tanh(c @ w['bridge%d_c_K' % n] + w['bridge%d_c_b' % n]).astype(c.dtype))


def cross_sum(x):
    """The Lambda of seq2seq.py:246-259 as the code computes it: the last axis (2W) is viewed as W pairs of neighbours, each pair is
    reversed, and the result is added -- so BOTH features 2k and 2k+1 become x[2k] + x[2k+1].  (The comment there announces
    fw[k] + bw[k]; what the reshape pairs are neighbours inside the concatenation [fw | bw].)"""
    return x + x.reshape(x.shape[:-1] + (x.shape[-1] // 2, 2))[..., ::-1].reshape(x.shape)


def encode(cfg, w, x):
    """encoder_model (seq2seq.py:403-406): x (B,T,V) dense rows (one-hot / confidences / zeros
    for padding) -> [enc_out (B,T,C), h1, c1, ..., hd, cd, a0 (B,T)].

    Layer 1 is bidirectional and hands its BACKWARD final state to decoder layer 1
    (seq2seq.py:280-281); layers n>=2 are forward LSTMs handing over their final state after
    the last (possibly padded) position (seq2seq.py:283,302).

    residual_connections (seq2seq.py:284-291): from the third layer on a layer's output sequence is its LSTM output PLUS its
    input sequence (the second layer's input is 2W wide: no sum there); the states handed over are the LSTM's own.
    bridge_dense: see `bridge`."""
    dt = w['E'].dtype
    x0 = x.astype(dt) @ w['E']
    fw, _, _ = _run_lstm(x0, w['enc1_fw_K'], w['enc1_fw_R'], w['enc1_fw_b'])
    bw, hb, cb = _run_lstm(x0, w['enc1_bw_K'], w['enc1_bw_R'], w['enc1_bw_b'], reverse=True)
    out = np.concatenate([fw, bw], axis=2)
    states = list(bridge(cfg, w, 1, hb, cb))
    for n in range(2, cfg.depth + 1):
        if getattr(cfg, 'deep_bidirectional_encoder', False):
            # every layer bidirectional, fed the cross sum of the layer below, handing on its BACKWARD final state (seq2seq.py:273-281)
            xin = cross_sum(out)
            fw, _, _ = _run_lstm(xin, w['enc%d_fw_K' % n], w['enc%d_fw_R' % n], w['enc%d_fw_b' % n])
            bw, h, c = _run_lstm(xin, w['enc%d_bw_K' % n], w['enc%d_bw_R' % n], w['enc%d_bw_b' % n], reverse=True)
            out = np.concatenate([fw, bw], axis=2)
            states += list(bridge(cfg, w, n, h, c))
            continue
        out2, h, c = _run_lstm(out, w['enc%d_K' % n], w['enc%d_R' % n], w['enc%d_b' % n])
        out = out2 + out if (getattr(cfg, 'residual_connections', False) and n >= 3) else out2
        states += list(bridge(cfg, w, n, h, c))
    a0 = np.zeros(out.shape[:2], dt)          # attention_state_init, seq2seq.py:307-309
    return [out] + states + [a0]


def attention(cfg, w, h, a_prev, enc_out, u):
    """DenseAnnotationAttention.attention_call (attention.py:526-575), dense over all T then
    masked by the window around t' = sum_s a_prev[s]*s + 1.

    Spec decision: t' is accumulated in float64 and rounded once to the working
    dtype, so it does not depend on a summation order."""
    dt = h.dtype
    T = enc_out.shape[1]
    wq = h @ w['att_Wa'] + w['att_bUW']                                   # att:539
    e = np.exp(np.tanh(wq[:, None, :] + u) @ w['att_va'] + w['att_bv'][0])   # att:540, (R,T)
    steps = np.arange(T)
    tprime = (a_prev.astype(np.float64) @ steps.astype(np.float64) + 1.0).astype(dt)   # att:553
    dist = np.abs(tprime[:, None] - steps[None, :].astype(dt))
    mask = dist <= dt.type(cfg.window)             # K.relu(max=5, threshold=5) == 0, att:562-567
    e = e * mask.astype(dt)
    a = e / e.sum(axis=1, keepdims=True)          # att:571 (0/0 -> NaN if window is off the line)
    ctx = (a[:, :, None] * enc_out).sum(axis=1)   # att:572
    return ctx.astype(dt), a.astype(dt)


def decoder_step(cfg, w, p_in, enc_out, states, u=None):
    """decoder_model.predict_on_batch for one character (seq2seq.py:416-480).

    p_in (R,V): zeros at step 0, otherwise the fed-back distribution.  enc_out (R or 1,T,C).
    states = [h1,c1,...,hd,cd,a].  u = enc_out.U_a; the reference recomputes it inside every
    step (seq2seq.py:459-460) -- pass u=None to do the same.
    Returns (probs (R,V), new_states).

    residual_connections: the reference's INFERENCE decoder has none (seq2seq.py:421-436 builds `decoder_model` layer by layer
    without the `add` of the training graph, seq2seq.py:359-360) -- restated as it is; the training graph's sums are in train.py."""
    d = cfg.depth
    dt = w['E'].dtype
    with np.errstate(invalid='ignore', divide='ignore', over='ignore'):
        y = p_in.astype(dt) @ w['E']                   # char_input_proj, s2s:319,418
        new_states = []
        for n in range(1, d):
            h, c = lstm_step(y, states[2 * n - 2], states[2 * n - 1],
                             w['dec%d_K' % n], w['dec%d_R' % n], w['dec%d_b' % n])
            new_states += [h, c]
            y = h
        if u is None:
            u = enc_out @ w['att_U']                   # attention_dense, s2s:313,460
        hd, cd, a_prev = states[2 * d - 2], states[2 * d - 1], states[2 * d]
        ctx, a = attention(cfg, w, hd, a_prev, enc_out, u)
        x = np.concatenate([y, ctx], axis=1)           # input_mode="concatenate", att:341-342
        h, c = lstm_step(x, hd, cd, w['dec%d_K' % d], w['dec%d_R' % d], w['dec%d_b' % d])
        new_states += [h, c, a]
        logits = h @ w['E'].T                          # tied projection, s2s:379
        logits = logits - logits.max(axis=1, keepdims=True)
        p = np.exp(logits)
        p = p / p.sum(axis=1, keepdims=True)
    return p.astype(dt), new_states
