"""Teacher-forced train step restated in numpy -- test infrastructure.

Forward = the training graph `encoder_decoder_model` (seq2seq.py:237-390): encoder (A.3), decoder LSTMs
over the whole target sequence with the encoder's final states as initial states (seq2seq.py:338-339,
351-354), attention cell unrolled over the target (attention.py:255-305, 526-575), tied softmax
(seq2seq.py:379).  Loss = Keras `categorical_crossentropy` with temporal sample weights
(seq2seq.py:494-497) + the embedding regulariser (seq2seq.py:530-553).
Backward is derived by hand (BPTT); tests/test_oracle_train.py checks it against torch autograd.
Update = Keras Adam(lr 1e-3, beta 0.9/0.999, eps 1e-7) with global-norm clipping at 5 (seq2seq.py:496).

Dropout (seq2seq.py:293-298, 345, 363-367) enters as explicit masks: `masks['enc'][n]` (F,) on the output
sequence of encoder layer n+1, `masks['dec'][n]` (W,) on decoder layer n+1 < depth, `masks['cell']`
(B, W+C) on the attention cell's input [y | ctx]; None = inference-like (all ones).
"""
import numpy as np

EPS = 1e-7          # K.epsilon()


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_forward(x_seq, K, R, b, h0=None, c0=None, reverse=False):
    B, T, _ = x_seq.shape
    W = R.shape[0]
    dt = x_seq.dtype
    h = np.zeros((B, W), dt) if h0 is None else h0
    c = np.zeros((B, W), dt) if c0 is None else c0
    cache = {'x': x_seq, 'h0': h, 'c0': c, 'K': K, 'R': R, 'reverse': reverse,
             'i': np.empty((B, T, W), dt), 'f': np.empty((B, T, W), dt), 'g': np.empty((B, T, W), dt),
             'o': np.empty((B, T, W), dt), 'c': np.empty((B, T, W), dt), 'h': np.empty((B, T, W), dt)}
    xK = x_seq @ K
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = xK[:, t] + h @ R + b
        i, f, g, o = _sig(z[:, :W]), _sig(z[:, W:2 * W]), np.tanh(z[:, 2 * W:3 * W]), _sig(z[:, 3 * W:])
        c = f * c + i * g
        h = o * np.tanh(c)
        for k, v in (('i', i), ('f', f), ('g', g), ('o', o), ('c', c), ('h', h)):
            cache[k][:, t] = v
    return cache['h'], h, c, cache


def lstm_backward(cache, dH, dh_fin=None, dc_fin=None):
    """dH (B,T,W): gradient w.r.t. the output sequence; dh_fin/dc_fin: w.r.t. the final state.
    Returns dX (B,T,in), dK, dR, db, dh0, dc0."""
    x, K, R = cache['x'], cache['K'], cache['R']
    B, T, W = cache['h'].shape
    dt = x.dtype
    dh = np.zeros((B, W), dt) if dh_fin is None else dh_fin.copy()
    dc = np.zeros((B, W), dt) if dc_fin is None else dc_fin.copy()
    dZ = np.empty((B, T, 4 * W), dt)
    Hprev = np.empty((B, T, W), dt)
    order = list(range(T - 1, -1, -1) if cache['reverse'] else range(T))
    for k in range(T - 1, -1, -1):        # reverse of the processing order
        t = order[k]
        tp = order[k - 1] if k > 0 else None
        h_prev = cache['h'][:, tp] if tp is not None else cache['h0']
        c_prev = cache['c'][:, tp] if tp is not None else cache['c0']
        i, f, g, o, c = (cache[n][:, t] for n in 'ifgoc')
        dh_t = dh + dH[:, t]
        tc = np.tanh(c)
        do = dh_t * tc
        dct = dh_t * o * (1 - tc * tc) + dc
        dz = np.concatenate([dct * g * i * (1 - i), dct * c_prev * f * (1 - f), dct * i * (1 - g * g),
                             do * o * (1 - o)], axis=1)
        dZ[:, t] = dz
        Hprev[:, t] = h_prev
        dc = dct * f
        dh = dz @ R.T
    dX = dZ @ K.T
    dK = np.einsum('bti,btj->ij', x, dZ)
    dR = np.einsum('bti,btj->ij', Hprev, dZ)
    db = dZ.sum(axis=(0, 1))
    return dX, dK, dR, db, dh, dc


def forward_backward(cfg, w, enc_in, dec_in, dec_out, weights, masks=None, want_grads=True):
    """Returns (loss, grads dict with the keys of `w`, aux)."""
    d, W, V, C = cfg.depth, cfg.width, cfg.voc_size, cfg.ctx_width
    dt = w['E'].dtype
    E = w['E']
    B, T, _ = enc_in.shape
    U = dec_in.shape[1]
    one = lambda n: np.ones(n, dt)
    deep = bool(getattr(cfg, 'deep_bidirectional_encoder', False))
    m_enc = [one(2 * W if (n == 0 or deep) else W) for n in range(d)] if masks is None else [np.asarray(m, dt) for m in masks['enc']]
    m_dec = [one(W) for _ in range(d - 1)] if masks is None else [np.asarray(m, dt) for m in masks['dec']]
    m_cell = np.ones((B, W + C), dt) if masks is None else np.asarray(masks['cell'], dt)

    # ---------------- encoder ----------------
    Xe = enc_in.astype(dt)
    x0 = Xe @ E
    _, _, _, cf = lstm_forward(x0, w['enc1_fw_K'], w['enc1_fw_R'], w['enc1_fw_b'])
    _, hb, cb, cbw = lstm_forward(x0, w['enc1_bw_K'], w['enc1_bw_R'], w['enc1_bw_b'], reverse=True)
    H = [np.concatenate([cf['h'], cbw['h']], axis=2)]
    O = [H[0] * m_enc[0]]
    enc_caches = [None]
    fin = [(hb, cb)]
    res = bool(getattr(cfg, 'residual_connections', False))
    cross = lambda x: x + x.reshape(x.shape[:-1] + (x.shape[-1] // 2, 2))[..., ::-1].reshape(x.shape)     # seq2seq.py:246-259 (model.cross_sum)
    for n in range(2, d + 1):
        if deep:            # every layer bidirectional on the cross sum of the (dropped-out) layer below; backward final states handed on
            xin = cross(O[-1])
            _, _, _, cfw = lstm_forward(xin, w['enc%d_fw_K' % n], w['enc%d_fw_R' % n], w['enc%d_fw_b' % n])
            _, h, c, cbw2 = lstm_forward(xin, w['enc%d_bw_K' % n], w['enc%d_bw_R' % n], w['enc%d_bw_b' % n], reverse=True)
            enc_caches.append((cfw, cbw2))
            Hn = np.concatenate([cfw['h'], cbw2['h']], axis=2)
            H.append(Hn); O.append(Hn * m_enc[n - 1]); fin.append((h, c))
            continue
        Hn, h, c, cache = lstm_forward(O[-1], w['enc%d_K' % n], w['enc%d_R' % n], w['enc%d_b' % n])
        enc_caches.append(cache)
        # residual_connections (seq2seq.py:284-291): output sequence = LSTM output + the layer's (already dropped-out) input
        # sequence, from the third layer on
        Yn = Hn + O[-1] if (res and n >= 3) else Hn
        H.append(Hn); O.append(Yn * m_enc[n - 1]); fin.append((h, c))
    enc_out = O[-1]
    u = enc_out @ w['att_U']
    # bridge_dense (seq2seq.py:299-301): the final states through Dense(width, tanh) on their way to the decoder
    bridged = bool(getattr(cfg, 'bridge_dense', False))
    fin_raw = fin
    if bridged:
        fin = [(np.tanh(h @ w['bridge%d_h_K' % (n + 1)] + w['bridge%d_h_b' % (n + 1)]),
                np.tanh(c @ w['bridge%d_c_K' % (n + 1)] + w['bridge%d_c_b' % (n + 1)])) for n, (h, c) in enumerate(fin_raw)]

    # ---------------- decoder ----------------
    Yd = dec_in.astype(dt)
    y = Yd @ E
    dec_caches = []
    Y = [y]
    for n in range(1, d):
        G, _, _, cache = lstm_forward(Y[-1], w['dec%d_K' % n], w['dec%d_R' % n], w['dec%d_b' % n],
                                      h0=fin[n - 1][0], c0=fin[n - 1][1])
        dec_caches.append(cache)
        # (seq2seq.py:359-360: `if n > 0 and residual_connections` with n counted from 0 -- from the second decoder layer on)
        Y.append((G + Y[-1] if (res and n >= 2) else G) * m_dec[n - 1])
    # attention cell (top layer)
    Kd, Rd, bd = w['dec%d_K' % d], w['dec%d_R' % d], w['dec%d_b' % d]
    Wa, va, bUW, bv = w['att_Wa'], w['att_va'], w['att_bUW'], w['att_bv'][0]
    h, c = fin[d - 1]
    a = np.zeros((B, T), dt)
    steps = np.arange(T)
    top = {k: np.empty((B, U, W), dt) for k in 'ifgoch'}
    top.update({'hprev': np.empty((B, U, W), dt), 'cprev': np.empty((B, U, W), dt), 'a': np.empty((B, U, T), dt),
                'x': np.empty((B, U, W + C), dt), 'th': [None] * U})
    for t in range(U):
        wq = h @ Wa + bUW
        th = np.tanh(wq[:, None, :] + u)                     # (B,T,W)
        e = np.exp(th @ va + bv)
        tprime = (a.astype(np.float64) @ steps + 1.0).astype(dt)
        mask = (np.abs(tprime[:, None] - steps[None, :].astype(dt)) <= dt.type(cfg.window)).astype(dt)
        e = e * mask
        a = e / e.sum(axis=1, keepdims=True)
        ctx = (a[:, :, None] * enc_out).sum(axis=1)
        x = np.concatenate([Y[-1][:, t], ctx], axis=1) * m_cell
        top['hprev'][:, t], top['cprev'][:, t], top['a'][:, t], top['x'][:, t], top['th'][t] = h, c, a, x, th
        z = x @ Kd + h @ Rd + bd
        i, f, g, o = _sig(z[:, :W]), _sig(z[:, W:2 * W]), np.tanh(z[:, 2 * W:3 * W]), _sig(z[:, 3 * W:])
        c = f * c + i * g
        h = o * np.tanh(c)
        for k, v in (('i', i), ('f', f), ('g', g), ('o', o), ('c', c), ('h', h)):
            top[k][:, t] = v
    G = top['h']
    if res and d >= 2:          # the top layer's sum: cell output + the cell's input sequence (before the cell's own input dropout)
        G = G + Y[-1]
    logits = G @ E.T
    logits = logits - logits.max(axis=2, keepdims=True)
    P = np.exp(logits)
    P = P / P.sum(axis=2, keepdims=True)

    # ---------------- loss ----------------
    Yt = dec_out.astype(dt)
    wts = weights.astype(dt)
    Pn = P / P.sum(axis=2, keepdims=True)
    Pc = np.clip(Pn, EPS, 1 - EPS)
    ce = -(Yt * np.log(Pc)).sum(axis=2)
    cnt = max(np.count_nonzero(wts), 1)
    loss_ce = (ce * wts).sum() / cnt
    mean_rest = E[1:].mean(axis=0)
    norms = (E * E).sum(axis=1)
    reg = ((E[0] - mean_rest) ** 2).sum() + 0.01 * ((1 - norms) ** 2).sum()
    loss = loss_ce + reg
    aux = {'probs': P, 'loss_ce': loss_ce, 'reg': reg, 'enc_out': enc_out}
    if not want_grads:
        return loss, None, aux

    # ---------------- backward ----------------
    g_ = {k: np.zeros_like(v) for k, v in w.items()}
    inrange = ((Pn > EPS) & (Pn < 1 - EPS)).astype(dt)       # tf.clip_by_value passes gradient inside only
    tgt_ok = (Yt * inrange).sum(axis=2, keepdims=True)        # 1 where the target's probability is unclipped
    dlogits = (P * Yt.sum(axis=2, keepdims=True) - Yt) * tgt_ok * (wts / cnt)[:, :, None]
    g_['E'] += np.einsum('buv,buw->vw', dlogits, G)
    dG = dlogits @ E
    # regulariser
    g_['E'][0] += 2 * (E[0] - mean_rest)
    g_['E'] += (-0.04 * (1 - norms))[:, None] * E

    d_enc_out = np.zeros_like(enc_out)
    du = np.zeros_like(u)
    dYtop = np.zeros((B, U, W), dt)
    dh = np.zeros((B, W), dt)
    dc = np.zeros((B, W), dt)
    dKd = np.zeros_like(Kd); dRd = np.zeros_like(Rd); dbd = np.zeros_like(bd)
    for t in range(U - 1, -1, -1):
        i, f, g, o, c = (top[n][:, t] for n in 'ifgoc')
        h_prev, c_prev, a, x, th = top['hprev'][:, t], top['cprev'][:, t], top['a'][:, t], top['x'][:, t], top['th'][t]
        dh_t = dh + dG[:, t]
        tc = np.tanh(c)
        do = dh_t * tc
        dct = dh_t * o * (1 - tc * tc) + dc
        dz = np.concatenate([dct * g * i * (1 - i), dct * c_prev * f * (1 - f), dct * i * (1 - g * g),
                             do * o * (1 - o)], axis=1)
        dKd += x.T @ dz; dRd += h_prev.T @ dz; dbd += dz.sum(axis=0)
        dc = dct * f
        dh = dz @ Rd.T
        dx = (dz @ Kd.T) * m_cell
        dYtop[:, t] = dx[:, :W]
        dctx = dx[:, W:]
        # attention backward (no gradient through the window mask nor through a_prev, attention.py:567)
        d_enc_out += a[:, :, None] * dctx[:, None, :]
        da = np.einsum('btc,bc->bt', enc_out, dctx)
        dscore = a * (da - (a * da).sum(axis=1, keepdims=True))
        g_['att_bv'][0] += dscore.sum()
        g_['att_va'] += np.einsum('bt,btw->w', dscore, th)
        dpre = dscore[:, :, None] * va[None, None, :] * (1 - th * th)
        du += dpre
        dwq = dpre.sum(axis=1)
        g_['att_bUW'] += dwq.sum(axis=0)
        g_['att_Wa'] += h_prev.T @ dwq
        dh = dh + dwq @ Wa.T
    g_['dec%d_K' % d], g_['dec%d_R' % d], g_['dec%d_b' % d] = dKd, dRd, dbd
    if res and d >= 2:
        dYtop = dYtop + dG
    dfin = [[np.zeros((B, W), dt), np.zeros((B, W), dt)] for _ in range(d)]
    dfin[d - 1] = [dh, dc]
    # decoder-independent half of the attention: u = enc_out . U_a
    g_['att_U'] += np.einsum('btc,btw->cw', enc_out, du)
    d_enc_out += du @ w['att_U'].T
    # lower decoder layers
    dYn = dYtop
    for n in range(d - 1, 0, -1):
        dGn = dYn * m_dec[n - 1]
        dX, dK, dR, db, dh0, dc0 = lstm_backward(dec_caches[n - 1], dGn)
        g_['dec%d_K' % n], g_['dec%d_R' % n], g_['dec%d_b' % n] = dK, dR, db
        dfin[n - 1][0] += dh0; dfin[n - 1][1] += dc0
        dYn = dX + dGn if (res and n >= 2) else dX
    g_['E'] += np.einsum('buv,buw->vw', Yd, dYn)
    if bridged:                 # back through the bridges: tanh', the Dense layers' gradients, the encoder's own final states
        for n in range(d):
            for k, part in enumerate('hc'):
                dpre = dfin[n][k] * (1 - fin[n][k] ** 2)
                g_['bridge%d_%s_K' % (n + 1, part)] += fin_raw[n][k].T @ dpre
                g_['bridge%d_%s_b' % (n + 1, part)] += dpre.sum(axis=0)
                dfin[n][k] = dpre @ w['bridge%d_%s_K' % (n + 1, part)].T
    # encoder
    dO = d_enc_out
    for n in range(d, 1, -1):
        dHn = dO * m_enc[n - 1]
        if deep:
            cfw, cbw2 = enc_caches[n - 1]
            dXf, g_['enc%d_fw_K' % n], g_['enc%d_fw_R' % n], g_['enc%d_fw_b' % n], _, _ = lstm_backward(cfw, dHn[:, :, :W])
            dXb, g_['enc%d_bw_K' % n], g_['enc%d_bw_R' % n], g_['enc%d_bw_b' % n], _, _ = lstm_backward(cbw2, dHn[:, :, W:], dfin[n - 1][0], dfin[n - 1][1])
            dO = cross(dXf + dXb)           # (the cross sum is its own adjoint: x + P x with a symmetric permutation P)
            continue
        dX, dK, dR, db, _, _ = lstm_backward(enc_caches[n - 1], dHn, dfin[n - 1][0], dfin[n - 1][1])
        g_['enc%d_K' % n], g_['enc%d_R' % n], g_['enc%d_b' % n] = dK, dR, db
        dO = dX + dHn if (res and n >= 3) else dX
    dH1 = dO * m_enc[0]
    dXf, g_['enc1_fw_K'], g_['enc1_fw_R'], g_['enc1_fw_b'], _, _ = lstm_backward(cf, dH1[:, :, :W])
    dXb, g_['enc1_bw_K'], g_['enc1_bw_R'], g_['enc1_bw_b'], _, _ = lstm_backward(cbw, dH1[:, :, W:], dfin[0][0], dfin[0][1])
    g_['E'] += np.einsum('btv,btw->vw', Xe, dXf + dXb)
    return loss, g_, aux


def adam_step(w, grads, state, lr=1e-3, beta1=0.9, beta2=0.999, eps=EPS, clipnorm=5.0, frozen=()):
    """Keras Adam with global-norm clipping; `state` = {'t': int, 'm': {...}, 'v': {...}}."""
    names = [k for k in w if not any(k.startswith(p) for p in frozen)]
    norm = np.sqrt(sum(float((grads[k].astype(np.float64) ** 2).sum()) for k in names))
    scale = clipnorm / norm if norm >= clipnorm else 1.0
    state['t'] += 1
    t = state['t']
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in names:
        g = grads[k] * w[k].dtype.type(scale)
        m = state['m'].setdefault(k, np.zeros_like(w[k]))
        v = state['v'].setdefault(k, np.zeros_like(w[k]))
        m[...] = beta1 * m + (1 - beta1) * g
        v[...] = beta2 * v + (1 - beta2) * g * g
        w[k] = w[k] - (lr_t * m / (np.sqrt(v) + eps)).astype(w[k].dtype)
    return norm
