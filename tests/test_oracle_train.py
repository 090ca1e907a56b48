"""The oracle's hand-derived backward (oracle/train.py) against torch autograd on an independent
torch restatement of the same forward, in float64."""
import numpy as np
import pytest
import torch

from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
from oracle.decode import OracleModel
from oracle.train import forward_backward, adam_step, EPS


def _cell(x, h, c, K, R, b, W):
    z = x @ K + h @ R + b
    i, f, g, o = torch.sigmoid(z[:, :W]), torch.sigmoid(z[:, W:2 * W]), torch.tanh(z[:, 2 * W:3 * W]), torch.sigmoid(z[:, 3 * W:])
    c = f * c + i * g
    return o * torch.tanh(c), c


def _lstm(x, K, R, b, W, h=None, c=None, reverse=False):
    B, T, _ = x.shape
    h = torch.zeros(B, W, dtype=x.dtype) if h is None else h
    c = torch.zeros(B, W, dtype=x.dtype) if c is None else c
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        h, c = _cell(x[:, t], h, c, K, R, b, W)
        out[t] = h
    return torch.stack(out, 1), h, c


def torch_loss(cfg, w, enc_in, dec_in, dec_out, weights, masks):
    d, W, C = cfg.depth, cfg.width, cfg.ctx_width
    E = w['E']
    x0 = enc_in @ E
    f, _, _ = _lstm(x0, w['enc1_fw_K'], w['enc1_fw_R'], w['enc1_fw_b'], W)
    bseq, hb, cb = _lstm(x0, w['enc1_bw_K'], w['enc1_bw_R'], w['enc1_bw_b'], W, reverse=True)
    out = torch.cat([f, bseq], 2) * masks['enc'][0]
    fin = [(hb, cb)]
    res, bridged = cfg.residual_connections, cfg.bridge_dense
    for n in range(2, d + 1):
        if cfg.deep_bidirectional_encoder:          # seq2seq.py:246-281, written as the Keras layers are
            xin = out + out.reshape(out.shape[:-1] + (out.shape[-1] // 2, 2)).flip(-1).reshape(out.shape)
            f2, _, _ = _lstm(xin, w['enc%d_fw_K' % n], w['enc%d_fw_R' % n], w['enc%d_fw_b' % n], W)
            b2, h, c = _lstm(xin, w['enc%d_bw_K' % n], w['enc%d_bw_R' % n], w['enc%d_bw_b' % n], W, reverse=True)
            out = torch.cat([f2, b2], 2) * masks['enc'][n - 1]
            fin.append((h, c))
            continue
        hs, h, c = _lstm(out, w['enc%d_K' % n], w['enc%d_R' % n], w['enc%d_b' % n], W)
        if res and n >= 3:                  # the training graph of seq2seq.py:284-291, written as the Keras layers are
            hs = hs + out
        out = hs * masks['enc'][n - 1]
        fin.append((h, c))
    if bridged:                             # seq2seq.py:299-301
        fin = [(torch.tanh(h @ w['bridge%d_h_K' % (n + 1)] + w['bridge%d_h_b' % (n + 1)]),
                torch.tanh(c @ w['bridge%d_c_K' % (n + 1)] + w['bridge%d_c_b' % (n + 1)])) for n, (h, c) in enumerate(fin)]
    enc_out = out
    u = enc_out @ w['att_U']
    y = dec_in @ E
    for n in range(1, d):
        hs, _, _ = _lstm(y, w['dec%d_K' % n], w['dec%d_R' % n], w['dec%d_b' % n], W, fin[n - 1][0], fin[n - 1][1])
        if res and n >= 2:                  # seq2seq.py:359-360 (n > 0 counted from 0)
            hs = hs + y
        y = hs * masks['dec'][n - 1]
    h, c = fin[d - 1]
    B, T = enc_in.shape[:2]
    a = torch.zeros(B, T, dtype=E.dtype)
    steps = torch.arange(T, dtype=E.dtype)
    outs = []
    for t in range(dec_in.shape[1]):
        wq = h @ w['att_Wa'] + w['att_bUW']
        e = torch.exp(torch.tanh(wq[:, None, :] + u) @ w['att_va'] + w['att_bv'][0])
        with torch.no_grad():
            tprime = a @ steps + 1.0
            mask = ((tprime[:, None] - steps[None, :]).abs() <= cfg.window).to(E.dtype)
        e = e * mask
        a = e / e.sum(1, keepdim=True)
        ctx = (a[:, :, None] * enc_out).sum(1)
        x = torch.cat([y[:, t], ctx], 1) * masks['cell']
        h, c = _cell(x, h, c, w['dec%d_K' % d], w['dec%d_R' % d], w['dec%d_b' % d], W)
        outs.append(h)
    top = torch.stack(outs, 1)
    if res and d >= 2:
        top = top + y
    P = torch.softmax(top @ E.T, dim=2)
    P = P / P.sum(2, keepdim=True)
    ce = -(dec_out * torch.log(torch.clamp(P, EPS, 1 - EPS))).sum(2)
    cnt = max(int((weights != 0).sum()), 1)
    loss = (ce * weights).sum() / cnt
    mean_rest = E[1:].mean(0).detach()
    reg = ((E[0] - mean_rest) ** 2).sum() + 0.01 * ((1 - (E * E).sum(1)) ** 2).sum()
    return loss + reg


@pytest.mark.parametrize('d,with_masks,flags', [(1, False, {}), (2, True, {}), (3, False, {}),
                                                (4, True, dict(residual_connections=True)), (2, True, dict(residual_connections=True)),
                                                (3, True, dict(bridge_dense=True)), (1, False, dict(bridge_dense=True)),
                                                (4, True, dict(residual_connections=True, bridge_dense=True)),
                                                (3, True, dict(deep_bidirectional_encoder=True)), (2, False, dict(deep_bidirectional_encoder=True)),
                                                (3, True, dict(deep_bidirectional_encoder=True, bridge_dense=True, residual_connections=True))])
def test_backward_matches_autograd(d, with_masks, flags):
    W, V = 16, 12
    cfg = ModelConfig(depth=d, width=W, voc_size=V, **flags)
    w = make_weights(cfg, dtype=np.float64, emb_scale=3.0)
    rng = np.random.default_rng(4)
    for k in w:
        if k.endswith('_b') or k in ('att_bUW', 'att_bv'):
            w[k] = rng.normal(size=w[k].shape) * 0.3
    m = OracleModel(cfg, w)
    src, _ = make_lines(4, 9, 1, voc_size=V)
    tgt, _ = make_lines(4, 9, 2, voc_size=V)
    tgt[1] = tgt[1][:5] + '\n'                      # ragged targets: padded steps carry weight 0
    src[2] = src[2][:6] + '\n'
    enc_in, dec_in, dec_out, wts = vectorize_lines(m, src, tgt)
    C = cfg.ctx_width
    if with_masks:
        keep = lambda shape: (rng.random(shape) > 0.2) / 0.8
        masks = {'enc': [keep(2 * W if (n == 0 or cfg.deep_bidirectional_encoder) else W) for n in range(d)], 'dec': [keep(W) for _ in range(d - 1)],
                 'cell': keep((4, W + C))}
    else:
        masks = {'enc': [np.ones(2 * W if (n == 0 or cfg.deep_bidirectional_encoder) else W) for n in range(d)], 'dec': [np.ones(W) for _ in range(d - 1)],
                 'cell': np.ones((4, W + C))}
    loss, grads, aux = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts, masks)
    tw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    tm = {'enc': [torch.tensor(x) for x in masks['enc']], 'dec': [torch.tensor(x) for x in masks['dec']],
          'cell': torch.tensor(masks['cell'])}
    tl = torch_loss(cfg, tw, torch.tensor(enc_in.astype(np.float64)), torch.tensor(dec_in.astype(np.float64)),
                    torch.tensor(dec_out.astype(np.float64)), torch.tensor(wts.astype(np.float64)), tm)
    tl.backward()
    assert abs(float(tl) - loss) < 1e-10
    for k in w:
        tg = tw[k].grad.numpy()
        assert np.allclose(grads[k], tg, rtol=1e-7, atol=1e-10), (k, np.abs(grads[k] - tg).max())


def test_adam_clipnorm_known_answer():
    w = {'a': np.array([1.0, 2.0]), 'b': np.array([[3.0]])}
    g = {'a': np.array([3.0, 4.0]), 'b': np.array([[12.0]])}           # global norm 13 > 5
    st = {'t': 0, 'm': {}, 'v': {}}
    norm = adam_step(w, g, st)
    assert abs(norm - 13.0) < 1e-12
    gs = 3.0 * 5 / 13
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    m, v = 0.1 * gs, 0.001 * gs * gs
    assert abs(w['a'][0] - (1.0 - lr_t * m / (np.sqrt(v) + 1e-7))) < 1e-15
    # below the threshold gradients pass unchanged
    w2 = {'a': np.array([1.0])}
    st2 = {'t': 0, 'm': {}, 'v': {}}
    adam_step(w2, {'a': np.array([0.5])}, st2)
    assert abs(st2['m']['a'][0] - 0.05) < 1e-15


def test_training_reduces_loss():
    cfg = ModelConfig(depth=2, width=32, voc_size=20)
    w = make_weights(cfg, dtype=np.float64, emb_scale=1.0)
    m = OracleModel(cfg, w)
    src, _ = make_lines(8, 8, 5, voc_size=20)
    enc_in, dec_in, dec_out, wts = vectorize_lines(m, src, src)      # copy task
    st = {'t': 0, 'm': {}, 'v': {}}
    losses = []
    for _ in range(12):
        loss, grads, _ = forward_backward(cfg, w, enc_in, dec_in, dec_out, wts)
        losses.append(loss)
        adam_step(w, grads, st, lr=5e-3)
    assert losses[-1] < losses[0] - 0.05
