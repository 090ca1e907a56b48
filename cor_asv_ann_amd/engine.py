"""Object wrapper over the C ABI (include/cor_asv_ann_hip.h): numpy in, numpy out.

This is plumbing between the `Sequence2Sequence` facade and the HIP library; all arithmetic of
the hot path happens in the library's kernels.
"""
import ctypes
from ctypes import byref, c_double, c_int64, c_void_p

import numpy as np

from . import _native as nv


def weight_shapes(depth, width, voc_size, bridge_dense=False, deep_bidirectional_encoder=False):
    """Ordered {name: shape} of the model's tensors in Keras layout (layer creation
    order of seq2seq.py:239-350 and attention.py:598-609; the bridge_dense layers of seq2seq.py:299-301 last)."""
    d, W, V = depth, width, voc_size
    deep = bool(deep_bidirectional_encoder)
    C = 2 * W if (d == 1 or deep) else W
    shapes = {'E': (V, W)}
    for direction in ('fw', 'bw'):
        shapes['enc1_%s_K' % direction] = (W, 4 * W)
        shapes['enc1_%s_R' % direction] = (W, 4 * W)
        shapes['enc1_%s_b' % direction] = (4 * W,)
    for n in range(2, d + 1):
        if deep:                        # every layer bidirectional, 2W-wide inputs (seq2seq.py:273-276)
            for direction in ('fw', 'bw'):
                shapes['enc%d_%s_K' % (n, direction)] = (2 * W, 4 * W)
                shapes['enc%d_%s_R' % (n, direction)] = (W, 4 * W)
                shapes['enc%d_%s_b' % (n, direction)] = (4 * W,)
            continue
        shapes['enc%d_K' % n] = (2 * W if n == 2 else W, 4 * W)
        shapes['enc%d_R' % n] = (W, 4 * W)
        shapes['enc%d_b' % n] = (4 * W,)
    shapes['att_U'] = (C, W)
    for n in range(1, d):
        shapes['dec%d_K' % n] = (W, 4 * W)
        shapes['dec%d_R' % n] = (W, 4 * W)
        shapes['dec%d_b' % n] = (4 * W,)
    shapes['att_Wa'] = (W, W)
    shapes['att_va'] = (W,)
    shapes['att_bUW'] = (W,)
    shapes['att_bv'] = (1,)
    shapes['dec%d_K' % d] = (W + C, 4 * W)
    shapes['dec%d_R' % d] = (W, 4 * W)
    shapes['dec%d_b' % d] = (4 * W,)
    if bridge_dense:
        for n in range(1, d + 1):
            for part in ('h', 'c'):
                shapes['bridge%d_%s_K' % (n, part)] = (W, W)
                shapes['bridge%d_%s_b' % (n, part)] = (W,)
    return shapes


def _pad_blocks(a, axis, blocks, W, Wp):
    """Along `axis`, `a` consists of `blocks` consecutive blocks of W entries: widen every block to Wp with zeros."""
    if W == Wp:
        return a
    a = np.asarray(a)
    shape = list(a.shape)
    assert shape[axis] == blocks * W, (shape, axis, blocks, W)
    a = np.moveaxis(a, axis, -1).reshape(a.shape[:axis] + a.shape[axis + 1:] + (blocks, W))
    out = np.zeros(a.shape[:-1] + (Wp,), a.dtype)
    out[..., :W] = a
    out = out.reshape(out.shape[:-2] + (blocks * Wp,))
    return np.ascontiguousarray(np.moveaxis(out, -1, axis))


def _strip_blocks(a, axis, blocks, W, Wp):
    """Inverse of _pad_blocks."""
    if W == Wp:
        return a
    a = np.asarray(a)
    a = np.moveaxis(a, axis, -1)
    a = a.reshape(a.shape[:-1] + (blocks, Wp))[..., :W]
    a = a.reshape(a.shape[:-2] + (blocks * W,))
    return np.ascontiguousarray(np.moveaxis(a, -1, axis))


def _width_blocks(name, depth, deep=False):
    """(blocks along axis 0, blocks along axis 1 or None) of the hidden width in a Keras-layout tensor of weight_shapes()."""
    d = depth
    top = 'dec%d_' % d
    if name == 'E':
        return (None, 1)
    if name.startswith('bridge'):
        return (1, 1) if name.endswith('_K') else (1, None)
    if name in ('att_va', 'att_bUW'):
        return (1, None)
    if name == 'att_bv':
        return (None, None)
    if name == 'att_U':
        return (2 if (d == 1 or deep) else 1, 1)
    if name == 'att_Wa':
        return (1, 1)
    if name.endswith('_b'):
        return (4, None)
    if name.endswith('_R'):
        return (1, 4)
    if name.endswith('_K'):
        if name == 'enc2_K' or (deep and name.startswith('enc') and not name.startswith('enc1_')):
            return (2, 4)
        if name == top + 'K':
            return (3 if (d == 1 or deep) else 2, 4)       # input + context (2W wide at depth 1 / with a deep bidirectional encoder)
        return (1, 4)
    raise KeyError(name)


class HipEngine(object):
    """One model handle on one HIP device.

    Hidden widths that are not a multiple of 32 (the C ABI's tile granularity; the reference's `--width` takes any integer) are
    padded here with dead units: every tensor crosses the ABI widened to the next multiple of 32, block by block, with zeros.
    That is exact, not approximate: a unit whose incoming weights, outgoing weights and biases are all zero has gate
    pre-activations 0, hence c' = f*c + i*tanh(0) = 0 and h = o*tanh(0) = 0 at every step, feeds nothing into any real unit,
    reads nothing from one, has zero attention weight (v_a = 0) -- and receives exactly zero gradient, so Adam leaves it at zero."""

    def __init__(self, depth, width, voc_size, device=0, window_width=5, residual_connections=False,
                 deep_bidirectional_encoder=False, bridge_dense=False, lm=False, stateful=False):
        self.lib = nv.load()
        self.depth, self.width, self.voc_size = int(depth), int(width), int(voc_size)
        if self.width < 1:
            raise ValueError('width must be positive')
        self.pwidth = (self.width + 31) // 32 * 32            # what the device sees
        self.deep = bool(deep_bidirectional_encoder)
        if self.deep and self.width % 32:
            # (the "cross sum" of seq2seq.py:246-259 pairs neighbouring features of [fw | bw]: dead-unit padding would move the pairs)
            raise ValueError('deep_bidirectional_encoder needs a width that is a multiple of 32')
        self.ctx_width = 2 * self.width if (self.depth == 1 or self.deep) else self.width
        self.cblocks = 2 if (self.depth == 1 or self.deep) else 1
        self.window_width = int(window_width)
        cfg = nv.Config(self.depth, self.pwidth, self.voc_size, int(window_width), int(bool(residual_connections)),
                        int(bool(deep_bidirectional_encoder)), int(bool(bridge_dense)), int(bool(lm)),
                        int(bool(stateful)))
        handle = c_void_p()
        nv.check(self.lib.casv_model_create(byref(cfg), int(device), byref(handle)))
        self.handle = handle
        self.residual_connections, self.bridge_dense = bool(residual_connections), bool(bridge_dense)
        self.shapes = weight_shapes(self.depth, self.width, self.voc_size, self.bridge_dense, self.deep)
        self.pshapes = weight_shapes(self.depth, self.pwidth, self.voc_size, self.bridge_dense, self.deep)
        self.B = self.T = 0

    # -- dead-unit padding (class docstring) ---------------------------------------------------
    def _pad_weight(self, name, a):
        W, Wp = self.width, self.pwidth
        if W == Wp:
            return a
        b0, b1 = _width_blocks(name, self.depth, self.deep)
        a = np.asarray(a, np.float32).reshape(self.shapes[name])
        if b0:
            a = _pad_blocks(a, 0, b0, W, Wp)
        if b1:
            a = _pad_blocks(a, 1, b1, W, Wp)
        return a

    def _strip_weight(self, name, a):
        W, Wp = self.width, self.pwidth
        if W == Wp:
            return a
        b0, b1 = _width_blocks(name, self.depth, self.deep)
        a = np.asarray(a).reshape(self.pshapes[name])
        if b0:
            a = _strip_blocks(a, 0, b0, W, Wp)
        if b1:
            a = _strip_blocks(a, 1, b1, W, Wp)
        return a

    def _padw(self, a, blocks=1):
        """Widen the LAST axis (blocks x width) of a state / output array."""
        return _pad_blocks(np.asarray(a, np.float32), np.asarray(a).ndim - 1, blocks, self.width, self.pwidth)

    def _stripw(self, a, blocks=1):
        return _strip_blocks(a, np.asarray(a).ndim - 1, blocks, self.width, self.pwidth)

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.casv_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------
    def set_weights(self, weights):
        for name, shape in self.shapes.items():
            if name not in weights:
                raise KeyError('missing weight "%s"' % name)
            a = nv.carray(weights[name], np.float32)
            if tuple(a.shape) != tuple(shape) and a.size != int(np.prod(shape)):
                raise ValueError('weight "%s" has shape %s, expected %s' % (name, a.shape, shape))
            a = nv.carray(self._pad_weight(name, a), np.float32)
            nv.check(self.lib.casv_set_weight(self.handle, name.encode(), nv.ptr(a), a.size))
        nv.check(self.lib.casv_commit_weights(self.handle))

    def get_weights(self):
        out = {}
        for name, shape in self.pshapes.items():
            a = np.empty(shape, np.float32)
            nv.check(self.lib.casv_get_weight(self.handle, name.encode(), nv.ptr(a), a.size))
            out[name] = self._strip_weight(name, a)
        return out

    # -- encoder -------------------------------------------------------------------------------
    @staticmethod
    def source_rejection(idx, val):
        """The rejection candidate of every input position: argmax of its input row, -1 for all-zero rows (seq2seq.py:1458-1462).
        idx / val (B,T,A).  (Host arithmetic on the whole batch: callers that prepare batches ahead of the device stage compute
        it there and hand it to encode().)"""
        masked = np.where(idx >= 0, val, -np.inf)
        best = masked.max(axis=2, keepdims=True)
        cand = np.where((masked == best) & (idx >= 0), idx, np.iinfo(np.int32).max)
        src_rej = cand.min(axis=2)
        anyval = ((idx >= 0) & (val != 0)).any(axis=2)
        return np.where(anyval, src_rej, -1).astype(np.int32)

    def encode(self, idx, val=None, src_rej=None):
        """idx int32 (B,T) or (B,T,A) with -1 for empty slots; val float32 same shape (default 1)."""
        idx = nv.carray(idx, np.int32)
        if idx.ndim == 2:
            idx = idx[:, :, None]
        idx = np.ascontiguousarray(idx)
        B, T, A = idx.shape
        if val is None:
            val = np.ones(idx.shape, np.float32)
        val = nv.carray(np.broadcast_to(np.asarray(val, np.float32).reshape(B, T, -1), idx.shape), np.float32)
        if src_rej is None:
            src_rej = self.source_rejection(idx, val)
        src_rej = nv.carray(src_rej, np.int32)
        nv.check(self.lib.casv_encode(self.handle, B, T, A, nv.ptr(idx), nv.ptr(val), nv.ptr(src_rej)))
        self.B, self.T = B, T

    def set_encoder_outputs(self, enc_out, states, a0=None, src_rej=None):
        """Install encoder outputs computed elsewhere: enc_out (B,T,C), states [h1,c1,...,hd,cd] each (B,W), a0 (B,T) or None."""
        enc_out = nv.carray(self._padw(enc_out, self.cblocks), np.float32)
        B, T = enc_out.shape[:2]
        st = nv.carray(self._padw(np.stack([np.asarray(x, np.float32).reshape(B, self.width) for x in states[:2 * self.depth]])), np.float32)
        a0 = None if a0 is None else nv.carray(np.asarray(a0, np.float32).reshape(B, T), np.float32)
        src_rej = None if src_rej is None else nv.carray(src_rej, np.int32)
        nv.check(self.lib.casv_set_encoder_outputs(self.handle, B, T, nv.ptr(enc_out), nv.ptr(st), nv.ptr(a0), nv.ptr(src_rej)))
        self.B, self.T = B, T

    def encoder_outputs(self):
        enc = np.empty((self.B, self.T, self.cblocks * self.pwidth), np.float32)
        st = np.empty((2 * self.depth, self.B, self.pwidth), np.float32)
        nv.check(self.lib.casv_get_encoder_outputs(self.handle, nv.ptr(enc), nv.ptr(st)))
        enc, st = self._stripw(enc, self.cblocks), self._stripw(st)
        return enc, [st[i] for i in range(2 * self.depth)]

    def decoder_step(self, line, p_in, states, a_in):
        line = nv.carray(line, np.int32)
        R = line.shape[0]
        p_in = nv.carray(p_in, np.float32)
        st = nv.carray(self._padw(np.stack(states[:2 * self.depth])), np.float32)
        a_in = nv.carray(a_in, np.float32)
        probs = np.empty((R, self.voc_size), np.float32)
        st_out = np.empty_like(st)
        a_out = np.empty((R, self.T), np.float32)
        nv.check(self.lib.casv_decoder_step(self.handle, R, nv.ptr(line), nv.ptr(p_in), nv.ptr(st), nv.ptr(a_in),
                                            nv.ptr(probs), nv.ptr(st_out), nv.ptr(a_out)))
        st_out = self._stripw(st_out)
        return probs, [st_out[i] for i in range(2 * self.depth)] + [a_out]

    # -- decode loops --------------------------------------------------------------------------
    def decode_greedy(self, mode=0, steps=None, want_align=False):
        S = int(steps or 2 * self.T)
        idx = np.empty((self.B, S), np.int32)
        prob = np.empty((self.B, S), np.float32)
        length = np.empty((self.B,), np.int32)
        sparse = want_align == 'sparse'        # (lo, w) windows instead of (B, S, T) rows
        align = np.empty((self.B, S, self.T), np.float32) if want_align and not sparse else None
        nv.check(self.lib.casv_decode_greedy(self.handle, int(mode), S, nv.ptr(idx), nv.ptr(prob), nv.ptr(length),
                                             nv.ptr(align)))
        if sparse:
            align = self.alignments_sparse(self.B, S)
        return idx, prob, length, align

    def decode_beam(self, batch_size=8, beam_width_in=15, beam_threshold_in=0.2, beam_width_out=16,
                    rejection_threshold=0.3, cost0=3.0, max_results=1, steps=None, want_align=False):
        S = int(steps or 2 * self.T)
        MR = int(max_results)
        p = nv.BeamParams(int(batch_size), int(beam_width_in), int(beam_width_out), MR, float(beam_threshold_in),
                          float(rejection_threshold or 0.0), float(cost0))
        n = self.B * MR
        out = {'idx': np.empty((n, S), np.int32), 'prob': np.empty((n, S), np.float32),
               'len': np.empty((n,), np.int32), 'score': np.empty((n,), np.float64),
               'rej': np.empty((n, S), np.int32),
               'align': np.empty((n, S, self.T), np.float32) if want_align and want_align != 'sparse' else None,
               'n_found': np.empty((self.B,), np.int32), 'n_steps': np.empty((self.B,), np.int32)}
        nv.check(self.lib.casv_decode_beam(self.handle, byref(p), S, nv.ptr(out['idx']), nv.ptr(out['prob']),
                                           nv.ptr(out['len']), nv.ptr(out['score']), nv.ptr(out['rej']),
                                           nv.ptr(out['align']), nv.ptr(out['n_found']), nv.ptr(out['n_steps'])))
        if want_align == 'sparse':
            out['align_sparse'] = self.alignments_sparse(n, S)
        return out

    # -- training ------------------------------------------------------------------------------
    def train_begin(self, lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-7, clipnorm=5.0, frozen=()):
        p = nv.AdamParams(lr, beta1, beta2, epsilon, clipnorm)
        csv = ','.join(frozen).encode() if frozen else None
        nv.check(self.lib.casv_train_begin(self.handle, byref(p), csv))

    def train_step(self, enc_idx, enc_val, dec_in, dec_out, weights, masks=None, mode=1):
        """One train_on_batch (mode 1), test_on_batch (mode 0) or loss+gradients (mode 2).
        enc_idx (B,T[,A]) int32, dec_in/dec_out (B,U) int32 (-1 = zero row), weights (B,U).
        masks: {'enc': [..], 'dec': [..], 'cell': (B,W+C)} of scaled keep-masks, or None."""
        enc_idx = nv.carray(enc_idx, np.int32)
        if enc_idx.ndim == 2:
            enc_idx = np.ascontiguousarray(enc_idx[:, :, None])
        B, T, A = enc_idx.shape
        enc_val = None if enc_val is None else nv.carray(np.asarray(enc_val, np.float32).reshape(B, T, A), np.float32)
        dec_in = nv.carray(dec_in, np.int32)
        dec_out = nv.carray(dec_out, np.int32)
        weights = nv.carray(weights, np.float32)
        U = dec_in.shape[1]
        m_enc = m_dec = m_cell = None
        if masks is not None:
            # (dead units are multiplied by zero whatever their mask says)
            m_enc = nv.carray(np.concatenate([self._padw(np.asarray(x, np.float32).ravel(), 2 if (n == 0 or self.deep) else 1)
                                              for n, x in enumerate(masks['enc'])]), np.float32)
            if self.depth > 1:
                m_dec = nv.carray(np.concatenate([self._padw(np.asarray(x, np.float32).ravel()) for x in masks['dec']]), np.float32)
            m_cell = nv.carray(self._padw(np.asarray(masks['cell'], np.float32), 1 + self.cblocks), np.float32)
        loss, norm = c_double(), c_double()
        nv.check(self.lib.casv_train_step(self.handle, int(mode), B, T, U, A, nv.ptr(enc_idx), nv.ptr(enc_val), nv.ptr(dec_in),
                                          nv.ptr(dec_out), nv.ptr(weights), nv.ptr(m_enc), nv.ptr(m_dec), nv.ptr(m_cell),
                                          byref(loss), byref(norm)))
        return loss.value, norm.value

    def train_gradients(self):
        out = {}
        for name, shape in self.pshapes.items():
            a = np.empty(shape, np.float32)
            nv.check(self.lib.casv_train_get_gradient(self.handle, name.encode(), nv.ptr(a), a.size))
            out[name] = self._strip_weight(name, a)
        return out

    def train_weights(self):
        nv.check(self.lib.casv_train_sync_weights(self.handle))
        return self.get_weights()

    def train_end(self):
        nv.check(self.lib.casv_train_end(self.handle))

    # -- measurement ---------------------------------------------------------------------------
    def profile(self, enable=True):
        """0/False = off, 1/True = every kernel class, 2 = only the dominant kernel (fused LSTM GEMM)."""
        nv.check(self.lib.casv_profile(self.handle, int(enable)))

    def profile_read(self, name):
        launches, ms, fl, by = c_int64(), c_double(), c_double(), c_double()
        nv.check(self.lib.casv_profile_read(self.handle, name.encode(), byref(launches), byref(ms), byref(fl),
                                            byref(by)))
        return {'launches': launches.value, 'ms': ms.value, 'flops': fl.value, 'bytes': by.value}

    def set_option(self, key, value):
        nv.check(self.lib.casv_set_option(self.handle, key.encode(), int(value)))

    def debug_contract(self, A, Bt, bias=None, split_k=False, wave_groups=False, weight=False, k_major=False):
        """Test support (casv_debug_contract): C = A . Bt^T (+ bias) through the launcher all GEMMs of the path go through;
        k_major: A (K,M), Bt (K,N), C = A^T . Bt through the weight-gradient kernels."""
        A, Bt = nv.carray(A, np.float32), nv.carray(Bt, np.float32)
        if k_major:
            (K, M), N = A.shape, Bt.shape[1]
            assert Bt.shape[0] == K and bias is None
            C = np.empty((M, N), np.float32)
            nv.check(self.lib.casv_debug_contract(self.handle, 8, M, N, K, nv.ptr(A), nv.ptr(Bt), None, nv.ptr(C)))
            return C
        bias = None if bias is None else nv.carray(bias, np.float32)
        (M, K), N = A.shape, Bt.shape[0]
        assert Bt.shape[1] == K and (bias is None or bias.shape == (N,))
        C = np.empty((M, N), np.float32)
        flags = (1 if split_k else 0) | (2 if wave_groups else 0) | (4 if weight else 0)
        nv.check(self.lib.casv_debug_contract(self.handle, flags, M, N, K, nv.ptr(A), nv.ptr(Bt), nv.ptr(bias), nv.ptr(C)))
        return C

    def alignments_sparse(self, rows, steps, K=None):
        """Window form of the last decode call's soft alignments: (lo int32 (rows, S), w float32 (rows, S, K))."""
        K = int(K or 2 * self.window_width + 1)
        lo = np.empty((rows, steps), np.int32)
        w = np.empty((rows, steps, K), np.float32)
        nv.check(self.lib.casv_get_alignments_sparse(self.handle, K, nv.ptr(lo), nv.ptr(w)))
        return lo, w

    # -- result records on the device (sharding.py) ------------------------------------------
    def records_reset(self, rows, steps):
        """A zeroed device buffer of `rows` result records of `steps` steps (2*steps+4 int32 each)."""
        self._rec_shape = (int(rows), 2 * int(steps) + 4)
        nv.check(self.lib.casv_records_reset(self.handle, int(rows), int(steps)))

    def records_append(self, row_offset):
        """Pack the best result of every line of the last decode call into records [row_offset, row_offset + B)."""
        nv.check(self.lib.casv_records_append(self.handle, int(row_offset)))

    def records_read(self):
        out = np.empty(self._rec_shape, np.int32)
        nv.check(self.lib.casv_records_read(self.handle, nv.ptr(out)))
        return out

    def records_device_ptr(self):
        """(device address, bytes) of the record buffer, after the handle's stream has drained."""
        p, n = c_void_p(), c_int64()
        nv.check(self.lib.casv_records_device_ptr(self.handle, byref(p), byref(n)))
        return p.value, n.value

    def stat(self, key):
        v = c_int64()
        nv.check(self.lib.casv_get_stat(self.handle, key.encode(), byref(v)))
        return v.value

    def synchronize(self):
        nv.check(self.lib.casv_synchronize(self.handle))
