"""`Sequence2Sequence` facade: the reference's Python API (ocrd_cor_asv_ann/lib/seq2seq.py:13-1544)
on top of the HIP hot path.

Same attributes, method names, argument meaning, return types and error behaviour as the reference
class, so `cor-asv-ann-proc`, `cor-asv-ann-train`, `cor-asv-ann-repl` and the OCR-D processor
(`wrapper/transcode.py:56-115`) keep working.  What differs is underneath: where the reference calls
`encoder_model.predict_on_batch` / `decoder_model.predict_on_batch` once per character
(seq2seq.py:1231,1245,1321,1428), this class hands index tensors to the C ABI once per batch and reads
back finished strings.  There is no CPU fallback: without the HIP library every compute call raises.
"""
import logging
import math
import os
import pickle
import unicodedata
from collections import OrderedDict

import numpy as np

from . import GAP
from . import _native as nv
from . import hdf5, keras_h5
from .engine import HipEngine, weight_shapes
from .realign import SparseAlignment

_UNSUPPORTED = ('lm_loss', 'lm_predict', 'scheduled_sampling', 'stateful')


class _ShortSwitchInterval(object):
    """The interpreter's thread switch interval is process-wide: a counted guard, so that overlapping pipelines (two
    `correct_batches` generators, or one that is never exhausted) neither restore each other's saved value in the wrong
    order nor leave the short interval behind.  Entered around the stretches in which a pipeline's consumer thread runs
    Python beside its device thread, left before every yield."""
    _lock = __import__('threading').Lock()
    _users = 0
    _saved = None

    def __enter__(self):
        import sys
        cls = _ShortSwitchInterval
        with cls._lock:
            if cls._users == 0:
                cls._saved = sys.getswitchinterval()
                sys.setswitchinterval(min(cls._saved, float(os.environ.get('CASV_SWITCH_INTERVAL', '1e-4'))))
            cls._users += 1
        return self

    def __exit__(self, *exc):
        import sys
        cls = _ShortSwitchInterval
        with cls._lock:
            cls._users -= 1
            if cls._users == 0 and cls._saved is not None:
                sys.setswitchinterval(cls._saved)
                cls._saved = None
        return False


class Sequence2Sequence(object):
    """Character-level encoder-attention-decoder corrector (API of seq2seq.py:13)."""

    def __init__(self, logger=None, progbars=True, device=0):
        # model parameters (seq2seq.py:108-132)
        self.batch_size = 256
        self.stateful = False
        self.width = 512
        self.depth = 2
        self.mapping = ({'': 0}, {0: ''})
        self.voc_size = 1
        self.residual_connections = False
        self.deep_bidirectional_encoder = False
        self.bridge_dense = False
        # training parameters (seq2seq.py:134-157)
        self.epochs = 100
        self.lm_loss = False
        self.lm_predict = False
        self.scheduled_sampling = None
        self.dropout = 0.2
        # beam decoder parameters (seq2seq.py:159-169)
        self.rejection_threshold = 0.3
        self.beam_width_in = 15
        self.beam_threshold_in = 0.2
        self.beam_width_out = 16
        # runtime
        self.logger = logger or logging.getLogger(__name__)
        self.progbars = progbars
        self.device = device
        # GEMM arithmetic of the device path (csrc/engine.h, arithmetic_of): 'auto' = the beam search's decoder steps contract
        # bf16x3-split fp32 operands on the bf16 matrix instruction (fp32-accurate sums), everything else -- encoder, greedy
        # decodes, train step -- the fp32-input instruction; 'fp32' / 'split' = one of the two everywhere.  Never by batch size.
        self.arithmetic = 'auto'
        self.engine = None
        self.status = 0   # empty / configured / trained (seq2seq.py:179)
        self._weights = None
        self.frozen_prefixes = []
        self._dirty = True
        self._rng = np.random.default_rng()

    def __repr__(self):
        return (__name__ + " (width: %d)" % self.width + " (depth: %d)" % self.depth +
                " (chars: %d)" % self.voc_size + " (attention)" +
                (" (stateful)" if self.stateful else " (stateless)") +
                " status: %s" % ("empty" if self.status < 1 else "configured" if self.status < 2 else "trained"))

    # ------------------------------------------------------------------------------------------
    # configuration and weights
    # ------------------------------------------------------------------------------------------
    def configure(self, batch_size=None):
        """Allocate the model for the current width/depth/voc_size with freshly initialised weights
        (seq2seq.py:190-489).  Non-default topology variants are refused, not ignored."""
        if batch_size:
            self.batch_size = batch_size
        for flag in _UNSUPPORTED:
            if getattr(self, flag):
                raise NotImplementedError('%s is not implemented in the MI355X hot path (only the default '
                                          'topology of the published models is)' % flag)
        self.logger.info('using HIP/gfx950 implementation to compile %s model of depth %d width %d size %d '
                         'with attention', 'stateless', self.depth, self.width, self.voc_size)
        if self.engine is not None:
            self.engine.close()
            self.engine = None
        self._weights = self._initial_weights()
        self._dirty = True      # the device copy is created / refreshed at the first compute call
        self.status = 1

    def _initial_weights(self):
        """Keras default initialisers of the reference's layers: embedding N(0, 0.001^2)
        (seq2seq.py:240), glorot_uniform kernels, orthogonal recurrent kernels, zero biases with
        unit forget gate, glorot_uniform / zeros in the attention cell (attention.py:509-510)."""
        rng = self._rng
        W = self.width
        out = {}
        for name, shape in weight_shapes(self.depth, W, max(self.voc_size, 1), self.bridge_dense, self.deep_bidirectional_encoder).items():
            if name == 'E':
                w = rng.standard_normal(shape) * 0.001
            elif name.endswith('_R'):
                a = rng.standard_normal(shape)
                u, _, vt = np.linalg.svd(a, full_matrices=False)
                w = u if u.shape == shape else vt
            elif name.endswith('_b'):
                w = np.zeros(shape)
                w[W:2 * W] = 1.0
            elif name in ('att_bUW', 'att_bv'):
                w = np.zeros(shape)
            else:
                fan = (shape[0] + shape[1]) if len(shape) == 2 else (shape[0] + 1)
                lim = math.sqrt(6.0 / fan)
                w = rng.uniform(-lim, lim, shape)
            out[name] = np.asarray(w, np.float32)
        return out

    def get_weights(self):
        assert self.status >= 1
        return {k: v.copy() for k, v in self._weights.items()}

    def set_weights(self, weights):
        """Install all tensors (Keras layout, names of engine.weight_shapes) and resync the device."""
        assert self.status >= 1
        shapes = weight_shapes(self.depth, self.width, self.voc_size, self.bridge_dense, self.deep_bidirectional_encoder)
        new = {}
        for name, shape in shapes.items():
            a = np.asarray(weights[name], np.float32).reshape(shape)
            new[name] = np.ascontiguousarray(a)
        self._weights = new
        self._dirty = True

    def reset_encoder(self):
        """Re-initialise the encoder tensors, keeping the decoder (the `--reset-encoder` option,
        scripts/train.py:84-93)."""
        fresh = self._initial_weights()
        w = self.get_weights()
        for name in w:
            if name.startswith('enc') or name == 'att_U':
                w[name] = fresh[name]
        self.set_weights(w)

    def _reconfigure_for_mapping(self):
        """Grow the embedding after the vocabulary grew, keeping the rows already trained
        (seq2seq.py:499-525)."""
        assert self.status >= 1
        old = self._weights
        old_voc = old['E'].shape[0] if old is not None else 0
        if old_voc < self.voc_size:
            keep = self.status >= 2
            status = self.status
            self.configure()
            if keep:
                self.logger.warning('transferring weights from previous model with only %d character types', old_voc)
                new = self.get_weights()
                for name in new:
                    if name == 'E':
                        new['E'][:old_voc] = old['E']
                    else:
                        new[name] = old[name]
                self.set_weights(new)
                self.status = status

    def _config_dict(self):
        return OrderedDict([
            ('width', np.array(self.width)), ('depth', np.array(self.depth)), ('stateful', np.array(bool(self.stateful))),
            ('residual_connections', np.array(bool(self.residual_connections))),
            ('deep_bidirectional_encoder', np.array(bool(self.deep_bidirectional_encoder))),
            ('bridge_dense', np.array(bool(self.bridge_dense))),
            ('mapping', np.fromiter((ord(self.mapping[1][i]) if i in self.mapping[1] and self.mapping[1][i] else 0
                                     for i in range(self.voc_size)), dtype=np.uint32))])

    def save(self, filename):
        """Store weights + configuration (seq2seq.py:1121-1141) in the reference's container: the Keras
        `save_weights` HDF5 layout plus the `config` group (keras_h5.py; written without h5py).
        A filename ending in `.npz` selects a plain numpy archive of the same tensors instead."""
        assert self.status > 1
        self.logger.info('Saving model under "%s"', filename)
        if str(filename).endswith('.npz'):
            data = {k: v for k, v in self._weights.items()}
            for key, value in self._config_dict().items():
                data['config/' + key] = value
            with open(filename, 'wb') as f:
                np.savez(f, **data)
        else:
            keras_h5.write_model(filename, self._config_dict(), self._weights)

    def _read_container(self, filename):
        """(config dict, {keras layer name: {tensor name: array}}) from a Keras HDF5 or an .npz model file."""
        if hdf5.is_hdf5(filename):
            return keras_h5.read_model(filename, self.logger)
        with np.load(filename) as data:
            src = {k: data[k] for k in data.files}
        config = {k[len('config/'):]: v for k, v in src.items() if k.startswith('config/')}
        depth = int(config['depth']) if 'depth' in config else self.depth
        bridge = bool(np.asarray(config['bridge_dense']).item()) if 'bridge_dense' in config else self.bridge_dense
        deep = bool(np.asarray(config['deep_bidirectional_encoder']).item()) if 'deep_bidirectional_encoder' in config else self.deep_bidirectional_encoder
        layers = OrderedDict()
        for lname, tensors in keras_h5.layer_tensors(depth, bridge, deep).items():
            if all(t in src for t in tensors):
                layers[lname] = OrderedDict((t, src[t]) for t in tensors)
        return config, layers

    def _set_mapping_from_codes(self, codes):
        c_i = dict((chr(c), i) if c > 0 else ('', 0) for i, c in enumerate(codes))
        i_c = dict((i, chr(c)) if c > 0 else (0, '') for i, c in enumerate(codes))
        self.mapping = (c_i, i_c)
        self.voc_size = len(c_i)

    def load_config(self, filename):
        """seq2seq.py:1143-1162."""
        if hdf5.is_hdf5(filename):
            config = keras_h5.read_config(filename)
        else:
            config, _ = self._read_container(filename)
        if 'width' not in config or 'mapping' not in config:
            raise KeyError('model file "%s" has no config group' % filename)
        self.width = int(config['width'])
        self.depth = int(config['depth'])
        self.stateful = bool(config['stateful'])
        self.residual_connections = bool(config.get('residual_connections', False))           # old default
        self.deep_bidirectional_encoder = bool(config.get('deep_bidirectional_encoder', False))
        self.bridge_dense = bool(config.get('bridge_dense', False))
        self._set_mapping_from_codes(config['mapping'])

    def _assign_layers(self, layers, table, skip_mismatch):
        """Keras' by-name weight loading: a layer of the file goes into the layer of the same name if the number
        and the shapes of its weights agree (ValueError, or a warning with skip_mismatch); layers the file does
        not have keep their values."""
        shapes = weight_shapes(self.depth, self.width, self.voc_size, self.bridge_dense, self.deep_bidirectional_encoder)
        w = self.get_weights()
        taken = []
        for lname, tensors in table.items():
            if lname not in layers:
                continue
            src = list(layers[lname].values())
            problem = None
            if len(src) != len(tensors):
                problem = 'layer "%s" expects %d weight(s), but the saved weights have %d element(s)' % (
                    lname, len(tensors), len(src))
            else:
                for name, arr in zip(tensors, src):
                    if int(np.prod(arr.shape)) != int(np.prod(shapes[name])) or (
                            arr.ndim == len(shapes[name]) and tuple(arr.shape) != tuple(shapes[name])):
                        problem = 'layer "%s": weight %s has shape %s, but the saved weight has shape %s' % (
                            lname, name, shapes[name], arr.shape)
                        break
            if problem:
                if not skip_mismatch:
                    raise ValueError(problem)
                self.logger.warning('skipping loading of weights for %s', problem)
                continue
            for name, arr in zip(tensors, src):
                w[name] = np.asarray(arr, np.float32).reshape(shapes[name])
            taken.append(lname)
        self.set_weights(w)
        return taken

    def load_weights(self, filename):
        """seq2seq.py:1164-1174 (`load_weights(filename, by_name=True)` + decoder resync)."""
        assert self.status > 0
        self.logger.info('Loading model from "%s"', filename)
        _, layers = self._read_container(filename)
        table = keras_h5.layer_tensors(self.depth, self.bridge_dense, self.deep_bidirectional_encoder)
        taken = self._assign_layers(layers, table, skip_mismatch=False)
        for lname in table:
            if lname not in taken:
                self.logger.warning('model file has no weights for layer "%s"', lname)
        self.status = 2

    def load_transfer_weights(self, filename):
        """Initialise matching layers from another (possibly shallower) model (seq2seq.py:1176-1213)."""
        assert self.status > 0
        assert self.depth > 1
        config, layers = self._read_container(filename)
        was_shallow = False
        if 'mapping' in config:
            self._set_mapping_from_codes(config['mapping'])
            self._reconfigure_for_mapping()
            was_shallow = 'depth' in config and int(config['depth']) == self.depth - 1
        self.logger.info('Transferring model from "%s"', filename)
        table = keras_h5.layer_tensors(self.depth, self.bridge_dense, self.deep_bidirectional_encoder)
        # the reference hands keras the attention CELL in place of the top decoder layer (seq2seq.py:1200-1204);
        # its name matches no layer of the file, so that layer is never transferred
        del table['decoder_lstm_%d' % self.depth]
        self._assign_layers(layers, table, skip_mismatch=True)
        self.frozen_prefixes = []
        if was_shallow:
            # layers taken over from a model one layer shallower stay fixed (seq2seq.py:1206-1211)
            self.logger.info('fixing weights from shallower model')
            self.frozen_prefixes = ['enc%d_' % i for i in range(1, self.depth)] + \
                                   ['dec%d_' % i for i in range(1, self.depth)]
        self.status = 1

    # ------------------------------------------------------------------------------------------
    # input layouts
    # ------------------------------------------------------------------------------------------
    def _index(self, char, what, i):
        idx = self.mapping[0].get(char)
        if idx is None:
            if char != GAP:
                self.logger.error('unmapped character "%s" at %s sequence %d', char, what, i)
            return 0   # underspecification
        return idx

    def vectorize_lines(self, encoder_input_sequences, decoder_input_sequences, encoder_conf_sequences=None):
        """Strings (or confidence lines / confusion networks) -> dense arrays, exactly the layouts of
        seq2seq.py:1020-1119: (enc (B,T,V), dec_in (B,Tt+1,V), dec_out (B,Tt+1,V), weights (B,Tt+1))."""
        assert len(encoder_input_sequences) == len(decoder_input_sequences)
        idx, val, conf = self._sparse_lines(encoder_input_sequences, encoder_conf_sequences)
        B, T, A = idx.shape
        enc = np.zeros((B, T, self.voc_size), dtype=np.float32 if conf else np.uint32)
        b, t, a = np.nonzero(idx >= 0)
        enc[b, t, idx[b, t, a]] = val[b, t, a]       # later alternatives overwrite, as the reference's loop does
        Tt = max(map(len, decoder_input_sequences))
        dec_in = np.zeros((B, Tt + 1, self.voc_size), dtype=np.uint32)
        dec_out = np.zeros((B, Tt + 1, self.voc_size), dtype=np.uint32)
        for i, seq in enumerate(decoder_input_sequences):
            for j, char in enumerate(seq):
                k = self._index(char, 'decoder input', i)
                dec_in[i, j + 1, k] = 1
                dec_out[i, j, k] = 1
        weights = np.ones(dec_out.shape[:-1], dtype=np.float32)
        weights[np.all(dec_out == 0, axis=2)] = 0.
        return enc, dec_in, dec_out, weights

    def _sparse_lines(self, lines, conf=None):
        """The same three input forms as index/value tensors (B,T,A), -1 = empty slot: this is what the
        device consumes (the dense one-hot array of the reference is 4*V bytes per character)."""
        B = len(lines)
        with_confmat = bool(conf) and type(conf[0][0]) is list
        if with_confmat:
            seqs = conf
            T = max(sum(max(len(x[0]) for x in chunk) if chunk else 0 for chunk in seq) for seq in seqs)
            A = max([len(chunk) for seq in seqs for chunk in seq] + [1])
        else:
            T = max(map(len, lines))
            A = 1
        idx = np.full((B, T, A), -1, np.int32)
        val = np.zeros((B, T, A), np.float32)
        if with_confmat:
            for i, seq in enumerate(seqs):
                j = 0
                for chunk in seq:
                    width = max(len(x[0]) for x in chunk) if chunk else 0
                    for a, (chars, p) in enumerate(chunk):
                        for k, char in enumerate(chars):
                            ci = self._index(char, 'encoder input', i)
                            # a later alternative with the same index replaces the earlier one, as the
                            # dense assignment of seq2seq.py:1079 does
                            same = np.nonzero(idx[i, j + k, :a] == ci)[0]
                            slot = int(same[0]) if len(same) else a
                            idx[i, j + k, slot] = ci
                            val[i, j + k, slot] = p
                    j += width
        else:
            # plain strings: ONE table lookup over all code points of the batch
            lut = self._codepoint_lut()
            lens = np.fromiter((len(line) for line in lines), dtype=np.int64, count=B)
            total = int(lens.sum())
            if total:
                cps = np.frombuffer(''.join(lines).encode('utf-32-le', 'surrogatepass'), dtype=np.uint32)
                found = lut[np.minimum(cps, len(lut) - 1)]
                hit = found >= 0
                rows = np.repeat(np.arange(B), lens)
                cols = np.arange(total) - np.repeat(np.cumsum(lens) - lens, lens)
                if not hit.all():
                    for j in np.nonzero(~hit)[0]:
                        self._index(lines[int(rows[j])][int(cols[j])], 'encoder input', int(rows[j]))   # logs like the reference
                idx[rows, cols, 0] = np.where(hit, found, 0)
                if conf:
                    val[rows, cols, 0] = np.concatenate([np.asarray(c, np.float32) for c in conf if len(c)])
                else:
                    val[rows, cols, 0] = 1.0
        return idx, val, conf

    @staticmethod
    def _dense_to_sparse(data):
        """(B,T,V) rows -> (idx, val) with A = the largest number of non-zeros of any row."""
        data = np.asarray(data)
        nz = data != 0
        A = max(1, int(nz.sum(axis=2).max()) if data.size else 1)
        order = np.argsort(~nz, axis=2, kind='stable')[:, :, :A]
        picked = np.take_along_axis(nz, order, axis=2)
        idx = np.where(picked, order, -1).astype(np.int32)
        val = np.where(picked, np.take_along_axis(data, order, axis=2), 0).astype(np.float32)
        return idx, val

    # ------------------------------------------------------------------------------------------
    # decoding
    # ------------------------------------------------------------------------------------------
    def _require_engine(self):
        """The device model, created on first use and re-synchronised after weight changes
        (the reference's `_resync_decoder`, seq2seq.py:526-528).  Raises if the HIP library or a
        GPU is missing: there is no CPU fallback."""
        assert self.status >= 1, 'configure() first'
        if self.voc_size < 2:
            raise RuntimeError('model has no vocabulary yet (load or train a model first)')
        if self.engine is None:
            self.engine = HipEngine(self.depth, self.width, self.voc_size, device=self.device,
                                    residual_connections=self.residual_connections, bridge_dense=self.bridge_dense,
                                    deep_bidirectional_encoder=self.deep_bidirectional_encoder)
            self._dirty = True
        if self._dirty:
            self.engine.set_weights(self._weights)
            self._dirty = False
        self._eos = self.mapping[0].get('\n', 1)
        self.engine.set_option('eos', self._eos)
        self.engine.set_option('arithmetic', {'auto': -1, 'fp32': 0, 'split': 2}[self.arithmetic])
        return self.engine

    def _codepoint_table(self):
        """Sorted code points of the single-character vocabulary entries and their indices."""
        mapping = self.mapping[0]
        cached = getattr(self, '_cp_cache', None)
        if cached is None or cached[0] is not mapping or cached[1] != len(mapping):
            items = sorted((ord(c), i) for c, i in mapping.items() if len(c) == 1)
            keys = np.array([k for k, _ in items] or [0], np.uint32)
            values = np.array([v for _, v in items] or [0], np.int32)
            out = np.zeros(max(self.voc_size, 1), np.uint32)
            for i, c in self.mapping[1].items():
                if len(c) == 1 and i < len(out):
                    out[i] = ord(c)
            lut = np.full(int(keys.max()) + 2, -1, np.int32)       # code point -> index, -1 = unmapped (last slot: beyond)
            lut[keys] = values
            self._cp_cache = (mapping, len(mapping), keys, values, out, lut)
            cached = self._cp_cache
        return cached[2], cached[3]

    def _codepoint_lut(self):
        """Dense code point -> vocabulary index table (-1 = unmapped); the last slot stands for all larger code points."""
        self._codepoint_table()
        return self._cp_cache[5]

    def _chars(self, indexes):
        self._codepoint_table()
        cps = self._cp_cache[4][np.asarray(indexes, np.int64)]
        return cps[cps != 0].astype('<u4').tobytes().decode('utf-32-le', 'surrogatepass')

    def _texts(self, idx, n):
        """Strings of all rows at once: row j = the characters of idx[j, :n[j]] (index 0 maps to no character).  ONE table
        lookup and ONE utf-32 decode for the whole batch, then slices."""
        B, S = idx.shape
        n = np.asarray(n, np.int64)
        keep = np.arange(S)[None, :] < n[:, None]
        self._codepoint_table()
        cps = self._cp_cache[4][idx[keep]]
        present = cps != 0
        text = cps[present].astype('<u4').tobytes().decode('utf-32-le', 'surrogatepass')
        upto = np.concatenate([[0], np.cumsum(present)])
        stop = np.cumsum(n)
        ends = np.cumsum(upto[stop] - upto[stop - n]).tolist()
        out, start = [], 0
        for end in ends:
            out.append(text[start:end])
            start = end
        return out, keep

    def _greedy_results(self, idx, prob, align, nonpad, T=None):
        """Per-line bookkeeping of seq2seq.py:1254-1263 on the index/probability matrices, for all lines at once: length up
        to the first end-of-line, string, mean -log p."""
        B, S = idx.shape
        is_eos = idx == self._eos
        n = np.where(is_eos.any(axis=1), is_eos.argmax(axis=1) + 1, S)
        n = np.where(nonpad, n, 0)
        texts, keep = self._texts(idx, n)
        with np.errstate(divide='ignore', invalid='ignore'):
            cost = np.where(keep, -np.log(prob), 0).sum(axis=1, dtype=np.float64) / np.maximum(n, 1)
        # probability lists: ONE conversion for the whole batch, then list slices
        flat = prob[keep].tolist()
        ends = np.cumsum(n).tolist()
        costs = cost.tolist()
        lines, probs, scores, aligns = [], [], [], []
        start = 0
        for j in range(B):
            end = ends[j]
            if not nonpad[j]:
                lines.append(''); probs.append([]); scores.append(0.); aligns.append([])
            else:
                lines.append(texts[j])
                probs.append(flat[start:end])
                scores.append(costs[j])
                aligns.append(self._alignment_rows(align, j, end - start, T))
            start = end
        return lines, probs, scores, aligns

    def _alignment_rows(self, align, j, n, T=None):
        """Alignment of result row j, first n steps: a SparseAlignment view over the window form, the reference's list
        of T-wide rows for a dense array, [] when alignments were not requested.  T = the padded length of the batch the
        row was decoded in (default: the engine's last batch)."""
        if align is None:
            return []
        if isinstance(align, tuple):
            lo, w = align
            return SparseAlignment(lo[j, :n], w[j, :n], self.engine.T if T is None else T)
        return [align[j, k] for k in range(n)]

    def decode_batch_greedy(self, encoder_input_data):
        """seq2seq.py:1215-1286: all lines at once, 2T steps, argmax without index 0, soft feedback.
        Returns (decoder_output_data, strings, probability lists, scores, alignments)."""
        eng = self._require_engine()
        encoder_input_data = np.asarray(encoder_input_data)
        idx, val = self._dense_to_sparse(encoder_input_data)
        eng.encode(idx, val)
        gi, gp, _, ga = eng.decode_greedy(mode=0, want_align=True)
        nonpad = ((idx >= 0) & (val != 0)).any(axis=(1, 2))     # np.any(encoder_input_data[j]), seq2seq.py:1255
        lines, probs, scores, aligns = self._greedy_results(gi, gp, ga, nonpad)
        B, T = encoder_input_data.shape[:2]
        # the reference stores the fed-back softmax in a uint32 array (seq2seq.py:1237,1244): all zeros
        decoder_output_data = np.zeros((B, 2 * T, self.voc_size), dtype=np.uint32)
        return decoder_output_data, lines, probs, scores, aligns

    def _encode_or_install(self, eng, source_seq, encoder_outputs):
        """Run the encoder on `source_seq` (T,V), or install `encoder_outputs` = what `encoder_model.predict_on_batch`
        returns for one line ([enc_out (1,T,C), h1, c1, ..., hd, cd, a0 (1,T)], seq2seq.py:1305-1308,1382-1386)."""
        if encoder_outputs is None:
            if source_seq is None:
                raise ValueError('need source_seq or encoder_outputs')
            idx, val = self._dense_to_sparse(np.asarray(source_seq)[None])
            eng.encode(idx, val)
            return
        outs = list(encoder_outputs)
        src_rej = None
        if source_seq is not None:      # the beam reads the source line back for its rejection candidates (seq2seq.py:1458)
            src = np.asarray(source_seq)
            src_rej = np.where(src.any(axis=1), src.argmax(axis=1), -1).astype(np.int32)[None]
        eng.set_encoder_outputs(outs[0], outs[1:1 + 2 * self.depth], a0=outs[1 + 2 * self.depth] if len(outs) > 1 + 2 * self.depth else None,
                                src_rej=src_rej)

    @property
    def encoder_model(self):
        """Stand-in for the Keras `encoder_model` (seq2seq.py:403-406): `.predict_on_batch(x)` -> [enc_out, h1, c1, ..., a0]."""
        return _EncoderModel(self)

    @property
    def decoder_model(self):
        """Stand-in for the Keras `decoder_model` (seq2seq.py:470-473): `.predict_on_batch([p, enc_out] + states)` ->
        [scores (R,1,V)] + new states."""
        return _DecoderModel(self)

    def decode_sequence_greedy(self, source_seq=None, encoder_outputs=None):
        """seq2seq.py:1288-1354 for one line given as a (T,V) array (or as the encoder's outputs for it)."""
        eng = self._require_engine()
        self._encode_or_install(eng, source_seq, encoder_outputs)
        return self._sequence_greedy_results(eng, 1)[0]

    def _sequence_greedy_results(self, eng, B, want_align=True):
        try:
            gi, gp, gl, ga = eng.decode_greedy(mode=1, want_align=want_align)
        except nv.NativeError as err:
            if err.code == nv.CASV_ERR_NAN:
                raise ValueError('All-NaN slice encountered')   # what np.nanargmax raises, seq2seq.py:1335
            raise
        out = []
        for j in range(B):
            n = int(gl[j])
            p = gp[j, :n]
            with np.errstate(divide='ignore', invalid='ignore'):
                score = float(np.sum(-np.log(p), dtype=np.float64)) / n
            out.append((self._chars(gi[j, :n]), list(p), score, self._alignment_rows(ga, j, n)))
        return out

    def _beam_results(self, res, j, max_results, T):
        for k in range(max_results):
            r = j * max_results + k
            n = int(res['len'][r])
            if n == 0:
                return
            aligns = self._alignment_rows(res.get('align_sparse', res['align']), r, n)
            yield (self._chars(res['idx'][r, :n]), res['prob'][r, :n].tolist(), float(res['score'][r]), aligns)

    def _beam_kwargs(self):
        return dict(batch_size=self.batch_size, beam_width_in=self.beam_width_in,
                    beam_threshold_in=self.beam_threshold_in, beam_width_out=self.beam_width_out,
                    rejection_threshold=self.rejection_threshold)

    def decode_sequence_beam(self, source_seq=None, encoder_outputs=None):
        """seq2seq.py:1356-1544: generator of (string, probabilities, score, alignments), best first.
        The search itself runs on the device when the first result is requested."""
        eng = self._require_engine()
        self._encode_or_install(eng, source_seq, encoder_outputs)
        res = eng.decode_beam(max_results=64, want_align=True, **self._beam_kwargs())
        for item in self._beam_results(res, 0, 64, eng.T):
            yield item

    def correct_lines(self, lines, conf=None, fast=True, greedy=True, alignments=True):
        """seq2seq.py:782-842.  Each line must end in a newline.  Returns (lines, probability lists,
        scores, alignments).  The alignment of a line is a `SparseAlignment`: a list of T-wide rows (`alignment[j][i]`, `len`,
        iteration, `numpy.asarray`) over the window form the device returns -- 12 instead of T floats per character cross
        PCIe, the rows are built when first looked at, and `realign.alignment2path` consumes the windows directly.
        Extensions: `alignments=False` skips the soft alignments altogether, `alignments='dense'` returns the reference's
        lists of T-wide numpy rows."""
        assert not fast or greedy, "cannot decode in fast mode with beam search enabled"
        if not lines:
            return [], [], [], []
        prepared = self._prepare_lines(lines, conf)
        raw = self._decode_prepared(prepared, fast, greedy, alignments, [bool(line) for line in lines])
        return self._results_of(lines, prepared, raw, fast, greedy, alignments)

    def correct_batches(self, batches, fast=True, greedy=True, alignments=True, after_decode=None):
        """`correct_lines` over a stream of batches -- an iterable of `lines` or of `(lines, conf)` -- as a three-stage pipeline:
        a worker thread turns the next batch into index arrays, a second one drives the device (its C-ABI calls release the GIL),
        and the caller's thread builds strings and lists from the batch before -- so the device does not wait for the Python on
        either side of it (the reference decodes batch after batch, seq2seq.py:756-780).  Yields what `correct_lines` returns,
        batch by batch, in order; identical results.  `after_decode(k)`, if given, runs in the device thread right after batch
        k's decode call (its results still lie in the engine's buffers: e.g. `engine.records_append`)."""
        assert not fast or greedy, "cannot decode in fast mode with beam search enabled"
        from .training import prefetch
        self._require_engine()
        self._codepoint_table()                 # (the lookup tables exist before the stages' threads ask for them)

        def prepared():
            for item in batches:
                # a pair (lines, conf) is a 2-tuple whose first member is itself a sequence of lines -- a batch handed over as a
                # tuple of strings is a batch of lines
                pair = isinstance(item, tuple) and len(item) == 2 and isinstance(item[0], (list, tuple))
                lines, conf = item if pair else (item, None)
                yield lines, (self._prepare_lines(lines, conf) if lines else None)

        import threading
        closing, in_call = threading.Event(), threading.Event()

        def decoded():
            for k, (lines, prep) in enumerate(prefetch(prepared(), depth=2, cancel=closing, detach_after=5.0)):
                if closing.is_set():            # (the consumer has left: no further call on the engine)
                    return
                in_call.set()
                try:
                    raw = self._decode_prepared(prep, fast, greedy, alignments, [bool(line) for line in lines]) if lines else None
                    # (no decode ran -- no lines, nothing but padding, or no live line in the per-line greedy mode --: the engine
                    # still holds the batch before)
                    ran = raw is not None and (fast or bool(raw[0]))
                    if after_decode is not None and ran:
                        after_decode(k)
                finally:
                    in_call.clear()
                yield lines, prep, raw

        # The device stage needs the interpreter lock only between its C-ABI calls; the string building of the stage behind it
        # is pure Python and gives the lock up once per switch interval (5 ms by default: up to half a millisecond of idle GPU
        # between two batches of configs[1], rocprofv3 trace of round 4).  A short interval while THIS generator runs -- not
        # while it is suspended at a yield or abandoned: the setting is process-wide (_ShortSwitchInterval counts its users).
        stage = prefetch(decoded(), depth=1, in_call=in_call, detach_after=5.0)
        try:
            while True:
                with _ShortSwitchInterval():
                    try:
                        lines, prep, raw = next(stage)
                    except StopIteration:
                        return
                    result = self._results_of(lines, prep, raw, fast, greedy, alignments) if lines else ([], [], [], [])
                yield result
        finally:
            closing.set()
            stage.close()

    # the three stages of correct_lines ------------------------------------------------------------
    def _prepare_lines(self, lines, conf):
        """Host, before the device: strings / confidences -> index and value arrays, and the rejection candidate of every
        position (batch-wide numpy arithmetic that would otherwise run in the device stage, between two C-ABI calls)."""
        idx, val, _ = self._sparse_lines(lines, conf)
        from .engine import HipEngine
        rej = HipEngine.source_rejection(idx, val) if idx.ndim == 3 and idx.shape[1] else None
        return idx, val, rej

    def _decode_prepared(self, prepared, fast, greedy, alignments, nonempty=None):
        """The device part: encode + decode, raw result arrays.  (The only stage that talks to the engine.)
        `nonempty[j]`: line j of the batch is not empty (decided on the LINES, as seq2seq.py:815 does: a line of nothing
        but unmapped symbols is decoded, an empty padding line is not)."""
        eng = self._require_engine()
        want_align = False if not alignments else (True if alignments == 'dense' else 'sparse')
        idx, val = prepared[:2]
        rej = prepared[2] if len(prepared) > 2 else None
        B, T = idx.shape[:2]
        if T == 0:                  # nothing but padding lines
            return None
        if fast:
            eng.encode(idx, val, rej)
            gi, gp, _, ga = eng.decode_greedy(mode=0, want_align=want_align)
            return gi, gp, ga
        # the per-line modes never decode the empty padding lines of a partial batch (seq2seq.py:815-816) -- an
        # all-zero input row would also trip the greedy mode's NaN rule for the whole batch
        live = [j for j in range(B) if (nonempty[j] if nonempty is not None else (idx[j] >= 0).any())]
        if greedy:
            if not live:
                return live, []
            eng.encode(idx[live], val[live], None if rej is None else rej[live])
            return live, self._sequence_greedy_results(eng, len(live), want_align)
        # The search keeps every expansion's state on the device (nothing is recomputed, nothing crosses to the host):
        # S x (lines x N) rows of h, c per layer, scores and alignments.  Large beams (the reference's default
        # batch_size = 256 hypotheses per step) are therefore decoded in chunks of lines that fit a memory budget;
        # lines are independent, so chunking does not change any result.
        # plus the trie: up to min(beam_width_in, V) + 1 child records of 60 bytes per expansion
        children = min(self.beam_width_in, self.voc_size) + 1
        per_line = 2 * T * self.batch_size * (((2 * self.depth + 1) * self.width + self.voc_size + 32 + T) * 4 + 60 * children)
        budget = float(os.environ.get('CASV_BEAM_MEMORY_GB', '96')) * 2 ** 30
        chunk = int(max(1, min(B, budget // max(per_line, 1))))
        out = []
        for lo in range(0, len(live), chunk):
            rows = live[lo:lo + chunk]
            eng.encode(idx[rows], val[rows], None if rej is None else rej[rows])
            res = eng.decode_beam(max_results=1, want_align=want_align, **self._beam_kwargs())
            out.append((rows, res, eng.T))
        return live, out

    def _results_of(self, lines, prepared, raw, fast, greedy, alignments):
        """Host, after the device: raw arrays -> strings, probability lists, scores, alignment views."""
        idx, val = prepared[:2]
        B = idx.shape[0]
        if raw is None:
            return self._finish(lines, [('', [], 0, []) for _ in range(B)])
        if fast:
            gi, gp, ga = raw
            nonpad = ((idx >= 0) & (val != 0)).any(axis=(1, 2))     # np.any(encoder_input_data[j]), seq2seq.py:1255
            return self._greedy_results(gi, gp, ga, nonpad, idx.shape[1])
        live, out = raw
        results = [('', [], 0, []) for _ in range(B)]
        if greedy:
            for j, r in zip(live, out):
                results[j] = r
            return self._finish(lines, results)
        for rows, res, T in out:
            texts, keep = self._texts(res['idx'], res['len'])       # best result of every line of the chunk, in one go
            flat = res['prob'][keep].tolist()                       # likewise the probability lists: one conversion, then slices
            ends = np.cumsum(res['len']).tolist()
            lens, scores_ = res['len'].tolist(), res['score'].tolist()
            for k, j in enumerate(rows):
                input_line = lines[j]
                n = lens[k]
                item = None
                if n:
                    item = (texts[k], flat[ends[k] - n:ends[k]], scores_[k],
                            self._alignment_rows(res.get('align_sparse', res['align']), k, n, T))
                if item is None:
                    # the generator of the reference raises StopIteration here (seq2seq.py:826-836)
                    self.logger.error('cannot beam-decode input line %d: "%s"', j, input_line)
                    if isinstance(input_line[0], tuple):
                        line = ''.join(chunk_[0] for chunk_ in input_line)
                    if isinstance(input_line[0], list):
                        line = ''.join(chunk_[0][0] if chunk_ else '' for chunk_ in input_line)
                    else:
                        line = input_line
                    item = (line, [1.0] * len(line), 0, [] if not alignments else
                            (self._identity_alignment(len(line)) if alignments == 'dense' else SparseAlignment.identity(len(line))))
                results[j] = item
        return self._finish(lines, results)

    def _identity_alignment(self, n):
        """`np.eye(n).tolist()` of the beam fallback (seq2seq.py:834).  The rows are built once per length and
        shared between the lines of that length (a fresh outer list per line): read-only in every caller."""
        cache = self.__dict__.setdefault('_eye_cache', {})
        rows = cache.get(n)
        if rows is None:
            rows = cache[n] = np.eye(n).tolist()
        return list(rows)

    @staticmethod
    def _finish(lines, results):
        out = ([], [], [], [])
        for line, probs, score, alignment in results:
            out[0].append(line.replace(GAP, ''))   # seq2seq.py:837
            out[1].append(probs)
            out[2].append(score)
            out[3].append(alignment)
        return out

    # ------------------------------------------------------------------------------------------
    # file level
    # ------------------------------------------------------------------------------------------
    def predict(self, filenames, fast=False, greedy=False, charmap=None):
        """seq2seq.py:756-780: generator of (filenames, lines, scores) per batch."""
        assert self.status == 2
        import collections
        names = collections.deque()            # file names of the batches in flight (the producer runs a few batches ahead)

        def batches():
            for lines_source, lines_sourceconf, _, lines_filename in self.gen_lines(filenames, repeat=False, unsupervised=True, charmap=charmap):
                names.append(lines_filename)
                yield lines_source, lines_sourceconf
        # batch k + 1 is read and vectorised, and the device runs on it, while this generator's consumer handles batch k
        for lines_result, _, scores_result, _ in self.correct_batches(batches(), fast=fast, greedy=greedy, alignments=False):
            yield (names.popleft(), lines_result, scores_result)

    def map_files(self, filenames):
        """Collect the character set of the files and grow the mapping (seq2seq.py:555-588)."""
        num_lines = 0
        chars = set(self.mapping[0].keys())
        for filename in filenames:
            for source_text, source_conf, target_text in self._read_file(filename, unsupervised=False, keep_raw=True):
                text = unicodedata.normalize('NFC', source_text + target_text)
                chars.update(text)
                if GAP in chars:
                    self.logger.warning('ignoring gap character "%s" in input file "%s"', GAP, filename)
                    chars.remove(GAP)
                num_lines += 1
        chars = sorted(chars)
        if len(chars) > self.voc_size:
            self.mapping = ({c: i for i, c in enumerate(chars)}, {i: c for i, c in enumerate(chars)})
            self.voc_size = len(chars)
            self._reconfigure_for_mapping()
        return num_lines

    @staticmethod
    def _read_file(filename, unsupervised, keep_raw=False):
        """Yield (source_text, source_conf or None, target_text) per line of a TSV or pickle file
        (formats of seq2seq.py:936-972)."""
        if filename.endswith('.pkl'):
            with open(filename, 'rb') as f:
                records = pickle.load(f)
            for source, target_text in records:
                if not source:
                    source_text, source_conf = '', []
                elif type(source[0]) is tuple:       # prob line
                    chars, probs = zip(*source)
                    source_text, source_conf = ''.join(chars), list(probs)
                else:                                # confmat
                    source_conf = source
                    if keep_raw:
                        source_text = ''.join(c for chunk in source for c, _ in chunk)
                    else:
                        source_text = ''.join(chunk[0][0] if chunk else '' for chunk in source)
                if not keep_raw and not source_text.endswith('\n'):
                    source_conf, source_text = [[('\n', 1.0)]], '\n'
                yield source_text, source_conf, target_text
        else:
            with open(filename, 'r') as f:
                for line in f:
                    if keep_raw:
                        yield line, None, ''
                    elif unsupervised and '\t' not in line:
                        yield line, None, line
                    else:
                        source_text, target_text = line.split('\t')
                        yield source_text + '\n', None, target_text

    def gen_data(self, filenames, split=None, train=False, unsupervised=False, charmap=None, reset_cb=None):
        """Dense batches ([encoder_input, decoder_input], decoder_output, weights) with False after every pass
        over the files (seq2seq.py:846-917), for callers that want the Keras-style arrays.  `train()` itself
        feeds the device with the index form of the same batches (training.batch_to_indices)."""
        for batch in self.gen_lines(filenames, True, split, train, unsupervised, charmap):
            if not batch:
                yield False
                continue
            lines_source, lines_sourceconf, lines_target, _ = batch
            enc, dec_in, dec_out, weights = self.vectorize_lines(lines_source, lines_target, lines_sourceconf)
            if train:
                rand = (enc.shape[1] * np.random.uniform(0, 1, len(lines_source)) / 0.01).astype(int)
                rows = np.nonzero(rand < enc.shape[1])[0]
                enc[rows, rand[rows], :] = np.eye(self.voc_size, dtype=enc.dtype)[0]
            yield ([enc, dec_in], dec_out, weights)

    def gen_lines(self, filenames, repeat=True, split=None, train=False, unsupervised=False, charmap=None):
        """Batches of (source lines, source confidences or None, target lines, filenames)
        (seq2seq.py:919-1018).  `repeat` loops over the files yielding False after every pass;
        otherwise the last partial batch is padded with '' / None."""
        split_ratio = 0.2
        table = str.maketrans(charmap) if charmap else None
        epoch = 0
        while True:
            src, cnf, tgt, names = [], [], [], []
            with_confidence = False
            for filename in filenames:
                with_confidence = filename.endswith('.pkl')
                for line_no, (s, c, t) in enumerate(self._read_file(filename, unsupervised)):
                    if isinstance(split, np.ndarray) and (split[line_no] < split_ratio) == train:
                        continue      # this line belongs to the other (train/validation) generator
                    if unsupervised:
                        t = s
                    if table:
                        s, t = s.translate(table), t.translate(table)
                    s = unicodedata.normalize('NFC', s)
                    t = unicodedata.normalize('NFC', t)
                    if train and self._is_bad_pair(s, t):
                        if epoch == 0:
                            self.logger.debug('ignoring bad line "%s\t%s"', s.rstrip(), t.rstrip())
                        continue
                    src.append(s); tgt.append(t); names.append(filename)
                    if with_confidence:
                        cnf.append(c)
                    if len(src) == self.batch_size:
                        yield (src, cnf if with_confidence else None, tgt, names)
                        src, cnf, tgt, names = [], [], [], []
            epoch += 1
            if repeat:
                yield False
            else:
                if src:
                    pad = self.batch_size - len(src)
                    src.extend(pad * ['']); tgt.extend(pad * [''])
                    if with_confidence:
                        cnf.extend(pad * [[]])
                    names.extend(pad * [None])
                    yield (src, cnf if with_confidence else None, tgt, names)
                break

    @staticmethod
    def _is_bad_pair(source_text, target_text):
        """Training filter for hopeless OCR lines: the criterion of `Alignment.is_bad()` (lib/alignment.py:160-163) --
        difflib's quick similarity bound below one half on a source line of more than 5 characters."""
        import difflib
        matcher = difflib.SequenceMatcher(isjunk=None, a=source_text, b=target_text, autojunk=False)
        return bool(matcher.quick_ratio() < 0.5 and len(source_text) > 5)

    def evaluate(self, filenames, fast=False, normalization='historic_latin', charmap=None, gt_level=1,
                 confusion=10, histogram=True):
        """Character/word error rates of source (OCR), greedy and beamed output against the target (seq2seq.py:651-754).

        Decoding runs on the device; alignment, normalisation and the running statistics are this package's own
        (`metrics.py`, restating lib/alignment.py:140-486) -- nothing is imported from the reference."""
        assert self.status == 2
        from .metrics import Alignment, Edits, splitwords
        names = ('origin', 'greedy', 'beamed')
        c_counts = {k: Edits(self.logger, histogram=histogram) for k in names}
        w_counts = {k: Edits(self.logger) for k in names}
        c_aligner = {k: Alignment(0, logger=self.logger, confusion=confusion > 0) for k in names}
        w_aligner = {k: Alignment(0, logger=self.logger) for k in names}

        def get_counts(aligner, line_source, line_target):
            dist, length = aligner.get_adjusted_distance(line_source, line_target, normalization=normalization, gtlevel=gt_level)
            return dist, length, line_source, line_target

        for batch in self.gen_lines(filenames, False, charmap=charmap):
            lines_source, lines_sourceconf, lines_target, _ = batch
            lines_greedy, _, scores_greedy, _ = self.correct_lines(lines_source, lines_sourceconf, fast=fast, greedy=True)
            if fast:
                lines_beamed, scores_beamed = lines_greedy, scores_greedy
            else:
                lines_beamed, _, scores_beamed, _ = self.correct_lines(lines_source, lines_sourceconf, fast=False, greedy=False)
            for j in range(len(lines_source)):
                if not lines_source[j] or not lines_target[j]:
                    continue                    # from partially filled batch
                self.logger.info('Source input              : %s', lines_source[j].rstrip(u'\n'))
                self.logger.info('Target output             : %s', lines_target[j].rstrip(u'\n'))
                self.logger.info('Target prediction (greedy): %s [%.2f]', lines_greedy[j].rstrip(u'\n'), scores_greedy[j])
                self.logger.info('Target prediction (beamed): %s [%.2f]', lines_beamed[j].rstrip(u'\n'), scores_beamed[j])
                lines = {'origin': lines_source[j], 'greedy': lines_greedy[j], 'beamed': lines_beamed[j]}
                tokens_target = splitwords(lines_target[j])
                for k in names:
                    c_counts[k].add(*get_counts(c_aligner[k], lines[k], lines_target[j]))
                    w_counts[k].add(*get_counts(w_aligner[k], splitwords(lines[k]), tokens_target))
            c_counts['greedy'].score += sum(scores_greedy)
            c_counts['beamed'].score += sum(scores_beamed)

        self.logger.info('finished %d lines', c_counts['origin'].length)
        if normalization == 'historic_latin':        # (the two behaviours give different CER / WER on ligatures and PUA letters)
            from .metrics import normalization_mode
            self.logger.info('historic_latin normalisation: %s (cor_asv_ann_amd.metrics.reference_quirks)', normalization_mode())
        labels = {'origin': 'OCR   ', 'greedy': 'greedy', 'beamed': 'beamed'}
        if confusion > 0:
            for k in names:
                self.logger.info('%s confusion: %s', labels[k], c_aligner[k].get_confusion(confusion))
        if histogram:
            for k in names:
                self.logger.info('%s histogram: %s', labels[k], repr(c_counts[k].hist()))
        for k in ('greedy', 'beamed'):
            self.logger.info('ppl %s: %.3f', k, math.exp(c_counts[k].score / max(c_counts[k].length, 1)))
        for what, counts in (('CER', c_counts), ('WER', w_counts)):
            for k in names:
                self.logger.info('%s %-7s %.3f±%.3f', what, labels[k].strip() + ':', counts[k].mean, math.sqrt(counts[k].varia))

    def train(self, filenames, val_filenames=None):
        from .training import train_files
        return train_files(self, filenames, val_filenames)


class _EncoderModel(object):
    def __init__(self, s2s):
        self.s2s = s2s

    def predict_on_batch(self, x):
        s2s = self.s2s
        eng = s2s._require_engine()
        x = np.asarray(x)
        idx, val = s2s._dense_to_sparse(x)
        eng.encode(idx, val)
        enc, states = eng.encoder_outputs()
        return [enc] + states + [np.zeros(x.shape[:2], np.float32)]

    predict = predict_on_batch


class _DecoderModel(object):
    def __init__(self, s2s):
        self.s2s = s2s

    def predict_on_batch(self, inputs):
        s2s = self.s2s
        eng = s2s._require_engine()
        p_in, attended = np.asarray(inputs[0], np.float32), np.asarray(inputs[1], np.float32)
        states = [np.asarray(x, np.float32) for x in inputs[2:]]
        R = p_in.shape[0]
        d = s2s.depth
        # the attended sequence may have batch size 1 against R rows of state (the beam relies on it, seq2seq.py:1428-1429)
        Ba = attended.shape[0]
        zero = [np.zeros((Ba, s2s.width), np.float32)] * (2 * d)
        eng.set_encoder_outputs(attended, zero)
        line = np.arange(R, dtype=np.int32) if Ba == R else np.zeros(R, np.int32)
        probs, new = eng.decoder_step(line, p_in.reshape(R, -1), states[:2 * d], states[2 * d])
        return [probs[:, None, :]] + new

    predict = predict_on_batch


class Node(object):
    """Host-side view of one beam hypothesis (seq2seq.py:1546-1608).  The search itself keeps its
    trie on the device; this class exists for API compatibility of `lib/__init__.py:8`."""

    def __init__(self, state, value, scores, cost, parent=None, prob=1.0, alignment=None, length0=None, cost0=None):
        self.value = value
        self.parent = parent
        self.state = state
        self.cum_cost = parent.cum_cost + cost if parent else cost
        self.length = 1 if parent is None else parent.length + 1
        self.length0 = length0 or (parent.length0 if parent else 1)
        self.cost0 = cost0 or (parent.cost0 if parent else 0)
        self.prob = prob
        self.scores = scores
        self.alignment = (parent.alignment if parent else []) if alignment is None else alignment

    def to_sequence(self):
        seq, cur = [], self
        while cur:
            seq.insert(0, cur)
            cur = cur.parent
        return seq

    def __str__(self):
        return ''.join(n.value for n in self.to_sequence()[1:])

    def pro_cost(self):
        return - (self.cum_cost + self.cost0 * abs(self.length - self.length0))

    def __lt__(self, other): return self.pro_cost() < other.pro_cost()
    def __le__(self, other): return self.pro_cost() <= other.pro_cost()
    def __eq__(self, other): return self.pro_cost() == other.pro_cost()
    def __ne__(self, other): return self.pro_cost() != other.pro_cost()
    def __gt__(self, other): return self.pro_cost() > other.pro_cost()
    def __ge__(self, other): return self.pro_cost() >= other.pro_cost()
    __hash__ = object.__hash__
