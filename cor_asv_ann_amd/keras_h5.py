"""The reference's model container: Keras-2.3 `save_weights` HDF5 layout + the `config` group.

Layout (keras/engine/saving.py `save_weights_to_hdf5_group`, Keras 2.3.1, restated; the reference calls it
at seq2seq.py:1129 and adds the `config` group at seq2seq.py:1130-1141):

    /                         attrs: layer_names (fixed-length byte strings), backend, keras_version
    /<layer>/                 attrs: weight_names, in `layer.weights` order
    /<layer>/<weight name>    one float32 dataset per weight; the weight name contains '/', so it lives in a
                              nested group, e.g. /encoder_lstm_1/encoder_lstm_1/forward_encoder_lstm_1/kernel:0
    /config/{width,depth,stateful,residual_connections,deep_bidirectional_encoder,bridge_dense}  scalars
    /config/mapping           uint32[voc_size]: code point of every index, 0 = unmapped

Loading is by layer name and, inside a layer, by ORDER of `weight_names` (keras `load_weights(by_name=True)`,
seq2seq.py:1172): the names themselves only address the datasets.  Layers trained with CuDNNLSTM on a GPU
carry the cuDNN weight format (bias of 8W); `_from_cudnn` restates keras' `_convert_rnn_weights`.

Tensor names on this side are those of `engine.weight_shapes`.
"""
from collections import OrderedDict

import numpy as np

from . import hdf5


def _bridge(out, n, names):
    """The bridge_dense layers of encoder layer n (seq2seq.py:299-301): Dense(width, tanh) 'bridge_h_<n>', 'bridge_c_<n>'."""
    for part in ('h', 'c'):
        out['bridge_%s_%d' % (part, n)] = (['bridge_%s_%d/kernel:0' % (part, n), 'bridge_%s_%d/bias:0' % (part, n)] if names
                                            else ['bridge%d_%s_K' % (n, part), 'bridge%d_%s_b' % (n, part)])


def layer_tensors(depth, bridge_dense=False, deep=False):
    """Ordered {keras layer name: [tensor names in layer.weights order]} (seq2seq.py:239-350, attention.py:418-421,598-609)."""
    d = int(depth)
    out = OrderedDict()
    out['char_input_projection'] = ['E']
    out['encoder_lstm_1'] = ['enc1_%s_%s' % (direction, part) for direction in ('fw', 'bw') for part in 'KRb']
    if bridge_dense:
        _bridge(out, 1, False)
    for n in range(2, d + 1):
        out['encoder_lstm_%d' % n] = (['enc%d_%s_%s' % (n, direction, part) for direction in ('fw', 'bw') for part in 'KRb'] if deep
                                      else ['enc%d_%s' % (n, part) for part in 'KRb'])
        if bridge_dense:
            _bridge(out, n, False)
    out['attention_dense'] = ['att_U']
    for n in range(1, d):
        out['decoder_lstm_%d' % n] = ['dec%d_%s' % (n, part) for part in 'KRb']
    out['decoder_lstm_%d' % d] = ['att_Wa', 'att_va', 'att_bUW', 'att_bv'] + ['dec%d_%s' % (d, part) for part in 'KRb']
    return out


def _keras_weight_names(depth, bridge_dense=False, deep=False):
    """Variable names keras/TF1 gives the weights (`<scope>/<name>:0`), per layer, in order."""
    d = int(depth)
    lstm = ['kernel:0', 'recurrent_kernel:0', 'bias:0']
    out = OrderedDict()
    out['char_input_projection'] = ['char_input_projection/kernel:0']
    out['encoder_lstm_1'] = ['encoder_lstm_1/%s_encoder_lstm_1/%s' % (direction, w)
                             for direction in ('forward', 'backward') for w in lstm]
    if bridge_dense:
        _bridge(out, 1, True)
    for n in range(2, d + 1):
        out['encoder_lstm_%d' % n] = (['encoder_lstm_%d/%s_encoder_lstm_%d/%s' % (n, direction, n, w) for direction in ('forward', 'backward') for w in lstm]
                                      if deep else ['encoder_lstm_%d/%s' % (n, w) for w in lstm])
        if bridge_dense:
            _bridge(out, n, True)
    out['attention_dense'] = ['attention_dense/kernel:0']
    for n in range(1, d):
        out['decoder_lstm_%d' % n] = ['decoder_lstm_%d/%s' % (n, w) for w in lstm]
    out['decoder_lstm_%d' % d] = ['decoder_lstm_%d/%s' % (d, w) for w in ['W_a:0', 'v_a:0', 'b_UW:0', 'b_v:0'] + lstm]
    return out


def _from_cudnn(kernel, recurrent, bias):
    """CuDNNLSTM -> LSTM weights (keras/engine/saving.py `_convert_rnn_weights`, n_gates = 4): every gate block of
    the kernels is stored transposed in Fortran order, and the input and recurrent biases are separate."""
    def per_gate(mat, func):
        return np.hstack([func(k) for k in np.hsplit(mat, 4)])
    kernel = per_gate(kernel, lambda k: k.T.reshape(k.shape, order='F'))
    recurrent = per_gate(recurrent, lambda k: k.T)
    bias = np.sum(np.split(bias, 2, axis=0), axis=0)
    return kernel, recurrent, bias


def _as_names(attr):
    if attr is None:
        return None
    return [n.decode('utf-8') if isinstance(n, bytes) else str(n) for n in np.asarray(attr).ravel().tolist()]


def _guard(func):
    """Truncated or damaged files surface as hdf5.H5Error, not as struct/index errors from the parser."""
    import functools
    import struct
    from zlib import error as zlib_error

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        try:
            return func(*args, **kwargs)
        except (struct.error, IndexError, ValueError, OverflowError, UnicodeDecodeError, RecursionError, MemoryError, zlib_error) as err:
            raise hdf5.H5Error('damaged or unsupported HDF5 file "%s": %s' % (args[0] if args else '?', err))
    return wrapper


@_guard
def read_config(filename):
    """The `config` group as a dict (seq2seq.py:1143-1162); {} if the file has none (plain keras weight files)."""
    with hdf5.File(filename) as f:
        root = f
        if 'config' not in root and 'model_weights' in root and 'config' in root['model_weights']:
            root = root['model_weights']
        if 'config' not in root:
            return {}
        cfg = root['config']
        out = {}
        for key in cfg.keys():
            node = cfg[key]
            if isinstance(node, hdf5.Dataset):
                out[key] = node.read()
        return out


@_guard
def read_layers(filename):
    """{keras layer name: [arrays in weight_names order]} for every layer group that has weights."""
    with hdf5.File(filename) as f:
        root = f
        if 'layer_names' not in root.attrs and 'model_weights' in root:        # full `model.save` files (seq2seq.py:1186-1187)
            root = root['model_weights']
        names = _as_names(root.attrs.get('layer_names'))
        if names is None:
            names = [k for k in root.keys() if k != 'config']
        layers = OrderedDict()
        for lname in names:
            if lname not in root:
                continue
            g = root[lname]
            if not isinstance(g, hdf5.Group):
                continue
            wnames = _as_names(g.attrs.get('weight_names'))
            if wnames is None:
                wnames = [p for p, _ in g.visit_datasets()]
            arrays = [np.asarray(g[w].read(), np.float32) for w in wnames]
            if arrays:
                layers[lname] = arrays
        return layers


def layers_to_tensors(layers, logger=None):
    """Keras layer weight lists -> {tensor name: array}, keyed per layer: {layer: {tensor: array}}.
    The attention cell is recognised by its 7 weights, Bidirectional by 6, plain LSTM by 3."""
    out = OrderedDict()
    for lname, arrays in layers.items():
        arrays = list(arrays)
        if lname in ('char_input_projection', 'attention_dense') and len(arrays) == 1:
            out[lname] = OrderedDict([('E' if lname == 'char_input_projection' else 'att_U', arrays[0])])
            continue
        if (lname.startswith('bridge_h_') or lname.startswith('bridge_c_')) and len(arrays) == 2:
            try:
                n = int(lname.rsplit('_', 1)[1])
            except ValueError:
                continue
            part = lname[len('bridge_')]
            out[lname] = OrderedDict([('bridge%d_%s_K' % (n, part), arrays[0]), ('bridge%d_%s_b' % (n, part), arrays[1].reshape(-1))])
            continue
        if not (lname.startswith('encoder_lstm_') or lname.startswith('decoder_lstm_')):
            continue
        try:
            n = int(lname.rsplit('_', 1)[1])
        except ValueError:
            continue
        prefix = ('enc%d' if lname.startswith('encoder') else 'dec%d') % n
        tensors = OrderedDict()
        if len(arrays) == 7:
            tensors['att_Wa'], tensors['att_va'], tensors['att_bUW'], tensors['att_bv'] = arrays[:4]
            tensors['att_va'] = tensors['att_va'].reshape(-1)
            arrays = arrays[4:]
        groups = [('', arrays)]
        if len(arrays) == 6:
            groups = [('_fw', arrays[:3]), ('_bw', arrays[3:])]
        elif len(arrays) != 3:
            if logger:
                logger.warning('layer "%s" has %d weights, expected 3, 6 or 7: skipped', lname, len(arrays))
            continue
        for suffix, (k, r, b) in groups:
            units = r.shape[0]
            if b.shape == (8 * units,):
                k, r, b = _from_cudnn(k, r, b)
            tensors[prefix + suffix + '_K'], tensors[prefix + suffix + '_R'], tensors[prefix + suffix + '_b'] = k, r, b
        out[lname] = tensors
    return out


def read_model(filename, logger=None):
    """(config dict, {layer: {tensor: array}})."""
    return read_config(filename), layers_to_tensors(read_layers(filename), logger)


def write_model(filename, config, weights):
    """Write `weights` ({tensor name: array}, all tensors of the model) and `config` in the reference's layout."""
    depth = int(config['depth'])
    bridge = bool(np.asarray(config.get('bridge_dense', False)).item()) if 'bridge_dense' in config else False
    deep = bool(np.asarray(config.get('deep_bidirectional_encoder', False)).item()) if 'deep_bidirectional_encoder' in config else False
    table, knames = layer_tensors(depth, bridge, deep), _keras_weight_names(depth, bridge, deep)
    w = hdf5.Writer()
    w.set_attr('/', 'layer_names', np.array([n.encode('utf-8') for n in table], dtype='S'))
    w.set_attr('/', 'backend', np.bytes_(b'tensorflow'))
    w.set_attr('/', 'keras_version', np.bytes_(b'2.3.1'))
    for lname, tensors in table.items():
        w.create_group(lname)
        w.set_attr(lname, 'weight_names', np.array([n.encode('utf-8') for n in knames[lname]], dtype='S'))
        for tname, kname in zip(tensors, knames[lname]):
            a = np.asarray(weights[tname], np.float32)
            if tname == 'att_va':
                a = a.reshape(-1, 1)                    # attention.py:600
            w.create_dataset(lname + '/' + kname, a)
    for key, value in config.items():
        w.create_dataset('config/' + key, np.asarray(value))
    w.save(filename)
