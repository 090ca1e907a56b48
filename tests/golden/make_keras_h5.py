"""Writes tests/golden/keras_*.h5 with h5py/libhdf5 in the layout keras 2.3.1 `save_weights` + the reference's
`Sequence2Sequence.save` produce (keras/engine/saving.py save_weights_to_hdf5_group; seq2seq.py:1121-1141).

Needs an interpreter with h5py (this image: /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py).  The files are
DATA for the dependency-free reader in cor_asv_ann_amd/hdf5.py: they are written by libhdf5 itself, so the reader is
checked against the real container format rather than against its own writer.

  keras_d2_w32_v12.h5        every layer of the training model incl. the weight-less ones, fixed-length string
                             attributes (h5py 2.x, the version of the keras 2.3 era), LSTM-format weights
  keras_d2_w16_v12_cudnn.h5  same tensors, encoder/decoder LSTM layers in CuDNNLSTM format (models trained on a GPU,
                             seq2seq.py:216-219), variable-length string attributes (h5py 3.x), gzip-chunked kernels
  keras_d1_w16_v12.h5        a depth-1 model (shallower-model transfer, seq2seq.py:1206-1211)
"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle.weights import ModelConfig, make_vocabulary, make_weights  # noqa: E402


def to_cudnn(kernel, recurrent, bias):
    """LSTM -> CuDNNLSTM (keras `_convert_rnn_weights`, from_cudnn=False)."""
    def per_gate(mat, func):
        return np.hstack([func(k) for k in np.hsplit(mat, 4)])
    return (per_gate(kernel, lambda k: k.T.reshape(k.shape)), per_gate(recurrent, lambda k: k.T),
            np.tile(0.5 * bias, 2))


def layer_list(d):
    """(layer name, [(weight name, tensor name)]) in the order of encoder_decoder_model.layers."""
    lstm = [('kernel:0', 'K'), ('recurrent_kernel:0', 'R'), ('bias:0', 'b')]
    layers = [('encoder_input', []), ('char_input_projection', [('char_input_projection/kernel:0', 'E')]),
              ('encoder_lstm_1', [('encoder_lstm_1/%s_encoder_lstm_1/%s' % (long, w), 'enc1_%s_%s' % (short, t))
                                  for long, short in (('forward', 'fw'), ('backward', 'bw')) for w, t in lstm]),
              ('dropout_1', [])]
    for n in range(2, d + 1):
        layers += [('encoder_lstm_%d' % n, [('encoder_lstm_%d/%s' % (n, w), 'enc%d_%s' % (n, t)) for w, t in lstm]),
                   ('dropout_%d' % n, [])]
    layers += [('decoder_input', []), ('attention_state_init', []), ('attention_dense', [('attention_dense/kernel:0', 'att_U')])]
    for n in range(1, d):
        layers += [('decoder_lstm_%d' % n, [('decoder_lstm_%d/%s' % (n, w), 'dec%d_%s' % (n, t)) for w, t in lstm]),
                   ('dropout_%d' % (d + n), [])]
    top = [('W_a:0', 'att_Wa'), ('v_a:0', 'att_va'), ('b_UW:0', 'att_bUW'), ('b_v:0', 'att_bv')] + \
          [(w, 'dec%d_%s' % (d, t)) for w, t in lstm]
    layers += [('decoder_lstm_%d' % d, [('decoder_lstm_%d/%s' % (d, w), t) for w, t in top]), ('char_output_projection', [])]
    return layers


def write(path, d, W, V, cudnn=False, vlen_attrs=False, libver='earliest'):
    cfg = ModelConfig(depth=d, width=W, voc_size=V)
    w = make_weights(cfg)
    _, i_c = make_vocabulary(V)

    def names(lst):
        return [n.encode('utf8') for n in lst] if vlen_attrs else np.array([n.encode('utf8') for n in lst] or [], dtype='S')

    with h5py.File(path, 'w', libver=libver) as f:
        layers = layer_list(d)
        f.attrs['layer_names'] = names([l for l, _ in layers])
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['keras_version'] = '2.3.1'.encode('utf8')
        for lname, weights in layers:
            g = f.create_group(lname)
            if vlen_attrs and not weights:
                g.attrs['weight_names'] = np.zeros((0,), dtype='S1')
            else:
                g.attrs['weight_names'] = names([n for n, _ in weights])
            vals = {t: np.asarray(w[t], np.float32) for _, t in weights}
            if 'att_va' in vals:
                vals['att_va'] = vals['att_va'].reshape(-1, 1)
            if cudnn:
                for prefix in set(t[:-2] for t in vals if t.endswith('_K') and not t.startswith('dec%d_' % d)):
                    vals[prefix + '_K'], vals[prefix + '_R'], vals[prefix + '_b'] = to_cudnn(
                        vals[prefix + '_K'], vals[prefix + '_R'], vals[prefix + '_b'])
            for wname, t in weights:
                val = vals[t]
                if cudnn and t.endswith('_K'):
                    g.create_dataset(wname, data=val, chunks=(max(1, val.shape[0] // 2), val.shape[1]), compression='gzip')
                else:
                    dset = g.create_dataset(wname, val.shape, dtype=val.dtype)
                    dset[:] = val
        config = f.create_group('config')
        config.create_dataset('width', data=np.array(W))
        config.create_dataset('depth', data=np.array(d))
        config.create_dataset('stateful', data=np.array(False))
        config.create_dataset('residual_connections', data=np.array(False))
        config.create_dataset('deep_bidirectional_encoder', data=np.array(False))
        config.create_dataset('bridge_dense', data=np.array(False))
        config.create_dataset('mapping', data=np.fromiter((ord(i_c[i]) if i in i_c and i_c[i] else 0 for i in range(V)),
                                                          dtype=np.uint32))


if __name__ == '__main__':
    write(os.path.join(HERE, 'keras_d2_w32_v12.h5'), 2, 32, 12)
    write(os.path.join(HERE, 'keras_d2_w16_v12_cudnn.h5'), 2, 16, 12, cudnn=True, vlen_attrs=True)
    write(os.path.join(HERE, 'keras_d1_w16_v12.h5'), 1, 16, 12)
    print('h5py', h5py.__version__, 'hdf5', h5py.version.hdf5_version)
