"""The optional topologies `residual_connections`, `bridge_dense` and `deep_bidirectional_encoder` (seq2seq.py:125-132,246-301,359-360)
through the C ABI against the oracle, on a real MI355X.

What they are: from the third encoder layer on a layer's output sequence is its LSTM output plus its input sequence; the final h and
c of every encoder layer pass through Dense(width, tanh) layers ('bridge_h_<n>', 'bridge_c_<n>') on their way to the decoder; with
a deep bidirectional encoder every layer is a BiLSTM that reads the "cross sum" of the layer below (neighbouring features of
[fw | bw] summed pairwise, as the reference's Lambda computes it) and hands on its backward final state, the attended width is 2W.  The
reference's INFERENCE decoder carries no residual sums (seq2seq.py:421-436 builds it layer by layer without the `add` of the training
graph) -- restated as it is, so the decoder steps of such a model are the default topology's.  Same tolerances as
tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
from oracle.decode import OracleModel, correct_lines, decode_batch_greedy

RT, AT = 2e-4, 2e-6
FLAGS = [dict(residual_connections=True), dict(bridge_dense=True), dict(residual_connections=True, bridge_dense=True),
         dict(deep_bidirectional_encoder=True), dict(deep_bidirectional_encoder=True, bridge_dense=True, residual_connections=True)]


def _engine(cfg, weights, **kw):
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(cfg.depth, cfg.width, cfg.voc_size, residual_connections=cfg.residual_connections,
                    bridge_dense=cfg.bridge_dense, deep_bidirectional_encoder=cfg.deep_bidirectional_encoder, **kw)
    eng.set_weights(weights)
    return eng


@pytest.mark.parametrize('flags', FLAGS)
@pytest.mark.parametrize('depth,width,B', [(4, 64, 9), (3, 96, 70), (2, 128, 5), (1, 64, 4)])
def test_encoder_outputs_equal_the_oracle(flags, depth, width, B):
    """enc_out and all final states (bridged where asked) of the per-step and -- where the topology has that form: bridge_dense alone,
    or depth < 3 -- of the persistent encoder, in both arithmetics."""
    cfg = ModelConfig(depth=depth, width=width, voc_size=48, **flags)
    weights = make_weights(cfg, emb_scale=12.0)
    lines, idx = make_lines(B, 19, 11, voc_size=48)
    om = OracleModel(cfg, weights)
    enc_in, _, _, _ = vectorize_lines(om, lines, [[] for _ in lines])
    want = om.encode(enc_in)
    eng = _engine(cfg, weights)
    outs = []
    for arith, persistent in ((0, -1), (0, 0), (2, -1)):
        eng.set_option('arithmetic', arith)
        eng.set_option('persistent', persistent)
        eng.encode(idx)
        enc, states = eng.encoder_outputs()
        assert np.allclose(enc, want[0], rtol=RT, atol=AT), (arith, persistent)
        assert np.allclose(np.stack(states), np.stack(want[1:-1]), rtol=RT, atol=AT), (arith, persistent)
        outs.append((enc, np.stack(states)))
    # persistent (where taken) and per-step fp32 encoders: the same bits
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    eng.close()


@pytest.mark.parametrize('flags', FLAGS)
def test_correct_lines_equals_the_oracle(flags):
    """Greedy and beamed `correct_lines` of a depth-3 model with the flags set, through the façade (attributes as the reference's
    `configure()` reads them), against the oracle: strings exact, scores 1e-4."""
    from tests.test_gpu_parity import _facade
    cfg = ModelConfig(depth=3, width=64, voc_size=64, **flags)
    weights = make_weights(cfg, emb_scale=12.0)
    om = OracleModel(cfg, weights, batch_size=4)
    lines, _ = make_lines(12, 15, 23, voc_size=64)
    s2s = _facade(cfg, weights, om.mapping, N=4, **flags)
    for fast, greedy in ((True, True), (False, False)):
        want = correct_lines(om, lines, fast=fast, greedy=greedy)
        got = s2s.correct_lines(lines, fast=fast, greedy=greedy)
        assert got[0] == want[0], (fast, greedy)
        assert np.allclose(got[2], want[2], atol=1e-4)
    s2s.engine.close()


def test_bridge_and_residual_change_the_results():
    """(the flags do something: against the default topology with the same tensors the encoder outputs / initial states differ)"""
    base = ModelConfig(depth=4, width=64, voc_size=48)
    full = ModelConfig(depth=4, width=64, voc_size=48, residual_connections=True, bridge_dense=True)
    w = make_weights(full, emb_scale=12.0)
    _, idx = make_lines(6, 19, 11, voc_size=48)
    e0, e1 = _engine(base, {k: v for k, v in w.items() if not k.startswith('bridge')}), _engine(full, w)
    e0.encode(idx); e1.encode(idx)
    a, b = e0.encoder_outputs(), e1.encoder_outputs()
    assert not np.allclose(a[0], b[0], atol=1e-3) and not np.allclose(np.stack(a[1]), np.stack(b[1]), atol=1e-3)
    e0.close(); e1.close()


def test_model_file_round_trip_with_bridge_layers(tmp_path):
    """save() / load_config() / load_weights() of a bridge_dense model: the Keras container carries 'bridge_h_<n>' / 'bridge_c_<n>'
    Dense layers (kernel, bias) and the two flags in its config group (seq2seq.py:1135-1137); a reloaded model decodes the same."""
    from tests.test_gpu_parity import _facade
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    flags = dict(residual_connections=True, bridge_dense=True)
    cfg = ModelConfig(depth=3, width=64, voc_size=64, **flags)
    weights = make_weights(cfg, emb_scale=12.0)
    om = OracleModel(cfg, weights, batch_size=4)
    lines, _ = make_lines(6, 15, 29, voc_size=64)
    s2s = _facade(cfg, weights, om.mapping, N=4, **flags)
    want = s2s.correct_lines(lines, fast=False, greedy=False)
    path = str(tmp_path / 'bridged.h5')
    s2s.save(path)
    s2s.engine.close()
    again = Sequence2Sequence()
    again.load_config(path)
    assert again.residual_connections and again.bridge_dense and again.depth == 3
    again.batch_size = 4
    again.configure()
    again.load_weights(path)
    got = again.correct_lines(lines, fast=False, greedy=False)
    assert got[0] == want[0] and np.allclose(got[2], want[2], atol=1e-6)
    again.engine.close()


@pytest.mark.parametrize('deep', [False, True])
def test_facade_trains_a_model_with_the_flags_set(tmp_path, deep):
    """cor-asv-ann-train's path (`Sequence2Sequence.train()`, seq2seq.py:590-649) on a small copy task with residual_connections and
    bridge_dense set -- and, second case, a deep bidirectional encoder on top: the validation loss falls, the trained model --
    bridges included -- decodes like the oracle on its weights and survives the reference's container."""
    import os
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    rng = np.random.default_rng(5)
    alphabet = 'abcdefghijklmnop '
    lines = [''.join(rng.choice(list(alphabet), size=int(rng.integers(6, 14)))) for _ in range(640)]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        (tmp_path / 'train.tsv').write_text(''.join('%s\t%s\n' % (l, l) for l in lines))
        s2s = Sequence2Sequence()
        s2s.depth, s2s.width, s2s.batch_size, s2s.epochs, s2s.dropout = 3, 64, 32, 16, 0.1
        s2s.residual_connections = s2s.bridge_dense = True
        s2s.deep_bidirectional_encoder = deep
        s2s._rng = np.random.default_rng(1)
        s2s.configure()
        s2s.train([str(tmp_path / 'train.tsv')])
    finally:
        os.chdir(cwd)
    hist = s2s.history
    assert s2s.status == 2 and min(h['val_loss'] for h in hist) < hist[0]['val_loss'] - 0.1, hist
    w = s2s.get_weights()
    assert 'bridge3_c_K' in w and np.abs(w['bridge1_h_b']).max() > 0          # (the Dense layers trained: their biases left zero)
    test = [l + '\n' for l in lines[:6]]
    s2s.batch_size = 4
    cfg = ModelConfig(depth=3, width=64, voc_size=s2s.voc_size, residual_connections=True, bridge_dense=True, deep_bidirectional_encoder=deep)
    assert ('enc3_bw_K' in w) == deep
    om = OracleModel(cfg, w, mapping=s2s.mapping, batch_size=4)
    got, want = s2s.correct_lines(test, fast=True, greedy=True), correct_lines(om, test, fast=True, greedy=True)
    assert got[0] == want[0] and np.allclose(got[2], want[2], atol=1e-4)
    s2s.save(str(tmp_path / 'm.h5'))
    other = Sequence2Sequence()
    other.load_config(str(tmp_path / 'm.h5')); other.configure(); other.load_weights(str(tmp_path / 'm.h5'))
    other.batch_size = 4
    assert other.correct_lines(test, fast=True, greedy=True)[0] == got[0]
