"""Host-side logic and the C-ABI surface, without a GPU: the library loads and exports every symbol
the header declares, and the facade's input layouts / file readers / containers behave like the
reference's (seq2seq.py:555-588, 919-1119, 1121-1162)."""
import os
import pickle
import re

import numpy as np
import pytest

from cor_asv_ann_amd import _native as nv
from cor_asv_ann_amd.seq2seq import Sequence2Sequence, Node
from cor_asv_ann_amd.engine import weight_shapes
from oracle import ModelConfig, make_vocabulary, weight_names
from oracle.decode import OracleModel, vectorize_lines

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cor_asv_ann_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(casv_[a-z_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = nv.load()
    names = _declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
        assert name in nv.SIGNATURES, 'no ctypes signature for %s' % name
    assert lib.casv_version().startswith(b'cor_asv_ann_amd')


def test_no_gpu_means_loud_failure():
    lib = nv.load()
    if lib.casv_device_count() > 0:
        pytest.skip('a GPU is present')
    s2s = _small_model()
    with pytest.raises(nv.NativeError):
        s2s.correct_lines(['ab\n'])


def test_weight_inventory_matches_oracle():
    for d in (1, 2, 4):
        cfg = ModelConfig(depth=d, width=64, voc_size=50)
        assert list(weight_shapes(d, 64, 50).items()) == [(n, tuple(s)) for n, s in weight_names(cfg)]


def _small_model(voc=12):
    s2s = Sequence2Sequence()
    s2s.depth, s2s.width = 2, 32
    chars = ['', '\n'] + [chr(ord('a') + i) for i in range(voc - 2)]
    s2s.mapping = ({c: i for i, c in enumerate(chars)}, {i: c for i, c in enumerate(chars)})
    s2s.voc_size = voc
    s2s.configure()
    return s2s


def test_vectorize_lines_equals_oracle_layouts():
    s2s = _small_model()
    om = OracleModel(ModelConfig(depth=2, width=32, voc_size=12), {}, mapping=s2s.mapping)
    plain = (['abc\n', 'b\n', ''], ['abd\n', 'bb\n', ''], None)
    prob = (['ab\n', 'c\n'], ['', ''], [[0.5, 0.25, 1.0], [0.75, 1.0]])
    cm = [[[('a', 0.6), ('bc', 0.4)], [], [('\n', 1.0)]], [[('zz', 0.9), ('a\a', 0.1)], [('\n', 1.0)]]]
    confmat = (cm, ['', ''], cm)
    for enc_seqs, dec_seqs, conf in (plain, prob, confmat):
        want = vectorize_lines(om, enc_seqs, dec_seqs, conf)
        got = s2s.vectorize_lines(enc_seqs, dec_seqs, conf)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        # the sparse form the device consumes describes the same dense rows
        idx, val = s2s._dense_to_sparse(want[0])
        dense = np.zeros(want[0].shape, np.float32)
        b_, t_, a_ = np.nonzero(idx >= 0)
        dense[b_, t_, idx[b_, t_, a_]] = val[b_, t_, a_]
        assert np.array_equal(dense, want[0].astype(np.float32))
        idx2, val2, _ = s2s._sparse_lines(enc_seqs, conf)
        dense2 = np.zeros(want[0].shape, np.float32)
        b_, t_, a_ = np.nonzero(idx2 >= 0)
        np.add.at(dense2, (b_, t_, idx2[b_, t_, a_]), val2[b_, t_, a_])
        assert np.array_equal(dense2, want[0].astype(np.float32))


def test_unsupported_topology_is_refused():
    for flag in ('lm_loss', 'lm_predict', 'stateful'):
        s2s = Sequence2Sequence()
        setattr(s2s, flag, True)
        with pytest.raises(NotImplementedError):
            s2s.configure()
    # residual_connections / bridge_dense / deep_bidirectional_encoder (seq2seq.py:246-301) are built since round 6: configure() takes
    # them, and a bridged model has the Dense layers' tensors
    s2s = Sequence2Sequence()
    s2s.residual_connections = s2s.bridge_dense = True
    s2s.depth, s2s.width, s2s.voc_size = 3, 32, 12
    s2s.configure()
    w = s2s.get_weights()
    assert w['bridge3_c_K'].shape == (32, 32) and w['bridge1_h_b'].shape == (32,) and not w['bridge1_h_b'].any()
    s2s = Sequence2Sequence()
    s2s.deep_bidirectional_encoder = True
    s2s.depth, s2s.width, s2s.voc_size = 3, 32, 12
    s2s.configure()
    w = s2s.get_weights()
    assert w['enc3_bw_K'].shape == (64, 128) and w['dec3_K'].shape == (96, 128) and w['att_U'].shape == (64, 32) and 'enc2_K' not in w
    s2s = Sequence2Sequence()
    s2s.scheduled_sampling = 'linear'
    with pytest.raises(NotImplementedError):
        s2s.configure()


def test_correct_lines_contract_without_device():
    s2s = _small_model()
    assert s2s.correct_lines([]) == ([], [], [], [])
    with pytest.raises(AssertionError):
        s2s.correct_lines(['a\n'], fast=True, greedy=False)


def test_save_load_roundtrip(tmp_path):
    s2s = _small_model()
    s2s.status = 2
    path = str(tmp_path / 'model.npz')
    s2s.save(path)
    other = Sequence2Sequence()
    other.load_config(path)
    assert (other.width, other.depth, other.voc_size) == (32, 2, 12) and other.mapping == s2s.mapping
    other.configure()
    other.load_weights(path)
    assert other.status == 2
    for k, v in s2s.get_weights().items():
        assert np.array_equal(other.get_weights()[k], v)


def test_gen_lines_and_map_files(tmp_path):
    tsv = tmp_path / 'a.tsv'
    tsv.write_text('abc\tabd\nb\tbb\ncab\tcab\n')
    s2s = Sequence2Sequence()
    s2s.depth, s2s.width, s2s.batch_size = 1, 32, 2
    s2s.configure()
    assert s2s.map_files([str(tsv)]) == 3
    assert s2s.mapping[0][''] == 0 and s2s.voc_size == len(set('abcd\t\n')) + 1
    batches = list(s2s.gen_lines([str(tsv)], repeat=False))
    assert batches[0][0] == ['abc\n', 'b\n'] and batches[0][2] == ['abd\n', 'bb\n'] and batches[0][1] is None
    assert batches[1][0] == ['cab\n', ''] and batches[1][3] == [str(tsv), None]      # padded last batch
    gen = s2s.gen_lines([str(tsv)], repeat=True)
    assert next(gen)[0] == ['abc\n', 'b\n'] and next(gen) is False                   # end-of-epoch signal
    # unsupervised text without tabs: source == target
    txt = tmp_path / 'b.txt'
    txt.write_text('hello\n')
    (src, conf, tgt, names), = list(s2s.gen_lines([str(txt)], repeat=False, unsupervised=True))
    assert src[0] == tgt[0] == 'hello\n'
    # pickled probability lines and confusion networks
    pkl = tmp_path / 'c.pkl'
    recs = [([('a', 0.9), ('b', 0.8), ('\n', 1.0)], 'ab\n'),
            ([[('a', 0.6), ('bc', 0.4)], [('\n', 1.0)]], 'a\n')]
    pkl.write_bytes(pickle.dumps(recs))
    (src, conf, tgt, names), = list(s2s.gen_lines([str(pkl)], repeat=False))
    assert src == ['ab\n', 'a\n'] and conf[0] == [0.9, 0.8, 1.0] and conf[1] == recs[1][0] and tgt == ['ab\n', 'a\n']


def test_node_api():
    root = Node(state=None, value='', scores=None, cost=0.0, length0=5, cost0=3.0)
    child = Node(state=None, value='x', scores=None, cost=0.5, parent=root)
    assert str(child) == 'x' and child.pro_cost() == -(0.5 + 3.0 * 3) and child > root


def test_evaluate_needs_nothing_from_the_reference(tmp_path, monkeypatch, caplog):
    """evaluate() (seq2seq.py:651-754): decoding (stubbed here: no GPU in this test) + this package's own metrics.
    Known answers: three lines, the OCR has one substitution in 'abc' -> 'abd' (4 symbols with the newline) and one
    missing character in 'b' -> 'bb'; the stub 'corrects' nothing, so all three columns report the same rates."""
    import logging
    import sys
    tsv = tmp_path / 'a.tsv'
    tsv.write_text('abc\tabd\nb\tbb\ncab\tcab\n')
    s2s = _small_model()
    s2s.status, s2s.batch_size = 2, 2
    s2s.logger = logging.getLogger('evaltest')
    monkeypatch.setattr(s2s, 'correct_lines', lambda lines, conf=None, fast=True, greedy=True:
                        (list(lines), [[1.0] * len(l) for l in lines], [0.5 if l else 0 for l in lines], [[] for _ in lines]))
    for name in list(sys.modules):
        if name.startswith('ocrd_cor_asv_ann'):
            monkeypatch.delitem(sys.modules, name, raising=False)
    with caplog.at_level(logging.INFO, logger='evaltest'):
        s2s.evaluate([str(tsv)], fast=True)
    assert not any(name.startswith('ocrd_cor_asv_ann') for name in sys.modules)
    text = '\n'.join(r.getMessage() for r in caplog.records)
    # characters: 1/4 + 1/3 + 0/4 errors over 11 aligned symbols = 2/11; weighted variance of the three rates
    rates, lens = [0.25, 1 / 3., 0.0], [4, 3, 4]
    mean = sum(r * n for r, n in zip(rates, lens)) / 11.
    var = sum(n * (r - mean) ** 2 for r, n in zip(rates, lens)) / 11.
    for col in ('OCR:   ', 'greedy:', 'beamed:'):                    # the reference's column layout (seq2seq.py:749-754)
        assert 'CER %s %.3f±%.3f' % (col, mean, var ** 0.5) in text, text
        # words: 'abc' != 'abd', 'b' != 'bb', 'cab' == 'cab': 2 of 3 one-word lines wrong
        assert 'WER %s %.3f±%.3f' % (col, 2 / 3., (2 / 9.) ** 0.5) in text, text
    assert 'finished 11 lines' in text                      # the reference logs the aligned length under that label
    # the inserted 'b' merges forward into the pair behind it; identical pairs are not counted under an equivalence predicate
    assert "OCR    confusion: ([(1, ('b', 'bb')), (1, ('c', 'd'))], 2)" in text, text
    assert 'ppl greedy: %.3f' % np.exp(1.5 / 11) in text


def test_gen_lines_pickle_formats_and_bad_line_filter(tmp_path):
    """The three pickled source forms of seq2seq.py:947-961 (plain string, probability line, confusion network), the
    missing-newline repair (:962-965), NFC normalisation (:979-980) and the training filter (:982-989)."""
    records = [
        ('abc\n', 'abd\n'),
        ([('a', 0.9), ('b', 0.8), ('\n', 1.0)], 'ab\n'),
        ([[('a', 0.6), ('o', 0.4)], [('b', 1.0)], [('\n', 1.0)]], 'ab\n'),
        ([[('x', 1.0)]], 'x\n'),                      # no trailing newline: replaced by a bare end-of-line
        ('', '\n'),
    ]
    pkl = tmp_path / 'a.pkl'
    with open(pkl, 'wb') as f:
        pickle.dump(records, f)
    s2s = Sequence2Sequence()
    s2s.batch_size = 8
    (src, conf, tgt, names), = list(s2s.gen_lines([str(pkl)], repeat=False))
    assert src[:5] == ['abc\n', 'ab\n', 'ab\n', '\n', '\n'] and tgt[:5] == ['abd\n', 'ab\n', 'ab\n', 'x\n', '\n']
    assert conf[1] == [0.9, 0.8, 1.0] and conf[2] == records[2][0] and conf[3] == [[('\n', 1.0)]] and conf[4] == [[('\n', 1.0)]]
    assert src[5:] == ['', '', ''] and conf[5:] == [[], [], []] and names[5:] == [None, None, None]
    # NFC: decomposed input comes out composed
    tsv = tmp_path / 'b.tsv'
    tsv.write_text('äb\täb\n')
    (src, _, tgt, _), = list(s2s.gen_lines([str(tsv)], repeat=False))
    assert src[0] == 'äb\n' and tgt[0] == 'äb\n'
    # training filter = Alignment.is_bad (lib/alignment.py:160-163): hopeless pairs are dropped only when training
    assert Sequence2Sequence._is_bad_pair('qwertzuiop\n', 'asdfghjkl\n')
    assert not Sequence2Sequence._is_bad_pair('qwert', 'asdfg')               # short lines are never dropped
    assert not Sequence2Sequence._is_bad_pair('the quick brown\n', 'the quick brovvn\n')
    bad = tmp_path / 'c.tsv'
    bad.write_text('qwertzuiop\tasdfghjkl\ngood line\tgood lime\n')
    (src, _, _, _), = list(s2s.gen_lines([str(bad)], repeat=False, train=True))
    assert src[0] == 'good line\n' and src[1] == ''
    (src, _, _, _), = list(s2s.gen_lines([str(bad)], repeat=False, train=False))
    assert src[:2] == ['qwertzuiop\n', 'good line\n']


def test_batch_prefetch_runs_ahead_keeps_order_and_hands_errors_over():
    """training.prefetch: the worker thread of `train()` (the reference's GeneratorEnqueuer, keras_train.py:133-145)."""
    import threading
    import time
    from cor_asv_ann_amd.training import prefetch
    seen = []

    def slow_producer(n):
        for i in range(n):
            time.sleep(0.02)
            seen.append((i, threading.current_thread().name))
            yield i

    t0 = time.perf_counter()
    out = []
    for item in prefetch(slow_producer(10), depth=2):
        time.sleep(0.02)                 # the "device step": overlaps with the production of the next items
        out.append(item)
    elapsed = time.perf_counter() - t0
    assert out == list(range(10)) and all(name == 'casv-batch-prefetch' for _, name in seen)
    assert elapsed < 0.33, elapsed       # serial would be 0.4 s

    def failing():
        yield 1
        raise KeyError('boom')
    got = []
    with pytest.raises(KeyError):
        for item in prefetch(failing()):
            got.append(item)
    assert got == [1]
    # a consumer that stops early releases the producer (it must not stay blocked on a full queue)
    gen = prefetch(iter(range(1000)), depth=1)
    assert next(gen) == 0
    gen.close()
    time.sleep(0.3)
    assert not any(t.name == 'casv-batch-prefetch' and t.is_alive() for t in threading.enumerate())


def test_prefetch_joins_its_worker_by_default():
    """without `detach_after` (train()'s stages: their producers draw from the model's shared random generator) a
    consumer that leaves early waits for the worker, however slow its current next() is -- no thread is left behind."""
    import threading
    import time
    from cor_asv_ann_amd.training import prefetch
    state = {'after_close': False, 'closed': False}

    def slow():
        yield 0
        time.sleep(0.8)                     # "loads a pickle at epoch start"
        state['after_close'] = state['closed']
        yield 1

    gen = prefetch(slow(), depth=1)
    assert next(gen) == 0
    gen.close()
    state['closed'] = True
    assert not any(t.name == 'casv-batch-prefetch' and t.is_alive() for t in threading.enumerate())
    assert state['after_close'] is False    # the producer's slow step ended BEFORE close() returned


def test_prefetch_leaves_a_blocked_producer_behind_but_waits_for_a_device_call():
    """a consumer that leaves early must not hang on a worker that is blocked in next(iterable) (a user
    generator reading a pipe, a nested stage waiting on q.get()), but it must wait as long as it takes while the worker is
    inside a call on the engine (`in_call`)."""
    import threading
    import time
    from cor_asv_ann_amd.training import prefetch
    release = threading.Event()

    def blocked():
        yield 0
        release.wait(30)                   # "reads a pipe"
        yield 1

    gen = prefetch(blocked(), depth=1, detach_after=0.3)
    assert next(gen) == 0
    t0 = time.perf_counter()
    gen.close()
    assert time.perf_counter() - t0 < 2.0
    release.set()

    # nested stages: the inner consumer loop (the outer stage's worker) ends when the caller's `cancel` is set
    cancel, release2 = threading.Event(), threading.Event()

    def inner_source():
        yield 0
        release2.wait(30)
        yield 1

    def outer_source():
        for item in prefetch(inner_source(), depth=1, cancel=cancel, detach_after=0.3):
            yield item

    gen = prefetch(outer_source(), depth=1, detach_after=5.0)
    assert next(gen) == 0
    t0 = time.perf_counter()
    cancel.set()
    gen.close()
    assert time.perf_counter() - t0 < 2.0
    release2.set()

    # a device call in flight is waited for, beyond detach_after
    in_call, finished = threading.Event(), []

    def device_stage():
        yield 0
        in_call.set()
        time.sleep(0.8)                    # "the C-ABI call"
        finished.append(True)
        in_call.clear()
        yield 1

    gen = prefetch(device_stage(), depth=1, in_call=in_call, detach_after=0.1)
    assert next(gen) == 0
    time.sleep(0.1)
    gen.close()
    assert finished == [True]


def test_switch_interval_guard_is_counted_and_not_held_across_yields():
    """correct_batches must not leave the process-wide switch interval changed while its generator is
    suspended or abandoned; overlapping pipelines must restore the original value."""
    import sys
    from cor_asv_ann_amd.seq2seq import _ShortSwitchInterval
    before = sys.getswitchinterval()
    a, b = _ShortSwitchInterval(), _ShortSwitchInterval()
    a.__enter__(); b.__enter__()
    assert sys.getswitchinterval() <= before and sys.getswitchinterval() <= 1e-3
    a.__exit__(None, None, None)           # (the first to enter leaves first: the "wrong" order)
    assert sys.getswitchinterval() <= 1e-3
    b.__exit__(None, None, None)
    assert sys.getswitchinterval() == before


def test_no_kernel_reads_a_register_whose_hidden_load_is_in_flight():
    """The GEMM kernels hide their tile loads from the compiler's wait bookkeeping (asm loads + counted s_waitcnt); a register
    copy the compiler inserts between such a load and its wait would move stale data.  csrc/check_asm_loads.py compiles the
    kernels to ISA and walks every path."""
    import os, subprocess, sys
    import pytest
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'cor_asv_ann_amd', 'csrc', 'check_asm_loads.py')],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr


def test_bench_counts_the_cpus_of_its_cgroup(tmp_path, monkeypatch):
    """bench.py's cpu_baseline sizes its worker pool to the CPUs the job may really use: the affinity mask capped by the cgroup's CPU
    quota (a one-GPU box shows 256 CPUs in the mask and hands out 16: `cpu.max` = "1600000 100000")."""
    import os
    import bench
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(256)))
    monkeypatch.delenv('CASV_BENCH_CPUS', raising=False)
    (tmp_path / 'cpu.max').write_text('1600000 100000\n')
    assert bench.host_cores(str(tmp_path)) == 16
    (tmp_path / 'cpu.max').write_text('max 100000\n')
    assert bench.host_cores(str(tmp_path)) == 256
    os.remove(tmp_path / 'cpu.max')
    os.makedirs(tmp_path / 'cpu')
    (tmp_path / 'cpu' / 'cpu.cfs_quota_us').write_text('800000\n')
    (tmp_path / 'cpu' / 'cpu.cfs_period_us').write_text('100000\n')
    assert bench.host_cores(str(tmp_path)) == 8                      # cgroup v1
    monkeypatch.setenv('CASV_BENCH_CPUS', '4')
    assert bench.host_cores(str(tmp_path)) == 4
    assert bench.host_cores(str(tmp_path / 'nowhere')) == 4
