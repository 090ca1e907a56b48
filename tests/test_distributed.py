"""The multi-GPU path is line sharding + one all-gather of result records.  Here it
runs as 2 gloo ranks on the CPU with the oracle standing in for the per-rank decoder."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cor_asv_ann_amd import sharding


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_records_roundtrip():
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 256, (5, 12)).astype(np.int32)
    prob = rng.random((5, 12)).astype(np.float32)
    length = rng.integers(0, 13, 5).astype(np.int32)
    score = rng.random(5)
    found = np.array([1, 0, 3, 1, 1], np.int32)
    out = sharding.unpack_records(sharding.pack_records(idx, prob, length, score, found))
    for a, b in zip(out, (idx, prob, length, score, found)):
        assert np.array_equal(a, b)


def test_records_from_lines():
    """Strings + probability lists (what correct_lines returns) -> records -> strings, incl. empty, truncated and
    lines whose probability list is shorter than the text (fallback lines)."""
    chars = ['', '\n'] + [chr(c) for c in range(0x61, 0x61 + 20)] + ['ß', '中']
    i_c = dict(enumerate(chars))
    lut = np.full(max(ord(c) for c in chars if c) + 2, -1, np.int32)
    for i, c in i_c.items():
        if c:
            lut[ord(c)] = i
    lines = ['abc\n', '', 'ß中a\n', 'abcdefghij\n', 'b']
    probs = [[0.5, 0.25, 1.0, 0.125], [], [0.1, 0.2, 0.3, 0.4], [0.9] * 11, []]
    scores = [0.1, 0.0, 0.3, 0.4, 0.0]
    rec = sharding.records_from_lines(lines, probs, scores, lut, 6)
    idx, prob, length, score, found = sharding.unpack_records(rec)
    assert list(length) == [4, 0, 4, 6, 1] and np.allclose(score, scores)
    assert sharding.records_to_strings(idx, length, i_c) == ['abc\n', '', 'ß中a\n', 'abcdef', 'b']
    assert np.allclose(prob[0, :4], probs[0]) and np.allclose(prob[3, :6], 0.9) and prob[4, 0] == 0 and not prob[1].any()


def _worker(rank, world, port, n_lines, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
    from oracle.decode import OracleModel, decode_batch_greedy
    cfg = ModelConfig(depth=2, width=32, voc_size=40)
    m = OracleModel(cfg, make_weights(cfg, emb_scale=6.0))
    lines, _ = make_lines(n_lines, 10, 21, voc_size=40)
    lo, hi = sharding.shard_bounds(n_lines, world, rank)
    enc_in, _, _, _ = vectorize_lines(m, lines[lo:hi], [[] for _ in range(hi - lo)])
    g = decode_batch_greedy(m, enc_in, return_indexes=True)
    S = g[5].shape[1]
    prob = np.zeros((hi - lo, S), np.float32)
    length = np.array([len(s) for s in g[1]], np.int32)
    for j, p in enumerate(g[2]):
        prob[j, :len(p)] = p
    rec = sharding.pack_records(g[5], prob, length, np.asarray(g[3]))
    allrec = sharding.all_gather_records(rec, n_lines)
    if rank == 0:
        np.save(os.path.join(tmpdir, 'gathered.npy'), allrec)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_decode_equals_unsharded(tmp_path):
    n_lines, world = 7, 2          # uneven shards: 4 + 3
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_lines, str(tmp_path)), nprocs=world, join=True)
    allrec = np.load(os.path.join(str(tmp_path), 'gathered.npy'))
    idx, prob, length, score, found = sharding.unpack_records(allrec)
    from oracle import ModelConfig, make_weights, make_lines, vectorize_lines
    from oracle.decode import OracleModel, decode_batch_greedy
    cfg = ModelConfig(depth=2, width=32, voc_size=40)
    m = OracleModel(cfg, make_weights(cfg, emb_scale=6.0))
    lines, _ = make_lines(n_lines, 10, 21, voc_size=40)
    enc_in, _, _, _ = vectorize_lines(m, lines, [[] for _ in lines])
    g = decode_batch_greedy(m, enc_in, return_indexes=True)
    assert idx.shape == g[5].shape and np.array_equal(idx, g[5])
    assert np.allclose(score, g[3], rtol=1e-5)
    assert sharding.records_to_strings(idx, length, m.mapping[1]) == g[1]


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(argv), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    return p.returncode, [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith('{')], p.stderr.decode()


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) is two ranks: launcher, sharding, record
    packing, all-gather and reporting rehearsed without a device (CASV_BENCH_DRY_RUN: ranks echo their lines)."""
    code, lines, err = _run_bench({'CASV_BENCH_DRY_RUN': '1', 'CASV_BENCH_BACKEND': 'gloo'},
                                  '--gpus', '2', '--steps', '2', '--warmup', '1', '--lines-per-gpu', '9')
    assert code == 0, err
    assert len(lines) == 1
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['gathered_records'] == 18
    assert len(out['ms_per_step_by_rank']) == 2 and out['gather_ms_per_step'] > 0
    assert out['config']['launcher'] == 'bench.py' and out['data'].startswith('dry-run')
    assert out['value'] == pytest.approx(18 * 100 * 2 / (out['ms_per_step'] * 2e-3), rel=1e-6)


def test_bench_under_torch_distributed_run_as_the_driver_starts_it():
    """The driver's N > 1 command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks, dry (no device; gloo carries the gather): every
    rank takes RANK / LOCAL_RANK / WORLD_SIZE from the launcher, pins itself to CPUs of its own, and exactly ONE JSON line comes out
    (rank 0's), with configs[4]'s shape, both ranks' own times and their host placement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CASV_BENCH_DRY_RUN='1', CASV_BENCH_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 1 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['config']['launcher'] == 'torch.distributed.run' and 'configs[4]' in out['config']['workload']
    assert out['config']['lines_per_gpu'] == 8192 and out['gathered_records'] == 16384
    assert len(out['ms_per_step_by_rank']) == 2 and len(set(out['ms_per_step_by_rank'])) == 2
    place = out['config']['host_placement_by_rank']
    assert len(place) == 2 and all(q['threads'] >= 1 for q in place)
    if len(os.sched_getaffinity(0)) >= 2:
        import bench
        a, b = (set(bench.parse_cpulist(q['cpus'])) for q in place)
        assert a and b and not (a & b)


def test_bench_fails_when_a_rank_fails():
    """Without the dry-run switch the ranks need a GPU: on a CPU box every rank fails and so does the launcher
    (no silent single-rank run, no fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    code, lines, err = _run_bench({}, '--gpus', '2', '--steps', '1', '--warmup', '0', '--lines-per-gpu', '4')
    assert code != 0 and not lines


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    code, lines, err = _run_bench({'CASV_BENCH_DRY_RUN': '1', 'WORLD_SIZE': '1', 'RANK': '0'}, '--gpus', '2', '--steps', '1')
    assert code != 0 and not lines and 'WORLD_SIZE' in err


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The real N-rank path of bench.py (two processes, each with its own model handle, decoding their shards on
    the same card; gloo carries the gather because two RCCL ranks cannot share a device)."""
    code, lines, err = _run_bench({'CASV_BENCH_BACKEND': 'gloo', 'CASV_BENCH_SAME_DEVICE': '1'},
                                  '--gpus', '2', '--steps', '1', '--warmup', '1', '--lines-per-gpu', '64')
    assert code == 0, err
    out = lines[0]
    assert out['n_gpus'] == 2 and out['gathered_records'] == 128 and out['data'] == 'synthetic'
    assert out['value'] > 0 and sum(out['kernel_ms_per_step'].values()) > 0 and len(out['ms_per_step_by_rank']) == 2


def _c5_lines():
    from cor_asv_ann_amd.synthetic import make_lines
    import bench
    wl = bench.WORKLOADS['c5']
    return wl, make_lines(wl['lines'], wl['length'], wl['seed'], voc_size=wl['voc'])[0]


@pytest.mark.gpu
def test_one_rank_of_configs4_through_the_native_rccl_gather(tmp_path):
    """BASELINE configs[4] as far as one GPU goes: ONE rank's share of the 64k-line job -- 8192 lines, decoded in eight
    1024-line batches, result records packed on the device after every batch, ONE RCCL all-gather per step through the C
    ABI (world = 1), exactly the code path rank r of `bench.py --gpus 8` runs.
      * all 8192 records arrive, in line order;
      * batch 3 (lines 3072..4095) equals those 1024 lines decoded alone (lines are independent units);
      * the device-packed records equal the records the host packs from the returned strings (the other gather path);
      * a second run gives the same bits (determinism)."""
    import numpy as np
    from cor_asv_ann_amd import sharding
    import bench
    env = {'CASV_BENCH_FORCE_DIST': '1', 'CASV_BENCH_GATHER': 'native', 'MASTER_PORT': '29611'}
    dump = str(tmp_path / 'rec.npy')
    code, lines, err = _run_bench(env, '--workload', 'c5', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--dump-records', dump)
    assert code == 0, err
    out = lines[0]
    assert out['gathered_records'] == 8192 and out['config']['lines_per_gpu'] == 8192 and out['config']['lines_per_decode_call'] == 1024
    assert out['config']['gather'].startswith('casv_comm') and out['config']['records'] == 'device-packed'
    assert out['metric'] == bench.METRIC and out['value'] > 0
    rec = np.load(dump)
    assert rec.shape == (8192, 2 * 202 + 4)
    idx, prob, length, score, found = sharding.unpack_records(rec)
    assert (length > 0).all() and (length <= 202).all() and (found == 1).all()

    # batch 3 alone, through the facade, and the host-side packing of its strings
    wl, all_lines = _c5_lines()
    s2s, _, _ = bench.make_model(0, wl['depth'], wl['width'], wl['n'], wl['emb'], wl['voc'])
    chunk = all_lines[3 * 1024:4 * 1024]
    o, p, s, _ = s2s.correct_lines(chunk, fast=False, greedy=False, alignments=False)
    want = sharding.records_from_lines(o, p, s, s2s._codepoint_lut(), 202)
    assert np.array_equal(rec[3 * 1024:4 * 1024], want)
    # ... and once more from the device, after a different batch has been through the same handle in between
    eng = s2s._require_engine()
    s2s.correct_lines(all_lines[:64], fast=False, greedy=False, alignments=False)
    s2s.correct_lines(chunk, fast=False, greedy=False, alignments=False)
    eng.records_reset(1024, 202)
    eng.records_append(0)
    assert np.array_equal(eng.records_read(), want)
    s2s.engine.close()

    dump2 = str(tmp_path / 'rec2.npy')
    code, lines2, err = _run_bench(dict(env, MASTER_PORT='29613', CASV_BENCH_RECORDS='host'), '--workload', 'c5', '--steps', '1', '--warmup', '0',
                                   '--no-cpu-baseline', '--dump-records', dump2)
    assert code == 0, err
    assert lines2[0]['config']['records'] == 'host-packed'
    assert np.array_equal(np.load(dump2), rec)


@pytest.mark.gpu
def test_torch_rccl_gather_reads_the_device_records_in_place():
    """The default gather of `bench.py --gpus N` (torch.distributed, backend nccl = RCCL) fed from the library's device-resident
    record buffer through `__cuda_array_interface__`: one rank here, same code as rank r of N."""
    code, lines, err = _run_bench({'CASV_BENCH_FORCE_DIST': '1', 'MASTER_PORT': '29615'}, '--workload', 'c5', '--lines-per-gpu', '256',
                                  '--steps', '1', '--warmup', '0', '--no-cpu-baseline')
    assert code == 0, err
    out = lines[0]
    assert out['gathered_records'] == 256 and out['config']['gather'] == 'torch.distributed/nccl' and out['config']['records'] == 'device-packed'


@pytest.mark.gpu
def test_two_gpus_over_real_rccl_when_the_box_has_them():
    """`bench.py --gpus 2` over RCCL/xGMI -- runs wherever two devices are visible (the one-GPU lease of the build skips)."""
    import cor_asv_ann_amd._native as nv
    if nv.load().casv_device_count() < 2:
        pytest.skip('one GPU on this box: the N > 1 RCCL launch is the driver\'s (8-GPU node)')
    for gather in ('', 'native'):
        code, lines, err = _run_bench({'CASV_BENCH_GATHER': gather} if gather else {}, '--gpus', '2', '--lines-per-gpu', '1024', '--steps', '1',
                                      '--warmup', '1', '--no-cpu-baseline')
        assert code == 0, err
        out = lines[0]
        assert out['n_gpus'] == 2 and out['gathered_records'] == 2048 and out['config']['records'] == 'device-packed'
        assert len(out['ms_per_step_by_rank']) == 2


def test_multi_gpu_launch_defaults_to_the_configs4_shape():
    """`--gpus N > 1` without a --workload is BASELINE configs[4]: 8192 lines per GPU per step in 1024-line batches."""
    code, lines, err = _run_bench({'CASV_BENCH_DRY_RUN': '1', 'CASV_BENCH_BACKEND': 'gloo'}, '--gpus', '2', '--steps', '1', '--warmup', '0')
    assert code == 0, err
    out = lines[0]
    assert out['config']['lines_per_gpu'] == 8192 and out['gathered_records'] == 16384 and 'configs[4]' in out['config']['workload']
    assert out['config']['lines_per_decode_call'] == 1024


def test_rank_placement_follows_the_numa_node_of_the_gpu(tmp_path, monkeypatch):
    """bench.py's host placement (what the eight processes share on the host is the scaling risk): the NUMA
    node of every GPU from the KFD topology in sysfs -- no GPU call -- and per rank a disjoint share of that node's CPUs."""
    import bench
    sysfs = tmp_path / 'sys'
    # a host like an 8-GPU node: KFD nodes 0, 1 = the two CPU sockets, nodes 2..9 = GPUs on render minors 128..135
    for i in range(10):
        d = sysfs / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes' / str(i)
        d.mkdir(parents=True)
        gpu = i >= 2
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n' % (0 if gpu else 96, 1024 if gpu else 0, 126 + i if gpu else 0))
        if gpu:
            r = sysfs / 'class' / 'drm' / ('renderD%d' % (126 + i)) / 'device'
            r.mkdir(parents=True)
            (r / 'numa_node').write_text('%d\n' % (0 if i < 6 else 1))
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    assert bench.gpu_numa_nodes(str(sysfs)) == [0, 0, 0, 0, 1, 1, 1, 1]
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,1')
    assert bench.gpu_numa_nodes(str(sysfs)) == [1, 0]
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert bench.gpu_numa_nodes(str(tmp_path / 'nothing')) is None
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    cpus_of_node = {0: list(range(0, 48)) + list(range(96, 144)), 1: list(range(48, 96)) + list(range(144, 192))}
    shares = [bench.rank_cpus(r, 8, range(192), numa, cpus_of_node) for r in range(8)]
    assert [n for _, n in shares] == numa
    assert all(len(c) == 24 and set(c) <= set(cpus_of_node[n]) for c, n in shares)
    assert len(set().union(*[set(c) for c, _ in shares])) == 192                       # disjoint, nothing left idle
    # a cpuset that leaves a node fewer CPUs than it has ranks: the even share of everything allowed instead
    tight = [bench.rank_cpus(r, 8, list(range(0, 2)) + list(range(48, 96)), numa, cpus_of_node)[0] for r in range(8)]
    assert all(tight) and sum(len(c) for c in tight) == len(set().union(*[set(c) for c in tight])) == 50
    # no topology at all (this container): contiguous shares of the allowed CPUs
    plain = [bench.rank_cpus(r, 4, range(8))[0] for r in range(4)]
    assert plain == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert bench.format_cpulist(bench.parse_cpulist('0-3,8,10-11')) == '0-3,8,10-11'


def test_eight_ranks_of_configs4_rehearsed_without_a_device(tmp_path):
    """BASELINE configs[4] with all EIGHT ranks on this host, dry (CASV_BENCH_DRY_RUN: every rank echoes its 8192 lines instead of
    decoding them; gloo carries the gather): launcher, sharding, record packing, the all-gather of 65 536 records and the
    reporting -- everything of `bench.py --gpus 8` but the device.  All records arrive, in line order; and eight ranks side by
    side do not slow each other's host work down (what the first real 8-GPU run must not discover): the median rank's time
    per step stays within 1.2x of one rank alone on the same host, the slowest within 2x."""
    import numpy as np
    from cor_asv_ann_amd import sharding
    from cor_asv_ann_amd.synthetic import make_lines, make_vocabulary
    import bench
    env = {'CASV_BENCH_DRY_RUN': '1', 'CASV_BENCH_BACKEND': 'gloo'}
    dump = str(tmp_path / 'rec8.npy')
    code, lines, err = _run_bench(env, '--gpus', '8', '--steps', '2', '--warmup', '1', '--dump-records', dump)
    assert code == 0, err
    out = lines[0]
    wl = bench.WORKLOADS['c5']
    assert out['n_gpus'] == 8 and out['gathered_records'] == 65536 and out['config']['lines_per_gpu'] == 8192
    assert len(out['ms_per_step_by_rank']) == 8
    # every rank reports its OWN time (not rank 0's eight times) and ran on CPUs of its own: disjoint sets that cover no more
    # than the host offers, BLAS / OpenMP threads capped at the rank's share
    assert len(set(out['ms_per_step_by_rank'])) == 8, out['ms_per_step_by_rank']
    place = out['config']['host_placement_by_rank']
    assert len(place) == 8 and all(p and p['threads'] >= 1 for p in place), place
    sets = [set(bench.parse_cpulist(p['cpus'])) for p in place]
    if len(os.sched_getaffinity(0)) >= 8:
        assert all(sets) and sum(len(x) for x in sets) == len(set().union(*sets)), place          # pairwise disjoint
        assert set().union(*sets) <= set(os.sched_getaffinity(0))
        assert all(p['threads'] == len(x) for p, x in zip(place, sets))
    rec = np.load(dump)
    S = 2 * (wl['length'] + 1)
    assert rec.shape == (65536, 2 * S + 4)
    idx, prob, length, score, found = sharding.unpack_records(rec)
    _, want = make_lines(65536, wl['length'], wl['seed'], voc_size=wl['voc'])
    assert (length == wl['length'] + 1).all()
    assert np.array_equal(idx[:, :wl['length'] + 1], want)            # record j is line j of the global job, on every rank's view
    # one rank alone through the same path (process group of one: packing, gather, reporting)
    one = _run_bench(dict(env, CASV_BENCH_FORCE_DIST='1', MASTER_PORT='29641'), '--gpus', '1', '--workload', 'c5', '--steps', '2',
                     '--warmup', '1', '--no-cpu-baseline')
    assert one[0] == 0, one[2]
    # a rank's own host work = its time per step without the collective (which moves 8x the bytes at 8 ranks and waits for the slowest)
    alone = one[1][0]['ms_per_step_by_rank'][0] - one[1][0]['gather_ms_per_step']
    packing8 = sorted(t - out['gather_ms_per_step'] for t in out['ms_per_step_by_rank'])
    print('host ms per step without the gather: one rank alone %.1f; eight ranks side by side: median %.1f, slowest %.1f; '
          'gather %.1f ms alone, %.1f ms at eight ranks (gloo, 107 MB per rank)'
          % (alone, packing8[4], packing8[-1], one[1][0]['gather_ms_per_step'], out['gather_ms_per_step']))
    # (a sanity bound on a shared 8-CPU container, eight processes on eight CPUs: the figure that matters is printed above)
    assert packing8[4] <= 1.5 * alone + 30.0 and packing8[-1] <= 3.0 * alone + 120.0, (alone, packing8)

