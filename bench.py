#!/usr/bin/env python3
"""Benchmark of the hot path on the metric of BASELINE.json:
corrected chars/sec (whole node) at beam=8, depth-4 width-512, 100-char lines.

One "step" = one pass of `Sequence2Sequence.correct_lines(..., fast=False, greedy=False)` (vectorise ->
encode -> beamed decode -> strings) over one batch of 1024 synthetic 100-character lines per GPU
(BASELINE.json configs[2]).  With N > 1 every rank (one process per GPU) decodes its own lines per step --
lines are independent, so the path shards with no data-path collective -- and one RCCL all-gather of the
result records makes all decoded lines available on all ranks (BASELINE.json configs[4], weak scaling).

`python bench.py --gpus N` starts the N ranks itself (the parent never touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks already
(RANK / LOCAL_RANK / WORLD_SIZE from the environment).

Workloads (`--workload`; default c3 on one GPU, c5 on several): c2 = greedy decode of BASELINE configs[1], c3 = the metric's
beamed decode (configs[2]), c4 = train step of configs[3], c5 = configs[4]'s shape (8192 lines per GPU per step, decoded in
1024-line batches, one all-gather per step), page = the OCR-D processor's call (wrapper/transcode.py:110-115 with the
defaults of wrapper/ocrd-tool.json: depth 2, width 512, V 640, one page of 40 confusion-network lines, 256 hypotheses per
step, alignments requested).  The default one-GPU run also times c2, c4 and page after the headline's timed region (child
processes, bounded) and reports them under "other_workloads".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, WIDTH, VOC, LINES, LENGTH, BEAM_N = 4, 512, 256, 1024, 100, 8
LINE_SEED = 103
# Embedding scale of the synthetic weights.  BASELINE.md asks for N(0,(4/sqrt(W))^2); with the tiny
# activations of a depth-4 random model that gives a FLAT softmax (max p = 0.004), the beam then keeps only
# the rejection candidate and the search degenerates to 1 hypothesis x T steps per line (16x less work).
# 128/sqrt(W) gives a peaky distribution (median p = 0.8) and the full 8-hypotheses x 2T-steps search the
# metric is about ("Synthetic weights").
EMB_SCALE = 128.0
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2516.6         # dense bf16: 16x the fp32-input rate (v_mfma_f32_32x32x16_bf16: 32 768 FLOP in 32 cycles per SIMD at 2.4 GHz)
PEAK_HBM_BYTES_PER_S = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
METRIC = 'corrected chars/sec (whole node) at beam=8, depth-4 width-512, 100-char lines'


# ------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a WORLD_SIZE in the environment -> N child ranks of this script
# ------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Start ranks 0..n-1 as child processes (this process has made no GPU call and makes none), hand
    rank 0's JSON line through, exit non-zero if any rank fails.  Children are stopped by exact PID."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    while failed is None and any(p.poll() is None for p in procs[1:]):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if procs[0].poll() is not None and failed is None:
            break                     # rank 0 is done (its pipe is read below); the others follow
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write('bench.py: rank %d exited with code %s\n' % (failed, procs[failed].returncode))
        return 1
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    # rank 0 prints the one JSON line; anything else a library wrote to its stdout (gloo's connection notice) goes to stderr
    for text in out.decode().splitlines():
        (sys.stdout if text.startswith('{') else sys.stderr).write(text + '\n')
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write('bench.py: rank exit codes %s\n' % codes)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------------
# host placement of a rank (the scaling risk of the line-sharded job is what the processes share on the host)
# ------------------------------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs='/sys'):
    """NUMA node of every GPU in HIP's device order, read from sysfs WITHOUT touching the GPU: the KFD topology lists the
    nodes in the order the runtime enumerates them (GPU nodes have simd_count > 0 and a drm_render_minor), the render node's
    PCI device says which NUMA node it hangs on.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES given as plain index lists are
    applied.  None where the information is not there (no GPU, a container without the topology)."""
    base = os.path.join(sysfs, 'class', 'kfd', 'kfd', 'topology', 'nodes')
    try:
        ids = sorted(int(x) for x in os.listdir(base) if x.isdigit())
    except OSError:
        return None
    nodes = []
    for i in ids:
        props = {}
        try:
            with open(os.path.join(base, str(i), 'properties')) as f:
                for row in f:
                    k, _, v = row.strip().partition(' ')
                    props[k] = v
        except OSError:
            continue
        if int(props.get('simd_count', '0') or 0) <= 0:
            continue                    # a CPU node
        numa = -1
        try:
            with open(os.path.join(sysfs, 'class', 'drm', 'renderD%s' % props.get('drm_render_minor', ''), 'device', 'numa_node')) as f:
                numa = int(f.read().strip())
        except (OSError, ValueError):
            pass
        nodes.append(numa)
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'):
        sel = os.environ.get(var)
        if sel:
            try:
                nodes = [nodes[int(x)] for x in sel.split(',') if x.strip()]
            except (ValueError, IndexError):
                return None
    return nodes or None


def rank_cpus(local_rank, local_world, allowed, numa_of_gpu=None, cpus_of_node=None):
    """The CPUs of local rank `local_rank` of `local_world` on this host: the allowed CPUs of its GPU's NUMA node, shared out
    evenly (contiguous slices) among the ranks whose GPUs hang on the same node; without NUMA information -- or where some
    node has fewer allowed CPUs than ranks -- an even share of all allowed CPUs, for EVERY rank (one rule per host, so that the
    shares are disjoint by construction).  Returns (cpus, numa node or -1)."""
    allowed = sorted(allowed)
    node_of = [(numa_of_gpu[r] if numa_of_gpu and r < len(numa_of_gpu) else -1) for r in range(local_world)]
    by_node = all(n >= 0 for n in node_of) and bool(cpus_of_node)
    pools = {}
    if by_node:
        ok = set(allowed)
        for n in set(node_of):
            pools[n] = [c for c in cpus_of_node.get(n, []) if c in ok]
            if len(pools[n]) < node_of.count(n):
                by_node = False
    if by_node:
        node = node_of[local_rank]
        peers, pool = [r for r in range(local_world) if node_of[r] == node], pools[node]
    else:
        peers, pool = list(range(local_world)), allowed
    k, n = peers.index(local_rank), len(peers)
    if len(pool) < n:
        return pool, node_of[local_rank]             # fewer CPUs than ranks: no pinning worth the name
    return pool[k * len(pool) // n:(k + 1) * len(pool) // n], node_of[local_rank]


def place_rank(local_rank, local_world):
    """Called by every rank of a multi-rank run BEFORE numpy / torch / the GPU library are loaded: CPU affinity to the NUMA node
    of the rank's GPU and a cap on the BLAS / OpenMP threads at the rank's share of the host, so that eight processes do not
    each start a thread per core of the whole machine.  Returns what the line reports."""
    if os.environ.get('CASV_BENCH_NO_PIN'):
        return {'cpus': 'unpinned (CASV_BENCH_NO_PIN)', 'numa_node': None, 'threads': None}
    allowed = sorted(os.sched_getaffinity(0))
    numa = gpu_numa_nodes()
    cpus_of_node = {}
    for node in set(numa or []):
        if node >= 0:
            try:
                with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
                    cpus_of_node[node] = parse_cpulist(f.read())
            except OSError:
                pass
    cpus, node = rank_cpus(local_rank, local_world, allowed, numa, cpus_of_node)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return {'cpus': 'unpinned (sched_setaffinity refused)', 'numa_node': node, 'threads': None}
    threads = max(1, len(cpus))
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ[var] = str(threads)
    return {'cpus': format_cpulist(cpus), 'numa_node': node if node >= 0 else None, 'threads': threads}


def format_cpulist(cpus):
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else '%d-%d' % (cpus[i], cpus[j]))
        i = j + 1
    return ','.join(out)


# ------------------------------------------------------------------------------------------------------
def dtype_label(arith, beamed):
    """The line's `dtype`: the arithmetic type the path computes in.  Operands and results are float32 everywhere; what differs is the
    matrix instruction the products run on."""
    split = 'f32 as three bf16 parts per operand value (exact), six bf16-MFMA products per term, fp32 accumulation'
    if arith == 'fp32' or (arith == 'auto' and not beamed):
        return 'f32'
    if arith == 'split':
        return split
    return 'f32; beam-search decoder steps: ' + split + '; encoder: fp32-input MFMA'


def make_model(device, depth=DEPTH, width=WIDTH, batch_size=BEAM_N, emb_scale=EMB_SCALE, voc=VOC):
    from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_vocabulary
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    cfg = ModelConfig(depth=depth, width=width, voc_size=voc)
    weights = make_weights(cfg, emb_scale=emb_scale)
    import logging
    logger = logging.getLogger('bench')
    logger.setLevel(logging.CRITICAL)     # lines without a finished hypothesis fall back to the input (seq2seq.py:826-836)
    s2s = Sequence2Sequence(logger=logger, device=device)   # and are logged as errors: hundreds per step with random weights
    s2s.depth, s2s.width, s2s.batch_size = depth, width, batch_size
    s2s.mapping, s2s.voc_size = make_vocabulary(voc), voc
    s2s.configure()
    s2s.set_weights(weights)
    s2s.status = 2
    return s2s, cfg, weights


def survey_flop_per_char(d=DEPTH, W=WIDTH, N=BEAM_N, V=VOC, L=LENGTH):
    """F = F_enc + N*S*F_row per line, divided by the L corrected characters."""
    K, T = 11, L + 1
    C = 2 * W if d == 1 else W
    f_row = 2 * V * W + (d - 1) * 16 * W * W + 2 * W * W + K * (4 * W + 2 * C) + 8 * W * (2 * W + C) + 2 * W * V
    f_enc = T * (32 * W * W + (24 * W * W if d >= 2 else 0) + 16 * W * W * max(d - 2, 0) + 2 * C * W)
    return (f_enc + N * 2 * T * f_row) / float(L)


def executed_flop_per_char(d=DEPTH, W=WIDTH, N=BEAM_N, V=VOC, L=LENGTH):
    """What the kernels execute per corrected character: as survey_flop_per_char, except that decoder layer 1 contracts the
    fed-back distribution with the folded E.K (K = Vp + W instead of an embedding GEMV plus K = 2W)."""
    K, T = 11, L + 1
    C = 2 * W if d == 1 else W
    Vp = -(-V // 32) * 32
    if d == 1:
        lstm = 8 * W * (Vp + C + W)
    else:
        lstm = 8 * W * (Vp + W) + (d - 2) * 16 * W * W + 8 * W * (2 * W + C)
    f_row = lstm + 2 * W * W + K * (4 * W + 2 * C) + 2 * W * V
    f_enc = T * (32 * W * W + (24 * W * W if d >= 2 else 0) + 16 * W * W * max(d - 2, 0) + 2 * C * W)
    return (f_enc + N * 2 * T * f_row) / float(L)


def survey_hbm_bytes_per_char(d=DEPTH, W=WIDTH, N=BEAM_N, V=VOC, L=LENGTH):
    """Q = Q_enc + N*S*Q_row per line (per-beam state in HBM, weights on chip), per corrected character."""
    K, T = 11, L + 1
    C = 2 * W if d == 1 else W
    q_row = 4 * (4 * d * W + K * (W + C) + 2 * V + 2 * T)
    q_enc = 4 * T * (1 + 2 * 2 * W + 2 * W * max(d - 2, 0) + C + W)
    return (q_enc + N * 2 * T * q_row) / float(L)


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def host_cores(cgroup='/sys/fs/cgroup'):
    """CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota where there is one (a GPU box hands a
    one-GPU job a share of the host -- 16 CPUs -- while the affinity mask still shows all 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in (os.path.join(cgroup, 'cpu.max'), os.path.join(cgroup, 'cpu', 'cpu.cfs_quota_us')):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith('cpu.max'):
                if parts[0] != 'max':
                    n = min(n, max(1, int(round(int(parts[0]) / float(parts[1])))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open(os.path.join(cgroup, 'cpu', 'cpu.cfs_period_us')) as f:
                        n = min(n, max(1, int(round(quota / float(f.read().split()[0])))))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = os.environ.get('CASV_BENCH_CPUS')          # the operator's word where neither says it
    return max(1, min(n, int(cap))) if cap and cap.isdigit() else n


_CPU_MODEL = None


def _cpu_init(depth, width, voc, emb, batch_size, beam, threads):
    """Worker start (a fresh interpreter: `spawn`): the oracle model of this workload, BLAS threads capped at the worker's share."""
    global _CPU_MODEL
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(threads)
    except Exception:
        pass
    from oracle import ModelConfig, make_weights
    from oracle.decode import OracleModel
    cfg = ModelConfig(depth=depth, width=width, voc_size=voc)
    _CPU_MODEL = OracleModel(cfg, make_weights(cfg, emb_scale=emb), batch_size=batch_size, recompute_u=True, **beam)


def _cpu_decode(job):
    """One worker's share of a sample: `correct_lines` of the oracle in the reference's dataflow."""
    chunk, fast, confmat = job
    from oracle.decode import correct_lines
    kw = dict(fast=True, greedy=True) if fast else dict(fast=False, greedy=False)
    if not chunk:
        return 0
    (correct_lines(_CPU_MODEL, chunk, conf=chunk, **kw) if confmat else correct_lines(_CPU_MODEL, chunk, **kw))
    return len(chunk)


def cpu_baseline(cfg, emb, lines, batch_size, fast, length=LENGTH, budget_s=60.0, repeats=5, warmups=2, min_lines=16, confmat=False, **beam):
    """The oracle in the reference's dataflow (per-character decoder call, dense-T attention, u recomputed every step, per-line
    best-first search / batched greedy loop) on the host cores, BASELINE.md section 3's recipe: best of `repeats` runs over the
    same sample of >= `min_lines` lines after `warmups` warm-up runs -- as far as the time budget carries (the sample line says
    what was run).  Lines are independent, so the sample is spread over worker processes, each with its share of the BLAS threads
    (the search's GEMMs have 8 rows: one process does not keep 64 cores busy; the reference itself is one process whose Keras CPU
    path would parallelise inside each op).  confmat: the lines are confusion networks, handed over as
    `correct_lines(lines, conf=lines)` (wrapper/transcode.py:111-115)."""
    import multiprocessing as mp
    cores = host_cores()
    # wide searches (the page call's 256 rows per step) get more threads per process, narrow ones more processes
    workers = max(1, min(16, cores // (16 if batch_size >= 64 and not fast else 4), len(lines)))
    threads = max(1, cores // workers)
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    for k in saved:                       # read by the workers' BLAS when THEY load it (this process has loaded its own long ago)
        os.environ[k] = str(threads)
    t_start = time.perf_counter()
    try:
        pool = mp.get_context('spawn').Pool(workers, initializer=_cpu_init,
                                            initargs=(cfg.depth, cfg.width, cfg.voc_size, emb, batch_size, beam, threads))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def run(sample):
        share = -(-len(sample) // workers)
        jobs = [(sample[w * share:(w + 1) * share], fast, confmat) for w in range(workers)]
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_decode, jobs, chunksize=1))
        assert done == len(sample)
        return time.perf_counter() - t0
    try:
        n0 = workers * (8 if fast else 1)           # first warm-up: BLAS threads, page-in; also sizes the sample
        n0 = min(n0, len(lines))
        first = run(lines[:n0])
        left = lambda: budget_s - (time.perf_counter() - t_start)
        per_round = first / (n0 / float(workers))           # seconds per line of a worker's share
        # the sample: >= min_lines lines, more if `repeats` runs of it still fit the budget
        n = int(min(len(lines), max(min_lines, workers * int(max(1.0, left() / (repeats + warmups - 1) / per_round)))))
        n = max(workers, n - n % workers) if n >= workers else n
        t_run = per_round * -(-n // workers)
        done_warm = 1
        while done_warm < warmups and left() > (repeats + 1) * t_run:
            run(lines[:n])
            done_warm += 1
        best, runs = None, 0
        while runs < repeats and (runs == 0 or left() > t_run):
            if runs == 0 and left() < t_run and n0 == n:
                break                       # (a wide search takes its whole budget for one round: the warm-up run is the sample)
            dt = run(lines[:n])
            best = dt if best is None or dt < best else best
            runs += 1
        if best is None:
            best, n, runs, done_warm = first, n0, 1, 0
    finally:
        pool.close()
        pool.join()
    return {'value': n * length / best, 'unit': 'chars/s', 'cores': workers * threads, 'kind': 'port',
            'sample': 'best of %d runs after %d warm-up runs over %d lines of the same workload (numpy fp32 oracle, reference dataflow: '
                      'per-character decoder call, dense-T attention, u recomputed per step), lines spread over %d worker processes x %d BLAS '
                      'threads, %.1f s per run' % (runs, done_warm, n, workers, threads, best)}


# ------------------------------------------------------------------------------------------------------
def train_bench(args):
    """BASELINE configs[3]: depth 4, width 512, teacher-forced train step (forward + backward + clip + Adam) on 512
    lines of 100 characters (targets = sources with 5 % substitutions), dropout 0.2.  One GPU."""
    import numpy as np
    from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_lines
    from cor_asv_ann_amd.engine import HipEngine
    B = 512
    cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
    eng = HipEngine(DEPTH, WIDTH, VOC)
    eng.set_weights(make_weights(cfg, emb_scale=4.0))
    _, sidx = make_lines(B, LENGTH, 104, voc_size=VOC)
    rng = np.random.default_rng(1104)
    tidx = sidx.copy()
    sub = rng.random(tidx.shape) < 0.05
    sub[:, -1] = False
    tidx[sub] = rng.integers(2, VOC, size=int(sub.sum()))
    U = LENGTH + 2
    dec_in = np.full((B, U), -1, np.int32)
    dec_out = np.full((B, U), -1, np.int32)
    dec_in[:, 1:LENGTH + 2] = tidx
    dec_out[:, :LENGTH + 1] = tidx
    wts = (dec_out >= 0).astype(np.float32)
    keep = lambda shape: ((rng.random(shape) >= 0.2) / 0.8).astype(np.float32)
    masks = {'enc': [keep(2 * WIDTH if n == 0 else WIDTH) for n in range(DEPTH)], 'dec': [keep(WIDTH) for _ in range(DEPTH - 1)],
             'cell': keep((B, 2 * WIDTH))}
    for opt in ('persistent', 'fused_backward', 'vendor_gemm'):    # A/B switches: 0 = one launch per time step and operation
        if os.environ.get('CASV_OPT_' + opt.upper()):
            eng.set_option(opt, int(os.environ['CASV_OPT_' + opt.upper()]))
    eng.train_begin()
    facade = None
    if args.facade:
        # the batches as `Sequence2Sequence.train()` gets them: read from a TSV file, vectorised, degraded, dropout masks drawn --
        # by the worker thread of training.prefetch while the device runs the step before (keras_train.py:133-145)
        import tempfile
        from cor_asv_ann_amd import training
        from cor_asv_ann_amd.seq2seq import Sequence2Sequence
        from cor_asv_ann_amd.synthetic import make_vocabulary
        i_c = make_vocabulary(VOC)[1]
        n_batches = args.warmup + 2 * args.steps          # (the timed steps, and again for the per-kernel pass)
        src_lines, sidx_all = make_lines(B * n_batches, LENGTH, 104, voc_size=VOC)
        tgt_idx = sidx_all[:, :LENGTH].copy()
        sub2 = rng.random(tgt_idx.shape) < 0.05
        tgt_idx[sub2] = rng.integers(2, VOC, size=int(sub2.sum()))
        tmp = tempfile.NamedTemporaryFile('w', suffix='.tsv', delete=False, encoding='utf-8')
        for a, row in zip(src_lines, tgt_idx):
            tmp.write('%s\t%s\n' % (a[:-1], ''.join(i_c[int(c)] for c in row)))
        tmp.close()
        s2s = Sequence2Sequence(device=0)
        s2s.depth, s2s.width, s2s.batch_size, s2s.dropout = DEPTH, WIDTH, B, 0.2
        s2s.mapping, s2s.voc_size = make_vocabulary(VOC), VOC
        s2s.status = 1
        facade = training.prefetch(training.train_batches(s2s, [tmp.name], None, np.random.default_rng(7)))

    def one_step():
        if facade is None:
            return eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=1)
        idx_, val_, din_, dout_, w_, masks_ = next(facade)
        return eng.train_step(idx_, val_, din_, dout_, w_, masks_, mode=1)
    for _ in range(args.warmup):
        one_step()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, norm = one_step()
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    # the GEMMs' own time: a second pass over the same steps with HIP events around every launch (kept out of the timed
    # region: ~1300 event pairs per step keep consecutive kernels from overlapping and cost 6.5 ms of a 75 ms step)
    eng.profile(True)
    for _ in range(args.steps):
        one_step()
    eng.synchronize()
    if facade is not None:
        facade.close()
        os.unlink(tmp.name)
    pl, pg, ps = eng.profile_read('lstm_gemm'), eng.profile_read('gemm'), eng.profile_read('lstm_gemm_small')
    pp = eng.profile_read('persist')
    eng.profile(False)
    # calibration, outside the timed region: the same step with the plain whole-sequence contractions handed to the vendor's
    # library (hipBLASLt, option "vendor_gemm"; never the default, never in `value`)
    vendor_default = int(os.environ.get('CASV_OPT_VENDOR_GEMM', '0'))
    calibration = None
    if facade is None and not vendor_default and not os.environ.get('CASV_BENCH_NO_CALIBRATION'):    # (the variable: profiles of the default path alone)
        eng.set_option('vendor_gemm', 1)
        for _ in range(2):
            one_step()
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        eng.synchronize()
        calibration = {'vendor_gemm_ms_per_step': 1e3 * (time.perf_counter() - t1) / args.steps,
                       'note': 'plain whole-sequence contractions through hipBLASLt instead of csrc/gemm.hip; everything else unchanged'}
        eng.set_option('vendor_gemm', 0)
    fl, ms = pl['flops'] + pg['flops'] + ps['flops'] + pp['flops'], pl['ms'] + pg['ms'] + ps['ms'] + pp['ms']
    cpu = None
    if not args.no_cpu_baseline and facade is None:
        # the oracle's train step (numpy fp32: forward, hand-derived BPTT, clip, Adam) on a bounded sample of the same batch, host cores
        from oracle import make_weights as oracle_weights
        from oracle.train import forward_backward, adam_step

        def one_hot(idx):
            out = np.zeros(idx.shape + (VOC,), np.float32)
            b, t = np.nonzero(idx >= 0)
            out[b, t, idx[b, t]] = 1.0
            return out
        ow = oracle_weights(cfg, emb_scale=4.0)

        def cpu_step(n):
            mk = {'enc': masks['enc'], 'dec': masks['dec'], 'cell': masks['cell'][:n]}
            t0 = time.perf_counter()
            _, grads, _ = forward_backward(cfg, ow, one_hot(sidx[:n]), one_hot(dec_in[:n]), one_hot(dec_out[:n]), wts[:n], mk)
            adam_step({k: v.copy() for k, v in ow.items()}, grads, {'t': 0, 'm': {}, 'v': {}})
            return time.perf_counter() - t0
        t1 = cpu_step(2)                                          # warm-up; sizes the sample
        n = int(max(2, min(64, args.cpu_budget / 2.0 / max(t1 / 2, 1e-6))))
        best = min(cpu_step(n) for _ in range(2))
        cpu = {'value': n * LENGTH / best, 'unit': 'chars/s', 'cores': min(host_threads(), host_cores()), 'kind': 'port',
               'sample': 'best of 2 train steps of the numpy fp32 oracle (forward, BPTT, clip, Adam) on the first %d lines of the same batch, %.1f s per step' % (n, best)}
    traffic, traffic_source = None, None        # HBM-side bytes per train step: a committed constant from PMC passes, not measured by this run
    try:
        with open(os.path.join(ROOT, 'profiles', 'c4_step_traffic.json')) as f:
            tj = json.load(f)
        traffic = tj.get('hbm_bytes_per_launch')
        traffic_source = ('profiles/c4_step_traffic.json: constant from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s (commit %s) '
                          'over this command, all kernels of a step, not a measurement of this run' % (tj.get('round', '?'), tj.get('commit', '?')))
    except Exception:
        pass
    emit(json.dumps({
        'metric': 'trained chars/sec (1 GPU), depth-4 width-512 teacher-forced train step, 100-char lines',
        'value': B * LENGTH * args.steps / elapsed, 'unit': 'chars/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[3]: depth=4 width=512 V=256 train step, batch 512 x 100 chars, dropout 0.2, Adam(clipnorm 5)',
                   'batches': 'read from a TSV file and vectorised by the worker thread of train() (training.prefetch)' if args.facade
                              else 'one synthetic batch, resident on the host', 'last_loss': loss, 'last_grad_norm': norm,
                   'vendor_gemm': vendor_default},
        'calibration': calibration, 'cpu_baseline': cpu,
        'roofline': {'bound': 'mfma', 'kernel': 'all GEMMs of the step (incl. the persistent recurrences: %.1f ms in %d launches)' % (pp['ms'] / max(args.steps, 1), pp['launches'] // max(args.steps, 1)), 'achieved': fl / max(ms, 1e-9) / 1e9,
                     'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / max(ms, 1e-9) / 1e9 / PEAK_F32_MFMA_TFLOPS,
                     'traffic': traffic, 'traffic_per': 'train step (all kernels)', 'traffic_source': traffic_source,
                     'launches': pl['launches'] + pg['launches'] + ps['launches'] + pp['launches'],
                     # the whole step priced with SURVEY.md section 8(d)'s ~133 MFLOP per trained character
                     'whole_path': {'flop_per_char': 133e6,
                                    'frac': B * LENGTH * args.steps / elapsed * 133e6 / 1e12 / PEAK_F32_MFMA_TFLOPS}}}))
    return 0


# ------------------------------------------------------------------------------------------------------
WORKLOADS = {
    # lines = lines per GPU per step, batch = lines per decode call, n = hypotheses per line and step (batch_size)
    'c2': dict(depth=2, width=256, voc=VOC, length=LENGTH, lines=256, batch=256, n=1, fast=True, emb=EMB_SCALE, seed=102,
               text='BASELINE configs[1]: depth=2 width=256 V=256 greedy decode (decode_batch_greedy, 2T steps), '
                    '256 lines x 100 chars per GPU per step'),
    'c3': dict(depth=DEPTH, width=WIDTH, voc=VOC, length=LENGTH, lines=LINES, batch=LINES, n=BEAM_N, fast=False, emb=EMB_SCALE, seed=LINE_SEED,
               text='BASELINE configs[2]: depth=4 width=512 V=256 beamed decode (N=8 hypotheses/step, defaults otherwise), '
                    '1024 lines x 100 chars per GPU per step'),
    'c5': dict(depth=DEPTH, width=WIDTH, voc=VOC, length=LENGTH, lines=8192, batch=LINES, n=BEAM_N, fast=False, emb=EMB_SCALE, seed=105,
               text='BASELINE configs[4]: depth=4 width=512 V=256 beamed decode (N=8), 8192 lines x 100 chars per GPU per step '
                    'in 1024-line batches, one all-gather of result records per step'),
    # the OCR-D processor's call (wrapper/transcode.py:110-115; parameter defaults of wrapper/ocrd-tool.json:38-56; the
    # published models are depth 2, width 512, ocrd-tool.json:61-74; `batch_size` keeps its default 256 = hypotheses per step,
    # seq2seq.py:111,1414): one page = one correct_lines call on confusion-network lines, alignments requested
    'page': dict(depth=2, width=512, voc=640, length=60, lines=40, batch=40, n=256, fast=False, emb=EMB_SCALE, seed=106, confmat=True,
                 alignments=True, rejection=0.5, beam_width_in=15, beam_threshold_in=0.2,
                 text='OCR-D processor call (wrapper/transcode.py:110-115, ocrd-tool.json defaults): depth=2 width=512 V=640, one page = '
                      '40 confusion-network lines x 60 positions, beamed decode with batch_size=256 hypotheses/step, fixed_beam_width 15, '
                      'relative_beam_width 0.2, rejection_threshold 0.5, soft alignments returned'),
}


def decode_bench(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # several ranks on one host: each to the CPUs next to its GPU, BLAS / OpenMP threads capped at its share -- before numpy, torch
    # or the GPU library are loaded (one rank keeps the whole host: its cpu_baseline uses all cores)
    placement = place_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world))) if world > 1 else None
    import numpy as np
    wl = WORKLOADS[args.workload]
    L, V = wl['length'], wl['voc']
    # GEMM arithmetic: the library's default is by entry point -- the beam search's decoder steps on
    # bf16x3-split operands (mode 2), everything else fp32-input; --arithmetic fp32 / split puts the handle on one of them,
    # --split-bf16 the whole process.  split_mode = what the workload's dominant launches take.
    arith, beamed = args.arithmetic, not wl['fast']
    if args.split_bf16 >= 0:
        arith = 'split' if args.split_bf16 else 'fp32'
    split_mode = args.split_bf16 if args.split_bf16 >= 0 else {'auto': 2 if beamed else 0, 'fp32': 0, 'split': 2}[arith]
    if world != args.gpus and 'WORLD_SIZE' in os.environ:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        return 2
    dist = None
    torch = None
    backend = os.environ.get('CASV_BENCH_BACKEND', 'nccl')     # 'gloo' + CASV_BENCH_SAME_DEVICE=1: rehearsal of the
    if os.environ.get('CASV_BENCH_SAME_DEVICE'):               # N-rank path on a one-GPU box (all ranks on device 0)
        local_rank = 0
    # CASV_BENCH_DRY_RUN=1: no device at all -- every rank echoes its input lines instead of decoding them.  Rehearses
    # the launcher, sharding, record packing, all-gather and reporting on a CPU-only box; the line says "dry-run" and
    # its value means nothing.
    dry = bool(os.environ.get('CASV_BENCH_DRY_RUN'))
    # CASV_BENCH_FORCE_DIST=1: take the multi-rank path (process group, barrier, all-gather, max-reduce) with ONE rank,
    # to exercise the RCCL calls on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get('CASV_BENCH_FORCE_DIST'))
    # CASV_BENCH_GATHER=native: barrier, all-gather and max-reduce through the C ABI's RCCL leg (casv_comm_*) instead of
    # torch.distributed -- no torch in the process at all
    native = dist_on and os.environ.get('CASV_BENCH_GATHER') == 'native' and not dry
    # Where the result records are packed: 'device' (default on the GPU paths) = by a kernel where the decode results
    # already lie, the collective reads them there (casv_records_*); 'host' = from the returned strings, copied to the
    # device for the collective (the only way for the gloo rehearsal and the dry run)
    records = os.environ.get('CASV_BENCH_RECORDS', 'device')
    if dry or (dist_on and not native and backend != 'nccl') or wl.get('confmat'):
        records = 'host'
    comm = None
    if dist_on and not native:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from cor_asv_ann_amd.synthetic import make_lines, make_confmat_lines, make_vocabulary
    from cor_asv_ann_amd import sharding
    per_gpu, batch = (args.lines_per_gpu or wl['lines']), wl['batch']
    # weak scaling: the global job is world x per_gpu lines, rank r decodes lines [r*per_gpu, (r+1)*per_gpu)
    if wl.get('confmat'):
        all_lines = make_confmat_lines(per_gpu * world, L, wl['seed'], voc_size=V)
    else:
        all_lines, _ = make_lines(per_gpu * world, L, wl['seed'], voc_size=V)
    lo, hi = sharding.shard_bounds(len(all_lines), world, rank)
    lines = all_lines[lo:hi]
    per_rank = -(-len(all_lines) // world)
    S = 2 * (L + 1)
    device = ('cuda:%d' % local_rank) if (dist_on and backend == 'nccl') else None
    want_align = bool(args.alignments) or bool(wl.get('alignments'))
    eng = None
    t_realign = [0.0]
    if dry:
        mapping = make_vocabulary(V)
        lut = np.full(max(ord(c) for c in mapping[0] if c) + 2, -1, np.int32)
        for c, i in mapping[0].items():
            if c:
                lut[ord(c)] = i
        cfg = weights = None

        def decode(chunk):
            return chunk, [[1.0] * len(t) for t in chunk], [0.0] * len(chunk)
        sync_dev = lambda: None
    else:
        s2s, cfg, weights = make_model(local_rank, wl['depth'], wl['width'], wl['n'], wl['emb'], V)
        s2s.arithmetic = args.arithmetic
        for key in ('rejection', 'beam_width_in', 'beam_threshold_in'):
            if key in wl:
                setattr(s2s, 'rejection_threshold' if key == 'rejection' else key, wl[key])
        eng = s2s._require_engine()
        if args.graph:
            eng.set_option('graph', 1)
        for opt in ('persistent', 'tile'):      # A/B switches for experiments: CASV_OPT_<NAME>=value
            if os.environ.get('CASV_OPT_' + opt.upper()):
                eng.set_option(opt, int(os.environ['CASV_OPT_' + opt.upper()]))
        lut = s2s._codepoint_lut()
        if native:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            comm = sharding.NativeComm(eng, rank, world)

        def decode(chunk):
            if wl.get('confmat'):
                out, probs, scores, al = s2s.correct_lines(chunk, conf=chunk, fast=False, greedy=False, alignments=want_align)
                if want_align:
                    # the wrapper's next step on every line: hard path from the soft alignment (transcode.py:126), here on
                    # the windows the device returned
                    from cor_asv_ann_amd.realign import alignment2path
                    t0 = time.perf_counter()
                    for line, text, a in zip(chunk, out, al):
                        if len(a):
                            alignment2path(a, sum(max((len(x[0]) for x in c), default=0) for c in line), len(text), 1. / V)
                    t_realign[0] += time.perf_counter() - t0
                return out, probs, scores
            out, probs, scores, al = s2s.correct_lines(chunk, fast=wl['fast'], greedy=wl['fast'], alignments=want_align)
            return out, probs, scores
        sync_dev = eng.synchronize
    t_gather = [0.0]

    chunks = [lines[b0:b0 + batch] for b0 in range(0, len(lines), batch)]

    def append_records(k):                      # in the device thread, right behind batch k's decode call: a small kernel
        if dist_on and records == 'device':     # on the same stream packs its results where they lie
            if eng.B != len(chunks[k]):         # (correct_lines decodes in several chunks only under a memory budget)
                raise RuntimeError('the last decode call covered %d of %d lines: records must be appended per decode call' % (eng.B, len(chunks[k])))
            eng.records_append(k * batch)

    def step():
        out_lines, probs, scores = [], [], []
        if dist_on and records == 'device':
            eng.records_reset(per_rank, S)
        if dry or wl.get('confmat') or len(chunks) == 1:
            for k, chunk in enumerate(chunks):
                o, p, s = decode(chunk)
                append_records(k)
                out_lines += o; probs += p; scores += s
        else:
            # several batches per step (configs[4]): batch k + 1 is vectorised and decoded while the strings of batch k are built
            for o, p, s, _ in s2s.correct_batches(chunks, fast=wl['fast'], greedy=wl['fast'], alignments=want_align, after_decode=append_records):
                out_lines += o; probs += p; scores += s
        if dist_on:
            # fixed-width records (characters, probabilities, length, score) -> RCCL all-gather
            t0 = time.perf_counter()
            if records == 'device':
                got = comm.all_gather_device_records(len(all_lines)) if comm else \
                    sharding.all_gather_device_records(eng, len(all_lines), device)
            else:
                rec = sharding.records_from_lines(out_lines, probs, scores, lut, S)
                got = comm.all_gather_records(rec, len(all_lines)) if comm else sharding.all_gather_records(rec, len(all_lines), device=device)
            t_gather[0] += time.perf_counter() - t0
            return got
        return out_lines

    def sync():
        sync_dev()
        if comm:
            comm.max(0.0)
        elif dist_on:
            if backend == 'nccl':
                torch.cuda.synchronize()
            dist.barrier()
            if backend == 'nccl':
                torch.cuda.synchronize()

    # One GPU: the steps are independent of each other, and they run the way `Sequence2Sequence.predict()` runs its batches --
    # through `correct_batches`, where batch k + 1 is vectorised while batch k is on the device and the strings of batch k - 1
    # are built (--pipeline 0: one `correct_lines` call after the other).  All K steps lie inside the timed region either way.
    pipelined = bool(args.pipeline) and eng is not None and not dist_on and not dry

    def run_steps(n):
        if not pipelined:
            last = None
            for _ in range(n):
                last = step()
            return last
        confmat = bool(wl.get('confmat'))
        fast = False if confmat else wl['fast']

        def batches():
            for _ in range(n):
                for chunk in chunks:
                    yield (chunk, chunk) if confmat else chunk
        out_lines, last = [], None
        for k, (o, p, sc, al) in enumerate(s2s.correct_batches(batches(), fast=fast, greedy=fast, alignments=want_align)):
            if confmat and want_align:
                from cor_asv_ann_amd.realign import alignment2path
                t1 = time.perf_counter()
                for line, text, a in zip(chunks[k % len(chunks)], o, al):
                    if len(a):
                        alignment2path(a, sum(max((len(x[0]) for x in c), default=0) for c in line), len(text), 1. / V)
                t_realign[0] += time.perf_counter() - t1
            out_lines += o
            if (k + 1) % len(chunks) == 0:
                last, out_lines = out_lines, []
        return last

    dom = 'lstm_gemm' if (args.workload != 'c2' or split_mode) else 'persist'
    run_steps(args.warmup)
    if eng:
        # HIP events around launches of the dominant kernel, on the library's stream: around every 13th of them (level 3) --
        # an event pair keeps the next launch from overlapping the kernel's tail, which costs ~8 us per pair: around every
        # launch (level 2) the timed region of c3 is 2 % slower than without events
        eng.profile(int(os.environ.get('CASV_BENCH_PROFILE', '2' if dom == 'persist' else '3')))
    t_gather[0] = 0.0
    t_realign[0] = 0.0
    sync()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    sync_dev()
    mine = time.perf_counter() - t0                       # this rank's own time, before it waits for the others
    sync()
    elapsed = time.perf_counter() - t0
    realign_ms = 1e3 * t_realign[0] / max(args.steps, 1)
    prof, others = None, {}
    if eng:
        prof = eng.profile_read(dom)
        eng.profile(1)                 # one extra, untimed step with events around every kernel class
        run_steps(1)
        sync()
        others = {k: eng.profile_read(k) for k in ('lstm_gemm', 'lstm_gemm_small', 'gemm', 'attention', 'softmax', 'beam', 'embed', 'persist')}
        eng.profile(False)
    per_rank_s = [mine]
    gather_ms = 1e3 * t_gather[0] / max(args.steps, 1)
    placements = [placement]
    if comm:
        elapsed = comm.max(elapsed)
        # every rank's own time and placement: one more all-gather of a small fixed-width record
        report = comm.all_gather_objects({'s': mine, 'placement': placement})
        per_rank_s = [r['s'] for r in report]
        placements = [r['placement'] for r in report]
    elif dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tt = torch.zeros(world, dtype=torch.float64, device=device or 'cpu')
        tt[rank] = mine
        dist.all_reduce(tt)
        per_rank_s = [float(x) for x in tt.cpu()]
        placements = [None] * world
        dist.all_gather_object(placements, placement)

    result = None
    if rank == 0:
        chars = len(all_lines) * L * args.steps
        fpc = survey_flop_per_char(wl['depth'], wl['width'], wl['n'], V, L)
        qpc = survey_hbm_bytes_per_char(wl['depth'], wl['width'], wl['n'], V, L)
        xpc = executed_flop_per_char(wl['depth'], wl['width'], wl['n'], V, L)
        result = {
            'metric': METRIC if args.workload in ('c3', 'c5') else
                      ('corrected chars/sec (1 GPU) greedy, depth-2 width-256, 100-char lines' if args.workload == 'c2' else
                       'corrected chars/sec (1 GPU), OCR-D processor call: depth-2 width-512 V=640, beamed N=256, 40-line pages'),
            'value': chars / elapsed, 'unit': 'chars/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype_label(arith, beamed),
            'data': 'dry-run (no decoding)' if dry else 'synthetic',
            'config': {'workload': wl['text'] + ', 2T=%d steps max' % S,
                       # deviation from BASELINE.md section 3 (emb_scale 4): with 4 the softmax is flat and the search collapses to one
                       # row x T steps per line, 16 x less work
                       'emb_scale': wl['emb'], 'weights': 'synthetic, seed 20250614',
                       'arithmetic': arith, 'lines_per_gpu': per_gpu, 'lines_per_decode_call': batch, 'line_length': L, 'beam_n': wl['n'],
                       'parallelism': 'lines sharded x%d' % world,
                       'gather': 'casv_comm (RCCL, C ABI)' if comm else ('torch.distributed/' + backend if dist_on else 'none'),
                       'records': (records + '-packed') if dist_on else 'none',
                       'graph': bool(args.graph), 'alignments': want_align, 'split_bf16_override': args.split_bf16,
                       'steps_run': 'through correct_batches, as predict() does: host work of neighbouring steps overlaps the device'
                                    if pipelined else 'one correct_lines call after the other',
                       'launcher': 'bench.py' if os.environ.get('CASV_BENCH_CHILD') else
                                   ('torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ else 'direct')},
            'ms_per_step_by_rank': [1e3 * x / args.steps for x in per_rank_s],
            'gather_ms_per_step': gather_ms if dist_on else 0.0,
        }
        if world > 1:
            # where each rank's host side ran: CPU list (NUMA node of its GPU where the host says which) and BLAS / OpenMP threads
            result['config']['host_placement_by_rank'] = placements
        if wl.get('confmat') and want_align:
            result['realign_ms_per_step'] = realign_ms       # inside ms_per_step: host time of the Viterbi re-alignment
        if dist_on:
            result['gathered_records'] = int(last.shape[0])
            if args.dump_records:
                np.save(args.dump_records, np.asarray(last, np.int32))
        if prof is not None and prof['launches']:
            achieved = prof['flops'] / max(prof['ms'], 1e-9) / 1e9            # TFLOP/s
            # HBM-side bytes per launch of the dominant kernel: NOT measured by this run -- a constant from the committed rocprofv3
            # PMC passes over this same command (profiles/pmc_traffic.py; FETCH_SIZE / WRITE_SIZE in separate passes), named here
            traffic, traffic_source = None, None
            tname = {'persist': 'persist_decode_traffic.json'}.get(dom) or \
                (('page_split_gemm_traffic.json' if split_mode == 2 else None if split_mode else 'page_gemm_traffic.json') if args.workload == 'page' else
                 ('split256_gemm_traffic.json' if split_mode == 2 else (None if split_mode == 1 else 'lstm_gemm_traffic.json')))
            try:
                with open(os.path.join(ROOT, 'profiles', tname or 'none')) as f:
                    tj = json.load(f)
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_source = ('profiles/%s: constant from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s (commit %s) over this '
                                  'command, not a measurement of this run' % (tname, tj.get('round', '?'), tj.get('commit', '?')))
            except Exception:
                pass
            result['roofline'] = {
                'bound': 'mfma',
                'kernel': ('gemm_kernel<EPI_LSTM, 1> (fused LSTM-cell GEMM, 128x128 tiles, fp32 MFMA)' if not split_mode else
                           'fused LSTM-cell GEMM on v_mfma_f32_32x32x16_bf16, bf16x3-split operands: ' +
                           ('gemm_kernel<EPI_LSTM, 1, split> 128x128 tiles' if split_mode == 1 else 'gemm_split256_kernel 256x256 tiles') +
                           '; achieved / peak / frac are the EXECUTED bf16 FLOP (6 per algorithmic fp32 one) against the dense bf16-MFMA peak')
                          if dom == 'lstm_gemm' else 'persist_decode_kernel (all 2T greedy steps of the batch in one launch: 16x16x4 fp32-MFMA tiles, '
                               'row-block hand-offs between workgroups)',
                'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                'launches': prof['launches'], 'avg_launch_us': 1e3 * prof['ms'] / max(prof['launches'], 1),
                'flops_per_launch': prof['flops'] / max(prof['launches'], 1),
                # the whole path priced with SURVEY.md section 8(d)'s algorithmic FLOP per corrected character
                'whole_path': {'flop_per_char': fpc,
                               'achieved': chars / elapsed * fpc / 1e12 / world,
                               'frac': chars / elapsed * fpc / 1e12 / world / PEAK_F32_MFMA_TFLOPS,
                               # ... and with the FLOP the kernels execute (embedding folded into decoder layer 1: fewer)
                               'flop_per_char_executed': xpc,
                               'frac_executed': chars / elapsed * xpc / 1e12 / world / PEAK_F32_MFMA_TFLOPS,
                               # the other roofline of SURVEY 8(d): not the binding one at fp32
                               'hbm_bytes_per_char': qpc,
                               'hbm_frac': chars / elapsed * qpc / world / PEAK_HBM_BYTES_PER_S}}
        if split_mode and dom == 'lstm_gemm' and 'roofline' in result:
            # The split kernels run on the bf16 matrix instruction: six bf16 products per algorithmic fp32 one.  Their roofline is the
            # dense bf16-MFMA peak; the algorithmic fp32 rate stays in the line as a rate, not as a fraction of a peak it is not bound by.
            rl = result['roofline']
            rl['algorithmic_fp32_tflops'] = rl['achieved']
            rl['achieved'], rl['peak'] = 6.0 * rl['achieved'], PEAK_BF16_MFMA_TFLOPS
            rl['frac'] = rl['achieved'] / PEAK_BF16_MFMA_TFLOPS
            rl['bf16_mfma'] = {'achieved': rl['achieved'], 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': rl['frac']}
            for k in ('frac', 'frac_executed'):        # (whole-path prices against the fp32 peak mean nothing here either)
                rl['whole_path'].pop(k, None)
            rl['whole_path']['note'] = ('achieved = algorithmic fp32 TFLOP/s of the whole path; no fraction of the fp32-input peak: the decoder steps '
                                        'do not run on that instruction.  frac_bf16 = the FLOP the path executes, the split launches\' six-fold, against the bf16 peak')
            # the whole path against the bf16 peak: decoder FLOP executed six-fold on the bf16 instruction; the encoder's share (fp32-input
            # instruction under 'auto') is priced as if it ran there too -- an upper bound on what the peak allows
            rl['whole_path']['frac_bf16'] = 6.0 * chars / elapsed * xpc / 1e12 / world / PEAK_BF16_MFMA_TFLOPS
        if others:
            result['kernel_ms_per_step'] = {k: v['ms'] for k, v in others.items()}     # from one extra untimed step
        if world == 1 and not args.no_cpu_baseline and not dry:
            beam = {('rejection_threshold' if k == 'rejection' else k): wl[k] for k in ('rejection', 'beam_width_in', 'beam_threshold_in') if k in wl}
            try:
                result['cpu_baseline'] = cpu_baseline(cfg, wl['emb'], all_lines[:256], wl['n'] if not wl['fast'] else 256, wl['fast'], L,
                                                      budget_s=args.cpu_budget, confmat=bool(wl.get('confmat')),
                                                      min_lines=1 if wl.get('confmat') else 16, **beam)
            except Exception as err:        # (a reported baseline: its failure must not take the measured line down)
                result['cpu_baseline'] = {'value': None, 'unit': 'chars/s', 'cores': host_cores(), 'kind': 'port', 'sample': 'failed: %r' % (err,)}
    if comm:
        comm.close()
    elif dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if eng is not None:
        s2s.engine.close()              # free the device memory before the other workloads run as child processes
        s2s.engine = None
    if rank == 0:
        if world == 1 and not dist_on and args.workload == 'c3' and not args.no_others and not dry:
            result['other_workloads'] = other_workloads(not args.no_cpu_baseline)
        emit(json.dumps(result))
    return 0


def other_workloads(with_cpu_baseline=True):
    """The other single-GPU workloads, each as a child process of this (finished) run so that the driver's one default
    invocation times them too: configs[1] (c2), configs[3] (c4) and the OCR-D processor's call (page).  Bounded: a few
    steps each, no CPU baseline; a workload that fails reports its error instead of failing the headline."""
    out = {}
    for name, extra in (('c2', ['--steps', '20', '--warmup', '3']), ('c4', ['--steps', '5', '--warmup', '2']),
                        ('page', ['--steps', '3', '--warmup', '1']),
                        # the headline's workload with every launch on the fp32-input matrix instruction (the library's only arithmetic
                        # until round 5; Sequence2Sequence.arithmetic = 'fp32'): the figure beside the default's
                        ('c3_fp32', ['--steps', '5', '--warmup', '2', '--arithmetic', 'fp32'])):
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', name.split('_')[0], '--no-others'] + extra
        cmd += ['--cpu-budget', '8'] if name in ('c2', 'c4', 'page') and with_cpu_baseline else ['--no-cpu-baseline']
        env = {k: v for k, v in os.environ.items() if not k.startswith('CASV_BENCH_')}
        try:
            t0 = time.perf_counter()
            proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            line = [x for x in proc.stdout.decode().splitlines() if x.startswith('{')]
            if proc.returncode != 0 or not line:
                out[name] = {'error': 'exit code %d: %s' % (proc.returncode, proc.stderr.decode()[-300:])}
                continue
            r = json.loads(line[-1])
            keep = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype') if k in r}
            keep['workload'] = r['config']['workload']
            if 'roofline' in r:
                keep['roofline'] = {k: r['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'launches', 'avg_launch_us', 'flops_per_launch', 'algorithmic_fp32_tflops')
                                    if k in r['roofline']}
                for k in ('frac', 'frac_bf16'):
                    if k in r['roofline'].get('whole_path', {}):
                        keep['roofline']['whole_path_' + k] = r['roofline']['whole_path'][k]
                if 'bf16_mfma' in r['roofline']:
                    keep['roofline']['bf16_mfma'] = r['roofline']['bf16_mfma']
            for k in ('kernel_ms_per_step', 'realign_ms_per_step', 'cpu_baseline', 'calibration'):
                if k in r:
                    keep[k] = r[k]
            if name == 'c4':
                keep['vendor_gemm'] = r['config'].get('vendor_gemm')
            keep['arithmetic'] = r['config'].get('arithmetic')
            keep['wall_s'] = time.perf_counter() - t0
            out[name] = keep
        except Exception as err:            # a measurement aid must not take the headline down
            out[name] = {'error': repr(err)}
    return out


_REAL_STDOUT = None


def claim_stdout():
    """Keep this process's stdout for the ONE JSON line: file descriptor 1 is pointed at stderr for everything else (RCCL prints
    a version banner to stdout when a communicator is created, libraries may print notices), the line itself goes to a private
    duplicate of the original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, (line + '\n').encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=60.0, help='seconds of host time for the cpu_baseline sample')
    ap.add_argument('--no-others', action='store_true', help='do not time c2 / c4 / page after the default c3 run')
    ap.add_argument('--pipeline', type=int, default=1, help='1 GPU: 1 = the steps run through correct_batches (vectorising, device and '
                    'string building of neighbouring steps overlap, as in predict()), 0 = one correct_lines call after the other')
    ap.add_argument('--graph', type=int, default=0, help='replay the decode step from a hipGraph')
    ap.add_argument('--workload', default=None, choices=['c2', 'c3', 'c4', 'c5', 'page'],
                    help='c3 = beamed decode (the BASELINE metric; default on one GPU); c5 = 8192 lines per GPU per step '
                         '(configs[4]; default with --gpus > 1); c2 = greedy decode (configs[1]); c4 = train step (configs[3]); '
                         'page = the OCR-D processor call (40 confusion-network lines, 256 hypotheses per step, alignments)')
    ap.add_argument('--lines-per-gpu', type=int, default=0, help='override the lines each GPU decodes per step')
    ap.add_argument('--alignments', type=int, default=0,
                    help='1 = also return the soft alignments (window form), as the OCR-D processor asks for (wrapper/transcode.py:110-115)')
    ap.add_argument('--facade', type=int, default=0,
                    help='c4 only: 1 = every batch comes the way Sequence2Sequence.train() gets it (file -> lines -> index arrays, '
                         'degradation, dropout masks), prepared by train()\'s worker thread while the device runs the step before')
    ap.add_argument('--arithmetic', default='auto', choices=['auto', 'fp32', 'split'],
                    help='GEMM arithmetic of the model handle (Sequence2Sequence.arithmetic): auto = the library default, by entry point '
                         '(beam-search decoder steps: bf16x3-split fp32 operands on the bf16 matrix instruction, fp32 accumulation; encoder, '
                         'greedy decodes: the fp32-input instruction); fp32 / split = one of them for everything the handle launches')
    ap.add_argument('--split-bf16', type=int, default=-1, choices=[-1, 0, 1, 2],
                    help='A/B measurements: process-wide override of every launch of the decode path (library option split_bf16; 0 = fp32-input, '
                         '1 = split operands on 128x128 tiles, 2 = on 256x256 tiles where they fill the chip)')
    ap.add_argument('--dump-records', default=None, help='rank 0 saves the gathered records of the last step to this .npy file (tests)')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be positive')
    if args.workload is None:
        # BASELINE configs[4] is what the multi-GPU metric is quoted on: 64k lines over 8 GPUs = 8192 per GPU per step
        args.workload = 'c5' if (args.gpus > 1 or int(os.environ.get('WORLD_SIZE', '1')) > 1) else 'c3'
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        if args.workload in ('c4', 'page'):
            ap.error('the train step (c4) and the page call are single-GPU workloads')
        os.environ['CASV_BENCH_CHILD'] = '1'
        argv = sys.argv[1:]
        if '--workload' not in argv:
            argv = argv + ['--workload', args.workload]
        return launch_ranks(args.gpus, argv)
    claim_stdout()
    if args.split_bf16 >= 0:
        os.environ['CASV_SPLIT_BF16'] = str(args.split_bf16)       # read by the library when it is loaded (process-wide override)
    if args.workload == 'c4':
        return train_bench(args)
    return decode_bench(args)


if __name__ == '__main__':
    sys.exit(main())
