#!/usr/bin/env python3
"""Benchmark of the hot path on the metric of BASELINE.json:
corrected chars/sec (whole node) at beam=8, depth-4 width-512, 100-char lines.

One "step" = one pass of `Sequence2Sequence.correct_lines(..., fast=False, greedy=False)` (vectorise ->
encode -> beamed decode -> strings) over one batch of 1024 synthetic 100-character lines per GPU
(BASELINE.json configs[2]).  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
decodes its own 1024 lines per step -- lines are independent, so the path shards with no data-path
collective -- and one RCCL all-gather of the result records makes all decoded lines available on all
ranks (BASELINE.json configs[4], weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

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
PEAK_HBM_BYTES_PER_S = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def make_model(device):
    from cor_asv_ann_amd.synthetic import ModelConfig, make_weights, make_vocabulary
    from cor_asv_ann_amd.seq2seq import Sequence2Sequence
    cfg = ModelConfig(depth=DEPTH, width=WIDTH, voc_size=VOC)
    weights = make_weights(cfg, emb_scale=EMB_SCALE)
    import logging
    logger = logging.getLogger('bench')
    logger.setLevel(logging.CRITICAL)     # lines without a finished hypothesis fall back to the input (seq2seq.py:826-836)
    s2s = Sequence2Sequence(logger=logger, device=device)   # and are logged as errors: hundreds per step with random weights
    s2s.depth, s2s.width, s2s.batch_size = DEPTH, WIDTH, BEAM_N
    s2s.mapping, s2s.voc_size = make_vocabulary(VOC), VOC
    s2s.configure()
    s2s.set_weights(weights)
    s2s.status = 2
    return s2s, cfg, weights


def survey_flop_per_char():
    """F = F_enc + N*S*F_row per line, divided by the L corrected characters."""
    W, V, d, K, T, N = WIDTH, VOC, DEPTH, 11, LENGTH + 1, BEAM_N
    C = 2 * W if d == 1 else W
    f_row = 2 * V * W + (d - 1) * 16 * W * W + 2 * W * W + K * (4 * W + 2 * C) + 8 * W * (2 * W + C) + 2 * W * V
    f_enc = T * (32 * W * W + (24 * W * W if d >= 2 else 0) + 16 * W * W * max(d - 2, 0) + 2 * C * W)
    return (f_enc + N * 2 * T * f_row) / float(LENGTH)


def survey_hbm_bytes_per_char():
    """Q = Q_enc + N*S*Q_row per line (per-beam state in HBM, weights on chip), per corrected character."""
    W, V, d, K, T, N = WIDTH, VOC, DEPTH, 11, LENGTH + 1, BEAM_N
    C = 2 * W if d == 1 else W
    q_row = 4 * (4 * d * W + K * (W + C) + 2 * V + 2 * T)
    q_enc = 4 * T * (1 + 2 * 2 * W + 2 * W * max(d - 2, 0) + C + W)
    return (q_enc + N * 2 * T * q_row) / float(LENGTH)


def cpu_baseline(cfg, weights, lines, budget_s=20.0):
    """The oracle in the reference's dataflow (per-character decoder call, dense-T attention, u recomputed
    every step, per-line best-first search) on the host cores, on as many lines of the same workload as
    fit the time budget."""
    from oracle.decode import OracleModel, correct_lines
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    om = OracleModel(cfg, weights, batch_size=BEAM_N, recompute_u=True)
    correct_lines(om, lines[:1], fast=False, greedy=False)      # warm-up (BLAS threads, page-in)
    t0 = time.perf_counter()
    n = 0
    while n < len(lines):
        correct_lines(om, lines[n:n + 2], fast=False, greedy=False)
        n += 2
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {'value': n * LENGTH / dt, 'unit': 'chars/s', 'cores': int(cores), 'kind': 'port',
            'sample': '%d lines of the same workload (numpy fp32 oracle, reference dataflow), %.1f s' % (n, dt)}


def train_bench(args):
    """BASELINE configs[3]: depth 4, width 512, teacher-forced train step (forward + backward + clip + Adam) on 512
    lines of 100 characters (targets = sources with 5 % substitutions), dropout 0.2.  One GPU."""
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
    eng.train_begin()
    for _ in range(args.warmup):
        eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=1)
    eng.profile(True)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, norm = eng.train_step(sidx, None, dec_in, dec_out, wts, masks, mode=1)
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    pl, pg, ps = eng.profile_read('lstm_gemm'), eng.profile_read('gemm'), eng.profile_read('lstm_gemm_small')
    eng.profile(False)
    fl, ms = pl['flops'] + pg['flops'] + ps['flops'], pl['ms'] + pg['ms'] + ps['ms']
    print(json.dumps({
        'metric': 'trained chars/sec (1 GPU), depth-4 width-512 teacher-forced train step, 100-char lines',
        'value': B * LENGTH * args.steps / elapsed, 'unit': 'chars/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[3]: depth=4 width=512 V=256 train step, batch 512 x 100 chars, dropout 0.2, Adam(clipnorm 5)',
                   'last_loss': loss, 'last_grad_norm': norm},
        'roofline': {'bound': 'mfma', 'kernel': 'gemm_kernel (all GEMMs of the step)', 'achieved': fl / max(ms, 1e-9) / 1e9,
                     'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / max(ms, 1e-9) / 1e9 / PEAK_F32_MFMA_TFLOPS,
                     'traffic': None, 'launches': pl['launches'] + pg['launches'] + ps['launches']}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', type=int, default=0, help='replay the decode step from a hipGraph')
    ap.add_argument('--workload', default='c3', choices=['c3', 'c4'],
                    help='c3 = beamed decode (the BASELINE metric, default); c4 = train step (BASELINE configs[3])')
    args = ap.parse_args()

    if args.workload == 'c4':
        return train_bench(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    torch = None
    backend = os.environ.get('CASV_BENCH_BACKEND', 'nccl')     # 'gloo' + CASV_BENCH_SAME_DEVICE=1: rehearsal of the
    if os.environ.get('CASV_BENCH_SAME_DEVICE'):               # N-rank path on a one-GPU box (all ranks on device 0)
        local_rank = 0
    # CASV_BENCH_FORCE_DIST=1: take the multi-rank path (process group, barrier, all-gather, max-reduce) with ONE rank,
    # to exercise the RCCL calls on a one-GPU box (launch under torch.distributed.run --nproc-per-node 1)
    dist_on = world > 1 or bool(os.environ.get('CASV_BENCH_FORCE_DIST'))
    if dist_on:
        import torch
        import torch.distributed as dist
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from cor_asv_ann_amd.synthetic import make_lines
    from cor_asv_ann_amd import sharding
    s2s, cfg, weights = make_model(local_rank)
    # weak scaling: the global job is world x 1024 lines, rank r decodes lines [r*1024, (r+1)*1024)
    all_lines, _ = make_lines(LINES * world, LENGTH, LINE_SEED, voc_size=VOC)
    lo, hi = sharding.shard_bounds(len(all_lines), world, rank)
    lines = all_lines[lo:hi]
    eng = s2s._require_engine()
    if args.graph:
        eng.set_option('graph', 1)
    S = 2 * (LENGTH + 1)
    device = ('cuda:%d' % local_rank) if (dist_on and backend == 'nccl') else None

    def step():
        out_lines, probs, scores, _ = s2s.correct_lines(lines, fast=False, greedy=False, alignments=False)
        if dist_on:
            # fixed-width records (characters, probabilities, length, score) -> RCCL all-gather
            rec = sharding.records_from_lines(out_lines, probs, scores, s2s._codepoint_lut(), S)
            return sharding.all_gather_records(rec, len(all_lines), device=device)
        return out_lines

    def sync():
        eng.synchronize()
        if dist_on:
            if backend == 'nccl':
                torch.cuda.synchronize()
            dist.barrier()
            if backend == 'nccl':
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile(2)                 # HIP events around every launch of the dominant kernel, on the library's stream
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read('lstm_gemm')
    eng.profile(1)                 # one extra, untimed step with events around every kernel class
    step()
    sync()
    others = {k: eng.profile_read(k) for k in ('lstm_gemm', 'lstm_gemm_small', 'gemm', 'attention', 'softmax', 'beam', 'embed')}
    eng.profile(False)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        chars = len(all_lines) * LENGTH * args.steps
        achieved = prof['flops'] / max(prof['ms'], 1e-9) / 1e9            # TFLOP/s
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'lstm_gemm_traffic.json')) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
        except Exception:
            pass
        result = {
            'metric': 'corrected chars/sec (whole node) at beam=8, depth-4 width-512, 100-char lines',
            'value': chars / elapsed, 'unit': 'chars/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: depth=4 width=512 V=256 beamed decode (N=8 hypotheses/step, '
                                   'defaults otherwise), %d lines x %d chars per GPU per step, 2T=%d search iterations max, '
                                   'synthetic weights seed 20250614 emb_scale=%g' % (LINES, LENGTH, S, EMB_SCALE),
                       'lines_per_gpu': LINES, 'line_length': LENGTH, 'beam_n': BEAM_N, 'parallelism': 'lines sharded x%d' % world,
                       'graph': bool(args.graph)},
            'roofline': {'bound': 'mfma', 'kernel': 'gemm_kernel<EPI_LSTM, 1> (fused LSTM-cell GEMM, 128x128 tiles, fp32 MFMA)',
                         'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'launches': prof['launches'], 'avg_launch_us': 1e3 * prof['ms'] / max(prof['launches'], 1),
                         'flops_per_launch': prof['flops'] / max(prof['launches'], 1),
                         # the whole path priced with SURVEY.md section 8(d)'s algorithmic FLOP per corrected character
                         'whole_path': {'flop_per_char': survey_flop_per_char(),
                                        'achieved': chars / elapsed * survey_flop_per_char() / 1e12 / world,
                                        'frac': chars / elapsed * survey_flop_per_char() / 1e12 / world / PEAK_F32_MFMA_TFLOPS,
                                        # the other roofline of SURVEY 8(d): not the binding one at fp32 (AI = 262 FLOP/B)
                                        'hbm_bytes_per_char': survey_hbm_bytes_per_char(),
                                        'hbm_frac': chars / elapsed * survey_hbm_bytes_per_char() / world / PEAK_HBM_BYTES_PER_S}},
            'kernel_ms_per_step': {k: v['ms'] for k, v in others.items()},     # from one extra untimed step
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(cfg, weights, all_lines[:64])
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
