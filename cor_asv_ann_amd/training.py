"""Host side of `Sequence2Sequence.train()` (seq2seq.py:590-649 + lib/keras_train.py:27-438): epochs over the
generator, validation, early stopping / NaN termination / per-epoch checkpoints.  Each batch is ONE call into
the C ABI (`casv_train_step`), which runs forward, backward, clipping and Adam on the device."""
import queue
import signal
import threading

import numpy as np

from . import keras_h5


def prefetch(iterable, depth=2, cancel=None, in_call=None, detach_after=None):
    """Run `iterable` in a worker thread, `depth` items ahead of the consumer -- the reference feeds `train_on_batch` from a
    `GeneratorEnqueuer` worker (keras_train.py:133-145) so that vectorising the next batch overlaps the device step; here the
    C ABI call releases the GIL for the whole step, so a thread does.  Exceptions of the producer surface in the consumer;
    a consumer that stops early (NaN loss, stop signal) releases the producer.

    Leaving early: the worker is joined, however long its current `next(iterable)` takes (the default: train()'s producers draw
    from the model's random generator and read its state -- a worker left behind would keep doing so beside whatever the caller
    does next).  Stages whose producer may block for good on something outside this process (correct_batches: a user generator
    reading a pipe, a nested stage) pass `detach_after`: such a worker is left behind after that many seconds (a daemon thread; its
    next put() sees `stop`) -- except while `in_call` (an Event the producer sets around its C-ABI calls) is set: the caller must
    not get the engine back while a call runs on it, the handle is not thread-safe.  `cancel`: an Event of the caller that ends
    the consumer loop as well (a nested stage's consumer is another stage's worker: it must not sit in q.get() for good)."""
    q = queue.Queue(maxsize=max(1, depth))
    done, stop = object(), threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for item in iterable:
                if not put(('item', item)):
                    return
        except BaseException as err:            # handed to the consumer
            put(('error', err))
            return
        put(('done', done))

    thread = threading.Thread(target=work, name='casv-batch-prefetch', daemon=True)
    thread.start()
    try:
        while True:
            if cancel is None:
                kind, item = q.get()
            else:
                try:
                    kind, item = q.get(timeout=0.1)
                except queue.Empty:
                    if cancel.is_set():
                        return
                    continue
            if kind == 'done':
                return
            if kind == 'error':
                raise item
            yield item
    finally:
        stop.set()
        waited = 0.0
        while thread.is_alive():
            thread.join(0.1)
            if in_call is not None and in_call.is_set():
                continue                        # a device call of the worker is in flight: wait it out, however long it takes
            waited += 0.1
            if detach_after is not None and waited >= detach_after:
                break


def train_batches(s2s, filenames, split_rand, rng):
    """One epoch of training batches in the form `casv_train_step` takes: index arrays of vectorize_lines, the random
    degradation of seq2seq.py:909-915, the dropout keep-masks (drawn in this order per batch)."""
    for batch in s2s.gen_lines(filenames, True, split_rand, True):
        if not batch:
            return                                 # end of epoch (kt:160-162)
        src, conf, tgt, _ = batch
        idx, val, dec_in, dec_out, w = batch_to_indices(s2s, src, tgt, conf)
        idx, val = degrade(idx, val, rng)
        yield idx, val, dec_in, dec_out, w, dropout_masks(s2s, len(src), rng)


def validation_batches(s2s, filenames, split_rand):
    for batch in s2s.gen_lines(filenames, True, split_rand, False):
        if not batch:
            return
        src, conf, tgt, _ = batch
        yield batch_to_indices(s2s, src, tgt, conf)


def batch_to_indices(s2s, lines_source, lines_target, lines_conf):
    """The arrays of vectorize_lines (seq2seq.py:1020-1119) in index form: encoder (idx, val) (B,T,A),
    decoder input / target character indices (B,U) with -1 for true-zero rows, temporal weights (B,U)."""
    idx, val, _ = s2s._sparse_lines(lines_source, lines_conf)
    B = len(lines_target)
    U = max(map(len, lines_target)) + 1
    dec_in = np.full((B, U), -1, np.int32)
    dec_out = np.full((B, U), -1, np.int32)
    for i, line in enumerate(lines_target):
        codes = [s2s._index(c, 'decoder input', i) for c in line]
        dec_in[i, 1:len(codes) + 1] = codes
        dec_out[i, :len(codes)] = codes
    weights = (dec_out >= 0).astype(np.float32)
    return idx, val, dec_in, dec_out, weights


def degrade(idx, val, rng):
    """Random degradation of one encoder position per line to index 0 for learning underspecification
    (seq2seq.py:909-915): position = int(T * U(0,1) / 0.01), applied when it falls inside the line."""
    B, T = idx.shape[:2]
    pos = (T * rng.uniform(0, 1, B) / 0.01).astype(int)
    for b in np.nonzero(pos < T)[0]:
        idx[b, pos[b], :] = -1
        idx[b, pos[b], 0] = 0
        val[b, pos[b], :] = 0
        val[b, pos[b], 0] = 1
    return idx, val


def dropout_masks(s2s, B, rng):
    """Keep-masks of the reference's dropout layers, scaled by 1/(1-rate): time-constant feature masks after
    every encoder layer and every hidden decoder layer (noise_shape (1, F), seq2seq.py:293-298,363-367), a
    per-sample mask on the attention cell's input (LSTMCell(dropout), seq2seq.py:345)."""
    rate = float(s2s.dropout or 0.0)
    if rate <= 0.0:
        return None
    W, d = s2s.width, s2s.depth
    deep = bool(getattr(s2s, 'deep_bidirectional_encoder', False))      # every encoder layer 2W wide (seq2seq.py:292-295)
    C = 2 * W if (d == 1 or deep) else W
    keep = lambda shape: ((rng.uniform(0, 1, shape) >= rate) / (1.0 - rate)).astype(np.float32)
    return {'enc': [keep(2 * W if (n == 0 or deep) else W) for n in range(d)], 'dec': [keep(W) for _ in range(d - 1)],
            'cell': keep((B, W + C))}


def train_files(s2s, filenames, val_filenames=None):
    num_lines = s2s.map_files(filenames)
    s2s.logger.info('Training on "%d" files with %d lines', len(filenames), num_lines)
    if val_filenames:
        num_lines = s2s.map_files(val_filenames)
        s2s.logger.info('Validating on "%d" files with %d lines', len(val_filenames), num_lines)
        split_rand = None
    else:
        s2s.logger.info('Validating on random 20% lines from those files')
        split_rand = s2s._rng.uniform(0, 1, (num_lines,))
    rng = s2s._rng
    engine = s2s._require_engine()
    engine.train_begin(frozen=tuple(s2s.frozen_prefixes))
    stop = {'flag': False}
    old_handler = None
    try:
        old_handler = signal.signal(signal.SIGINT, lambda *a: stop.__setitem__('flag', True))   # StopSignalCallback
    except ValueError:
        pass                                           # not in the main thread
    history = []
    best, best_weights, wait = np.inf, None, 0
    try:
        for epoch in range(s2s.epochs):
            total, nb = 0.0, 0
            nan = False
            # the next batches are vectorised by a worker thread while the device runs this one (kt:133-145)
            for idx, val, dec_in, dec_out, w, masks in prefetch(train_batches(s2s, filenames, split_rand, rng)):
                loss, _ = engine.train_step(idx, val, dec_in, dec_out, w, masks, mode=1)
                if not np.isfinite(loss):
                    s2s.logger.warning('Batch %d: Invalid loss, terminating training', nb)   # TerminateOnNaN
                    nan = True
                    break
                total += loss; nb += 1
                if stop['flag']:
                    break
            vtotal, vn = 0.0, 0
            if not nan:
                for idx, val, dec_in, dec_out, w in prefetch(validation_batches(s2s, val_filenames or filenames, split_rand)):
                    loss, _ = engine.train_step(idx, val, dec_in, dec_out, w, None, mode=0)
                    vtotal += loss; vn += 1
            val_loss = vtotal / vn if vn else float('nan')
            history.append({'loss': total / max(nb, 1), 'val_loss': val_loss})
            s2s.logger.info('epoch %d: loss %.4f val_loss %.4f (%d/%d batches)', epoch + 1, history[-1]['loss'], val_loss, nb, vn)
            if nan or not np.isfinite(val_loss):
                break
            weights = engine.train_weights()
            # ModelCheckpoint("model.ckpt.weights-{epoch:02d}-{val_loss:.2f}.h5", save_weights_only=True), seq2seq.py:621-622
            keras_h5.write_model('model.ckpt.weights-%02d-%.2f.h5' % (epoch + 1, val_loss), s2s._config_dict(), weights)
            if val_loss < best:
                best, best_weights, wait = val_loss, weights, 0
            else:
                wait += 1
                if wait >= 3:                          # EarlyStopping(patience=3, restore_best_weights)
                    s2s.logger.info('Epoch %05d: early stopping', epoch + 1)
                    break
            if stop['flag']:
                break
    finally:
        if old_handler is not None:
            signal.signal(signal.SIGINT, old_handler)
        engine.train_end()
    s2s.history = history
    if best_weights is not None:
        s2s.logger.info('training finished with val_loss %f', best)
        s2s._weights = {k: np.array(v) for k, v in best_weights.items()}
        s2s._dirty = True
        s2s.status = 2
    else:
        s2s.logger.critical('training failed')
        s2s._weights = engine.get_weights()
        s2s._dirty = True
        s2s.status = 1
