"""Line sharding across the GPUs of one node.

Lines are independent at inference (no cross-line state, seq2seq.py:113 `stateful=False`), so the
path shards with no data-path collective: every rank holds a replica of the weights, decodes a
contiguous range of lines on its own GPU, and one all-gather of fixed-width result records
(RCCL over xGMI when the backend is "nccl") makes the decoded lines available everywhere.
"""
import numpy as np


def shard_bounds(n_items, world_size, rank):
    """Contiguous [lo, hi) of rank's share; the first `n_items % world_size` ranks get one more."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def record_width(steps):
    return 2 * steps + 4


def pack_records(idx, prob, length, score, found=None):
    """(n,S) int32 characters, (n,S) float32 probabilities, (n,) lengths, (n,) float64 scores
    -> (n, 2S+4) int32 records (floats travel as their bit patterns)."""
    idx = np.ascontiguousarray(idx, np.int32)
    n, S = idx.shape
    rec = np.zeros((n, record_width(S)), np.int32)
    rec[:, :S] = idx
    rec[:, S:2 * S] = np.ascontiguousarray(prob, np.float32).view(np.int32)
    rec[:, 2 * S] = np.asarray(length, np.int32)
    rec[:, 2 * S + 1:2 * S + 3] = np.ascontiguousarray(score, np.float64).reshape(n, 1).view(np.int32)
    rec[:, 2 * S + 3] = 1 if found is None else np.asarray(found, np.int32)
    return rec


def records_from_lines(lines, probs, scores, lut, steps):
    """Decoded strings + per-character probabilities (the return values of correct_lines) -> records.
    lut: code point -> vocabulary index, last slot for everything beyond (Sequence2Sequence._codepoint_lut)."""
    from itertools import chain
    n, S = len(lines), int(steps)
    idx = np.zeros((n, S), np.int32)
    prob = np.zeros((n, S), np.float32)
    lens = np.fromiter((min(len(t), S) for t in lines), dtype=np.int64, count=n)
    total = int(lens.sum())
    if total:
        cps = np.frombuffer(''.join(t[:S] for t in lines).encode('utf-32-le', 'surrogatepass'), dtype=np.uint32)
        starts = np.cumsum(lens) - lens
        pos = np.repeat(np.arange(n) * S - starts, lens) + np.arange(total)        # flat position of every character
        np.put(idx, pos, np.maximum(lut[np.minimum(cps, len(lut) - 1)], 0))
        plens = np.fromiter((min(len(p), m) for p, m in zip(probs, lens)), dtype=np.int64, count=n)
        ptotal = int(plens.sum())
        if ptotal:
            flat = np.fromiter(chain.from_iterable(p[:m] for p, m in zip(probs, plens)), dtype=np.float32, count=ptotal)
            ppos = np.repeat(np.arange(n) * S - (np.cumsum(plens) - plens), plens) + np.arange(ptotal)
            np.put(prob, ppos, flat)
    return pack_records(idx, prob, lens.astype(np.int32), np.asarray(scores, np.float64))


def unpack_records(rec):
    rec = np.ascontiguousarray(rec, np.int32)
    S = (rec.shape[1] - 4) // 2
    idx = rec[:, :S].copy()
    prob = rec[:, S:2 * S].copy().view(np.float32)
    length = rec[:, 2 * S].copy()
    score = rec[:, 2 * S + 1:2 * S + 3].copy().view(np.float64).reshape(-1)
    found = rec[:, 2 * S + 3].copy()
    return idx, prob, length, score, found


def all_gather_records(rec, n_total, device=None, group=None):
    """All-gather every rank's records into the (n_total, width) array in line order.  Shards are
    padded to the size of the largest one because all_gather needs equal shapes."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-n_total // world)
    width = rec.shape[1]
    mine = torch.zeros((per, width), dtype=torch.int32, device=device or 'cpu')
    mine[:rec.shape[0]] = torch.from_numpy(np.ascontiguousarray(rec)).to(mine.device)
    out = torch.empty((world * per, width), dtype=torch.int32, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().numpy().reshape(world, per, width)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(out[r, :hi - lo])
    return np.concatenate(parts, axis=0)


class _DeviceRecords(object):
    """`__cuda_array_interface__` view of the engine's record buffer: lets torch wrap the device memory the library
    packed the records into, without a copy."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': '<i4', 'data': (int(ptr), False), 'version': 2,
                                         'strides': None}


def all_gather_device_records(engine, n_total, device, group=None):
    """The all-gather of `all_gather_records`, fed from the engine's device-resident record buffer (engine.records_reset /
    records_append: every rank holds ceil(n_total / world) records, zero-padded): no host-side packing, no host-to-device
    copy.  `device` = the torch device of this rank's GPU (the one the engine runs on)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    ptr, nbytes = engine.records_device_ptr()               # (waits for the engine's stream)
    width = engine._rec_shape[1]
    assert engine._rec_shape[0] == per and nbytes == per * width * 4
    mine = torch.as_tensor(_DeviceRecords(ptr, (per, width)), device=device)
    out = torch.empty((world * per, width), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().numpy().reshape(world, per, width)
    return np.concatenate([out[r, :shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0]] for r in range(world)], axis=0)


class NativeComm(object):
    """The all-gather of result records through the C ABI (`casv_comm_*`: RCCL on the model handle's device and stream) --
    no torch in the product path.  The 128-byte RCCL id travels from rank 0 to the others over a plain TCP socket
    (`MASTER_ADDR` / `MASTER_PORT` + 1 of the usual launcher environment)."""

    def __init__(self, engine, rank=None, world=None, addr=None, port=None):
        import ctypes
        import os
        from . import _native as nv
        self.engine, self.nv = engine, nv
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port or int(os.environ.get('MASTER_PORT', '29500')) + 1)
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            nv.check(engine.lib.casv_comm_unique_id(uid))
        if self.world > 1:
            uid.raw = exchange_bytes(uid.raw, self.rank, self.world, addr, port)
        nv.check(engine.lib.casv_comm_init(engine.handle, self.rank, self.world, uid))

    def all_gather_records(self, rec, n_total):
        per = -(-n_total // self.world)
        width = rec.shape[1]
        mine = np.zeros((per, width), np.int32)
        mine[:rec.shape[0]] = rec
        out = np.empty((self.world * per, width), np.int32)
        self.nv.check(self.engine.lib.casv_comm_all_gather(self.engine.handle, self.nv.ptr(mine), self.nv.ptr(out), mine.nbytes))
        out = out.reshape(self.world, per, width)
        return np.concatenate([out[r, :shard_bounds(n_total, self.world, r)[1] - shard_bounds(n_total, self.world, r)[0]]
                               for r in range(self.world)], axis=0)

    def all_gather_device_records(self, n_total):
        """The same from the engine's device-resident record buffer (engine.records_reset / records_append): the
        collective reads the records where the pack kernel wrote them."""
        per = -(-n_total // self.world)
        width = self.engine._rec_shape[1]
        assert self.engine._rec_shape[0] == per
        out = np.empty((self.world * per, width), np.int32)
        self.nv.check(self.engine.lib.casv_comm_all_gather_records(self.engine.handle, self.nv.ptr(out)))
        out = out.reshape(self.world, per, width)
        return np.concatenate([out[r, :shard_bounds(n_total, self.world, r)[1] - shard_bounds(n_total, self.world, r)[0]]
                               for r in range(self.world)], axis=0)

    def all_gather_objects(self, obj, width=1024):
        """A small JSON-serialisable object of every rank, in rank order (reports: per-rank times, host placement) -- one
        casv_comm_all_gather of a fixed-width record."""
        import json
        raw = json.dumps(obj).encode()
        if len(raw) > width - 4:
            raise ValueError('object of %d bytes does not fit a %d-byte record' % (len(raw), width))
        mine = np.zeros(width, np.uint8)
        mine[:4] = np.frombuffer(np.int32(len(raw)).tobytes(), np.uint8)
        mine[4:4 + len(raw)] = np.frombuffer(raw, np.uint8)
        out = np.empty(self.world * width, np.uint8)
        self.nv.check(self.engine.lib.casv_comm_all_gather(self.engine.handle, self.nv.ptr(mine), self.nv.ptr(out), width))
        res = []
        for r in range(self.world):
            rec = out[r * width:(r + 1) * width]
            n = int(np.frombuffer(rec[:4].tobytes(), np.int32)[0])
            res.append(json.loads(rec[4:4 + n].tobytes().decode()))
        return res

    def max(self, value):
        """Maximum of a float over the ranks (also a barrier)."""
        from ctypes import byref, c_double
        v = c_double(float(value))
        self.nv.check(self.engine.lib.casv_comm_all_reduce_max(self.engine.handle, byref(v)))
        return v.value

    def close(self):
        self.nv.check(self.engine.lib.casv_comm_destroy(self.engine.handle))


def exchange_bytes(payload, rank, world, addr, port, timeout=120.0):
    """Rank 0 sends `payload` to every other rank over TCP; returns the payload on every rank."""
    import socket
    import time
    if rank == 0:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(payload)
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                buf = b''
                while len(buf) < len(payload):
                    chunk = c.recv(len(payload) - len(buf))
                    if not chunk:
                        raise ConnectionError('rank 0 closed the connection early')
                    buf += chunk
                return buf
        except (ConnectionRefusedError, socket.timeout, OSError):
            if time.time() > deadline:
                raise
            time.sleep(0.2)


def records_to_strings(idx, length, i_c):
    return [''.join(i_c[int(c)] for c in idx[j, :int(length[j])]) for j in range(idx.shape[0])]
