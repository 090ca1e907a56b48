"""Dependency-free reader/writer for the subset of HDF5 that Keras 2.3 / h5py model files use.

The reference stores models with `keras.Model.save_weights` + a `config` group written through h5py
(seq2seq.py:1121-1141) and reads them back with h5py (seq2seq.py:1143-1213).  h5py/libhdf5 are not a
dependency of this package, so the container format is restated here from the HDF5 File Format
Specification (version 1.1/2.0 structures, i.e. what libhdf5 writes with its default `libver='earliest'`):

  reader: superblock v0/v1, version-1 object headers (+ continuation blocks), old-style groups (symbol
          table message -> v1 B-tree -> SNOD nodes -> local heap) and compact new-style link messages,
          dataspace v1/v2, datatypes fixed-point / IEEE float / fixed-length string / enum (numpy bool),
          data layout v1-v3 compact / contiguous / chunked (v1 chunk B-tree; deflate + shuffle filters),
          attribute messages v1-v3 (numeric and fixed-length-string arrays; variable-length -> None).
  writer: superblock v0, one symbol-table node per group, contiguous little-endian datasets, attributes
          with numeric or fixed-length-string arrays -- the layout h5py itself produces for such files.

`tests/test_hdf5.py` checks the reader against files written by libhdf5 (fixtures under tests/golden made
with h5py by tests/golden/make_keras_h5.py) and, where an h5py interpreter exists, the writer's files
against h5py.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


def is_hdf5(filename):
    try:
        with open(filename, 'rb') as f:
            for off in (0, 512, 1024, 2048):
                f.seek(off)
                if f.read(8) == SIGNATURE:
                    return True
    except OSError:
        pass
    return False


# =============================================================================================
# reader
# =============================================================================================
class _Datatype(object):
    def __init__(self, cls, size, dtype=None, strpad=0, base=None, vlen=False):
        self.cls, self.size, self.dtype, self.strpad, self.base, self.vlen = cls, size, dtype, strpad, base, vlen


def _parse_datatype(buf, pos):
    """Datatype message (spec IV.A.2.d) -> (_Datatype, bytes consumed)."""
    cv, b0, b1, b2, size = struct.unpack_from('<BBBBI', buf, pos)
    cls, version = cv & 0x0f, cv >> 4
    p = pos + 8
    if cls == 0:        # fixed-point
        order = '>' if b0 & 1 else '<'
        signed = bool(b0 & 8)
        p += 4
        return _Datatype(0, size, np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size))), p - pos
    if cls == 1:        # floating point
        order = '>' if b0 & 1 else '<'
        p += 12
        return _Datatype(1, size, np.dtype('%sf%d' % (order, size))), p - pos
    if cls == 3:        # fixed-length string
        return _Datatype(3, size, np.dtype('S%d' % size), strpad=b0 & 0x0f), p - pos
    if cls == 8:        # enumeration: base type, names, values
        nmemb = b0 | (b1 << 8)
        base, used = _parse_datatype(buf, p)
        p += used
        for _ in range(nmemb):
            end = buf.index(b'\0', p)
            n = end - p + 1
            p += (n + 7) & ~7 if version < 3 else n
        p += nmemb * base.size
        dt = base.dtype
        if base.size == 1 and nmemb == 2:
            dt = np.dtype('bool')        # h5py's mapping of numpy bool: enum {FALSE=0, TRUE=1} over int8
        return _Datatype(8, size, dt, base=base), p - pos
    if cls == 9:        # variable length (strings of Python `str` attributes): not materialised
        base, used = _parse_datatype(buf, p)
        return _Datatype(9, size, None, base=base, vlen=True), p + used - pos
    if cls == 4:        # bit field
        return _Datatype(4, size, np.dtype('%su%d' % ('>' if b0 & 1 else '<', size))), 12
    if cls == 6:        # compound: skipped (never part of a Keras weight file)
        return _Datatype(6, size, np.dtype('V%d' % size)), 8
    return _Datatype(cls, size, np.dtype('V%d' % size)), 8


def _parse_dataspace(buf, pos, L):
    version, rank, flags = struct.unpack_from('<BBB', buf, pos)
    if version == 1:
        p = pos + 8
    elif version == 2:
        p = pos + 4
        if buf[pos + 3] == 2:           # null dataspace
            return None
    else:
        raise H5Error('dataspace message version %d is not supported' % version)
    fmt = '<%d%s' % (rank, 'Q' if L == 8 else 'I')
    return tuple(struct.unpack_from(fmt, buf, p)) if rank else ()


class _Object(object):
    """A parsed object header: messages grouped by type."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._read_header(addr)

    def first(self, mtype):
        for t, body in self.msgs:
            if t == mtype:
                return body
        return None

    @property
    def attrs(self):
        out = {}
        for t, body in self.msgs:
            if t == 0x000C:
                name, value = self.f._parse_attribute(body)
                out[name] = value
        return out


class Dataset(_Object):
    def __init__(self, f, addr):
        _Object.__init__(self, f, addr)
        sp, dt = self.first(0x0001), self.first(0x0003)
        if sp is None or dt is None:
            raise H5Error('object at %#x is not a dataset' % addr)
        self.shape = _parse_dataspace(sp, 0, f.L)
        self.type, _ = _parse_datatype(dt, 0)
        self.dtype = self.type.dtype

    def read(self):
        f = self.f
        lay = self.first(0x0008)
        if lay is None:
            raise H5Error('dataset without a layout message')
        if self.type.vlen or self.dtype is None:
            raise H5Error('variable-length datasets are not supported')
        shape = self.shape if self.shape is not None else (0,)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        nbytes = count * self.type.size
        version = lay[0]
        if version in (3, 4):
            cls = lay[1]
            if version == 4 and cls == 2:
                raise H5Error('version-4 chunked layouts (libver="latest") are not supported; re-save contiguous or with libver="earliest"')
            if cls == 0:                                    # compact
                size = struct.unpack_from('<H', lay, 2)[0]
                raw = lay[4:4 + size]
            elif cls == 1:                                  # contiguous
                addr = f._uint(lay, 2, f.O)
                raw = b'\0' * nbytes if addr == f._undef else f._bytes(addr, nbytes)
            elif cls == 2:                                  # chunked
                rank = lay[2]
                btree = f._uint(lay, 3, f.O)
                cdims = struct.unpack_from('<%dI' % rank, lay, 3 + f.O)
                raw = self._read_chunks(btree, cdims[:-1], shape)
            else:
                raise H5Error('data layout class %d is not supported' % cls)
        elif version in (1, 2):
            rank, cls = lay[1], lay[2]
            p = 8
            addr = None
            if cls != 0:
                addr = f._uint(lay, p, f.O)
                p += f.O
            dims = struct.unpack_from('<%dI' % rank, lay, p)
            p += 4 * rank
            if cls == 0:
                size = struct.unpack_from('<I', lay, p)[0]
                raw = lay[p + 4:p + 4 + size]
            elif cls == 1:
                raw = b'\0' * nbytes if addr == f._undef else f._bytes(addr, nbytes)
            else:
                raw = self._read_chunks(addr, dims[:-1], shape)
        else:
            raise H5Error('data layout message version %d is not supported' % version)
        a = np.frombuffer(raw, dtype=self.dtype if self.type.cls != 8 else self.type.base.dtype, count=count)
        if self.type.cls == 8 and self.dtype == np.dtype('bool'):
            a = a != 0
        a = a.reshape(shape)
        if a.dtype.byteorder == '>':
            a = a.astype(a.dtype.newbyteorder('<'))
        return a.copy() if shape else a.reshape(()).copy()

    def __getitem__(self, key):
        a = self.read()
        if key == ():
            return a[()]
        return a[key]

    def _filters(self):
        body = self.first(0x000B)
        if body is None:
            return []
        version, n = body[0], body[1]
        p = 8 if version == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from('<H', body, p)[0]
            p += 2
            namelen = 0
            if version == 1 or fid >= 256:
                namelen = struct.unpack_from('<H', body, p)[0]
                p += 2
            flags, ncd = struct.unpack_from('<HH', body, p)
            p += 4
            if namelen:
                p += (namelen + 7) & ~7 if version == 1 else namelen
            cd = struct.unpack_from('<%dI' % ncd, body, p)
            p += 4 * ncd
            if version == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _read_chunks(self, btree, cdims, shape):
        f = self.f
        esize = self.type.size
        rank = len(cdims)
        out = np.zeros(shape, dtype=np.dtype('V%d' % esize))
        filters = self._filters()
        if btree == f._undef:
            return out.tobytes()
        chunk_bytes = int(np.prod(cdims)) * esize

        def walk(addr):
            head = f._bytes(addr, 8 + 2 * f.O)
            if head[:4] != b'TREE' or head[4] != 1:
                raise H5Error('bad chunk B-tree node at %#x' % addr)
            level, used = head[5], struct.unpack_from('<H', head, 6)[0]
            keysize = 8 + 8 * (rank + 1)
            body = f._bytes(addr + 8 + 2 * f.O, used * (keysize + f.O) + keysize)
            for i in range(used):
                kp = i * (keysize + f.O)
                csize, fmask = struct.unpack_from('<II', body, kp)
                offs = struct.unpack_from('<%dQ' % (rank + 1), body, kp + 8)[:rank]
                child = f._uint(body, kp + keysize, f.O)
                if level > 0:
                    walk(child)
                    continue
                raw = f._bytes(child, csize)
                for j, (fid, cd) in reversed(list(enumerate(filters))):
                    if fmask & (1 << j):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:                      # shuffle
                        n = len(raw) // esize
                        raw = np.frombuffer(raw, np.uint8)[:n * esize].reshape(esize, n).T.tobytes()
                    elif fid == 3:                      # fletcher32: checksum trails the data
                        raw = raw[:-4]
                    else:
                        raise H5Error('HDF5 filter %d is not supported' % fid)
                chunk = np.frombuffer(raw[:chunk_bytes], dtype=out.dtype).reshape(cdims)
                sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
                out[sel_out] = chunk[sel_in]

        walk(btree)
        return out.tobytes()


class Group(_Object):
    def __init__(self, f, addr, name='/'):
        _Object.__init__(self, f, addr)
        self.name = name
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        f = self.f
        links = {}
        st = self.first(0x0011)
        if st is not None:
            btree, heap = f._uint(st, 0, f.O), f._uint(st, f.O, f.O)
            heap_head = f._bytes(heap, 8 + 2 * f.L + f.O)
            if heap_head[:4] != b'HEAP':
                raise H5Error('bad local heap at %#x' % heap)
            dsize = f._uint(heap_head, 8, f.L)
            daddr = f._uint(heap_head, 8 + 2 * f.L, f.O)
            hdata = f._bytes(daddr, dsize)

            def walk(addr):
                head = f._bytes(addr, 8 + 2 * f.O)
                if head[:4] == b'SNOD':
                    n = struct.unpack_from('<H', head, 6)[0]
                    esz = 2 * f.O + 24
                    body = f._bytes(addr + 8, n * esz)
                    for i in range(n):
                        noff = f._uint(body, i * esz, f.O)
                        oaddr = f._uint(body, i * esz + f.O, f.O)
                        ctype = struct.unpack_from('<I', body, i * esz + 2 * f.O)[0]
                        end = hdata.index(b'\0', noff)
                        if ctype == 2:
                            continue                     # symbolic link: not followed
                        links[hdata[noff:end].decode('utf-8')] = oaddr
                    return
                if head[:4] != b'TREE' or head[4] != 0:
                    raise H5Error('bad group B-tree node at %#x' % addr)
                used = struct.unpack_from('<H', head, 6)[0]
                body = f._bytes(addr + 8 + 2 * f.O, used * (f.L + f.O) + f.L)
                for i in range(used):
                    walk(f._uint(body, f.L + i * (f.L + f.O), f.O))

            if btree != f._undef:
                walk(btree)
        for t, body in self.msgs:                          # compact new-style groups: link messages
            if t != 0x0006:
                continue
            flags = body[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = body[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            lsz = 1 << (flags & 3)
            nlen = int.from_bytes(body[p:p + lsz], 'little')
            p += lsz
            lname = body[p:p + nlen].decode('utf-8')
            p += nlen
            if ltype == 0:
                links[lname] = f._uint(body, p, f.O)
        if self.first(0x0002) is not None and not links and st is None:
            info = self.first(0x0002)
            p = 2 + (8 if info[1] & 1 else 0)
            if f._uint(info, p, f.O) != f._undef:
                raise H5Error('densely stored (fractal-heap) groups are not supported; re-save with libver="earliest"')
        self._links = links

    def keys(self):
        self._load()
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError(path)
            node = node.f._open(node._links[part], node.name.rstrip('/') + '/' + part)
        return node

    def visit_datasets(self, prefix=''):
        """Yield (path, Dataset) for every dataset below this group."""
        for k in self.keys():
            child = self[k]
            if isinstance(child, Group):
                for item in child.visit_datasets(prefix + k + '/'):
                    yield item
            else:
                yield prefix + k, child


class File(Group):
    def __init__(self, filename):
        with open(filename, 'rb') as fh:
            self.buf = fh.read()
        self.filename = filename
        base = -1
        for off in (0, 512, 1024, 2048, 4096):
            if self.buf[off:off + 8] == SIGNATURE:
                base = off
                break
        if base < 0:
            raise H5Error('"%s" is not an HDF5 file' % filename)
        version = self.buf[base + 8]
        if version in (0, 1):
            self.O, self.L = self.buf[base + 13], self.buf[base + 14]
            p = base + 24 + (4 if version == 1 else 0)
            self.base = self._uint(self.buf, p, self.O)
            p += 4 * self.O
            root = self._uint(self.buf, p + self.O, self.O)        # symbol table entry: name offset, header address
        elif version in (2, 3):
            self.O, self.L = self.buf[base + 9], self.buf[base + 10]
            p = base + 12
            self.base = self._uint(self.buf, p, self.O)
            root = self._uint(self.buf, p + 3 * self.O, self.O)
        else:
            raise H5Error('superblock version %d is not supported' % version)
        self.userblock = base
        self._undef = (1 << (8 * self.O)) - 1
        self._cache = {}
        Group.__init__(self, self, root, '/')

    # -- low level --------------------------------------------------------------------------------
    @staticmethod
    def _uint(buf, pos, n):
        return int.from_bytes(buf[pos:pos + n], 'little')

    def _bytes(self, addr, n):
        a = addr + self.base
        if a < 0 or a + n > len(self.buf):
            raise H5Error('address %#x+%d outside the file' % (addr, n))
        return self.buf[a:a + n]

    def _read_header(self, addr):
        head = self._bytes(addr, 16)
        if head[:4] == b'OHDR':
            return self._read_header_v2(addr)
        if head[0] != 1:
            raise H5Error('object header version %d at %#x is not supported' % (head[0], addr))
        nmsg = struct.unpack_from('<H', head, 2)[0]
        size = struct.unpack_from('<I', head, 8)[0]
        blocks = [(addr + 16, size)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            baddr, bsize = blocks.pop(0)
            data = self._bytes(baddr, bsize)
            p = 0
            while p + 8 <= bsize and len(msgs) < nmsg:
                mtype, msize, mflags = struct.unpack_from('<HHB', data, p)
                body = data[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:
                    blocks.append((self._uint(body, 0, self.O), self._uint(body, self.O, self.L)))
                msgs.append((mtype, body))
        return msgs

    def _read_header_v2(self, addr):
        head = self._bytes(addr, 6)
        flags = head[5]
        p = 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        szlen = 1 << (flags & 3)
        size = self._uint(self._bytes(addr + p, szlen), 0, szlen)
        p += szlen
        blocks = [(addr + p, size)]
        msgs = []
        track = bool(flags & 0x04)
        while blocks:
            baddr, bsize = blocks.pop(0)
            data = self._bytes(baddr, bsize)
            q = 0
            while q + 4 <= bsize:
                mtype = data[q]
                msize = struct.unpack_from('<H', data, q + 1)[0]
                q += 4 + (2 if track else 0)
                body = data[q:q + msize]
                q += msize
                if mtype == 0x10:
                    caddr, clen = self._uint(body, 0, self.O), self._uint(body, self.O, self.L)
                    blocks.append((caddr + 4, clen - 8))        # skip 'OCHK', drop checksum
                msgs.append((mtype, body))
        return msgs

    def _open(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._read_header(addr)
        types = set(t for t, _ in msgs)
        node = Dataset(self, addr) if 0x0008 in types else Group(self, addr, name)
        self._cache[addr] = node
        return node

    def _parse_attribute(self, body):
        version = body[0]
        nsz, tsz, ssz = struct.unpack_from('<HHH', body, 2)
        p = 8
        if version == 3:
            p += 1
        pad = (lambda n: (n + 7) & ~7) if version == 1 else (lambda n: n)
        name = body[p:p + nsz].split(b'\0')[0].decode('utf-8')
        p += pad(nsz)
        dt, _ = _parse_datatype(body, p)
        p += pad(tsz)
        shape = _parse_dataspace(body, p, self.L)
        p += pad(ssz)
        if shape is None:
            return name, None
        if dt.vlen:
            if body[p - pad(ssz) - pad(tsz)] & 0x0f != 9 or dt.base is None:
                return name, None
            count = int(np.prod(shape, dtype=np.int64)) if shape else 1
            items = []
            for i in range(count):              # (length, global heap collection address, object index)
                q = p + i * (8 + self.O)
                n = struct.unpack_from('<I', body, q)[0]
                items.append(self._global_heap_object(self._uint(body, q + 4, self.O), struct.unpack_from('<I', body, q + 4 + self.O)[0])[:n])
            a = np.array(items, dtype=object).reshape(shape) if shape else items[0]
            return name, a
        if dt.dtype is None:
            return name, None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        a = np.frombuffer(body, dtype=dt.dtype if dt.cls != 8 else dt.base.dtype, count=count, offset=p)
        if dt.cls == 8 and dt.dtype == np.dtype('bool'):
            a = a != 0
        a = a.reshape(shape).copy()
        return name, (a if shape else a[()])

    def _global_heap_object(self, addr, index):
        head = self._bytes(addr, 8 + self.L)
        if head[:4] != b'GCOL':
            raise H5Error('bad global heap collection at %#x' % addr)
        size = self._uint(head, 8, self.L)
        data = self._bytes(addr, size)
        p = 8 + self.L
        while p + 8 + self.L <= size:
            idx = struct.unpack_from('<H', data, p)[0]
            osize = self._uint(data, p + 8, self.L)
            if idx == 0:
                break
            if idx == index:
                return data[p + 8 + self.L:p + 8 + self.L + osize]
            p += 8 + self.L + ((osize + 7) & ~7)
        raise H5Error('global heap object %d not found' % index)

    def close(self):
        self.buf = b''

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# =============================================================================================
# writer
# =============================================================================================
def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        props = {4: (0, 32, 23, 8, 0, 23, 127), 8: (0, 64, 52, 11, 0, 52, 1023), 2: (0, 16, 10, 5, 0, 10, 15)}[dt.itemsize]
        sign = {4: 31, 8: 63, 2: 15}[dt.itemsize]
        return struct.pack('<BBBBI', 0x11, 0x20, sign, 0, dt.itemsize) + struct.pack('<HHBBBBI', *props)
    if dt.kind in 'iu':
        return struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
    if dt.kind == 'b':          # numpy bool as h5py stores it: enum {FALSE=0, TRUE=1} over int8
        base = struct.pack('<BBBBI', 0x10, 0x08, 0, 0, 1) + struct.pack('<HH', 0, 8)
        names = b'FALSE\0\0\0' + b'TRUE\0\0\0\0'
        return struct.pack('<BBBBI', 0x18, 2, 0, 0, 1) + base + names + b'\x00\x01'
    if dt.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dt.itemsize)       # null-padded ASCII
    raise H5Error('cannot store dtype %s' % dt)


def _dataspace_message(shape):
    if shape == ():
        return struct.pack('<BBBB4x', 1, 0, 0, 0)
    return struct.pack('<BBBB4x', 1, len(shape), 1, 0) + b''.join(struct.pack('<Q', s) for s in shape) * 2


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _attribute_message(name, value):
    a = np.asarray(value)
    if a.dtype.kind == 'U':
        a = np.char.encode(a, 'utf-8')
    if a.dtype.kind == 'O':
        raise H5Error('cannot store object arrays')
    if a.shape:
        a = np.ascontiguousarray(a)
    if a.dtype.byteorder == '>':
        a = a.astype(a.dtype.newbyteorder('<'))
    nm = name.encode('utf-8') + b'\0'
    dt, sp = _dtype_message(a.dtype), _dataspace_message(a.shape)
    body = struct.pack('<BxHHH', 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + a.tobytes()
    return _message(0x000C, body)


class Writer(object):
    """Build a tree with `create_group` / `create_dataset` / `attrs`, then `save(filename)`."""

    class _Node(object):
        def __init__(self):
            self.children = {}      # name -> _Node
            self.attrs = {}
            self.data = None        # ndarray for datasets

    def __init__(self):
        self.root = Writer._Node()

    def _node(self, path, create=True):
        node = self.root
        for part in [p for p in path.split('/') if p]:
            if part not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[part] = Writer._Node()
            node = node.children[part]
            if node.data is not None:
                raise H5Error('"%s" is a dataset, not a group' % part)
        return node

    def create_group(self, path):
        self._node(path)

    def create_dataset(self, path, data):
        parts = [p for p in path.split('/') if p]
        parent = self._node('/'.join(parts[:-1]))
        node = Writer._Node()
        a = np.asarray(data)
        if a.dtype.byteorder == '>':
            a = a.astype(a.dtype.newbyteorder('<'))
        node.data = np.ascontiguousarray(a) if a.shape else a
        parent.children[parts[-1]] = node

    def set_attr(self, path, name, value):
        node = self.root
        for part in [p for p in path.split('/') if p]:
            node = node.children[part]
        node.attrs[name] = value

    # -- layout -----------------------------------------------------------------------------------
    def save(self, filename):
        O = 8
        leaf_k = 4

        def count(node):
            n = len(node.children)
            for c in node.children.values():
                if c.data is None:
                    n = max(n, count(c))
            return n
        widest = count(self.root)
        while 2 * leaf_k < widest:
            leaf_k *= 2
        chunks = []                  # (address, bytes)
        state = {'pos': 0}

        def alloc(nbytes, align=8):
            state['pos'] = (state['pos'] + align - 1) // align * align
            addr = state['pos']
            state['pos'] += nbytes
            return addr

        super_size = 24 + 4 * O + (2 * O + 24)
        alloc(super_size)

        def emit_header(msgs):
            body = b''.join(msgs)
            head = struct.pack('<BxHII4x', 1, len(msgs), 1, len(body))
            addr = alloc(len(head) + len(body))
            chunks.append((addr, head + body))
            return addr

        def emit(node):
            attr_msgs = [_attribute_message(k, v) for k, v in node.attrs.items()]
            if node.data is not None:
                a = node.data
                raw = a.tobytes()
                daddr = alloc(max(len(raw), 1)) if len(raw) else UNDEF
                if len(raw):
                    chunks.append((daddr, raw))
                msgs = [_message(0x0001, _dataspace_message(a.shape)),
                        _message(0x0003, _dtype_message(a.dtype), flags=1),
                        _message(0x0005, struct.pack('<BBBB', 2, 2, 2, 0)),           # fill value: v2, alloc late, never written, undefined
                        _message(0x0008, struct.pack('<BBQQ', 3, 1, daddr, len(raw)))]
                return emit_header(msgs + attr_msgs), None
            # group: children first, then local heap, symbol node, B-tree, header
            names = sorted(node.children, key=lambda s: s.encode('utf-8'))
            entries = []
            heap = bytearray(b'\0' * 8)              # offset 0: the empty name every heap starts with
            for nm in names:
                child = node.children[nm]
                caddr, scratch = emit(child)
                off = len(heap)
                heap += _pad8(nm.encode('utf-8') + b'\0')
                entries.append((off, caddr, scratch))
            heap_free = len(heap)
            heap += struct.pack('<QQ', 1, 16)        # one free block: next = 1 (none), size 16
            heap_data_addr = alloc(len(heap))
            chunks.append((heap_data_addr, bytes(heap)))
            heap_addr = alloc(8 + 2 * 8 + O)
            chunks.append((heap_addr, b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), heap_free, heap_data_addr)))
            snod = bytearray(b'SNOD' + struct.pack('<BxH', 1, len(entries)))
            for off, caddr, scratch in entries:
                if scratch is None:
                    snod += struct.pack('<QQI4x16x', off, caddr, 0)
                else:
                    snod += struct.pack('<QQI4xQQ', off, caddr, 1, scratch[0], scratch[1])
            snod += b'\0' * ((2 * leaf_k - len(entries)) * (2 * O + 24))
            snod_addr = alloc(len(snod))
            chunks.append((snod_addr, bytes(snod)))
            internal_k = 16
            btree = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1 if entries else 0, UNDEF, UNDEF))
            if entries:
                btree += struct.pack('<QQQ', 0, snod_addr, entries[-1][0])
                used = 1
            else:
                btree += struct.pack('<Q', 0)
                used = 0
            btree += b'\0' * ((2 * internal_k - used) * (8 + O) + (0 if used else 0))
            btree_addr = alloc(len(btree))
            chunks.append((btree_addr, bytes(btree)))
            msgs = [_message(0x0011, struct.pack('<QQ', btree_addr, heap_addr))]
            return emit_header(msgs + attr_msgs), (btree_addr, heap_addr)

        root_addr, root_scratch = emit(self.root)
        eof = alloc(0)
        sb = SIGNATURE + struct.pack('<BBBxBBBxHHI', 0, 0, 0, 0, O, 8, leaf_k, 16, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQI4xQQ', 0, root_addr, 1, root_scratch[0], root_scratch[1])
        out = bytearray(eof)
        out[:len(sb)] = sb
        for addr, data in chunks:
            out[addr:addr + len(data)] = data
        with open(filename, 'wb') as fh:
            fh.write(bytes(out))
