"""Soft alignments in window form and the post-decode re-alignment that consumes them.

The attention of the decoder (attention.py:553-571) is zero outside a window of at most 2*window+1 input positions,
so one output step's alignment row is fully described by (first position, <= 11 weights).  `correct_lines` returns
one `SparseAlignment` per line: it IS a list of T-wide rows (`alignment[j][i]`, `len`, iteration, `numpy.asarray`)
whose rows are built, all at once, only when somebody first looks at one -- and `alignment2path` below runs the
Viterbi re-alignment of the OCR-D wrapper (wrapper/transcode.py:279-349) directly on the windows.
"""
import numpy as np


class SparseAlignment(list):
    """Alignment rows of one decoded line: row j is zero outside [lo[j], lo[j] + K) (lo[j] < 0: an all-NaN row).

    A `list` (the reference hands out a list of T-wide rows, transcode.py:308,316,325 index it cell by cell) that stays
    EMPTY until somebody looks at a row: the first indexing / iteration / comparison builds all rows with one scatter and
    turns the object into a plain list of numpy rows (`_DenseAlignment`: the list type's own C slots, so the wrapper's
    unchanged `_alignment2path` pays nothing per cell).  `len()`, `numpy.asarray`, `cells_above` / `value` and
    `alignment2path` below work on the windows and never build the rows."""

    __slots__ = ('lo', 'w', 'width')

    def __init__(self, lo, w, width):
        list.__init__(self)
        self.lo = np.asarray(lo, np.int32)
        self.w = np.asarray(w, np.float32)
        self.width = int(width)

    @classmethod
    def identity(cls, n):
        """`np.eye(n)` of the beam fallback (seq2seq.py:834)."""
        w = np.zeros((n, 1), np.float32)
        w[:, 0] = 1.0
        return cls(np.arange(n, dtype=np.int32), w, n)

    def dense(self):
        """All rows as one (n, width) float32 array (one scatter over the windows)."""
        n, K = self.lo.shape[0], self.w.shape[1] if self.w.ndim == 2 else 0
        out = np.zeros((n, self.width), np.float32)
        if n and K and self.width:
            lo = self.lo.astype(np.int64)
            cols = lo[:, None] + np.arange(K)[None, :]
            ok = (lo[:, None] >= 0) & (cols < self.width)
            rows = np.broadcast_to(np.arange(n)[:, None], cols.shape)
            out[rows[ok], cols[ok]] = self.w[ok]
        if n:
            out[self.lo < 0] = np.nan
        return out

    def _materialise(self):
        """Build the rows once and become a plain list of them."""
        list.extend(self, self.dense())
        self.__class__ = _DenseAlignment
        return self

    def __len__(self):
        return self.lo.shape[0]

    def row(self, j):
        out = np.zeros(self.width, np.float32)
        lo = int(self.lo[j])
        if lo < 0:
            out[:] = np.nan
            return out
        k = min(self.w.shape[1], self.width - lo)
        if k > 0:
            out[lo:lo + k] = self.w[j, :k]
        return out

    # anything that looks at rows builds them first (then the list type answers by itself)
    def __getitem__(self, j):
        return list.__getitem__(self._materialise(), j)

    def __iter__(self):
        return list.__iter__(self._materialise())

    def __reversed__(self):
        return list.__reversed__(self._materialise())

    def __contains__(self, x):
        return list.__contains__(self._materialise(), x)

    def __eq__(self, other):
        return list.__eq__(self._materialise(), other)

    def __ne__(self, other):
        return list.__ne__(self._materialise(), other)

    __hash__ = None

    def __repr__(self):
        return list.__repr__(self._materialise())

    def __add__(self, other):
        return list.__add__(self._materialise(), other)

    def __reduce__(self):
        return (SparseAlignment, (self.lo, self.w, self.width))

    def copy(self):
        return list(self._materialise())

    def __array__(self, dtype=None, copy=None):
        a = self.dense()
        return a.astype(dtype) if dtype is not None else a

    def tolist(self):
        return self.dense().tolist()

    def cells_above(self, j, min_score, i_max):
        """Positions i < i_max of row j whose weight exceeds min_score, ascending, with their weights."""
        lo = int(self.lo[j])
        if lo < 0:
            return [], []
        w = self.w[j]
        pos = lo + np.nonzero(w > min_score)[0]
        pos = pos[pos < min(i_max, self.width)]
        return pos.tolist(), w[pos - lo].tolist()

    def value(self, j, i):
        lo = int(self.lo[j])
        if lo < 0:
            return float('nan')
        k = i - lo
        return float(self.w[j, k]) if 0 <= k < self.w.shape[1] and i < self.width else 0.0


class _DenseAlignment(SparseAlignment):
    """A SparseAlignment whose rows have been built: every list operation is the list type's own again."""

    __slots__ = ()
    __len__ = list.__len__
    __getitem__ = list.__getitem__
    __iter__ = list.__iter__
    __reversed__ = list.__reversed__
    __contains__ = list.__contains__
    __eq__ = list.__eq__
    __ne__ = list.__ne__
    __repr__ = list.__repr__
    __add__ = list.__add__
    copy = list.copy

    def _materialise(self):
        return self

    def __array__(self, dtype=None, copy=None):
        a = np.stack(list(self)) if list.__len__(self) else np.zeros((0, self.width), np.float32)
        return a.astype(dtype) if dtype is not None else a


def alignment2path(alignment, i_max, j_max, min_score):
    """The Viterbi re-alignment of wrapper/transcode.py:279-349 on window-form alignments: `casv_realign_path` of the C ABI
    (host code of the library, csrc/realign_host.hip: the same search, ~40x faster than the Python loop below) when the
    library is there and the call is inside its envelope, `alignment2path_py` otherwise -- identical results (tested)."""
    if not isinstance(alignment, SparseAlignment):
        alignment = dense_to_sparse(alignment)
    n = len(alignment)
    if 0 < i_max and 0 < j_max <= n and alignment.w.ndim == 2 and alignment.w.shape[1] >= 1 and alignment.width >= 1:
        lib = _native_lib()
        if lib is not None:
            from ctypes import byref, c_double, c_void_p
            lo = np.ascontiguousarray(alignment.lo, np.int32)
            w = np.ascontiguousarray(alignment.w, np.float32)
            path = np.empty(i_max + 1, np.int32)
            dist = c_double()
            rc = lib.casv_realign_path(n, alignment.width, w.shape[1], lo.ctypes.data_as(c_void_p), w.ctypes.data_as(c_void_p),
                                       int(i_max), int(j_max), float(min_score), path.ctypes.data_as(c_void_p), byref(dist))
            if rc == 0:
                hit = np.nonzero(path >= 0)[0]
                return dict(zip(hit.tolist(), path[hit].tolist())), dist.value
    return alignment2path_py(alignment, i_max, j_max, min_score)


_LIB = []


def _native_lib():
    """The HIP library for its host-side helper (no device needed); None if it has not been built."""
    if not _LIB:
        try:
            from . import _native
            _LIB.append(_native.load())
        except Exception:
            _LIB.append(None)
    return _LIB[0]


def alignment2path_py(alignment, i_max, j_max, min_score):
    """The Viterbi re-alignment of wrapper/transcode.py:279-349 -- same forward scores, same back-tracking rule, same
    returned (realignment dict input position -> output position, distance) -- visiting only the cells inside the
    attention windows instead of testing all i_max * j_max cells (the reference's forward pass skips cells with a
    score <= min_score too, transcode.py:316, it just has to look at each one to find out).

    `alignment`: a SparseAlignment, or the reference's list of rows (converted on the fly)."""
    if not isinstance(alignment, SparseAlignment):
        alignment = dense_to_sparse(alignment)
    fw = np.zeros((i_max, j_max), dtype=np.float32)

    def visit(i, j, a):
        im1 = fw[i - 1, j] if i > 0 else 0
        jm1 = fw[i, j - 1] if j > 0 else 0
        ijm1 = fw[i - 1, j - 1] if i > 0 and j > 0 else 0
        fw[i, j] = a + max(im1, jm1, ijm1)

    if i_max > 0 and j_max > 0:
        visit(0, 0, alignment.value(0, 0))                   # the scan starts here whatever the score (transcode.py:296)
        for j in range(j_max):
            pos, val = alignment.cells_above(j, min_score, i_max)
            for i, a in zip(pos, val):
                if j == 0 and i == 0:
                    continue
                visit(i, j, a)
    # backward pass (transcode.py:320-340), including numpy's wrap-around of index -1 at the borders
    i = i_max - 1 if i_max <= j_max else j_max - 2 + int(np.argmax(fw[j_max - i_max - 2:, j_max - 1]))
    j = j_max - 1 if j_max <= i_max else i_max - 2 + int(np.argmax(fw[i_max - 1, i_max - j_max - 2:]))
    realignment = {i_max: j_max}
    dist = 0
    while i >= 0 and j >= 0:
        dist += 1.0 - alignment.value(j, i)
        realignment[i] = j
        if fw[i - 1, j] > fw[i, j - 1]:
            if fw[i - 1, j] > fw[i - 1, j - 1]:
                i -= 1
            else:
                i -= 1
                j -= 1
        elif fw[i, j - 1] > fw[i - 1, j - 1]:
            j -= 1
        else:
            j -= 1
            i -= 1
    realignment[0] = 0
    return realignment, dist


def dense_to_sparse(rows):
    """List of T-wide rows -> SparseAlignment holding every non-zero entry (window = widest non-zero span)."""
    a = np.asarray([np.asarray(r, np.float32) for r in rows], np.float32)
    if a.ndim != 2 or a.shape[0] == 0:
        return SparseAlignment(np.zeros(0, np.int32), np.zeros((0, 1), np.float32), a.shape[1] if a.ndim == 2 else 0)
    n, T = a.shape
    nan_row = np.isnan(a).all(axis=1)
    nz = (a != 0) & ~np.isnan(a)
    any_nz = nz.any(axis=1)
    first = np.where(any_nz, nz.argmax(axis=1), 0)
    last = np.where(any_nz, T - 1 - nz[:, ::-1].argmax(axis=1), 0)
    K = int((last - first).max()) + 1
    w = np.zeros((n, K), np.float32)
    for j in range(n):
        k = min(K, T - first[j])
        w[j, :k] = np.nan_to_num(a[j, first[j]:first[j] + k], nan=0.0)
    lo = np.where(nan_row, -1, first).astype(np.int32)
    return SparseAlignment(lo, w, T)
