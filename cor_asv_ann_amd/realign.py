"""Soft alignments in window form and the post-decode re-alignment that consumes them.

The attention of the decoder (attention.py:553-571) is zero outside a window of at most 2*window+1 input positions,
so one output step's alignment row is fully described by (first position, <= 11 weights).  `correct_lines` returns
one `SparseAlignment` per line: it behaves like the reference's list of T-wide rows (`alignment[j][i]`, `len`,
iteration, `numpy.asarray`) -- rows are materialised only when somebody asks -- and `alignment2path` below runs the
Viterbi re-alignment of the OCR-D wrapper (wrapper/transcode.py:279-349) directly on the windows.
"""
import numpy as np


class SparseAlignment(object):
    """Alignment rows of one decoded line: row j is zero outside [lo[j], lo[j] + K) (lo[j] < 0: an all-NaN row)."""

    __slots__ = ('lo', 'w', 'width')

    def __init__(self, lo, w, width):
        self.lo = np.asarray(lo, np.int32)
        self.w = np.asarray(w, np.float32)
        self.width = int(width)

    @classmethod
    def identity(cls, n):
        """`np.eye(n)` of the beam fallback (seq2seq.py:834)."""
        w = np.zeros((n, 1), np.float32)
        w[:, 0] = 1.0
        return cls(np.arange(n, dtype=np.int32), w, n)

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

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self.row(k) for k in range(*j.indices(len(self)))]
        if j < 0:
            j += len(self)
        if not 0 <= j < len(self):
            raise IndexError(j)
        return self.row(j)

    def __iter__(self):
        return (self.row(j) for j in range(len(self)))

    def __array__(self, dtype=None, copy=None):
        a = np.stack([self.row(j) for j in range(len(self))]) if len(self) else np.zeros((0, self.width), np.float32)
        return a.astype(dtype) if dtype is not None else a

    def tolist(self):
        return [self.row(j).tolist() for j in range(len(self))]

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


def alignment2path(alignment, i_max, j_max, min_score):
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
