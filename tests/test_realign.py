"""Window form of the soft alignments and the re-alignment on it (wrapper/transcode.py:279-349) against the dense,
cell-by-cell restatement in oracle/realign.py."""
import numpy as np

from cor_asv_ann_amd.realign import SparseAlignment, alignment2path, dense_to_sparse
from oracle.realign import alignment2path as oracle_path


def _random_alignment(rng, n_out, T, window=5, jump=0.15, nan_rows=False):
    """Rows like the decoder's attention: a normalised window of <= 2*window+1 weights around a position that moves
    forward by about one per step, with occasional stalls and jumps."""
    lo = np.zeros(n_out, np.int32)
    w = np.zeros((n_out, 2 * window + 1), np.float32)
    pos = 0.0
    for j in range(n_out):
        pos += rng.choice([0.0, 1.0, 1.0, 1.0, 2.0]) if rng.random() > jump else rng.integers(-2, 4)
        pos = float(np.clip(pos, 0, T - 1))
        first = max(0, int(np.ceil(pos - window)))
        last = min(T - 1, int(np.floor(pos + window)))
        cnt = last - first + 1
        e = np.exp(rng.normal(0, 1.5, cnt) - 0.6 * np.abs(np.arange(first, last + 1) - pos)).astype(np.float32)
        lo[j] = first
        w[j, :cnt] = e / e.sum()
        if nan_rows and rng.random() < 0.05:
            lo[j] = -1
            w[j] = np.nan
    return SparseAlignment(lo, w, T)


def test_sparse_alignment_is_a_list_of_rows():
    rng = np.random.default_rng(0)
    sp = _random_alignment(rng, 9, 14)
    dense = np.asarray(sp)
    assert dense.shape == (9, 14) and len(sp) == 9
    assert np.allclose(dense.sum(axis=1), 1.0, atol=1e-6)
    for j in range(9):
        assert np.array_equal(sp[j], dense[j]) and sp[j][int(sp.lo[j])] == dense[j, sp.lo[j]]
        for i in (0, 5, 13):
            assert sp.value(j, i) == dense[j, i]
    assert [r.tolist() for r in sp] == dense.tolist() == sp.tolist()
    assert np.array_equal(np.asarray(sp[-1]), dense[-1]) and len(sp[2:5]) == 3
    back = dense_to_sparse(list(dense))
    assert np.array_equal(np.asarray(back), dense)
    eye = SparseAlignment.identity(4)
    assert np.array_equal(np.asarray(eye), np.eye(4, dtype=np.float32))
    nan = SparseAlignment(np.array([-1, 0]), np.array([[np.nan, np.nan], [0.5, 0.5]], np.float32), 3)
    assert np.isnan(nan[0]).all() and nan[1].tolist() == [0.5, 0.5, 0.0] and np.isnan(nan.value(0, 1))


def test_path_from_windows_equals_dense_viterbi():
    rng = np.random.default_rng(1)
    checked = 0
    for case in range(300):
        T = int(rng.integers(1, 40))
        n_out = int(rng.integers(1, 40))
        sp = _random_alignment(rng, n_out, T, window=int(rng.choice([1, 3, 5])), nan_rows=case % 3 == 0)
        dense = [row for row in np.asarray(sp)]
        i_max = int(rng.integers(1, T + 1))
        j_max = int(rng.integers(1, n_out + 1))
        min_score = float(rng.choice([1 / 256., 0.02, 0.2]))
        want = oracle_path(dense, i_max, j_max, min_score)
        got = alignment2path(sp, i_max, j_max, min_score)
        assert got[0] == want[0], (case, T, n_out, i_max, j_max)
        assert (np.isnan(got[1]) and np.isnan(want[1])) or abs(got[1] - want[1]) < 1e-5
        got2 = alignment2path(dense, i_max, j_max, min_score)           # the reference's list form works too
        assert got2[0] == want[0]
        checked += 1
    assert checked == 300


def test_identity_alignment_gives_the_diagonal():
    n = 12
    path, dist = alignment2path(SparseAlignment.identity(n), n, n, 1 / 50.)
    assert path == dict([(i, i) for i in range(n)] + [(n, n)]) and dist == 0
