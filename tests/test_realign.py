"""Window form of the soft alignments and the re-alignment on it (wrapper/transcode.py:279-349) against the dense,
cell-by-cell restatement in oracle/realign.py."""
import numpy as np

from cor_asv_ann_amd.realign import SparseAlignment, alignment2path, alignment2path_py, dense_to_sparse
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
        got3 = alignment2path_py(sp, i_max, j_max, min_score)           # the Python loop behind the native search
        assert got3[0] == want[0] and ((np.isnan(got3[1]) and np.isnan(want[1])) or abs(got3[1] - want[1]) < 1e-5)
        checked += 1
    assert checked == 300


def test_identity_alignment_gives_the_diagonal():
    n = 12
    path, dist = alignment2path(SparseAlignment.identity(n), n, n, 1 / 50.)
    assert path == dict([(i, i) for i in range(n)] + [(n, n)]) and dist == 0


def _reference_style_forward(alignment, i_max, j_max, min_score):
    """The cell-by-cell access pattern of the wrapper's unchanged `_alignment2path` forward pass (transcode.py:293-318):
    `alignment[j][i]` for every cell it looks at."""
    fw = np.zeros((i_max, j_max), dtype=np.float32)
    i, j = 0, 0
    while i < i_max and j < j_max:
        im1 = fw[i - 1, j] if i > 0 else 0
        jm1 = fw[i, j - 1] if j > 0 else 0
        ijm1 = fw[i - 1, j - 1] if i > 0 and j > 0 else 0
        fw[i, j] = alignment[j][i] + max(im1, jm1, ijm1)
        while True:
            i += 1
            if i == i_max:
                j += 1
                if j == j_max:
                    break
                i = 0
            if alignment[j][i] > min_score:
                break
    return fw


def test_cell_by_cell_access_costs_what_a_list_of_rows_costs():
    """The OCR-D wrapper indexes `alignment[j][i]` cell by cell (transcode.py:308,316,325).  A SparseAlignment builds its
    rows once, at the first such access, and is a plain list from then on: the reference's loop over a 101 x 101 line
    must not take more than 1.5x what it takes on a list of rows (round 2: 8x, a fresh row per access)."""
    import time
    rng = np.random.default_rng(5)
    n = T = 101
    sp = _random_alignment(rng, n, T)
    rows = list(np.asarray(sp))

    def best_of(make, repeats=7):
        best = float('inf')
        for _ in range(repeats):
            a = make()
            t0 = time.perf_counter()
            fw = _reference_style_forward(a, T, n, 1 / 256.)
            best = min(best, time.perf_counter() - t0)
        return best, fw

    t_list, fw_list = best_of(lambda: list(rows))
    t_sparse, fw_sparse = best_of(lambda: SparseAlignment(sp.lo, sp.w, T))
    assert np.array_equal(fw_list, fw_sparse)
    assert t_sparse <= 1.5 * t_list + 2e-4, (t_sparse, t_list)


def test_sparse_alignment_stays_lazy_until_rows_are_looked_at():
    rng = np.random.default_rng(6)
    sp = _random_alignment(rng, 7, 9)
    assert isinstance(sp, list) and len(sp) == 7 and bool(sp) and list.__len__(sp) == 0       # nothing built yet
    dense = np.asarray(sp)                                                                      # ... nor by asarray,
    alignment2path(sp, 9, 7, 1 / 50.)                                                           # the windows' own Viterbi,
    assert sp.value(2, int(sp.lo[2])) == dense[2, sp.lo[2]] and list.__len__(sp) == 0           # or single values
    first = sp[0]                                                                               # the first row access builds all
    assert list.__len__(sp) == 7 and isinstance(sp, SparseAlignment) and np.array_equal(first, dense[0])
    assert np.array_equal(np.asarray(sp), dense) and len(sp) == 7 and sp.tolist() == dense.tolist()
    assert [r.tolist() for r in sp] == dense.tolist() and len(sp[1:3]) == 2
    fresh = _random_alignment(np.random.default_rng(6), 7, 9)
    assert fresh != [] and not fresh == []                 # comparing looks at the rows too (an empty list otherwise)
    assert all(np.array_equal(a, b) for a, b in zip(fresh, dense))
    empty = SparseAlignment(np.zeros(0, np.int32), np.zeros((0, 11), np.float32), 9)
    assert len(empty) == 0 and not empty and list(empty) == [] and np.asarray(empty).shape == (0, 9)


def test_native_search_is_the_one_that_runs_and_agrees_on_long_lines():
    """`alignment2path` goes through `casv_realign_path` (host code of the library: no device needed); on page-sized lines it
    returns what the Python loop and the dense restatement return, a good deal faster."""
    import time
    from cor_asv_ann_amd import realign
    assert realign._native_lib() is not None
    rng = np.random.default_rng(11)
    t_native = t_py = 0.0
    for case in range(40):
        T = int(rng.integers(30, 120)); n_out = int(rng.integers(30, 120))
        sp = _random_alignment(rng, n_out, T, nan_rows=case % 5 == 0)
        i_max, j_max = int(rng.integers(1, T + 1)), int(rng.integers(1, n_out + 1))
        t0 = time.perf_counter(); a = alignment2path(sp, i_max, j_max, 1 / 640.); t1 = time.perf_counter()
        b = alignment2path_py(sp, i_max, j_max, 1 / 640.); t2 = time.perf_counter()
        t_native += t1 - t0; t_py += t2 - t1
        want = oracle_path([row for row in np.asarray(sp)], i_max, j_max, 1 / 640.)
        assert a[0] == b[0] == want[0], case
        assert (np.isnan(a[1]) and np.isnan(want[1])) or (abs(a[1] - want[1]) < 1e-5 + 1e-6 * abs(a[1]) and abs(a[1] - b[1]) < 1e-9), case   # (the dense restatement sums in float32)
    assert t_native < 0.5 * t_py, (t_native, t_py)
