"""The GEMM launcher alone, on caller-given operands (C ABI: casv_debug_contract), against a float64 product on the host.

What the end-to-end parity tests cannot see: launch forms that only the train step's big contractions take (K split over
workgroups and over the two wave groups of a workgroup -- ADVICE round 4 found a wrong prologue shortcut there by reading the
code), and HOW accurate each arithmetic is: the fp32-input kernels' k-ordered fmaf chain against the split-bf16 kernels' six
bf16 products per K tile, both measured against float64 on the bench's operand shapes.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def engine():
    from cor_asv_ann_amd.engine import HipEngine
    eng = HipEngine(1, 32, 8)
    try:
        yield eng
    finally:
        eng.set_option('split_bf16', -1)
        eng.set_option('tile', -1)
        eng.close()


def _operands(M, N, K, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((M, K)) * scale).astype(np.float32)
    Bt = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    return A, Bt, bias


# A sequential fp32 chain over K random-sign terms leaves an error of about sum|a||b| * 2^-24 * 0.64 (rms; each of the K additions
# rounds the running sum, which grows like sqrt(k)) whatever K is -- the fp32-input MFMA's k-ordered fmaf chain measures 0.45 in
# these units; a wrong or missing K tile would show as 1e4 and more.
RMS_BOUND, MAX_BOUND = 1.5, 12.0


def _errors(C, A, Bt, bias):
    """(rms, max) error of C against the float64 product, in units of 2^-24 * sum |a||b| (what a forward error bound of any
    summation order is stated in)."""
    ref = A.astype(np.float64) @ Bt.astype(np.float64).T + bias.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(Bt).astype(np.float64).T + np.abs(bias)
    err = (C.astype(np.float64) - ref) / (mag * 2.0 ** -24)
    assert np.isfinite(C).all(), 'an element was not written'
    return float(np.sqrt(np.mean(err ** 2))), float(np.abs(err).max())


@pytest.mark.parametrize('shape', [(2048, 1024, 512), (2048, 1024, 1024), (1536, 1024, 768), (512, 2048, 512), (51712 // 8, 512, 512)])
@pytest.mark.parametrize('tile', [-1, 0])
def test_split_k_forms_of_the_plain_contraction(engine, shape, tile):
    """K split over workgroups (float atomics) and over the two wave groups of a workgroup (KS = 2: the form a plain
    contraction takes when its 128x128 grid leaves CUs idle, e.g. M = 2048, N = 1024, K = 512 -> two K shares of 16 tiles, grid
    256): every element equals the float64 product to fp32 rounding.  (Round 4's prologue shortcut for K tiles 2 and 3 loaded
    the wrong tiles under KS = 2: K tiles 2 and 3 counted twice, 4..7 dropped.)"""
    M, N, K = shape
    A, Bt, bias = _operands(M, N, K, seed=M + N + K)
    engine.set_option('tile', tile)
    for split_k, groups in ((False, False), (True, False), (True, True)):
        C = engine.debug_contract(A, Bt, bias, split_k=split_k, wave_groups=groups)
        rms, mx = _errors(C, A, Bt, bias)
        assert rms < RMS_BOUND and mx < MAX_BOUND, (shape, tile, split_k, groups, rms, mx)


@pytest.mark.parametrize('M,N,K', [(51712, 512, 512), (52224, 2048, 512), (51712, 256, 512), (51712, 1024, 2048)])
def test_partial_round_of_a_big_contraction_goes_as_small_tiles_with_the_same_bits(engine, M, N, K):
    """The train step's whole-sequence contractions (M = 51 712 / 52 224 rows; the caller leaves the launch form open): a 128x128 tile
    grid that ends in a partial round of workgroups has the row blocks of that round launched as 32- or 64-row tiles instead
    (gemm.hip, cut_partial_round).  Same k-ordered chain per element: the result equals the one-launch form bit for bit, with
    and without `C +=`-free bias; and the float64 product to rounding."""
    A, Bt, bias = _operands(M, N, K, seed=M + N)
    one = engine.debug_contract(A, Bt, bias)                      # ksplit = 0: one block per tile, one launch
    cut = engine.debug_contract(A, Bt, bias, split_k=True)        # ksplit = -1: the launcher may cut the partial round off
    assert np.array_equal(one, cut)
    if N <= 512:
        rms, mx = _errors(cut, A, Bt, bias)
        assert rms < RMS_BOUND and mx < MAX_BOUND, (rms, mx)


@pytest.mark.parametrize('M,N,K', [(8192, 2048, 768), (8192, 2048, 1024), (8192, 2048, 1536), (1024, 512, 512), (333, 200, 96)])
def test_plain_contraction_equals_float64_to_rounding(engine, M, N, K):
    """The shapes of the decoder's launches (and a ragged one), every tile shape: same bits from all of them, float64 to rounding."""
    A, Bt, bias = _operands(M, N, K, seed=K)
    first = None
    for tile in (0, 1, 2, -1):
        engine.set_option('tile', tile)
        C = engine.debug_contract(A, Bt, bias)
        rms, mx = _errors(C, A, Bt, bias)
        assert rms < RMS_BOUND and mx < MAX_BOUND, (tile, rms, mx)
        if first is None:
            first = C
        assert np.array_equal(first, C), 'tile shape %d changes the bits' % tile


@pytest.mark.parametrize('M', [10240, 10300, 16640, 8448, 300])
def test_split_tile_shapes_and_the_partial_round_cut_give_the_same_bits(engine, M):
    """The split arithmetic's two tile shapes contract an element with the same instruction sequence, so a launch may mix them:
    256x256 tiles for the whole rounds of workgroups, 128x128 tiles for the rows of a partial last round (the page call: 10 240 rows
    x 2048 columns = 320 tiles on 256 CUs) or a ragged last row block (gemm.hip, split256_cut_rows).  Mode 2 (with the cut) equals
    mode 1 (128x128 tiles only) bit for bit, and float64 to rounding."""
    N, K = 2048, 768
    A, Bt, bias = _operands(M, N, K, seed=M)
    engine.set_option('split_bf16', 1)
    one = engine.debug_contract(A, Bt, bias, weight=True)
    engine.set_option('split_bf16', 2)
    two = engine.debug_contract(A, Bt, bias, weight=True)
    engine.set_option('split_bf16', -1)
    assert np.array_equal(one, two)
    rms, mx = _errors(two, A, Bt, bias)
    assert rms < RMS_BOUND and mx < MAX_BOUND, (rms, mx)


@pytest.mark.parametrize('K', [768, 1024, 1536])
@pytest.mark.parametrize('scale', [1.0, 37.0])
def test_split_bf16_is_as_accurate_as_the_fp32_chain(engine, K, scale):
    """the split-bf16 arithmetic (three bf16 parts per value, six products per K tile, fp32
    accumulation) against float64, beside the fp32-input kernels' k-ordered fmaf chain against float64, on the bench's operand
    shapes (M = 8192 rows, N = 2048 gate columns, K = 768 / 1024 / 1536): RMS and maximum error of the split sums at most 1.1 x
    the chain's.  Both modes of the option, weights as a pre-split image and staged per tile."""
    M, N = 8192, 2048
    A, Bt, bias = _operands(M, N, K, seed=7 * K, scale=scale)
    engine.set_option('split_bf16', 0)
    chain = _errors(engine.debug_contract(A, Bt, bias), A, Bt, bias)
    for mode in (1, 2):
        engine.set_option('split_bf16', mode)
        for weight in (False, True):
            C = engine.debug_contract(A, Bt, bias, weight=weight)
            rms, mx = _errors(C, A, Bt, bias)
            assert rms <= 1.1 * chain[0] and mx <= 1.1 * chain[1], (mode, weight, K, (rms, mx), chain)
    engine.set_option('split_bf16', -1)


@pytest.mark.parametrize('M,N,K', [(2048, 512, 8192), (2048, 1024, 4096), (512, 512, 2048), (256, 768, 16384)])
def test_weight_gradient_contraction_on_split_operands_is_as_accurate_as_the_fp32_one(engine, M, N, K):
    """The train step's weight gradients, C = A^T . B on K-major operands: the split kernel (csrc/gemm_tn_split.hip: 256x256 tiles, the
    transposition in the staging's load pattern, K shared out over workgroups with float atomics) against float64, beside the
    fp32-input kernel (csrc/gemm_tn.hip) against float64 -- error at most 1.1 x the fp32 kernel's plus the noise floor of a sum in
    blocks -- and never a missing or doubled K tile (that would show as 1e4 in these units)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)

    def errors(C):
        assert np.isfinite(C).all(), 'an element was not written'
        err = (C.astype(np.float64) - ref) / (mag * 2.0 ** -24)
        return float(np.sqrt(np.mean(err ** 2))), float(np.abs(err).max())
    engine.set_option('split_bf16', 0)
    fp32 = errors(engine.debug_contract(A, B, k_major=True))
    engine.set_option('split_bf16', 2)
    split = errors(engine.debug_contract(A, B, k_major=True))
    engine.set_option('split_bf16', -1)
    assert fp32[0] < RMS_BOUND and fp32[1] < MAX_BOUND, fp32
    assert split[0] <= 1.1 * fp32[0] + 0.05 and split[1] <= 1.1 * fp32[1] + 0.5, (split, fp32)
