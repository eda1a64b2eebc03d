"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit-exact (integer outputs)."""
import numpy as np
import pytest

from oracle import model as oracle
from pymasc_amd import ffi
from . import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def check_block(out, ref, max_shift, has_m, skip_ncc=False):
    S = max_shift
    if not skip_ncc:
        assert int(out[ffi.PMX_ROW_SCALARS, 0]) == ref["ncc_forward_sum"]
        assert int(out[ffi.PMX_ROW_SCALARS, 1]) == ref["ncc_reverse_sum"]
        np.testing.assert_array_equal(out[ffi.PMX_ROW_NCC_CCBINS, :S + 1].astype(np.int64), ref["ncc_ccbins"])
    if has_m:
        np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_FSUM].astype(np.int64), ref["mscc_forward_sum"])
        np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_RSUM].astype(np.int64), ref["mscc_reverse_sum"])
        np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_CCBINS].astype(np.int64), ref["mscc_ccbins"])
        np.testing.assert_array_equal(out[ffi.PMX_ROW_MLEN].astype(np.int64), ref["mappable_len_by_shift"])


CASES = [
    # seed, chrom_len, S, L, fd, rd, with_m
    (1, 5000, 100, 36, 0.02, 0.02, True),
    (2, 70000, 300, 36, 0.01, 0.01, True),
    (3, 200000, 1000, 36, 0.005, 0.005, True),
    (4, 33000, 64, 50, 0.3, 0.3, True),        # dense reads
    (5, 40000, 255, 100, 0.01, 0.02, True),    # S < 2L-1
    (6, 40000, 1023, 20, 0.01, 0.02, False),   # NCC only
    (7, 900, 700, 36, 0.05, 0.05, True),       # shift range ~ chromosome length
    (8, 65536 - 36 - 300 - 100, 300, 36, 0.01, 0.01, True),  # nbits multiple of 64
    (9, 120000, 2500, 100, 0.01, 0.01, True),   # max_shift > 1023: shift chunks (BASELINE config 5 shape)
    (10, 90000, 5000, 100, 0.005, 0.02, True),  # -d 5000
    (11, 70000, 1024, 36, 0.01, 0.01, False),   # first shift of the second chunk, NCC only
    (12, 40000, 3000, 1024, 0.02, 0.01, True),  # longest supported read length
    (13, 150000, 65535, 36, 0.004, 0.004, True),   # largest max_shift of the ABI: 64 chunks of 1024 shifts
    (14, 20000, 40000, 50, 0.02, 0.02, True),   # shift range longer than the chromosome
]


@pytest.mark.parametrize("flags", [ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE, 0])
@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_calc_correlation_matches_oracle(ctx, case, flags):
    seed, clen, S, L, fd, rd, with_m = case
    nbits, F, R, M = synth.make_case(seed, clen, S, L, fd, rd, with_m)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
    check_block(out, ref, S, with_m)
    # the path that actually ran is reported, so a silent fallback cannot pass for the other kernel
    want = ffi.PMX_PATH_DENSE if flags == ffi.PMX_FLAG_FORCE_DENSE else ffi.PMX_PATH_SPARSE
    assert int(out[ffi.PMX_ROW_SCALARS, 3]) == want
    if with_m:
        assert int(out[ffi.PMX_ROW_SCALARS, 2]) == int(oracle.lib().pmo_count(oracle._p(M), M.size))


def test_skip_ncc(ctx):
    nbits, F, R, M = synth.make_case(11, 30000, 200, 36)
    ref = oracle.calc_correlation(F, R, M, nbits, 200, 36, skip_ncc=True)
    out = ctx.calc_correlation(F, R, M, nbits, 200, 36, ffi.PMX_FLAG_SKIP_NCC)
    check_block(out, ref, 200, True, skip_ncc=True)
    assert not out[ffi.PMX_ROW_NCC_CCBINS].any()


@pytest.mark.parametrize("max_shift", [300, 1023, 1024, 4000])
@pytest.mark.parametrize("flags", [ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE, 0])
def test_mappable_len_readless(ctx, flags, max_shift):
    nbits, _, _, M = synth.make_case(12, 50000, max_shift, 36)
    ref = oracle.mappable_len_readless(M, nbits, max_shift)
    out = ctx.mappable_len(M, nbits, max_shift, flags)
    np.testing.assert_array_equal(out.astype(np.int64), ref)


def test_empty_vectors(ctx):
    nbits = 10000
    z = np.zeros(synth.nwords(nbits), dtype=np.uint64)
    out = ctx.calc_correlation(z, z, z, nbits, 100, 36)
    assert not out[:ffi.PMX_ROW_SCALARS].any()


def test_bits_builders_match_oracle(ctx):
    rng = np.random.default_rng(5)
    nbits = 100000
    pos = rng.integers(0, nbits, size=5000)
    d = ctx.bits_alloc(nbits)
    ctx.bits_set_positions(d, nbits, pos)
    got = ctx.bits_download(d, nbits)
    np.testing.assert_array_equal(got, oracle.bits_from_positions(pos, nbits))
    assert ctx.bits_count(d, nbits) == int(oracle.lib().pmo_count(oracle._p(got), got.size))
    ctx.bits_clear(d, nbits)
    starts = np.sort(rng.integers(0, nbits - 3000, size=200))
    lens = rng.integers(1, 2500, size=200)
    iv = [(int(s), int(s + l)) for s, l in zip(starts, lens)]       # (begin, end) as BigWig intervals
    ctx.bits_set_regions(d, nbits, np.array([b + 1 for b, e in iv]), np.array([e for b, e in iv]))
    got = ctx.bits_download(d, nbits)
    np.testing.assert_array_equal(got, oracle.bits_from_intervals(iv, nbits))
    ctx.bits_free(d)
    with pytest.raises(ffi.PmxError):
        d = ctx.bits_alloc(100)
        try:
            ctx.bits_set_positions(d, 100, np.array([5, 100]))
        finally:
            ctx.bits_free(d)
