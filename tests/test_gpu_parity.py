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


# bits anywhere in [0, nbits): reads longer than read_len, a track longer than the BAM's chromosome, bit 0, bit nbits-1
FULL_RANGE_CASES = [
    # seed, chrom_len, S, L, fd, rd, with_m, mean_on, mean_off
    (21, 32768 - 1000 - 36 - 100 + 7, 1000, 36, 0.01, 0.01, True, 300, 80),    # the vector ends 7 bits into a new tile
    (22, 65536 - 300 - 36 - 100, 300, 36, 0.02, 0.02, True, 2000, 500),        # nbits = 65536: full last dword, sparse edges
    (23, 70001, 1000, 50, 0.005, 0.02, True, 30, 5),                            # partial last dword, dense edges
    (24, 40000, 2500, 100, 0.01, 0.01, True, 300, 80),                          # shift chunks
    (25, 90011, 1023, 36, 0.02, 0.01, False, 300, 80),                          # NCC-only instantiation
    (26, 131072 + 31, 5000, 100, 0.01, 0.01, False, 300, 80),                   # NCC-only, shift chunks
    (27, 500, 300, 36, 0.3, 0.3, True, 5, 5),                                   # everything inside one tile
]


@pytest.mark.parametrize("flags", [ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE, 0])
@pytest.mark.parametrize("case", FULL_RANGE_CASES, ids=[f"seed{c[0]}" for c in FULL_RANGE_CASES])
def test_bits_anywhere_in_the_vector(ctx, case, flags):
    seed, clen, S, L, fd, rd, with_m, mean_on, mean_off = case
    nbits, F, R, M = synth.make_case(seed, clen, S, L, fd, rd, with_m, mean_on=mean_on, mean_off=mean_off,
                                     full_range=True)
    for w in (F, R) + ((M,) if with_m else ()):
        assert (int(w[(nbits - 1) >> 6]) >> ((nbits - 1) & 63)) & 1 and int(w[0]) & 1
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
    check_block(out, ref, S, with_m)


@pytest.mark.parametrize("with_m", [True, False])
@pytest.mark.parametrize("read_len", [1025, 1500])
def test_reads_longer_than_1024_take_the_dense_kernels(ctx, read_len, with_m):
    """read_len > 1024: beyond the staged halo of the set-bit kernels; the default path must pick the dense kernels
    (and say so), FORCE_SPARSE must refuse."""
    S = 300
    nbits, F, R, M = synth.make_case(31, 60000, S, read_len, 0.01, 0.01, with_m, full_range=True)
    ref = oracle.calc_correlation(F, R, M, nbits, S, read_len)
    out = ctx.calc_correlation(F, R, M, nbits, S, read_len, 0)
    check_block(out, ref, S, with_m)
    assert int(out[ffi.PMX_ROW_SCALARS, 3]) == ffi.PMX_PATH_DENSE
    with pytest.raises(ffi.PmxError):
        ctx.calc_correlation(F, R, M, nbits, S, read_len, ffi.PMX_FLAG_FORCE_SPARSE)


@pytest.mark.parametrize("flags", [ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE, 0])
@pytest.mark.parametrize("max_shift", [0, 1, 2])
def test_max_shift_below_three(ctx, max_shift, flags):
    """The reference takes any -d (mscc.pyx:288); the kernels need 3 shifts for their scalar row, the ABI pads
    internally and cuts the scalar row to max_shift + 1 entries (include/pymasc_amd.h)."""
    L = 36
    nbits, F, R, M = synth.make_case(41, 50000, max_shift, L, 0.02, 0.02, True, full_range=True)
    ref = oracle.calc_correlation(F, R, M, nbits, max_shift, L)
    out = ctx.calc_correlation(F, R, M, nbits, max_shift, L, flags)
    assert out.shape == (ffi.PMX_NROWS, max_shift + 1)
    np.testing.assert_array_equal(out[ffi.PMX_ROW_NCC_CCBINS].astype(np.int64), ref["ncc_ccbins"])
    np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_FSUM].astype(np.int64), ref["mscc_forward_sum"])
    np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_RSUM].astype(np.int64), ref["mscc_reverse_sum"])
    np.testing.assert_array_equal(out[ffi.PMX_ROW_MSCC_CCBINS].astype(np.int64), ref["mscc_ccbins"])
    np.testing.assert_array_equal(out[ffi.PMX_ROW_MLEN].astype(np.int64), ref["mappable_len_by_shift"])
    scal = [ref["ncc_forward_sum"], ref["ncc_reverse_sum"], int(oracle.lib().pmo_count(oracle._p(M), M.size))]
    assert [int(x) for x in out[ffi.PMX_ROW_SCALARS]] == scal[:max_shift + 1]


def test_max_shift_limit_is_an_error_not_a_wrong_answer(ctx):
    nbits, F, R, M = synth.make_case(42, 5000, 100, 36)
    with pytest.raises(ffi.PmxError):
        ctx.calc_correlation(F, R, M, nbits, 65536, 36, 0)


def test_mappable_len_sparse_and_dense_edge_regions_in_one_track(ctx):
    """The autocorrelation runs a pair-enumeration pass over sparse-edge tiles and the window kernel over the tiles
    it flags as dense: a track with both kinds of regions (and a dense region straddling tile boundaries) must add up."""
    S, L, G = 1000, 36, 700000
    nbits = G + L + S + 100
    rng = np.random.default_rng(77)
    M = synth.run_bits(rng, nbits, 3000, 900, 1, G + 1)                       # long runs: ~35 edges per 64-Kbit tile
    for lo, hi in ((60000, 140000), (300000, 310000), (520000, 660000)):     # dense-edge islands
        island = synth.run_bits(rng, nbits, 6, 4, lo, hi)
        keep = synth.run_bits(np.random.default_rng(1), nbits, 10**9, 1, lo, hi)   # ones on [lo, hi)
        M = (M & ~keep) | island
    ref = oracle.mappable_len_readless(M, nbits, S)
    np.testing.assert_array_equal(ctx.mappable_len(M, nbits, S, 0).astype(np.int64), ref)
    nb2, F, R, _ = synth.make_case(5, G, S, L, 0.004, 0.004, False)
    assert nb2 == nbits
    out = ctx.calc_correlation(F, R, M, nbits, S, L, 0)
    check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


@pytest.mark.parametrize("mean_on,mean_off", [(800, 224), (400, 112), (1500, 548)])
@pytest.mark.parametrize("max_shift", [1000, 4000])
def test_mappable_len_around_the_dense_tile_threshold(ctx, mean_on, mean_off, max_shift):
    """Run periods of ~1024 / 512 / 2048 bits put the 128-Kbit tiles of the pair pass right at, above and below its
    256-edge threshold (geometric run lengths: neighbouring tiles fall on different sides); every lag must add up
    whichever pass took a tile.  max_shift 4000: histograms of 4 K lags per workgroup, window kernel in lag chunks."""
    G = 1_500_000
    nbits = G + 36 + max_shift + 100
    M = synth.run_bits(np.random.default_rng(1000 + mean_on + max_shift), nbits, mean_on, mean_off, 1, nbits)
    ref = oracle.mappable_len_readless(M, nbits, max_shift)
    np.testing.assert_array_equal(ctx.mappable_len(M, nbits, max_shift, 0).astype(np.int64), ref)


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


@pytest.mark.parametrize("max_lag", [7900, 8191, 8192])
def test_mappable_len_at_the_largest_pair_pass_lags(ctx, max_lag):
    """The pair pass keeps 2 x (max_lag + 1) histogram words in LDS: 64 KB is passed at ~7870 lags, 8191 is its last lag
    (8192: the window kernel in lag chunks takes over); with reads (calc_correlation) and without (mappable_len)."""
    L = 36
    nbits, F, R, M = synth.make_case(max_lag, 400000, max_lag, L, 0.003, 0.003, True, mean_on=2500, mean_off=600)
    ref = oracle.mappable_len_readless(M, nbits, max_lag)
    np.testing.assert_array_equal(ctx.mappable_len(M, nbits, max_lag, 0).astype(np.int64), ref)
    S = max_lag + L - 1        # lags |L - 1 - d| up to max_lag
    if S <= 65535:
        nb2, F2, R2, M2 = synth.make_case(max_lag + 1, 150000, S, L, 0.003, 0.003, True, mean_on=2500, mean_off=600)
        check_block(ctx.calc_correlation(F2, R2, M2, nb2, S, L, 0), oracle.calc_correlation(F2, R2, M2, nb2, S, L), S, True)


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


# ---- the event kernel (sparse tiles) and its hand-over to the window kernel (dense tiles) ----------------------------
EVENT_TILE = 65536      # kernels_events.h: EV_TB
EVENT_CAP_E = 1536      # EV_CAPE_SMALL: run edges of everything staged for the tile (max_shift <= 1023)
EVENT_POOL_M = 2416                     # EV_POOL_SMALL: forward reads + reverse reads (tile + max_shift bits above) + run edges, with a track
EVENT_POOL_NCC = 768 + 1000 + EVENT_CAP_E   # ... NCC only (EV_POOL_ENTRIES)
EVENT_POOL_DEEP = 4328                  # EV_POOL_DEEP: with a track and PMX_FLAG_DEEP_LISTS


def _exact_count_bits(rng, nbits, lo, hi, k):
    w = np.zeros(synth.nwords(nbits), dtype=np.uint64)
    pos = rng.choice(hi - lo, size=k, replace=False).astype(np.int64) + lo
    np.bitwise_or.at(w, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
    return w


@pytest.mark.parametrize("deep", [False, True])
@pytest.mark.parametrize("n_edges", [None, 0, 300])
@pytest.mark.parametrize("n_f,over", [(1200, 0), (1200, 1), (2050, 0), (2050, 1), (40, 0), (40, 1)])
def test_event_lists_exactly_full_and_one_over(ctx, n_f, over, n_edges, deep):
    """The three lists of a tile share one pool.  Tile 1 of three holds n_f forward reads, n_edges run edges (None: no
    mappability track) and as many reverse reads as fill the pool exactly (stays on the event kernel) or one more
    (flagged, summed by the window kernel and ADDED to what the event kernel wrote for tiles 0 and 2)."""
    S, L = 700, 36
    rng = np.random.default_rng(700 + n_f + over + (n_edges or 7))
    nbits = 3 * EVENT_TILE - 1000
    sparse = lambda: synth.random_bits(rng, nbits, 0.003, 1, nbits - 200)
    F, R = sparse(), sparse()
    lo, hi = EVENT_TILE, 2 * EVENT_TILE
    # clear tile 1 (and, for R, the max_shift bits above it, which count towards its list), then place the bits in it
    for target, clear_hi in ((F, hi), (R, hi + 1024)):
        for wd in range(lo // 64, clear_hi // 64):
            target[wd] = 0
    if deep and n_edges is None:
        pytest.skip("PMX_FLAG_DEEP_LISTS selects an instantiation of the kernel with a track only")
    pool = EVENT_POOL_NCC if n_edges is None else (EVENT_POOL_DEEP if deep else EVENT_POOL_M)
    n_r = pool - n_f - (n_edges or 0) + over
    F |= _exact_count_bits(rng, nbits, lo, hi, n_f)
    R |= _exact_count_bits(rng, nbits, lo, hi, n_r)
    M = None
    if n_edges is not None:
        # mappable throughout what is staged for tile 1 (2048 bits below it, 1152 above) except n_edges / 2 holes inside it
        bits = np.zeros(synth.nwords(nbits) * 64, dtype=np.uint8)
        bits[:nbits] = np.unpackbits(synth.run_bits(rng, nbits, 3000, 900, 1, nbits - 300).view(np.uint8), bitorder="little")[:nbits]
        bits[lo - 4096:hi + 4096] = 1
        for i in range(n_edges // 2):
            bits[lo + 50 + 200 * i:lo + 90 + 200 * i] = 0
        M = np.packbits(bits, bitorder="little").view(np.uint64).copy()
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE | (ffi.PMX_FLAG_DEEP_LISTS if deep else 0))
    check_block(out, ref, S, M is not None)


@pytest.mark.parametrize("n_edges", [EVENT_CAP_E - 2, EVENT_CAP_E, EVENT_CAP_E + 2])
def test_event_edge_list_around_its_capacity(ctx, n_edges):
    """Mappability runs placed so that the edges staged for tile 1 (the tile, 2048 bits below, 1152 above) number
    n_edges: at most the capacity -> event kernel, more -> window kernel for that tile only."""
    S, L = 500, 50
    rng = np.random.default_rng(n_edges)
    nbits = 3 * EVENT_TILE
    F = synth.random_bits(rng, nbits, 0.004, 1, nbits - 700)
    R = synth.random_bits(rng, nbits, 0.004, 1, nbits - 700)
    bits = np.zeros(nbits, dtype=np.uint8)
    # n_edges / 2 runs of 40 bits, 100 bits apart, from the start of tile 1 on; elsewhere two long runs
    for i in range(n_edges // 2):
        p = EVENT_TILE + 10 + 100 * i
        bits[p:p + 40] = 1
    bits[1000:30000] = 1
    bits[2 * EVENT_TILE + 5000:2 * EVENT_TILE + 40000] = 1
    M = np.packbits(bits, bitorder="little").view(np.uint64).copy()
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)
    check_block(out, ref, S, True)


@pytest.mark.parametrize("skip_ncc,deep", [(False, False), (True, False), (False, True)])
def test_dense_and_sparse_tiles_in_one_chromosome(ctx, skip_ncc, deep):
    """Read density 0.05 in the middle tiles, 0.004 elsewhere; mappability edges dense in another region: both kernels
    contribute to every output row of the same result block."""
    S, L = 1000, 36
    rng = np.random.default_rng(99)
    nbits = 6 * EVENT_TILE + 12345
    F = synth.random_bits(rng, nbits, 0.004, 1, nbits - 1200)
    R = synth.random_bits(rng, nbits, 0.004, 1, nbits - 1200)
    F |= synth.random_bits(rng, nbits, 0.05, 2 * EVENT_TILE + 300, 3 * EVENT_TILE + 777)
    R |= synth.random_bits(rng, nbits, 0.05, 3 * EVENT_TILE - 5000, 4 * EVENT_TILE)
    M = synth.run_bits(rng, nbits, 2500, 700, 1, 4 * EVENT_TILE)
    M |= synth.run_bits(rng, nbits, 30, 20, 4 * EVENT_TILE + 100, 5 * EVENT_TILE)
    flags = ffi.PMX_FLAG_FORCE_SPARSE | (ffi.PMX_FLAG_SKIP_NCC if skip_ncc else 0) | (ffi.PMX_FLAG_DEEP_LISTS if deep else 0)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
    check_block(out, ref, S, True, skip_ncc=skip_ncc)


@pytest.mark.parametrize("S,L", [(3, 1), (5, 1024), (1023, 1), (1023, 1024), (64, 700), (700, 64)])
def test_event_kernel_shift_and_read_length_corners(ctx, S, L):
    """Smallest / largest shift range and read length of the event kernel (max_shift <= 1023, read_len <= 1024), and
    read_len - 1 above / below max_shift (the two edge ranges of a reverse read then overlap differently)."""
    rng = np.random.default_rng(S * 2048 + L)
    nbits = 2 * EVENT_TILE + 4321
    F = synth.random_bits(rng, nbits, 0.005, 0, nbits)
    R = synth.random_bits(rng, nbits, 0.005, 0, nbits)
    M = synth.run_bits(rng, nbits, 900, 300, 0, nbits)
    for w in (F, R, M):
        synth.set_bit(w, 0)
        synth.set_bit(w, nbits - 1)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)
    check_block(out, ref, S, True)


def test_window_kernel_alone_matches_the_event_path(ctx):
    """PMX_CC_EVENTS=0 (read once per process) keeps every tile on the window kernel: a child process computes the same
    case that way and both agree with the oracle."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import json, sys, numpy as np\n"
        "from pymasc_amd import ffi\n"
        "from tests import synth\n"
        "nbits, F, R, M = synth.make_case(3, 200000, 1000, 36, 0.005, 0.005, True)\n"
        "c = ffi.Context(0)\n"
        "out = c.calc_correlation(F, R, M, nbits, 1000, 36, ffi.PMX_FLAG_FORCE_SPARSE)\n"
        "print(json.dumps(np.asarray(out).astype(np.int64).tolist()))\n"
    )
    env = dict(os.environ, PMX_CC_EVENTS="0", PMX_AUTOCORR_PAIRS="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    child = np.asarray(json.loads(res.stdout.strip().splitlines()[-1]), dtype=np.int64)
    nbits, F, R, M = synth.make_case(3, 200000, 1000, 36, 0.005, 0.005, True)
    here = np.asarray(ctx.calc_correlation(F, R, M, nbits, 1000, 36, ffi.PMX_FLAG_FORCE_SPARSE)).astype(np.int64)
    np.testing.assert_array_equal(child, here)
    check_block(here, oracle.calc_correlation(F, R, M, nbits, 1000, 36), 1000, True)


@pytest.mark.parametrize("pattern", [0x00000000, 0xffffffff, 0x80808080, 0x7fffffff, 0x5a5a5a5a])
def test_results_do_not_depend_on_stale_scratch_or_lds(ctx, pattern):
    """Every scratch buffer of the context and the LDS of every CU are filled with a pattern before the call: a kernel
    that reads memory it has not written fails here deterministically.  The first geometry is the one a fuzz run caught
    (a dense tile followed by a short sparse tile whose edges sit in the halo and which holds no reverse read: the
    edge-driven loops then walk an EMPTY reverse list and must stop on its sentinel for range starts below zero)."""
    cases = [
        (1846347302, 65536, 1023, 151, 1.0, 0.0005, 30.0, 5.0, False),
        (802336588, 70001, 1023, 151, 0.02, 0.0, 30.0, 5.0, False),
        (698984044, 65536, 33, 36, 0.0005, 0.005, 30.0, 5.0, False),
        (3, 200000, 1000, 36, 0.005, 0.005, 300.0, 80.0, False),
        (1511311730, 65536, 512, 175, 0.0, 0.0, 300.0, 5.0, True),
    ]
    for seed, clen, S, L, fd, rd, on, off, full in cases:
        nbits, F, R, M = synth.make_case(seed, clen, S, L, fd, rd, True, mean_on=on, mean_off=off, full_range=full)
        ref = oracle.calc_correlation(F, R, M, nbits, S, L)
        ctx.debug_poison(pattern)
        out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)
        check_block(out, ref, S, True)
        want_lag = max(L - 1, S - (L - 1) if S > L - 1 else 0)
        ctx.debug_poison(pattern)
        got = ctx.mappable_len(M, nbits, want_lag, 0)
        np.testing.assert_array_equal(got.astype(np.int64), oracle.mappable_len_readless(M, nbits, want_lag).astype(np.int64))
