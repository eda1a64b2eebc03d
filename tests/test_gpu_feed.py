"""GPU parity of the stream-ordered feeders against the oracle's restatement of the reference's per-read rules
(feed_forward_read / feed_reverse_read, mscc.pyx:370-418; _load_mappability, mscc.pyx:327-349): bit-exact vectors,
read-length sums, kept-read counts and the first offending read, for one chunk and for chunk boundaries anywhere."""
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


def make_reads(rng, n, glen, dup_frac=0.3, lens=(36,), sort=True):
    """Sorted reads with plenty of duplicates: repeated positions on both strands, variable read lengths so that
    different (pos, len) pairs hit the same reverse bit."""
    base = rng.integers(1, glen, size=n)
    pick = rng.random(n) < dup_frac
    base[pick] = rng.choice(base[~pick] if (~pick).any() else base, size=int(pick.sum()))
    pos = np.sort(base) if sort else base
    rlen = rng.choice(np.asarray(lens), size=n)
    rev = rng.random(n) < 0.5
    return pos.astype(np.int64), rlen.astype(np.int64), rev


def reference_feed(pos, rlen, rev, S, L, glen):
    oc = oracle.OracleCalculator(S, L, ["c"], [glen])
    for p, l, r in zip(pos.tolist(), rlen.tolist(), rev.tolist()):
        (oc.feed_reverse_read if r else oc.feed_forward_read)("c", p, l)
    return oc._F, oc._R, oc._f_rls, oc._r_rls, oc._nbits


def garbage_vectors(ctx, nbits):
    """F and R as the context's pool may hand them out: full of somebody else's bits (PMX_FEED_WHOLE_VECTORS writes every word)."""
    d_F, d_R = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits)
    junk = np.full(ffi.nwords(nbits), 0xdeadbeefcafef00d, dtype=np.uint64)
    ctx.bits_upload(d_F, junk, nbits)
    ctx.bits_upload(d_R, ~junk, nbits)
    return d_F, d_R


def device_feed(ctx, pos, rlen, rev, nbits, cuts, pdt, ldt, whole=False):
    d_F, d_R = garbage_vectors(ctx, nbits) if whole else (ctx.bits_alloc(nbits), ctx.bits_alloc(nbits))
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    keep = []
    fed = 0
    for a, b in zip([0] + cuts, cuts + [pos.size]):
        if b > a:
            keep.append(ctx.feed_reads(d_F, d_R, nbits, pos[a:b].astype(pdt), rlen[a:b].astype(ldt), rev[a:b], fed, d_st,
                                       whole_vectors=whole and fed == 0))
            fed += b - a
    F, R = ctx.bits_download(d_F, nbits), ctx.bits_download(d_R, nbits)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    return F, R, st


@pytest.mark.parametrize("pdt,ldt", [(np.int32, np.int32), (np.int64, np.int64), (np.int32, np.int64), (np.int64, np.int32),
                                     (np.int32, np.uint16), (np.int64, np.uint16)])
@pytest.mark.parametrize("nchunks", [1, 2, 7])
def test_feed_reads_matches_the_reference_rules(ctx, nchunks, pdt, ldt):
    S, L, glen = 300, 36, 200000
    rng = np.random.default_rng(100 * nchunks + np.dtype(pdt).itemsize + np.dtype(ldt).itemsize)
    pos, rlen, rev = make_reads(rng, 30000, glen, lens=(20, 36, 36, 36, 50, 101))
    pos[:3] = 0                                          # position 0: the reference's _last_forward_pos starts at 0 (a duplicate)
    pos.sort()
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    cuts = sorted(rng.choice(np.arange(1, pos.size), size=nchunks - 1, replace=False).tolist()) if nchunks > 1 else []
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, cuts, pdt, ldt)
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    popc = lambda w: int(oracle.lib().pmo_count(oracle._p(np.ascontiguousarray(w)), w.size))
    assert int(st[ffi.PMX_FEED_FORWARD_KEPT]) == popc(wF) and int(st[ffi.PMX_FEED_REVERSE_KEPT]) == popc(wR)
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
    assert int(st[ffi.PMX_FEED_READS]) == pos.size and int(st[ffi.PMX_FEED_LAST_POS]) == int(pos[-1])


def test_chunk_boundaries_inside_runs_of_equal_positions(ctx):
    """Every cut falls inside a run of reads at one position (both strands, several lengths): the duplicate rules must see
    the reads of the previous chunk -- forward through the carried last forward position, reverse through the vector."""
    S, L, glen = 100, 36, 5000
    rng = np.random.default_rng(7)
    pos = np.sort(rng.integers(1, 60, size=4000)).astype(np.int64) * 50      # 60 positions x ~66 reads
    rlen = rng.choice(np.asarray([30, 36, 36, 40]), size=pos.size).astype(np.int64)
    rev = rng.random(pos.size) < 0.5
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    cuts = list(range(13, pos.size, 97))
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, cuts, np.int32, np.int32)
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr


@pytest.mark.parametrize("pdt", [np.int32, np.int64])
def test_strand_packed_into_the_position_word(ctx, pdt):
    S, L, glen = 200, 36, 150000
    rng = np.random.default_rng(23)
    pos, rlen, rev = make_reads(rng, 25000, glen, lens=(20, 36, 50))
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    packed = ffi.pack_strand(pos.astype(pdt), rev)
    assert packed.dtype == np.dtype(pdt) and (packed < 0).sum() == rev.sum()
    d_F, d_R, d_st = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    k1 = ctx.feed_reads(d_F, d_R, nbits, packed[:11000], rlen[:11000].astype(np.uint16), None, 0, d_st)
    k2 = ctx.feed_reads(d_F, d_R, nbits, packed[11000:], rlen[11000:].astype(np.int32), None, 11000, d_st)
    np.testing.assert_array_equal(ctx.bits_download(d_F, nbits), wF)
    np.testing.assert_array_equal(ctx.bits_download(d_R, nbits), wR)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_LAST_POS]) == int(pos[-1])
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    del k1, k2


def test_uniform_read_length_as_a_scalar(ctx):
    S, L, glen = 200, 36, 100000
    rng = np.random.default_rng(17)
    pos, rlen, rev = make_reads(rng, 20000, glen, lens=(50,))
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    d_F, d_R, d_st = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    k1 = ctx.feed_reads(d_F, d_R, nbits, pos[:9000].astype(np.int32), 50, rev[:9000], 0, d_st)
    k2 = ctx.feed_reads(d_F, d_R, nbits, pos[9000:], 50, rev[9000:], 9000, d_st)
    np.testing.assert_array_equal(ctx.bits_download(d_F, nbits), wF)
    np.testing.assert_array_equal(ctx.bits_download(d_R, nbits), wR)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    del k1, k2


def test_first_unsorted_and_first_out_of_range_read_are_recorded(ctx):
    S, L, glen = 100, 36, 50000
    nbits = glen + L + S + 100
    rng = np.random.default_rng(3)
    pos, rlen, rev = make_reads(rng, 5000, glen)
    bad_sort = 3210
    pos[bad_sort] = pos[bad_sort - 1] - 1                # below its predecessor (mscc.pyx:362-363)
    pos[bad_sort + 1:] = np.maximum(pos[bad_sort + 1:], pos[bad_sort - 1])
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [1000, bad_sort], np.int64, np.int64)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == bad_sort
    pos, rlen, rev = make_reads(rng, 5000, glen)
    pos[-2:] = nbits + 5
    rev[-2:] = False
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [2500], np.int64, np.int64)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == pos.size - 2
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0


@pytest.mark.parametrize("dt,offset", [(np.uint32, 1), (np.int64, 1), (np.int64, 0)])
def test_set_regions_async_matches_oracle(ctx, dt, offset):
    rng = np.random.default_rng(11)
    nbits = 300000
    starts = np.sort(rng.integers(0, nbits - 3000, size=900))
    ends = starts + rng.integers(1, 2500, size=900)
    d = ctx.bits_alloc(nbits)
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    keep = ctx.bits_set_regions_async(d, nbits, starts.astype(dt), ends.astype(dt), offset, d_st)
    got = ctx.bits_download(d, nbits)
    iv = [(int(s) + offset - 1, int(e)) for s, e in zip(starts, ends)]     # bits_from_intervals sets (b + 1 .. e)
    np.testing.assert_array_equal(got, oracle.bits_from_intervals(iv, nbits))
    assert int(ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
    # an interval beyond the vector is clipped and recorded
    ctx.bits_clear(d, nbits)
    keep = ctx.bits_set_regions_async(d, nbits, np.array([10, nbits - 5], dtype=dt), np.array([20, nbits + 50], dtype=dt), 0, d_st)
    got = ctx.bits_download(d, nbits)
    assert ffi.PMX_FEED_ERR_BASE - int(ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 1
    assert (int(got[(nbits - 1) >> 6]) >> ((nbits - 1) & 63)) & 1
    ctx.bits_free(d)
    ctx.bits_free(d_st)
    del keep


def _intervals(rng, nbits, n, max_len, max_gap):
    """n sorted, disjoint (begin, end] BigWig pairs inside [0, nbits): set(begin + 1, end)."""
    gaps = rng.integers(0, max_gap + 1, size=n)          # (gap 0: a run that ends where the next begins)
    lens = rng.integers(1, max_len + 1, size=n)
    begin = np.cumsum(gaps + np.concatenate([[0], lens[:-1]]))
    end = begin + lens
    keep = end < nbits
    return begin[keep], end[keep]


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
@pytest.mark.parametrize("nbits,n,max_len,max_gap", [(300000, 900, 300, 60), (65536 * 3 + 17, 40, 9000, 3), (1000, 30, 20, 20),
                                                     (65536 * 5, 3, 200000, 1000), (64 * 1024, 2000, 3, 40), (4_000_000, 60000, 90, 40)])
def test_a_vector_built_from_sorted_intervals_matches_oracle(ctx, dt, nbits, n, max_len, max_gap):
    """PMX_REGIONS_SORTED (k_regions_build): every word written by its owner over whatever the vector held; runs that touch,
    runs across several workgroups' words, single bits, the vector's first and last bit."""
    rng = np.random.default_rng(nbits % 1000 + n)
    begin, end = _intervals(rng, nbits, n, max_len, max_gap)
    if begin.size:       # ... up to the very last bit, and from the very first (begin + 1 = 0 cannot be said in uint32: bit 1)
        end[-1] = nbits - 1
    d = ctx.bits_alloc(nbits)
    ctx.bits_upload(d, np.full(ffi.nwords(nbits), 0xdeadbeefcafef00d, dtype=np.uint64), nbits)
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    keep = ctx.bits_set_regions_async(d, nbits, begin.astype(dt), end.astype(dt), 1, d_st, sorted_disjoint=True)
    got = ctx.bits_download(d, nbits)
    np.testing.assert_array_equal(got, oracle.bits_from_intervals([(int(b), int(e)) for b, e in zip(begin, end)], nbits))
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    assert int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
    # offset 0 (set(first, last)), bit 0 included
    if dt is np.int64 and begin.size:
        first = begin + 1
        first[0] = 0
        ctx.bits_set_regions_async(d, nbits, first.astype(dt), end.astype(dt), 0, d_st, sorted_disjoint=True)
        want = oracle.bits_from_intervals([(int(a) - 1, int(e)) for a, e in zip(first, end)], nbits)
        np.testing.assert_array_equal(ctx.bits_download(d, nbits), want)
    ctx.bits_free(d)
    ctx.bits_free(d_st)
    del keep


def test_sorted_builder_without_intervals_beyond_the_vector_and_out_of_order(ctx):
    nbits = 200000
    d = ctx.bits_alloc(nbits)
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    junk = np.full(ffi.nwords(nbits), 0xffffffffffffffff, dtype=np.uint64)
    # no intervals: zeros over whatever was there
    ctx.bits_upload(d, junk, nbits)
    ctx.bits_set_regions_async(d, nbits, np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32), 1, d_st, sorted_disjoint=True)
    assert not ctx.bits_download(d, nbits).any()
    # an interval that runs beyond the vector is clipped and recorded, as by the general setter
    ctx.bits_upload(d, junk, nbits)
    b_, e_ = np.array([10, 5000, nbits - 50], dtype=np.uint32), np.array([20, 70000, nbits + 500], dtype=np.uint32)
    ctx.bits_set_regions_async(d, nbits, b_, e_, 1, d_st, sorted_disjoint=True)
    want = oracle.bits_from_intervals([(10, 20), (5000, 70000), (nbits - 50, nbits - 1)], nbits)
    np.testing.assert_array_equal(ctx.bits_download(d, nbits), want)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 2 and int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) == 0
    # order violations are recorded (first index): overlap, wrong order, an empty interval
    for b_, e_, first_bad in (([10, 15, 400], [20, 30, 500], 0), ([300, 10, 600], [350, 20, 700], 0),
                              ([10, 50, 70, 90], [20, 60, 70, 95], 2), ([10, 30], [20, 40], None)):
        ctx.bits_clear(d_st, ffi.PMX_FEED_WORDS * 64)
        ctx.bits_set_regions_async(d, nbits, np.array(b_, dtype=np.uint32), np.array(e_, dtype=np.uint32), 1, d_st, sorted_disjoint=True)
        st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
        if first_bad is None:
            assert int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) == 0
        else:
            assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) == first_bad
    ctx.bits_free(d)
    ctx.bits_free(d_st)


@pytest.mark.parametrize("side", [False, True])
def test_regions_cleared_and_set_beside_the_read_feeders(ctx, side):
    """pmx_bits_set_regions_ex: PMX_REGIONS_CLEAR on vectors full of somebody else's bits, PMX_REGIONS_SIDE while reads are fed to
    other vectors on the context's stream; the first other entry point (here: the download) waits for the side stream."""
    rng = np.random.default_rng(19)
    glen, S, L = 900000, 300, 36
    nbits = glen + L + S + 100
    tracks, keep, fed = [], [], []
    for k in range(6):
        d_M = ctx.bits_alloc(nbits)
        ctx.bits_upload(d_M, np.full(ffi.nwords(nbits), 0xdeadbeefcafef00d, dtype=np.uint64), nbits)
        n_iv = 0 if k == 3 else 4000                       # (no intervals at all: the vector is still cleared)
        starts = np.sort(rng.integers(0, nbits - 300, size=n_iv))
        ends = starts + rng.integers(1, 250, size=n_iv)
        if k % 2:     # (the general setter: intervals in any order, overlapping)
            keep.append(ctx.bits_set_regions_async(d_M, nbits, starts.astype(np.uint32), ends.astype(np.uint32), 1, None, clear=True, side=side))
        else:         # (the builder: BigWig order)
            starts, ends = _intervals(rng, nbits, n_iv, 250, 200)
            d_ms = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
            keep.append(ctx.bits_set_regions_async(d_M, nbits, starts.astype(np.uint32), ends.astype(np.uint32), 1, d_ms, side=side,
                                                   sorted_disjoint=True))
        tracks.append((d_M, starts, ends))
        # ... and a chromosome's reads on the context's own stream meanwhile
        pos, rlen, rev = make_reads(rng, 20000, glen, lens=(L,))
        d_F, d_R = garbage_vectors(ctx, nbits)
        d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
        keep.append(ctx.feed_reads(d_F, d_R, nbits, pos.astype(np.int32), rlen.astype(np.int32), rev, 0, d_st, whole_vectors=True))
        fed.append((d_F, d_R, d_st, pos, rlen, rev))
    for d_M, starts, ends in tracks:
        got = ctx.bits_download(d_M, nbits)
        np.testing.assert_array_equal(got, oracle.bits_from_intervals([(int(s), int(e)) for s, e in zip(starts, ends)], nbits))
        ctx.bits_free(d_M)
    for d_F, d_R, d_st, pos, rlen, rev in fed:
        F, R, _fr, _rr, _nb = reference_feed(pos, rlen, rev, S, L, glen)
        np.testing.assert_array_equal(ctx.bits_download(d_F, nbits), F)
        np.testing.assert_array_equal(ctx.bits_download(d_R, nbits), R)
        for d in (d_F, d_R, d_st):
            ctx.bits_free(d)
    del keep


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
def test_build_batch_matches_oracle_and_reports_range_errors(ctx, dt):
    rng = np.random.default_rng(21)
    jobs, want, ptrs = [], [], []
    for k, nbits in enumerate([70001, 65536, 300, 123457]):
        fpos = np.sort(rng.integers(0, nbits, size=nbits // 50))
        rpos = np.sort(rng.integers(0, nbits, size=nbits // 40))
        starts = np.sort(rng.integers(0, max(nbits - 400, 1), size=max(nbits // 500, 1)))
        ends = np.minimum(starts + rng.integers(0, 300, size=starts.size), nbits - 1)
        dF, dR, dM = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(nbits)
        ctx.bits_set_positions(dF, nbits, np.arange(0, nbits, 7))       # stale content: the batch call clears
        jobs.append((dF, dR, dM if k != 2 else None, nbits, fpos, rpos, starts if k != 2 else None, ends if k != 2 else None))
        ptrs.append((dF, dR, dM))
        want.append((oracle.bits_from_positions(fpos, nbits), oracle.bits_from_positions(rpos, nbits),
                     oracle.bits_from_intervals([(int(s) - 1, int(e)) for s, e in zip(starts, ends)], nbits) if k != 2 else None, nbits))
    keep = ctx.bits_build_batch(jobs, dt)
    ctx.bits_build_status()
    for (dF, dR, dM), (wF, wR, wM, nbits) in zip(ptrs, want):
        np.testing.assert_array_equal(ctx.bits_download(dF, nbits), wF)
        np.testing.assert_array_equal(ctx.bits_download(dR, nbits), wR)
        if wM is not None:
            np.testing.assert_array_equal(ctx.bits_download(dM, nbits), wM)
    # a position beyond its vector: reported by the status call, naming the job
    bad = list(jobs[1])
    bad[4] = np.array([5, 65536], dtype=np.int64)
    keep = ctx.bits_build_batch([jobs[0], tuple(bad)], dt)
    with pytest.raises(ffi.PmxError) as e:
        ctx.bits_build_status()
    assert "job 1" in str(e.value)
    ctx.bits_build_status()                               # cleared by the report
    for tri in ptrs:
        for d in tri:
            ctx.bits_free(d)
    del keep


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
@pytest.mark.parametrize("with_m", [True, False])
def test_packed_host_arrays_are_copied_in_one_piece_with_the_same_result(ctx, dt, with_m):
    """Context.host_packed lays the arrays of a chromosome out as the staging slot does (back to back, 16-byte padded):
    the feeders then issue ONE copy.  Sizes that are not multiples of 16 bytes exercise the padding."""
    rng = np.random.default_rng(31 + (1 if with_m else 0))
    jobs, want, ptrs = [], [], []
    for nbits in (70001, 131072 + 13, 901):
        fpos = np.sort(rng.integers(0, nbits, size=nbits // 50 + 1))
        rpos = np.sort(rng.integers(0, nbits, size=nbits // 40 + 3))
        starts = np.sort(rng.integers(0, max(nbits - 400, 1), size=max(nbits // 500, 1) + 2))
        ends = np.minimum(starts + rng.integers(0, 300, size=starts.size), nbits - 1)
        src = [fpos, rpos] + ([starts, ends] if with_m else [])
        dst = ctx.host_packed([a.size for a in src], dt)
        for d_, a in zip(dst, src):
            d_[:] = a
        assert all(int(b.ctypes.data - a.ctypes.data) == ((a.nbytes + 15) & ~15) for a, b in zip(dst, dst[1:]))
        dF, dR, dM = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(nbits) if with_m else None
        jobs.append((dF, dR, dM, nbits, dst[0], dst[1], dst[2] if with_m else None, dst[3] if with_m else None))
        ptrs.append((dF, dR, dM))
        want.append((oracle.bits_from_positions(fpos, nbits), oracle.bits_from_positions(rpos, nbits),
                     oracle.bits_from_intervals([(int(s) - 1, int(e)) for s, e in zip(starts, ends)], nbits) if with_m else None, nbits))
    keep = ctx.bits_build_batch(jobs, dt)
    ctx.bits_build_status()
    for (dF, dR, dM), (wF, wR, wM, nbits) in zip(ptrs, want):
        np.testing.assert_array_equal(ctx.bits_download(dF, nbits), wF)
        np.testing.assert_array_equal(ctx.bits_download(dR, nbits), wR)
        if wM is not None:
            np.testing.assert_array_equal(ctx.bits_download(dM, nbits), wM)
    # the interval feeder alone, from a packed pair
    if with_m:
        nbits = 70001
        starts = np.sort(rng.integers(0, nbits - 400, size=77))
        ends = starts + rng.integers(1, 300, size=77)
        b_, e_ = ctx.host_packed([77, 77], dt)
        b_[:], e_[:] = starts, ends
        d = ctx.bits_alloc(nbits)
        keep2 = ctx.bits_set_regions_async(d, nbits, b_, e_, 1, None)
        np.testing.assert_array_equal(ctx.bits_download(d, nbits),
                                      oracle.bits_from_intervals([(int(s), int(e)) for s, e in zip(starts, ends)], nbits))
        ctx.bits_free(d)
        del keep2
    for dF, dR, dM in ptrs:
        ctx.bits_free(dF)
        ctx.bits_free(dR)
        if dM:
            ctx.bits_free(dM)
    del keep


@pytest.mark.parametrize("max_shift", [300, 2000])
def test_mappable_len_batch_matches_oracle(ctx, max_shift):
    rng = np.random.default_rng(5 + max_shift)
    sizes = [40000, 900, 200003, 65536] + [3000 + 517 * i for i in range(36)]     # more jobs than one job table
    Ms = [synth.run_bits(rng, n, 300, 80, 0, n) for n in sizes]
    dM = [ctx.bits_alloc(n) for n in sizes]
    dO = [ctx.bits_alloc((max_shift + 1) * 64) for _ in sizes]
    for d, M, n in zip(dM, Ms, sizes):
        ctx.bits_upload(d, M, n)
    ctx.mappable_len_batch_dev(dM, sizes, max_shift, 0, dO)
    for d, M, n in zip(dO, Ms, sizes):
        got = ctx.bits_download(d, (max_shift + 1) * 64)
        np.testing.assert_array_equal(got.astype(np.int64), oracle.mappable_len_readless(M, n, max_shift).astype(np.int64))
    for d in dM + dO:
        ctx.bits_free(d)


# ---- the two-bytes-per-read form (pmx_feed_reads_delta16): the same cases through ffi.pack_delta16 --------------------
def device_feed_delta16(ctx, pos, rlen, rev, nbits, cuts, ldt, pinned=False, whole=False):
    d_F, d_R = garbage_vectors(ctx, nbits) if whole else (ctx.bits_alloc(nbits), ctx.bits_alloc(nbits))
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    keep, fed = [], 0
    for a, b in zip([0] + cuts, cuts + [pos.size]):
        if b > a:
            reads = ffi.pack_delta16(pos[a:b], rev[a:b], ctx if pinned else None)
            keep.append(ctx.feed_reads_delta16(d_F, d_R, nbits, reads, rlen[a:b].astype(ldt) if ldt else int(rlen[0]), fed, d_st,
                                               whole_vectors=whole and fed == 0))
            fed += b - a
    F, R = ctx.bits_download(d_F, nbits), ctx.bits_download(d_R, nbits)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    return F, R, st


@pytest.mark.parametrize("ldt", [np.uint16, np.int32, np.int64])
@pytest.mark.parametrize("nchunks", [1, 2, 7])
def test_delta16_feed_matches_the_reference_rules(ctx, nchunks, ldt):
    S, L, glen = 300, 36, 200000
    rng = np.random.default_rng(900 * nchunks + np.dtype(ldt).itemsize)
    pos, rlen, rev = make_reads(rng, 30000, glen, lens=(20, 36, 36, 36, 50, 101))
    pos[:3] = 0
    pos.sort()
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    cuts = sorted(rng.choice(np.arange(1, pos.size), size=nchunks - 1, replace=False).tolist()) if nchunks > 1 else []
    F, R, st = device_feed_delta16(ctx, pos, rlen, rev, nbits, cuts, ldt, pinned=(nchunks == 2))
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
    assert int(st[ffi.PMX_FEED_READS]) == pos.size and int(st[ffi.PMX_FEED_LAST_POS]) == int(pos[-1])


def test_delta16_segments_at_wide_gaps_and_inside_runs_of_equal_positions(ctx):
    """Gaps of 32766 / 32767 / 32768 / 10^6 bp between neighbours (the distance field holds 32766 at most), runs of equal
    positions across segment boundaries (every 1024 reads), chunk cuts inside them, one read length for the run."""
    S, L, glen = 100, 36, 30_000_000
    rng = np.random.default_rng(31)
    parts, at = [], 1
    for gap in (32766, 32767, 32768, 1_000_000, 5, 32767, 70000):
        n = int(rng.integers(3000, 9000))
        p = at + np.sort(rng.integers(0, 4000, size=n))
        p[rng.random(n) < 0.4] = at + 1777                      # a pile-up: thousands of reads at one position
        p = np.sort(p)
        p[0] = at                                                  # (the gap to the part before is exact)
        parts.append(p)
        at = int(parts[-1][-1]) + gap
    pos = np.concatenate(parts).astype(np.int64)
    rlen = np.full(pos.size, 36, dtype=np.int64)
    rev = rng.random(pos.size) < 0.5
    d = np.diff(pos)
    assert set((32766, 32767, 32768, 1_000_000)) <= set(d.tolist())
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    cuts = [1023, 1024, 1025, 4096, 20000]
    F, R, st = device_feed_delta16(ctx, pos, rlen, rev, nbits, cuts, None)
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_LAST_POS]) == int(pos[-1])


def test_delta16_reports_a_run_that_starts_below_the_reads_fed_before_and_a_read_beyond_the_vector(ctx):
    S, L, glen = 100, 36, 50000
    nbits = glen + L + S + 100
    rng = np.random.default_rng(5)
    pos, rlen, rev = make_reads(rng, 6000, glen)
    # second run begins below the end of the first one: its first read is the offender (mscc.pyx:362-363)
    a, b = pos[:3000].copy(), pos[2000:5000].copy()
    d_F, d_R, d_st = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    k1 = ctx.feed_reads_delta16(d_F, d_R, nbits, ffi.pack_delta16(a, rev[:3000]), 36, 0, d_st)
    k2 = ctx.feed_reads_delta16(d_F, d_R, nbits, ffi.pack_delta16(b, rev[2000:5000]), 36, 3000, d_st)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 3000
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    pos[-2:] = nbits + 5
    rev[-2:] = False
    F, R, st = device_feed_delta16(ctx, pos, rlen, rev, nbits, [2500], np.int64)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == pos.size - 2
    with pytest.raises(ffi.PmxError):          # a segment table that does not end with n
        bad = ffi.pack_delta16(pos[:100], rev[:100])
        bad.seg_start[-1] = 99
        ctx.feed_reads_delta16(ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), nbits, bad, 36, 0, ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64))
    del k1, k2


# ---- PMX_FEED_WHOLE_VECTORS: the first run of a chromosome writes every word of uncleared vectors (k_feed_build) ----------
@pytest.mark.parametrize("glen,nreads", [(200000, 30000), (70, 40), (65536 * 3 - 436, 9000), (1_500_000, 200)])
@pytest.mark.parametrize("nchunks", [1, 3])
def test_whole_vector_feed_matches_the_reference_rules(ctx, nchunks, glen, nreads):
    """Vectors full of junk, first run with PMX_FEED_WHOLE_VECTORS (every word written by the workgroup that owns it: words
    without reads too, the tail of the last word, vectors of one workgroup and of many), later runs the ordinary way."""
    S, L = 300, 36
    rng = np.random.default_rng(7000 + nchunks + glen)
    pos, rlen, rev = make_reads(rng, nreads, glen, lens=(20, 36, 36, 36, 50, 101))
    pos[:3] = 0
    pos.sort()
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    cuts = sorted(rng.choice(np.arange(1, pos.size), size=nchunks - 1, replace=False).tolist()) if nchunks > 1 else []
    for fmt in ("pos", "delta16"):
        if fmt == "pos":
            F, R, st = device_feed(ctx, pos, rlen, rev, nbits, cuts, np.int32, np.uint16, whole=True)
        else:
            F, R, st = device_feed_delta16(ctx, pos, rlen, rev, nbits, cuts, np.int32, whole=True)
        np.testing.assert_array_equal(F, wF)
        np.testing.assert_array_equal(R, wR)
        assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
        popc = lambda w: int(oracle.lib().pmo_count(oracle._p(np.ascontiguousarray(w)), w.size))
        assert int(st[ffi.PMX_FEED_FORWARD_KEPT]) == popc(wF) and int(st[ffi.PMX_FEED_REVERSE_KEPT]) == popc(wR)
        assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
        assert int(st[ffi.PMX_FEED_READS]) == pos.size and int(st[ffi.PMX_FEED_LAST_POS]) == int(pos[-1])


def test_whole_vector_feed_pile_ups_and_errors(ctx):
    """Runs of equal positions with mixed strands and lengths (the look-backs cross workgroup boundaries of the word-wise
    deal), a chunk cut inside one, then the first unsorted / first out-of-range read of a whole-vector run."""
    S, L, glen = 100, 36, 500000
    rng = np.random.default_rng(77)
    pos = np.sort(rng.integers(1, 80, size=20000)).astype(np.int64) * 5000 + 65500      # pile-ups next to 64-Kbit boundaries
    rlen = rng.choice(np.asarray([30, 36, 36, 40, 75]), size=pos.size).astype(np.int64)
    rev = rng.random(pos.size) < 0.5
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [9999], np.int64, np.int64, whole=True)
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    pos, rlen, rev = make_reads(rng, 5000, glen)
    bad = 3210
    pos[bad] = pos[bad - 1] - 1
    pos[bad + 1:] = np.maximum(pos[bad + 1:], pos[bad - 1])
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [], np.int32, np.int32, whole=True)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == bad
    pos, rlen, rev = make_reads(rng, 5000, glen)
    pos[-2:] = nbits + 5
    rev[-2:] = False
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [], np.int64, np.uint16, whole=True)
    assert ffi.PMX_FEED_ERR_BASE - int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == pos.size - 2
    assert int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0


@pytest.mark.parametrize("whole", [False, True])
def test_a_pile_up_of_a_hundred_thousand_reads_inside_one_read_length(ctx, whole):
    """chrM / rDNA / satellite style: 1.2 x 10^5 reads within 150 bp, both strands, mixed lengths (soft clips), then ordinary
    reads.  The look-backs of the duplicate rules walked back read by read on ONE lane (K dependent loads per read, K^2 in
    all: round-3 advisor finding); the wavefront serves long walks together now.  Exact, and bounded in time."""
    import time
    S, L, glen = 100, 36, 300000
    rng = np.random.default_rng(123)
    pile = np.sort(rng.integers(5000, 5150, size=120000))
    rest = np.sort(rng.integers(6000, glen, size=20000))
    pos = np.concatenate([pile, rest]).astype(np.int64)
    rlen = rng.choice(np.asarray([30, 36, 36, 50, 75, 101]), size=pos.size).astype(np.int64)
    rev = rng.random(pos.size) < 0.5
    rev[:60000] = rng.random(60000) < 0.97            # a long stretch that is nearly all reverse: forward look-backs over it
    wF, wR, wf, wr, nbits = reference_feed(pos, rlen, rev, S, L, glen)
    ctx.sync()
    t0 = time.perf_counter()
    F, R, st = device_feed(ctx, pos, rlen, rev, nbits, [70000], np.int32, np.uint16, whole=whole)
    dt = time.perf_counter() - t0
    np.testing.assert_array_equal(F, wF)
    np.testing.assert_array_equal(R, wR)
    assert int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == wf and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == wr
    assert dt < 2.0, f"pile-up feed took {dt:.2f} s"
