"""Host logic of CCHipCalculator (feed / dedup / sortedness / flush / lag re-indexing / results) on
CPU, with tests/fake_context.py standing in for the device, compared with the oracle's restatement of
CCBitArrayCalculator and with the reference-generated vectors."""
import json
import os
import pickle

import numpy as np
import pytest

from oracle import model as oracle
from pymasc_amd.calculator import CCHipCalculator
from pymasc_amd.exceptions import ReadUnsortedError
from pymasc_amd import result as R
from . import fixtures as fx
from .fake_context import FakeContext
from .helpers import DictFeeder, assert_matches_oracle, feed_all

CASES = json.load(open(os.path.join(fx.GOLDEN, "ref_successive_ncc.json")))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_ncc_against_reference_vectors(case):
    names = [n for n, _ in case["chroms"]]
    lens = [l for _, l in case["chroms"]]
    calc = CCHipCalculator(case["max_shift"], 36, names, lens, context=FakeContext())
    for name, pos, rev, rl in case["reads"]:
        (calc.feed_reverse_read if rev else calc.feed_forward_read)(name, pos, rl)
    calc.finishup_calculation()
    whole = calc.get_whole_result()
    exp = case["expected"]
    assert isinstance(whole, R.NCCGenomeWideResult)
    assert (whole.genomelen, whole.forward_sum, whole.reverse_sum) == (exp["genomelen"], exp["forward_sum"],
                                                                       exp["reverse_sum"])
    assert set(whole.chroms) == set(exp["chroms"])
    for ch, e in exp["chroms"].items():
        r = whole.chroms[ch]
        assert (r.forward_sum, r.reverse_sum) == (e["forward_sum"], e["reverse_sum"])
        assert (r.forward_read_len_sum, r.reverse_read_len_sum) == (e["forward_read_len_sum"],
                                                                    e["reverse_read_len_sum"])
        assert [int(x) for x in r.ccbins] == e["ccbins"]
        ecc = np.array([np.nan if x is None else x for x in e["cc"]])
        m = ~np.isnan(ecc)
        assert np.array_equal(np.isnan(r.cc), ~m)
        np.testing.assert_allclose(r.cc[m], ecc[m], rtol=0, atol=1e-15)


def _small_mscc_setup(seed=3, skip_ncc=False):
    rng = np.random.default_rng(seed)
    chroms = [("a", 20000), ("b", 12000), ("c", 9000), ("d", 4000)]
    tracks = {}
    for name, ln in chroms[:3]:      # "d" has no mappability track -> KeyError path
        iv, p = [], int(rng.integers(0, 50))
        while p < ln - 400:
            on = int(rng.integers(1, 300))
            iv.append((p, p + on, float(rng.choice([0.5, 1.0, 1.0, 1.0]))))
            p += on + int(rng.integers(1, 120))
        tracks[name] = iv
    reads = []
    for name, ln in [chroms[0], chroms[2], chroms[3]]:   # "b" gets no reads
        n = 1500
        pos = np.sort(rng.integers(1, ln - 80, size=n))
        pos[rng.integers(0, n, 100)] = pos[rng.integers(0, n, 100)]
        pos.sort()
        for p, rv, rl in zip(pos.tolist(), (rng.random(n) < 0.5).tolist(), rng.choice([30, 36, 50], n).tolist()):
            reads.append((rv, name, p, rl))
    names = [n for n, _ in chroms]
    lens = [l for _, l in chroms]
    return names, lens, tracks, reads


@pytest.mark.parametrize("early_batch", [10, 1, 0])
@pytest.mark.parametrize("skip_ncc", [False, True])
def test_both_against_oracle_calculator(skip_ncc, early_batch):
    """early_batch: chromosomes queued before their kernels are launched ahead of the fetch (1: every flush launches, 0: one
    launch at the fetch) -- the same results whichever way."""
    names, lens, tracks, reads = _small_mscc_setup(skip_ncc=skip_ncc)
    S, L = 120, 36
    calc = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(tracks), skip_ncc=skip_ncc, context=FakeContext())
    calc.early_batch = early_batch
    ocalc = oracle.OracleCalculator(S, L, names, lens, mappability={
        c: [x for x in iv if np.float32(x[2]) >= 1] for c, iv in tracks.items()}, skip_ncc=skip_ncc)
    feed_all(calc, reads)
    feed_all(ocalc, reads)
    calc.finishup_calculation()
    ocalc.finishup_calculation()
    assert_matches_oracle(calc, ocalc, names)
    whole = calc.get_whole_result()
    assert isinstance(whole, R.BothGenomeWideResult)
    assert set(whole.chroms) == set(names) and set(whole.mappable_chroms) == set(names)
    # read-less chromosome with a track: lag table of length S+1 (mscc.pyx:207-215); without: zeros
    assert len(whole.mappable_chroms["b"].mappable_len) == S + 1 and sum(whole.mappable_chroms["b"].mappable_len) > 0
    assert isinstance(whole.mappable_chroms["b"], R.EmptyMSCCResult)
    # chromosome with reads but no track: NCC only (mscc.pyx:254-259), placeholder MSCC from _fill_result
    assert isinstance(whole.mappable_chroms["d"], R.EmptyMSCCResult)
    assert np.isnan(whole.mappable_chroms["d"].cc).all()
    # results must survive the worker -> parent queue (handler/worker.py:234)
    back = pickle.loads(pickle.dumps(calc.get_result("a")))
    assert list(back.mappable_chrom.ccbins) == list(calc.get_result("a").mappable_chrom.ccbins)


def test_worker_style_flush_per_chromosome():
    """handler/worker.py:217-234: feed one chromosome, flush(chrom), get_result(chrom); a chromosome
    without reads is flushed too and must come back as placeholders."""
    names, lens, tracks, reads = _small_mscc_setup(seed=9)
    S, L = 64, 36
    calc = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(tracks), context=FakeContext())
    ocalc = oracle.OracleCalculator(S, L, names, lens, mappability={
        c: [x for x in iv if np.float32(x[2]) >= 1] for c, iv in tracks.items()})
    for chrom in names:
        mine = [r for r in reads if r[1] == chrom]
        feed_all(calc, mine)
        feed_all(ocalc, mine)
        calc.flush(chrom)
        ocalc.flush(chrom)
        res = calc.get_result(chrom)
        assert res.chrom is not None
    assert_matches_oracle(calc, ocalc, names)
    agg = R.aggregate_results({c: calc.get_result(c) for c in names})
    assert isinstance(agg, R.BothGenomeWideResult)
    assert agg.genomelen == sum(lens)
    assert agg.forward_sum == calc.forward_sum and agg.reverse_sum == calc.reverse_sum


def test_unsorted_reads_raise():
    calc = CCHipCalculator(50, 36, ["a", "b"], [5000, 5000], context=FakeContext())
    calc.feed_forward_read("a", 100, 36)
    with pytest.raises(ReadUnsortedError):
        calc.feed_reverse_read("a", 99, 36)
    calc = CCHipCalculator(50, 36, ["a", "b"], [5000, 5000], context=FakeContext())
    calc.feed_forward_read("a", 100, 36)
    calc.feed_forward_read("b", 10, 36)
    with pytest.raises(ReadUnsortedError):      # a finished chromosome reappears (mscc.pyx:354-355)
        calc.feed_forward_read("a", 200, 36)
    assert issubclass(ReadUnsortedError, IndexError)


def test_a_read_beyond_the_vector_is_refused_when_it_is_fed():
    """nbits = chromosome length + read_len + max_shift + 100 (mscc.pyx:165-167); the reference sets the bit unchecked.
    The error comes at feed time, before anything of the chromosome is queued, and leaves the calculator usable."""
    S, L, G = 50, 36, 5000
    nbits = G + L + S + 100
    calc = CCHipCalculator(S, L, ["a", "b"], [G, G], context=FakeContext())
    calc.feed_forward_read("a", 100, 36)
    calc.feed_reverse_read("a", nbits - 36, 36)          # reverse bit = pos + readlen - 1 = nbits - 1: the last bit, fine
    with pytest.raises(IndexError):
        calc.feed_reverse_read("a", nbits - 35, 36)
    calc.feed_forward_read("a", nbits - 1, 36)           # last bit of the vector
    with pytest.raises(IndexError):
        calc.feed_forward_read("a", nbits, 36)
    # a chunk fed in bulk is walked by the device only: its out-of-range read (dropped there) is reported when the results
    # of the chromosome are fetched, after everything fetched has been stored
    calc.feed_reads("b", np.array([10, nbits + 5]), np.array([36, 36]), np.array([False, False]))
    with pytest.raises(IndexError):
        calc.finishup_calculation()
    assert calc.get_result("a").chrom.forward_sum == 2 and calc.get_result("b").chrom.forward_sum == 1


def test_an_unsorted_read_inside_a_bulk_chunk_is_reported_when_results_are_fetched():
    calc = CCHipCalculator(50, 36, ["a", "b"], [5000, 5000], context=FakeContext())
    calc.feed_reads("a", np.array([10, 300, 200, 400]), np.array([36] * 4), np.array([False, True, False, True]))
    with pytest.raises(ReadUnsortedError):
        calc.flush("a")
        calc.get_result("a")
    calc = CCHipCalculator(50, 36, ["a", "b"], [5000, 5000], context=FakeContext())
    calc.feed_reads("a", np.array([10, 300]), np.array([36] * 2), np.array([False, True]))
    with pytest.raises(ReadUnsortedError):       # the first read of a chunk against the reads fed before: at once
        calc.feed_reads("a", np.array([299, 400]), np.array([36] * 2), np.array([False, True]))


def test_get_result_unknown_chrom_is_keyerror():
    calc = CCHipCalculator(50, 36, ["a"], [5000], context=FakeContext())
    with pytest.raises(KeyError):
        calc.get_result("zzz")


def test_bulk_feed_equals_per_read_feed():
    names, lens, tracks, reads = _small_mscc_setup(seed=21)
    S, L = 80, 36
    one = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(tracks), context=FakeContext())
    two = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(tracks), context=FakeContext())
    feed_all(one, reads)
    for chrom in names:
        mine = [r for r in reads if r[1] == chrom]
        if mine:
            two.feed_reads(chrom, np.array([r[2] for r in mine]), np.array([r[3] for r in mine]),
                           np.array([r[0] for r in mine]))
    one.finishup_calculation()
    two.finishup_calculation()
    for c in names:
        a, b = one.get_result(c), two.get_result(c)
        assert list(a.chrom.ccbins) == list(b.chrom.ccbins) and a.chrom.forward_read_len_sum == b.chrom.forward_read_len_sum
        assert list(a.mappable_chrom.ccbins) == list(b.mappable_chrom.ccbins)


def test_calc_cc_formulas():
    r = R.NCCResult(max_shift=3, read_len=5, genomelen=1000, forward_sum=40, reverse_sum=50,
                    forward_read_len_sum=0, reverse_read_len_sum=0, ccbins=[5, 4, 3, 2])
    r.calc_cc()
    np.testing.assert_array_equal(r.cc, oracle.ncc_cc(40, 50, [5, 4, 3, 2], 1000, 3))
    e = R.EmptyNCCResult.create_empty(1000, 3, 5)
    assert np.isnan(e.cc).all() and e.ccbins == [0.0] * 4
    m = R.MSCCResult(max_shift=5, read_len=3, genomelen=1000, forward_sum=[9, 8, 7, 6, 5, 4],
                     reverse_sum=[5, 6, 7, 8, 9, 9], forward_read_len_sum=0, reverse_read_len_sum=0,
                     ccbins=[1, 2, 3, 2, 1, 0], mappable_len=[500, 490, 480, 470])
    m.calc_cc()
    np.testing.assert_array_equal(m.cc, oracle.mscc_cc([9, 8, 7, 6, 5, 4], [5, 6, 7, 8, 9, 9],
                                                       [1, 2, 3, 2, 1, 0], [500, 490, 480, 470], 5, 3))


def _check_tiny_shift_ranges(context_factory):
    """max_shift 0..2 (and shift ranges shorter than the read): legal in the reference; the kernels are run with
    three shifts and the rows cut back (pymasc_amd/calculator.py)."""
    names, lens, tracks, reads = _small_mscc_setup(seed=21)
    for S in (0, 1, 2, 5, 35, 36, 70, 71, 72):
        ctx = context_factory()
        calc = CCHipCalculator(S, 36, names, lens, bwfeeder=DictFeeder(tracks), context=ctx)
        ocalc = oracle.OracleCalculator(S, 36, names, lens, mappability={
            c: [x for x in iv if np.float32(x[2]) >= 1] for c, iv in tracks.items()})
        feed_all(calc, reads)
        feed_all(ocalc, reads)
        calc.finishup_calculation()
        ocalc.finishup_calculation()
        assert_matches_oracle(calc, ocalc, names)
        r = calc.get_result("a")
        assert len(r.chrom.ccbins) == S + 1 and len(r.chrom.cc) == S + 1
        assert len(r.mappable_chrom.ccbins) == S + 1 and len(r.mappable_chrom.forward_sum) == S + 1
        ctx.close()


def test_bulk_feed_with_packed_strand_and_scalar_read_length():
    """feed_reads(chrom, pos, readlen, None): the strand in the top bit of every position (ffi.pack_strand), and ONE read
    length for the chunk -- the leanest form a reader can hand over -- give the results of the per-read protocol."""
    from pymasc_amd import ffi
    rng = np.random.default_rng(4)
    names, lens = ["a", "b"], [30000, 9000]
    one = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    two = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    for chrom, glen in zip(names, lens):
        pos = np.sort(rng.integers(1, glen, size=800)).astype(np.int32)
        rev = rng.random(800) < 0.5
        for p, r in zip(pos.tolist(), rev.tolist()):
            (one.feed_reverse_read if r else one.feed_forward_read)(chrom, p, 36)
        packed = ffi.pack_strand(pos, rev)
        two.feed_reads(chrom, packed[:300], 36, None)
        two.feed_reads(chrom, packed[300:], 36, None)
    one.finishup_calculation()
    two.finishup_calculation()
    for c in names:
        a, b = one.get_result(c).chrom, two.get_result(c).chrom
        assert list(a.ccbins) == list(b.ccbins) and (a.forward_sum, a.reverse_sum) == (b.forward_sum, b.reverse_sum)
        assert (a.forward_read_len_sum, a.reverse_read_len_sum) == (b.forward_read_len_sum, b.reverse_read_len_sum)


def test_calc_cc_batch_is_bit_identical_to_calc_cc():
    """Results of a fetch get their cc curves in one vectorised pass (result.calc_cc_batch): the same float64 operations
    as NCCResult.calc_cc / MSCCResult.calc_cc element by element, all-zero bins -> NaN included."""
    import copy
    rng = np.random.default_rng(1)
    ncc, mscc = [], []
    for k, (S, L) in enumerate([(300, 36), (300, 36), (300, 36), (20, 36), (100, 50), (300, 36)]):
        bins = rng.integers(0, 50, S + 1).tolist() if k != 2 else [0] * (S + 1)
        ncc.append(R.NCCResult(max_shift=S, read_len=L, genomelen=10**6 + k, forward_sum=5000 + k, reverse_sum=4000,
                               forward_read_len_sum=1, reverse_read_len_sum=1, ccbins=bins))
        by = rng.integers(1000, 5000, S + 1).tolist()
        head = by[:min(L, S + 1)][::-1]
        mlen = [None] * (L - len(head)) + head + (by[2 * L - 1:] if S >= 2 * L - 1 else [])
        mscc.append(R.MSCCResult(max_shift=S, read_len=L, genomelen=10**6, forward_sum=rng.integers(0, 900, S + 1).tolist(),
                                 reverse_sum=rng.integers(0, 900, S + 1).tolist(), forward_read_len_sum=1, reverse_read_len_sum=1,
                                 ccbins=bins, mappable_len=mlen))
    one = copy.deepcopy(ncc) + copy.deepcopy(mscc)
    for r in one:
        r.calc_cc()
    R.calc_cc_batch(ncc, mscc)
    for a, b in zip(one, ncc + mscc):
        assert a.cc.dtype == b.cc.dtype == np.float64 and np.array_equal(a.cc, b.cc, equal_nan=True)
    # ... and from the integer rows as arrays (what the calculator holds when it fetches a genome's results)
    same = [k for k in (0, 1, 2, 5)]
    n2, m2 = [copy.deepcopy(ncc[k]) for k in same], [copy.deepcopy(mscc[k]) for k in same]
    for r in n2 + m2:
        r.cc = None
    by_shift = [[m.mappable_len[abs(d - 35)] for d in range(301)] for m in m2]
    R.calc_cc_batch(n2, m2, np.array([r.ccbins for r in n2]),
                    (np.array([r.ccbins for r in m2]), np.array([r.forward_sum for r in m2]),
                     np.array([r.reverse_sum for r in m2]), np.array(by_shift)))
    for k, a, b in zip(same, n2, m2):
        assert np.array_equal(a.cc, one[k].cc, equal_nan=True) and np.array_equal(b.cc, one[len(ncc) + k].cc, equal_nan=True)


def test_tiny_shift_ranges_host():
    _check_tiny_shift_ranges(FakeContext)


@pytest.mark.gpu
def test_tiny_shift_ranges_gpu():
    from pymasc_amd import ffi
    _check_tiny_shift_ranges(lambda: ffi.Context(0))


def test_dense_data_hint_from_the_counts_the_calculator_holds():
    """PMX_FLAG_WINDOW_ONLY (skip the event kernel) is passed for a chromosome whose reads or mappable runs are too dense
    for its lists, and only for that one; results are the oracle's either way."""
    from pymasc_amd import calculator as C, ffi

    class Recording(FakeContext):
        def __init__(self):
            super().__init__()
            self.flags = []

        def cc_dev(self, d_F, d_R, d_M, nbits, max_shift, read_len, flags, d_out):
            self.flags.append(int(flags))
            return super().cc_dev(d_F, d_R, d_M, nbits, max_shift, read_len, flags, d_out)

    names, lens = ["sparse", "deep", "deeper", "shortruns"], [6000, 6000, 6000, 6000]
    rng = np.random.default_rng(5)
    tracks = {"sparse": [(100, 5000, 1.0)], "deep": [(100, 5000, 1.0)], "deeper": [(100, 5000, 1.0)],
              "shortruns": [(b, b + 20, 1.0) for b in range(100, 5900, 50)]}           # 116 runs / 6000 bp: 2500 edges per 64 Kbit
    assert not C.window_only_hint(30, 30, 1, 6000, 100) and C.window_only_hint(30, 30, 116, 6000, 100)
    assert not C.window_only_hint(300, 300, 0, 65536, 100) and C.window_only_hint(1700, 1700, 0, 65536, 100)
    # with a track: the ordinary pool (2416 entries per 64 Kbit) -> the DEEP instantiation (4328) -> the window kernels
    assert not C.deep_lists_hint(900, 900, 30, 65536, 100) and C.deep_lists_hint(1100, 1100, 30, 65536, 100)
    assert C.deep_lists_hint(2000, 2000, 30, 65536, 100) and not C.window_only_hint(2000, 2000, 30, 65536, 100)
    assert C.window_only_hint(2200, 2200, 30, 65536, 100) and not C.deep_lists_hint(2200, 2200, 30, 65536, 100)
    assert not C.deep_lists_hint(1100, 1100, 0, 65536, 100) and not C.deep_lists_hint(1100, 1100, 30, 65536, 5000)
    assert C.window_only_hint(800, 300, 0, 65536, 5000) and C.window_only_hint(30, 30, 200, 65536, 5000)   # fixed shares above 1023
    reads = []
    for chrom, n in (("sparse", 30), ("deep", 170), ("deeper", 240), ("shortruns", 30)):    # 170 / 240 reads per strand / 6000 bp: 3700 / 5200 per tile
        pos = np.sort(rng.choice(np.arange(1, 5900), size=n, replace=False))
        reads += [(False, chrom, int(p), 36) for p in pos]
        reads += [(True, chrom, int(p), 36) for p in pos]
    reads.sort(key=lambda r: (names.index(r[1]), r[2]))
    ctx = Recording()
    calc = CCHipCalculator(100, 36, names, lens, bwfeeder=DictFeeder(tracks), context=ctx)
    ocalc = oracle.OracleCalculator(100, 36, names, lens, mappability=tracks)
    feed_all(calc, reads)
    feed_all(ocalc, reads)
    calc.finishup_calculation()
    ocalc.finishup_calculation()
    assert_matches_oracle(calc, ocalc, names)
    # ONE batched pass for the four chromosomes (their hints do not split it: round-3 advisor finding), with the hint most of
    # its positions ask for: sparse / deep / window / window at equal lengths -> half window, three quarters deep or beyond
    hints = ffi.PMX_FLAG_WINDOW_ONLY | ffi.PMX_FLAG_DEEP_LISTS | ffi.PMX_FLAG_EVENTS_HINT
    assert [f & hints for f in ctx.flags] == [ffi.PMX_FLAG_DEEP_LISTS] * 4
    # a long sparse chromosome beside a short dense one: the pass is the sparse one's (and a hint is ALWAYS given: without
    # one the library samples the vectors and synchronises the stream)
    ctx2 = Recording()
    calc2 = CCHipCalculator(100, 36, ["long", "mito"], [60000, 3000], context=ctx2)
    p1 = np.sort(rng.choice(np.arange(1, 59000), size=200, replace=False))
    p2 = np.sort(rng.choice(np.arange(1, 2900), size=900, replace=False))
    feed_all(calc2, [(False, "long", int(p), 36) for p in p1] + [(True, "mito", int(p), 36) for p in p2])
    calc2.finishup_calculation()
    assert [f & hints for f in ctx2.flags] == [ffi.PMX_FLAG_EVENTS_HINT] * 2


def test_fetching_results_between_two_bulk_chunks_keeps_the_feed_state_of_the_chromosome_being_fed():
    """A chromosome in the middle of its bulk feed may hold a RECYCLED arena slot below slots that are being fetched: the
    fetch must clear only the slots it returns (round-3 advisor finding: it cleared the whole arena prefix, and the read-
    length sums / duplicate rule / look-back bound of the chromosome being fed were lost without an error)."""
    rng = np.random.default_rng(11)
    names, lens = ["a", "b", "c"], [20000, 15000, 12000]
    chunks = {}
    for chrom, glen in zip(names, lens):
        pos = np.sort(rng.integers(1, glen, size=600)).astype(np.int64)
        pos[300] = pos[299]                       # a duplicate position across the chunk boundary
        rev = rng.random(600) < 0.5
        rev[299] = rev[300] = False               # ... both forward: the second is a duplicate (mscc.pyx:388-392)
        rl = rng.integers(30, 60, size=600)       # mixed read lengths: the reverse look-back bound matters
        chunks[chrom] = (pos, rl, rev)
    one = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    for chrom in names:
        pos, rl, rev = chunks[chrom]
        for p, l, r in zip(pos.tolist(), rl.tolist(), rev.tolist()):
            (one.feed_reverse_read if r else one.feed_forward_read)(chrom, p, l)
    one.finishup_calculation()

    two = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    prev = None
    for chrom in names:
        pos, rl, rev = chunks[chrom]
        two.feed_reads(chrom, pos[:300], rl[:300], rev[:300])
        if prev is not None:
            two.get_result(prev)                  # fetch the finished chromosome: its slot is recycled by the next one
        two.feed_reads(chrom, pos[300:], rl[300:], rev[300:])
        prev = chrom
    two.finishup_calculation()
    for c in names:
        a, b = one.get_result(c).chrom, two.get_result(c).chrom
        assert list(a.ccbins) == list(b.ccbins) and (a.forward_sum, a.reverse_sum) == (b.forward_sum, b.reverse_sum)
        assert (a.forward_read_len_sum, a.reverse_read_len_sum) == (b.forward_read_len_sum, b.reverse_read_len_sum), c
    assert (one.forward_read_len_sum, one.reverse_read_len_sum) == (two.forward_read_len_sum, two.reverse_read_len_sum)


def test_packed_strand_keeps_the_width_of_the_position_word():
    """The strand in the top bit: a uint32 packing is the same bits as the int32 one (not converted to int64, which would make
    bit 31 part of the position), narrower words are refused (round-3 advisor finding)."""
    from pymasc_amd import ffi
    rng = np.random.default_rng(8)
    pos = np.sort(rng.integers(1, 20000, size=500)).astype(np.int32)
    rev = rng.random(500) < 0.5
    packed = ffi.pack_strand(pos, rev)
    res = []
    for arr in (packed, packed.view(np.uint32)):
        calc = CCHipCalculator(60, 36, ["a"], [21000], context=FakeContext())
        calc.feed_reads("a", arr, 36, None)
        calc.finishup_calculation()
        r = calc.get_result("a").chrom
        res.append((list(r.ccbins), r.forward_sum, r.reverse_sum))
    assert res[0] == res[1] and res[0][2] > 0          # (reverse reads were taken as reverse reads)
    with pytest.raises(TypeError):
        ffi.Context.feed_reads(ffi.Context.__new__(ffi.Context), 0, 0, 100, np.array([1, 2], dtype=np.int16), 36, None, 0, 0)


def test_bulk_feed_in_two_bytes_per_read():
    """feed_reads(chrom, ffi.Delta16Reads, readlen, None): the distance form of a sorted run gives the results of the per-read
    protocol (the encoding round-trips; the calculator looks at the run's first and last position only)."""
    from pymasc_amd import ffi
    rng = np.random.default_rng(44)
    names, lens = ["a", "b"], [3_000_000, 90_000]
    one = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    two = CCHipCalculator(60, 36, names, lens, context=FakeContext())
    for chrom, glen in zip(names, lens):
        pos = np.sort(rng.integers(1, glen // 3, size=9000)).astype(np.int64)
        pos[6000:] += glen // 2                      # a gap wider than the distance field
        rev = rng.random(9000) < 0.5
        for p, r in zip(pos.tolist(), rev.tolist()):
            (one.feed_reverse_read if r else one.feed_forward_read)(chrom, p, 36)
        reads = ffi.pack_delta16(pos[:5000], rev[:5000])
        back, brev = ffi.unpack_delta16(reads)
        assert (back == pos[:5000]).all() and (brev == rev[:5000]).all() and reads.words.dtype == np.uint16
        two.feed_reads(chrom, reads, 36, None)
        two.feed_reads(chrom, ffi.pack_delta16(pos[5000:], rev[5000:]), 36, None)
    with pytest.raises(ReadUnsortedError):            # a run that starts below the reads fed before
        two.feed_reads("b", ffi.pack_delta16(np.array([5, 9]), np.array([False, True])), 36, None)
    one.finishup_calculation()
    two.finishup_calculation()
    for c in names:
        a, b = one.get_result(c).chrom, two.get_result(c).chrom
        assert list(a.ccbins) == list(b.ccbins) and (a.forward_sum, a.reverse_sum) == (b.forward_sum, b.reverse_sum)
        assert (a.forward_read_len_sum, a.reverse_read_len_sum) == (b.forward_read_len_sum, b.reverse_read_len_sum)


def test_track_intervals_out_of_order_or_overlapping_go_the_general_way():
    """The reference ORs intervals in whatever order they come (set(begin + 1, end) per interval, mscc.pyx:343-344): only tracks
    in BigWig order are handed to the builder (PMX_REGIONS_SORTED), the others to the general setter -- same results."""
    names, lens, tracks, reads = _small_mscc_setup(seed=21)
    S, L = 90, 36
    rng = np.random.default_rng(5)
    messy = {}
    for c, iv in tracks.items():
        iv = list(iv)
        iv += [(b + 3, e + 40, v) for b, e, v in iv[::5]]        # overlapping copies
        rng.shuffle(iv)                                          # ... in any order
        messy[c] = iv

    class Spy(FakeContext):
        def __init__(self):
            super().__init__()
            self.sorted_calls = []

        def bits_set_regions_async(self, *a, **kw):
            self.sorted_calls.append(bool(kw.get("sorted_disjoint")))
            return super().bits_set_regions_async(*a, **kw)

    for which, want_sorted in ((tracks, True), (messy, False)):
        ctx = Spy()
        calc = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(which), context=ctx)
        ocalc = oracle.OracleCalculator(S, L, names, lens, mappability={
            c: [x for x in iv if np.float32(x[2]) >= 1] for c, iv in which.items()})
        feed_all(calc, reads)
        feed_all(ocalc, reads)
        calc.finishup_calculation()
        ocalc.finishup_calculation()
        assert_matches_oracle(calc, ocalc, names)
        assert ctx.sorted_calls and all(s == want_sorted for s in ctx.sorted_calls)
