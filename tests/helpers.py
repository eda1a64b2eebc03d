"""Shared assertions: product results (pymasc_amd.result dataclasses) vs oracle dicts."""
import numpy as np


class DictFeeder:
    """bwfeeder stand-in over {chrom: [(begin, end, value)]}: the contract of BigWigReader.fetch
    (reader/bigwig.pyx:147-177): KeyError for unknown chromosomes, intervals filtered by value >= threshold."""

    def __init__(self, tracks):
        self.tracks = tracks

    def fetch(self, valfilter, chrom):
        if chrom not in self.tracks:
            raise KeyError(chrom)
        iv = self.tracks[chrom]
        if iv and valfilter > 0:
            iv = [x for x in iv if np.float32(x[2]) >= np.float32(valfilter)]
        return iter(iv)


def feed_all(calc, reads):
    for rev, chrom, pos, rl in reads:
        (calc.feed_reverse_read if rev else calc.feed_forward_read)(chrom, pos, rl)


def assert_cc_equal(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    # same integers + same float64 operation order => identical doubles
    assert np.array_equal(a[m], b[m]), np.abs(a[m] - b[m]).max()


def assert_matches_oracle(calc, ocalc, chroms):
    for c in chroms:
        got = calc.get_result(c)
        ncc, mscc = ocalc.get_result(c)
        if ncc is None:
            assert got.chrom is None
        else:
            r = got.chrom
            for k in ("max_shift", "read_len", "genomelen", "forward_sum", "reverse_sum",
                      "forward_read_len_sum", "reverse_read_len_sum"):
                assert getattr(r, k) == ncc[k], (c, k, getattr(r, k), ncc[k])
            assert [int(x) for x in r.ccbins] == [int(x) for x in ncc["ccbins"]], c
            assert_cc_equal(r.cc, ncc["cc"])
        if mscc is None:
            assert got.mappable_chrom is None
        else:
            r = got.mappable_chrom
            for k in ("max_shift", "read_len", "genomelen", "forward_read_len_sum", "reverse_read_len_sum"):
                assert getattr(r, k) == mscc[k], (c, k)
            assert [int(x) for x in r.forward_sum] == [int(x) for x in mscc["forward_sum"]], c
            assert [int(x) for x in r.reverse_sum] == [int(x) for x in mscc["reverse_sum"]], c
            assert [int(x) for x in r.ccbins] == [int(x) for x in mscc["ccbins"]], c
            assert list(r.mappable_len) == list(mscc["mappable_len"]), c
            assert_cc_equal(r.cc, mscc["cc"])
    assert calc.forward_sum == ocalc.forward_sum and calc.reverse_sum == ocalc.reverse_sum
    assert calc.forward_read_len_sum == ocalc.forward_read_len_sum
    assert calc.reverse_read_len_sum == ocalc.reverse_read_len_sum
