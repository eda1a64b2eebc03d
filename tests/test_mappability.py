"""Mappable-length pre-calculation and its JSON cache.

Known answer: the reference ships tests/data/hg19_36mer-test_mappability.json next to the 36-mer track whose
bedGraph dump is tests/golden/hg19_36mer-test.bedGraph (both committed under tests/golden/): recomputing the
table from the intervals must reproduce the file byte for byte (same keys, indent, integer lists).  The CPU
variants run the host logic over tests/fake_context.py; the GPU variants use the HIP autocorrelation kernel."""
import json
import os
import shutil

import numpy as np
import pytest

from pymasc_amd import mappability as MP
from pymasc_amd.calculator import CCHipCalculator
from . import fixtures as fx
from .helpers import DictFeeder, assert_cc_equal, feed_all

GOLDEN_JSON = os.path.join(fx.GOLDEN, "hg19_36mer-test_mappability.json")


class TrackFeeder(DictFeeder):
    """DictFeeder + the ``chromsizes`` attribute of the reference's BigWigReader (reader/bigwig.pyx:60-75)."""

    def __init__(self, tracks, chromsizes):
        super().__init__(tracks)
        self.chromsizes = chromsizes


def _golden_feeder():
    names, lengths = fx.load_refs()
    sizes = dict(zip(names, lengths))
    return TrackFeeder(fx.load_bedgraph(), {"chr1": sizes["chr1"]})   # the test BigWig holds chr1 only


def _ctx(kind):
    if kind == "fake":
        from .fake_context import FakeContext
        return FakeContext()
    from pymasc_amd import ffi
    return ffi.Context(0)


def test_required_shift_size():
    # handler/mappability.py:122-136
    assert MP.required_shift_size(36, 300) == 265
    assert MP.required_shift_size(36, 71) == 36
    assert MP.required_shift_size(36, 72) == 37
    assert MP.required_shift_size(50, 10) == 50
    for L in (1, 5, 36):
        for S in range(0, 200, 7):
            need = max(abs(d - (L - 1)) for d in range(S + 1))
            assert MP.required_shift_size(L, S) >= need


def test_default_stats_path():
    assert str(MP.default_stats_path("/a/b/hg19_36mer.bigwig")) == "/a/b/hg19_36mer_mappability.json"
    assert str(MP.default_stats_path("t.x.bw")) == "t.x_mappability.json"


def _check_precalc_reproduces_golden(tmp_path, kind):
    ctx = _ctx(kind)
    try:
        st = MP.MappabilityStats(_golden_feeder(), max_shift=300, readlen=36,
                                 track_path=tmp_path / "hg19_36mer-test.bigwig", context=ctx)
        assert st.max_shift == 265 and st.need_save_stats and not st.is_called
        assert st.map_path == tmp_path / "hg19_36mer-test_mappability.json"
        with pytest.raises(KeyError):
            st.get_mappable_len("chr1")
        assert st.get_mappable_len("chrNope") is None
        st.calc_mappability()
        assert st.is_called and st.chrom2is_called == {"chr1": True}
        st.save_mappability_stats()
        assert not st.need_save_stats
        with open(st.map_path, "rb") as a, open(GOLDEN_JSON, "rb") as b:
            assert a.read() == b.read()
        assert st.get_mappable_len(shift_from=0, shift_to=3) == [34858, 33873, 32997]
        assert st.get_mappable_len("chr1", 1, 3) == (33873, 32997)
    finally:
        ctx.close()


def test_precalc_reproduces_golden_json_host(tmp_path):
    _check_precalc_reproduces_golden(tmp_path, "fake")


@pytest.mark.gpu
def test_precalc_reproduces_golden_json_gpu(tmp_path):
    _check_precalc_reproduces_golden(tmp_path, "gpu")


def test_cache_rules(tmp_path):
    """handler/mappability.py:201-262: load when valid; recompute when short, malformed or incomplete."""
    from .fake_context import FakeContext
    path = tmp_path / "m.json"
    shutil.copy(GOLDEN_JSON, path)
    feeder = _golden_feeder()
    # valid and long enough: loaded, truncated to the lags this run needs, nothing to save
    st = MP.MappabilityStats(feeder, max_shift=100, readlen=36, map_path=path, context=FakeContext())
    assert st.is_called and not st.need_save_stats and st.max_shift == 65
    gold = json.load(open(GOLDEN_JSON))
    assert st.mappable_len == gold["__whole__"][:66]
    assert st.chrom2mappable_len["chr1"] == tuple(gold["references"]["chr1"][:66])
    st.save_mappability_stats()                       # no-op
    assert json.load(open(path)) == gold
    # a longer run than the cache covers: recompute and overwrite
    st = MP.MappabilityStats(feeder, max_shift=400, readlen=36, map_path=path, context=FakeContext())
    assert st.need_save_stats and not st.is_called and st.max_shift == 365
    st.save_mappability_stats()                       # computes what is missing first
    new = json.load(open(path))
    assert new["max_shift"] == 365 and len(new["__whole__"]) == 366
    assert new["references"]["chr1"][:266] == gold["references"]["chr1"]
    assert new["__whole__"] == new["references"]["chr1"]
    # broken files are recomputed, never trusted
    for breaker in (lambda d: d.pop("__whole__"), lambda d: d["__whole__"].pop(),
                    lambda d: d["references"].pop("chr1"), lambda d: d["references"]["chr1"].append(1)):
        bad = json.load(open(GOLDEN_JSON))
        breaker(bad)
        json.dump(bad, open(path, "w"))
        st = MP.MappabilityStats(feeder, max_shift=300, readlen=36, map_path=path, context=FakeContext())
        assert st.need_save_stats and not st.is_called
    path.write_text("{ not json")
    st = MP.MappabilityStats(feeder, max_shift=300, readlen=36, map_path=path, context=FakeContext())
    assert st.need_save_stats
    # a directory in place of the file / an unwritable location
    with pytest.raises(MP.JSONIOError):
        MP.MappabilityStats(feeder, 300, 36, map_path=tmp_path, context=FakeContext())
    with pytest.raises(MP.JSONIOError):
        MP.MappabilityStats(feeder, 300, 36, map_path=tmp_path / "no" / "such" / "dir.json", context=FakeContext())
    # read_stats raises what the reference raises
    json.dump(gold, open(path, "w"))
    with pytest.raises(MP.NeedUpdate):
        MP.read_stats(path, 266, ["chr1"])
    with pytest.raises(KeyError):
        MP.read_stats(path, 100, ["chr1", "chr2"])


def _check_calculator_with_cache(kind):
    """A calculator given the cached lag tables skips the autocorrelation and still returns identical results."""
    names, lengths = fx.load_refs()
    gold = json.load(open(GOLDEN_JSON))
    results = []
    for cache in (None, gold["references"]):
        ctx = _ctx(kind)
        calls = []
        if kind == "fake":
            orig = ctx.cc_dev
            ctx.cc_dev = lambda *a, _o=orig: (calls.append(a[6]), _o(*a))[1]
        calc = CCHipCalculator(300, 36, names, lengths, bwfeeder=DictFeeder(fx.load_bedgraph()), context=ctx,
                               chrom2mappable_len=cache)
        feed_all(calc, fx.load_reads(10))
        calc.finishup_calculation()
        results.append(calc.get_result("chr1").mappable_chrom)
        if kind == "fake":
            from pymasc_amd import ffi
            assert [bool(f & ffi.PMX_FLAG_SKIP_MLEN) for f in calls] == [cache is not None]
        ctx.close()
    a, b = results
    assert list(a.mappable_len) == list(b.mappable_len) == gold["references"]["chr1"]
    assert list(a.forward_sum) == list(b.forward_sum) and list(a.ccbins) == list(b.ccbins)
    assert_cc_equal(a.cc, b.cc)


def test_calculator_with_cache_host():
    _check_calculator_with_cache("fake")


@pytest.mark.gpu
def test_calculator_with_cache_gpu():
    _check_calculator_with_cache("gpu")


@pytest.mark.gpu
def test_skip_mlen_flag_leaves_row_zero():
    from pymasc_amd import ffi
    from oracle import model as oracle
    from . import synth
    S, L = 500, 50
    nbits, F, R, M = synth.make_case(5, 300_000, S, L)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    with ffi.Context(0) as ctx:
        for flags in (ffi.PMX_FLAG_SKIP_MLEN, ffi.PMX_FLAG_SKIP_MLEN | ffi.PMX_FLAG_FORCE_DENSE):
            out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
            assert np.array_equal(out[ffi.PMX_ROW_MSCC_CCBINS], ref["mscc_ccbins"])
            assert np.array_equal(out[ffi.PMX_ROW_MSCC_FSUM], ref["mscc_forward_sum"])
            assert np.array_equal(out[ffi.PMX_ROW_MSCC_RSUM], ref["mscc_reverse_sum"])
            assert not out[ffi.PMX_ROW_MLEN].any() and out[ffi.PMX_ROW_SCALARS, 2] == 0
