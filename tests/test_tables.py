"""Table writers: replay the reference's golden run (tests/integration/test_golden_outputs.py:
107-226: -d 300 -q 10 -r 36 + 36-mer mappability) through the calculator, write _cc/_mscc/_nreads tables and
compare them with the committed golden files: headers and integers exactly, coefficients to decimal=15 (the
reference's own tolerance, test_golden_outputs.py:159-163), plus the Fisher-z merge against hand-computed values.

The CPU variant drives the host logic with tests/fake_context.py; the GPU variant is the same run on the HIP path."""
import csv
import os

import numpy as np
import pytest

from pymasc_amd import tables as T
from pymasc_amd import result as R
from pymasc_amd.calculator import CCHipCalculator
from . import fixtures as fx
from .helpers import DictFeeder, feed_all


def _golden_result(context=None):
    names, lengths = fx.load_refs()
    kw = {} if context is None else {"context": context}
    calc = CCHipCalculator(300, 36, names, lengths, bwfeeder=DictFeeder(fx.load_bedgraph()), **kw)
    feed_all(calc, fx.load_reads(10))
    calc.finishup_calculation()
    whole = calc.get_whole_result()
    if context is None:
        calc.close()
    return names, whole


def _rows(path):
    with open(path, newline="") as fp:
        return list(csv.reader(fp, dialect="excel-tab"))


def _check_cc(got_path, golden_name):
    got, exp = _rows(got_path), _rows(os.path.join(fx.GOLDEN, golden_name))
    assert got[0] == exp[0]
    assert len(got) == len(exp)
    for g, e in zip(got[1:], exp[1:]):
        assert g[0] == e[0]
        np.testing.assert_almost_equal(np.array(g[1:], dtype=np.float64), np.array(e[1:], dtype=np.float64),
                                       decimal=15)


def _check_golden_tables(tmp_path, context):
    names, whole = _golden_result(context)
    assert isinstance(whole, R.BothGenomeWideResult)
    paths = T.write_tables(tmp_path / "ENCFF000RMB-test.bam", whole, references=names)
    assert [p.name for p in paths] == ["ENCFF000RMB-test_cc.tab", "ENCFF000RMB-test_mscc.tab",
                                       "ENCFF000RMB-test_nreads.tab"]
    _check_cc(paths[0], "ENCFF000RMB-test_cc.tab")
    _check_cc(paths[1], "ENCFF000RMB-test_mscc.tab")
    # integers only: byte for byte
    with open(paths[2], "rb") as a, open(os.path.join(fx.GOLDEN, "ENCFF000RMB-test_nreads.tab"), "rb") as b:
        assert a.read() == b.read()
    # the default column set is the reference's current one: chromosomes that produced statistics
    p = T.write_nreads_table(tmp_path / "cur.bam", T.build_tables(whole))
    assert _rows(p)[0] == ["shift", "whole", "chr1"]
    # and the tables read back
    cc = T.load_cc_table(paths[0])
    assert list(cc) == ["chr1"] and len(cc["chr1"]) == 301
    fw, rv, mfw, mrv = T.load_nreads_table(paths[2])
    assert (fw["chr1"], rv["chr1"]) == (622, 670)
    assert len(mfw["chr1"]) == 301 and len(mrv["chr1"]) == 301


def test_golden_tables_host(tmp_path):
    from .fake_context import FakeContext
    _check_golden_tables(tmp_path, FakeContext())


@pytest.mark.gpu
def test_golden_tables_gpu(tmp_path):
    _check_golden_tables(tmp_path, None)


def test_merge_cc_fisher_z():
    # two chromosomes, one shift with a NaN, one with |r| = 1 (infinite z is dropped)
    n = [1003, 503]
    a = np.array([0.5, np.nan, 1.0, 0.1])
    b = np.array([0.3, 0.2, 0.25, -0.1])
    merged, lo, hi = T.merge_cc(n, [a, b])
    z0 = (1000 * np.arctanh(0.5) + 500 * np.arctanh(0.3)) / 1500
    assert merged[0] == pytest.approx(np.tanh(z0), abs=1e-16)
    assert merged[1] == pytest.approx(0.2, abs=1e-16)
    assert merged[2] == pytest.approx(0.25, abs=1e-16)
    z3 = (1000 * np.arctanh(0.1) + 500 * np.arctanh(-0.1)) / 1500
    assert merged[3] == pytest.approx(np.tanh(z3), abs=1e-16)
    half = 2.5758293035489004 / np.sqrt(1500)
    assert lo[0] == pytest.approx(np.tanh(z0 - half), abs=1e-15)
    assert hi[0] == pytest.approx(np.tanh(z0 + half), abs=1e-15)
    assert (lo <= merged).all() and (merged <= hi).all()
    from scipy.stats import norm
    assert T._Z_99 == norm.ppf(1 - (1 - 0.99) / 2)


def test_multi_chromosome_tables(tmp_path):
    """Columns: sorted, all-NaN chromosomes dropped, empty chromosomes absent; whole = Fisher-z merge with the
    representative lengths (chromosome length for NCC, mappable_len[read_len - 1] for MSCC)."""
    S, L = 8, 4
    def ncc(glen, f, r, bins):
        x = R.NCCResult(S, L, glen, f, r, f * L, r * L, bins)
        x.calc_cc()
        return x
    def mscc(glen, bins, scale):
        x = R.MSCCResult(S, L, glen, np.full(S + 1, 40 * scale), np.full(S + 1, 50 * scale), None, None, bins,
                         tuple(glen - 10 * i for i in range(S + 1)))
        x.calc_cc()
        return x
    chroms = {"chrB": ncc(5000, 100, 120, [9, 8, 7, 6, 5, 4, 3, 2, 1]),
              "chrA": ncc(9000, 300, 280, [30, 28, 26, 24, 22, 20, 18, 16, 14]),
              "chrZ": ncc(1000, 5, 5, [0] * 9),                       # all NaN -> no column
              "chrE": R.EmptyNCCResult.create_empty(700, S, L)}
    mchroms = {"chrB": mscc(4000, [5, 4, 4, 3, 3, 2, 2, 1, 1], 1),
               "chrA": mscc(8000, [12, 11, 10, 9, 8, 7, 6, 5, 4], 2),
               "chrE": R.EmptyMSCCResult.create_empty(700, S, L)}
    whole = R.BothGenomeWideResult(15700, 0, 0, 405, 405, chroms, mchroms)
    tabs = T.build_tables(whole)
    m, _, _ = T.merge_cc([5000, 9000], [chroms["chrB"].cc, chroms["chrA"].cc])
    np.testing.assert_array_equal(tabs.ncc_whole, m)      # chrZ is all NaN: takes no part
    m, _, _ = T.merge_cc([4000 - 30, 8000 - 30], [mchroms["chrB"].cc, mchroms["chrA"].cc])
    np.testing.assert_array_equal(tabs.mscc_whole, m)
    paths = T.write_tables(tmp_path / "x.bam", whole)
    assert _rows(paths[0])[0] == ["shift", "whole", "chrA", "chrB"]
    assert _rows(paths[1])[0] == ["shift", "whole", "chrA", "chrB"]
    nr = _rows(paths[2])
    assert nr[0] == ["shift", "whole", "chrA", "chrB", "chrZ"]
    assert nr[1] == ["raw", "405-405", "300-280", "100-120", "5-5"]
    assert nr[2] == ["0", "120-150", "80-100", "40-50", "0-0"]
    assert len(nr) == 2 + S + 1
    # floats are written with their shortest round-trip representation
    assert float(_rows(paths[0])[1][2]) == chroms["chrA"].cc[0]


def test_merge_cc_against_reference_vectors():
    """Vectors produced by the reference's merge_correlations (oracle/make_ref_merge_vectors.py)."""
    import json
    cases = json.load(open(os.path.join(fx.GOLDEN, "ref_merge_correlations.json")))
    unhex = lambda col: np.array([np.nan if x is None else float.fromhex(x) for x in col])
    for case in cases:
        got = T.merge_cc(case["n"], [unhex(c) for c in case["cc"]])
        for g, key in zip(got, ("merged", "lower", "upper")):
            # same numpy here and on the GPU box gives identical doubles; 1e-15 (the reference's table
            # tolerance) allows for a libm that rounds tanh/arctanh differently
            np.testing.assert_allclose(g, unhex(case[key]), rtol=0, atol=1e-15)
