"""Pins the CPU oracle (oracle/) to everything the reference's own tests hold for this path:
the CLI goldens of tests/integration/test_golden_outputs.py:44-226 (-d 300 -q 10 -r 36 with the
36-mer mappability track), the lag table JSON, and vectors produced by the reference's compiled
`successive` NCC calculator (oracle/make_ref_vectors.py)."""
import json
import os

import numpy as np
import pytest

from oracle import model as oracle
from . import fixtures as fx
from .helpers import feed_all


@pytest.fixture(scope="module")
def golden_run():
    names, lengths = fx.load_refs()
    calc = oracle.OracleCalculator(300, 36, names, lengths, mappability=fx.load_bedgraph())
    feed_all(calc, fx.load_reads(10))
    calc.flush("chr1")
    return calc


def test_ncc_integers_and_cc_table(golden_run):
    ncc, _ = golden_run.get_result("chr1")
    assert (ncc["forward_sum"], ncc["reverse_sum"]) == (622, 670)
    assert ncc["ccbins"][:5] == [28, 26, 19, 22, 26]
    raw, _ = fx.load_nreads_table()
    assert raw["chr1"] == (622, 670) and raw["whole"] == (622, 670)
    _, cc = fx.load_cc_table("ENCFF000RMB-test_cc.tab")
    np.testing.assert_allclose(ncc["cc"], cc["chr1"], rtol=0, atol=1e-15)


def test_mscc_integers_and_cc_table(golden_run):
    _, mscc = golden_run.get_result("chr1")
    _, per = fx.load_nreads_table()
    np.testing.assert_array_equal(np.array(mscc["forward_sum"]), per["chr1"][0])
    np.testing.assert_array_equal(np.array(mscc["reverse_sum"]), per["chr1"][1])
    assert mscc["ccbins"][:4] == [16, 20, 14, 15]
    _, cc = fx.load_cc_table("ENCFF000RMB-test_mscc.tab")
    np.testing.assert_allclose(mscc["cc"], cc["chr1"], rtol=0, atol=1e-15)


def test_mappable_len_lag_table(golden_run):
    _, mscc = golden_run.get_result("chr1")
    table = fx.load_mappability_json()
    assert table["max_shift"] == 265
    assert list(mscc["mappable_len"]) == table["references"]["chr1"]
    assert len(mscc["mappable_len"]) == max(36, 300 - 36 + 2)


def test_readless_chromosome_lag_table():
    # the same track loaded as a chromosome WITHOUT reads goes through _fill_result's loop
    # (mscc.pyx:207-215) and must give the same lags (symmetry of the autocorrelation)
    names, lengths = ["chr1"], [249250621]
    bg = fx.load_bedgraph()
    # restrict to a window so the (S+1) x 4 full passes stay fast: shift coordinates to a small chromosome
    lo = min(b for b, e, v in bg["chr1"])
    hi = max(e for b, e, v in bg["chr1"])
    small = {"chr1": [(b - lo, e - lo, v) for b, e, v in bg["chr1"]]}
    calc = oracle.OracleCalculator(265, 36, names, [hi - lo], mappability=small)
    calc._fill_result("chr1")
    table = fx.load_mappability_json()
    assert list(calc.ref2mscc["chr1"]["mappable_len"]) == table["references"]["chr1"]


CASES = json.load(open(os.path.join(fx.GOLDEN, "ref_successive_ncc.json")))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_against_reference_successive_ncc(case):
    names = [n for n, _ in case["chroms"]]
    lens = [l for _, l in case["chroms"]]
    calc = oracle.OracleCalculator(case["max_shift"], 36, names, lens)
    for name, pos, rev, rl in case["reads"]:
        (calc.feed_reverse_read if rev else calc.feed_forward_read)(name, pos, rl)
    calc.finishup_calculation()
    exp = case["expected"]
    assert calc.forward_sum == exp["forward_sum"] and calc.reverse_sum == exp["reverse_sum"]
    for ch, e in exp["chroms"].items():
        ncc, _ = calc.get_result(ch)
        for k in ("forward_sum", "reverse_sum", "forward_read_len_sum", "reverse_read_len_sum"):
            assert ncc[k] == e[k], (ch, k)
        assert [int(x) for x in ncc["ccbins"]] == e["ccbins"]
        ecc = np.array([np.nan if x is None else x for x in e["cc"]])
        assert np.array_equal(np.isnan(ecc), np.isnan(ncc["cc"]))
        m = ~np.isnan(ecc)
        np.testing.assert_allclose(ncc["cc"][m], ecc[m], rtol=0, atol=1e-15)


def test_bruteforce_definition_small():
    """cc_oracle.c (shift-and-count passes) vs the index-form definitions of SURVEY.md section 0,
    evaluated bit by bit in pure Python on a tiny case."""
    rng = np.random.default_rng(7)
    G, S, L = 300, 90, 12
    nbits = G + L + S + 100
    f = np.zeros(nbits, dtype=np.int64)
    r = np.zeros(nbits, dtype=np.int64)
    m = np.zeros(nbits, dtype=np.int64)
    f[rng.integers(1, G + 1, 60)] = 1
    r[rng.integers(1, G + L, 60)] = 1
    for s in range(1, G, 25):
        m[s:s + int(rng.integers(3, 22))] = 1
    m[G + 1:] = 0
    pack = lambda b: np.packbits(np.concatenate([b, np.zeros((-nbits) % 64, dtype=np.int64)]).astype(np.uint8),
                                 bitorder="little").view(np.uint64).copy()
    out = oracle.calc_correlation(pack(f), pack(r), pack(m), nbits, S, L)
    M = lambda i: int(m[i]) if 0 <= i < nbits else 0
    R = lambda i: int(r[i]) if 0 <= i < nbits else 0
    for d in range(S + 1):
        D = [M(j) & M(j + L - 1 - d) for j in range(nbits)]
        assert out["ncc_ccbins"][d] == sum(int(f[j]) & R(j + d) for j in range(nbits))
        assert out["mappable_len_by_shift"][d] == sum(D)
        assert out["mscc_forward_sum"][d] == sum(int(f[j]) & D[j] for j in range(nbits))
        assert out["mscc_reverse_sum"][d] == sum(R(j + d) & D[j] for j in range(nbits))
        assert out["mscc_ccbins"][d] == sum(int(f[j]) & R(j + d) & D[j] for j in range(nbits))


def test_fixture_read_counts_match_the_reference_summary():
    """The reference's own summary of its test BAM (tests/integration/expected_results/encode_data_summary.json:
    total_reads 2501, all on chr1, chr1_valid_reads 1292) against the fixture the
    parity tests are fed from: 1292 = 622 forward + 670 reverse reads after the -q 10 filter."""
    import csv
    import os
    with open(os.path.join(fx.GOLDEN, "ENCFF000RMB-test.reads.tsv")) as fh:
        rows = list(csv.reader(fh, delimiter="\t"))[1:]
    assert len(rows) == 2501 and {r[1] for r in rows} == {"chr1"}
    reads = fx.load_reads(mapq=10)
    assert len(reads) == 1292
    assert (sum(1 for r in reads if not r[0]), sum(1 for r in reads if r[0])) == (622, 670)
