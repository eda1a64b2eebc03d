"""GPU: the drop-in calculator end to end on the reference's own test data -- replays
tests/integration/test_golden_outputs.py:44-226 (-d 300 -q 10 -r 36 + 36-mer mappability) through
CCHipCalculator and compares with the committed golden tables (cc to 1e-15, integers exactly)."""
import numpy as np
import pytest

from oracle import model as oracle
from pymasc_amd import ffi
from pymasc_amd import result as R
from pymasc_amd.calculator import CCHipCalculator
from . import fixtures as fx
from .helpers import DictFeeder, assert_matches_oracle, feed_all
from .test_host_logic import _small_mscc_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_calc():
    names, lengths = fx.load_refs()
    calc = CCHipCalculator(300, 36, names, lengths, bwfeeder=DictFeeder(fx.load_bedgraph()))
    feed_all(calc, fx.load_reads(10))
    calc.flush("chr1")
    yield calc
    calc.close()


def test_golden_ncc(golden_calc):
    r = golden_calc.get_result("chr1").chrom
    assert (r.forward_sum, r.reverse_sum) == (622, 670)
    assert list(r.ccbins[:5]) == [28, 26, 19, 22, 26]
    _, cc = fx.load_cc_table("ENCFF000RMB-test_cc.tab")
    np.testing.assert_allclose(r.cc, cc["chr1"], rtol=0, atol=1e-15)


def test_golden_mscc(golden_calc):
    r = golden_calc.get_result("chr1").mappable_chrom
    _, per = fx.load_nreads_table()
    np.testing.assert_array_equal(np.array(r.forward_sum), per["chr1"][0])
    np.testing.assert_array_equal(np.array(r.reverse_sum), per["chr1"][1])
    assert list(r.ccbins[:4]) == [16, 20, 14, 15]
    assert list(r.mappable_len) == fx.load_mappability_json()["references"]["chr1"]
    _, cc = fx.load_cc_table("ENCFF000RMB-test_mscc.tab")
    np.testing.assert_allclose(r.cc, cc["chr1"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("flags", [0, ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE])
@pytest.mark.parametrize("skip_ncc,early_batch", [(False, 0), (True, 0), (False, 1), (False, 2)])
def test_multi_chromosome_vs_oracle_calculator(flags, skip_ncc, early_batch):
    """early_batch 1 / 2: the kernels of the chromosomes queued so far are launched at every (second) flush, the rest at
    the fetch; 0 (the default): everything at the fetch."""
    names, lens, tracks, reads = _small_mscc_setup(seed=17)
    S, L = 200, 36
    calc = CCHipCalculator(S, L, names, lens, bwfeeder=DictFeeder(tracks), skip_ncc=skip_ncc, kernel_flags=flags)
    calc.early_batch = early_batch
    ocalc = oracle.OracleCalculator(S, L, names, lens, mappability={
        c: [x for x in iv if np.float32(x[2]) >= 1] for c, iv in tracks.items()}, skip_ncc=skip_ncc)
    feed_all(calc, reads)
    feed_all(ocalc, reads)
    calc.finishup_calculation()
    ocalc.finishup_calculation()
    assert_matches_oracle(calc, ocalc, names)
    assert isinstance(calc.get_whole_result(), R.BothGenomeWideResult)
    calc.close()
