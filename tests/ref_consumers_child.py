"""Child process of tests/test_reference_consumers.py (container only: needs /root/reference on PYTHONPATH).

Feeds the reference's golden run (-d 300 -q 10 -r 36 -m <track>) through CCHipCalculator (host logic over the
test-only FakeContext; the kernels are covered by the -m gpu parity tests) and hands what the calculator returns
to the REFERENCE's own consumers, imported from /root/reference:
  * PyMaSC/handler/calc.py:210-218   isinstance(obj, ChromResult) on the pickled-and-unpickled worker payload
  * PyMaSC/result.py:301-356         aggregate_results over per-chromosome payloads
  * PyMaSC/stats.py:600-712          make_genome_wide_stat (dispatches on *GenomeWideResultModel / *ResultModel /
                                     Empty*Result types)
  * PyMaSC/output/stats.py:48-120    output_stats -> _stats.tab
and prints the _stats.tab rows as JSON on the last line of stdout."""
import json
import os
import pickle
import sys
import tempfile
from dataclasses import dataclass
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import PyMaSC.result as ref_result  # noqa: E402
from PyMaSC.interfaces.result import (BothGenomeWideResultModel, ChromResult, MSCCResultModel,  # noqa: E402
                                      NCCGenomeWideResultModel, NCCResultModel)
from PyMaSC.output.stats import output_stats  # noqa: E402
from PyMaSC.stats import make_genome_wide_stat  # noqa: E402

import fixtures  # noqa: E402
from fake_context import FakeContext  # noqa: E402
from helpers import DictFeeder, feed_all  # noqa: E402
from pymasc_amd import result as R  # noqa: E402
from pymasc_amd.calculator import CCHipCalculator  # noqa: E402


@dataclass
class StatConfig:                    # the fields of PyMaSC/interfaces/config.py:70-77, CLI defaults parsearg.py:241-258
    read_length: int = 36
    chi2_pval: float = 0.05
    mv_avr_filter_len: int = 15
    filter_mask_len: int = 5
    min_calc_width: int = 50
    expected_library_length: Optional[int] = None


def golden_calc(skip_ncc=False, with_track=True):
    names, lengths = fixtures.load_refs()
    feeder = DictFeeder(fixtures.load_bedgraph()) if with_track else None
    calc = CCHipCalculator(300, 36, names, lengths, feeder, skip_ncc, context=FakeContext())
    feed_all(calc, fixtures.load_reads(mapq=10))
    calc.finishup_calculation()
    return calc, names


def stats_rows(whole):
    stats = make_genome_wide_stat(whole, StatConfig(), output_warnings=False)
    with tempfile.TemporaryDirectory() as td:
        base = os.path.join(td, "ENCFF000RMB-test")
        output_stats(base, stats)
        with open(base + "_stats.tab") as fh:
            return dict(line.rstrip("\n").split("\t", 1) for line in fh if "\t" in line)


def main():
    assert R.REFERENCE_TYPES, "PyMaSC importable but pymasc_amd.result did not bind the reference's classes"
    out = {}
    calc, names = golden_calc()

    # --- per-chromosome payloads: what CalcWorker._report_result puts on the queue (worker.py:226-234) ---
    per_chrom = {}
    for c in names:
        obj = pickle.loads(pickle.dumps(calc.get_result(c)))          # the multiprocessing.Queue hop
        assert isinstance(obj, ChromResult), c                        # handler/calc.py:218
        assert type(obj) is ref_result.BothChromResult
        assert isinstance(obj.chrom, NCCResultModel) and isinstance(obj.mappable_chrom, MSCCResultModel)
        per_chrom[c] = obj
    empty = [c for c in names if isinstance(per_chrom[c].chrom, ref_result.EmptyNCCResult)]
    assert len(empty) == len(names) - 1 and "chr1" not in empty
    assert all(isinstance(per_chrom[c].mappable_chrom, ref_result.EmptyMSCCResult) for c in empty)

    # --- genome-wide payload of the single-process path (handler/calc.py:159-161) ---
    whole = calc.get_whole_result()
    assert type(whole) is ref_result.BothGenomeWideResult and isinstance(whole, BothGenomeWideResultModel)
    out["single"] = stats_rows(whole)

    # --- the -p path: aggregate_results over the reported payloads (handler/calc.py:235) ---
    agg = ref_result.aggregate_results(per_chrom)
    assert type(agg) is ref_result.BothGenomeWideResult
    for k in ("genomelen", "forward_sum", "reverse_sum", "forward_read_len_sum", "reverse_read_len_sum"):
        assert getattr(agg, k) == getattr(whole, k), k
    out["aggregated"] = stats_rows(agg)

    # --- NCC only (no track) and MSCC only (--skip-ncc) ---
    calc_n, _ = golden_calc(with_track=False)
    wn = calc_n.get_whole_result()
    assert type(wn) is ref_result.NCCGenomeWideResult and isinstance(wn, NCCGenomeWideResultModel)
    out["ncc_only"] = stats_rows(wn)
    calc_m, _ = golden_calc(skip_ncc=True)
    per_m = {c: pickle.loads(pickle.dumps(calc_m.get_result(c))) for c in names}
    assert all(isinstance(o, ChromResult) for o in per_m.values())
    # with --skip-ncc the reference still creates EmptyNCCResult placeholders for every reference in
    # _fill_result (mscc.pyx:186-190): same here, so the genome-wide payload is the Both type
    wm = calc_m.get_whole_result()
    assert type(wm) is ref_result.BothGenomeWideResult
    out["skip_ncc"] = stats_rows(wm)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
