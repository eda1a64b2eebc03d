"""pymasc_amd.pipeline.run: the reference's golden run (`-d 300 -q 10 -r 36 -m bigwig`) in one call, from the binary
inputs to the three tables, including the mappable-length cache round trip (GPU)."""
import csv
import os
import shutil

import numpy as np
import pytest

from pymasc_amd import pipeline
from . import fixtures as fx

pytestmark = pytest.mark.gpu


def _rows(path):
    with open(path, newline="") as fp:
        return list(csv.reader(fp, dialect="excel-tab"))


def test_golden_run_in_one_call(tmp_path):
    bam = tmp_path / "ENCFF000RMB-test.bam"
    bw = tmp_path / "hg19_36mer-test.bigwig"
    shutil.copy(os.path.join(fx.GOLDEN, "ENCFF000RMB-test.bam"), bam)
    shutil.copy(os.path.join(fx.GOLDEN, "ENCFF000RMB-test.bam.bai"), str(bam) + ".bai")
    shutil.copy(os.path.join(fx.GOLDEN, "hg19_36mer-test.bigwig"), bw)
    for attempt in range(2):                       # 1st: computes and saves the cache; 2nd: loads it (SKIP_MLEN)
        out = tmp_path / ("out%d" % attempt)
        result, written = pipeline.run(bam, out, max_shift=300, read_len=36, mapq_criteria=10, mappability_path=bw)
        assert [p.name for p in written] == ["ENCFF000RMB-test_cc.tab", "ENCFF000RMB-test_mscc.tab",
                                             "ENCFF000RMB-test_nreads.tab"]
        cache = tmp_path / "hg19_36mer-test_mappability.json"
        assert open(cache, "rb").read() == open(os.path.join(fx.GOLDEN, "hg19_36mer-test_mappability.json"), "rb").read()
        for p in written[:2]:
            got, exp = _rows(p), _rows(os.path.join(fx.GOLDEN, p.name))
            assert got[0] == exp[0] and len(got) == len(exp)
            np.testing.assert_almost_equal(np.array([r[1:] for r in got[1:]], dtype=float),
                                           np.array([r[1:] for r in exp[1:]], dtype=float), decimal=15)
        got, exp = _rows(written[2]), _rows(os.path.join(fx.GOLDEN, "ENCFF000RMB-test_nreads.tab"))
        col = exp[0].index("chr1")
        assert got[0] == ["shift", "whole", "chr1"]
        assert [r[:2] + [r[2]] for r in got[1:]] == [[r[0], r[1], r[col]] for r in exp[1:]]
        assert (result.forward_sum, result.reverse_sum) == (622, 670)
