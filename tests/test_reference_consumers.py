"""Drop-in proof at the Python boundary: the reference's OWN consumers accept what CCHipCalculator returns.

Container only (skipped where /root/reference does not exist, e.g. on the GPU box).  Runs in a child process
with /root/reference on PYTHONPATH so that ``pymasc_amd.result`` binds the reference's classes there, while this
pytest process keeps the stand-alone dataclasses (what the GPU box runs).  The child is
tests/ref_consumers_child.py; the expected numbers are the reference's committed golden
``ENCFF000RMB-test_stats.tab`` (tests/integration/test_golden_outputs.py:44-105, decimal=10)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PyMaSC")),
                                reason="reference checkout not present (GPU box)")


def _golden_stats():
    with open(os.path.join(HERE, "golden", "ENCFF000RMB-test_stats.tab")) as fh:
        return dict(line.rstrip("\n").split("\t", 1) for line in fh if "\t" in line)


def _assert_rows(got, want, keys=None):
    assert set(got) == set(want)
    for k in (keys or want):
        try:
            g, w = float(got[k]), float(want[k])
        except ValueError:
            assert got[k] == want[k], k
            continue
        if np.isnan(w):
            assert np.isnan(g), (k, got[k])
        else:
            np.testing.assert_almost_equal(g, w, decimal=10, err_msg=k)


@pytest.fixture(scope="module")
def child_rows():
    env = dict(os.environ, PYTHONPATH=REF + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(HERE, "ref_consumers_child.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_single_process_payload_through_reference_stats(child_rows):
    """get_whole_result() -> make_genome_wide_stat -> output_stats == the reference's golden _stats.tab."""
    _assert_rows(child_rows["single"], _golden_stats())


def test_worker_payloads_through_reference_aggregation(child_rows):
    """pickled get_result() per chromosome -> isinstance(ChromResult) -> aggregate_results -> same _stats.tab."""
    _assert_rows(child_rows["aggregated"], _golden_stats())


def test_ncc_only_and_skip_ncc_payloads(child_rows):
    """NCCGenomeWideResult (no track) and the --skip-ncc payload go through the same consumers.  Without MSCC the
    reference estimates the library length from the NCC curve itself, so only the rows that do not depend on that
    estimate are comparable with the golden (NCC+MSCC) run."""
    want = _golden_stats()
    got = child_rows["ncc_only"]
    _assert_rows(got, dict(got, **{k: want[k] for k in want}),
                 ["Name", "Read length", "Genome length", "Forward reads", "Reverse reads", "Minimum NCC",
                  "NCC at read length"])
    assert got["DMP length"] == "nan" and got["Estimated library length"] != "nan"
    got = child_rows["skip_ncc"]
    _assert_rows(got, dict(got, **{k: want[k] for k in want}),
                 [k for k in want if "MSCC" in k or "DMP" in k] + ["Estimated library length"])


def test_standalone_types_here():
    """This process has no PyMaSC on the path: the stand-alone dataclasses are bound (what the GPU box runs)."""
    from pymasc_amd import result as R
    assert not R.REFERENCE_TYPES and R.NCCResult.__module__ == "pymasc_amd.result"
