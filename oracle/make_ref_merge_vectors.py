#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/ref_merge_correlations.json by calling the
REFERENCE's pure-Python Fisher-z merge (PyMaSC/utils/calc.py:172-241, imported from /root/reference in the
build container) on seeded coefficient tables.  The committed JSON holds inputs (representative lengths,
per-chromosome coefficients with NaN as null) and the reference's float64 outputs as hex strings -- data only.
It pins pymasc_amd/tables.py:merge_cc, i.e. the ``whole`` column of the _cc/_mscc tables.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, "/root/reference")

from PyMaSC.utils.calc import merge_correlations  # noqa: E402


def _hex(a):
    return [None if np.isnan(x) else float(x).hex() for x in a]


def main():
    rng = np.random.default_rng(20250912)
    cases = []
    for trial in range(6):
        k = [1, 2, 3, 7, 24, 40][trial]
        nshift = 32
        n = rng.integers(2000, 250_000_000, size=k)
        ccs = [rng.uniform(-0.15, 0.7, size=nshift) for _ in range(k)]
        if k > 2:
            for c in ccs:
                c[rng.integers(0, nshift, size=4)] = np.nan
            ccs[1][5] = 1.0          # infinite z: dropped by the merge
            ccs[2][9] = -1.0
            ccs[0][:] = np.where(np.isnan(ccs[0]), 0.01, ccs[0])   # keep one finite value per shift
        with np.errstate(divide="ignore"):
            m, lo, hi = merge_correlations(np.array(n, dtype=np.int64), ccs, 36)
        cases.append({"n": [int(x) for x in n], "cc": [_hex(c) for c in ccs],
                      "merged": _hex(m), "lower": _hex(lo), "upper": _hex(hi)})
    out = os.path.join(ROOT, "tests", "golden", "ref_merge_correlations.json")
    with open(out, "w") as fp:
        json.dump(cases, fp)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
