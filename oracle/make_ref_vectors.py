#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/ref_successive_ncc.json by running the
REFERENCE's own compiled `successive` NaiveCCCalculator (built by oracle/build_ref.sh into
oracle/_ref/) on seeded synthetic reads.  Runs in the build container only; the committed JSON
holds inputs (reads) and the reference's integer outputs -- data, no reference code.

The reference asserts bitarray == successive on its own data
(/root/reference/tests/integration/test_golden_outputs.py:672-718), so these vectors pin the
bitarray semantics that oracle/cc_oracle.c restates (the bitarray extension itself cannot be
built: its C library is an empty submodule).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, "/root/reference")

import PyMaSC.core.successive as _succ  # noqa: E402
_succ.__path__.append(os.path.join(HERE, "_ref"))
from PyMaSC.core.successive.ncc import NaiveCCCalculator  # noqa: E402


def gen_case(seed, chroms, max_shift, n_reads, readlen_choices, cluster=False, dup_frac=0.1):
    rng = np.random.default_rng(seed)
    reads = []
    for name, length in chroms:
        n = int(n_reads)
        hi = length - max(readlen_choices)
        if cluster:
            centers = rng.integers(1, hi, size=max(1, n // 40))
            pos = np.clip(rng.choice(centers, size=n) + rng.integers(-150, 150, size=n), 1, hi)
        else:
            pos = rng.integers(1, hi, size=n)
        ndup = int(n * dup_frac)
        if ndup:
            pos[rng.integers(0, n, size=ndup)] = pos[rng.integers(0, n, size=ndup)]
        pos.sort()
        rev = rng.random(n) < 0.5
        rl = rng.choice(readlen_choices, size=n)
        for p, r, l in zip(pos.tolist(), rev.tolist(), rl.tolist()):
            reads.append([name, int(p), bool(r), int(l)])
    return reads


def run_reference(chroms, max_shift, reads):
    calc = NaiveCCCalculator(max_shift, [c for c, _ in chroms], [l for _, l in chroms])
    for name, pos, rev, rl in reads:
        (calc.feed_reverse_read if rev else calc.feed_forward_read)(name, pos, rl)
    calc.finishup_calculation()
    whole = calc.get_whole_result()
    out = {"genomelen": int(whole.genomelen), "forward_sum": int(whole.forward_sum),
           "reverse_sum": int(whole.reverse_sum), "chroms": {}}
    for name, res in whole.chroms.items():
        out["chroms"][name] = {
            "forward_sum": int(res.forward_sum), "reverse_sum": int(res.reverse_sum),
            "forward_read_len_sum": int(res.forward_read_len_sum),
            "reverse_read_len_sum": int(res.reverse_read_len_sum),
            "ccbins": [int(x) for x in np.asarray(res.ccbins)],
            "cc": [None if np.isnan(x) else float(x) for x in np.asarray(res.cc, dtype=np.float64)],
        }
    return out


CASES = [
    dict(name="uniform_one_chrom", seed=1, chroms=[("chrA", 50000)], max_shift=100, n_reads=3000,
         readlen_choices=[36]),
    dict(name="clustered_two_chroms", seed=2, chroms=[("chrA", 80000), ("chrB", 30000)], max_shift=300,
         n_reads=2500, readlen_choices=[36], cluster=True),
    dict(name="variable_readlen", seed=3, chroms=[("c1", 20000), ("c2", 20000), ("c3", 5000)], max_shift=150,
         n_reads=1500, readlen_choices=[25, 36, 50, 76], dup_frac=0.3),
    dict(name="dense_small", seed=4, chroms=[("d", 3000)], max_shift=64, n_reads=4000, readlen_choices=[20],
         dup_frac=0.0),
    dict(name="shift_longer_than_chrom_fraction", seed=5, chroms=[("e", 1500)], max_shift=700, n_reads=400,
         readlen_choices=[36]),
    dict(name="empty_chrom_between", seed=6, chroms=[("x", 10000), ("y", 7000), ("z", 9000)], max_shift=63,
         n_reads=600, readlen_choices=[36], skip_chrom="y"),
]


def main():
    out = []
    for case in CASES:
        kw = {k: v for k, v in case.items() if k not in ("name", "skip_chrom")}
        reads = gen_case(**kw)
        if case.get("skip_chrom"):
            reads = [r for r in reads if r[0] != case["skip_chrom"]]
        res = run_reference(case["chroms"], case["max_shift"], reads)
        out.append({"name": case["name"], "chroms": case["chroms"], "max_shift": case["max_shift"],
                    "reads": reads, "expected": res})
        print(case["name"], len(reads), "reads; fsum/rsum", res["forward_sum"], res["reverse_sum"])
    path = os.path.join(ROOT, "tests", "golden", "ref_successive_ncc.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
