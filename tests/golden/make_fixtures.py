#!/usr/bin/env python3
"""Builds the data fixtures under tests/golden/ from the reference's own test DATA files.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

Inputs (data files held by the reference's tests, no source code):
  tests/data/ENCFF000RMB-test.sam            -> ENCFF000RMB-test.reads.tsv (+ .refs.tsv)
        one line per alignment record: flag, rname, pos(1-based), mapq, query length inferred from
        the CIGAR (sum of M/I/S/=/X, what pysam's infer_query_length() returns)
  tests/data/hg19_36mer-test.bedGraph        -> copied (mappability intervals, text twin of the bigwig)
  tests/data/hg19_36mer-test_mappability.json-> copied (lag table golden)
  tests/golden/ENCFF000RMB-test_{cc,mscc,nreads,stats}.tab -> copied (expected outputs)
  tests/data/ENCFF000RMB-test.bam, tests/data/hg19_36mer-test.bigwig -> copied (binary twins of the .sam and the
        .bedGraph above: inputs of the native readers' parity tests, tests/test_io_readers.py)
"""
import os
import re
import shutil

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    refs = []
    rows = []
    with open(os.path.join(REF, "data", "ENCFF000RMB-test.sam")) as fh:
        for line in fh:
            if line.startswith("@"):
                if line.startswith("@SQ"):
                    f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                    refs.append((f["SN"], int(f["LN"])))
                continue
            c = line.rstrip("\n").split("\t")
            flag, rname, pos, mapq, cigar = int(c[1]), c[2], int(c[3]), int(c[4]), c[5]
            qlen = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar) if op in "MIS=X")
            rows.append((flag, rname, pos, mapq, qlen))
    with open(os.path.join(HERE, "ENCFF000RMB-test.refs.tsv"), "w") as out:
        for name, ln in refs:
            out.write(f"{name}\t{ln}\n")
    with open(os.path.join(HERE, "ENCFF000RMB-test.reads.tsv"), "w") as out:
        out.write("flag\trname\tpos\tmapq\tqlen\n")
        for r in rows:
            out.write("\t".join(str(x) for x in r) + "\n")
    for rel in ["data/hg19_36mer-test.bedGraph", "data/hg19_36mer-test_mappability.json",
                "data/ENCFF000RMB-test.bam", "data/ENCFF000RMB-test.bam.bai", "data/hg19_36mer-test.bigwig",
                "golden/ENCFF000RMB-test_cc.tab", "golden/ENCFF000RMB-test_mscc.tab",
                "golden/ENCFF000RMB-test_nreads.tab", "golden/ENCFF000RMB-test_stats.tab"]:
        shutil.copyfile(os.path.join(REF, rel), os.path.join(HERE, os.path.basename(rel)))
        os.chmod(os.path.join(HERE, os.path.basename(rel)), 0o644)
    print(f"{len(refs)} references, {len(rows)} records")


if __name__ == "__main__":
    main()
