"""Loaders for the data fixtures in tests/golden/ (made by tests/golden/make_fixtures.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_refs():
    names, lengths = [], []
    with open(os.path.join(GOLDEN, "ENCFF000RMB-test.refs.tsv")) as fh:
        for line in fh:
            n, l = line.split()
            names.append(n)
            lengths.append(int(l))
    return names, lengths


def load_reads(mapq=10):
    """Records that survive the reference's read filter (handler/read.py:62-90):
    skip read2 (0x80), mapq < criteria, unmapped (0x4), duplicate (0x400).
    Yields (is_reverse, chrom, pos_1based, readlen) in file order."""
    out = []
    with open(os.path.join(GOLDEN, "ENCFF000RMB-test.reads.tsv")) as fh:
        next(fh)
        for line in fh:
            flag, rname, pos, mq, qlen = line.split("\t")
            flag, pos, mq, qlen = int(flag), int(pos), int(mq), int(qlen)
            if flag & 0x80 or mq < mapq or flag & 0x4 or flag & 0x400:
                continue
            out.append((bool(flag & 0x10), rname, pos, qlen))
    return out


def load_bedgraph():
    """chrom -> list of (begin, end, value): what BigWigReader.fetch sees before the value filter."""
    d = {}
    with open(os.path.join(GOLDEN, "hg19_36mer-test.bedGraph")) as fh:
        for line in fh:
            c, b, e, v = line.split()
            d.setdefault(c, []).append((int(b), int(e), float(v)))
    return d


def load_cc_table(name):
    """_cc.tab / _mscc.tab -> (shifts, {column: float64 array})."""
    with open(os.path.join(GOLDEN, name)) as fh:
        header = fh.readline().rstrip("\n").split("\t")
        cols = {h: [] for h in header}
        for line in fh:
            for h, v in zip(header, line.rstrip("\n").split("\t")):
                cols[h].append(float(v))
    shifts = np.array(cols.pop("shift"), dtype=np.int64)
    return shifts, {k: np.array(v, dtype=np.float64) for k, v in cols.items()}


def load_nreads_table():
    """_nreads.tab -> (raw: {col: (f, r)}, per_shift: {col: (f[], r[])})."""
    with open(os.path.join(GOLDEN, "ENCFF000RMB-test_nreads.tab")) as fh:
        header = fh.readline().rstrip("\n").split("\t")[1:]
        raw = None
        per = {h: ([], []) for h in header}
        for line in fh:
            c = line.rstrip("\n").split("\t")
            vals = [tuple(int(x) for x in v.split("-")) for v in c[1:]]
            if c[0] == "raw":
                raw = dict(zip(header, vals))
            else:
                for h, (f, r) in zip(header, vals):
                    per[h][0].append(f)
                    per[h][1].append(r)
    return raw, {h: (np.array(f), np.array(r)) for h, (f, r) in per.items()}


def load_mappability_json():
    with open(os.path.join(GOLDEN, "hg19_36mer-test_mappability.json")) as fh:
        return json.load(fh)
