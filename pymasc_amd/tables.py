"""Tab-separated result tables written from the calculator's results.

What the reference writes after the hot path (PyMaSC/output/table.py):

  <name>_cc.tab    shift | whole | <chrom>...   naive cross-correlation per shift        (table.py:175-209)
  <name>_mscc.tab  same layout for the mappability-sensitive cross-correlation           (table.py:208-209)
  <name>_nreads.tab  "forward-reverse" read-count pairs: a ``raw`` row (NCC totals) and one row per shift
                   (mappable counts)                                                     (table.py:301-333, 369-406)

The ``whole`` column is not a sum: per shift the per-chromosome coefficients are merged in Fisher-z space
weighted by (n - 3) (PyMaSC/utils/calc.py:172-241), with n the chromosome's representative length: the
chromosome length for NCC (stats.py:93-96) and ``mappable_len[read_len - 1]`` for MSCC (stats.py:113-116).
Chromosomes without reads (Empty*Result) take no part in the merge nor in the columns
(stats.py:474-487, 689-696); per-chromosome columns whose coefficients are all NaN are dropped and the rest
sorted by name (table.py:198-201).

Everything here is host-side float64/text over a few thousand numbers; the integers it consumes come from the
HIP path (pymasc_amd/calculator.py).
"""
from __future__ import annotations

import csv
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from .result import (BothGenomeWideResult, EmptyResult, MSCCGenomeWideResult, MSCCResult, NCCGenomeWideResult,
                     NCCResult)

CC_SUFFIX = "_cc.tab"
MSCC_SUFFIX = "_mscc.tab"
NREADS_SUFFIX = "_nreads.tab"

#: two-sided 99 % normal quantile, norm.ppf(1 - (1 - 0.99) / 2)  (utils/calc.py:176,224)
_Z_99 = 2.5758293035489004

GenomeWideResult = Union[NCCGenomeWideResult, MSCCGenomeWideResult, BothGenomeWideResult]


def merge_cc(weights_n: Sequence[int], cc_arrays: Sequence[np.ndarray]
             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Genome-wide coefficient per shift and its 99 % interval: (merged, lower, upper).

    Per shift: drop NaN coefficients, z = arctanh(r), drop infinite z, weighted mean with weights n - 3,
    back through tanh; interval = tanh(mean ± z99 / sqrt(Σ weights))  (utils/calc.py:205-235).  The weighted mean
    is taken with np.average per shift so that the summation order, and with it the last bit, is the reference's.
    """
    n = np.asarray(weights_n, dtype=np.int64)
    if n.ndim != 1 or len(n) != len(cc_arrays):
        raise ValueError("one representative length per correlation array expected")
    table = np.asarray(cc_arrays, dtype=np.float64)          # chromosomes x shifts
    if table.ndim != 2:
        raise ValueError("correlation arrays must share one length")
    nshift = table.shape[1]
    merged = np.empty(nshift, dtype=np.float64)
    lower = np.empty(nshift, dtype=np.float64)
    upper = np.empty(nshift, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        for d in range(nshift):
            col = table[:, d]
            keep = ~np.isnan(col)
            z = np.arctanh(col[keep])
            w = n[keep] - 3
            finite = ~np.isinf(z)
            z, w = z[finite], w[finite]
            if len(z) == 0 or w.sum() == 0:
                merged[d] = lower[d] = upper[d] = np.nan
                continue
            mean = np.average(z, weights=w)
            half = _Z_99 * np.sqrt(1 / np.sum(w))
            merged[d] = np.tanh(mean)
            lower[d] = np.tanh(mean - half)
            upper[d] = np.tanh(mean + half)
    return merged, lower, upper


@dataclass
class TableSet:
    """Everything the three tables hold, already in column order."""
    ncc_whole: Optional[np.ndarray]
    ncc_chroms: Dict[str, np.ndarray]
    mscc_whole: Optional[np.ndarray]
    mscc_chroms: Dict[str, np.ndarray]
    #: column -> (forward, reverse) totals; ``whole`` first
    raw_reads: Optional[Dict[str, Tuple[int, int]]]
    #: column -> (forward[shift], reverse[shift])
    mappable_reads: Optional[Dict[str, Tuple[np.ndarray, np.ndarray]]]
    references: Tuple[str, ...]
    ncc_interval: Optional[Tuple[np.ndarray, np.ndarray]] = None
    mscc_interval: Optional[Tuple[np.ndarray, np.ndarray]] = None


def _regular(chroms: Optional[Mapping[str, Union[NCCResult, MSCCResult]]]):
    if chroms is None:
        return None
    return {c: r for c, r in chroms.items() if r is not None and not isinstance(r, EmptyResult)}


def build_tables(result: GenomeWideResult) -> TableSet:
    """Aggregate a genome-wide result the way stats.py:456-547 does, keeping only what the tables print."""
    if isinstance(result, BothGenomeWideResult):
        ncc, mscc = _regular(result.chroms), _regular(result.mappable_chroms)
    elif isinstance(result, MSCCGenomeWideResult):
        ncc, mscc = None, _regular(result.chroms)
    elif isinstance(result, NCCGenomeWideResult):
        ncc, mscc = _regular(result.chroms), None
    else:
        raise TypeError("unsupported genome-wide result: {!r}".format(type(result)))

    ts = TableSet(None, {}, None, {}, None, None, ())

    if ncc:
        for r in ncc.values():
            if r.cc is None:
                r.calc_cc()
        ts.ncc_whole, lo, hi = merge_cc([r.genomelen for r in ncc.values()], [r.cc for r in ncc.values()])
        ts.ncc_interval = (lo, hi)
        ts.ncc_chroms = {c: np.asarray(r.cc, dtype=np.float64) for c, r in ncc.items()}
        raw = {"whole": (int(sum(int(r.forward_sum) for r in ncc.values())),
                         int(sum(int(r.reverse_sum) for r in ncc.values())))}
        raw.update({c: (int(r.forward_sum), int(r.reverse_sum)) for c, r in ncc.items()})
        ts.raw_reads = raw

    if mscc:
        for r in mscc.values():
            if r.cc is None:
                r.calc_cc()
        # representative length = lag table at read_len - 1 (stats.py:113-116)
        reps = [int(np.asarray(r.mappable_len, dtype=np.int64)[r.read_len - 1]) for r in mscc.values()]
        ts.mscc_whole, lo, hi = merge_cc(reps, [r.cc for r in mscc.values()])
        ts.mscc_interval = (lo, hi)
        ts.mscc_chroms = {c: np.asarray(r.cc, dtype=np.float64) for c, r in mscc.items()}
        fw = np.sum(np.asarray([r.forward_sum for r in mscc.values()], dtype=np.int64), axis=0)
        rv = np.sum(np.asarray([r.reverse_sum for r in mscc.values()], dtype=np.int64), axis=0)
        per = {"whole": (fw, rv)}
        per.update({c: (np.asarray(r.forward_sum, dtype=np.int64), np.asarray(r.reverse_sum, dtype=np.int64))
                    for c, r in mscc.items()})
        ts.mappable_reads = per

    # the reference takes the column set from the NCC statistics when it has them (interfaces/stats.py:186-194)
    ts.references = tuple(ncc.keys()) if ncc is not None else tuple(mscc.keys() if mscc else ())
    return ts


def _table_path(outfile: Union[str, os.PathLike], suffix: str) -> Path:
    """``/dir/name.ext`` -> ``/dir/name<suffix>``  (table.py:185-188)."""
    p = Path(outfile)
    return p.parent / (p.stem + suffix)


def _fmt(v) -> str:
    # csv.writer stringifies with str(); numpy >= 2 prints float64 scalars with the shortest round-trip repr,
    # which is what Python's float repr gives too.
    return repr(float(v))


def _write_cc(path: Path, whole: np.ndarray, chroms: Mapping[str, np.ndarray]) -> Path:
    cols = sorted(c for c, cc in chroms.items() if not np.isnan(cc).all())
    with open(path, "w", newline="") as fp:
        tab = csv.writer(fp, dialect="excel-tab")
        tab.writerow(["shift", "whole"] + cols)
        for d in range(len(whole)):
            tab.writerow([d, _fmt(whole[d])] + [_fmt(chroms[c][d]) for c in cols])
    return path


def write_cc_table(outfile, tables: TableSet) -> Path:
    """``_cc.tab``  (table.py:175-208 with target 'ncc')."""
    if tables.ncc_whole is None:
        raise ValueError("no naive cross-correlation to write")
    return _write_cc(_table_path(outfile, CC_SUFFIX), tables.ncc_whole, tables.ncc_chroms)


def write_mscc_table(outfile, tables: TableSet) -> Path:
    """``_mscc.tab``  (table.py:175-209 with target 'mscc')."""
    if tables.mscc_whole is None:
        raise ValueError("no mappability-sensitive cross-correlation to write")
    return _write_cc(_table_path(outfile, MSCC_SUFFIX), tables.mscc_whole, tables.mscc_chroms)


def write_nreads_table(outfile, tables: TableSet, references: Optional[Iterable[str]] = None) -> Path:
    """``_nreads.tab``  (table.py:301-333, 369-406).

    Columns are ``whole`` + the sorted chromosomes that produced statistics; pass ``references`` (e.g. every BAM
    reference) to list others as ``0-0`` columns, which is how the golden file shipped with the reference's tests
    (tests/golden/ENCFF000RMB-test_nreads.tab) was laid out.
    """
    header = ["whole"] + sorted(tables.references if references is None else references)
    path = _table_path(outfile, NREADS_SUFFIX)
    with open(path, "w", newline="") as fp:
        tab = csv.writer(fp, dialect="excel-tab")
        tab.writerow(["shift"] + header)
        raw = tables.raw_reads
        # the raw row is only written when both whole-genome totals are non-empty dicts (table.py:323-325)
        if raw:
            tab.writerow(["raw"] + ["{}-{}".format(*raw.get(c, (0, 0))) for c in header])
        per = tables.mappable_reads
        if per:
            nshift = len(per["whole"][0])
            zeros = np.zeros(nshift, dtype=np.int64)
            fcols = [per.get(c, (zeros, zeros))[0] for c in header]
            rcols = [per.get(c, (zeros, zeros))[1] for c in header]
            for d in range(nshift):
                tab.writerow([d] + ["{}-{}".format(int(f[d]), int(r[d])) for f, r in zip(fcols, rcols)])
    return path


def write_tables(outfile, result: GenomeWideResult, references: Optional[Iterable[str]] = None) -> List[Path]:
    """Write every table the result supports; returns the paths written."""
    tables = build_tables(result)
    out: List[Path] = []
    if tables.ncc_whole is not None:
        out.append(write_cc_table(outfile, tables))
    if tables.mscc_whole is not None:
        out.append(write_mscc_table(outfile, tables))
    out.append(write_nreads_table(outfile, tables, references))
    return out


# ---------------------------------------------------------------------------------------------------------------
# readers (table.py:112-129, 252-283, 336-366): used by the tests and by callers that replot saved tables
# ---------------------------------------------------------------------------------------------------------------

def load_cc_table(path) -> Dict[str, List[float]]:
    """column -> coefficients, without the ``whole`` column (table.py:143-168)."""
    with open(path, newline="") as fp:
        rows = list(csv.reader(fp, dialect="excel-tab"))
    header = rows[0][1:]
    cols = {h: [float(r[i + 1]) for r in rows[1:]] for i, h in enumerate(header)}
    cols.pop("whole", None)
    return cols


def load_nreads_table(path):
    """(forward, reverse, mappable_forward, mappable_reverse) keyed by chromosome, ``whole`` removed
    (table.py:252-283, 336-366).  Raises KeyError when the file holds no pair at all."""
    with open(path, newline="") as fp:
        rows = list(csv.reader(fp, dialect="excel-tab"))
    header = rows[0][1:]
    fw: Dict[str, int] = {}
    rv: Dict[str, int] = {}
    mfw: Dict[str, List[int]] = {}
    mrv: Dict[str, List[int]] = {}
    for row in rows[1:]:
        for key, pair in zip(header, row[1:]):
            if "-" not in pair:
                continue
            f, r = (int(x) for x in pair.split("-"))
            if row[0] == "raw":
                fw[key], rv[key] = f, r
            else:
                mfw.setdefault(key, []).append(f)
                mrv.setdefault(key, []).append(r)
    for d in (fw, rv, mfw, mrv):
        d.pop("whole", None)
    if not (fw or rv or mfw or mrv):
        raise KeyError("nothing to load from {}".format(path))
    return fw, rv, mfw, mrv
