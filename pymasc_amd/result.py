"""Result payloads of the calculator boundary.

Inside a PyMaSC installation (``import PyMaSC.result`` works) the names exported here ARE the reference's own
classes -- ``PyMaSC.result.NCCResult / MSCCResult / BothChromResult / Empty* / *GenomeWideResult`` and its
``aggregate_results`` -- because the reference's consumers dispatch on nominal types:
``PyMaSC/stats.py:384-395,433,610-627`` (``isinstance(..., NCCResultModel / MSCCResultModel / Empty*Result /
*GenomeWideResultModel)``), ``PyMaSC/handler/calc.py:218`` (``assert isinstance(obj, ChromResult)`` on what a
worker reports) and ``PyMaSC/result.py:329-351`` (``aggregate_results``).  What ``CCHipCalculator`` returns is then
exactly what ``CCBitArrayCalculator`` returns, type included (tests/test_reference_consumers.py).

Stand-alone (no PyMaSC on the path, e.g. the GPU box) the same names are the dataclasses below: field-for-field
value-equal to the reference's (PyMaSC/result.py:68-118 per chromosome, :121-126 BothChromResult, :144-257 empty
placeholders, :260-298 genome-wide), plain and picklable (they cross a multiprocessing.Queue, worker.py:234).
``REFERENCE_TYPES`` says which of the two is bound.

The integer fields come from the GPU; ``calc_cc`` turns them into float64 on the host with the SAME
operation order as PyMaSC/result.py:42-65 (binomial-variance normalisation), which is what holds the
reference's decimal=15 table tolerance.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Mapping, Optional, Tuple, Union

import numpy as np


def normalised_cc(forward_sum, reverse_sum, ccbins, totlen, denom) -> np.ndarray:
    """(ccbins/denom - f*r) / sqrt(f(1-f) r(1-r)) with f = forward_sum/totlen, r = reverse_sum/totlen.

    Mirrors PyMaSC/result.py:42-65 including the all-zero -> all-NaN rule (:54-55)."""
    bins = np.array(ccbins, dtype=np.int64)
    if bins.sum() == 0:
        return np.full_like(bins, np.nan, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        fmean = forward_sum / totlen
        rmean = reverse_sum / totlen
        fvar = fmean * (1 - fmean)
        rvar = rmean * (1 - rmean)
        prod = fmean * rmean
        geo = (fvar * rvar) ** 0.5
        return (bins / denom - prod) / geo


@dataclass
class NCCResult:
    """Naive cross-correlation of one chromosome (reference: result.py:68-89)."""
    max_shift: int
    read_len: int
    genomelen: int
    forward_sum: int
    reverse_sum: int
    forward_read_len_sum: int
    reverse_read_len_sum: int
    ccbins: Union[List[float], np.ndarray]
    cc: np.ndarray = field(init=False, default=None)

    def calc_cc(self) -> None:
        denom = self.genomelen - np.array(range(self.max_shift + 1), dtype=np.float64)
        self.cc = normalised_cc(float(self.forward_sum), float(self.reverse_sum),
                                self.ccbins[:self.max_shift + 1], self.genomelen, denom)


@dataclass
class MSCCResult:
    """Mappability-sensitive cross-correlation of one chromosome (reference: result.py:92-118).

    ``mappable_len`` is indexed by LAG like the reference's list (mscc.pyx:271,292-298)."""
    max_shift: int
    read_len: int
    genomelen: int
    forward_sum: Union[List[int], np.ndarray]
    reverse_sum: Union[List[int], np.ndarray]
    forward_read_len_sum: Optional[int]
    reverse_read_len_sum: Optional[int]
    ccbins: Union[List[float], np.ndarray]
    mappable_len: Optional[Union[Tuple[int, ...], List[Optional[int]]]]
    cc: np.ndarray = field(init=False, default=None)

    def calc_cc(self) -> None:
        assert self.mappable_len is not None, "mappable_len must be set before calculating CC."
        lag = np.array(self.mappable_len, dtype=np.float64)
        # lag table -> per-shift denominator T[d] = mappable_len[|d - (L-1)|]  (result.py:107-110)
        totlen = np.concatenate((lag[:self.read_len][::-1], lag[1:]))[:self.max_shift + 1]
        self.cc = normalised_cc(np.array(self.forward_sum[:self.max_shift + 1], dtype=np.float64),
                                np.array(self.reverse_sum[:self.max_shift + 1], dtype=np.float64),
                                self.ccbins[:self.max_shift + 1], totlen, totlen)


@dataclass
class BothChromResult:
    """What a worker reports per chromosome (reference: result.py:121-126, mscc.pyx:441-447)."""
    chrom: Optional[NCCResult]
    mappable_chrom: Optional[MSCCResult]


class EmptyResult:
    """Marker base of placeholders for chromosomes without reads (reference: result.py:129-141)."""


@dataclass
class EmptyNCCResult(EmptyResult, NCCResult):
    @classmethod
    def create_empty(cls, genome_length: int, max_shift: int, read_len: int) -> "EmptyNCCResult":
        res = cls(max_shift=max_shift, read_len=read_len, genomelen=genome_length, forward_sum=0, reverse_sum=0,
                  forward_read_len_sum=0, reverse_read_len_sum=0, ccbins=[0.0] * (max_shift + 1))
        res.calc_cc()
        return res


@dataclass
class EmptyMSCCResult(EmptyResult, MSCCResult):
    @classmethod
    def create_empty(cls, genome_length: int, max_shift: int, read_len: int) -> "EmptyMSCCResult":
        res = cls(max_shift=max_shift, read_len=read_len, genomelen=genome_length,
                  forward_sum=np.zeros(max_shift + 1, dtype=np.int64),
                  reverse_sum=np.zeros(max_shift + 1, dtype=np.int64),
                  forward_read_len_sum=0, reverse_read_len_sum=0, ccbins=[0.0] * (max_shift + 1),
                  mappable_len=tuple([0] * (max_shift + 1)))
        res.calc_cc()
        return res


@dataclass
class EmptyBothChromResult(EmptyResult, BothChromResult):
    @classmethod
    def create_empty(cls, genome_length: int, max_shift: int, read_len: int) -> "EmptyBothChromResult":
        return cls(chrom=EmptyNCCResult.create_empty(genome_length, max_shift, read_len),
                   mappable_chrom=EmptyMSCCResult.create_empty(genome_length, max_shift, read_len))


@dataclass
class NCCGenomeWideResult:
    genomelen: int
    forward_read_len_sum: int
    reverse_read_len_sum: int
    forward_sum: int
    reverse_sum: int
    chroms: Dict[str, NCCResult]


@dataclass
class MSCCGenomeWideResult:
    genomelen: int
    forward_read_len_sum: int
    reverse_read_len_sum: int
    chroms: Dict[str, MSCCResult]


@dataclass
class BothGenomeWideResult:
    genomelen: int
    forward_read_len_sum: int
    reverse_read_len_sum: int
    forward_sum: int
    reverse_sum: int
    chroms: Dict[str, NCCResult]
    mappable_chroms: Dict[str, MSCCResult]


GenomeWideResult = Union[NCCGenomeWideResult, MSCCGenomeWideResult, BothGenomeWideResult]


def _sum_ncc(results: Mapping[str, NCCResult]) -> NCCGenomeWideResult:
    vals = list(results.values())
    return NCCGenomeWideResult(
        genomelen=sum(r.genomelen for r in vals),
        forward_read_len_sum=sum(r.forward_read_len_sum for r in vals),
        reverse_read_len_sum=sum(r.reverse_read_len_sum for r in vals),
        forward_sum=sum(r.forward_sum for r in vals),
        reverse_sum=sum(r.reverse_sum for r in vals),
        chroms=dict(results))


def _sum_mscc(results: Mapping[str, MSCCResult]) -> MSCCGenomeWideResult:
    vals = list(results.values())
    return MSCCGenomeWideResult(
        genomelen=sum(r.genomelen for r in vals),
        forward_read_len_sum=sum(r.forward_read_len_sum for r in vals),
        reverse_read_len_sum=sum(r.reverse_read_len_sum for r in vals),
        chroms=dict(results))


def aggregate_results(results: Mapping[str, Any]) -> GenomeWideResult:
    """Per-chromosome worker results -> genome-wide result (reference: result.py:301-356,360-464).

    Integer totals are summed; the per-chromosome rows are KEPT because the reference's genome-wide
    curve is a Fisher-z merge of per-chromosome cc (utils/calc.py:172-241), not a ratio of summed bins."""
    if not results:
        raise ValueError("Cannot aggregate empty results dictionary")
    first = next(iter(results.values()))
    if isinstance(first, BothChromResult):
        real = [r for r in results.values() if not isinstance(r, EmptyResult)]
        if all(r.chrom is None for r in real):
            return _sum_mscc({c: r.mappable_chrom for c, r in results.items()})
        if all(r.mappable_chrom is None for r in real):
            return _sum_ncc({c: r.chrom for c, r in results.items()})
        ncc = _sum_ncc({c: r.chrom for c, r in results.items()})
        mscc = _sum_mscc({c: r.mappable_chrom for c, r in results.items()})
        return BothGenomeWideResult(genomelen=ncc.genomelen, forward_read_len_sum=ncc.forward_read_len_sum,
                                    reverse_read_len_sum=ncc.reverse_read_len_sum, forward_sum=ncc.forward_sum,
                                    reverse_sum=ncc.reverse_sum, chroms=ncc.chroms, mappable_chroms=mscc.chroms)
    if isinstance(first, NCCResult):
        return _sum_ncc(results)
    if isinstance(first, MSCCResult):
        return _sum_mscc(results)
    raise TypeError(f"Unknown result type: {type(first)}")


def calc_cc_batch(ncc: List["NCCResult"], mscc: List["MSCCResult"], ncc_bins: Optional[np.ndarray] = None,
                  mscc_rows: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]] = None) -> None:
    """calc_cc() of many chromosomes' results at once: the same elementwise float64 operations as normalised_cc, row by
    row of one 2-D array instead of a Python call (and a handful of small numpy calls) per chromosome -- a genome's worth
    of results takes a fraction of a millisecond instead of a few.  Bit-identical values (no reductions are reordered).
    Inside a PyMaSC installation the results are the reference's own objects: their own calc_cc() is called.
    ncc_bins [len(ncc), S+1] / mscc_rows = (ccbins, forward_sum, reverse_sum, mappable length by SHIFT), each
    [len(mscc), S+1]: the integer rows the results were built from, when the caller still holds them as arrays (all results
    of one geometry then) -- saves turning the results' lists back into arrays."""
    if REFERENCE_TYPES:
        for r in list(ncc) + list(mscc):
            r.calc_cc()
        return

    def rows(cc2d, zero, results):
        for r, row, z in zip(results, cc2d, zero):
            r.cc = np.full(row.shape, np.nan, dtype=np.float64) if z else row.copy()

    with np.errstate(divide="ignore", invalid="ignore"):
        if ncc_bins is not None and len(ncc):
            bins = np.asarray(ncc_bins, dtype=np.int64)
            glen = np.array([r.genomelen for r in ncc], dtype=np.float64)[:, None]
            denom = glen - np.arange(bins.shape[1], dtype=np.float64)[None, :]
            fmean = np.array([float(r.forward_sum) for r in ncc])[:, None] / glen
            rmean = np.array([float(r.reverse_sum) for r in ncc])[:, None] / glen
            geo = ((fmean * (1 - fmean)) * (rmean * (1 - rmean))) ** 0.5       # (one value per chromosome)
            # (bins / denom - fmean * rmean) / geo, in place: a [chromosomes, shifts] float64 temporary is 200 KB for a genome
            # at 1000 shifts -- above the allocator's mmap threshold, so every temporary of the plain expression costs a
            # mapping and its page faults; the same operations in the same order on one buffer
            cc = np.divide(bins, denom, out=denom)
            cc -= fmean * rmean
            cc /= geo
            rows(cc, bins.sum(axis=1) == 0, ncc)
            ncc = []
        if mscc_rows is not None and len(mscc):
            bins = np.asarray(mscc_rows[0], dtype=np.int64)
            tot = np.asarray(mscc_rows[3], dtype=np.float64)
            fmean = np.asarray(mscc_rows[1], dtype=np.float64)
            fmean /= tot
            rmean = np.asarray(mscc_rows[2], dtype=np.float64)
            rmean /= tot
            # geo = ((fmean * (1 - fmean)) * (rmean * (1 - rmean))) ** 0.5 and (bins / tot - fmean * rmean) / geo: the same
            # operations in the same order (IEEE products commute; x ** 0.5 IS numpy's sqrt), on three buffers instead of ten
            geo = np.subtract(1, fmean)
            geo *= fmean
            tmp = np.subtract(1, rmean)
            tmp *= rmean
            geo *= tmp
            np.sqrt(geo, out=geo)
            cc = np.divide(bins, tot, out=tmp)
            fmean *= rmean
            cc -= fmean
            cc /= geo
            rows(cc, bins.sum(axis=1) == 0, mscc)
            mscc = []
        for group in _by_shape(ncc):
            S1 = group[0].max_shift + 1
            bins = np.array([r.ccbins[:S1] for r in group], dtype=np.int64)
            glen = np.array([r.genomelen for r in group], dtype=np.float64)[:, None]
            denom = glen - np.arange(S1, dtype=np.float64)[None, :]
            fmean = np.array([float(r.forward_sum) for r in group])[:, None] / glen
            rmean = np.array([float(r.reverse_sum) for r in group])[:, None] / glen
            geo = ((fmean * (1 - fmean)) * (rmean * (1 - rmean))) ** 0.5
            rows((bins / denom - fmean * rmean) / geo, bins.sum(axis=1) == 0, group)
        for group in _by_shape(mscc):
            S1, L = group[0].max_shift + 1, group[0].read_len
            bins = np.array([r.ccbins[:S1] for r in group], dtype=np.int64)
            lag = [np.array(r.mappable_len, dtype=np.float64) for r in group]
            tot = np.array([np.concatenate((g[:L][::-1], g[1:]))[:S1] for g in lag])
            fmean = np.array([r.forward_sum[:S1] for r in group], dtype=np.float64) / tot
            rmean = np.array([r.reverse_sum[:S1] for r in group], dtype=np.float64) / tot
            geo = ((fmean * (1 - fmean)) * (rmean * (1 - rmean))) ** 0.5
            rows((bins / tot - fmean * rmean) / geo, bins.sum(axis=1) == 0, group)


def _by_shape(results):
    groups: Dict[Tuple[int, int, int], list] = {}
    for r in results:
        n = len(r.mappable_len) if hasattr(r, "mappable_len") else 0
        groups.setdefault((r.max_shift, r.read_len, n), []).append(r)
    return list(groups.values())


# ---- inside a PyMaSC installation: emit the reference's own types ------------------------------------------------
REFERENCE_TYPES = False
try:
    import PyMaSC.result as _ref
except Exception:          # stand-alone: the dataclasses above
    _ref = None
if _ref is not None:
    _standalone = {k: v for k, v in globals().items() if isinstance(v, type) and v.__module__ == __name__}
    NCCResult, MSCCResult, BothChromResult = _ref.NCCResult, _ref.MSCCResult, _ref.BothChromResult
    EmptyResult, EmptyNCCResult, EmptyMSCCResult = _ref.EmptyResult, _ref.EmptyNCCResult, _ref.EmptyMSCCResult
    EmptyBothChromResult = _ref.EmptyBothChromResult
    NCCGenomeWideResult, MSCCGenomeWideResult = _ref.NCCGenomeWideResult, _ref.MSCCGenomeWideResult
    BothGenomeWideResult = _ref.BothGenomeWideResult
    GenomeWideResult = Union[NCCGenomeWideResult, MSCCGenomeWideResult, BothGenomeWideResult]
    aggregate_results = _ref.aggregate_results
    REFERENCE_TYPES = True
