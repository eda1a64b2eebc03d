"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the PyMaSC BitArray cross-correlation path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package (``pymasc_amd``) never does.

It restates, on the CPU, what the reference does between ``feed_*_read`` and the result objects:

* host logic of ``CCBitArrayCalculator`` -- /root/reference/PyMaSC/core/bitarray/mscc.pyx:97-483
  (vector sizing :134,:165-167; chromosome switching / sortedness :351-366; forward dedup :388-393;
  reverse dedup :416-418; mappability load :327-349; lag re-indexing of mappable_len :271,:292-298;
  read-less chromosomes :181-215; finishup :420-439),
* the per-shift loop itself through ``cc_oracle.c`` (compiled to ``oracle/libcc_oracle.so``),
* ``calc_cc`` -- /root/reference/PyMaSC/result.py:42-65 (formula), :80-89 (NCC), :104-118 (MSCC).

Parity pin (see tests/test_oracle_golden.py): the reference's own goldens
``tests/golden/ENCFF000RMB-test_{cc,mscc,nreads}.tab`` and ``hg19_36mer-test_mappability.json``
(copied as data fixtures into tests/golden/), plus vectors generated here by the reference's
compiled ``successive`` NCC calculator (tests/golden/ref_successive_ncc.json, made by
oracle/make_ref_vectors.py).  The BitArray C library the reference links is not in
/root/reference (empty submodule), so the reference's bitarray extension itself cannot be built.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcc_oracle.so")
_lib = None

MAPPABILITY_THRESHOLD = 1.0   # mscc.pyx:115
EXTRA_ALLOCATE_SIZE = 100     # mscc.pyx:117


class OracleReadUnsortedError(IndexError):
    """Restates PyMaSC/core/exceptions.py:4 (ReadUnsortedError(IndexError))."""


ORACLE_CFLAGS = ["-O3", "-mpopcnt", "-fPIC", "-shared"]     # reported by bench.py's cpu_baseline


def build_oracle_lib(force: bool = False) -> str:
    src = os.path.join(_HERE, "cc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", *ORACLE_CFLAGS, "-o", _LIB_PATH, src])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build_oracle_lib()
        L = ctypes.CDLL(_LIB_PATH)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.pmo_calc_correlation.restype = ctypes.c_int
        L.pmo_calc_correlation.argtypes = [u64p, u64p, u64p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int, u64p, u64p, u64p, u64p, u64p, u64p, u64p]
        L.pmo_mappable_len.restype = ctypes.c_int
        L.pmo_mappable_len.argtypes = [u64p, ctypes.c_uint64, ctypes.c_int64, u64p]
        L.pmo_set_region.restype = None
        L.pmo_set_region.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint64]
        L.pmo_count.restype = ctypes.c_uint64
        L.pmo_count.argtypes = [u64p, ctypes.c_uint64]
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def nwords(nbits: int) -> int:
    return (int(nbits) + 63) // 64


def bits_from_positions(positions: Iterable[int], nbits: int) -> np.ndarray:
    """bitarray.__setitem__ per position (bitarray.pyx:72-79)."""
    w = np.zeros(nwords(nbits), dtype=np.uint64)
    pos = np.asarray(list(positions) if not isinstance(positions, np.ndarray) else positions, dtype=np.int64)
    if pos.size:
        assert pos.min() >= 0 and pos.max() < nbits
        np.bitwise_or.at(w, pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))
    return w


def bits_from_intervals(intervals: Iterable[Tuple[int, int]], nbits: int) -> np.ndarray:
    """mscc.pyx:343-344: for (begin, end, val) in feeder: mappability.set(begin + 1, end)."""
    w = np.zeros(nwords(nbits), dtype=np.uint64)
    L = lib()
    for begin, end in intervals:
        if end >= begin + 1:
            L.pmo_set_region(_p(w), int(begin) + 1, int(end))
    return w


def calc_correlation(F: np.ndarray, R: np.ndarray, M: Optional[np.ndarray], nbits: int, max_shift: int,
                     read_len: int, skip_ncc: bool = False) -> Dict[str, object]:
    """One chromosome through cc_oracle.c (mscc.pyx:217-325). Returns raw integer arrays by shift d."""
    S = int(max_shift)
    z = lambda: np.zeros(S + 1, dtype=np.uint64)
    fs = np.zeros(1, dtype=np.uint64)
    rs = np.zeros(1, dtype=np.uint64)
    ncc, mf, mr, mc, ml = z(), z(), z(), z(), z()
    rc = lib().pmo_calc_correlation(_p(F), _p(R), _p(M), int(nbits), S, int(read_len), int(bool(skip_ncc)),
                                    _p(fs), _p(rs), _p(ncc), _p(mf), _p(mr), _p(mc), _p(ml))
    if rc != 0:
        raise MemoryError("pmo_calc_correlation failed")
    out: Dict[str, object] = {}
    if not skip_ncc:
        out["ncc_forward_sum"] = int(fs[0])
        out["ncc_reverse_sum"] = int(rs[0])
        out["ncc_ccbins"] = ncc.astype(np.int64)
    if M is not None:
        out["mscc_forward_sum"] = mf.astype(np.int64)
        out["mscc_reverse_sum"] = mr.astype(np.int64)
        out["mscc_ccbins"] = mc.astype(np.int64)
        out["mappable_len_by_shift"] = ml.astype(np.int64)
    return out


def mappable_len_readless(M: np.ndarray, nbits: int, max_shift: int) -> np.ndarray:
    """mscc.pyx:207-215 through cc_oracle.c."""
    out = np.zeros(int(max_shift) + 1, dtype=np.uint64)
    rc = lib().pmo_mappable_len(_p(M), int(nbits), int(max_shift), _p(out))
    if rc != 0:
        raise MemoryError("pmo_mappable_len failed")
    return out.astype(np.int64)


def mappable_len_by_lag(by_shift: Sequence[int], max_shift: int, read_len: int) -> List[Optional[int]]:
    """mscc.pyx:271,292-298: the list the reference stores, indexed by lag."""
    L = int(read_len)
    out: List[Optional[int]] = [None] * L
    for i in range(int(max_shift) + 1):
        if i < L:
            out[L - i - 1] = int(by_shift[i])
        elif i < L * 2 - 1:
            pass
        else:
            out.append(int(by_shift[i]))
    return out


# ---- calc_cc (result.py) -------------------------------------------------------------------------

def _calc_cc(forward_sum, reverse_sum, ccbins, totlen, denom) -> np.ndarray:
    """result.py:42-65, same operation order."""
    ccbins = np.array(ccbins, dtype=np.int64)
    if ccbins.sum() == 0:
        return np.full_like(ccbins, np.nan, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        forward_mean = forward_sum / totlen
        reverse_mean = reverse_sum / totlen
        forward_var = forward_mean * (1 - forward_mean)
        reverse_var = reverse_mean * (1 - reverse_mean)
        sum_prod = forward_mean * reverse_mean
        var_geomean = (forward_var * reverse_var) ** 0.5
        return (ccbins / denom - sum_prod) / var_geomean


def ncc_cc(forward_sum: int, reverse_sum: int, ccbins, genomelen: int, max_shift: int) -> np.ndarray:
    """result.py:80-89."""
    denom = genomelen - np.array(range(max_shift + 1), dtype=np.float64)
    return _calc_cc(float(forward_sum), float(reverse_sum), list(ccbins)[:max_shift + 1], genomelen, denom)


def mscc_cc(forward_sum, reverse_sum, ccbins, mappable_len, max_shift: int, read_len: int) -> np.ndarray:
    """result.py:104-118."""
    totlen = np.array(mappable_len, dtype=np.float64)
    totlen = np.concatenate((totlen[:read_len][::-1], totlen[1:]))[:max_shift + 1]
    return _calc_cc(np.array(forward_sum[:max_shift + 1], dtype=np.float64),
                    np.array(reverse_sum[:max_shift + 1], dtype=np.float64),
                    list(ccbins)[:max_shift + 1], totlen, totlen)


# ---- host logic of CCBitArrayCalculator ----------------------------------------------------------

class OracleCalculator:
    """CPU restatement of CCBitArrayCalculator (mscc.pyx:41-483) returning plain dicts.

    ``mappability``: optional dict chrom -> list of (begin, end, value) intervals, i.e. what
    ``BigWigReader.fetch`` yields before filtering (reader/bigwig.pyx:147-177); chromosomes missing
    from the dict behave like a KeyError from ``fetch`` (mscc.pyx:254-259, :200-205).
    """

    def __init__(self, max_shift: int, read_len: int, references: Sequence[str], lengths: Sequence[int],
                 mappability: Optional[Dict[str, Sequence[Tuple[int, int, float]]]] = None,
                 skip_ncc: bool = False):
        self.max_shift = int(max_shift)
        self.read_len = int(read_len)
        self.references = list(references)
        self.ref2genomelen = dict(zip(references, lengths))
        self.genomelen = int(sum(lengths))
        self.skip_ncc = bool(skip_ncc)
        self.mappability = mappability
        self.ref2ncc: Dict[str, dict] = {}
        self.ref2mscc: Dict[str, dict] = {}
        self.forward_sum = self.reverse_sum = 0
        self.forward_read_len_sum = self.reverse_read_len_sum = 0
        self._chr = ""
        self._solved: List[str] = []
        self._flushed = False
        self._ext = self.read_len + self.max_shift + EXTRA_ALLOCATE_SIZE   # :134

    # :161-171
    def _init_buff(self):
        self._nbits = self.ref2genomelen[self._chr] + self._ext
        self._F = np.zeros(nwords(self._nbits), dtype=np.uint64)
        self._R = np.zeros(nwords(self._nbits), dtype=np.uint64)
        self._last_pos = 0
        self._last_fpos = 0
        self._f_rls = self._r_rls = 0

    # :327-349
    def _load_mappability(self, chrom: str) -> Optional[np.ndarray]:
        if self.mappability is None:
            return None
        if chrom not in self.mappability:
            raise KeyError(chrom)
        nbits = self.ref2genomelen[chrom] + self._ext
        iv = [(b, e) for (b, e, v) in self.mappability[chrom] if np.float32(v) >= np.float32(MAPPABILITY_THRESHOLD)]
        return bits_from_intervals(iv, nbits)

    # :351-366
    def _check_pos(self, chrom: str, pos: int):
        if chrom != self._chr:
            if self._chr != "":
                if chrom in self._solved:
                    raise OracleReadUnsortedError
                self._solved.append(self._chr)
                self.flush()
                self._flushed = False
            self._chr = chrom
            self._init_buff()
        if pos < self._last_pos:
            raise OracleReadUnsortedError
        self._last_pos = pos

    # :370-393
    def feed_forward_read(self, chrom: str, pos: int, readlen: int):
        self._check_pos(chrom, pos)
        if self._last_fpos == pos:
            return
        self._last_fpos = pos
        self._f_rls += readlen
        self._F[pos >> 6] |= np.uint64(1) << np.uint64(pos & 63)

    # :397-418
    def feed_reverse_read(self, chrom: str, pos: int, readlen: int):
        self._check_pos(chrom, pos)
        p = pos + readlen - 1
        if not (int(self._R[p >> 6]) >> (p & 63)) & 1:
            self._R[p >> 6] |= np.uint64(1) << np.uint64(p & 63)
            self._r_rls += readlen

    # :173-179
    def flush(self, chrom: Optional[str] = None):
        if self._chr != "" and not self._flushed:
            self._calc_correlation()
        if chrom is not None:
            self._fill_result(chrom)
        self._flushed = True

    # :217-325
    def _calc_correlation(self):
        c = self._chr
        self.forward_read_len_sum += self._f_rls
        self.reverse_read_len_sum += self._r_rls
        try:
            M = self._load_mappability(c)
        except KeyError:
            M = None
        raw = calc_correlation(self._F, self._R, M, self._nbits, self.max_shift, self.read_len, self.skip_ncc)
        glen = self.ref2genomelen[c]
        if not self.skip_ncc:
            self.forward_sum += raw["ncc_forward_sum"]
            self.reverse_sum += raw["ncc_reverse_sum"]
            self.ref2ncc[c] = dict(
                max_shift=self.max_shift, read_len=self.read_len, genomelen=glen,
                forward_sum=raw["ncc_forward_sum"], reverse_sum=raw["ncc_reverse_sum"],
                forward_read_len_sum=self._f_rls, reverse_read_len_sum=self._r_rls,
                ccbins=[int(x) for x in raw["ncc_ccbins"]],
                cc=ncc_cc(raw["ncc_forward_sum"], raw["ncc_reverse_sum"], raw["ncc_ccbins"], glen, self.max_shift))
        if M is not None:
            mlen = mappable_len_by_lag(raw["mappable_len_by_shift"], self.max_shift, self.read_len)
            fs = [int(x) for x in raw["mscc_forward_sum"]]
            rs = [int(x) for x in raw["mscc_reverse_sum"]]
            cb = [int(x) for x in raw["mscc_ccbins"]]
            self.ref2mscc[c] = dict(
                max_shift=self.max_shift, read_len=self.read_len, genomelen=glen,
                forward_sum=fs, reverse_sum=rs,
                forward_read_len_sum=self._f_rls, reverse_read_len_sum=self._r_rls,
                ccbins=cb, mappable_len=mlen,
                cc=mscc_cc(fs, rs, cb, mlen, self.max_shift, self.read_len))

    # :181-215
    def _fill_result(self, chrom: str):
        self._chr = chrom
        S = self.max_shift
        glen = self.ref2genomelen[chrom]
        if chrom not in self.ref2ncc:
            zeros = [0.0] * (S + 1)
            self.ref2ncc[chrom] = dict(
                max_shift=S, read_len=self.read_len, genomelen=glen, forward_sum=0, reverse_sum=0,
                forward_read_len_sum=0, reverse_read_len_sum=0, ccbins=zeros,
                cc=ncc_cc(0, 0, zeros, glen, S), empty=True)
        if self.mappability is None or chrom in self.ref2mscc:
            return
        zeros = [0.0] * (S + 1)
        res = self.ref2mscc[chrom] = dict(
            max_shift=S, read_len=self.read_len, genomelen=glen,
            forward_sum=np.zeros(S + 1, dtype=np.int64), reverse_sum=np.zeros(S + 1, dtype=np.int64),
            forward_read_len_sum=0, reverse_read_len_sum=0, ccbins=zeros,
            mappable_len=tuple([0] * (S + 1)), empty=True)
        res["cc"] = mscc_cc(res["forward_sum"], res["reverse_sum"], zeros, res["mappable_len"], S, self.read_len)
        try:
            M = self._load_mappability(chrom)
        except KeyError:
            return
        nbits = glen + self._ext
        res["mappable_len"] = tuple(int(x) for x in mappable_len_readless(M, nbits, S))

    # :420-439
    def finishup_calculation(self):
        self.flush(self._chr)
        for chrom in self.references:
            self._fill_result(chrom)

    # :441-447
    def get_result(self, chrom: str):
        if chrom not in self.ref2ncc and chrom not in self.ref2mscc:
            raise KeyError(chrom)
        return self.ref2ncc.get(chrom), self.ref2mscc.get(chrom)
