"""CCHipCalculator -- drop-in for the reference's ``CCBitArrayCalculator`` on one MI355X.

Same constructor and the same six-method calculator protocol
(PyMaSC/interfaces/calculator.py:15-79,110-124; PyMaSC/core/bitarray/mscc.pyx:97-98,173-179,370-483),
same exceptions (ReadUnsortedError / KeyError) and value-equal result objects, so
``PyMaSC/handler/factory.py:224-258`` can construct it instead of the Cython class.

What is different underneath (MI355X-first, not a translation):
* reads are not written into a host bit array one Python call at a time; positions are appended to
  host arrays and, when the chromosome ends, uploaded once and scattered into HBM bit-vectors by a
  HIP kernel (``pmx_bits_set_positions``); mappability intervals go through ``pmx_bits_set_regions``;
* the (max_shift+1)-iteration loop of full-vector passes (mscc.pyx:288-317) is ONE call,
  ``pmx_cc_dev``, whose kernels read each vector from HBM once;
* read-length sums and duplicate rules (mscc.pyx:388-392, :416-418) are evaluated vectorised on the
  host at flush time from the same position arrays.

There is no CPU compute fallback: constructing the calculator without a visible GPU raises.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import ffi
from .exceptions import ReadUnsortedError
from .result import (BothChromResult, BothGenomeWideResult, EmptyMSCCResult, EmptyNCCResult, MSCCGenomeWideResult,
                     MSCCResult, NCCGenomeWideResult, NCCResult)

logger = logging.getLogger(__name__)

_CHUNK = 1 << 16
DENSE_READS_PER_BP = 0.0105      # reads of one strand per position above which PMX_FLAG_WINDOW_ONLY is passed
DENSE_RUNS_PER_BP = 0.0025       # mappable runs per position above which it is passed (the event kernel lists 384 edges per 64 Kbit)


class _PosBuffer:
    """Append-only (position, readlen) arrays; amortised O(1) per read, no Python list of ints."""

    __slots__ = ("pos", "rlen", "n")

    def __init__(self):
        self.pos = np.empty(_CHUNK, dtype=np.int64)
        self.rlen = np.empty(_CHUNK, dtype=np.int64)
        self.n = 0

    def append(self, pos: int, rlen: int):
        if self.n == self.pos.size:
            self.pos = np.concatenate([self.pos, np.empty(self.pos.size, dtype=np.int64)])
            self.rlen = np.concatenate([self.rlen, np.empty(self.rlen.size, dtype=np.int64)])
        self.pos[self.n] = pos
        self.rlen[self.n] = rlen
        self.n += 1

    def extend(self, pos: np.ndarray, rlen: np.ndarray):
        k = pos.size
        while self.n + k > self.pos.size:
            self.pos = np.concatenate([self.pos, np.empty(self.pos.size, dtype=np.int64)])
            self.rlen = np.concatenate([self.rlen, np.empty(self.rlen.size, dtype=np.int64)])
        self.pos[self.n:self.n + k] = pos
        self.rlen[self.n:self.n + k] = rlen
        self.n += k

    def view(self):
        return self.pos[:self.n], self.rlen[:self.n]


class CCHipCalculator:
    """NCC + MSCC calculator backed by libpymasc_hip.so (see module docstring)."""

    MAPPABILITY_THRESHOLD = 1.0     # mscc.pyx:115
    EXTRA_ALLOCATE_SIZE = 100       # mscc.pyx:117

    def __init__(self, max_shift: int, read_len: int, references: Sequence[str], lengths: Sequence[int],
                 bwfeeder: Any = None, skip_ncc: bool = False, logger_lock: Any = None, progress_bar: Any = None,
                 device: int = 0, kernel_flags: int = 0, context: Optional[ffi.Context] = None,
                 chrom2mappable_len: Optional[Dict[str, Sequence[int]]] = None):
        self.max_shift = int(max_shift)
        self.read_len = int(read_len)
        self.references = list(references)
        self.ref2genomelen: Dict[str, int] = dict(zip(references, (int(x) for x in lengths)))
        self.genomelen = int(sum(lengths))
        self.skip_ncc = bool(skip_ncc)
        self.logger_lock = logger_lock
        self._bwfeeder = bwfeeder
        self._progress = progress_bar
        self._kernel_flags = int(kernel_flags)
        # the result block of the C ABI keeps four scalars in a row of max_shift + 1 words, so the kernels are run
        # with at least 3 shifts; shifts are independent, rows are cut back to max_shift + 1 below
        self._kshift = max(self.max_shift, 3)
        # lag tables known beforehand (the *_mappability.json cache, pymasc_amd/mappability.py): chromosomes
        # found here skip the autocorrelation pass
        self._known_mlen: Dict[str, Sequence[int]] = dict(chrom2mappable_len or {})

        self.ref2ncc_result: Dict[str, NCCResult] = {}
        self.ref2mscc_result: Dict[str, MSCCResult] = {}
        self.forward_sum = self.reverse_sum = 0
        self.forward_read_len_sum = self.reverse_read_len_sum = 0

        self._chr = ""
        self._solved_chr: List[str] = []
        self._buff_flashed = False
        self._array_extend_size = self.read_len + self.max_shift + self.EXTRA_ALLOCATE_SIZE   # mscc.pyx:134
        self._last_pos = 0
        self._fwd = _PosBuffer()
        self._rev = _PosBuffer()

        # the one and only compute back-end; raises ffi.PmxError when no GPU / no library
        self._ctx = context if context is not None else ffi.Context(device)
        self._own_ctx = context is None
        self._dev_bits: Dict[str, List[int]] = {}     # name -> [device pointer, capacity in bits]
        self._d_out = 0
        self._d_out_words = 0

    # ---- plumbing ------------------------------------------------------------------------------
    def close(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is None:
            return
        for ptr, _cap in self._dev_bits.values():
            ctx.bits_free(ptr)
        self._dev_bits.clear()
        if self._d_out:
            ctx.bits_free(self._d_out)
            self._d_out = 0
        if self._own_ctx:
            ctx.close()
        self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _logging_info(self, msg: str):
        if self.logger_lock:
            self.logger_lock.acquire()
        logger.info(msg)
        if self.logger_lock:
            self.logger_lock.release()

    def _device_vector(self, name: str, nbits: int) -> int:
        """A zeroed HBM bit-vector of >= nbits, reused across chromosomes (grow-only)."""
        slot = self._dev_bits.get(name)
        if slot is None or slot[1] < nbits:
            if slot is not None:
                self._ctx.bits_free(slot[0])
            cap = int(nbits * 1.25) + 4096 if slot is not None else int(nbits)
            self._dev_bits[name] = slot = [self._ctx.bits_alloc(cap), cap]
        else:
            self._ctx.bits_clear(slot[0], slot[1])
        return slot[0]

    def _device_out(self, words: int) -> int:
        if self._d_out_words < words:
            if self._d_out:
                self._ctx.bits_free(self._d_out)
            self._d_out = self._ctx.bits_alloc(words * 64)
            self._d_out_words = words
        return self._d_out

    # ---- feeding (mscc.pyx:351-418) --------------------------------------------------------------
    def _init_buff(self):
        self._last_pos = 0
        self._fwd.n = 0
        self._rev.n = 0

    def _check_pos(self, chrom: str, pos: int, bit: Optional[int] = None, what: str = ""):
        if chrom != self._chr:
            if self._chr != "":
                if chrom in self._solved_chr:
                    raise ReadUnsortedError
                self._solved_chr.append(self._chr)
                self.flush()
                self._buff_flashed = False
            self._chr = chrom
            self._init_buff()
            self._cur_nbits = self.ref2genomelen[chrom] + self._array_extend_size   # mscc.pyx:165-167
            self._logging_info("Loading {} reads to bit array...".format(chrom))
        if bit is not None:
            self._check_bit(chrom, bit, what)
        if pos < self._last_pos:
            raise ReadUnsortedError
        self._last_pos = pos

    def _check_bit(self, chrom: str, bit: int, what: str):
        """The reference sets bit_array[bit] unchecked (bitarray.pyx:72-79: undefined beyond the array); here a read whose
        bit falls outside the chromosome's vector is refused when it is fed, before anything is queued for the GPU."""
        if bit < 0 or bit >= self._cur_nbits:
            raise IndexError("{} read beyond the bit array of {} (position {}, {} bits)".format(
                what, chrom, bit, self._cur_nbits))

    def feed_forward_read(self, chrom: str, pos: int, readlen: int) -> None:
        """1-based 5' position of a forward read (mscc.pyx:370-393)."""
        self._check_pos(chrom, pos, pos, "forward")
        self._fwd.append(pos, readlen)

    def feed_reverse_read(self, chrom: str, pos: int, readlen: int) -> None:
        """1-based leftmost position of a reverse read; its bit is pos + readlen - 1 (mscc.pyx:397-418)."""
        self._check_pos(chrom, pos, pos + readlen - 1, "reverse")
        self._rev.append(pos, readlen)

    def feed_reads(self, chrom: str, pos: np.ndarray, readlen: np.ndarray, is_reverse: np.ndarray) -> None:
        """Bulk variant of the two methods above for one chromosome: arrays in file order.
        Not part of the reference protocol; it removes the per-read Python call for vectorised readers."""
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        readlen = np.ascontiguousarray(readlen, dtype=np.int64)
        is_reverse = np.asarray(is_reverse, dtype=bool)
        if pos.size == 0:
            return
        self._check_pos(chrom, int(pos[0]))
        if pos.size > 1 and (np.diff(pos) < 0).any():
            raise ReadUnsortedError
        fpos, rbit = pos[~is_reverse], pos[is_reverse] + readlen[is_reverse] - 1
        if fpos.size:
            self._check_bit(chrom, int(fpos.min()), "forward")
            self._check_bit(chrom, int(fpos.max()), "forward")
        if rbit.size:
            self._check_bit(chrom, int(rbit.min()), "reverse")
            self._check_bit(chrom, int(rbit.max()), "reverse")
        self._last_pos = int(pos[-1])
        self._fwd.extend(fpos, readlen[~is_reverse])
        self._rev.extend(pos[is_reverse], readlen[is_reverse])

    # ---- per-chromosome calculation ---------------------------------------------------------------
    def _load_mappability(self, chrom: str, nbits: int) -> Optional[int]:
        """mscc.pyx:327-349 -> device vector, or None without a feeder; KeyError if the track is missing."""
        if not self._bwfeeder:
            return None
        bulk = getattr(self._bwfeeder, "fetch_arrays", None)
        if bulk is not None:          # pymasc_amd.bigwig.BigWigReader: arrays, no per-interval Python objects
            begin, end, _v = bulk(self.MAPPABILITY_THRESHOLD, chrom)
            first, last = begin.astype(np.int64) + 1, end.astype(np.int64)
        else:
            iv = [(b, e) for b, e, _v in self._bwfeeder.fetch(self.MAPPABILITY_THRESHOLD, chrom)]
            arr = np.asarray(iv, dtype=np.int64).reshape(-1, 2)
            first, last = arr[:, 0] + 1, arr[:, 1].copy()
        self._logging_info("Loading {} mappability to bit array...".format(chrom))
        d_m = self._device_vector("M", nbits)
        if first.size:
            self._ctx.bits_set_regions(d_m, nbits, first, last)   # set(begin + 1, end), mscc.pyx:343-344
        self._n_runs = int(first.size)   # two run edges per interval: the other density the event kernel's lists depend on
        return d_m

    def _calc_correlation(self):
        chrom = self._chr
        S, L = self.max_shift, self.read_len
        glen = self.ref2genomelen[chrom]
        nbits = glen + self._array_extend_size

        fpos, flen = self._fwd.view()
        rpos, rlen = self._rev.view()
        # forward: a read at the same position as the previous forward read is a duplicate (:388-392)
        if fpos.size:
            keep = np.empty(fpos.size, dtype=bool)
            keep[0] = fpos[0] != 0
            np.not_equal(fpos[1:], fpos[:-1], out=keep[1:])
            f_read_len_sum = int(flen[keep].sum())
            fbits = fpos[keep]
        else:
            f_read_len_sum, fbits = 0, fpos
        # reverse: first read to hit a 3' position sets the bit and counts (:416-418)
        if rpos.size:
            p3 = rpos + rlen - 1
            uniq, first = np.unique(p3, return_index=True)
            r_read_len_sum = int(rlen[first].sum())
            rbits = uniq
        else:
            r_read_len_sum, rbits = 0, rpos
        self.forward_read_len_sum += f_read_len_sum
        self.reverse_read_len_sum += r_read_len_sum

        for arr, what in ((fbits, "forward"), (rbits, "reverse")):
            if arr.size and (arr.min() < 0 or arr.max() >= nbits):
                raise IndexError("{} read beyond the bit array of {} (position {}, {} bits)".format(
                    what, chrom, int(arr.max()), nbits))

        d_f = self._device_vector("F", nbits)
        d_r = self._device_vector("R", nbits)
        self._ctx.bits_set_positions(d_f, nbits, fbits)
        self._ctx.bits_set_positions(d_r, nbits, rbits)

        try:
            d_m = self._load_mappability(chrom, nbits)
        except KeyError as e:
            self._logging_info("Mappability for '{}' not found. "
                               "Skip calc mappability sensitive CC.".format(e.args[0] if e.args else chrom))
            d_m = None

        self._logging_info("Calculate cross-correlation for {}...".format(chrom))
        flags = self._kernel_flags | (ffi.PMX_FLAG_SKIP_NCC if self.skip_ncc else 0)
        # deep data: above ~1 % read starts per position and strand every tile overflows the event kernel's lists
        # (EV_CAPF / EV_CAPR per 64 Kbit) -- say so instead of letting it find out (same integers either way)
        if max(fbits.size, rbits.size) > DENSE_READS_PER_BP * max(glen, 1):
            flags |= ffi.PMX_FLAG_WINDOW_ONLY
        if d_m is not None and getattr(self, "_n_runs", 0) > DENSE_RUNS_PER_BP * max(glen, 1):
            flags |= ffi.PMX_FLAG_WINDOW_ONLY   # a track of short runs (> ~330 run edges per 64 Kbit on average)
        c = L - 1
        known = self._known_mlen.get(chrom) if d_m is not None else None
        if known is not None and len(known) <= max(c, S - c):      # cache too short for this run: recompute
            known = None
        if known is not None:
            flags |= ffi.PMX_FLAG_SKIP_MLEN
        KS = self._kshift
        words = ffi.PMX_NROWS * (KS + 1)
        d_out = self._device_out(words)
        self._ctx.cc_dev(d_f, d_r, d_m, nbits, KS, L, flags, d_out)
        out = self._ctx.bits_download(d_out, words * 64).reshape(ffi.PMX_NROWS, KS + 1)

        if not self.skip_ncc:
            fsum = int(out[ffi.PMX_ROW_SCALARS, 0])
            rsum = int(out[ffi.PMX_ROW_SCALARS, 1])
            self.forward_sum += fsum
            self.reverse_sum += rsum
            res = self.ref2ncc_result[chrom] = NCCResult(
                max_shift=S, read_len=L, genomelen=glen, forward_sum=fsum, reverse_sum=rsum,
                forward_read_len_sum=f_read_len_sum, reverse_read_len_sum=r_read_len_sum,
                ccbins=[int(x) for x in out[ffi.PMX_ROW_NCC_CCBINS, :S + 1]])
            res.calc_cc()
        if d_m is not None:
            by_shift = out[ffi.PMX_ROW_MLEN] if known is None else [known[abs(c - d)] for d in range(S + 1)]
            # the reference stores mappable_len by LAG: d < L -> index L-1-d, L <= d < 2L-1 skipped
            # (same value by symmetry), d >= 2L-1 appended (mscc.pyx:271,292-298)
            mlen: List[Optional[int]] = [None] * L
            for d in range(S + 1):
                if d < L:
                    mlen[L - d - 1] = int(by_shift[d])
                elif d >= 2 * L - 1:
                    mlen.append(int(by_shift[d]))
            mres = self.ref2mscc_result[chrom] = MSCCResult(
                max_shift=S, read_len=L, genomelen=glen,
                forward_sum=[int(x) for x in out[ffi.PMX_ROW_MSCC_FSUM, :S + 1]],
                reverse_sum=[int(x) for x in out[ffi.PMX_ROW_MSCC_RSUM, :S + 1]],
                forward_read_len_sum=f_read_len_sum, reverse_read_len_sum=r_read_len_sum,
                ccbins=[int(x) for x in out[ffi.PMX_ROW_MSCC_CCBINS, :S + 1]], mappable_len=mlen)
            mres.calc_cc()

    def _fill_result(self, chrom: str):
        """Placeholders / read-less mappable_len (mscc.pyx:181-215)."""
        self._chr = chrom
        S, L = self.max_shift, self.read_len
        glen = self.ref2genomelen[chrom]
        if chrom not in self.ref2ncc_result:
            self.ref2ncc_result[chrom] = EmptyNCCResult.create_empty(glen, S, L)
        if not self._bwfeeder or chrom in self.ref2mscc_result:
            return
        result = self.ref2mscc_result[chrom] = EmptyMSCCResult.create_empty(glen, S, L)
        nbits = glen + self._array_extend_size
        known = self._known_mlen.get(chrom)
        if known is not None and len(known) > S:      # cached table long enough: no track load, no kernel
            result.mappable_len = tuple(int(x) for x in known[:S + 1])
            return
        try:
            d_m = self._load_mappability(chrom, nbits)
            if d_m is None:
                raise KeyError(chrom)
        except KeyError:
            return
        self._logging_info("Calc {} mappable length...".format(chrom))
        KS = self._kshift
        d_out = self._device_out(ffi.PMX_NROWS * (KS + 1))
        self._ctx.mappable_len_dev(d_m, nbits, KS, self._kernel_flags, d_out)
        out = self._ctx.bits_download(d_out, (KS + 1) * 64)
        result.mappable_len = tuple(int(x) for x in out[:S + 1])

    # ---- lifecycle (mscc.pyx:173-179, :420-483) -----------------------------------------------------
    def flush(self, chrom: Optional[str] = None) -> None:
        if self._chr != "" and not self._buff_flashed:
            self._calc_correlation()
        if chrom is not None:
            self._fill_result(chrom)
        self._buff_flashed = True

    def finishup_calculation(self) -> None:
        self.flush(self._chr)
        for chrom in self.references:
            self._fill_result(chrom)

    def get_result(self, chrom: str) -> BothChromResult:
        if chrom not in self.ref2ncc_result and chrom not in self.ref2mscc_result:
            raise KeyError(chrom)
        return BothChromResult(chrom=self.ref2ncc_result.get(chrom), mappable_chrom=self.ref2mscc_result.get(chrom))

    def get_whole_result(self):
        if not self.ref2mscc_result:
            assert self.ref2ncc_result, "No results available for either NCC or MSCC."
            return NCCGenomeWideResult(
                genomelen=self.genomelen, forward_sum=self.forward_sum, reverse_sum=self.reverse_sum,
                chroms=self.ref2ncc_result.copy(), forward_read_len_sum=self.forward_read_len_sum,
                reverse_read_len_sum=self.reverse_read_len_sum)
        if not self.ref2ncc_result:
            return MSCCGenomeWideResult(
                genomelen=self.genomelen, chroms=self.ref2mscc_result.copy(),
                forward_read_len_sum=self.forward_read_len_sum, reverse_read_len_sum=self.reverse_read_len_sum)
        return BothGenomeWideResult(
            genomelen=self.genomelen, forward_sum=self.forward_sum, reverse_sum=self.reverse_sum,
            chroms=self.ref2ncc_result.copy(), mappable_chroms=self.ref2mscc_result.copy(),
            forward_read_len_sum=self.forward_read_len_sum, reverse_read_len_sum=self.reverse_read_len_sum)
