"""CCHipCalculator -- drop-in for the reference's ``CCBitArrayCalculator`` on one MI355X.

Same constructor and the same six-method calculator protocol
(PyMaSC/interfaces/calculator.py:15-79,110-124; PyMaSC/core/bitarray/mscc.pyx:97-98,173-179,370-483),
same exceptions (ReadUnsortedError / KeyError) and value-equal result objects, so
``PyMaSC/handler/factory.py:224-258`` can construct it instead of the Cython class.

What is different underneath (MI355X-first, not a translation):
* nothing waits for the GPU until results are asked for.  Reads fed one by one are appended to host arrays; reads fed in
  bulk (``feed_reads``) go straight to the device, where ``pmx_feed_reads`` applies the reference's per-read rules --
  sortedness, the forward / reverse duplicate rules, read-length sums (mscc.pyx:362-366,388-392,416-418) -- and sets the
  bits; mappability intervals go through ``pmx_bits_set_regions_async``.  The host never walks the reads;
* every chromosome keeps HBM bit-vectors of its own (a pool on the context; 288 GB hold many genomes) and ``flush`` only
  queues it: the (max_shift+1)-iteration loop of full-vector passes (mscc.pyx:288-317) of ALL chromosomes queued since
  the last fetch is ONE batched call, ``pmx_cc_batch_dev``, whose kernels read each vector from HBM once; the lag pass
  of the read-less references (mscc.pyx:207-215, one per reference in ``finishup_calculation``) is one batched call too;
* ``get_result`` / ``finishup_calculation`` / ``get_whole_result`` (and the public result attributes) run what is
  queued, synchronise ONCE, copy the result arena back in one piece and build the reference's result objects.  Errors
  the device found inside a bulk chunk (``ReadUnsortedError``, ``IndexError``) are raised then; the per-read methods
  raise them at once, like the reference.

There is no CPU compute fallback: constructing the calculator without a visible GPU raises.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import ffi
from .exceptions import ReadUnsortedError
from .result import (BothChromResult, BothGenomeWideResult, EmptyMSCCResult, EmptyNCCResult, MSCCGenomeWideResult,
                     MSCCResult, NCCGenomeWideResult, NCCResult, calc_cc_batch)

logger = logging.getLogger(__name__)

_CHUNK = 1 << 16
# The event kernel's list capacities per 64-Kbit tile (csrc/kernels_events.h: EV_CAPF, EV_CAPR, EV_CAPE_SMALL).  max_shift <=
# 1023: forward reads + reverse reads + run edges share one pool (EVENT_POOL), and run edges alone may fill EVENT_EDGES of it; above:
# fixed shares.  A chromosome whose AVERAGE tile is beyond them goes to the window kernels unseen (PMX_FLAG_WINDOW_ONLY).
EVENT_TILE_BITS = 65536
EVENT_FORWARD, EVENT_REVERSE, EVENT_EDGES, EVENT_EDGES_BIG = 768, 1000, 1536, 384
EVENT_POOL, EVENT_POOL_NCC = 2416, 768 + 1000 + 1536      # EV_POOL_SMALL (with a track) / without
EVENT_POOL_DEEP = 4328                                    # EV_POOL_DEEP: the instantiation PMX_FLAG_DEEP_LISTS selects


def window_only_hint(n_forward, n_reverse, n_runs, length, max_shift):
    """True when the event kernel would find (nearly) every tile of this chromosome dense: the caller then skips it.
    n_forward / n_reverse: reads kept per strand, n_runs: mappable runs (two edges each; 0 without a track)."""
    per_tile = EVENT_TILE_BITS / float(max(length, 1))
    f, r, e = n_forward * per_tile, n_reverse * per_tile, 2.0 * n_runs * per_tile
    if max_shift <= 1023:
        return e > EVENT_EDGES or f + r + e > (EVENT_POOL_DEEP if n_runs else EVENT_POOL_NCC)
    return f > EVENT_FORWARD or r > EVENT_REVERSE or e > EVENT_EDGES_BIG


def deep_lists_hint(n_forward, n_reverse, n_runs, length, max_shift):
    """True when the average tile comes close to the ordinary list pool (max_shift <= 1023, with a track) but stays within the
    larger one: the caller then asks for the event kernel's DEEP instantiation (PMX_FLAG_DEEP_LISTS)."""
    if max_shift > 1023 or not n_runs or window_only_hint(n_forward, n_reverse, n_runs, length, max_shift):
        return False
    per_tile = EVENT_TILE_BITS / float(max(length, 1))
    return (n_forward + n_reverse + 2.0 * n_runs) * per_tile > 0.85 * EVENT_POOL


class _ReadBuffer:
    """Append-only (position, readlen, strand) arrays in file order for the per-read protocol methods; amortised O(1) per
    read, no Python list of ints.  Handed to the device in one piece when the chromosome ends."""

    __slots__ = ("pos", "rlen", "rev", "n")

    def __init__(self):
        self.pos = np.empty(_CHUNK, dtype=np.int64)
        self.rlen = np.empty(_CHUNK, dtype=np.int64)
        self.rev = np.empty(_CHUNK, dtype=np.uint8)
        self.n = 0

    def append(self, pos: int, rlen: int, rev: int):
        if self.n == self.pos.size:
            self.pos = np.concatenate([self.pos, np.empty(self.pos.size, dtype=np.int64)])
            self.rlen = np.concatenate([self.rlen, np.empty(self.rlen.size, dtype=np.int64)])
            self.rev = np.concatenate([self.rev, np.empty(self.rev.size, dtype=np.uint8)])
        self.pos[self.n] = pos
        self.rlen[self.n] = rlen
        self.rev[self.n] = rev
        self.n += 1

    def take(self):
        """The buffered reads as arrays of their own (the buffer is reused while the copy to the device may still run)."""
        n, self.n = self.n, 0
        return self.pos[:n].copy(), self.rlen[:n].copy(), self.rev[:n].copy()


class _Pending:
    """One chromosome whose kernels are queued: where its rows / feed state will be, and what to build from them."""

    __slots__ = ("chrom", "slot", "kind", "has_m", "known", "glen", "nreads", "vecs", "nbits", "flags", "hint", "launched")

    def __init__(self, chrom, slot, kind, has_m=False, known=None, glen=0, nreads=0, vecs=(), nbits=0, flags=0, hint=0):
        self.chrom, self.slot, self.kind, self.has_m, self.known, self.glen, self.nreads = (
            chrom, slot, kind, has_m, known, glen, nreads)
        self.vecs, self.nbits, self.flags, self.hint = vecs, nbits, flags, hint     # [(device pointer, pool capacity)]: F, R[, M]
        self.launched = False                                       # its kernels are in the stream already (_run_cc)


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
        self._progress = progress_bar       # accepted for the reference's constructor signature; not driven
        self._kernel_flags = int(kernel_flags)
        # the result block of the C ABI keeps four scalars in a row of max_shift + 1 words, so the kernels are run
        # with at least 3 shifts; shifts are independent, rows are cut back to max_shift + 1 below
        self._kshift = max(self.max_shift, 3)
        # lag tables known beforehand (the *_mappability.json cache, pymasc_amd/mappability.py): chromosomes
        # found here skip the autocorrelation pass
        self._known_mlen: Dict[str, Sequence[int]] = dict(chrom2mappable_len or {})

        self._ncc: Dict[str, NCCResult] = {}
        self._mscc: Dict[str, MSCCResult] = {}
        self._forward_sum = self._reverse_sum = 0
        self._forward_read_len_sum = self._reverse_read_len_sum = 0

        self._chr = ""
        self._solved_chr: List[str] = []
        self._buff_flashed = False
        self._array_extend_size = self.read_len + self.max_shift + self.EXTRA_ALLOCATE_SIZE   # mscc.pyx:134
        self._last_pos = 0
        self._cur_nbits = 0
        self._buf = _ReadBuffer()
        self._fed = 0                 # reads of the current chromosome handed to the device so far
        self._cur_slot = -1
        self._cur_vecs: List[Tuple[int, int]] = []

        # the one and only compute back-end; raises ffi.PmxError when no GPU / no library
        self._ctx = context if context is not None else ffi.Context(device)
        self._own_ctx = context is None
        # Every chromosome gets bit-vectors of its own from the context's pool and keeps them until results are fetched:
        # the cross-correlation of ALL chromosomes queued by then is ONE batched pass of the kernels (pmx_cc_batch_dev),
        # not a launch chain per chromosome.  288 GB of HBM hold hundreds of genomes' worth of vectors; `max_resident_bytes`
        # bounds what one calculator keeps before it runs the batch early.
        self.max_resident_bytes = 64 << 30
        # chromosomes queued before their kernels are launched ahead of the fetch (0: never).  Off by default: on one GPU the
        # feeders' kernels and copies keep the stream busy while a genome arrives, and an early pass only moves time from
        # the fetch into the feed (hg38: feed 3.7 -> 4.15 ms, fetch 1.75 -> 1.2 ms).  For callers whose reads arrive slowly.
        self.early_batch = 0
        # how a chromosome's mappability vector is made (A/B knobs; same bits either way): on the context's side stream, beside
        # the next chromosome's reads; built whole from intervals in BigWig order instead of cleared and OR-ed into
        self.track_side_stream = True
        self.track_builder = True
        self._resident_bytes = 0
        # result arena: one slot per reference = a result block + a feed state, filled by the queued kernels and read
        # back in ONE copy when results are asked for
        self._slot_words = ffi.PMX_NROWS * (self._kshift + 1) + ffi.PMX_FEED_WORDS
        self._arena = 0
        self._arena_slots = 0
        self._free_slots: List[int] = []
        self._pending: List[_Pending] = []
        self._readless: List[Tuple[str, int, Tuple[int, int]]] = []     # (chrom, slot, M vector) awaiting the batched lag pass
        self._inflight: List[Any] = []                      # host arrays whose copies may still be running

    # ---- the reference's public attributes, current as soon as they are read -------------------------------
    @property
    def ref2ncc_result(self) -> Dict[str, NCCResult]:
        self._materialize()
        return self._ncc

    @property
    def ref2mscc_result(self) -> Dict[str, MSCCResult]:
        self._materialize()
        return self._mscc

    @property
    def forward_sum(self) -> int:
        self._materialize()
        return self._forward_sum

    @property
    def reverse_sum(self) -> int:
        self._materialize()
        return self._reverse_sum

    @property
    def forward_read_len_sum(self) -> int:
        self._materialize()
        return self._forward_read_len_sum

    @property
    def reverse_read_len_sum(self) -> int:
        self._materialize()
        return self._reverse_read_len_sum

    # ---- plumbing ------------------------------------------------------------------------------
    def close(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is None:
            return
        try:
            ctx.sync()
        except Exception:
            pass
        for p in self._pending:
            for ptr, cap in p.vecs:
                ctx.pool_free(ptr, cap)
        self._pending = []
        for v in getattr(self, "_cur_vecs", []):
            ctx.pool_free(*v)
        self._cur_vecs = []
        for _c, _s, (d_m, cap) in self._readless:
            ctx.pool_free(d_m, cap)
        self._readless = []
        if self._arena:
            ctx.bits_free(self._arena)
            self._arena = 0
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

    def _vector(self, nbits: int, clear: bool = True):
        """(pointer, pool capacity) of a zeroed HBM bit-vector of >= nbits (the clear is queued on the stream; clear=False: the
        caller's next kernel writes every word of it)."""
        ptr, cap = self._ctx.pool_alloc(nbits)
        if clear:
            self._ctx.bits_clear(ptr, nbits)
        self._resident_bytes += cap // 8
        return ptr, cap

    def _new_slot(self) -> int:
        """A cleared arena slot (result block + feed state)."""
        if not self._arena:
            self._arena_slots = len(self.references) + 2
            self._arena = self._ctx.bits_alloc(self._arena_slots * self._slot_words * 64)
            self._free_slots = list(range(self._arena_slots - 1, -1, -1))
        if not self._free_slots:
            self._materialize()                     # (gives the slots of everything fetched back)
        return self._free_slots.pop()            # (cleared: a fresh arena is zero-filled, _materialize clears what it fetched)

    def _slot_ptr(self, slot: int) -> int:
        return self._arena + slot * self._slot_words * 8

    def _state_ptr(self, slot: int) -> int:
        return self._slot_ptr(slot) + ffi.PMX_NROWS * (self._kshift + 1) * 8

    # ---- feeding (mscc.pyx:351-418) --------------------------------------------------------------
    def _init_buff(self):
        self._last_pos = 0
        self._buf.n = 0
        self._fed = 0

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
            self._cur_slot = -1             # its vectors are cleared and its slot taken when the first reads go to the device
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
        self._buf.append(pos, readlen, 0)

    def feed_reverse_read(self, chrom: str, pos: int, readlen: int) -> None:
        """1-based leftmost position of a reverse read; its bit is pos + readlen - 1 (mscc.pyx:397-418)."""
        self._check_pos(chrom, pos, pos + readlen - 1, "reverse")
        self._buf.append(pos, readlen, 1)

    def _start_chromosome_on_device(self, clear: bool = True) -> bool:
        """Slot + F / R vectors of the current chromosome; True if they were taken just now.  clear=False: the vectors are
        handed out as the pool left them, for a first run that writes every word (PMX_FEED_WHOLE_VECTORS)."""
        if self._cur_slot < 0:
            self._cur_slot = self._new_slot()
            self._cur_vecs = [self._vector(self._cur_nbits, clear), self._vector(self._cur_nbits, clear)]
            self._d_f, self._d_r = self._cur_vecs[0][0], self._cur_vecs[1][0]
            return True
        return False

    def _to_device(self, pos: np.ndarray, readlen: np.ndarray, rev: np.ndarray):
        """Queues a run of reads of the current chromosome: copy + the reference's duplicate rules + bit set on the device
        (pmx_feed_reads).  Nothing is read back here."""
        if pos.size == 0:
            return
        # the first run of a chromosome writes every word of its vectors itself (no clear, no atomics: k_feed_build)
        fresh = self._start_chromosome_on_device(clear=False)
        self._inflight.append(self._ctx.feed_reads(self._d_f, self._d_r, self._cur_nbits, pos, readlen, rev, self._fed,
                                                   self._state_ptr(self._cur_slot), whole_vectors=fresh))
        self._fed += int(pos.size)

    def feed_reads(self, chrom: str, pos: np.ndarray, readlen: np.ndarray, is_reverse: Optional[np.ndarray]) -> None:
        """Bulk variant of the two methods above for one chromosome: arrays in file order (int32 or int64 positions -- or an
        ffi.Delta16Reads, the two-bytes-per-read form of a sorted run, with is_reverse = None --,
        uint16 / int32 / int64 read lengths or ONE int when every read of the chunk has that length, strand as bool / uint8, or
        None with the strand packed into the top bit of every position: ffi.pack_strand).  Not part of the reference protocol; it removes the per-read Python call for
        vectorised readers, and the host does not walk the reads at all: the order of the chunk against the reads fed
        before is checked here (first position), everything else -- order inside the chunk, range, duplicates, read-length
        sums -- by the device, so ReadUnsortedError / IndexError for a read inside the chunk are raised when the results
        of the chromosome are fetched (flush + get_result, finishup_calculation), not by this call."""
        if isinstance(pos, ffi.Delta16Reads):
            # two bytes per read (ffi.pack_delta16 / a reader that emits the form): the host sees the first and the last
            # position of the run, the device everything else
            if pos.size == 0:
                return
            self._check_pos(chrom, pos.first_pos)
            if self._buf.n:
                self._to_device(*self._buf.take())
            fresh = self._start_chromosome_on_device(clear=False)
            self._inflight.append(self._ctx.feed_reads_delta16(self._d_f, self._d_r, self._cur_nbits, pos,
                                                               readlen if np.ndim(readlen) == 0 else np.asarray(readlen),
                                                               self._fed, self._state_ptr(self._cur_slot), whole_vectors=fresh))
            self._fed += pos.size
            self._last_pos = max(self._last_pos, pos.last_pos)
            return
        pos = np.asarray(pos)
        if pos.size == 0:
            return
        packed = is_reverse is None
        top = (1 << (8 * pos.dtype.itemsize - 1)) - 1 if packed else -1      # (mask of the position bits)
        self._check_pos(chrom, int(pos[0]) & top if packed else int(pos[0]))
        if self._buf.n:                              # reads fed one by one before this chunk go first
            self._to_device(*self._buf.take())
        self._to_device(pos, readlen if np.ndim(readlen) == 0 else np.asarray(readlen), None if packed else np.asarray(is_reverse))
        last = int(pos[-1]) & top if packed else int(pos[-1])
        self._last_pos = max(self._last_pos, last)

    def feed_reads_device(self, chrom: str, d_pos: int, d_readlen: int, d_is_reverse: int, n: int, first_pos: int, last_pos: int) -> None:
        """feed_reads for a run of one chromosome that lies in DEVICE memory (addresses of int32 positions, int32 query lengths,
        uint8 strands; `first_pos` / `last_pos`: the positions of its first and last read, for the order check against what was
        fed before -- everything inside the run is checked by the device, as in feed_reads).  The producer is
        pymasc_amd.bam_device.DeviceBamReader: a BAM file inflated, decoded and filtered on the GPU never visits the host.
        The arrays must stay as they are until the results of the chromosome have been fetched."""
        if n == 0:
            return
        self._check_pos(chrom, int(first_pos))
        if self._buf.n:
            self._to_device(*self._buf.take())
        fresh = self._start_chromosome_on_device(clear=False)
        self._ctx.feed_reads_dev(self._d_f, self._d_r, self._cur_nbits, d_pos, d_readlen, d_is_reverse, n, self._fed,
                                 self._state_ptr(self._cur_slot), whole_vectors=fresh)
        self._fed += int(n)
        self._last_pos = max(self._last_pos, int(last_pos))

    # ---- per-chromosome calculation ---------------------------------------------------------------
    def _load_mappability(self, chrom: str, nbits: int, slot: int):
        """mscc.pyx:327-349 -> (device vector, pool capacity), queued, not waited for; None without a feeder; KeyError if
        the track is missing."""
        if not self._bwfeeder:
            return None
        on_device = getattr(self._bwfeeder, "fetch_device", None)
        if on_device is not None and hasattr(self._ctx, "bits_set_regions_dev_ex"):
            # pymasc_amd.bigwig_device.DeviceBigWigReader: the file was decoded on the GPU and the chromosome's intervals are in HBM
            # already -- set(begin + 1, end) (mscc.pyx:343-344) straight from there; sorted, disjoint intervals (BigWig order, checked
            # by the reader on the device) BUILD the vector as below
            d_first, d_last, n, in_order = on_device(self.MAPPABILITY_THRESHOLD, chrom)
            self._logging_info("Loading {} mappability to bit array...".format(chrom))
            vec = self._vector(nbits, clear=False)
            self._ctx.bits_set_regions_dev_ex(vec[0], nbits, d_first, d_last, n, 1, self._state_ptr(slot), clear=True,
                                              sorted_disjoint=self.track_builder and in_order and n > 0)
            self._n_runs = int(n)
            return vec
        bulk = getattr(self._bwfeeder, "fetch_arrays", None)
        if bulk is not None:          # pymasc_amd.bigwig.BigWigReader: arrays, no per-interval Python objects
            first, last, _v = bulk(self.MAPPABILITY_THRESHOLD, chrom)
        else:
            iv = [(b, e) for b, e, _v in self._bwfeeder.fetch(self.MAPPABILITY_THRESHOLD, chrom)]
            arr = np.asarray(iv, dtype=np.int64).reshape(-1, 2)
            first, last = arr[:, 0].copy(), arr[:, 1].copy()
        self._logging_info("Loading {} mappability to bit array...".format(chrom))
        vec = self._vector(nbits, clear=False)
        # set(begin + 1, end), mscc.pyx:343-344 (the + 1 is applied on the device; an end beyond the vector is clipped).  The
        # vector is cleared and filled on the context's side stream: this chromosome's track is built while the next one's
        # reads are fed (one queue of small kernels paced the feed of a genome); the batched pass waits for it.
        # BigWig intervals come sorted and disjoint (begin_i < end_i <= begin_(i+1)): the vector is then BUILT, every word
        # written once by the workgroup that owns it, instead of cleared and OR-ed into (include/pymasc_amd.h:
        # PMX_REGIONS_SORTED).  Checked here -- two comparisons per interval, a few microseconds per chromosome --, because
        # the reference takes intervals in any order (set(begin + 1, end) per interval, mscc.pyx:343-344): anything else goes
        # the general way.  (The device checks again and records a violation in the chromosome's feed state.)
        first, last = np.asarray(first), np.asarray(last)
        in_order = self.track_builder and bool((first < last).all()) and (first.size < 2 or bool((first[1:] >= last[:-1]).all()))
        self._inflight.append(self._ctx.bits_set_regions_async(vec[0], nbits, first, last, 1, self._state_ptr(slot), clear=True,
                                                               side=self.track_side_stream, sorted_disjoint=in_order))
        self._n_runs = int(first.size)   # two run edges per interval: the other density the event kernel's lists depend on
        return vec

    def _calc_correlation(self):
        chrom = self._chr
        S, L = self.max_shift, self.read_len
        glen = self.ref2genomelen[chrom]
        nbits = glen + self._array_extend_size
        if self._buf.n:
            self._to_device(*self._buf.take())
        self._start_chromosome_on_device()          # (a chromosome whose reads were all refused still gets its rows)
        slot = self._cur_slot

        try:
            m_vec = self._load_mappability(chrom, nbits, slot)
        except KeyError as e:
            self._logging_info("Mappability for '{}' not found. "
                               "Skip calc mappability sensitive CC.".format(e.args[0] if e.args else chrom))
            m_vec = None
        d_m = m_vec[0] if m_vec is not None else None

        self._logging_info("Calculate cross-correlation for {}...".format(chrom))
        flags = self._kernel_flags | (ffi.PMX_FLAG_SKIP_NCC if self.skip_ncc else 0)
        # deep data or a track of very short runs: the average tile overflows the event kernel's lists -- say so instead
        # of letting it find out (same integers either way).  The strand split is not known on the host (the device walks
        # the reads): half the reads fed stands for a strand.
        counts = (0.5 * self._fed, 0.5 * self._fed, getattr(self, "_n_runs", 0) if d_m is not None else 0, glen, self.max_shift)
        # (kept apart from the flags: one batched pass serves chromosomes whose hints differ -- chrM beside the autosomes --,
        # with the hint most of its positions ask for, see _run_cc; a hint is always given, so that the library does not
        # take one itself from a sample of the vectors: that costs a synchronisation of the stream, pymasc_amd.h)
        if window_only_hint(*counts):
            hint = ffi.PMX_FLAG_WINDOW_ONLY
        elif deep_lists_hint(*counts):
            hint = ffi.PMX_FLAG_DEEP_LISTS       # deep, but within the event kernel's larger list pool
        else:
            hint = ffi.PMX_FLAG_EVENTS_HINT
        c = L - 1
        known = self._known_mlen.get(chrom) if d_m is not None else None
        if known is not None and len(known) <= max(c, S - c):      # cache too short for this run: recompute
            known = None
        if known is not None:
            flags |= ffi.PMX_FLAG_SKIP_MLEN
        # queued for the batched pass (_run_cc): the vectors stay resident until then
        vecs = self._cur_vecs + ([m_vec] if m_vec is not None else [])
        self._cur_vecs = []
        self._pending.append(_Pending(chrom, slot, "cc", has_m=d_m is not None, known=known, glen=glen, nreads=self._fed,
                                      vecs=vecs, nbits=nbits, flags=flags, hint=hint))
        self._cur_slot = -1
        # every `early_batch` chromosomes the kernels of those queued so far are launched (nothing is waited for or read
        # back), so that the fetch at the end finds only the last few to do
        if self.early_batch and sum(1 for p in self._pending if p.kind == "cc" and not p.launched) >= self.early_batch:
            self._run_cc()
        if self._resident_bytes > self.max_resident_bytes:
            self._materialize()

    def _fill_result(self, chrom: str):
        """Placeholders / read-less mappable_len (mscc.pyx:181-215).  The lag pass of read-less chromosomes is queued for ONE
        batched launch (finishup_calculation walks every reference: 86 in the reference's test BAM, mscc.pyx:436-439)."""
        self._chr = chrom
        S, L = self.max_shift, self.read_len
        glen = self.ref2genomelen[chrom]
        queued = {p.chrom for p in self._pending}
        if chrom not in self._ncc and not (chrom in queued and not self.skip_ncc):
            self._ncc[chrom] = EmptyNCCResult.create_empty(glen, S, L)
        has_mscc = chrom in self._mscc or any(p.chrom == chrom and (p.has_m or p.kind == "mlen") for p in self._pending)
        if not self._bwfeeder or has_mscc:
            return
        result = self._mscc[chrom] = EmptyMSCCResult.create_empty(glen, S, L)
        nbits = glen + self._array_extend_size
        known = self._known_mlen.get(chrom)
        if known is not None and len(known) > S:      # cached table long enough: no track load, no kernel
            result.mappable_len = tuple(int(x) for x in known[:S + 1])
            return
        slot = self._new_slot()
        try:
            m_vec = self._load_mappability(chrom, nbits, slot)
            if m_vec is None:
                raise KeyError(chrom)
        except KeyError:
            self._free_slots.append(slot)       # (nothing was queued: the slot is as clean as it came)
            return
        self._logging_info("Calc {} mappable length...".format(chrom))
        self._readless.append((chrom, slot, m_vec))

    def _run_cc(self):
        """_calc_correlation's loop (mscc.pyx:288-317) for every chromosome queued since the last fetch: one batched pass per
        group of chromosomes that share the kernel flags -- with / without a track, with / without a cached lag table:
        normally one group; the density hints of the chromosomes do NOT split a pass --."""
        groups: Dict[Tuple[int, bool], List[_Pending]] = {}
        for p in self._pending:
            if p.kind == "cc" and not p.launched:
                groups.setdefault((p.flags, p.has_m), []).append(p)
                p.launched = True
        for (flags, has_m), ps in groups.items():
            # the density hint of the pass: what most of its positions ask for (same integers whatever it says)
            total = float(sum(p.nbits for p in ps)) or 1.0
            w_window = sum(p.nbits for p in ps if p.hint == ffi.PMX_FLAG_WINDOW_ONLY) / total
            w_deep = sum(p.nbits for p in ps if p.hint == ffi.PMX_FLAG_DEEP_LISTS) / total
            flags |= (ffi.PMX_FLAG_WINDOW_ONLY if w_window > 0.5 else
                      ffi.PMX_FLAG_DEEP_LISTS if w_window + w_deep > 0.5 else ffi.PMX_FLAG_EVENTS_HINT)
            self._ctx.cc_batch_dev([p.vecs[0][0] for p in ps], [p.vecs[1][0] for p in ps],
                                   [p.vecs[2][0] for p in ps] if has_m else None, [p.nbits for p in ps], self._kshift,
                                   self.read_len, flags, [self._slot_ptr(p.slot) for p in ps])

    def _run_readless(self):
        if not self._readless:
            return
        nb = [self.ref2genomelen[c] + self._array_extend_size for c, _s, _m in self._readless]
        self._ctx.mappable_len_batch_dev([m[0] for _c, _s, m in self._readless], nb, self._kshift, self._kernel_flags,
                                         [self._slot_ptr(s) for _c, s, _m in self._readless])
        for chrom, slot, m_vec in self._readless:
            self._pending.append(_Pending(chrom, slot, "mlen", glen=self.ref2genomelen[chrom], vecs=[m_vec]))

    def _materialize(self):
        """Everything queued -> result objects: ONE synchronisation and ONE copy of the arena, then the deferred errors of
        the chunks fed in bulk, then the reference's result classes."""
        if not self._pending and not self._readless:
            return
        self._run_cc()
        self._run_readless()
        S, L, KS = self.max_shift, self.read_len, self._kshift
        pending, self._pending = self._pending, []
        readless, self._readless = self._readless, []
        hi = max(p.slot for p in pending) + 1
        raw = self._ctx.bits_download(self._arena, hi * self._slot_words * 64)     # synchronises
        self._free_slots.extend(p.slot for p in pending)
        # cleared before they are handed out again -- ONLY the slots fetched: a chromosome that is in the middle of its bulk
        # feed may hold a recycled slot below `hi`, and its feed state (read-length sums, last forward position, look-back
        # bound, first-error words) lives there.  Runs of neighbouring slots share a clear (normally one for everything).
        fetched = sorted(p.slot for p in pending)
        run0 = prev = fetched[0]
        for s in fetched[1:] + [None]:
            if s is not None and s == prev + 1:
                prev = s
                continue
            self._ctx.bits_clear(self._slot_ptr(run0), (prev - run0 + 1) * self._slot_words * 64)
            run0 = prev = s
        self._inflight = []
        for p in pending:                            # (the synchronising copy above is behind every kernel that read them)
            for ptr, cap in p.vecs:
                self._ctx.pool_free(ptr, cap)
                self._resident_bytes -= cap // 8
            p.vecs = ()
        raw = raw.reshape(-1, self._slot_words)
        nrow = ffi.PMX_NROWS * (KS + 1)
        # (all values lie far below 2^63: the same words seen as int64, no copy; the feed states as Python ints in one go)
        raw_i64 = raw.view(np.int64)
        states = raw[:, nrow:nrow + ffi.PMX_FEED_WORDS].tolist()
        c = L - 1
        error: Optional[BaseException] = None
        new_ncc: List[NCCResult] = []
        new_mscc: List[MSCCResult] = []
        ncc_slots: List[int] = []
        mscc_slots: List[int] = []
        known_rows: List[Tuple[int, np.ndarray]] = []      # (index into new_mscc, cached mappable length by shift)
        for p in pending:
            out = raw[p.slot, :nrow].reshape(ffi.PMX_NROWS, KS + 1)
            st = states[p.slot]
            if st[ffi.PMX_FEED_REGIONS_UNSORTED] and error is None:      # (cannot happen: _load_mappability checks the order)
                error = RuntimeError("mappability interval {} of {} is out of order".format(
                    ffi.PMX_FEED_ERR_BASE - st[ffi.PMX_FEED_REGIONS_UNSORTED], p.chrom))
            if p.kind == "mlen":
                self._mscc[p.chrom].mappable_len = tuple(out.reshape(-1)[:S + 1].tolist())
                continue
            # what the device found in the chunks fed in bulk: raised once everything fetched has been stored
            if st[ffi.PMX_FEED_FIRST_UNSORTED] and error is None:
                error = ReadUnsortedError("read {} of {} is below its predecessor".format(
                    ffi.PMX_FEED_ERR_BASE - st[ffi.PMX_FEED_FIRST_UNSORTED], p.chrom))
            if st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE] and error is None:
                error = IndexError("read {} of {} beyond its bit array ({} bits)".format(
                    ffi.PMX_FEED_ERR_BASE - st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE], p.chrom, p.glen + self._array_extend_size))
            # the per-shift rows go into the result objects as int64 ARRAYS (the reference's models take lists or arrays,
            # PyMaSC/result.py:34,77,95-101, and its own placeholders hold arrays, :207-208): one copy of the block per
            # chromosome instead of a Python int per shift -- 0.6 ms per hg38 genome at 1000 shifts, 80 ms for config 5
            # (one copy of the chromosome's block: the result objects must not keep the whole fetched arena alive)
            i64 = raw_i64[p.slot, :nrow].reshape(ffi.PMX_NROWS, KS + 1).copy()
            f_rls, r_rls = st[ffi.PMX_FEED_FORWARD_LEN_SUM], st[ffi.PMX_FEED_REVERSE_LEN_SUM]
            self._forward_read_len_sum += f_rls
            self._reverse_read_len_sum += r_rls
            if not self.skip_ncc:
                fsum, rsum = i64[ffi.PMX_ROW_SCALARS, :2].tolist()
                self._forward_sum += fsum
                self._reverse_sum += rsum
                res = self._ncc[p.chrom] = NCCResult(
                    max_shift=S, read_len=L, genomelen=p.glen, forward_sum=fsum, reverse_sum=rsum,
                    forward_read_len_sum=f_rls, reverse_read_len_sum=r_rls,
                    ccbins=i64[ffi.PMX_ROW_NCC_CCBINS, :S + 1])
                new_ncc.append(res)
                ncc_slots.append(p.slot)
            if p.has_m:
                by_shift = out[ffi.PMX_ROW_MLEN].tolist() if p.known is None else [int(p.known[abs(c - d)]) for d in range(S + 1)]
                if p.known is not None:
                    known_rows.append((len(mscc_slots), np.array(by_shift[:S + 1], dtype=np.int64)))
                mscc_slots.append(p.slot)
                # the reference stores mappable_len by LAG: d < L -> index L-1-d, L <= d < 2L-1 skipped
                # (same value by symmetry), d >= 2L-1 appended (mscc.pyx:271,292-298)
                head = by_shift[:min(L, S + 1)][::-1]
                mlen: List[Optional[int]] = [None] * (L - len(head)) + head
                if S >= 2 * L - 1:
                    mlen += by_shift[2 * L - 1:S + 1]
                mres = self._mscc[p.chrom] = MSCCResult(
                    max_shift=S, read_len=L, genomelen=p.glen,
                    forward_sum=i64[ffi.PMX_ROW_MSCC_FSUM, :S + 1],
                    reverse_sum=i64[ffi.PMX_ROW_MSCC_RSUM, :S + 1],
                    forward_read_len_sum=f_rls, reverse_read_len_sum=r_rls,
                    ccbins=i64[ffi.PMX_ROW_MSCC_CCBINS, :S + 1], mappable_len=mlen)
                new_mscc.append(mres)
        # NCCResult.calc_cc / MSCCResult.calc_cc (mscc.pyx:320-323), all chromosomes at once, from the rows as fetched
        # (the integer rows of all chromosomes in one gather per row kind from the fetched arena)
        blocks = raw_i64[:, :nrow].reshape(-1, ffi.PMX_NROWS, KS + 1)
        ncc_bins = blocks[ncc_slots, ffi.PMX_ROW_NCC_CCBINS, :S + 1] if ncc_slots else None
        mscc_rows = None
        if mscc_slots:
            mscc_rows = tuple(blocks[mscc_slots, row, :S + 1] for row in
                              (ffi.PMX_ROW_MSCC_CCBINS, ffi.PMX_ROW_MSCC_FSUM, ffi.PMX_ROW_MSCC_RSUM, ffi.PMX_ROW_MLEN))
            for k, row in known_rows:
                mscc_rows[3][k] = row
        calc_cc_batch(new_ncc, new_mscc, ncc_bins, mscc_rows)
        if error is not None:
            raise error

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
        self._materialize()

    def get_result(self, chrom: str) -> BothChromResult:
        self._materialize()
        if chrom not in self._ncc and chrom not in self._mscc:
            raise KeyError(chrom)
        return BothChromResult(chrom=self._ncc.get(chrom), mappable_chrom=self._mscc.get(chrom))

    def get_whole_result(self):
        self._materialize()
        if not self._mscc:
            assert self._ncc, "No results available for either NCC or MSCC."
            return NCCGenomeWideResult(
                genomelen=self.genomelen, forward_sum=self._forward_sum, reverse_sum=self._reverse_sum,
                chroms=self._ncc.copy(), forward_read_len_sum=self._forward_read_len_sum,
                reverse_read_len_sum=self._reverse_read_len_sum)
        if not self._ncc:
            return MSCCGenomeWideResult(
                genomelen=self.genomelen, chroms=self._mscc.copy(),
                forward_read_len_sum=self._forward_read_len_sum, reverse_read_len_sum=self._reverse_read_len_sum)
        return BothGenomeWideResult(
            genomelen=self.genomelen, forward_sum=self._forward_sum, reverse_sum=self._reverse_sum,
            chroms=self._ncc.copy(), mappable_chroms=self._mscc.copy(),
            forward_read_len_sum=self._forward_read_len_sum, reverse_read_len_sum=self._reverse_read_len_sum)
