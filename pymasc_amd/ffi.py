"""ctypes binding of libpymasc_hip.so (include/pymasc_amd.h).

There is NO CPU fallback: if the library is missing it is built with hipcc, and if no MI355X is
visible ``Context()`` raises ``PmxError`` -- the product path fails loudly instead of computing on
the host.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

from . import build as _build

PMX_ROW_NCC_CCBINS = 0
PMX_ROW_MSCC_FSUM = 1
PMX_ROW_MSCC_RSUM = 2
PMX_ROW_MSCC_CCBINS = 3
PMX_ROW_MLEN = 4
PMX_ROW_SCALARS = 5
PMX_NROWS = 6

PMX_FLAG_SKIP_NCC = 1
PMX_FLAG_WINDOW_ONLY = 16
PMX_FLAG_DEEP_LISTS = 32
RANGE_TILE_BITS = 65536        # PMX_RANGE_TILE_BITS: the unit of pmx_cc_batch_ranges_dev
PMX_FLAG_EVENTS_HINT = 64     # the caller knows the data fits the event kernel's lists (no density probe, no synchronisation)
PMX_FLAG_FORCE_DENSE = 2
PMX_FLAG_FORCE_SPARSE = 4
PMX_FLAG_SKIP_MLEN = 8

# feed state words (include/pymasc_amd.h: PMX_FEED_*)
PMX_FEED_FORWARD_LEN_SUM, PMX_FEED_REVERSE_LEN_SUM, PMX_FEED_FORWARD_KEPT, PMX_FEED_REVERSE_KEPT = 0, 1, 2, 3
PMX_FEED_FIRST_UNSORTED, PMX_FEED_FIRST_OUT_OF_RANGE, PMX_FEED_LAST_POS, PMX_FEED_LAST_FORWARD_POS = 4, 5, 6, 7
PMX_FEED_READS, PMX_FEED_MAX_REVERSE_LEN, PMX_FEED_CHUNK_FORWARD_POS, PMX_FEED_WORDS = 8, 9, 10, 16
PMX_FEED_REGIONS_UNSORTED = 11   # first interval that breaks the order PMX_REGIONS_SORTED promises
PMX_FEED_ERR_BASE = 1 << 62
PMX_REGIONS_CLEAR = 1        # pmx_bits_set_regions_ex: clear the vector first
PMX_REGIONS_SIDE = 2         # ... on the context's side stream, beside the read feeders
PMX_REGIONS_SORTED = 4       # ... sorted, disjoint intervals: the vector is built (every word written once, no clear)
PMX_FEED_WHOLE_VECTORS = 1   # pmx_feed_reads_ex / pmx_feed_reads_delta16: the first run of a chromosome writes every word
PMX_PATH_DENSE = 1
PMX_PATH_SPARSE = 2

PMX_KERNEL_CC_DENSE = 0
PMX_KERNEL_CC_SPARSE = 1
PMX_KERNEL_AUTOCORR = 2
PMX_KERNEL_CC_EVENTS = 3
PMX_KERNEL_COUNT = 4

# every symbol include/pymasc_amd.h declares (tests check the library exports all of them)
EXPORTS = [
    "pmx_last_error", "pmx_version", "pmx_build_id", "pmx_device_count",
    "pmx_ctx_create", "pmx_ctx_destroy", "pmx_ctx_sync",
    "pmx_bits_alloc", "pmx_bits_free", "pmx_bits_clear", "pmx_bits_upload", "pmx_bits_download",
    "pmx_bits_set_positions", "pmx_bits_set_positions_dev", "pmx_bits_set_regions", "pmx_bits_set_regions_dev",
    "pmx_bits_count",
    "pmx_host_alloc", "pmx_host_free", "pmx_feed_reads", "pmx_feed_reads_ex", "pmx_feed_reads_delta16", "pmx_feed_reads_dev", "pmx_bits_set_regions_async", "pmx_bits_set_regions_ex", "pmx_bits_set_regions_dev_ex", "pmx_bits_build_batch",
    "pmx_bits_build_status",
    "pmx_cc_dev", "pmx_cc_batch_dev", "pmx_cc_batch_ranges_dev", "pmx_calc_correlation", "pmx_mappable_len_dev", "pmx_mappable_len",
    "pmx_mappable_len_batch_dev",
    "pmx_ctx_set_profiling", "pmx_ctx_reset_kernel_times", "pmx_ctx_kernel_time", "pmx_kernel_name",
    "pmx_debug_poison", "pmx_debug_read_slab", "pmx_debug_set_max_workgroups",
]


class PmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pymasc_amd HIP library error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the C-ABI library (building it in-tree first if needed) and declare prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PYMASC_AMD_LIB") or _build.LIB
    if path is None and not os.environ.get("PYMASC_AMD_LIB"):
        _build.build()
    L = ctypes.CDLL(p)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    L.pmx_last_error.restype = ctypes.c_char_p
    L.pmx_last_error.argtypes = []
    L.pmx_build_id.argtypes = []
    L.pmx_build_id.restype = ctypes.c_char_p
    L.pmx_version.restype = i32
    L.pmx_device_count.argtypes = [ctypes.POINTER(i32)]
    L.pmx_ctx_create.argtypes = [i32, vp, ctypes.POINTER(vp)]
    L.pmx_ctx_destroy.argtypes = [vp]
    L.pmx_ctx_sync.argtypes = [vp]
    L.pmx_bits_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
    L.pmx_bits_free.argtypes = [vp, vp]
    L.pmx_bits_clear.argtypes = [vp, vp, u64]
    L.pmx_bits_upload.argtypes = [vp, vp, vp, u64]
    L.pmx_bits_download.argtypes = [vp, vp, vp, u64]
    L.pmx_bits_set_positions.argtypes = [vp, vp, u64, vp, u64]
    L.pmx_bits_set_positions_dev.argtypes = [vp, vp, u64, vp, u64]
    L.pmx_bits_set_regions.argtypes = [vp, vp, u64, vp, vp, u64]
    L.pmx_bits_set_regions_dev.argtypes = [vp, vp, u64, vp, vp, u64]
    L.pmx_bits_count.argtypes = [vp, vp, u64, ctypes.POINTER(u64)]
    L.pmx_host_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
    L.pmx_host_free.argtypes = [vp, vp]
    L.pmx_feed_reads.argtypes = [vp, vp, vp, u64, vp, u32, vp, u32, vp, u64, u64, vp]
    L.pmx_feed_reads_delta16.argtypes = [vp, vp, vp, u64, vp, u64, vp, vp, u32, vp, u32, u64, vp, u32]
    L.pmx_feed_reads_ex.argtypes = [vp, vp, vp, u64, vp, u32, vp, u32, vp, u64, u64, vp, u32]
    L.pmx_feed_reads_dev.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, u64, vp, u32]
    L.pmx_bits_set_regions_async.argtypes = [vp, vp, u64, vp, vp, u32, u64, ctypes.c_int64, vp]
    L.pmx_bits_set_regions_ex.argtypes = [vp, vp, u64, vp, vp, u32, u64, ctypes.c_int64, vp, u32]
    L.pmx_bits_set_regions_dev_ex.argtypes = [vp, vp, u64, vp, vp, u64, ctypes.c_int64, vp, u32]
    L.pmx_bits_build_batch.argtypes = [vp, u32, vp, u32]
    L.pmx_bits_build_status.argtypes = [vp]
    L.pmx_mappable_len_batch_dev.argtypes = [vp, u32, vp, vp, u32, u32, vp]
    L.pmx_cc_dev.argtypes = [vp, vp, vp, vp, u64, u32, u32, u32, vp]
    L.pmx_cc_batch_dev.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32, u32, vp]
    L.pmx_cc_batch_ranges_dev.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, u32, vp]
    L.pmx_calc_correlation.argtypes = [vp, vp, vp, vp, u64, u32, u32, u32, vp]
    L.pmx_mappable_len_dev.argtypes = [vp, vp, u64, u32, u32, vp]
    L.pmx_mappable_len.argtypes = [vp, vp, u64, u32, u32, vp]
    L.pmx_ctx_set_profiling.argtypes = [vp, i32]
    L.pmx_ctx_reset_kernel_times.argtypes = [vp]
    L.pmx_ctx_kernel_time.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]
    L.pmx_kernel_name.restype = ctypes.c_char_p
    L.pmx_kernel_name.argtypes = [i32]
    L.pmx_debug_poison.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32]
    L.pmx_debug_read_slab.argtypes = [vp, u64, vp, u64]
    L.pmx_debug_set_max_workgroups.argtypes = [vp, ctypes.c_uint32]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("pmx_version",):
            fn.restype = i32
    if path is None:
        _lib = L
    return L


def pack_strand(pos: np.ndarray, is_reverse: np.ndarray) -> np.ndarray:
    """Positions (int32 / int64) with the strand in their top bit: the 4-bytes-per-read form pmx_feed_reads takes with
    h_is_reverse = NULL (a reader can emit it directly)."""
    pos = np.asarray(pos)
    dt = pos.dtype if pos.dtype in (np.dtype(np.int32), np.dtype(np.int64)) else np.dtype(np.int64)
    out = pos.astype(dt, copy=True)
    top = dt.type(-1) << dt.type(8 * dt.itemsize - 1)
    out[np.asarray(is_reverse, dtype=bool)] |= top
    return out


DELTA16_SEGMENT_READS = 1024     # include/pymasc_amd.h: pmx_feed_reads_delta16
DELTA16_MAX_GAP = 32766


class Delta16Reads:
    """A run of reads of one chromosome in file order, two bytes per read (pmx_feed_reads_delta16): `words` (uint16: strand in
    bit 15, distance to the previous read in bits 0..14), the segment table `seg_start` (uint32, nseg + 1 entries ending with n)
    / `seg_base` (int32: absolute position of each segment's first read), and the first / last position of the run -- what
    the host needs of it (order against the reads fed before; nothing else is looked at on the host)."""
    __slots__ = ("words", "seg_start", "seg_base", "first_pos", "last_pos", "readlen")

    def __init__(self, words, seg_start, seg_base, first_pos, last_pos, readlen=None):
        self.words, self.seg_start, self.seg_base, self.first_pos, self.last_pos = words, seg_start, seg_base, int(first_pos), int(last_pos)
        self.readlen = readlen      # (optional: per-read lengths laid out behind the table in the same page-locked block)

    @property
    def size(self):
        return int(self.words.size)


def pack_delta16(pos: np.ndarray, is_reverse: np.ndarray, context: "Context" = None, readlen: np.ndarray = None) -> Delta16Reads:
    """Sorted 1-based positions + strands -> the two-bytes-per-read form (a reader emits this directly; this is the numpy
    reference of the encoding).  With `context`, the three arrays are laid out in ONE page-locked block in the order the
    device slot takes them, so that the feed is a single copy."""
    pos = np.asarray(pos, dtype=np.int64)
    n = pos.size
    if n == 0:
        raise ValueError("pack_delta16: empty run")
    if (np.diff(pos) < 0).any():
        raise ValueError("pack_delta16: positions must be sorted (an unsorted run has no distance form; feed it as positions)")
    if pos[0] < 0 or pos[-1] >= 2**31:
        raise ValueError("pack_delta16: positions must fit 31 bits")
    d = np.diff(pos, prepend=pos[:1])
    starts = np.union1d(np.arange(0, n, DELTA16_SEGMENT_READS), np.flatnonzero(d > DELTA16_MAX_GAP))
    # (a stretch of far-apart reads can leave a segment longer than the grid allows only if a grid point was skipped: it is not)
    d[starts] = 0
    nseg = starts.size
    lens = None
    if context is not None and readlen is not None:     # per-read lengths (uint16) ride in the same block: still one copy
        words, seg_start, seg_base, lens = context.host_packed([n, nseg + 1, nseg, n], [np.uint16, np.uint32, np.int32, np.uint16])
        lens[:] = readlen
    elif context is not None:
        words, seg_start, seg_base = context.host_packed([n, nseg + 1, nseg], [np.uint16, np.uint32, np.int32])
    else:
        words, seg_start, seg_base = np.empty(n, np.uint16), np.empty(nseg + 1, np.uint32), np.empty(nseg, np.int32)
        lens = None if readlen is None else np.asarray(readlen).astype(np.uint16)
    words[:] = d.astype(np.uint16) | (np.asarray(is_reverse, dtype=bool).astype(np.uint16) << 15)
    seg_start[:nseg] = starts
    seg_start[nseg] = n
    seg_base[:] = pos[starts]
    return Delta16Reads(words, seg_start, seg_base, pos[0], pos[-1], lens)


def unpack_delta16(reads: Delta16Reads):
    """(positions int64, is_reverse bool): the inverse of pack_delta16 (tests, the CPU stand-in of the device)."""
    w = reads.words.astype(np.int64)
    d = w & 0x7fff
    nseg = reads.seg_base.size
    cum = np.cumsum(d)
    starts = reads.seg_start[:nseg].astype(np.int64)
    seg_of = np.repeat(np.arange(nseg), np.diff(reads.seg_start.astype(np.int64)))
    # position = the segment's base + the distances of the segment's reads up to and including this one (k_feed_expand16)
    pos = reads.seg_base.astype(np.int64)[seg_of] + cum - (cum[starts] - d[starts])[seg_of]
    return pos, (w >> 15).astype(bool)


def build_id() -> str:
    """Source hash compiled into the loaded library (include/pymasc_amd.h: pmx_build_id)."""
    return load_library().pmx_build_id().decode()


def _check(L, rc: int):
    if rc != 0:
        raise PmxError(rc, L.pmx_last_error().decode("utf-8", "replace"))


def nwords(nbits: int) -> int:
    return (int(nbits) + 63) // 64


def _np_ptr(a: Optional[np.ndarray], dtype) -> Optional[int]:
    if a is None:
        return None
    if a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
        raise TypeError(f"expected C-contiguous {np.dtype(dtype).name} array")
    return a.ctypes.data


class Context:
    """One per worker process / GPU. Thin object wrapper over the pmx_ctx handle."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._L = load_library()
        h = ctypes.c_void_p()
        _check(self._L, self._L.pmx_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None,
                                               ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    # -- lifecycle
    # -- a pool of device bit-vectors: calculators of successive samples on one context reuse each other's allocations
    # (hipMalloc / hipFree synchronise; 288 GB of HBM hold the vectors of many genomes)
    _POOL_GRANULE = 1 << 22          # bits

    def pool_alloc(self, nbits: int):
        """(device pointer, capacity in bits) of a bit-vector of >= nbits from the context's pool.  NOT cleared."""
        cap = -(-int(nbits) // self._POOL_GRANULE) * self._POOL_GRANULE
        free = self.__dict__.setdefault("_pool", {}).setdefault(cap, [])
        if free:
            return free.pop(), cap
        return self.bits_alloc(cap), cap

    def pool_free(self, d_words: int, cap: int):
        self.__dict__.setdefault("_pool", {}).setdefault(int(cap), []).append(d_words)

    def pool_release(self):
        for cap, ptrs in self.__dict__.get("_pool", {}).items():
            for p in ptrs:
                self.bits_free(p)
        self._pool = {}

    def close(self):
        if getattr(self, "_h", None):
            try:
                self.pool_release()
            except Exception:
                pass
            for p in list(getattr(self, "_pinned", {}).values()):
                self._L.pmx_host_free(self._h, ctypes.c_void_p(p))
            self._pinned = {}
            self._L.pmx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        _check(self._L, self._L.pmx_ctx_sync(self._h))

    # -- device bit-vectors (raw device pointers as ints)
    def bits_alloc(self, nbits: int) -> int:
        p = ctypes.c_void_p()
        _check(self._L, self._L.pmx_bits_alloc(self._h, int(nbits), ctypes.byref(p)))
        return p.value

    def bits_free(self, d_words: int):
        _check(self._L, self._L.pmx_bits_free(self._h, ctypes.c_void_p(d_words)))

    def bits_clear(self, d_words: int, nbits: int):
        _check(self._L, self._L.pmx_bits_clear(self._h, ctypes.c_void_p(d_words), int(nbits)))

    def bits_upload(self, d_words: int, h_words: np.ndarray, nbits: int):
        assert h_words.size >= nwords(nbits)
        _check(self._L, self._L.pmx_bits_upload(self._h, ctypes.c_void_p(d_words), _np_ptr(h_words, np.uint64),
                                                int(nbits)))

    def bits_download(self, d_words: int, nbits: int) -> np.ndarray:
        out = np.empty(nwords(nbits), dtype=np.uint64)
        _check(self._L, self._L.pmx_bits_download(self._h, ctypes.c_void_p(d_words), out.ctypes.data, int(nbits)))
        return out

    def bits_set_positions(self, d_words: int, nbits: int, pos: np.ndarray):
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        _check(self._L, self._L.pmx_bits_set_positions(self._h, ctypes.c_void_p(d_words), int(nbits),
                                                       pos.ctypes.data, pos.size))

    def bits_set_positions_dev(self, d_words: int, nbits: int, d_pos: int, n: int):
        _check(self._L, self._L.pmx_bits_set_positions_dev(self._h, ctypes.c_void_p(d_words), int(nbits),
                                                           ctypes.c_void_p(d_pos), int(n)))

    def bits_set_regions(self, d_words: int, nbits: int, first: np.ndarray, last: np.ndarray):
        first = np.ascontiguousarray(first, dtype=np.int64)
        last = np.ascontiguousarray(last, dtype=np.int64)
        assert first.size == last.size
        _check(self._L, self._L.pmx_bits_set_regions(self._h, ctypes.c_void_p(d_words), int(nbits),
                                                     first.ctypes.data, last.ctypes.data, first.size))

    def bits_set_regions_dev(self, d_words: int, nbits: int, d_first: int, d_last: int, n: int):
        _check(self._L, self._L.pmx_bits_set_regions_dev(self._h, ctypes.c_void_p(d_words), int(nbits),
                                                         ctypes.c_void_p(d_first), ctypes.c_void_p(d_last), int(n)))

    def bits_count(self, d_words: int, nbits: int) -> int:
        out = ctypes.c_uint64()
        _check(self._L, self._L.pmx_bits_count(self._h, ctypes.c_void_p(d_words), int(nbits), ctypes.byref(out)))
        return int(out.value)

    # -- stream-ordered feeding (nothing below synchronises)
    def host_array(self, n: int, dtype) -> np.ndarray:
        """A numpy array over page-locked host memory (pmx_host_alloc): copies from it run asynchronously.  The memory is
        released with the context (or host_free(array))."""
        dt = np.dtype(dtype)
        nbytes = max(int(n) * dt.itemsize, 1)
        p = ctypes.c_void_p()
        _check(self._L, self._L.pmx_host_alloc(self._h, nbytes, ctypes.byref(p)))
        buf = (ctypes.c_char * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(n))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_packed(self, sizes, dtype):
        """Arrays of `dtype` over ONE page-locked block, back to back with every array padded to 16 bytes: the layout the
        feeders give their staging slot, so feed_reads / bits_set_regions_async / bits_build_batch copy them in one piece
        (sizes e.g. (n_forward, n_reverse, n_intervals, n_intervals)).  Released with the context (or host_free(first array))."""
        dts = [np.dtype(d) for d in dtype] if isinstance(dtype, (list, tuple)) else [np.dtype(dtype)] * len(sizes)   # (one type, or one per array)
        offs, total = [], 0
        for n, dt in zip(sizes, dts):
            offs.append(total)
            total += (int(n) * dt.itemsize + 15) & ~15
        block = self.host_array(max(total, 16), np.uint8)
        return [block[o:o + int(n) * dt.itemsize].view(dt) for o, n, dt in zip(offs, sizes, dts)]

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            _check(self._L, self._L.pmx_host_free(self._h, ctypes.c_void_p(p)))

    @staticmethod
    def _int_array(a: np.ndarray, what: str) -> np.ndarray:
        a = np.asarray(a)
        ok = (np.dtype(np.int32), np.dtype(np.int64)) + ((np.dtype(np.uint16),) if what == "readlen" else ())
        if a.dtype not in ok:
            a = a.astype(np.int64)
        return np.ascontiguousarray(a)

    def feed_reads_delta16(self, d_F: int, d_R: int, nbits: int, reads: "Delta16Reads", readlen, reads_before: int, d_state: int,
                           whole_vectors: bool = False):
        """pmx_feed_reads_delta16: a run of reads in two bytes per read (pack_delta16).  whole_vectors: PMX_FEED_WHOLE_VECTORS
        (the first run of the chromosome; the vectors need not be cleared).  Returns the arrays handed over: keep them alive
        until the next synchronising call."""
        words = np.ascontiguousarray(reads.words, dtype=np.uint16)
        seg_start = np.ascontiguousarray(reads.seg_start, dtype=np.uint32)
        seg_base = np.ascontiguousarray(reads.seg_base, dtype=np.int32)
        if np.ndim(readlen) == 0:
            readlen = np.array([int(readlen)], dtype=np.int64)
            len_bytes = 0
        else:
            readlen = self._int_array(readlen, "readlen")
            len_bytes = readlen.dtype.itemsize
            assert readlen.size == words.size
        assert seg_start.size == seg_base.size + 1
        _check(self._L, self._L.pmx_feed_reads_delta16(self._h, ctypes.c_void_p(d_F), ctypes.c_void_p(d_R), int(nbits), words.ctypes.data,
                                                       int(words.size), seg_start.ctypes.data, seg_base.ctypes.data, int(seg_base.size),
                                                       readlen.ctypes.data, len_bytes, int(reads_before), ctypes.c_void_p(d_state),
                                                       PMX_FEED_WHOLE_VECTORS if whole_vectors else 0))
        return words, seg_start, seg_base, readlen

    def feed_reads_dev(self, d_F: int, d_R: int, nbits: int, d_pos: int, d_readlen: int, d_is_reverse: int, n: int,
                       reads_before: int, d_state: int, whole_vectors: bool = False):
        """pmx_feed_reads_dev: a run of reads that is already in device memory (int32 positions, int32 lengths, uint8 strand:
        what pymasc_amd.bam_device.DeviceBamReader leaves in HBM).  The arrays must stay untouched until the next synchronising call."""
        _check(self._L, self._L.pmx_feed_reads_dev(self._h, ctypes.c_void_p(d_F), ctypes.c_void_p(d_R), int(nbits), ctypes.c_void_p(d_pos),
                                                   ctypes.c_void_p(d_readlen), ctypes.c_void_p(d_is_reverse), int(n), int(reads_before),
                                                   ctypes.c_void_p(d_state), PMX_FEED_WHOLE_VECTORS if whole_vectors else 0))

    def feed_reads(self, d_F: int, d_R: int, nbits: int, pos: np.ndarray, readlen: np.ndarray, is_reverse: np.ndarray,
                   reads_before: int, d_state: int, whole_vectors: bool = False):
        """pmx_feed_reads: a run of reads of one chromosome in file order (int32 / int64 positions, uint16 / int32 / int64
        read lengths -- or ONE int for a run of reads of the same length --, strand as bool / uint8 -- or is_reverse = None with the strand packed into the top bit of
        every position (pack_strand) --; other integer types are converted).
        Returns the arrays actually handed over: keep them alive until the next synchronising call."""
        if is_reverse is None:                       # the strand travels in the top bit of the position word
            # ... so the word must keep its width: int32 / int64 as they are, the unsigned packings reinterpreted (a
            # conversion to int64 would turn bit 31 of a uint32 into part of the position), anything else refused
            pos = np.asarray(pos)
            if pos.dtype in (np.dtype(np.uint32), np.dtype(np.uint64)):
                pos = np.ascontiguousarray(pos).view(np.int32 if pos.dtype.itemsize == 4 else np.int64)
            elif pos.dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
                raise TypeError("feed_reads with the strand packed into the top bit takes 32- or 64-bit positions, not %s" % pos.dtype)
            pos = np.ascontiguousarray(pos)
            rev = None
        else:
            pos = self._int_array(pos, "pos")
            rev = np.ascontiguousarray(is_reverse)
            if rev.dtype != np.uint8:
                rev = rev.view(np.uint8) if rev.dtype == np.bool_ else rev.astype(np.uint8)
        if np.ndim(readlen) == 0:                    # one length for every read of the run: nothing to copy for it
            readlen = np.array([int(readlen)], dtype=np.int64)
            len_bytes = 0
        else:
            readlen = self._int_array(readlen, "readlen")
            len_bytes = readlen.dtype.itemsize
            assert pos.size == readlen.size
        assert rev is None or rev.size == pos.size
        _check(self._L, self._L.pmx_feed_reads_ex(self._h, ctypes.c_void_p(d_F), ctypes.c_void_p(d_R), int(nbits), pos.ctypes.data,
                                                  pos.dtype.itemsize, readlen.ctypes.data, len_bytes,
                                                  rev.ctypes.data if rev is not None else None,
                                                  pos.size, int(reads_before), ctypes.c_void_p(d_state),
                                                  PMX_FEED_WHOLE_VECTORS if whole_vectors else 0))
        return pos, readlen, rev

    def bits_set_regions_dev_ex(self, d_words: int, nbits: int, d_first: int, d_last: int, n: int, first_offset: int = 0,
                                d_state: Optional[int] = None, clear: bool = False, sorted_disjoint: bool = False):
        """pmx_bits_set_regions_dev_ex: set(first + first_offset, last) per interval for uint32 arrays that are already in device memory
        (pymasc_amd.bigwig_device.DeviceBigWigReader.fetch_device); clear / sorted_disjoint as in bits_set_regions_async."""
        flags = (PMX_REGIONS_CLEAR if clear else 0) | (PMX_REGIONS_SORTED if sorted_disjoint else 0)
        _check(self._L, self._L.pmx_bits_set_regions_dev_ex(self._h, ctypes.c_void_p(d_words), int(nbits), ctypes.c_void_p(d_first),
                                                            ctypes.c_void_p(d_last), int(n), int(first_offset),
                                                            ctypes.c_void_p(d_state) if d_state else None, flags))

    def bits_set_regions_async(self, d_words: int, nbits: int, first: np.ndarray, last: np.ndarray, first_offset: int = 0,
                               d_state: Optional[int] = None, clear: bool = False, side: bool = False, sorted_disjoint: bool = False):
        """set(first + first_offset, last) per interval, no synchronisation (uint32 or int64 arrays).  clear: the vector is
        cleared first; side: on the context's side stream, beside the read feeders; sorted_disjoint: the intervals are in
        BigWig order and the whole vector is BUILT from them (d_state required: a violation is recorded in
        d_state[PMX_FEED_REGIONS_UNSORTED]) -- pmx_bits_set_regions_ex, include/pymasc_amd.h."""
        first, last = np.asarray(first), np.asarray(last)
        if first.dtype != last.dtype or first.dtype not in (np.dtype(np.uint32), np.dtype(np.int64)):
            first, last = first.astype(np.int64), last.astype(np.int64)
        first, last = np.ascontiguousarray(first), np.ascontiguousarray(last)
        assert first.size == last.size
        flags = (PMX_REGIONS_CLEAR if clear else 0) | (PMX_REGIONS_SIDE if side else 0) | (PMX_REGIONS_SORTED if sorted_disjoint else 0)
        _check(self._L, self._L.pmx_bits_set_regions_ex(self._h, ctypes.c_void_p(d_words), int(nbits), first.ctypes.data,
                                                        last.ctypes.data, first.dtype.itemsize, first.size, int(first_offset),
                                                        ctypes.c_void_p(d_state) if d_state else None, flags))
        return first, last

    class _BuildJob(ctypes.Structure):
        _fields_ = [("d_F", ctypes.c_void_p), ("d_R", ctypes.c_void_p), ("d_M", ctypes.c_void_p), ("nbits", ctypes.c_uint64),
                    ("h_fpos", ctypes.c_void_p), ("h_rpos", ctypes.c_void_p), ("n_f", ctypes.c_uint64), ("n_r", ctypes.c_uint64),
                    ("h_first", ctypes.c_void_p), ("h_last", ctypes.c_void_p), ("n_iv", ctypes.c_uint64)]

    def bits_build_batch(self, jobs, pos_dtype=np.int64):
        """pmx_bits_build_batch.  jobs: iterable of (d_F, d_R, d_M, nbits, fpos, rpos, first, last) -- vectors as device
        pointers (or None), arrays of dtype `pos_dtype` (uint32 / int64) or None.  One call, no synchronisation; errors are
        reported by build_status()."""
        dt = np.dtype(pos_dtype)
        assert dt in (np.dtype(np.uint32), np.dtype(np.int64))
        jobs = list(jobs)
        arr = (self._BuildJob * max(len(jobs), 1))()
        keep = []
        for k, (d_F, d_R, d_M, nbits, fpos, rpos, first, last) in enumerate(jobs):
            def a(x):
                if x is None:
                    return None, 0
                x = np.ascontiguousarray(x, dtype=dt)
                keep.append(x)
                return x.ctypes.data, x.size
            j = arr[k]
            j.d_F, j.d_R, j.d_M, j.nbits = d_F, d_R, d_M, int(nbits)
            j.h_fpos, j.n_f = a(fpos)
            j.h_rpos, j.n_r = a(rpos)
            j.h_first, n1 = a(first)
            j.h_last, n2 = a(last)
            assert n1 == n2
            j.n_iv = n1
        _check(self._L, self._L.pmx_bits_build_batch(self._h, len(jobs), ctypes.cast(arr, ctypes.c_void_p), dt.itemsize))
        return keep

    def bits_build_status(self):
        _check(self._L, self._L.pmx_bits_build_status(self._h))

    def mappable_len_batch_dev(self, d_M, nbits, max_shift: int, flags: int, d_out):
        n = len(d_M)
        vpa = ctypes.c_void_p * n
        _check(self._L, self._L.pmx_mappable_len_batch_dev(self._h, n, ctypes.cast(vpa(*d_M), ctypes.c_void_p),
                                                           ctypes.cast((ctypes.c_uint64 * n)(*[int(x) for x in nbits]), ctypes.c_void_p),
                                                           int(max_shift), int(flags), ctypes.cast(vpa(*d_out), ctypes.c_void_p)))

    # -- hot path
    def cc_dev(self, d_F: int, d_R: int, d_M: Optional[int], nbits: int, max_shift: int, read_len: int,
               flags: int, d_out: int):
        _check(self._L, self._L.pmx_cc_dev(self._h, ctypes.c_void_p(d_F), ctypes.c_void_p(d_R),
                                           ctypes.c_void_p(d_M) if d_M else None, int(nbits), int(max_shift),
                                           int(read_len), int(flags), ctypes.c_void_p(d_out)))

    def cc_batch_dev(self, d_F, d_R, d_M, nbits, max_shift: int, read_len: int, flags: int, d_out):
        """One pass over a batch of chromosomes: sequences of device pointers / sizes (d_M None or full)."""
        n = len(d_F)
        assert len(d_R) == n and len(nbits) == n and len(d_out) == n and (d_M is None or len(d_M) == n)
        arr = lambda xs: (ctypes.c_uint64 * n)(*[int(x) for x in xs])
        aF, aR, aN, aO = arr(d_F), arr(d_R), arr(nbits), arr(d_out)
        aM = arr(d_M) if d_M is not None else None
        _check(self._L, self._L.pmx_cc_batch_dev(self._h, n, aF, aR, aM, aN, int(max_shift), int(read_len),
                                                 int(flags), aO))

    def cc_batch_ranges_dev(self, d_F, d_R, d_M, nbits, tile_first, tile_count, max_shift: int, read_len: int, flags: int, d_out):
        """pmx_cc_batch_ranges_dev: job i computes the share of its chromosome's sums that belongs to the RANGE_TILE_BITS-bit tiles
        [tile_first[i], tile_first[i] + tile_count[i]); the blocks of jobs that cover a chromosome add up to its result."""
        n = len(d_F)
        assert len(d_R) == n and len(nbits) == n and len(d_out) == n and (d_M is None or len(d_M) == n)
        assert len(tile_first) == n and len(tile_count) == n
        arr = lambda xs: (ctypes.c_uint64 * n)(*[int(x) for x in xs])
        arr32 = lambda xs: (ctypes.c_uint32 * n)(*[int(x) for x in xs])
        aF, aR, aN, aO = arr(d_F), arr(d_R), arr(nbits), arr(d_out)
        aM = arr(d_M) if d_M is not None else None
        _check(self._L, self._L.pmx_cc_batch_ranges_dev(self._h, n, aF, aR, aM, aN, arr32(tile_first), arr32(tile_count),
                                                        int(max_shift), int(read_len), int(flags), aO))

    def calc_correlation(self, F: np.ndarray, R: np.ndarray, M: Optional[np.ndarray], nbits: int, max_shift: int,
                         read_len: int, flags: int = 0) -> np.ndarray:
        """Host buffers in, result block [PMX_NROWS, max_shift+1] uint64 out."""
        nw = nwords(nbits)
        assert F.size >= nw and R.size >= nw and (M is None or M.size >= nw)
        out = np.zeros((PMX_NROWS, int(max_shift) + 1), dtype=np.uint64)
        _check(self._L, self._L.pmx_calc_correlation(self._h, _np_ptr(F, np.uint64), _np_ptr(R, np.uint64),
                                                     _np_ptr(M, np.uint64), int(nbits), int(max_shift),
                                                     int(read_len), int(flags), out.ctypes.data))
        return out

    def mappable_len_dev(self, d_M: int, nbits: int, max_shift: int, flags: int, d_out: int):
        _check(self._L, self._L.pmx_mappable_len_dev(self._h, ctypes.c_void_p(d_M), int(nbits), int(max_shift),
                                                     int(flags), ctypes.c_void_p(d_out)))

    def mappable_len(self, M: np.ndarray, nbits: int, max_shift: int, flags: int = 0) -> np.ndarray:
        assert M.size >= nwords(nbits)
        out = np.zeros(int(max_shift) + 1, dtype=np.uint64)
        _check(self._L, self._L.pmx_mappable_len(self._h, _np_ptr(M, np.uint64), int(nbits), int(max_shift),
                                                 int(flags), out.ctypes.data))
        return out

    # -- measurement
    def set_profiling(self, on):
        """False / 0: off; True / 1: kernels that do work; 2: also the fallback launches behind the event kernel."""
        _check(self._L, self._L.pmx_ctx_set_profiling(self._h, int(on)))

    def reset_kernel_times(self):
        _check(self._L, self._L.pmx_ctx_reset_kernel_times(self._h))

    def kernel_time(self, kernel_id: int):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        _check(self._L, self._L.pmx_ctx_kernel_time(self._h, int(kernel_id), ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def debug_poison(self, pattern: int, mask: int = 511) -> None:
        """Diagnostic: fill the context's scratch buffers (mask bits 0-7) and the LDS of every CU (bit 8) with a byte /
        dword pattern, so that a kernel reading memory it has not written fails deterministically (tests, fuzz)."""
        _check(self._L, self._L.pmx_debug_poison(self._h, int(pattern) & 0xffffffff, int(mask)))

    def debug_set_max_workgroups(self, n: int) -> None:
        """Diagnostic: cap the persistent workgroups of later launches (0: off), see include/pymasc_amd.h."""
        _check(self._L, self._L.pmx_debug_set_max_workgroups(self._h, int(n)))

    def kernel_name(self, kernel_id: int) -> str:
        return self._L.pmx_kernel_name(int(kernel_id)).decode()


def device_count() -> int:
    L = load_library()
    n = ctypes.c_int()
    _check(L, L.pmx_device_count(ctypes.byref(n)))
    return int(n.value)
