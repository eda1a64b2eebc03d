"""BAM ingest ON THE DEVICE: binding of libpymasc_ingest.so (include/pymasc_amd_ingest.h).

``DeviceBamReader`` has the surface of ``pymasc_amd.bam.BamReader`` (itself the part of the reference's
BAMFileProcessor the calculation touches, PyMaSC/reader/bam.py:84-165) -- ``references``, ``lengths``, ``batches``,
``fetch``, ``close`` -- so ``pymasc_amd.bam.feed_bam`` takes either; the difference is where the work happens: the whole
file is copied to HBM compressed, and BGZF inflate, CRC32, the record chain and the reference's read filter
(handler/read.py:62-155) run as HIP kernels.  There is no host fallback: without a GPU ``DeviceBamReader`` raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterator, Tuple

import numpy as np

from .bam import PMX_BAM_DEFAULT_EXCLUDE, PmxIOError

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libpymasc_ingest.so"

#: every symbol include/pymasc_amd_ingest.h declares (tests/test_abi.py checks the built library against this list)
INGEST_EXPORTS = [
    "pmx_dbam_last_error", "pmx_dbam_version", "pmx_dbam_open", "pmx_dbam_close", "pmx_dbam_nref", "pmx_dbam_ref_name",
    "pmx_dbam_ref_len", "pmx_dbam_header_text", "pmx_dbam_decode", "pmx_dbam_device_arrays", "pmx_dbam_fetch",
    "pmx_dbam_runs", "pmx_dbam_counters", "pmx_dbam_timings", "pmx_dbam_inflated",
    "pmx_dbw_open", "pmx_dbw_close", "pmx_dbw_nchrom", "pmx_dbw_chrom_name", "pmx_dbw_chrom_len", "pmx_dbw_fetch", "pmx_dbw_device_arrays",
    "pmx_dbw_sorted", "pmx_dbw_copy",
]

_lib = None


def ingest_library_path() -> str:
    return os.environ.get("PYMASC_AMD_INGEST_LIB", os.path.join(_HERE, _LIB_NAME))


def load_ingest_library():
    """dlopen libpymasc_ingest.so (built by pymasc_amd/build.py:build_ingest) and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = ingest_library_path()
    if not os.path.exists(path):
        raise PmxIOError(-1, "{} not found: run `python pymasc_amd/build.py`".format(path))
    L = ctypes.CDLL(path)
    vp, i32, i64, u32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
    L.pmx_dbam_last_error.restype = ctypes.c_char_p
    L.pmx_dbam_version.restype = ctypes.c_int
    L.pmx_dbam_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    L.pmx_dbam_open.restype = ctypes.c_int
    L.pmx_dbam_close.argtypes = [vp]
    L.pmx_dbam_close.restype = None
    L.pmx_dbam_nref.argtypes = [vp]
    L.pmx_dbam_nref.restype = i32
    L.pmx_dbam_ref_name.argtypes = [vp, i32]
    L.pmx_dbam_ref_name.restype = ctypes.c_char_p
    L.pmx_dbam_ref_len.argtypes = [vp, i32]
    L.pmx_dbam_ref_len.restype = i64
    L.pmx_dbam_header_text.argtypes = [vp, ctypes.POINTER(u32)]
    L.pmx_dbam_header_text.restype = ctypes.c_char_p
    L.pmx_dbam_decode.argtypes = [vp, u32, u32, i32]
    L.pmx_dbam_decode.restype = i64
    L.pmx_dbam_device_arrays.argtypes = [vp] + [ctypes.POINTER(vp)] * 4
    L.pmx_dbam_device_arrays.restype = ctypes.c_int
    L.pmx_dbam_fetch.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.pmx_dbam_fetch.restype = ctypes.c_int
    L.pmx_dbam_runs.argtypes = [vp, i64, vp, vp, vp, vp]
    L.pmx_dbam_runs.restype = i64
    L.pmx_dbam_counters.argtypes = [vp] + [ctypes.POINTER(u64)] * 6
    L.pmx_dbam_counters.restype = ctypes.c_int
    L.pmx_dbam_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.pmx_dbam_timings.restype = ctypes.c_int
    L.pmx_dbam_inflated.argtypes = [vp, u64, u64, vp]
    L.pmx_dbam_inflated.restype = ctypes.c_int
    L.pmx_dbw_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    L.pmx_dbw_open.restype = ctypes.c_int
    L.pmx_dbw_close.argtypes = [vp]
    L.pmx_dbw_close.restype = None
    L.pmx_dbw_nchrom.argtypes = [vp]
    L.pmx_dbw_nchrom.restype = i32
    L.pmx_dbw_chrom_name.argtypes = [vp, i32]
    L.pmx_dbw_chrom_name.restype = ctypes.c_char_p
    L.pmx_dbw_chrom_len.argtypes = [vp, i32]
    L.pmx_dbw_chrom_len.restype = i64
    L.pmx_dbw_fetch.argtypes = [vp, ctypes.c_char_p, ctypes.c_float]
    L.pmx_dbw_fetch.restype = i64
    L.pmx_dbw_device_arrays.argtypes = [vp] + [ctypes.POINTER(vp)] * 3
    L.pmx_dbw_device_arrays.restype = ctypes.c_int
    L.pmx_dbw_sorted.argtypes = [vp]
    L.pmx_dbw_sorted.restype = ctypes.c_int
    L.pmx_dbw_copy.argtypes = [vp, i64, i64, vp, vp, vp]
    L.pmx_dbw_copy.restype = ctypes.c_int
    _lib = L
    return L


def _raise(code: int):
    raise PmxIOError(int(code), load_ingest_library().pmx_dbam_last_error().decode("utf-8", "replace"))


class DeviceBamReader:
    """A BAM file inflated and decoded on the GPU; batches of filtered read arrays like ``BamReader``."""

    def __init__(self, path, device: int = 0, threads: int = 0):
        self._L = load_ingest_library()
        self.path = os.fspath(path)
        self._h = None
        h = ctypes.c_void_p()
        rc = self._L.pmx_dbam_open(self.path.encode(), int(device), int(threads), ctypes.byref(h))
        if rc:
            _raise(rc)
        self._h = h
        n = self._L.pmx_dbam_nref(h)
        self.references: Tuple[str, ...] = tuple(self._L.pmx_dbam_ref_name(h, i).decode() for i in range(n))
        self.lengths: Tuple[int, ...] = tuple(int(self._L.pmx_dbam_ref_len(h, i)) for i in range(n))

    def has_index(self) -> bool:
        """Every reference can be fetched on its own (the whole stream is resident): no .bai needed."""
        return True

    @property
    def closed(self) -> bool:
        return self._h is None

    @property
    def header_text(self) -> str:
        ln = ctypes.c_uint32()
        t = self._L.pmx_dbam_header_text(self._h, ctypes.byref(ln))
        return (t or b"").decode("utf-8", "replace")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            self._L.pmx_dbam_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counters(self) -> dict:
        v = [ctypes.c_uint64() for _ in range(6)]
        rc = self._L.pmx_dbam_counters(self._h, *[ctypes.byref(x) for x in v])
        if rc:
            _raise(rc)
        return dict(zip(("records", "kept", "bytes_out", "bytes_in", "members", "rewalked"), (int(x.value) for x in v)))

    def timings(self) -> dict:
        t = (ctypes.c_double * 6)()
        rc = self._L.pmx_dbam_timings(self._h, t)
        if rc:
            _raise(rc)
        return dict(zip(("upload_s", "inflate_s", "crc_s", "header_s", "chain_s", "write_s"), (float(x) for x in t)))

    def inflated(self, first: int = 0, n: int = None) -> bytes:
        """(test hook) bytes of the inflated stream."""
        if n is None:
            n = self.counters()["bytes_out"] - first
        buf = np.empty(max(n, 1), dtype=np.uint8)
        rc = self._L.pmx_dbam_inflated(self._h, int(first), int(n), buf.ctypes.data)
        if rc:
            _raise(rc)
        return buf[:n].tobytes()

    def decode(self, mapq_criteria: int = 0, flag_exclude: int = PMX_BAM_DEFAULT_EXCLUDE, reference: int = -1) -> int:
        """Runs the record walk + filter on the device; returns the number of kept records (they stay in HBM)."""
        if self._h is None:
            raise ValueError("I/O operation on closed BAM reader")
        n = self._L.pmx_dbam_decode(self._h, int(mapq_criteria), int(flag_exclude), int(reference))
        if n < 0:
            _raise(n)
        return int(n)

    def device_arrays(self) -> Tuple[int, int, int, int]:
        """Device addresses of (ref_id int32, pos1 int32, read_len int32, reverse uint8) of the last decode."""
        v = [ctypes.c_void_p() for _ in range(4)]
        rc = self._L.pmx_dbam_device_arrays(self._h, *[ctypes.byref(x) for x in v])
        if rc:
            _raise(rc)
        return tuple(int(x.value or 0) for x in v)

    def device_runs(self):
        """The kept records of the last decode as runs of one reference, in file order: a list of
        (ref_id, start index, count, first pos, last pos) -- None when the file has more than 65536 runs (unsorted)."""
        n = self._L.pmx_dbam_runs(self._h, 0, None, None, None, None)
        if n == -3:
            return None
        if n < 0:
            _raise(n)
        start = np.empty(max(n, 1), dtype=np.int64)
        ref = np.empty(max(n, 1), dtype=np.int32)
        first = np.empty(max(n, 1), dtype=np.int32)
        last = np.empty(max(n, 1), dtype=np.int32)
        m = self._L.pmx_dbam_runs(self._h, n, start.ctypes.data, ref.ctypes.data, first.ctypes.data, last.ctypes.data)
        if m < 0:
            _raise(m)
        total = self.counters()["kept"]
        ends = list(start[1:m]) + [total]
        return [(int(ref[r]), int(start[r]), int(ends[r] - start[r]), int(first[r]), int(last[r])) for r in range(m)]

    def feed(self, calculator, mapq_criteria: int, references=None, finish: bool = True) -> int:
        """handler/calc.py:131-161 with nothing on the host: decode + filter on the device, then every run of one chromosome is
        handed to ``calculator.feed_reads_device`` as three device addresses (the same duplicate / order rules, mscc.pyx:351-418,
        applied by the device feeders).  Falls back to ``pymasc_amd.bam.feed_bam`` (host arrays) for an unsorted file or a
        calculator without the device entry point.  Returns the number of reads fed."""
        from .bam import feed_bam
        wanted = set(calculator.references if references is None else references)
        if not hasattr(calculator, "feed_reads_device"):
            return feed_bam(calculator, self, mapq_criteria, references, finish, use_index=False)
        self.decode(mapq_criteria)
        runs = self.device_runs()
        if runs is None:
            return feed_bam(calculator, self, mapq_criteria, references, finish, use_index=False)
        d_ref, d_pos, d_len, d_rev = self.device_arrays()
        fed = 0
        for ref, start, count, first, last in runs:
            name = self.references[ref]
            if name not in wanted:
                continue
            calculator.feed_reads_device(name, d_pos + 4 * start, d_len + 4 * start, d_rev + start, count, first, last)
            fed += count
        if finish:
            calculator.finishup_calculation()
        else:
            calculator._ctx.sync()      # the feeders read this reader's arrays: they must be done before it may be closed
        return fed

    def _fetch(self, first: int, n: int):
        ref = np.empty(n, dtype=np.int32)
        pos = np.empty(n, dtype=np.int32)
        rlen = np.empty(n, dtype=np.int32)
        rev = np.empty(n, dtype=np.uint8)
        rc = self._L.pmx_dbam_fetch(self._h, first, n, ref.ctypes.data, pos.ctypes.data, rlen.ctypes.data, rev.ctypes.data)
        if rc:
            _raise(rc)
        return ref, pos, rlen, rev.astype(bool)

    def batches(self, mapq_criteria: int = 0, flag_exclude: int = PMX_BAM_DEFAULT_EXCLUDE, batch: int = 1 << 22,
                _reference: int = -1) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        """Yields (ref_id, pos_1based, read_len, is_reverse) of the reads that pass the reference's filter
        (handler/read.py:62-90,131-141), in file order, at most ``batch`` per round -- as ``BamReader.batches``."""
        total = self.decode(mapq_criteria, flag_exclude, _reference)
        for first in range(0, total, batch):
            yield self._fetch(first, min(batch, total - first))

    def fetch(self, reference: str, mapq_criteria: int = 0, flag_exclude: int = PMX_BAM_DEFAULT_EXCLUDE,
              batch: int = 1 << 22):
        """The reads of ONE reference (handler/worker.py:106-132: what a worker gets from ``AlignmentFile.fetch(chrom)``)."""
        if reference not in self.references:
            raise KeyError(reference)
        return self.batches(mapq_criteria, flag_exclude, batch, _reference=self.references.index(reference))
