"""BAM ingest for the calculator: native reader binding + the feeding loop.

Host mirror of what the reference does between the file and the calculator:

* ``BamReader`` has the surface of the reference's BAMFileProcessor that the calculation touches
  (PyMaSC/reader/bam.py:84-165: ``references``, ``lengths``, ``close``, context manager) over
  libpymasc_io.so's BGZF/BAM reader (include/pymasc_amd_io.h) instead of pysam.
* ``feed_bam`` is the single-process loop of PyMaSC/handler/calc.py:131-161 with the read filter and field
  extraction of handler/read.py:62-155 done natively and in bulk: records arrive as arrays in file order, are cut
  into runs of one chromosome and handed to ``CCHipCalculator.feed_reads`` (which applies the sortedness and
  duplicate rules of mscc.pyx:351-418), then ``finishup_calculation``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterator, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libpymasc_io.so"

PMX_BAM_FLAG_UNMAPPED = 0x4
PMX_BAM_FLAG_REVERSE = 0x10
PMX_BAM_FLAG_READ2 = 0x80
PMX_BAM_FLAG_DUPLICATE = 0x400
PMX_BAM_DEFAULT_EXCLUDE = PMX_BAM_FLAG_READ2 | PMX_BAM_FLAG_UNMAPPED | PMX_BAM_FLAG_DUPLICATE

#: every symbol include/pymasc_amd_io.h declares (tests/test_abi.py checks the built library against this list)
IO_EXPORTS = [
    "pmx_io_last_error", "pmx_io_version",
    "pmx_bam_open", "pmx_bam_close", "pmx_bam_nref", "pmx_bam_ref_name", "pmx_bam_ref_len", "pmx_bam_header_text",
    "pmx_bam_next_batch", "pmx_bam_counters", "pmx_bam_index_load", "pmx_bam_has_index", "pmx_bam_fetch_ref",
    "pmx_bigwig_open", "pmx_bigwig_close", "pmx_bigwig_nchrom", "pmx_bigwig_chrom_name", "pmx_bigwig_chrom_len",
    "pmx_bigwig_fetch",
]


class PmxIOError(IOError):
    """An error reported by libpymasc_io.so; ``code`` is the PMX_IO_ERR_* value."""

    def __init__(self, code: int, msg: str):
        super().__init__("[pmx_io {}] {}".format(code, msg))
        self.code = code


_lib = None


def io_library_path() -> str:
    return os.environ.get("PYMASC_AMD_IO_LIB", os.path.join(_HERE, _LIB_NAME))


def load_io_library():
    """dlopen libpymasc_io.so (built by pymasc_amd/build.py:build_io) and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = io_library_path()
    if not os.path.exists(path):
        raise PmxIOError(-1, "{} not found: run `python pymasc_amd/build.py`".format(path))
    L = ctypes.CDLL(path)
    vp, i32, i64, u32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
    L.pmx_io_last_error.restype = ctypes.c_char_p
    L.pmx_io_version.restype = ctypes.c_int
    L.pmx_bam_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp)]
    L.pmx_bam_open.restype = ctypes.c_int
    L.pmx_bam_close.argtypes = [vp]
    L.pmx_bam_close.restype = None
    L.pmx_bam_nref.argtypes = [vp]
    L.pmx_bam_nref.restype = i32
    L.pmx_bam_ref_name.argtypes = [vp, i32]
    L.pmx_bam_ref_name.restype = ctypes.c_char_p
    L.pmx_bam_ref_len.argtypes = [vp, i32]
    L.pmx_bam_ref_len.restype = i64
    L.pmx_bam_header_text.argtypes = [vp, ctypes.POINTER(u32)]
    L.pmx_bam_header_text.restype = ctypes.c_char_p
    L.pmx_bam_next_batch.argtypes = [vp, u32, u32, i64, vp, vp, vp, vp]
    L.pmx_bam_next_batch.restype = i64
    L.pmx_bam_counters.argtypes = [vp] + [ctypes.POINTER(u64)] * 4
    L.pmx_bam_counters.restype = ctypes.c_int
    L.pmx_bam_index_load.argtypes = [vp, ctypes.c_char_p]
    L.pmx_bam_index_load.restype = ctypes.c_int
    L.pmx_bam_has_index.argtypes = [vp]
    L.pmx_bam_has_index.restype = ctypes.c_int
    L.pmx_bam_fetch_ref.argtypes = [vp, i32]
    L.pmx_bam_fetch_ref.restype = ctypes.c_int
    L.pmx_bigwig_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.pmx_bigwig_open.restype = ctypes.c_int
    L.pmx_bigwig_close.argtypes = [vp]
    L.pmx_bigwig_close.restype = None
    L.pmx_bigwig_nchrom.argtypes = [vp]
    L.pmx_bigwig_nchrom.restype = i32
    L.pmx_bigwig_chrom_name.argtypes = [vp, i32]
    L.pmx_bigwig_chrom_name.restype = ctypes.c_char_p
    L.pmx_bigwig_chrom_len.argtypes = [vp, i32]
    L.pmx_bigwig_chrom_len.restype = i64
    L.pmx_bigwig_fetch.argtypes = [vp, ctypes.c_char_p, ctypes.c_float, i64, vp, vp, vp]
    L.pmx_bigwig_fetch.restype = i64
    _lib = L
    return L


def _raise(code: int):
    raise PmxIOError(int(code), load_io_library().pmx_io_last_error().decode("utf-8", "replace"))


class BamReader:
    """A coordinate-sorted BAM file as batches of filtered read arrays."""

    def __init__(self, path, threads: int = 0, index=None):
        """``index``: path of the .bai; None: ``<path>.bai`` or ``<stem>.bai`` when present (like pysam); False: none."""
        self._L = load_io_library()
        self.path = os.fspath(path)
        h = ctypes.c_void_p()
        rc = self._L.pmx_bam_open(self.path.encode(), int(threads), ctypes.byref(h))
        if rc:
            _raise(rc)
        self._h = h
        n = self._L.pmx_bam_nref(h)
        self.references: Tuple[str, ...] = tuple(self._L.pmx_bam_ref_name(h, i).decode() for i in range(n))
        self.lengths: Tuple[int, ...] = tuple(int(self._L.pmx_bam_ref_len(h, i)) for i in range(n))
        if index is None:
            for cand in (self.path + ".bai", os.path.splitext(self.path)[0] + ".bai"):
                if os.path.exists(cand):
                    index = cand
                    break
        if index:
            rc = self._L.pmx_bam_index_load(h, os.fspath(index).encode())
            if rc:
                self.close()
                _raise(rc)

    def has_index(self) -> bool:
        """reader/bam.py:128-135."""
        return bool(self._L.pmx_bam_has_index(self._h))

    @property
    def closed(self) -> bool:
        return self._h is None

    @property
    def header_text(self) -> str:
        ln = ctypes.c_uint32()
        t = self._L.pmx_bam_header_text(self._h, ctypes.byref(ln))
        return (t or b"").decode("utf-8", "replace")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            self._L.pmx_bam_close(self._h)
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
        v = [ctypes.c_uint64() for _ in range(4)]
        rc = self._L.pmx_bam_counters(self._h, *[ctypes.byref(x) for x in v])
        if rc:
            _raise(rc)
        return dict(zip(("records", "kept", "bytes_out", "bytes_in"), (int(x.value) for x in v)))

    def fetch(self, reference: str, mapq_criteria: int = 0, flag_exclude: int = PMX_BAM_DEFAULT_EXCLUDE,
              batch: int = 1 << 22) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        """The reads of ONE reference through the .bai index, as ``batches`` yields them -- what a worker of the
        reference's multi-process mode gets from ``AlignmentFile.fetch(chrom)`` (handler/worker.py:106-132)."""
        if self._h is None:
            raise ValueError("I/O operation on closed BAM reader")
        if reference not in self.references:
            raise KeyError(reference)
        if not self.has_index():
            raise ValueError("fetch() needs an index: {}.bai not found".format(self.path))
        rc = self._L.pmx_bam_fetch_ref(self._h, self.references.index(reference))
        if rc:
            _raise(rc)
        return self.batches(mapq_criteria, flag_exclude, batch, _region=True)

    def batches(self, mapq_criteria: int = 0, flag_exclude: int = PMX_BAM_DEFAULT_EXCLUDE, batch: int = 1 << 22,
                _region: bool = False) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        """Yields (ref_id, pos_1based, read_len, is_reverse) of the reads that pass the reference's filter
        (handler/read.py:62-90,131-141), in file order, at most ``batch`` per round."""
        if self._h is None:
            raise ValueError("I/O operation on closed BAM reader")
        if not _region:      # a plain pass always starts at the first record, whatever was fetched before
            rc = self._L.pmx_bam_fetch_ref(self._h, -1)
            if rc:
                _raise(rc)
        ref = np.empty(batch, dtype=np.int32)
        pos = np.empty(batch, dtype=np.int32)
        rlen = np.empty(batch, dtype=np.int32)
        rev = np.empty(batch, dtype=np.uint8)
        while True:
            n = self._L.pmx_bam_next_batch(self._h, int(mapq_criteria), int(flag_exclude), batch, ref.ctypes.data,
                                           pos.ctypes.data, rlen.ctypes.data, rev.ctypes.data)
            if n < 0:
                _raise(n)
            if n == 0:
                return
            yield ref[:n].copy(), pos[:n].copy(), rlen[:n].copy(), rev[:n].astype(bool)


def feed_bam(calculator, reader: BamReader, mapq_criteria: int, references: Optional[Sequence[str]] = None,
             finish: bool = True, use_index: Optional[bool] = None) -> int:
    """Stream every usable read of ``reader`` into ``calculator`` (handler/calc.py:131-161).

    ``references``: the chromosomes taken into account (config.references, calc.py:143-144); default: the
    calculator's.  ``use_index``: read only those chromosomes through the .bai (default: when an index is loaded
    and fewer than all references are wanted -- a rank of a multi-GPU run); otherwise one pass over the file.
    Returns the number of reads fed.  Raises what the calculator raises (ReadUnsortedError for unsorted input,
    mscc.pyx:351-364)."""
    names = reader.references
    wanted = set(calculator.references if references is None else references)
    use = np.array([n in wanted for n in names], dtype=bool)
    if use_index is None:
        use_index = reader.has_index() and not use.all()
    fed = 0

    def feed(ref, pos, rlen, rev):
        cuts = np.flatnonzero(np.diff(ref)) + 1           # runs of one chromosome, in file order
        starts = np.concatenate(([0], cuts))
        ends = np.concatenate((cuts, [ref.size]))
        for s, e in zip(starts.tolist(), ends.tolist()):
            calculator.feed_reads(names[int(ref[s])], pos[s:e], rlen[s:e], rev[s:e])
        return int(ref.size)

    if use_index:
        for name in (n for n in names if n in wanted):    # file order = header order for a sorted BAM
            for ref, pos, rlen, rev in reader.fetch(name, mapq_criteria):
                if ref.size:
                    fed += feed(ref, pos, rlen, rev)
    else:
        for ref, pos, rlen, rev in reader.batches(mapq_criteria):
            if not use.all():
                m = use[ref]
                ref, pos, rlen, rev = ref[m], pos[m], rlen[m], rev[m]
            if ref.size:
                fed += feed(ref, pos, rlen, rev)
    if finish:
        calculator.finishup_calculation()
    return fed
