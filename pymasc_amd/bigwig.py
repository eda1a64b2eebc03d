"""BigWig mappability track reader over libpymasc_io.so.

Same surface as the reference's BigWigReader (PyMaSC/reader/bigwig.pyx:100-200): ``chromsizes``,
``fetch(valfilter, chrom)`` -> iterator of ``(begin, end, value)`` (KeyError for an unknown chromosome, intervals
with value >= valfilter only when valfilter > 0), ``disable_progress_bar``, ``close`` -- so it drops into
``CCHipCalculator(bwfeeder=...)`` and ``MappabilityStats(feeder=...)`` where the reference passes its own reader.
``fetch_arrays`` is the bulk form the calculator prefers: numpy arrays straight into ``pmx_bits_set_regions``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Iterator, Tuple

import numpy as np

from .bam import PmxIOError, _raise, load_io_library  # noqa: F401  (PmxIOError re-exported)

PMX_IO_ERR_NOTFOUND = -4


class BigWigReader:
    def __init__(self, path):
        path_str = os.fspath(path)
        if not os.path.exists(path_str):
            raise IOError("input file '{0}' dose not exist.".format(path_str))     # bigwig.pyx:127-128
        self._L = load_io_library()
        self.path = path_str
        h = ctypes.c_void_p()
        rc = self._L.pmx_bigwig_open(path_str.encode(), ctypes.byref(h))
        if rc:
            _raise(rc)
        self._h = h
        self.closed = False
        n = self._L.pmx_bigwig_nchrom(h)
        self.chromsizes: Dict[str, int] = {
            self._L.pmx_bigwig_chrom_name(h, i).decode(): int(self._L.pmx_bigwig_chrom_len(h, i)) for i in range(n)}

    def fetch_arrays(self, valfilter: float, chrom: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(begin, end, value) arrays of the chromosome's intervals with value >= valfilter."""
        if self.closed:
            raise ValueError("I/O operation on closed BigWig reader")
        if chrom not in self.chromsizes:
            raise KeyError(chrom)
        name = chrom.encode()
        n = self._L.pmx_bigwig_fetch(self._h, name, float(valfilter), 0, None, None, None)
        if n == PMX_IO_ERR_NOTFOUND:
            raise KeyError(chrom)
        if n < 0:
            _raise(n)
        begin = np.empty(n, dtype=np.uint32)
        end = np.empty(n, dtype=np.uint32)
        value = np.empty(n, dtype=np.float32)
        if n:
            m = self._L.pmx_bigwig_fetch(self._h, name, float(valfilter), n, begin.ctypes.data, end.ctypes.data,
                                         value.ctypes.data)
            if m < 0:
                _raise(m)
            assert m == n
        return begin, end, value

    def fetch(self, valfilter: float, chrom: str) -> Iterator[Tuple[int, int, float]]:
        begin, end, value = self.fetch_arrays(valfilter, chrom)
        return iter(zip(begin.tolist(), end.tolist(), value.tolist()))

    def disable_progress_bar(self) -> None:
        pass

    def close(self) -> None:
        if not getattr(self, "closed", True):
            self._L.pmx_bigwig_close(self._h)
            self._h = None
            self.closed = True

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
