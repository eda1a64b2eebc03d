"""BigWig (bbi) ingest ON THE DEVICE: binding of the pmx_dbw_* entry points of libpymasc_ingest.so.

``DeviceBigWigReader`` has the surface of ``pymasc_amd.bigwig.BigWigReader`` -- the part of the reference's
PyMaSC/reader/bigwig.pyx the calculation touches: ``chromsizes``, ``fetch(valfilter, chrom)``, ``fetch_arrays``, ``close`` -- plus
``fetch_device``, which leaves a chromosome's intervals in HBM for ``CCHipCalculator`` to build its mappability vector from
(pmx_bits_set_regions_dev_ex).  The file is copied to the GPU once; data blocks are inflated (the BGZF kernel), Adler-32-checked and
decoded by HIP kernels.  No host fallback: without a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Iterator, Tuple

import numpy as np

from .bam import PmxIOError
from .bam_device import load_ingest_library

PMX_DBAM_ERR_NOTFOUND = -4


def _raise(code: int):
    raise PmxIOError(int(code), load_ingest_library().pmx_dbam_last_error().decode("utf-8", "replace"))


class DeviceBigWigReader:
    def __init__(self, path, device: int = 0, threads: int = 0):
        path_str = os.fspath(path)
        if not os.path.exists(path_str):
            raise IOError("input file '{0}' dose not exist.".format(path_str))     # bigwig.pyx:127-128
        self._L = load_ingest_library()
        self.path = path_str
        self._h = None
        h = ctypes.c_void_p()
        rc = self._L.pmx_dbw_open(path_str.encode(), int(device), int(threads), ctypes.byref(h))
        if rc:
            _raise(rc)
        self._h = h
        self.closed = False
        n = self._L.pmx_dbw_nchrom(h)
        self.chromsizes: Dict[str, int] = {self._L.pmx_dbw_chrom_name(h, i).decode(): int(self._L.pmx_dbw_chrom_len(h, i))
                                           for i in range(n)}

    def fetch_device(self, valfilter: float, chrom: str) -> Tuple[int, int, int, bool]:
        """The chromosome's intervals with value >= valfilter, left in device memory: (address of uint32 begin[], address of
        uint32 end[], count, sorted and disjoint?) -- valid until the next fetch on this reader."""
        if self.closed:
            raise ValueError("I/O operation on closed BigWig reader")
        if chrom not in self.chromsizes:
            raise KeyError(chrom)
        n = self._L.pmx_dbw_fetch(self._h, chrom.encode(), float(valfilter))
        if n == PMX_DBAM_ERR_NOTFOUND:
            raise KeyError(chrom)
        if n < 0:
            _raise(n)
        b, e, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._L.pmx_dbw_device_arrays(self._h, ctypes.byref(b), ctypes.byref(e), ctypes.byref(v))
        if rc:
            _raise(rc)
        return int(b.value or 0), int(e.value or 0), int(n), bool(self._L.pmx_dbw_sorted(self._h))

    def fetch_arrays(self, valfilter: float, chrom: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(begin, end, value) host arrays, as BigWigReader.fetch_arrays."""
        _b, _e, n, _s = self.fetch_device(valfilter, chrom)
        begin = np.empty(n, dtype=np.uint32)
        end = np.empty(n, dtype=np.uint32)
        value = np.empty(n, dtype=np.float32)
        if n:
            rc = self._L.pmx_dbw_copy(self._h, 0, n, begin.ctypes.data, end.ctypes.data, value.ctypes.data)
            if rc:
                _raise(rc)
        return begin, end, value

    def fetch(self, valfilter: float, chrom: str) -> Iterator[Tuple[int, int, float]]:
        begin, end, value = self.fetch_arrays(valfilter, chrom)
        return iter(zip(begin.tolist(), end.tolist(), value.tolist()))

    def disable_progress_bar(self) -> None:
        pass

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            self._L.pmx_dbw_close(self._h)
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
