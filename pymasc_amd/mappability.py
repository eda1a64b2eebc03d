"""Mappable-length statistics: GPU pre-calculation and the ``*_mappability.json`` cache.

Reference behaviour restated here:

* ``MappableLengthCalculator`` (PyMaSC/core/mappability.pyx:38-205): per chromosome and lag k = 0..max_shift,
  ``mappable_len[k] = #{j : M[j] and M[j+k]}`` over the BigWig intervals with value >= 1 (threshold, :93); the
  genome-wide table is the per-lag sum over chromosomes (:126-131).  The reference streams the intervals through a
  sliding buffer + np.correlate (:166-205); here the intervals become a bit-vector in HBM and the table comes from
  the run-edge autocorrelation kernel (``pmx_mappable_len_dev``), the same integers by construction.
* ``MappabilityHandler`` (PyMaSC/handler/mappability.py:99-309): the lag range needed for a run
  (``calc_mappable_len_required_shift_size``, :122-136), the cache path derived from the BigWig path (:196-199),
  the validity rules when reading a cache (:234-262), its JSON layout (:296-301: keys ``max_shift``,
  ``__whole__``, ``references``; indent 4, sorted keys) and the "recompute when the stored range is too short"
  rule (:245-247).

The interval source is any object with ``chromsizes`` (name -> length) and ``fetch(threshold, chrom)`` yielding
``(begin, end, value)`` -- the contract of the reference's BigWigReader (reader/bigwig.pyx:147-177).
"""
from __future__ import annotations

import json
import logging
import os
from pathlib import Path
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import ffi

logger = logging.getLogger(__name__)

MAPPABILITY_THRESHOLD = 1.0        # mappability.pyx:93
_PAD_BITS = 64                     # slack past the last interval; the autocorrelation needs none


class BWIOError(IOError):
    """The mappability track cannot be read (handler/mappability.py:43-47)."""


class JSONIOError(IOError):
    """The statistics cache cannot be read or written (handler/mappability.py:50-54)."""


class NeedUpdate(Exception):
    """The cache covers fewer lags than this run needs (handler/mappability.py:57-61)."""


class _IntEncoder(json.JSONEncoder):
    """numpy scalars / arrays as plain JSON numbers / lists (handler/mappability.py:64-86)."""

    def default(self, obj):
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return super().default(obj)


def required_shift_size(readlen: int, max_shift: int) -> int:
    """Largest lag the MSCC normalisation reads: |d - (L-1)| over d in [0, max_shift], expressed the way the
    reference does (handler/mappability.py:122-136)."""
    return max_shift - readlen + 1 if max_shift > 2 * readlen - 1 else readlen


def default_stats_path(track_path) -> Path:
    """``/dir/name.bw`` -> ``/dir/name_mappability.json`` (handler/mappability.py:196-199)."""
    p = Path(track_path)
    return p.parent / (p.with_suffix("").name + "_mappability.json")


def read_stats(path, need_shift: int, references: Iterable[str]) -> Dict[str, Any]:
    """Parse and validate a cache file (handler/mappability.py:234-262).

    KeyError: a mandatory key or a reference is missing.  IndexError: a table's length disagrees with
    ``max_shift``.  NeedUpdate: the stored range is shorter than ``need_shift``."""
    with open(path) as fp:
        stats = json.load(fp)
    for key in ("max_shift", "__whole__", "references"):
        if key not in stats:
            raise KeyError(key)
    if stats["max_shift"] < need_shift:
        raise NeedUpdate
    if stats["max_shift"] != len(stats["__whole__"]) - 1:
        raise IndexError("__whole__")
    for ref in references:
        if ref not in stats["references"]:
            raise KeyError(ref)
        if stats["max_shift"] != len(stats["references"][ref]) - 1:
            raise IndexError(ref)
    return stats


def write_stats(path, max_shift: int, whole: Sequence[int], per_chrom: Mapping[str, Sequence[int]]) -> None:
    """The cache file, laid out like the reference's (handler/mappability.py:294-301).  Written to a temporary file in
    the same directory and renamed into place, so a concurrent reader (another rank, another run) sees either the old
    file or the complete new one, never a truncated one."""
    path = str(path)
    tmp = "{}.tmp.{}".format(path, os.getpid())
    try:
        with open(tmp, "w") as fp:
            json.dump({"max_shift": max_shift, "__whole__": list(whole),
                       "references": {c: list(v) for c, v in per_chrom.items()}},
                      fp, indent=4, sort_keys=True, cls=_IntEncoder)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


class MappabilityStats:
    """Per-chromosome / genome-wide mappable length by lag, computed on the GPU, cached as JSON.

    Mirrors the public surface of the reference's MappabilityHandler that the rest of PyMaSC touches:
    ``chromsizes``, ``max_shift``, ``chrom2is_called``, ``chrom2mappable_len``, ``mappable_len``, ``is_called``,
    ``need_save_stats``, ``calc_mappability``, ``get_mappable_len``, ``save_mappability_stats``.
    """

    def __init__(self, feeder: Any, max_shift: int = 0, readlen: int = 0, map_path=None, track_path=None,
                 device: int = 0, context: Optional[ffi.Context] = None):
        self.feeder = feeder
        self.chromsizes: Dict[str, int] = dict(feeder.chromsizes)
        self.max_shift = required_shift_size(int(readlen), int(max_shift))
        self.chrom2is_called = {c: False for c in self.chromsizes}
        self.chrom2mappable_len: Dict[str, Tuple[int, ...]] = {}
        self.mappable_len: List[int] = [0] * (self.max_shift + 1)
        self.is_called = False
        self.need_save_stats = True
        self._ctx = context
        self._own_ctx = False
        self._device = device

        if map_path is not None:
            self.map_path: Optional[Path] = Path(map_path)
        elif track_path is not None:
            self.map_path = default_stats_path(track_path)
        else:
            self.map_path = None

        if self.map_path is None:
            return
        if not self.map_path.exists():
            parent = self.map_path.parent
            if not parent.is_dir() or not os.access(parent, os.W_OK):
                logger.critical("Directory is not writable: '{}'".format(parent))
                raise JSONIOError(str(parent))
            logger.info("Calcurate mappable length with max shift size {}.".format(self.max_shift))
        elif not self.map_path.is_file():
            logger.critical("Specified path is not file: '{}'".format(self.map_path))
            raise JSONIOError(str(self.map_path))
        elif not os.access(self.map_path, os.R_OK):
            logger.error("Failed to read '{}'".format(self.map_path))
        else:
            self._try_load()
            if self.need_save_stats:
                if not os.access(self.map_path, os.W_OK):
                    logger.critical("Failed to overwrite '{}'".format(self.map_path))
                    raise JSONIOError(str(self.map_path))
                logger.warning("Existing file '{}' will be overwritten.".format(self.map_path))
            else:
                logger.info("Use mappability stats read from '{}'".format(self.map_path))

    # ---- cache ------------------------------------------------------------------------------------------
    def _try_load(self) -> None:
        try:
            stats = read_stats(self.map_path, self.max_shift, self.chromsizes)
        except IOError as e:
            logger.error("Failed to read '{}'".format(self.map_path))
            logger.error("[Errno {}] {}".format(e.errno, str(e)))
        except (TypeError, OverflowError, ValueError, KeyError, IndexError):
            logger.error("Failed to load json file: '{}'".format(self.map_path))
        except NeedUpdate:
            logger.info("Specified shift length longer than former analysis. The stats will be updated.")
        else:
            n = self.max_shift + 1
            self.mappable_len = list(stats["__whole__"][:n])
            self.chrom2mappable_len = {ref: tuple(tab[:n]) for ref, tab in stats["references"].items()}
            self.chrom2is_called = {ref: True for ref in self.chromsizes}
            self.is_called = True
            self.need_save_stats = False

    def save_mappability_stats(self) -> None:
        if not self.need_save_stats:
            logger.info("Mappability stats updating is not required.")
            return
        if self.map_path is None:
            raise JSONIOError("no path to save the mappability stats to")
        if not self.is_called:
            self.calc_mappability()
        logger.info("Save mappable length to '{}'".format(self.map_path))
        try:
            write_stats(self.map_path, self.max_shift, self.mappable_len, self.chrom2mappable_len)
        except IOError as e:
            logger.error("Faild to output: {}\n[Errno {}] {}".format(e.filename, e.errno, str(e)))
        self.need_save_stats = False

    # ---- calculation ------------------------------------------------------------------------------------
    def _context(self) -> ffi.Context:
        if self._ctx is None:
            self._ctx = ffi.Context(self._device)      # raises PmxError without GPU / library: no CPU path
            self._own_ctx = True
        return self._ctx

    def close(self) -> None:
        if self._own_ctx and self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._own_ctx = False

    def _calc_chrom(self, chrom: str) -> Tuple[int, ...]:
        ctx = self._context()
        logger.info("Calc {} mappable length...".format(chrom))
        bulk = getattr(self.feeder, "fetch_arrays", None)
        if bulk is not None:
            begin, end, _v = bulk(MAPPABILITY_THRESHOLD, chrom)
            arr = np.stack((begin.astype(np.int64), end.astype(np.int64)), axis=1)
        else:
            iv = [(b, e) for b, e, _v in self.feeder.fetch(MAPPABILITY_THRESHOLD, chrom)]
            arr = np.asarray(iv, dtype=np.int64).reshape(-1, 2)
        nlag = self.max_shift + 1
        if arr.shape[0] == 0:
            return tuple([0] * nlag)
        if (arr[:, 1] <= arr[:, 0]).any() or arr.min() < 0:
            raise ValueError("malformed mappability interval on {}".format(chrom))
        nbits = int(max(arr[:, 1].max(), self.chromsizes.get(chrom, 0))) + _PAD_BITS
        d_m = ctx.bits_alloc(nbits)
        d_out = ctx.bits_alloc(nlag * 64)
        try:
            # [begin, end) 0-based -> bits begin .. end-1; the table depends on differences only
            ctx.bits_set_regions(d_m, nbits, arr[:, 0].copy(), arr[:, 1] - 1)
            ctx.mappable_len_dev(d_m, nbits, self.max_shift, 0, d_out)
            out = ctx.bits_download(d_out, nlag * 64)
        finally:
            ctx.bits_free(d_m)
            ctx.bits_free(d_out)
        return tuple(int(x) for x in out[:nlag])

    def calc_mappability(self, chrom: Optional[str] = None) -> None:
        """mappability.pyx:133-164: one chromosome, or every one not yet done."""
        if not chrom:
            chroms = [c for c, done in self.chrom2is_called.items() if not done]
        elif self.chrom2is_called[chrom]:
            return
        else:
            chroms = [chrom]
        for c in chroms:
            table = self._calc_chrom(c)
            self.chrom2mappable_len[c] = table
            for i, v in enumerate(table):
                self.mappable_len[i] += v
            self.chrom2is_called[c] = True
        if all(self.chrom2is_called.values()):
            self.is_called = True

    def get_mappable_len(self, chrom: Optional[str] = None, shift_from: Optional[int] = None,
                         shift_to: Optional[int] = None, force: bool = False):
        """mappability.pyx:207-246."""
        if chrom is not None:
            if chrom not in self.chrom2is_called:
                return None
            if self.chrom2is_called[chrom]:
                return self.chrom2mappable_len[chrom][shift_from:shift_to]
            if force:
                self.calc_mappability(chrom)
                return self.chrom2mappable_len[chrom][shift_from:shift_to]
            raise KeyError("Mappable length for '{}' is not calculated yet.".format(chrom))
        if not self.is_called:
            self.calc_mappability()
        return self.mappable_len[shift_from:shift_to]
