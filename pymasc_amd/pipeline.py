"""One call from the input files to the output tables: the hot path with the rows either side of it.

What `pymasc sample.bam -m track.bw -d MAX_SHIFT -q MAPQ -r READ_LEN -o OUTDIR` does between argument parsing and the
statistics / plots (PyMaSC/pymasc.py:90-160, handler/calc.py:100-161): open the BAM, open the mappability track,
load or compute the mappable-length cache, run the calculator over the reads, write the `_cc` / `_mscc` / `_nreads`
tables.  No CLI, no read-length estimation, no statistics: those stay the reference's.
Under `torch.distributed` (one process per GPU) the chromosomes are sharded over the ranks and rank 0 writes.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional, Sequence

from . import tables
from .mappability import MappabilityStats
from .sharding import run_sharded


def run(bam_path, outdir, max_shift: int, read_len: int, mapq_criteria: int = 1, mappability_path=None,
        mappability_stats_path=None, skip_ncc: bool = False, references: Optional[Sequence[str]] = None,
        device: Optional[int] = None, save_mappability_stats: bool = True, group=None, context=None,
        device_ingest: Optional[bool] = None):
    """Returns (genome-wide result, [paths written]).  ``outdir/<bam stem>_{cc,mscc,nreads}.tab`` are written by
    rank 0 (every rank holds the result).  ``context``: an existing pymasc_amd.ffi.Context to run on (default: one per
    call on ``device``).  ``device_ingest``: see sharding.run_sharded (default: the BAM file is inflated and decoded on the GPU
    when there is one rank on a real GPU)."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if on else 0

    # The mappable-length cache (handler/mappability.py:239-309): loaded when valid; otherwise computed ONCE, on rank 0,
    # written atomically, and broadcast -- the other ranks neither recompute it per chromosome nor read a file that is
    # being rewritten.
    known = None
    if mappability_path is not None:
        from .bigwig import BigWigReader
        from .sharding import _collective_device_setup
        _collective_device_setup(device, group)
        box = [None, None]          # [chrom2mappable_len, error message]
        if rank == 0:
            try:
                with BigWigReader(mappability_path) as bw:
                    stats = MappabilityStats(bw, max_shift, read_len, map_path=mappability_stats_path,
                                             track_path=mappability_path, device=device, context=context)
                    try:
                        if stats.is_called:                      # a valid cache: the autocorrelation pass is skipped
                            box[0] = stats.chrom2mappable_len
                        elif save_mappability_stats:
                            stats.calc_mappability()
                            stats.save_mappability_stats()
                            box[0] = stats.chrom2mappable_len
                    finally:
                        stats.close()
            except Exception as e:                               # every rank must learn about it (no hang below)
                box[1] = "{}: {}".format(type(e).__name__, e)
                if not on or dist.get_world_size(group) == 1:
                    raise
        if on and dist.get_world_size(group) > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[1] is not None:
            raise RuntimeError("mappability statistics failed on rank 0 [{}]".format(box[1]))
        known = box[0]
    result = run_sharded(bam_path, max_shift, read_len, mapq_criteria, bigwig_path=mappability_path,
                         references=references, skip_ncc=skip_ncc, device=device, chrom2mappable_len=known,
                         group=group, context=context, device_ingest=device_ingest)
    written: List[Path] = []
    if rank == 0:
        out = Path(outdir)
        out.mkdir(parents=True, exist_ok=True)
        written = tables.write_tables(out / Path(bam_path).name, result)
    return result, written
