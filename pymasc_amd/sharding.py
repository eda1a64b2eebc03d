"""Multi-GPU layer: chromosome jobs -> ranks, and the result exchange.

The reference parallelises over chromosomes with one worker process per chromosome task
(PyMaSC/handler/calc.py:163-235, utils/calc.py:95-145) and ships each per-chromosome result to the
parent over a multiprocessing.Queue (handler/worker.py:234).  Here: one process per GPU, chromosome
jobs assigned longest-processing-time-first, no collective on the data path, and ONE exchange at the
end: an all-gather of the per-chromosome result rows (the genome-wide curve is a Fisher-z merge of
per-chromosome cc, PyMaSC/utils/calc.py:172-241, so the rows must be kept) plus an all-reduce(sum)
of the genome-wide integer totals (mscc.pyx:238-239, stats.py:515-517).  Backend "nccl" is RCCL over
xGMI on the GPU box; the same code runs on "gloo" with CPU tensors in the tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple  # noqa: F401

import torch
import torch.distributed as dist


def lpt_assign(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: job indices per rank, deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return out


def tile_range_assign(nbits: Sequence[int], world_size: int, tile_bits: int = 65536) -> List[List[Tuple[int, int, int]]]:
    """Equal shares of the genome's TILES, not of its chromosomes (round 4): the tiles of all jobs laid end to end are cut into
    world_size contiguous stretches; rank r gets [(job, first tile, tile count), ...] -- whole chromosomes in the middle of
    its stretch, a share of one at either end.  Every sum of the hot path is owned by one tile, so the ranks' partial
    result blocks ADD UP to the chromosome's (pmx_cc_batch_ranges_dev) and ONE all-reduce(sum) of the per-chromosome rows
    is the whole exchange -- BASELINE.json's north star.  LPT over whole chromosomes leaves 3.6 % imbalance on hg38 at 8
    ranks; this leaves less than one tile.  Deterministic on every rank."""
    ntiles = [max(1, (int(b) + tile_bits - 1) // tile_bits) for b in nbits]
    total = sum(ntiles)
    out: List[List[Tuple[int, int, int]]] = [[] for _ in range(world_size)]
    start = 0                                     # first global tile of the current job
    for j, nt in enumerate(ntiles):
        for r in range(world_size):
            lo, hi = total * r // world_size, total * (r + 1) // world_size      # rank r's stretch of the global sequence
            a, b = max(lo, start), min(hi, start + nt)
            if b > a:
                out[r].append((j, a - start, b - a))
        start += nt
    return out


def exchange_partial_rows(partial: torch.Tensor, group=None, force_collectives: bool = False) -> torch.Tensor:
    """partial: int64 [njobs, nrows, stride] -- this rank's SHARE of every chromosome's result block (zeros where it holds
    nothing).  Returns the complete blocks on every rank: one all-reduce(sum) over RCCL / xGMI."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 or (force_collectives and dist.is_initialized()):
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def owner_table(assignment: List[List[int]], njobs: int) -> List[Tuple[int, int]]:
    """job -> (rank, slot on that rank)."""
    table = [(-1, -1)] * njobs
    for r, jobs in enumerate(assignment):
        for s, j in enumerate(jobs):
            table[j] = (r, s)
    return table


_INDEX_CACHE: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _owner_index(assignment: List[List[int]], njobs: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rank, slot) index tensors of every job, built once per assignment (no per-step host->device copies)."""
    key = (tuple(tuple(a) for a in assignment), njobs, str(device))
    hit = _INDEX_CACHE.get(key)
    if hit is None:
        table = owner_table(assignment, njobs)
        hit = (torch.tensor([t[0] for t in table], device=device), torch.tensor([t[1] for t in table], device=device))
        _INDEX_CACHE[key] = hit
    return hit


def exchange_results(local_rows: torch.Tensor, assignment: List[List[int]], njobs: int,
                     group=None, force_collectives: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """local_rows: int64 [max_slots, nrows, stride], this rank's jobs in slot order (unused slots zero).

    Returns (rows[njobs, nrows, stride] in job order on every rank, totals[nrows, stride] = sum over
    all jobs, computed by all-reduce so it can be cross-checked against rows.sum(0)).
    force_collectives: run the all-gather / all-reduce even in a group of one rank (the one-GPU boxes' way to put the
    RCCL code path under test and to time its fixed cost)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    max_slots = max(len(a) for a in assignment)
    assert local_rows.shape[0] == max_slots
    totals = local_rows.sum(dim=0)
    if world == 1 and not (force_collectives and dist.is_initialized()):
        gathered = local_rows.unsqueeze(0)
    else:
        flat = torch.empty((world * max_slots,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype,
                           device=local_rows.device)
        dist.all_gather_into_tensor(flat, local_rows.contiguous(), group=group)   # rank-major concatenation
        gathered = flat.view((world,) + tuple(local_rows.shape))
        dist.all_reduce(totals, op=dist.ReduceOp.SUM, group=group)
    idx_r, idx_s = _owner_index(assignment, njobs, local_rows.device)
    rows = gathered[idx_r, idx_s]
    return rows, totals


# ---------------------------------------------------------------------------------------------------------------
# From the input files: the reference's `-p N` flow (handler/calc.py:163-235, handler/worker.py:68-234), one
# process per GPU instead of one per chromosome task.
# ---------------------------------------------------------------------------------------------------------------

def _collective_device_setup(device, group=None):
    """With a GPU backend (nccl = RCCL) object collectives stage their payload on torch's CURRENT device, which is
    cuda:0 in every fresh process: bind it to this rank's GPU first, or two ranks collide on one device."""
    if not (dist.is_available() and dist.is_initialized()) or device is None:
        return
    try:
        backend = str(dist.get_backend(group))
    except Exception:
        return
    if "nccl" in backend and torch.cuda.is_available():
        torch.cuda.set_device(int(device))


def gather_chromosome_results(local: Dict[str, object], order: Sequence[str], group=None,
                              error: BaseException = None) -> Dict[str, object]:
    """Every rank's {chromosome: BothChromResult} -> the union on every rank, in ``order``.

    The payload is a few KB of integers per chromosome (what the reference pushes through a multiprocessing.Queue,
    worker.py:234); the fixed-shape tensor exchange used by the benchmark is ``exchange_results`` above.
    ``error``: this rank failed -- every rank learns about it from the same collective and raises, instead of the
    healthy ranks waiting for a peer that never arrives (the reference's '__ERROR__' report, worker.py:91-99 ->
    handler/calc.py:205-206)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if error is not None:
            raise error
        parts = [(None, local)]
    else:
        parts = [None] * dist.get_world_size(group)
        mine = (None if error is None else "{}: {}".format(type(error).__name__, error), None if error else local)
        dist.all_gather_object(parts, mine, group=group)
    failed = [(r, p[0]) for r, p in enumerate(parts) if p[0] is not None]
    if failed:
        if error is not None:
            raise error                                    # the failing rank re-raises its own exception (type kept)
        raise RuntimeError("Worker error on rank(s): " + "; ".join("{} [{}]".format(r, msg) for r, msg in failed))
    merged: Dict[str, object] = {}
    for _err, p in parts:
        for chrom, res in p.items():
            if chrom in merged:
                raise RuntimeError("chromosome {} was calculated by two ranks".format(chrom))
            merged[chrom] = res
    return {c: merged[c] for c in order if c in merged}


def reconcile_chromosome_sizes(bam_sizes: Dict[str, int], external_sizes: Dict[str, int]) -> Dict[str, int]:
    """The reference's rule when a mappability track is given (reader/bam.py:217-255 via handler/calc.py:100-115):
    for chromosomes present in both, a mismatch is warned about and the LONGER length is used."""
    import logging
    log = logging.getLogger(__name__)
    out = {}
    for ref, bam_size in bam_sizes.items():
        ext = external_sizes.get(ref)
        if ext is None:
            log.debug("External size for '%s' not found", ref)
            out[ref] = bam_size
            continue
        if ext != bam_size:
            log.warning("'%s' reference length mismatch: SAM/BAM -> %s, External -> %s", ref, format(bam_size, ","),
                        format(ext, ","))
            if bam_size < ext:
                log.warning("Use longer length '%d' for '%s' anyway", ext, ref)
        out[ref] = max(bam_size, ext)
    return out


def run_sharded(bam_path, max_shift: int, read_len: int, mapq_criteria: int, bigwig_path=None,
                references: Sequence[str] = None, skip_ncc: bool = False, device: int = None, context=None,
                chrom2mappable_len=None, group=None, device_ingest: Optional[bool] = None):
    """BAM (+ BigWig) -> genome-wide result on every rank; chromosomes LPT-sharded over the ranks by length.

    Launch: one process per GPU under ``torch.distributed`` (torchrun, or pymasc_amd.launch.spawn_ranks), the process
    group initialised BEFORE this call; ``device`` = this rank's GPU (default LOCAL_RANK).
    Every rank reads its own chromosomes through the .bai index when there is one (the reference requires it for
    its multi-process mode, reader/bam.py:246-262), otherwise it streams the whole BAM through the native reader
    (~30 M records/s on 16 host threads) and keeps its share; the kernels see only the rank's chromosomes; one object
    all-gather at the end, then the reference's aggregation (result.py:301-464 -> pymasc_amd.result.aggregate_results).
    A rank that fails reports through that same all-gather, so every rank raises instead of hanging.
    ``device_ingest``: inflate, walk and filter the BAM file on the GPU and hand the records to the feeders in HBM
    (pymasc_amd.bam_device) -- default: when this is the only rank and it runs on a real GPU; with several
    ranks each takes its own chromosomes through the host reader and the .bai instead of inflating the whole file N times."""
    from .bam import BamReader, feed_bam
    from .bigwig import BigWigReader
    from .calculator import CCHipCalculator
    from .result import aggregate_results

    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    if device is None and context is None and on:
        import os
        device = int(os.environ.get("LOCAL_RANK", "0"))
    _collective_device_setup(device, group)
    local: Dict[str, object] = {}
    names: List[str] = []
    error = None
    if device_ingest is None:
        from . import ffi
        device_ingest = world == 1 and (context is None or isinstance(context, ffi.Context)) and ffi.device_count() > 0
    if device_ingest:
        from .bam_device import DeviceBamReader

        def open_bam():
            return DeviceBamReader(bam_path, device=(context.device if context is not None else (device or 0)))

        def feed(calc, bam, mine):
            return bam.feed(calc, mapq_criteria, references=mine)
    else:
        def open_bam():
            return BamReader(bam_path)

        def feed(calc, bam, mine):
            return feed_bam(calc, bam, mapq_criteria, references=mine)
    try:
        with open_bam() as bam:
            names = [n for n in bam.references if references is None or n in set(references)]
            lengths = dict(zip(bam.references, bam.lengths))
            if bigwig_path is None:
                bw = None
            elif device_ingest:      # the track decoded on the GPU too: its intervals reach the calculator in HBM
                from .bigwig_device import DeviceBigWigReader
                bw = DeviceBigWigReader(bigwig_path, device=(context.device if context is not None else (device or 0)))
            else:
                bw = BigWigReader(bigwig_path)
            try:
                if bw is not None:      # the track's chromosome sizes win where they are longer (handler/calc.py:100-115)
                    lengths.update(reconcile_chromosome_sizes({n: lengths[n] for n in names}, bw.chromsizes))
                mine = [names[i] for i in sorted(lpt_assign([lengths[n] for n in names], world)[rank])]
                kw = {}
                if context is not None:
                    kw["context"] = context
                elif device is not None:
                    kw["device"] = device
                if mine:
                    calc = CCHipCalculator(max_shift, read_len, mine, [lengths[n] for n in mine], bwfeeder=bw,
                                           skip_ncc=skip_ncc, chrom2mappable_len=chrom2mappable_len, **kw)
                    try:
                        feed(calc, bam, mine)
                        local = {c: calc.get_result(c) for c in mine}
                    finally:
                        if context is None:
                            calc.close()
            finally:
                if bw is not None:
                    bw.close()
    except Exception as e:              # surfaced on every rank by the gather below
        if not on or world == 1:
            raise
        error = e
    merged = gather_chromosome_results(local, names, group, error=error)
    return aggregate_results(merged)
