"""Synthetic workloads of BASELINE.md section 4 (configs 4 and 5), generated ON the GPU with torch
and turned into HBM bit-vectors by the library's own builders (pmx_bits_set_positions_dev /
pmx_bits_set_regions_dev), i.e. through the same ingest kernels real reads and BigWig intervals use.

F, R: i.i.d. positions at density rho per strand plus a planted fragment peak (30 % of forward reads
get a reverse partner at +180); M: alternating runs, geometric lengths, mean 2000 mappable / 500 not
(~80 % ones).  Seed = seed_base + chromosome index.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

HG38 = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
    ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
    ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
    ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
]


def stress_genome(n_chroms: int = 200, total_bp: float = 1e10, seed: int = 0xBADC0DE):
    """Config 5: lengths log-uniform in [5e6, 2.5e8], rescaled to sum to total_bp."""
    g = torch.Generator().manual_seed(seed)
    import math
    u = torch.rand(n_chroms, generator=g, dtype=torch.float64)
    lens = torch.exp(math.log(5e6) + u * (math.log(2.5e8) - math.log(5e6)))
    lens = (lens * (total_bp / lens.sum())).round().to(torch.int64)
    return [(f"s{i}", int(l)) for i, l in enumerate(lens.tolist())]


_FIXTURE_RUNS = None


def fixture_run_lengths(path: Optional[str] = None):
    """Empirical run statistics of the only real mappability track available offline: the reference's test fixture
    `hg19_36mer-test.bedGraph` (committed under tests/golden/; 985 mappable runs -- maximal stretches of value >= 1, the
    filter of reader/bigwig.pyx:174-175 -- over 149 kb of chr1: mean run 35 bp, mean gap 117 bp, 23 % mappable, 860 run
    edges per 64 Kbit).  Returns (run lengths, gap lengths) as int64 numpy arrays."""
    global _FIXTURE_RUNS
    if _FIXTURE_RUNS is None:
        import os
        import numpy as np
        path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                    "hg19_36mer-test.bedGraph")
        runs = []
        for line in open(path):
            f = line.split()
            if len(f) < 4 or float(f[3]) < 1.0:
                continue
            b, e = int(f[1]), int(f[2])
            if runs and runs[-1][1] == b:
                runs[-1][1] = e
            else:
                runs.append([b, e])
        r = np.asarray(runs, dtype=np.int64)
        _FIXTURE_RUNS = (r[:, 1] - r[:, 0], r[1:, 0] - r[:-1, 1])
    return _FIXTURE_RUNS


@dataclass
class ChromVectors:
    name: str
    length: int
    nbits: int
    F: torch.Tensor          # int64 words on the GPU (bit layout of include/pymasc_amd.h)
    R: torch.Tensor
    M: Optional[torch.Tensor]
    n_forward: int
    n_reverse: int
    n_runs: int = 0          # mappable runs of M (the calculator's PMX_FLAG_WINDOW_ONLY hint looks at runs per bp)
    # host copies of what the vectors were built from (keep_host=True): read positions and set(first, last) intervals
    h_fpos: Optional["numpy.ndarray"] = None
    h_rpos: Optional["numpy.ndarray"] = None
    h_first: Optional["numpy.ndarray"] = None
    h_last: Optional["numpy.ndarray"] = None


def nwords(nbits: int) -> int:
    return (nbits + 63) // 64


def make_chromosome(ctx, device, name: str, length: int, max_shift: int, read_len: int, seed: int,
                    density: float = 0.005, with_m: bool = True, peak_frac: float = 0.3, peak_shift: int = 180,
                    mean_on: float = 2000.0, mean_off: float = 500.0, keep_host: bool = False,
                    track: str = "synthetic") -> ChromVectors:
    """track = "synthetic": BASELINE.md's geometric runs (mean_on / mean_off); "fixture": run and gap lengths drawn
    independently from the empirical distribution of the reference's test track (fixture_run_lengths), tiled to the
    chromosome's length -- the statistics of a real 36-mer track's worst stretch everywhere."""
    g = torch.Generator(device=device).manual_seed(seed)
    nbits = length + read_len + max_shift + 100
    nw = nwords(nbits)
    F = torch.zeros(nw, dtype=torch.int64, device=device)
    R = torch.zeros(nw, dtype=torch.int64, device=device)
    n = max(1, int(length * density))
    fpos = torch.randint(1, length + 1, (n,), generator=g, device=device, dtype=torch.int64)
    n_peak = int(n * peak_frac)
    rpos_bg = torch.randint(1, length + 1, (n - n_peak,), generator=g, device=device, dtype=torch.int64)
    pick = torch.randperm(n, generator=g, device=device)[:n_peak]
    rpos = torch.cat([rpos_bg, torch.clamp(fpos[pick] + peak_shift, max=length + read_len - 1)])
    torch.cuda.current_stream(device).synchronize()
    ctx.bits_set_positions_dev(F.data_ptr(), nbits, fpos.data_ptr(), fpos.numel())
    ctx.bits_set_positions_dev(R.data_ptr(), nbits, rpos.data_ptr(), rpos.numel())
    host = {}
    if keep_host:
        host.update(h_fpos=fpos.cpu().numpy(), h_rpos=rpos.cpu().numpy())
    M = None
    if with_m:
        M = torch.zeros(nw, dtype=torch.int64, device=device)
        if track == "fixture":
            e_on, e_off = fixture_run_lengths()
            nruns = int(length / (float(e_on.mean()) + float(e_off.mean())) * 1.1) + 64
            t_on = torch.from_numpy(e_on).to(device)
            t_off = torch.from_numpy(e_off).to(device)
            on = t_on[torch.randint(0, t_on.numel(), (nruns,), generator=g, device=device)]
            off = t_off[torch.randint(0, t_off.numel(), (nruns,), generator=g, device=device)]
        else:
            nruns = int(length / (mean_on + mean_off) * 1.3) + 16
            on = torch.empty(nruns, device=device, dtype=torch.float64).exponential_(1.0 / mean_on, generator=g)
            off = torch.empty(nruns, device=device, dtype=torch.float64).exponential_(1.0 / mean_off, generator=g)
            on = on.floor().to(torch.int64) + 1
            off = off.floor().to(torch.int64) + 1
        ends = torch.cumsum(on + off, 0)          # exclusive end of each on-run, 0-based like BigWig
        begins = ends - on
        keep = begins < length
        first = (begins[keep] + 1).contiguous()   # set(begin + 1, end)  (mscc.pyx:344)
        last = torch.clamp(ends[keep], max=length).contiguous()
        torch.cuda.current_stream(device).synchronize()
        ctx.bits_set_regions_dev(M.data_ptr(), nbits, first.data_ptr(), last.data_ptr(), first.numel())
        if keep_host:
            host.update(h_first=first.cpu().numpy(), h_last=last.cpu().numpy())
    ctx.sync()
    return ChromVectors(name, length, nbits, F, R, M, int(fpos.numel()), int(rpos.numel()),
                        n_runs=int(first.numel()) if with_m else 0, **host)


def make_genome(ctx, device, chroms, max_shift: int, read_len: int, seed_base: int = 0xC0FFEE,
                density: float = 0.005, with_m: bool = True, indices=None) -> List[ChromVectors]:
    out = []
    for i, (name, length) in enumerate(chroms):
        if indices is not None and i not in indices:
            continue
        out.append(make_chromosome(ctx, device, name, length, max_shift, read_len, seed_base + i, density, with_m))
    return out
