"""Multi-rank path on CPU: chromosome jobs LPT-sharded over world_size-2 `gloo` ranks, per-chromosome rows
all-gathered and genome totals all-reduced (pymasc_amd.sharding), compared with a single-process run.
The per-rank compute is the oracle here (no GPU in this container); on the GPU box the same exchange code
runs over RCCL with rows produced by the HIP kernels (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pymasc_amd import sharding

NROWS, S, L = 6, 40, 12
LENGTHS = [9000, 4000, 7000, 1200, 6500, 3000, 800]


def _rows_for(job):
    """Deterministic 'result block' of a job, computed with the CPU oracle on seeded synthetic vectors."""
    from oracle import model as oracle
    from tests import synth
    nbits, F, R, M = synth.make_case(1000 + job, LENGTHS[job], S, L, 0.03, 0.03, True, mean_on=60, mean_off=20)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = np.zeros((NROWS, S + 1), dtype=np.int64)
    out[0] = ref["ncc_ccbins"]
    out[1] = ref["mscc_forward_sum"]
    out[2] = ref["mscc_reverse_sum"]
    out[3] = ref["mscc_ccbins"]
    out[4] = ref["mappable_len_by_shift"]
    out[5, 0] = ref["ncc_forward_sum"]
    out[5, 1] = ref["ncc_reverse_sum"]
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assignment = sharding.lpt_assign(LENGTHS, world)
        max_slots = max(len(a) for a in assignment)
        local = torch.zeros((max_slots, NROWS, S + 1), dtype=torch.int64)
        for slot, job in enumerate(assignment[rank]):
            local[slot] = torch.from_numpy(_rows_for(job))
        rows, totals = sharding.exchange_results(local, assignment, len(LENGTHS))
        q.put((rank, rows.numpy(), totals.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_lpt_assign_balances_and_is_deterministic():
    a = sharding.lpt_assign(LENGTHS, 2)
    assert sorted(sum(a, [])) == list(range(len(LENGTHS)))
    loads = [sum(LENGTHS[j] for j in jobs) for jobs in a]
    assert max(loads) - min(loads) <= max(LENGTHS)
    assert a == sharding.lpt_assign(LENGTHS, 2)
    assert sharding.lpt_assign(LENGTHS, 1) == [sorted(range(len(LENGTHS)), key=lambda i: (-LENGTHS[i], i))]
    table = sharding.owner_table(a, len(LENGTHS))
    for r, jobs in enumerate(a):
        for s, j in enumerate(jobs):
            assert table[j] == (r, s)


@pytest.mark.timeout(300)
def test_two_rank_exchange_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.stack([_rows_for(j) for j in range(len(LENGTHS))])
    for rank, rows, totals in got:
        np.testing.assert_array_equal(rows, expect)                 # every rank holds every chromosome's rows
        np.testing.assert_array_equal(totals, expect.sum(axis=0))   # all-reduced genome totals


def test_single_process_exchange_is_identity():
    a = sharding.lpt_assign(LENGTHS, 1)
    local = torch.stack([torch.from_numpy(_rows_for(j)) for j in a[0]])
    rows, totals = sharding.exchange_results(local, a, len(LENGTHS))
    expect = np.stack([_rows_for(j) for j in range(len(LENGTHS))])
    np.testing.assert_array_equal(rows.numpy(), expect)
    np.testing.assert_array_equal(totals.numpy(), expect.sum(axis=0))


# ---- from the input files: BAM + BigWig -> per-rank calculators -> gathered genome-wide result -> tables ----------
def _write_inputs(tmp):
    from tests import io_writers as W
    rng = np.random.default_rng(77)
    refs = [("c1", 40000), ("c2", 25000), ("c3", 31000), ("c4", 9000), ("c5", 18000)]
    recs, meta = W.synth_bam_records(rng, refs[:4], 900)           # c5 has no reads
    bam = os.path.join(tmp, "s.bam")
    # indexed: a rank that owns a subset of the chromosomes reads them through the .bai, the single-process run
    # (all chromosomes) takes one pass over the file -- both must agree
    W.write_bam_indexed(bam, refs, recs, [int(x) for x in meta[:, 0]], block=3000)
    tracks = {}
    for name, size in refs[:3] + refs[4:]:                           # c4 has no mappability track
        iv, p = [], int(rng.integers(0, 40))
        while p < size - 300:
            ln = int(rng.integers(20, 400))
            iv.append((p, p + ln, float(rng.choice([0.5, 1.0, 1.0, 1.0]))))
            p += ln + int(rng.integers(1, 90))
        tracks[name] = iv
    bw = os.path.join(tmp, "m.bw")
    W.write_bigwig(bw, {n: s for n, s in refs if n in tracks}, tracks, items_per_block=40)
    return bam, bw


def _table_bytes(result, tmp, tag):
    from pymasc_amd import tables
    paths = tables.write_tables(os.path.join(tmp, tag + ".bam"), result)
    return [open(p, "rb").read() for p in paths]


def _file_worker(rank, world, port, q, bam, bw, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.fake_context import FakeContext
        res = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, context=FakeContext())
        q.put((rank, _table_bytes(res, tmp, "rank%d" % rank), res.forward_sum, res.reverse_sum, res.genomelen))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_run_from_files_matches_single_process(tmp_path):
    from tests.fake_context import FakeContext
    tmp = str(tmp_path)
    bam, bw = _write_inputs(tmp)
    single = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, context=FakeContext())
    assert set(single.chroms) == {"c1", "c2", "c3", "c4", "c5"}
    expect = _table_bytes(single, tmp, "single")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_file_worker, args=(r, world, port, q, bam, bw, tmp)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, tabs, fsum, rsum, glen in got:
        assert tabs == expect, rank                      # byte-identical _cc / _mscc / _nreads tables on every rank
        assert (fsum, rsum, glen) == (single.forward_sum, single.reverse_sum, single.genomelen)


@pytest.mark.gpu
def test_run_from_files_gpu_matches_host_restatement(tmp_path):
    """Same inputs through the HIP path (single rank) and through the oracle-backed stand-in: identical tables."""
    from tests.fake_context import FakeContext
    tmp = str(tmp_path)
    bam, bw = _write_inputs(tmp)
    host = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, context=FakeContext())
    gpu = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, device=0)
    assert _table_bytes(gpu, tmp, "gpu") == _table_bytes(host, tmp, "host")
    for skip_ncc in (True,):
        host = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, skip_ncc=skip_ncc, context=FakeContext())
        gpu = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, skip_ncc=skip_ncc, device=0)
        assert _table_bytes(gpu, tmp, "gpu2") == _table_bytes(host, tmp, "host2")


# ---- chromosome sizes: the track wins where it is longer (handler/calc.py:100-115, reader/bam.py:217-255) ----------
def test_reconcile_chromosome_sizes_rule():
    out = sharding.reconcile_chromosome_sizes({"a": 100, "b": 200, "c": 300}, {"a": 150, "b": 180, "z": 5})
    assert out == {"a": 150, "b": 200, "c": 300}


def test_track_chromosome_longer_than_the_bam_header(tmp_path):
    """A BigWig whose chromosome is longer than the BAM's, with mappable intervals beyond the BAM length: the reference
    uses the longer length for genomelen and the vectors; before the fix the intervals overran the bit-vector
    (PMX_ERR_INVALID) and genomelen differed."""
    from tests import io_writers as W
    from tests.fake_context import FakeContext
    from pymasc_amd.bam import BamReader, feed_bam
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.calculator import CCHipCalculator
    tmp = str(tmp_path)
    rng = np.random.default_rng(5)
    refs = [("c1", 30000), ("c2", 20000)]
    recs, meta = W.synth_bam_records(rng, refs, 600)
    bam = os.path.join(tmp, "s.bam")
    W.write_bam_indexed(bam, refs, recs, [int(x) for x in meta[:, 0]], block=3000)
    sizes = {"c1": 33000, "c2": 19000}                    # c1 longer in the track, c2 shorter
    tracks = {"c1": [(100, 9000, 1.0), (12000, 29990, 1.0), (30500, 32990, 1.0)], "c2": [(0, 18000, 1.0)]}
    bw = os.path.join(tmp, "m.bw")
    W.write_bigwig(bw, sizes, tracks, items_per_block=40)
    res = sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, context=FakeContext())
    assert res.genomelen == 33000 + 20000
    assert res.chroms["c1"].genomelen == 33000 and res.mappable_chroms["c2"].genomelen == 20000
    # the same through a calculator built with the reconciled lengths by hand
    with BamReader(bam) as b, BigWigReader(bw) as w:
        calc = CCHipCalculator(120, 36, ["c1", "c2"], [33000, 20000], bwfeeder=w, context=FakeContext())
        feed_bam(calc, b, 10)
    for c in ("c1", "c2"):
        assert list(res.mappable_chroms[c].ccbins) == list(calc.get_result(c).mappable_chrom.ccbins)
        assert list(res.mappable_chroms[c].mappable_len) == list(calc.get_result(c).mappable_chrom.mappable_len)
        assert list(res.chroms[c].ccbins) == list(calc.get_result(c).chrom.ccbins)
    assert res.mappable_chroms["c1"].mappable_len[0] == (9000 - 100) + (29990 - 12000) + (32990 - 30500)


# ---- pipeline.run under several ranks: the lag cache is computed once, written atomically, broadcast ---------------
def _pipeline_worker(rank, world, port, q, bam, bw, tmp, tag):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pymasc_amd import pipeline, mappability
        from tests.fake_context import FakeContext
        calls = []
        orig = mappability.MappabilityStats.calc_mappability
        mappability.MappabilityStats.calc_mappability = lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1]
        res, written = pipeline.run(bam, os.path.join(tmp, "out_%s_%d" % (tag, rank)), 120, 36, 10, mappability_path=bw,
                                    context=FakeContext())
        tabs = [open(p, "rb").read() for p in written]
        q.put((rank, tabs, len(calls), [os.path.basename(str(p)) for p in written]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_pipeline_computes_the_lag_cache_once(tmp_path):
    from pymasc_amd import pipeline
    from tests.fake_context import FakeContext
    tmp = str(tmp_path)
    bam, bw = _write_inputs(tmp)
    cache = os.path.join(tmp, "m_mappability.json")
    single, written = pipeline.run(bam, os.path.join(tmp, "single"), 120, 36, 10, mappability_path=bw,
                                   context=FakeContext(), save_mappability_stats=False)
    assert not os.path.exists(cache)
    expect = [open(p, "rb").read() for p in written]
    for tag in ("cold", "warm"):                           # cold: rank 0 computes + saves; warm: the file is loaded
        world = 2
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, bam, bw, tmp, tag)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert got[0][1] == expect and got[0][3] == ["s_cc.tab", "s_mscc.tab", "s_nreads.tab"]
        assert got[1][1] == [] and got[1][3] == []         # rank 0 writes the tables
        assert [g[2] for g in got] == ([1, 0] if tag == "cold" else [0, 0])   # computed once, on rank 0 only / loaded
        assert os.path.exists(cache) and not [f for f in os.listdir(tmp) if ".tmp." in f]
        json_ok = __import__("json").load(open(cache))
        assert set(json_ok) == {"max_shift", "__whole__", "references"}


# ---- a failing rank surfaces on every rank instead of leaving the others in the collective ---------------------------
def _failing_worker(rank, world, port, q, bam, bw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pymasc_amd import ffi
        from tests.fake_context import FakeContext

        class Broken(FakeContext):
            def cc_dev(self, *a, **k):
                raise ffi.PmxError(-2, "injected kernel failure")
        try:
            sharding.run_sharded(bam, 120, 36, 10, bigwig_path=bw, context=Broken() if rank == 1 else FakeContext())
            q.put((rank, "no error"))
        except Exception as e:
            q.put((rank, "{}: {}".format(type(e).__name__, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_failure_on_one_rank_raises_on_all(tmp_path):
    bam, bw = _write_inputs(str(tmp_path))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, bam, bw)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[1].startswith("PmxError") and "injected kernel failure" in got[1]
    assert got[0].startswith("RuntimeError") and "rank(s): 1" in got[0] and "injected kernel failure" in got[0]


# ---- tile-range sharding (round 4): equal shares of the genome's tiles, ONE all-reduce(sum) of per-chromosome rows --------
TILE = 512            # (a small tile for the CPU rehearsal: the assignment and the exchange do not depend on its size)
RLENGTHS = [9700, 4000, 7000, 300, 6500]


def _range_case(job):
    from tests import synth
    return synth.make_case(2000 + job, RLENGTHS[job], S, L, 0.03, 0.03, True, mean_on=60, mean_off=20)


def _share_of(job, first, count):
    """The share of a chromosome's result block that belongs to the tiles [first, first + count): every pair / event is owned
    by one tile -- the forward read's for ncc, mscc.fsum and mscc.ccbins, the reverse read's for mscc.rsum -- so masking the
    DRIVER vector to the range gives the share (what pmx_cc_batch_ranges_dev computes on the GPU; mappable_len and the path
    marker, which no mask expresses, are put with the share that holds tile 0: any split adds up)."""
    from oracle import model as oracle
    nbits, F, R, M = _range_case(job)
    lo, hi = first * TILE, min((first + count) * TILE, nbits)
    mask = np.zeros_like(F)
    for b in range(lo, hi):                     # (small vectors)
        mask[b >> 6] |= np.uint64(1) << np.uint64(b & 63)
    a = oracle.calc_correlation(F & mask, R, M, nbits, S, L)
    b = oracle.calc_correlation(F, R & mask, M, nbits, S, L)
    out = np.zeros((NROWS, S + 1), dtype=np.int64)
    out[0], out[1], out[3] = a["ncc_ccbins"], a["mscc_forward_sum"], a["mscc_ccbins"]
    out[2] = b["mscc_reverse_sum"]
    out[5, 0], out[5, 1] = a["ncc_forward_sum"], b["ncc_reverse_sum"]
    if first == 0:
        out[4] = a["mappable_len_by_shift"]
        out[5, 3] = 2
    return out


def _full_rows(job):
    from oracle import model as oracle
    nbits, F, R, M = _range_case(job)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out = np.zeros((NROWS, S + 1), dtype=np.int64)
    out[0], out[1], out[2], out[3], out[4] = (ref["ncc_ccbins"], ref["mscc_forward_sum"], ref["mscc_reverse_sum"], ref["mscc_ccbins"],
                                              ref["mappable_len_by_shift"])
    out[5, 0], out[5, 1], out[5, 3] = ref["ncc_forward_sum"], ref["ncc_reverse_sum"], 2
    return out


def _range_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nbits = [_range_case(j)[0] for j in range(len(RLENGTHS))]
        mine = sharding.tile_range_assign(nbits, world, TILE)[rank]
        partial = torch.zeros((len(RLENGTHS), NROWS, S + 1), dtype=torch.int64)
        for job, first, count in mine:
            partial[job] += torch.from_numpy(_share_of(job, first, count))
        q.put((rank, mine, sharding.exchange_partial_rows(partial).numpy()))
    finally:
        dist.destroy_process_group()


def test_tile_range_assignment_is_an_exact_cover_with_equal_shares():
    nbits = [248956422, 242193529, 50818468, 16569, 70000, 65536, 1]
    ntiles = [max(1, (b + 65535) // 65536) for b in nbits]
    for world in (1, 2, 3, 8, 64):
        a = sharding.tile_range_assign(nbits, world)
        seen = [np.zeros(n, dtype=np.int64) for n in ntiles]
        for r in a:
            for job, first, count in r:
                assert count >= 1
                seen[job][first:first + count] += 1
        assert all((s == 1).all() for s in seen)                    # every tile of every chromosome exactly once
        shares = [sum(c for _, _, c in r) for r in a]
        assert max(shares) - min(shares) <= 1                       # within one tile of each other
        for r in a:                                                  # a rank's stretch is contiguous: at most two partial chromosomes
            assert sum(1 for job, first, count in r if count != ntiles[job]) <= 2


@pytest.mark.timeout(300)
def test_two_rank_tile_ranges_all_reduce_to_the_whole_rows():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.stack([_full_rows(j) for j in range(len(RLENGTHS))])
    assert any(count != max(1, (_range_case(job)[0] + TILE - 1) // TILE) for _r, mine, _x in got for job, _f, count in mine)   # a chromosome IS split
    for _rank, _mine, rows in got:
        np.testing.assert_array_equal(rows, expect)
