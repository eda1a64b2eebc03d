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
