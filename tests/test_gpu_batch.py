"""GPU parity of the batched entry point (pmx_cc_batch_dev): several chromosomes in one pass of the kernels,
job boundaries falling inside a workgroup's tile range, more jobs than one launch's job table holds."""
import numpy as np
import pytest
import torch

from oracle import model as oracle
from pymasc_amd import ffi
from . import synth
from .test_gpu_parity import check_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.close()


def run_batch(ctx, cases, S, L, with_m, flags=0):
    dev = torch.device("cuda", 0)
    tens, pF, pR, pM, pN, pO, outs = [], [], [], [], [], [], []
    for nbits, F, R, M in cases:
        tF = torch.from_numpy(F.view(np.int64)).to(dev)
        tR = torch.from_numpy(R.view(np.int64)).to(dev)
        tM = torch.from_numpy(M.view(np.int64)).to(dev) if with_m else None
        tO = torch.full((ffi.PMX_NROWS, S + 1), -1, dtype=torch.int64, device=dev)   # garbage: must be overwritten
        tens += [tF, tR, tM, tO]
        pF.append(tF.data_ptr())
        pR.append(tR.data_ptr())
        if with_m:
            pM.append(tM.data_ptr())
        pN.append(nbits)
        pO.append(tO.data_ptr())
        outs.append(tO)
    torch.cuda.synchronize()
    ctx.cc_batch_dev(pF, pR, pM if with_m else None, pN, S, L, flags, pO)
    ctx.sync()
    return [o.cpu().numpy().view(np.uint64) for o in outs]


@pytest.mark.parametrize("with_m", [True, False])
def test_small_jobs_of_mixed_sizes(ctx, with_m):
    S, L = 300, 36
    lens = [700, 40000, 32768 * 3 + 5, 70000, 1500, 32768 - S - L - 100, 200000]
    cases = [synth.make_case(100 + i, n, S, L, 0.01, 0.012, with_m) for i, n in enumerate(lens)]
    outs = run_batch(ctx, cases, S, L, with_m)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, with_m)
        assert int(out[ffi.PMX_ROW_SCALARS, 3]) == ffi.PMX_PATH_SPARSE
        if not with_m:
            assert not out[ffi.PMX_ROW_MSCC_FSUM:ffi.PMX_ROW_MLEN + 1].any()


def test_job_boundaries_inside_workgroup_ranges(ctx):
    # ~1500 tiles > resident workgroups, so every workgroup walks several tiles and some ranges span two jobs
    S, L = 100, 36
    lens = [20_000_000, 9_000_001, 20_500_000]
    cases = [synth.make_case(200 + i, n, S, L, 0.004, 0.004, True, mean_on=2000, mean_off=500) for i, n in enumerate(lens)]
    outs = run_batch(ctx, cases, S, L, True)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


def test_job_boundaries_inside_workgroup_ranges_ncc_only(ctx):
    # the NCC-only instantiation (one counter, 6 waves per SIMD, its own popcount(R) path) over ranges that span jobs
    S, L = 1000, 36
    lens = [12_000_000, 5_000_001, 12_500_000]
    cases = [synth.make_case(400 + i, n, S, L, 0.004, 0.004, False, full_range=(i == 1)) for i, n in enumerate(lens)]
    outs = run_batch(ctx, cases, S, L, False)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, None, nbits, S, L), S, False)
        assert int(out[ffi.PMX_ROW_SCALARS, 3]) == ffi.PMX_PATH_SPARSE


def test_more_jobs_than_one_job_table(ctx):
    S, L = 64, 20
    cases = [synth.make_case(300 + i, 3000 + 517 * i, S, L, 0.02, 0.02, True) for i in range(45)]
    outs = run_batch(ctx, cases, S, L, True)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


@pytest.mark.parametrize("S", [300, 1500])
def test_dense_tiles_on_both_sides_of_a_job_table(ctx, S):
    """40 chromosomes = two launches of the max_shift <= 1023 kernels (one beyond): read-dense and edge-dense stretches in
    chromosomes of BOTH launches, so that the flagged-tile counters, the job statistics and the work split of the window
    kernels (k_plan_flagged: it scans one flag array shared by the launches) are per launch."""
    L = 36
    rng = np.random.default_rng(4040 + S)
    cases = []
    for i in range(40):
        n = 140000 + 7919 * i if i in (5, 31, 33, 39) else 4000 + 613 * i
        nbits, F, R, M = synth.make_case(500 + i, n, S, L, 0.006, 0.006, True, mean_on=1500, mean_off=400)
        if i in (5, 33):        # a read-dense tile in the middle
            F |= synth.random_bits(rng, nbits, 0.06, 70000, 125000)
            R |= synth.random_bits(rng, nbits, 0.06, 66000, 120000)
        if i in (31, 39):       # a stretch of very short runs
            keep = synth.run_bits(np.random.default_rng(1), nbits, 10**9, 1, 0, 66000)
            M = (M & keep) | synth.run_bits(rng, nbits, 6, 4, 66000, nbits)
        cases.append((nbits, F, R, M))
    outs = run_batch(ctx, cases, S, L, True)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


def test_dense_tiles_force_counter_spills(ctx):
    # every bit set: thousands of records per tile -> multi-round lists and mid-tile counter folds
    S, L = 200, 36
    G = 150000
    nbits = G + L + S + 100
    nw = synth.nwords(nbits)
    rng = np.random.default_rng(9)
    F = synth.random_bits(rng, nbits, 0.9, 1, G + 1)
    R = synth.random_bits(rng, nbits, 0.9, 1, G + L)
    M = synth.random_bits(rng, nbits, 0.7, 1, G + 1)
    out = run_batch(ctx, [(nbits, F, R, M)], S, L, True, ffi.PMX_FLAG_FORCE_SPARSE)[0]
    check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


def test_mixed_mappability_is_rejected(ctx):
    S, L = 64, 20
    a = synth.make_case(1, 5000, S, L)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (a[1], a[2], a[3])]
    o = torch.zeros((ffi.PMX_NROWS, S + 1), dtype=torch.int64, device=dev)
    with pytest.raises(ffi.PmxError):
        ctx.cc_batch_dev([t[0].data_ptr()] * 2, [t[1].data_ptr()] * 2, [t[2].data_ptr(), 0], [a[0]] * 2, S, L, 0,
                         [o.data_ptr()] * 2)


@pytest.mark.parametrize("S", [300, 1100])
def test_back_to_back_batches_whose_job_table_grows(ctx, S):
    """Two pmx_cc_batch_dev calls queued WITHOUT a synchronisation between them, the second with far more jobs than the
    first: beyond 1023 shifts the device-side job table of the stream has to be re-allocated for it while the first call's
    kernels may still be queued (pmx_upload_jobtab: the one device table per stream is overwritten by every upload, its
    page-locked staging slots go round a ring), below they travel in the kernel arguments (two launches for the second
    call).  Every scratch buffer poisoned first.  (the test an unexplained abort under a stream sync asked for.)"""
    L = 36
    small = [synth.make_case(700 + i, 2500 + 811 * i, S, L, 0.01, 0.01, True) for i in range(3)]
    large = [synth.make_case(720 + i, 1800 + 97 * i, S, L, 0.01, 0.01, True) for i in range(130)]
    dev = torch.device("cuda", 0)
    ctx.debug_poison(0xffffffff)

    def queue(cases):
        keep, args, outs = [], ([], [], [], []), []
        for nbits, F, R, M in cases:
            t = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (F, R, M)]
            o = torch.full((ffi.PMX_NROWS, S + 1), -1, dtype=torch.int64, device=dev)
            keep += t
            for a, x in zip(args, t + [o]):
                a.append(x.data_ptr())
            outs.append(o)
        return keep, args, outs, [c[0] for c in cases]

    k1, a1, o1, n1 = queue(small)
    k2, a2, o2, n2 = queue(large)
    torch.cuda.synchronize()
    ctx.cc_batch_dev(a1[0], a1[1], a1[2], n1, S, L, 0, a1[3])
    ctx.cc_batch_dev(a2[0], a2[1], a2[2], n2, S, L, 0, a2[3])     # (no sync in between)
    ctx.cc_batch_dev(a1[0], a1[1], a1[2], n1, S, L, 0, a1[3])     # and the small table again behind the large one
    ctx.sync()
    for cases, outs in ((small, o1), (large, o2)):
        for (nbits, F, R, M), o in zip(cases, outs):
            check_block(o.cpu().numpy().view(np.uint64), oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


@pytest.mark.parametrize("density,expect_events", [(0.004, True), (0.06, False)])
def test_a_call_without_a_hint_takes_one_from_a_sample_of_the_vectors(ctx, density, expect_events):
    """flags = 0: pmx_cc_batch_dev counts 16 sampled tiles per chromosome (k_density_probe) and gives itself the hint a caller
    with counts would give -- sparse data stays on the event kernel, data far beyond its lists goes straight to the window
    kernels (no event pass in front of them: the 5-15 % of round 3).  Same integers as with PMX_FLAG_FORCE_SPARSE (no probe)."""
    S, L = 300, 36
    lens = [300000, 120000, 70000]
    cases = [synth.make_case(900 + i, n, S, L, density, density, True, mean_on=2000, mean_off=500) for i, n in enumerate(lens)]
    ctx.set_profiling(2)
    ctx.reset_kernel_times()
    outs = run_batch(ctx, cases, S, L, True, 0)
    ev_launches = ctx.kernel_time(ffi.PMX_KERNEL_CC_EVENTS)[1]
    ctx.set_profiling(False)
    assert (ev_launches > 0) == expect_events
    forced = run_batch(ctx, cases, S, L, True, ffi.PMX_FLAG_FORCE_SPARSE)
    for (nbits, F, R, M), out, f in zip(cases, outs, forced):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)
        assert np.array_equal(out[:ffi.PMX_ROW_SCALARS], f[:ffi.PMX_ROW_SCALARS])


@pytest.mark.parametrize("with_m,skip_ncc", [(True, False), (False, False), (True, True)])
def test_tile_ranges_of_a_chromosome_add_up_to_it(ctx, with_m, skip_ncc):
    """pmx_cc_batch_ranges_dev: every sum of the hot path is owned by one 64-Kbit tile, so the result blocks of jobs that cover
    a chromosome between them add up to the block pmx_cc_batch_dev writes for it -- rows, popcounts, mappable lengths,
    path marker (what ranks of a multi-GPU run all-reduce).  Cuts anywhere (tile 1, the last tile, one-tile ranges), a
    read-dense and an edge-dense stretch inside ranges (those tiles go to the window kernels), a one-tile chromosome."""
    S, L = 300, 36
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(61)
    cases = []
    for i, n in enumerate([400000, 65536 * 3 - 436, 30000]):
        nbits, F, R, M = synth.make_case(950 + i, n, S, L, 0.006, 0.006, True, mean_on=1500, mean_off=400)
        if i == 0:
            F |= synth.random_bits(rng, nbits, 0.08, 140000, 200000)           # read-dense tiles
            R |= synth.random_bits(rng, nbits, 0.08, 138000, 198000)
            keep = synth.run_bits(np.random.default_rng(1), nbits, 10**9, 1, 0, 300000)
            M = (M & keep) | synth.run_bits(rng, nbits, 6, 4, 300000, 340000)   # very short runs: edge-dense tiles
        cases.append((nbits, F, R, M))
    flags = ffi.PMX_FLAG_SKIP_NCC if skip_ncc else 0
    full = run_batch(ctx, cases, S, L, with_m, flags | ffi.PMX_FLAG_EVENTS_HINT)
    cuts = {0: [0, 1, 2, 5, 7], 1: [0, 2, 3], 2: [0, 1]}      # tile boundaries of the ranges per chromosome
    keep, aF, aR, aM, aN, aO, first, count, outs, owner = [], [], [], [], [], [], [], [], [], []
    for j, (nbits, F, R, M) in enumerate(cases):
        ntiles = (nbits + ffi.RANGE_TILE_BITS - 1) // ffi.RANGE_TILE_BITS
        assert cuts[j][-1] == ntiles, (j, ntiles)
        t = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (F, R, M)]
        keep += t
        for a, b in zip(cuts[j][:-1], cuts[j][1:]):
            o = torch.full((ffi.PMX_NROWS, S + 1), -1, dtype=torch.int64, device=dev)
            aF.append(t[0].data_ptr()); aR.append(t[1].data_ptr()); aM.append(t[2].data_ptr()); aN.append(nbits)
            aO.append(o.data_ptr()); first.append(a); count.append(b - a); outs.append(o); owner.append(j)
    torch.cuda.synchronize()
    ctx.cc_batch_ranges_dev(aF, aR, aM if with_m else None, aN, first, count, S, L, flags, aO)
    ctx.sync()
    for j, (nbits, F, R, M) in enumerate(cases):
        total = sum(o.cpu().numpy().view(np.uint64) for o, w in zip(outs, owner) if w == j)
        assert np.array_equal(total, full[j]), j
        check_block(total, oracle.calc_correlation(F, R, M if with_m else None, nbits, S, L, skip_ncc=skip_ncc), S, with_m, skip_ncc)
    with pytest.raises(ffi.PmxError):          # a range beyond the chromosome's tiles
        ctx.cc_batch_ranges_dev(aF[:1], aR[:1], aM[:1] if with_m else None, aN[:1], [6], [3], S, L, flags, aO[:1])
    with pytest.raises(ffi.PmxError):          # beyond 1023 shifts the pair pass over M is not range-aware
        ctx.cc_batch_ranges_dev(aF[:1], aR[:1], aM[:1] if with_m else None, aN[:1], [0], [1], 2000, L, flags, aO[:1])
