"""GPU parity of the event kernel beyond 1023 shifts (max_shift 1024 .. 8191: 16-bit histogram cells, 1 / 2 / 4
sub-groups per workgroup, runtime halo geometry) and of the histogram flushes that keep its packed cells from
overflowing, against the CPU oracle, bit-exact (mscc.pyx:288-317 takes any max_shift)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import model as oracle
from pymasc_amd import ffi
from . import synth
from .test_gpu_batch import run_batch
from .test_gpu_parity import EVENT_TILE, check_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffi.Context(0)
    yield c
    c.set_profiling(False)
    c.debug_set_max_workgroups(0)
    c.close()


def events_ran(ctx, fn):
    """Runs fn with kernel timing on and returns (result, launches of the event kernel, launches of the window kernel)."""
    ctx.set_profiling(2)
    ctx.reset_kernel_times()
    out = fn()
    ev = ctx.kernel_time(ffi.PMX_KERNEL_CC_EVENTS)[1]
    win = ctx.kernel_time(ffi.PMX_KERNEL_CC_SPARSE)[1]
    ctx.set_profiling(False)
    return out, ev, win


@pytest.mark.parametrize("with_m", [True, False])
@pytest.mark.parametrize("S,L", [(1024, 36), (1024, 1024), (2047, 36), (2048, 100), (4095, 1), (4096, 1024), (5000, 100),
                                 (8191, 36), (8191, 1024)])
def test_shift_ranges_of_the_big_event_kernel(ctx, S, L, with_m):
    """First / last shift of every histogram geometry (2048, 4096, 8192 cells; 1, 2, 4 sub-groups), shortest and longest
    read, bits anywhere in the vector; three tiles and a bit so that sub-groups sit iterations out."""
    rng = np.random.default_rng(S * 2048 + L)
    nbits = 3 * EVENT_TILE + 4321
    F = synth.random_bits(rng, nbits, 0.005, 0, nbits)
    R = synth.random_bits(rng, nbits, 0.005, 0, nbits)
    M = synth.run_bits(rng, nbits, 900, 300, 0, nbits) if with_m else None
    for w in (F, R) + ((M,) if with_m else ()):
        synth.set_bit(w, 0)
        synth.set_bit(w, nbits - 1)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out, ev, _win = events_ran(ctx, lambda: ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE))
    check_block(out, ref, S, with_m)
    assert ev == 1, "max_shift in [1024, 8191] must run on the event kernel"


def test_first_shift_beyond_the_event_kernel_takes_the_window_kernel(ctx):
    S, L = 8192, 36
    nbits, F, R, M = synth.make_case(8192, 150000, S, L, 0.004, 0.004, True)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out, ev, win = events_ran(ctx, lambda: ctx.calc_correlation(F, R, M, nbits, S, L, 0))
    check_block(out, ref, S, True)
    assert ev == 0 and win >= 1


@pytest.mark.parametrize("skip_ncc", [False, True])
@pytest.mark.parametrize("S", [1500, 3000, 5000])
def test_dense_and_sparse_tiles_beyond_1023_shifts(ctx, S, skip_ncc):
    """Read-dense tiles in the middle, an edge-dense stretch of the track elsewhere: the event kernel flags those tiles,
    the window kernel takes them in chunks of 1024 shifts and a gated reduce adds its sums to every row."""
    L = 50
    rng = np.random.default_rng(4242 + S)
    nbits = 6 * EVENT_TILE + 12345
    F = synth.random_bits(rng, nbits, 0.004, 1, nbits - 1200)
    R = synth.random_bits(rng, nbits, 0.004, 1, nbits - 1200)
    F |= synth.random_bits(rng, nbits, 0.05, 2 * EVENT_TILE + 300, 3 * EVENT_TILE + 777)
    R |= synth.random_bits(rng, nbits, 0.05, 3 * EVENT_TILE - 5000, 4 * EVENT_TILE)
    M = synth.run_bits(rng, nbits, 2500, 700, 1, 4 * EVENT_TILE)
    M |= synth.run_bits(rng, nbits, 30, 20, 4 * EVENT_TILE + 100, 5 * EVENT_TILE)
    flags = ffi.PMX_FLAG_FORCE_SPARSE | (ffi.PMX_FLAG_SKIP_NCC if skip_ncc else 0)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    out, ev, win = events_ran(ctx, lambda: ctx.calc_correlation(F, R, M, nbits, S, L, flags))
    check_block(out, ref, S, True, skip_ncc=skip_ncc)
    assert ev == 1 and win >= 1


# One or two persistent workgroups over the whole input (pmx_debug_set_max_workgroups): every workgroup lists far more
# reads than a packed histogram cell can count, so the flushes before an overflow are what keeps the integers right.
FLUSH_CASES = [
    # S, L, chrom_len, read density, workgroups, with_m
    (300, 36, 7_000_000, 0.011, 1, True),      # max_shift <= 1023: ncc | mscc.ccbins cells of 16 bits, > 65535 forward reads
    (1500, 36, 2_000_000, 0.010, 2, True),     # one sub-group: GF / GR cells of 16 bits, |GR| <= 3 x reverse reads
    (3000, 100, 2_000_000, 0.010, 2, True),    # two sub-groups
    (5000, 100, 2_500_000, 0.008, 1, True),    # two or four sub-groups, BASELINE config 5's shift range
    (5000, 100, 2_500_000, 0.008, 1, False),   # NCC only: 32-bit cells, no flush needed (and none may hurt)
]


@pytest.mark.parametrize("case", FLUSH_CASES, ids=[f"S{c[0]}_wg{c[4]}_{'m' if c[5] else 'ncc'}" for c in FLUSH_CASES])
def test_histogram_flushes_keep_packed_cells_exact(ctx, case):
    S, L, clen, dens, nwg, with_m = case
    nbits, F, R, M = synth.make_case(S + nwg, clen, S, L, dens, dens, with_m, mean_on=2000, mean_off=500)
    ref = oracle.calc_correlation(F, R, M, nbits, S, L)
    ctx.debug_set_max_workgroups(nwg)
    try:
        out, ev, _ = events_ran(ctx, lambda: ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE))
    finally:
        ctx.debug_set_max_workgroups(0)
    check_block(out, ref, S, with_m)
    assert ev == 1


@pytest.mark.parametrize("S", [300, 2500, 5000])
def test_job_changes_inside_an_iteration_of_sub_groups(ctx, S):
    """Several chromosomes of odd tile counts in ONE workgroup's range: an iteration of 2 / 4 sub-groups ends at the end of
    its chromosome (histograms belong to a (workgroup, job) pair), sub-groups beyond it idle."""
    L = 36
    lens = [EVENT_TILE * 3 + 5000, 900, EVENT_TILE * 5 - 100, EVENT_TILE + 70, EVENT_TILE * 2 + 33000, 40000]
    cases = [synth.make_case(900 + i + S, n, S, L, 0.006, 0.006, True, mean_on=1500, mean_off=400) for i, n in enumerate(lens)]
    ctx.debug_set_max_workgroups(2)
    try:
        outs = run_batch(ctx, cases, S, L, True)
    finally:
        ctx.debug_set_max_workgroups(0)
    for (nbits, F, R, M), out in zip(cases, outs):
        check_block(out, oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


@pytest.mark.parametrize("nsg", [1, 2, 4])
def test_every_sub_group_count_gives_the_same_integers(nsg):
    """PMX_EV_NSG (read once per process) forces the number of sub-groups per workgroup: a child process per setting."""
    code = (
        "import json, numpy as np\n"
        "from pymasc_amd import ffi\n"
        "from tests import synth\n"
        "res = []\n"
        "c = ffi.Context(0)\n"
        "for S, L, n in ((1100, 36, 300000), (2500, 100, 400000)):\n"
        "    nbits, F, R, M = synth.make_case(77 + S, n, S, L, 0.007, 0.007, True, mean_on=1200, mean_off=300)\n"
        "    c.debug_set_max_workgroups(2)\n"
        "    res.append(np.asarray(c.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)).astype(np.int64).tolist())\n"
        "print(json.dumps(res))\n"
    )
    env = dict(os.environ, PMX_EV_NSG=str(nsg))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    child = json.loads(res.stdout.strip().splitlines()[-1])
    for (S, L, n), got in zip(((1100, 36, 300000), (2500, 100, 400000)), child):
        nbits, F, R, M = synth.make_case(77 + S, n, S, L, 0.007, 0.007, True, mean_on=1200, mean_off=300)
        check_block(np.asarray(got, dtype=np.int64).astype(np.uint64), oracle.calc_correlation(F, R, M, nbits, S, L), S, True)


@pytest.mark.parametrize("pattern", [0x00000000, 0xffffffff, 0x80808080, 0x7fffffff])
def test_big_event_kernel_does_not_depend_on_stale_scratch_or_lds(ctx, pattern):
    cases = [
        (1846347302, 65536, 2047, 151, 1.0, 0.0005, 30.0, 5.0, False),
        (802336588, 70001, 4000, 151, 0.02, 0.0, 30.0, 5.0, False),
        (698984044, 65536, 1033, 36, 0.0005, 0.005, 30.0, 5.0, False),
        (3, 200000, 6000, 36, 0.005, 0.005, 300.0, 80.0, False),
        (1511311730, 65536, 1512, 175, 0.0, 0.0, 300.0, 5.0, True),
    ]
    for seed, clen, S, L, fd, rd, on, off, full in cases:
        nbits, F, R, M = synth.make_case(seed, clen, S, L, fd, rd, True, mean_on=on, mean_off=off, full_range=full)
        ref = oracle.calc_correlation(F, R, M, nbits, S, L)
        ctx.debug_poison(pattern)
        out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)
        check_block(out, ref, S, True)
