"""GPU: the hot path at BASELINE.json's full sizes (config 4: 24 hg38-length chromosomes, max_shift 1000; one
chromosome of config 5 at max_shift 5000), checked through properties that do not need the CPU oracle to finish:

  * selected shifts recomputed independently with plain torch integer ops on the packed words (shift, AND, byte-table
    popcount) -- every output row, several shifts including 0, read_len - 1, read_len and max_shift;
  * scalars: popcount rows equal the torch popcounts; mappable_len at lag 0 equals popcount(M);
  * the two kernel families (set-bit windows vs dense word-parallel) agree on a whole chromosome;
  * one batched launch over the genome == per-chromosome launches (bit-exact).
All integers, all exact."""
import numpy as np
import pytest
import torch

from pymasc_amd import ffi, synth

pytestmark = pytest.mark.gpu

S, L = 1000, 36
ROWS = ffi.PMX_NROWS


@pytest.fixture(scope="module")
def env():
    ctx = ffi.Context(0)
    dev = torch.device("cuda", 0)
    yield ctx, dev
    ctx.close()


_POP8 = None


def _popcount(words: torch.Tensor) -> int:
    """Population count of int64 words through a 256-entry byte table (torch has no popcount)."""
    global _POP8
    if _POP8 is None or _POP8.device != words.device:
        _POP8 = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=words.device)
    total = 0
    flat = words.view(torch.uint8)
    for lo in range(0, flat.numel(), 1 << 28):
        total += int(_POP8[flat[lo:lo + (1 << 28)].long()].sum().item())
    return total


def _shr(words: torch.Tensor, d: int) -> torch.Tensor:
    """Bit-vector >> d (bit i of the result = bit i + d of the input), zero fill."""
    q, r = divmod(d, 64)
    n = words.numel()
    out = torch.zeros_like(words)
    if q >= n:
        return out
    lo = words[q:]
    if r == 0:
        out[:n - q] = lo
        return out
    mask = (1 << (64 - r)) - 1                                  # arithmetic shift on int64: clear the sign fill
    out[:n - q] = torch.bitwise_and(torch.bitwise_right_shift(lo, r), mask)
    out[:n - q - 1] |= torch.bitwise_left_shift(lo[1:], 64 - r)
    return out


def _shl(words: torch.Tensor, d: int) -> torch.Tensor:
    """Bit-vector << d (bit i of the result = bit i - d of the input)."""
    q, r = divmod(d, 64)
    n = words.numel()
    out = torch.zeros_like(words)
    if q >= n:
        return out
    hi = words[:n - q]
    if r == 0:
        out[q:] = hi
        return out
    mask = (1 << r) - 1
    out[q:] = torch.bitwise_left_shift(hi, r)
    out[q + 1:] |= torch.bitwise_and(torch.bitwise_right_shift(hi[:-1], 64 - r), mask)
    return out


def _torch_reference(v, d: int, read_len: int = L):
    """mscc.pyx:288-317 for ONE shift with whole-vector torch ops: F[j] & R[j+d] & D_d[j], D_d[j] = M[j] & M[j+c-d]."""
    c = read_len - 1
    F, R, M = v.F, v.R, v.M
    Rd = _shr(R, d)
    if M is None:
        return {"ncc": _popcount(F & Rd)}
    k = c - d
    Mk = _shr(M, k) if k >= 0 else _shl(M, -k)
    D = M & Mk
    return {
        "ncc": _popcount(F & Rd),
        "fsum": _popcount(F & D),
        "rsum": _popcount(Rd & D),
        "cc": _popcount(F & Rd & D),
        "mlen": _popcount(D),
    }


def _run_batch(ctx, dev, vecs, max_shift, flags=0, read_len=L):
    out = torch.full((len(vecs), ROWS, max_shift + 1), -1, dtype=torch.int64, device=dev)   # garbage: must be overwritten
    with_m = vecs[0].M is not None
    ctx.debug_poison(0xffffffff)   # stale scratch buffers / LDS must not matter (0xffffffff: ones as bits, -1 / huge as numbers)
    ctx.cc_batch_dev([v.F.data_ptr() for v in vecs], [v.R.data_ptr() for v in vecs],
                     [v.M.data_ptr() for v in vecs] if with_m else None,
                     [v.nbits for v in vecs], max_shift, read_len, flags, [out[i].data_ptr() for i in range(len(vecs))])
    ctx.sync()
    return out.cpu().numpy()


@pytest.fixture(scope="module")
def genome(env):
    ctx, dev = env
    vecs = synth.make_genome(ctx, dev, synth.HG38, S, L)
    rows = _run_batch(ctx, dev, vecs, S)
    return vecs, rows


def test_full_genome_selected_shifts_against_torch(env, genome):
    vecs, rows = genome
    assert sum(v.length for v in vecs) == sum(l for _, l in synth.HG38)      # 3.088 Gbp
    for i in (0, 7, 20, 23):                                                    # chr1, chr8, chr21, chrY
        v = vecs[i]
        for d in (0, 1, L - 1, L, 517, S):
            ref = _torch_reference(v, d)
            got = {"ncc": rows[i, ffi.PMX_ROW_NCC_CCBINS, d], "fsum": rows[i, ffi.PMX_ROW_MSCC_FSUM, d],
                   "rsum": rows[i, ffi.PMX_ROW_MSCC_RSUM, d], "cc": rows[i, ffi.PMX_ROW_MSCC_CCBINS, d],
                   "mlen": rows[i, ffi.PMX_ROW_MLEN, d]}
            assert {k: int(x) for k, x in got.items()} == ref, (v.name, d)


def test_full_genome_scalars_and_bounds(env, genome):
    vecs, rows = genome
    for i, v in enumerate(vecs):
        sc = rows[i, ffi.PMX_ROW_SCALARS]
        assert int(sc[3]) == ffi.PMX_PATH_SPARSE
        if i in (2, 11, 22):
            assert (int(sc[0]), int(sc[1]), int(sc[2])) == (_popcount(v.F), _popcount(v.R), _popcount(v.M))
        assert 0 < sc[0] <= v.n_forward and 0 < sc[1] <= v.n_reverse           # duplicates collapse into one bit
        mlen = rows[i, ffi.PMX_ROW_MLEN]
        assert int(mlen[L - 1]) == int(sc[2])                                   # lag 0 <-> shift read_len - 1
        assert (mlen <= sc[2]).all()
        # masked counts never exceed the unmasked ones, nor the strand totals
        assert (rows[i, ffi.PMX_ROW_MSCC_CCBINS] <= rows[i, ffi.PMX_ROW_NCC_CCBINS]).all()
        assert (rows[i, ffi.PMX_ROW_MSCC_FSUM] <= sc[0]).all() and (rows[i, ffi.PMX_ROW_MSCC_RSUM] <= sc[1]).all()
        # the planted fragment peak at +180 (synth.make_chromosome) dominates the naive curve
        assert int(np.argmax(rows[i, ffi.PMX_ROW_NCC_CCBINS])) == 180


def test_full_chromosome_dense_and_sparse_kernels_agree(env, genome):
    ctx, dev = env
    vecs, rows = genome
    for i in (0, 18):                                                           # chr1 (longest), chr19
        dense = _run_batch(ctx, dev, [vecs[i]], S, ffi.PMX_FLAG_FORCE_DENSE)[0]
        assert int(dense[ffi.PMX_ROW_SCALARS, 3]) == ffi.PMX_PATH_DENSE
        for r in range(5):
            np.testing.assert_array_equal(dense[r], rows[i, r])
        np.testing.assert_array_equal(dense[ffi.PMX_ROW_SCALARS, :3], rows[i, ffi.PMX_ROW_SCALARS, :3])


def test_batched_launch_equals_single_launches(env, genome):
    ctx, dev = env
    vecs, rows = genome
    for i in (1, 12, 23):
        single = _run_batch(ctx, dev, [vecs[i]], S)[0]
        np.testing.assert_array_equal(single, rows[i])


def test_stress_chromosome_max_shift_5000(env):
    """Config 5 shape: one 2.5e8-bp chromosome at max_shift 5000 (five 1024-shift chunks)."""
    ctx, dev = env
    S5 = 5000
    name, length = max(synth.stress_genome(), key=lambda t: t[1])
    v = synth.make_chromosome(ctx, dev, name, length, S5, L, 0xBADC0DE)
    rows = _run_batch(ctx, dev, [v], S5)[0]
    for d in (0, 1023, 1024, 2047, 2048, 3333, 4999, 5000):                     # chunk edges included
        ref = _torch_reference(v, d)
        got = {"ncc": rows[ffi.PMX_ROW_NCC_CCBINS, d], "fsum": rows[ffi.PMX_ROW_MSCC_FSUM, d],
               "rsum": rows[ffi.PMX_ROW_MSCC_RSUM, d], "cc": rows[ffi.PMX_ROW_MSCC_CCBINS, d],
               "mlen": rows[ffi.PMX_ROW_MLEN, d]}
        assert {k: int(x) for k, x in got.items()} == ref, d


def test_full_genome_ncc_only(env):
    """BASELINE config 2's shape: naive CC only over the whole hg38-sized genome, max_shift 1000 -- the NCC-only
    instantiation of the set-bit kernel (one counter, 6 waves per SIMD, popcount(R) counted per thread)."""
    ctx, dev = env
    vecs = synth.make_genome(ctx, dev, synth.HG38, S, L, with_m=False)
    rows = _run_batch(ctx, dev, vecs, S)
    for i, v in enumerate(vecs):
        sc = rows[i, ffi.PMX_ROW_SCALARS]
        assert int(sc[3]) == ffi.PMX_PATH_SPARSE and int(sc[2]) == 0
        assert not rows[i, ffi.PMX_ROW_MSCC_FSUM:ffi.PMX_ROW_MLEN + 1].any()
        assert 0 < sc[0] <= v.n_forward and 0 < sc[1] <= v.n_reverse
        assert (rows[i, ffi.PMX_ROW_NCC_CCBINS] <= min(int(sc[0]), int(sc[1]))).all()
        assert int(np.argmax(rows[i, ffi.PMX_ROW_NCC_CCBINS])) == 180
    for i in (0, 5, 16, 21, 23):                                                # chr1, chr6, chr17, chr22, chrY
        v = vecs[i]
        assert (int(rows[i, ffi.PMX_ROW_SCALARS, 0]), int(rows[i, ffi.PMX_ROW_SCALARS, 1])) == (_popcount(v.F), _popcount(v.R))
        for d in (0, 1, 31, 32, 180, 999, S):
            assert int(rows[i, ffi.PMX_ROW_NCC_CCBINS, d]) == _torch_reference(v, d)["ncc"], (v.name, d)
    # the dense kernels agree on one whole chromosome, and a single launch equals its slot of the batch
    dense = _run_batch(ctx, dev, [vecs[20]], S, ffi.PMX_FLAG_FORCE_DENSE)[0]
    np.testing.assert_array_equal(dense[ffi.PMX_ROW_NCC_CCBINS], rows[20, ffi.PMX_ROW_NCC_CCBINS])
    np.testing.assert_array_equal(_run_batch(ctx, dev, [vecs[7]], S)[0], rows[7])


@pytest.mark.parametrize("with_m", [True, False])
def test_stress_genome_as_specified(env, with_m):
    """BASELINE config 5 as specified: 10 Gbp, 200 chromosomes, max_shift 5000, read_len 100 -- 1000 (chromosome x
    1024-shift chunk) jobs, i.e. 7 batches of <= 32 chromosomes x 5 launches that reuse one slab.  Chunk-edge shifts
    are recomputed with torch on chromosomes of different batches; every row is checked for the invariants that do not
    need a reference."""
    ctx, dev = env
    S5, L5 = 5000, 100
    chroms = synth.stress_genome()
    assert len(chroms) == 200 and abs(sum(l for _, l in chroms) - 1e10) < 1e6
    vecs = synth.make_genome(ctx, dev, chroms, S5, L5, seed_base=0xBADC0DE, with_m=with_m)
    rows = _run_batch(ctx, dev, vecs, S5, read_len=L5)
    assert (rows >= 0).all()                                                    # every word of every block was written
    for i, v in enumerate(vecs):
        sc = rows[i, ffi.PMX_ROW_SCALARS]
        assert int(sc[3]) == ffi.PMX_PATH_SPARSE
        assert 0 < sc[0] <= v.n_forward and 0 < sc[1] <= v.n_reverse
        assert int(np.argmax(rows[i, ffi.PMX_ROW_NCC_CCBINS])) == 180          # the planted fragment peak
        if with_m:
            mlen = rows[i, ffi.PMX_ROW_MLEN]
            assert int(mlen[L5 - 1]) == int(sc[2]) and (mlen <= sc[2]).all()
            assert (rows[i, ffi.PMX_ROW_MSCC_CCBINS] <= rows[i, ffi.PMX_ROW_NCC_CCBINS]).all()
            assert (rows[i, ffi.PMX_ROW_MSCC_FSUM] <= sc[0]).all() and (rows[i, ffi.PMX_ROW_MSCC_RSUM] <= sc[1]).all()
        else:
            assert not rows[i, ffi.PMX_ROW_MSCC_FSUM:ffi.PMX_ROW_MLEN + 1].any()
    longest = max(range(200), key=lambda i: vecs[i].length)
    shortest = min(range(200), key=lambda i: vecs[i].length)
    for i in sorted({3, 40, 77, 150, 199, longest, shortest}):                  # launches of five different batches
        v = vecs[i]
        for d in (0, L5 - 1, L5, 1023, 1024, 2047, 2048, 4095, 4096, 4999, S5):
            ref = _torch_reference(v, d, L5)
            got = {"ncc": rows[i, ffi.PMX_ROW_NCC_CCBINS, d]}
            if with_m:
                got.update(fsum=rows[i, ffi.PMX_ROW_MSCC_FSUM, d], rsum=rows[i, ffi.PMX_ROW_MSCC_RSUM, d],
                           cc=rows[i, ffi.PMX_ROW_MSCC_CCBINS, d], mlen=rows[i, ffi.PMX_ROW_MLEN, d])
            assert {k: int(x) for k, x in got.items()} == ref, (v.name, d)
    # a chromosome alone == its slot of the 200-chromosome batch (job table reuse across launches)
    for i in (31, 32, 199):
        np.testing.assert_array_equal(_run_batch(ctx, dev, [vecs[i]], S5, read_len=L5)[0], rows[i])


def _pile_up(ctx, dev, v, lo, hi, density, seed):
    """More reads on both strands inside [lo, hi): a local pile-up on an otherwise ordinary chromosome."""
    g = torch.Generator(device=dev).manual_seed(seed)
    n = int((hi - lo) * density)
    for vec in (v.F, v.R):
        pos = torch.randint(lo, hi, (n,), generator=g, device=dev, dtype=torch.int64)
        torch.cuda.current_stream(dev).synchronize()
        ctx.bits_set_positions_dev(vec.data_ptr(), v.nbits, pos.data_ptr(), pos.numel())
    ctx.sync()


@pytest.mark.parametrize("with_m", [True, False])
def test_deep_and_marginal_data_hand_over_to_the_window_kernels(env, with_m):
    """Chromosomes of several tiles per workgroup at read densities around and far above the event kernel's list
    capacities (~1 % per strand): a deep one (every tile far above: the workgroups hand their whole ranges over after two
    tiles), a marginal one (tiles flagged one by one, the window kernels skip the others), an ordinary one with a 3-Mbp
    pile-up in the middle, an ordinary one.  The event path must give the integers of the window kernels alone
    (PMX_FLAG_WINDOW_ONLY), and both must match torch at selected shifts."""
    ctx, dev = env
    shape = [("deep", 90_000_000, 0.02), ("marginal", 70_000_000, 0.0118), ("pileup", 80_000_000, 0.004), ("plain", 40_000_000, 0.005)]
    vecs = [synth.make_chromosome(ctx, dev, n, ln, S, L, 0xD00D + i, density=rho, with_m=with_m)
            for i, (n, ln, rho) in enumerate(shape)]
    _pile_up(ctx, dev, vecs[2], 30_000_000, 33_000_000, 0.03, 77)
    ev = _run_batch(ctx, dev, vecs, S)
    win = _run_batch(ctx, dev, vecs, S, ffi.PMX_FLAG_WINDOW_ONLY)
    np.testing.assert_array_equal(ev, win)
    for i, v in enumerate(vecs):
        for d in (0, 35, 36, 500, S):
            ref = _torch_reference(v, d) if with_m else {"ncc": _torch_reference(v, d)["ncc"]}
            assert int(ev[i, ffi.PMX_ROW_NCC_CCBINS, d]) == ref["ncc"], (v.name, d)
            if with_m:
                assert int(ev[i, ffi.PMX_ROW_MSCC_FSUM, d]) == ref["fsum"], (v.name, d)
                assert int(ev[i, ffi.PMX_ROW_MSCC_RSUM, d]) == ref["rsum"], (v.name, d)
                assert int(ev[i, ffi.PMX_ROW_MSCC_CCBINS, d]) == ref["cc"], (v.name, d)
                assert int(ev[i, ffi.PMX_ROW_MLEN, d]) == ref["mlen"], (v.name, d)
