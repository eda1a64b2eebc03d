#!/usr/bin/env python3
"""GPU box: randomized parity sweep of the C-ABI hot path against the CPU oracle (test infrastructure).

Draws chromosome length, max_shift, read_len, read densities, mappability run statistics and the kernel-path flag at
random, runs pmx_calc_correlation / pmx_cc_batch_dev / pmx_mappable_len, and compares every output word with
oracle/cc_oracle.c.  Sizes are kept small enough for the oracle to keep up; the point is coverage of odd geometry
(tile edges, shift-chunk edges, S vs L vs chromosome length, empty / saturated vectors), not throughput.

    python tools/fuzz_parity.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import model as oracle  # noqa: E402
from pymasc_amd import ffi  # noqa: E402
from tests import synth  # noqa: E402


def draw_case(rng):
    kind = rng.integers(0, 10)
    S = int(rng.choice([3, 4, 31, 32, 33, 63, 64, 100, 300, 511, 512, 1000, 1023, 1024, 1025, 2047, 2048, 3000, 5000]))
    if kind == 0:
        S = int(rng.integers(3, 6000))
    if kind == 3:
        S = int(rng.integers(1024, 8193))       # the event kernel's range beyond 1023 shifts, and the first shift past it
    L = int(rng.choice([1, 2, 20, 36, 50, 100, 151, 250, 1000, 1024, 1025, 1500]))
    if kind == 1:
        L = int(rng.integers(1, 1025))
    clen = int(rng.choice([1, 40, 900, 5000, 32768 - 200, 32768, 32768 + 7, 65536, 70001, 131072 + 31, 200003]))
    if kind == 2:
        clen = int(rng.integers(1, 250000))
    fd = float(rng.choice([0.0, 0.0005, 0.005, 0.012, 0.02, 0.03, 0.1, 0.5, 1.0]))
    rd = float(rng.choice([0.0, 0.0005, 0.005, 0.012, 0.02, 0.03, 0.1, 0.5, 1.0]))
    with_m = bool(rng.random() < 0.75)
    mean_on = float(rng.choice([1.5, 5, 30, 300, 3000, 1e6]))
    mean_off = float(rng.choice([1.5, 5, 80, 500, 1e6]))
    return S, L, clen, fd, rd, with_m, mean_on, mean_off


def draw_full_range(rng):
    """Half of the cases put bits anywhere in [0, nbits) (tests/synth.py:make_case)."""
    return bool(rng.random() < 0.5)


def compare(out, ref, S, with_m, skip_ncc, tag):
    ok = True
    def chk(name, a, b):
        nonlocal ok
        if not np.array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)):
            bad = np.flatnonzero(np.asarray(a).astype(np.int64) != np.asarray(b).astype(np.int64))
            print("MISMATCH", tag, name, "first bad shift", bad[:5], "of", bad.size, flush=True)
            ok = False
    if not skip_ncc:
        chk("ncc", out[ffi.PMX_ROW_NCC_CCBINS, :S + 1], ref["ncc_ccbins"])
        chk("popF", out[ffi.PMX_ROW_SCALARS, 0], ref["ncc_forward_sum"])
        chk("popR", out[ffi.PMX_ROW_SCALARS, 1], ref["ncc_reverse_sum"])
    if with_m:
        chk("fsum", out[ffi.PMX_ROW_MSCC_FSUM], ref["mscc_forward_sum"])
        chk("rsum", out[ffi.PMX_ROW_MSCC_RSUM], ref["mscc_reverse_sum"])
        chk("cc", out[ffi.PMX_ROW_MSCC_CCBINS], ref["mscc_ccbins"])
        chk("mlen", out[ffi.PMX_ROW_MLEN], ref["mappable_len_by_shift"])
    return ok


def fuzz_feed(ctx, rng, clen, S, L):
    """pmx_feed_reads against the oracle's per-read restatement of feed_forward_read / feed_reverse_read (mscc.pyx:351-418):
    random reads with duplicates on both strands and variable lengths, fed in random chunks, every width combination."""
    n = int(rng.integers(1, 4000))
    glen = max(int(clen), 50)
    pos = np.sort(rng.integers(0, glen, size=n)).astype(np.int64)
    if rng.random() < 0.5:
        pos = np.sort(rng.choice(pos, size=n))           # many equal positions
    rlen = rng.choice(np.asarray([1, 20, 36, 36, 50, 101]), size=n).astype(np.int64)
    rev = rng.random(n) < rng.random()
    oc = oracle.OracleCalculator(S, L, ["c"], [glen])
    for p, l, r in zip(pos.tolist(), rlen.tolist(), rev.tolist()):
        (oc.feed_reverse_read if r else oc.feed_forward_read)("c", p, l)
    nbits = oc._nbits
    d_F, d_R, d_st = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    cuts = sorted(set(rng.integers(1, n, size=int(rng.integers(0, 5))).tolist())) if n > 1 else []
    pdt = [np.int32, np.int64][int(rng.integers(0, 2))]
    ldt = [np.uint16, np.int32, np.int64][int(rng.integers(0, 3))]
    packed = rng.random() < 0.3
    delta16 = rng.random() < 0.35 and nbits < 2**31          # round 4: the two-bytes-per-read form
    whole = rng.random() < 0.5                               # round 4: the first run writes every word of uncleared vectors
    if whole:
        junk = np.full(ffi.nwords(nbits), 0x5a5a5a5aa5a5a5a5, dtype=np.uint64)
        ctx.bits_upload(d_F, junk, nbits)
        ctx.bits_upload(d_R, ~junk, nbits)
    keep, fed = [], 0
    for a, b in zip([0] + cuts, cuts + [n]):
        if b > a:
            pp = pos[a:b].astype(pdt)
            w = whole and fed == 0
            if delta16:
                keep.append(ctx.feed_reads_delta16(d_F, d_R, nbits, ffi.pack_delta16(pos[a:b], rev[a:b]), rlen[a:b].astype(ldt), fed, d_st,
                                                   whole_vectors=w))
            else:
                keep.append(ctx.feed_reads(d_F, d_R, nbits, ffi.pack_strand(pp, rev[a:b]) if packed else pp, rlen[a:b].astype(ldt),
                                           None if packed else rev[a:b], fed, d_st, whole_vectors=w))
            fed += b - a
    F, R = ctx.bits_download(d_F, nbits), ctx.bits_download(d_R, nbits)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    for d in (d_F, d_R, d_st):
        ctx.bits_free(d)
    ok = (np.array_equal(F, oc._F) and np.array_equal(R, oc._R) and int(st[ffi.PMX_FEED_FORWARD_LEN_SUM]) == oc._f_rls
          and int(st[ffi.PMX_FEED_REVERSE_LEN_SUM]) == oc._r_rls and int(st[ffi.PMX_FEED_FIRST_UNSORTED]) == 0
          and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0)
    if not ok:
        print("MISMATCH feed_reads", n, glen, S, L, cuts, pdt, ldt, packed, delta16, whole, flush=True)
    return 0 if ok else 1


def fuzz_regions(ctx, rng, M, nbits):
    """pmx_bits_set_regions_ex (round 4): the case's mappability vector rebuilt from its own runs of ones, as BigWig (begin, end]
    pairs in order -- PMX_REGIONS_SORTED over a vector full of somebody else's bits, on the side stream or not -- and, now and
    then, the same intervals in disorder through the general setter (clear first) and through the builder (must be flagged)."""
    bits = np.unpackbits(M.view(np.uint8), bitorder="little")[:nbits].astype(np.int8)
    edges = np.diff(np.concatenate([[0], bits, [0]]))
    first = np.flatnonzero(edges == 1)            # first set bit of a run
    last = np.flatnonzero(edges == -1) - 1        # its last set bit
    begin, end = first - 1, last                  # set(begin + 1, end)
    dt = [np.uint32, np.int64][int(rng.integers(0, 2))]
    off = 1
    if begin.size and begin[0] < 0:
        if dt is np.uint32:
            dt = np.int64
    d = ctx.bits_alloc(nbits)
    d_st = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
    ctx.bits_upload(d, np.full(ffi.nwords(nbits), 0xa5a5a5a55a5a5a5a, dtype=np.uint64), nbits)
    side = bool(rng.random() < 0.5)
    keep = [ctx.bits_set_regions_async(d, nbits, begin.astype(dt), end.astype(dt), off, d_st, side=side, sorted_disjoint=True)]
    got = ctx.bits_download(d, nbits)
    st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
    ok = np.array_equal(got, M[:ffi.nwords(nbits)]) and int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) == 0 and int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]) == 0
    if ok and begin.size > 2 and rng.random() < 0.3:
        perm = rng.permutation(begin.size)
        if not np.array_equal(perm, np.arange(begin.size)):
            keep.append(ctx.bits_set_regions_async(d, nbits, begin[perm].astype(dt), end[perm].astype(dt), off, d_st, clear=True, side=side))
            ok = np.array_equal(ctx.bits_download(d, nbits), M[:ffi.nwords(nbits)])
            keep.append(ctx.bits_set_regions_async(d, nbits, begin[perm].astype(dt), end[perm].astype(dt), off, d_st, sorted_disjoint=True))
            st = ctx.bits_download(d_st, ffi.PMX_FEED_WORDS * 64)
            ok = ok and int(st[ffi.PMX_FEED_REGIONS_UNSORTED]) != 0
    ctx.bits_free(d)
    ctx.bits_free(d_st)
    if not ok:
        print("MISMATCH set_regions_ex", nbits, begin.size, dt, side, flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    poison_rng = np.random.default_rng(a.seed + 12345)   # separate stream: the case sequence of a seed stays what it was
    oracle.lib()
    t_end = time.time() + a.seconds
    n = bad = 0
    last = time.time()
    with ffi.Context(0) as ctx:
        while time.time() < t_end:
            S, L, clen, fd, rd, with_m, mean_on, mean_off = draw_case(rng)
            if poison_rng.random() < 0.2:
                # stale scratch buffers / LDS must not matter: leave a pattern in all of them now and then
                ctx.debug_poison(int(poison_rng.choice([0, 0xffffffff, 0x80808080, 0x7fffffff, 0x00010001,
                                                        int(poison_rng.integers(0, 2**32))])))
            # keep the oracle's (S+1) * words * passes affordable
            if (S + 1) * (clen + S + L + 100) > 1.5e9:
                continue
            case_seed = int(rng.integers(0, 2**31))
            full = draw_full_range(rng)
            nbits, F, R, M = synth.make_case(case_seed, clen, S, L, fd, rd, with_m, mean_on=mean_on, mean_off=mean_off,
                                             full_range=full)
            skip_ncc = with_m and rng.random() < 0.2
            ref = oracle.calc_correlation(F, R, M, nbits, S, L, skip_ncc=skip_ncc)
            for flags in (0, ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE, ffi.PMX_FLAG_DEEP_LISTS, ffi.PMX_FLAG_EVENTS_HINT):
                if flags == ffi.PMX_FLAG_FORCE_DENSE and (S + 1) * nbits > 3e8:
                    continue
                if flags == ffi.PMX_FLAG_FORCE_SPARSE and L > 1024:
                    continue                       # reads longer than 1024 take the dense kernels
                fl = flags | (ffi.PMX_FLAG_SKIP_NCC if skip_ncc else 0)
                tag = f"seed={case_seed} S={S} L={L} clen={clen} fd={fd} rd={rd} m={with_m}/{mean_on}/{mean_off} full={full} flags={fl}"
                try:
                    out = ctx.calc_correlation(F, R, M, nbits, S, L, fl)
                except ffi.PmxError as e:
                    print("ERROR", tag, e, flush=True)
                    bad += 1
                    continue
                if not compare(out, ref, S, with_m, skip_ncc, tag):
                    bad += 1
            if rng.random() < 0.15:
                bad += fuzz_feed(ctx, rng, clen, S, L)
            if with_m and rng.random() < 0.25:
                bad += fuzz_regions(ctx, rng, M, nbits)
            if with_m and rng.random() < 0.3:
                lag = int(rng.choice([S, 300, 1023, 1024, 4000]))
                if (lag + 1) * nbits < 1.5e9:
                    want = oracle.mappable_len_readless(M, nbits, lag)
                    got = ctx.mappable_len(M, nbits, lag, 0)
                    if not np.array_equal(got.astype(np.int64), want.astype(np.int64)):
                        print("MISMATCH mappable_len", case_seed, lag, flush=True)
                        bad += 1
            n += 1
            if time.time() - last > 30:
                print(f"[fuzz] {n} cases, {bad} bad", flush=True)
                last = time.time()
    print(f"[fuzz] done: {n} cases, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
