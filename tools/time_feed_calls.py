#!/usr/bin/env python3
"""GPU box diagnostic: host time of the feeders' calls (they only enqueue) for one hg38-chr1-sized chromosome."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymasc_amd import ffi
ctx = ffi.Context(0)
nbits = 248_956_422 + 2000
n = 2_480_000
rng = np.random.default_rng(1)
pos = np.sort(rng.integers(1, nbits - 100, size=n)).astype(np.int32)
rev = (rng.random(n) < 0.5).astype(np.uint8)
packed = ctx.host_array(n, np.int32); packed[:] = ffi.pack_strand(pos, rev)
b_, e_ = ctx.host_packed([50000, 50000], np.uint32)
st = np.sort(rng.integers(0, nbits - 5000, size=50000)); b_[:] = st; e_[:] = st + 1000
dF, dR, dM = ctx.bits_alloc(nbits), ctx.bits_alloc(nbits), ctx.bits_alloc(nbits)
dS = ctx.bits_alloc(ffi.PMX_FEED_WORDS * 64)
def t(f, reps=20):
    ctx.sync(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0); ctx.sync()
    ts.sort(); return ts[len(ts) // 2] * 1e6
print("bits_clear            %.1f us" % t(lambda: ctx.bits_clear(dF, nbits)))
print("feed_reads (enqueue)  %.1f us" % t(lambda: ctx.feed_reads(dF, dR, nbits, packed, 36, None, 0, dS)))
print("set_regions (enqueue) %.1f us" % t(lambda: ctx.bits_set_regions_async(dM, nbits, b_, e_, 1, None)))
t0 = time.perf_counter(); ctx.feed_reads(dF, dR, nbits, packed, 36, None, 0, dS); ctx.sync(); print("feed_reads + sync     %.1f us" % ((time.perf_counter() - t0) * 1e6))
# the calls as the calculator issues them, back to back (no sync): where does the host wait?
ctx.sync()
rows = []
T0 = time.perf_counter()
for k in range(12):
    a = time.perf_counter(); ctx.bits_clear(dF, nbits); ctx.bits_clear(dR, nbits)
    b = time.perf_counter(); ctx.feed_reads(dF, dR, nbits, packed, 36, None, 0, dS)
    c = time.perf_counter(); ctx.bits_clear(dM, nbits); ctx.bits_set_regions_async(dM, nbits, b_, e_, 1, None)
    d = time.perf_counter(); rows.append(((a - T0) * 1e6, (b - a) * 1e6, (c - b) * 1e6, (d - c) * 1e6))
ctx.sync(); print("total %.1f us" % ((time.perf_counter() - T0) * 1e6))
for r in rows: print("  start %8.1f  clears %6.1f  feed_reads %6.1f  regions %6.1f" % r)
