#!/usr/bin/env python3
"""Cost of the inflate kernel's paths: members of literals only, of short near matches, of far matches, and BAM-like records;
prints ns per output byte and per symbol of k_bgzf_inflate (whole grid, members in parallel)."""
import os, struct, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import io_writers as W
from pymasc_amd import bam_device as D

def member(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    c = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(c) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + c
            + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))

rng = np.random.default_rng(1)
N = 8192
def qual(n): return rng.integers(20, 41, n, dtype=np.uint8).tobytes()
cases = {
    "literals_4bit (huffman only, 21 values)": (lambda: qual(60000), 6, zlib.Z_HUFFMAN_ONLY),
    "literals_8bit (huffman only, 200 values)": (lambda: rng.integers(0, 200, 60000, dtype=np.uint8).tobytes(), 6, zlib.Z_HUFFMAN_ONLY),
    "short matches (text, level 6)": (lambda: b"".join(b"@r%07d\tchr1\t%d\t36M\n" % (i, i * 7) for i in range(2200))[:60000], 6, 0),
    "rle (dist 1)": (lambda: b"".join(bytes([int(v)]) * 200 for v in rng.integers(0, 256, 300)), 6, 0),
    "bam-like level 1": (lambda: b"".join(b"\x66\0\0\0" + b"\x01\0\0\0" + struct.pack("<i", 1000 + i) + b"read0000000\0" + rng.integers(0, 256, 18, dtype=np.uint8).tobytes() + qual(36) + b"\0" * 30 for i in range(560)), 1, 0),
}
head = member(W.bam_header([("c1", 1000)]))
for name, (gen, level, strat) in cases.items():
    uniq = [member(gen(), level, strat) for _ in range(8)]
    path = "/tmp/paths.bam"
    with open(path, "wb") as fp:
        fp.write(head + b"".join(uniq[i % 8] for i in range(N)) + W.BGZF_EOF)
    best = None
    for _ in range(3):
        with D.DeviceBamReader(path) as r:
            t = r.timings()["inflate_s"]; c = r.counters()
        best = t if best is None else min(best, t)
    print("%-45s %6.2f ms  %6.1f MB out  ratio %.2f  %.3f ns/byte (grid)  %.1f GB/s" % (name, best * 1e3, c["bytes_out"] / 1e6, c["bytes_out"] / c["bytes_in"], best * 1e9 / c["bytes_out"], c["bytes_out"] / best / 1e9), flush=True)
os.unlink(path)
