#!/usr/bin/env python3
"""Randomised parity of the device ingest (libpymasc_ingest.so) against Python's zlib and the host reader.

Every round writes one BGZF file of random members -- payload kinds (uniform bytes, skewed alphabets, text, repeats at
random distances up to 32 KB, runs, BAM-like records), zlib level 0..9, strategy (default / filtered / Huffman-only / RLE /
fixed), memLevel 1..9 (short blocks: several DEFLATE blocks per member), window 2^9..2^15, sync-flushed mixtures -- behind a
BAM header, and checks the inflated stream byte for byte; every other round is a BAM file of random records (record sizes from
tens of bytes to beyond a 16-KB piece, random flags / mapq / CIGARs / tags, member sizes 40 .. 65280) checked against the host
reader's records for two filters, and a random BigWig track (device reader == host reader, three thresholds).  usage: tools/fuzz_ingest.py --seconds 300 --seed 1"""
import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pymasc_amd import bam as B  # noqa: E402
from pymasc_amd import bam_device as D  # noqa: E402
from tests import io_writers as W  # noqa: E402


def raw_member(cdata, payload):
    bsize = 12 + 6 + len(cdata) + 8 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def payload(rng):
    kind = int(rng.integers(0, 7))
    n = int(rng.choice([0, 1, 7, 300, 5000, 30000, 60000], p=[0.02, 0.03, 0.05, 0.2, 0.3, 0.2, 0.2]))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:
        k = int(rng.integers(2, 60))
        p = rng.dirichlet(np.full(k, 0.3))
        return bytes(rng.choice(rng.integers(0, 256, k), n, p=p).astype(np.uint8))
    if kind == 2:
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(int(rng.integers(3, 200)))]
        out = b" ".join(words[int(i)] for i in rng.integers(0, len(words), n // 4 + 1))
        return out[:n]
    if kind == 3:
        piece = rng.integers(0, 256, int(rng.integers(1, 33000)), dtype=np.uint8).tobytes()
        return (piece * (n // max(len(piece), 1) + 2))[:n]
    if kind == 4:
        return b"".join(bytes([int(v)]) * int(c) for v, c in zip(rng.integers(0, 256, 400), rng.integers(1, 600, 400)))[:n]
    if kind == 5:
        recs = []
        while sum(map(len, recs)) < n:
            recs.append(W.bam_record(0, int(rng.integers(0, 1 << 28)), int(rng.integers(0, 61)), int(rng.integers(0, 4096)),
                                     [("M", int(rng.integers(20, 150)))], b"r%d" % int(rng.integers(0, 1 << 30))))
        return b"".join(recs)[:n]
    a = rng.integers(0, 256, n, dtype=np.uint8)
    a[rng.random(n) < 0.7] = 65
    return a.tobytes()


def member(rng, p):
    level = int(rng.integers(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    mem = int(rng.integers(1, 10))
    wbits = -int(rng.integers(9, 16))
    for _ in range(6):
        co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
        if rng.random() < 0.2 and len(p) > 10:      # sync-flushed pieces: blocks of different types in one member
            cuts = sorted(set(int(x) for x in rng.integers(1, len(p), int(rng.integers(1, 5)))))
            c = b"".join(co.compress(p[a:b]) + co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH)
                         for a, b in zip([0] + cuts, cuts + [len(p)])) + co.flush()
        else:
            c = co.compress(p) + co.flush()
        if 12 + 6 + len(c) + 8 - 1 < 65536:
            return raw_member(c, p), p
        p = p[:len(p) * 2 // 3]     # did not fit a member: shorter
    raise AssertionError


def round_members(rng, path):
    head = W.bam_header([("c1", 1000)])
    ms, want = [raw_member(zlib.compressobj(6, zlib.DEFLATED, -15).compress(head) + zlib.compressobj(6, zlib.DEFLATED, -15).flush(), head)], [head]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    ms[0] = raw_member(co.compress(head) + co.flush(), head)
    for _ in range(int(rng.integers(1, 400))):
        m, p = member(rng, payload(rng))
        ms.append(m)
        want.append(p)
    with open(path, "wb") as fp:
        fp.write(b"".join(ms) + (W.BGZF_EOF if rng.random() < 0.8 else b""))
    with D.DeviceBamReader(path) as r:
        got = r.inflated()
    exp = b"".join(want)
    if got != exp:
        first = next(i for i in range(min(len(exp), len(got))) if got[i] != exp[i]) if len(got) == len(exp) else -1
        raise AssertionError("inflated stream differs: lengths %d / %d, first difference %d" % (len(got), len(exp), first))
    return len(ms), len(exp)


def round_records(rng, path):
    nref = int(rng.integers(1, 6))
    refs = [("chr%d" % i, int(rng.integers(1000, 1 << 28))) for i in range(nref)]
    recs = []
    for rid, (_n, ln) in enumerate(refs):
        pos = 0
        for _ in range(int(rng.integers(0, 1500))):
            pos = min(pos + int(rng.integers(0, 300)), ln - 1)
            n = int(rng.choice([1, 20, 36, 101, 250, 3000, 30000], p=[0.02, 0.2, 0.3, 0.3, 0.15, 0.025, 0.005]))
            ops = [("M", n)] if rng.random() < 0.6 else [("S", int(rng.integers(1, 9))), ("M", n), ("I", int(rng.integers(1, 4))), ("D", 2), ("H", 3)]
            if rng.random() < 0.03:
                ops = []
            tags = b"" if rng.random() < 0.5 else b"NMi" + struct.pack("<i", 1) + b"XAZ" + bytes(rng.integers(65, 91, int(rng.integers(0, 50)), dtype=np.uint8)) + b"\0"
            ref = rid if rng.random() > 0.01 else -1
            recs.append(W.bam_record(ref, pos if ref >= 0 else -1, int(rng.integers(0, 256)), int(rng.choice([0, 16, 4, 0x400, 0x81, 0x41, 0x10 | 0x400])),
                                     ops, b"q%d" % len(recs), tags=tags))
    W.write_bam(path, refs, recs, block=int(rng.choice([40, 97, 500, 4096, 0xff00])), level=int(rng.integers(0, 10)), eof=bool(rng.random() < 0.8))
    for mapq, excl in ((int(rng.integers(0, 40)), B.PMX_BAM_DEFAULT_EXCLUDE), (0, int(rng.choice([0, 4, 0x404])))):
        with B.BamReader(path, threads=4, index=False) as h:
            exp = [tuple(a.tolist()) for b in h.batches(mapq, excl) for a in [np.stack([b[0], b[1], b[2], b[3].astype(np.int32)])]]
        with D.DeviceBamReader(path) as r:
            got = [tuple(a.tolist()) for b in r.batches(mapq, excl) for a in [np.stack([b[0], b[1], b[2], b[3].astype(np.int32)])]]
            c = r.counters()
        e = np.concatenate([np.array(x) for x in exp], axis=1) if exp else np.zeros((4, 0), int)
        g = np.concatenate([np.array(x) for x in got], axis=1) if got else np.zeros((4, 0), int)
        assert e.shape == g.shape and (e == g).all(), "records differ (mapq %d, exclude %#x)" % (mapq, excl)
        assert c["records"] == len(recs)
    return len(recs), c["rewalked"]


def round_bigwig(rng, path):
    """A random track (bedGraph / variableStep / fixedStep sections, compressed or not, block and tree sizes, values around the
    thresholds): every chromosome at three thresholds, device reader == host reader."""
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    kind = str(rng.choice(["bedgraph", "varstep", "fixedstep"]))
    chromsizes = {"c%d" % i: int(rng.integers(2000, 300000)) for i in range(int(rng.integers(1, 7)))}
    span = None if kind == "bedgraph" else int(rng.integers(1, 60))
    step = int(rng.integers(1, 80)) if kind == "fixedstep" else None
    tracks = {}
    for name, size in chromsizes.items():
        if rng.random() < 0.15:
            continue
        iv, p = [], int(rng.integers(0, 50))
        while p < size - 300:
            ln = int(rng.integers(1, 400)) if span is None else span
            iv.append((p, p + ln, float(rng.choice([0.0, 0.25, 0.5, 1.0, 1.0, 3.5]))))
            p += step if (step is not None and rng.random() < 0.8) else ln + int(rng.integers(0, 300))
        tracks[name] = iv
    if not tracks:
        return 0
    W.write_bigwig(path, chromsizes, tracks, kind=kind, compress=bool(rng.random() < 0.8), items_per_block=int(rng.choice([1, 5, 64, 1024])),
                   rtree_block=int(rng.choice([2, 4, 64])), bpt_block=int(rng.choice([2, 3, 16])), span=span or 1, step=step or 1)
    n = 0
    with BigWigReader(path) as h, DeviceBigWigReader(path) as d:
        assert d.chromsizes == h.chromsizes
        for thr in (1, 0, 0.5):
            for c in chromsizes:
                a, b = h.fetch_arrays(thr, c), d.fetch_arrays(thr, c)
                assert all((x == y).all() for x, y in zip(a, b)) and a[0].size == b[0].size, (kind, thr, c)
                n += a[0].size
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    path = "/tmp/fuzz_ingest_%d.bam" % os.getpid()
    t0 = time.time()
    rounds = members = nbytes = nrecs = rewalked = nivs = 0
    last = t0
    while time.time() - t0 < a.seconds:
        m, n = round_members(rng, path)
        members += m
        nbytes += n
        k, rw = round_records(rng, path)
        nrecs += k
        rewalked += rw
        nivs += round_bigwig(rng, path + ".bw")
        rounds += 1
        if time.time() - last > 30:
            last = time.time()
            print("[fuzz_ingest] %d rounds: %d members (%.1f MB inflated), %d records, %d pieces rewalked, %d BigWig intervals, 0 bad" % (rounds, members, nbytes / 1e6, nrecs, rewalked, nivs), flush=True)
    print("[fuzz_ingest] done (seed %d): %d rounds, %d members, %.1f MB inflated == zlib, %d records == host reader, %d pieces rewalked, "
          "%d BigWig intervals == host reader, 0 bad" % (a.seed, rounds, members, nbytes / 1e6, nrecs, rewalked, nivs), flush=True)
    os.unlink(path)
    if os.path.exists(path + ".bw"):
        os.unlink(path + ".bw")


if __name__ == "__main__":
    main()
