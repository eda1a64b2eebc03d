#!/usr/bin/env python3
"""Ingest benchmark: synthetic coordinate-sorted BAM -> native reader -> calculator.

Writes a BAM of N single-end reads over hg38-sized chromosomes (vectorised numpy record layout, BGZF level 1),
then reports
  * reader only: records/s and GB/s of uncompressed BAM for 1..T threads (libpymasc_io.so, no GPU involved),
  * end to end with --gpu: feed_bam into CCHipCalculator (NCC, max_shift 1000) wall-clock, split into ingest and
    device time.
pysam is not installed, so the reference's own per-read loop cannot be timed here; the comparable CPU figure is a
pure-Python loop over the decoded arrays calling feed_forward_read / feed_reverse_read per read (--pyloop), which
is what handler/calc.py:140-153 does minus pysam's own decode cost.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pymasc_amd import bam as B  # noqa: E402
from pymasc_amd.synth import HG38  # noqa: E402
from tests import io_writers as W  # noqa: E402


def synth_bam(path, n_reads, seed=1, readlen=36, chroms=None):
    rng = np.random.default_rng(seed)
    refs = [(n, l) for n, l in HG38] if chroms is None else [(n, l) for n, l in HG38][:chroms]
    total = sum(l for _, l in refs)
    name_len = 12
    # one fixed-size record layout: 32-byte core + name + 1 CIGAR op + seq + qual
    rec_dtype = np.dtype([("block_size", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                          ("bin", "<u2"), ("n_cig", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("nref", "<i4"),
                          ("npos", "<i4"), ("tlen", "<i4"), ("name", "S%d" % name_len), ("cigar", "<u4"),
                          ("seq", "u1", ((readlen + 1) // 2,)), ("qual", "u1", (readlen,))])
    header = W.bam_header(refs)
    t0 = time.time()
    with open(path, "wb") as fp, ThreadPoolExecutor(8) as pool:
        def flush(data):
            blocks = [data[i:i + 0xff00] for i in range(0, len(data), 0xff00)]
            for z in pool.map(lambda b: W.bgzf_block(b, 1), blocks):
                fp.write(z)
        pending = header
        for rid, (_n, ln) in enumerate(refs):
            k = int(round(n_reads * ln / total))
            if k == 0:
                continue
            rec = np.zeros(k, dtype=rec_dtype)
            rec["block_size"] = rec_dtype.itemsize - 4
            rec["ref"] = rid
            rec["pos"] = np.sort(rng.integers(0, ln - readlen - 1, size=k))
            rec["l_name"] = name_len
            rec["mapq"] = rng.integers(0, 61, size=k)
            rec["bin"] = 4680
            rec["n_cig"] = 1
            fl = np.where(rng.random(k) < 0.5, 16, 0)
            fl = np.where(rng.random(k) < 0.03, fl | 0x400, fl)
            rec["flag"] = fl
            rec["l_seq"] = readlen
            rec["nref"] = -1
            rec["npos"] = -1
            rec["name"] = b"read0000000"
            rec["cigar"] = (readlen << 4) | 0
            rec["seq"] = rng.integers(0, 256, size=(k, (readlen + 1) // 2), dtype=np.uint8)
            rec["qual"] = rng.integers(20, 41, size=(k, readlen), dtype=np.uint8)
            pending += rec.tobytes()
            cut = len(pending) - len(pending) % 0xff00
            flush(pending[:cut])
            pending = pending[cut:]
        flush(pending)
        fp.write(W.BGZF_EOF)
    return refs, time.time() - t0


def time_reader(path, threads, mapq):
    t0 = time.time()
    n = 0
    with B.BamReader(path, threads=threads) as r:
        for ref, _pos, _rl, _rev in r.batches(mapq):
            n += ref.size
        c = r.counters()
    dt = time.time() - t0
    return {"threads": threads, "seconds": round(dt, 3), "records_per_s": round(c["records"] / dt),
            "kept": n, "records": c["records"], "uncompressed_GBps": round(c["bytes_out"] / dt / 1e9, 3),
            "compressed_GBps": round(c["bytes_in"] / dt / 1e9, 3)}


def time_device_reader(path, mapq, check=None):
    """The device path (libpymasc_ingest.so): file -> HBM -> inflate + CRC32 -> record chain -> filtered arrays IN HBM (what a
    device-side feed consumes), and the same + the copy of the arrays to the host (what BamReader.batches hands out)."""
    from pymasc_amd import bam_device as D
    out = []
    for rep in range(3):      # the first open also page-locks the staging buffers and loads the code object
        t0 = time.time()
        with D.DeviceBamReader(path) as r:
            t1 = time.time()
            kept = r.decode(mapq)
            t2 = time.time()
            n = 0
            cs = 0
            for ref, pos, _rl, rev in r.batches(mapq):
                n += ref.size
                cs += int(pos.astype(np.int64).sum()) + int(rev.sum())
            t3 = time.time()
            c, tm = r.counters(), r.timings()
        assert n == kept
        if check is not None:
            assert (n, cs) == check, ((n, cs), check)
        out.append({"rep": rep, "open_s": round(t1 - t0, 4), "decode_s": round(t2 - t1, 4), "resident_total_s": round(t2 - t0, 4),
                    "decode_again_and_copy_to_host_s": round(t3 - t2, 4), "kept": kept, "records": c["records"], "members": c["members"],
                    "pieces_rewalked": c["rewalked"], "phases_s": {k: round(v, 4) for k, v in tm.items()},
                    "records_per_s": round(c["records"] / (t2 - t0)), "uncompressed_GBps": round(c["bytes_out"] / (t2 - t0) / 1e9, 3),
                    "compressed_GBps": round(c["bytes_in"] / (t2 - t0) / 1e9, 3)})
        print(json.dumps(out[-1]), flush=True)
    return out


def time_bigwig(path_bw, chroms_per_call=None):
    """A synthetic hg38-shaped mappability track (runs of ~500 bp every ~1250 bp, bedGraph sections of 1024 items, zlib) read
    chromosome by chromosome: the host reader (zlib on threads) against the device reader (intervals left in HBM), same intervals."""
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    rng = np.random.default_rng(3)
    sizes = {n: l for n, l in HG38}
    tracks = {}
    t0 = time.time()
    for n, l in sizes.items():
        k = l // 1250
        starts = np.arange(k, dtype=np.int64) * 1250 + rng.integers(0, 600, k)
        ends = starts + rng.integers(200, 640, k)
        tracks[n] = list(zip(starts.tolist(), ends.tolist(), [1.0] * k))
    W.write_bigwig(path_bw, sizes, tracks, items_per_block=1024, rtree_block=256)
    out = {"intervals": sum(map(len, tracks.values())), "file_bytes": os.path.getsize(path_bw), "generate_s": round(time.time() - t0, 1)}
    del tracks
    host = []
    for _ in range(3):
        t0 = time.time()
        with BigWigReader(path_bw) as bw:
            chk = [(a.size, int(a.sum()) + int(b.sum())) for a, b, _v in (bw.fetch_arrays(1, c) for c in sizes)]
        host.append(round(time.time() - t0, 4))
    dev, dev_open = [], []
    for _ in range(3):
        t0 = time.time()
        with DeviceBigWigReader(path_bw) as bw:
            t1 = time.time()
            ns = [bw.fetch_device(1, c)[2] for c in sizes]
            dev.append(round(time.time() - t0, 4))
            dev_open.append(round(t1 - t0, 4))
    with DeviceBigWigReader(path_bw) as bw:     # parity of the whole track
        chk2 = [(a.size, int(a.sum()) + int(b.sum())) for a, b, _v in (bw.fetch_arrays(1, c) for c in sizes)]
    assert chk == chk2 and ns == [c[0] for c in chk]
    out.update({"host_reader_s": host, "device_reader_s": dev, "device_open_s": dev_open, "speedup": round(min(host) / min(dev), 2)})
    os.unlink(path_bw)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--path", default="/tmp/pymasc_ingest_bench.bam")
    ap.add_argument("--mapq", type=int, default=10)
    ap.add_argument("--threads", type=int, nargs="*", default=[1, 2, 4, 8, 16])
    ap.add_argument("--chroms", type=int, default=None)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--device", action="store_true", help="time the device-side reader (BGZF inflate + decode as HIP kernels)")
    ap.add_argument("--bigwig", action="store_true", help="also time the mappability track: host reader against device reader")
    ap.add_argument("--pyloop", type=int, default=0, help="time a per-read Python feeding loop over this many reads")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    refs, gen_s = synth_bam(a.path, a.reads, chroms=a.chroms)
    res = {"reads": a.reads, "bam_bytes": os.path.getsize(a.path), "generate_s": round(gen_s, 1), "reader": []}
    for t in a.threads:
        if t <= (os.cpu_count() or 1):
            res["reader"].append(time_reader(a.path, t, a.mapq))
            print(json.dumps(res["reader"][-1]), flush=True)

    if a.device:
        with B.BamReader(a.path, threads=16) as r:       # the checker: same records, same fields
            n = cs = 0
            for ref, pos, _rl, rev in r.batches(a.mapq):
                n += ref.size
                cs += int(pos.astype(np.int64).sum()) + int(rev.sum())
        res["device_reader"] = time_device_reader(a.path, a.mapq, check=(n, cs))
        best = min(x["resident_total_s"] for x in res["device_reader"])
        host = min(x["seconds"] for x in res["reader"]) if res["reader"] else None
        res["device_vs_host_reader"] = {"device_resident_s": best, "host_best_s": host, "speedup": round(host / best, 2) if host else None}
        print(json.dumps(res["device_vs_host_reader"]), flush=True)

    if a.pyloop:
        from tests.fake_context import FakeContext     # host-only stand-in: the loop never reaches a flush here
        from pymasc_amd.calculator import CCHipCalculator
        with B.BamReader(a.path) as r:
            ref, pos, rl, rev = next(r.batches(a.mapq, batch=a.pyloop))
        calc = CCHipCalculator(1000, 36, [n for n, _ in refs], [l for _, l in refs], context=FakeContext())
        same = ref == ref[0]
        pos, rl, rev = pos[same].tolist(), rl[same].tolist(), rev[same].tolist()
        name = refs[int(ref[0])][0]
        t0 = time.time()
        for p, l, v in zip(pos, rl, rev):
            if v:
                calc.feed_reverse_read(name, p, l)
            else:
                calc.feed_forward_read(name, p, l)
        dt = time.time() - t0
        res["python_per_read_loop"] = {"reads": len(pos), "reads_per_s": round(len(pos) / dt)}
        print(json.dumps(res["python_per_read_loop"]), flush=True)

    if a.gpu:
        from pymasc_amd.calculator import CCHipCalculator
        calc = CCHipCalculator(1000, 36, [n for n, _ in refs], [l for _, l in refs])
        t0 = time.time()
        with B.BamReader(a.path) as r:
            fed = B.feed_bam(calc, r, a.mapq)
        dt = time.time() - t0
        whole = calc.get_whole_result()
        res["end_to_end"] = {"seconds": round(dt, 3), "reads_fed": fed, "reads_per_s": round(fed / dt),
                             "forward_sum": int(whole.forward_sum), "reverse_sum": int(whole.reverse_sum)}
        calc.close()
        print(json.dumps(res["end_to_end"]), flush=True)
        if a.device:      # the same file, inflated + decoded + filtered on the GPU, records handed to the feeders in HBM
            from pymasc_amd import bam_device as D
            res["end_to_end_device"] = []
            for rep in range(3):
                calc = CCHipCalculator(1000, 36, [n for n, _ in refs], [l for _, l in refs])
                t0 = time.time()
                with D.DeviceBamReader(a.path) as r:
                    fed2 = r.feed(calc, a.mapq)
                dt2 = time.time() - t0
                w2 = calc.get_whole_result()
                assert (fed2, int(w2.forward_sum), int(w2.reverse_sum)) == (fed, int(whole.forward_sum), int(whole.reverse_sum))
                calc.close()
                res["end_to_end_device"].append({"seconds": round(dt2, 3), "reads_fed": fed2, "reads_per_s": round(fed2 / dt2)})
                print(json.dumps(res["end_to_end_device"][-1]), flush=True)
    if a.bigwig:
        res["bigwig"] = time_bigwig(a.path + ".bw")
        print(json.dumps(res["bigwig"]), flush=True)
    if a.out:
        with open(a.out, "w") as fp:
            json.dump(res, fp, indent=1)
    os.unlink(a.path)


if __name__ == "__main__":
    main()
