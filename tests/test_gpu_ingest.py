"""Device-side BAM ingest (libpymasc_ingest.so: BGZF inflate, CRC32, record chain, read filter as HIP kernels) against the
host reader (libpymasc_io.so, zlib) and against Python's zlib, bit for bit.  SURVEY.md §8 row f1; the reference side is
PyMaSC/handler/calc.py:140-153 + handler/read.py:62-155 (pysam), pinned through the reference's own BAM <-> SAM twin."""
import os
import struct
import zlib

import numpy as np
import pytest

from pymasc_amd import bam as B
from pymasc_amd import bam_device as D
from . import fixtures as fx
from . import io_writers as W
from .test_io_readers import BAM, _all_reads, _expected

pytestmark = pytest.mark.gpu


def _member(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, crc=None, isize=None) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    cdata = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(cdata) + 8 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + cdata + struct.pack("<II", (zlib.crc32(payload) & 0xffffffff) if crc is None else crc,
                                  len(payload) if isize is None else isize))


def _payloads(rng):
    """Byte strings that exercise every part of the inflate kernel."""
    rnd = rng.integers(0, 256, 60000, dtype=np.uint8).tobytes()
    text = (b"@read/1\tchr1\t12345\t36M\tACGTTGCA\n" * 3000)[:65000]
    piece = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    far = (piece * 13)[:65000]                                 # matches 5000 back: beyond the 4-KB ring in LDS
    farther = (rng.integers(0, 256, 30000, dtype=np.uint8).tobytes() * 2)[:60000]   # 30000 back
    runs = b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 400), rng.integers(1, 400, 400)))[:65000]
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), 60000, p=np.r_[[0.5, 0.2, 0.1], np.full(253, 0.2 / 253)]))
    small = rng.integers(0, 4, 300, dtype=np.uint8).tobytes()
    return {"random": rnd, "text": text, "far": far, "farther": farther, "runs": runs, "skew": skew, "small": small,
            "one": b"x", "empty": b""}


def test_inflate_matches_zlib(tmp_path):
    """Members of every DEFLATE block type and match geometry, concatenated behind a BAM header: the inflated stream equals
    the payloads byte for byte (and every member's CRC32 / ISIZE was checked on the device)."""
    rng = np.random.default_rng(5)
    head = W.bam_header([("c1", 1000)])
    members, want = [_member(head)], [head]
    for name, p in _payloads(rng).items():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(p) > 60000:
                p = p[:60000]
            members.append(_member(p, level, strategy))
            want.append(p)
    path = tmp_path / "m.bam"
    path.write_bytes(b"".join(members) + W.BGZF_EOF)
    with D.DeviceBamReader(path) as r:
        c = r.counters()
        assert c["members"] == len(members) + 1 and c["bytes_out"] == sum(map(len, want))
        got = r.inflated()
    exp = b"".join(want)
    if got != exp:
        first = next(i for i in range(len(exp)) if got[i] != exp[i])
        off = np.cumsum([0] + [len(w) for w in want])
        raise AssertionError("first difference at byte %d (member %d)" % (first, int(np.searchsorted(off, first, "right")) - 1))


def test_multi_block_members(tmp_path):
    """zlib starts a new DEFLATE block when its symbol buffer fills: one member, several dynamic blocks; and a member whose
    blocks are of different types (sync-flushed pieces: stored + fixed + dynamic)."""
    rng = np.random.default_rng(6)
    head = W.bam_header([("c1", 1000)])
    a = bytes(rng.integers(0, 7, 65000, dtype=np.uint8))        # ~65000 symbols: several blocks at memLevel 1
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 1)
    ca = co.compress(a) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), b"abc" * 500, b"q" * 40, bytes(rng.integers(0, 3, 9000, dtype=np.uint8))]
    cb = b"".join(co.compress(p) + co.flush(zlib.Z_FULL_FLUSH) for p in parts) + co.flush()
    b = b"".join(parts)

    def raw_member(cdata, payload):
        bsize = 12 + 6 + len(cdata) + 8 - 1
        return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))

    path = tmp_path / "mb.bam"
    path.write_bytes(_member(head) + raw_member(ca, a) + raw_member(cb, b) + W.BGZF_EOF)
    with D.DeviceBamReader(path) as r:
        assert r.inflated() == head + a + b


@pytest.mark.parametrize("mapq", [0, 1, 10, 30, 255])
def test_reference_bam_matches_its_sam_twin(mapq):
    """The reference's own test BAM through the device path == the reads of its SAM twin (tests/golden/*.reads.tsv)."""
    names, lengths = fx.load_refs()
    with D.DeviceBamReader(BAM) as r:
        assert list(r.references) == names and list(r.lengths) == lengths
        assert r.header_text.startswith("@HD") or r.header_text.startswith("@SQ")
        got = _all_reads(r, mapq, batch=500)
        assert got == fx.load_reads(mapq)
        c = r.counters()
        assert c["records"] == 2501 and c["kept"] == len(got) and c["bytes_in"] == os.path.getsize(BAM)
    with B.BamReader(BAM) as h:
        assert _all_reads(h, mapq) == got


@pytest.mark.parametrize("block", [0xff00, 4096, 257, 61])
def test_synthetic_bam_records_straddling_blocks(tmp_path, block):
    rng = np.random.default_rng(block)
    refs = [("chrA", 50000), ("chrB_with_a_long_name" * 3, 30000), ("chrC", 999)]
    recs, meta = W.synth_bam_records(rng, refs, 700)
    path = tmp_path / "s.bam"
    W.write_bam(path, refs, recs, block=block)
    with D.DeviceBamReader(path, threads=4) as r:
        assert list(zip(r.references, r.lengths)) == refs
        for mapq in (0, 20):
            assert _all_reads(r, mapq, batch=333) == _expected(meta, refs, mapq)
        assert r.counters()["records"] == len(recs)
        for i, (name, _l) in enumerate(refs):      # one reference at a time == the host reader's filter on ref_id
            got = [x for b in r.fetch(name, 20) for x in zip(*[a.tolist() for a in b])]
            exp = [(i, p, l, v) for v, n, p, l in _expected(meta, refs, 20) if n == name]
            assert got == exp


def test_many_pieces_and_long_records(tmp_path):
    """A stream of many 16-KB pieces: records of very different sizes (a few longer than a piece, so that pieces without any
    record start exist), every piece boundary cutting a record somewhere; host reader == device reader."""
    rng = np.random.default_rng(12)
    refs = [("c1", 5_000_000)]
    recs, pos = [], 0
    for i in range(6000):
        pos += int(rng.integers(0, 50))
        n = int(rng.choice([20, 36, 101, 250, 2000, 40000], p=[0.3, 0.3, 0.2, 0.15, 0.045, 0.005]))
        cig = [("S", 2), ("M", n - 2)] if i % 3 else [("M", n)]
        recs.append(W.bam_record(0, pos, int(rng.integers(0, 61)), int(rng.choice([0, 16, 0x400, 4])), cig, b"r%d" % i,
                                 tags=b"XAZ" + bytes(rng.integers(65, 91, int(rng.integers(0, 40)), dtype=np.uint8)) + b"\0"))
    path = tmp_path / "l.bam"
    W.write_bam(path, refs, recs, level=1)
    with B.BamReader(path, threads=8) as h:
        exp = _all_reads(h, 5)
    with D.DeviceBamReader(path) as r:
        assert _all_reads(r, 5) == exp
        c = r.counters()
        assert c["records"] == len(recs)
        # pieces inside a record longer than a piece have no guess and take their neighbour's end, one round per piece of the
        # record; nothing else is walked twice (a repair once spread from every long record to the end of the file)
        pieces = (c["bytes_out"] + 16383) // 16384
        assert 0 < c["rewalked"] <= pieces // 2


def test_filter_and_field_corner_cases(tmp_path):
    refs = [("c1", 100000), ("c2", 100000)]
    R = W.bam_record
    recs = [
        R(0, 99, 30, 0, [("M", 36)]),
        R(0, 199, 30, 16, [("S", 3), ("M", 30), ("I", 2), ("D", 5), ("M", 4), ("H", 9)]),
        R(0, 299, 30, 0, [("=", 10), ("X", 1), ("N", 100), ("=", 5), ("P", 2)]),
        R(0, 399, 9, 0, [("M", 36)]),
        R(0, 499, 30, 4, [("M", 36)]),
        R(0, 599, 30, 0x400, [("M", 36)]),
        R(0, 699, 30, 0x81, [("M", 36)]),
        R(0, 799, 30, 0x41, [("M", 36)]),
        R(0, 899, 30, 0, []),
        R(0, 999, 30, 0, [("H", 5), ("D", 5)]),
        W.long_cigar_record(0, 1099, 30, 16, [("M", 20), ("I", 1), ("M", 20), ("S", 4)]),
        R(1, 4, 255, 16, [("M", 50)]),
        R(-1, -1, 30, 0, [("M", 36)]),
    ]
    path = tmp_path / "c.bam"
    W.write_bam(path, refs, recs)
    with D.DeviceBamReader(path) as r:
        assert _all_reads(r, 10) == [
            (False, "c1", 100, 36), (True, "c1", 200, 39), (False, "c1", 300, 16), (False, "c1", 800, 36),
            (True, "c1", 1100, 45), (True, "c2", 5, 50)]
        assert r.counters()["records"] == len(recs)
        got = _all_reads(r, 10, flag_exclude=B.PMX_BAM_FLAG_UNMAPPED)      # a second decode on the resident stream
        assert [g[2] for g in got] == [100, 200, 300, 600, 700, 800, 1100, 5]


def test_header_only_and_empty_cases(tmp_path):
    refs = [("c1", 1000)]
    p = tmp_path / "h.bam"
    W.write_bam(p, refs, [])
    with D.DeviceBamReader(p) as r:
        assert r.references == ("c1",) and _all_reads(r, 0) == []
    W.write_bam(p, [], [], text="@HD\tVN:1.0\n")
    with D.DeviceBamReader(p) as r:
        assert r.references == () and _all_reads(r, 0) == []
    W.write_bam(p, refs, [W.bam_record(0, 5, 1, 0, [("M", 10)])], eof=False)
    with D.DeviceBamReader(p) as r:
        assert _all_reads(r, 0) == [(False, "c1", 6, 10)]


def test_corrupt_input_is_reported(tmp_path):
    refs = [("c1", 100000)]
    recs, _ = W.synth_bam_records(np.random.default_rng(2), refs, 300)
    good = tmp_path / "g.bam"
    W.write_bam(good, refs, recs, block=2048)
    raw = bytearray(open(good, "rb").read())

    def reading(data):
        p = tmp_path / "bad.bam"
        p.write_bytes(bytes(data))
        with D.DeviceBamReader(p) as r:
            return _all_reads(r, 0)

    with pytest.raises(B.PmxIOError, match="truncated"):
        reading(raw[:len(raw) // 2])
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x55
    with pytest.raises(B.PmxIOError, match="CRC32|inflate|BGZF|DEFLATE|Huffman"):
        reading(flipped)
    # single-bit flips inside a member: the device path reports an error exactly where the host reader (zlib) does -- a flip in
    # the gzip header's unused bytes or in the padding bits behind the last code changes nothing for either -- and none hangs
    rng = np.random.default_rng(3)
    raised = 0
    for _ in range(60):
        bad = bytearray(raw)
        at = int(rng.integers(len(raw) // 3, len(raw) // 3 + 1500))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / "flip.bam"
        p.write_bytes(bytes(bad))
        try:
            with B.BamReader(p) as h:
                exp = _all_reads(h, 0)
        except B.PmxIOError:
            exp = None
        if exp is None:
            raised += 1
            with pytest.raises(B.PmxIOError):
                reading(bad)
        else:
            assert reading(bad) == exp
    assert raised >= 40
    with pytest.raises(B.PmxIOError, match="BGZF"):
        reading(b"\x1f\x8b\x08\x00" + bytes(raw[4:]))
    with pytest.raises(B.PmxIOError, match="magic"):
        p = tmp_path / "x.bam"
        p.write_bytes(W.bgzf_compress(b"SAM\1" + b"\0" * 100))
        D.DeviceBamReader(p)
    with pytest.raises(B.PmxIOError, match="inside an alignment record"):
        data = W.bam_header(refs) + b"".join(recs)
        p = tmp_path / "t.bam"
        p.write_bytes(W.bgzf_compress(data[:-7]))
        with D.DeviceBamReader(p) as r:
            _all_reads(r, 0)
    with pytest.raises(B.PmxIOError, match="cannot open"):
        D.DeviceBamReader(tmp_path / "missing.bam")
    with pytest.raises(B.PmxIOError, match="unknown reference"):
        p = tmp_path / "r.bam"
        W.write_bam(p, refs, [W.bam_record(3, 5, 1, 0, [("M", 10)])])
        with D.DeviceBamReader(p) as r:
            _all_reads(r, 0)
    with pytest.raises(B.PmxIOError, match="CRC32"):
        p = tmp_path / "crc.bam"
        p.write_bytes(_member(W.bam_header(refs)) + _member(b"".join(recs[:50]), crc=12345) + W.BGZF_EOF)
        D.DeviceBamReader(p)
    with pytest.raises(B.PmxIOError, match="recorded size"):
        p = tmp_path / "isz.bam"
        body = b"".join(recs[:50])
        p.write_bytes(_member(W.bam_header(refs)) + _member(body, isize=len(body) - 3) + W.BGZF_EOF)
        D.DeviceBamReader(p)


def test_files_to_tables_through_the_device_reader(tmp_path):
    """The reference's golden run (-d 300 -q 10 -r 36 -m bigwig) with the BAM inflated and decoded on the GPU: pymasc_amd.bam.feed_bam
    takes the device reader as it takes the host reader, and the output tables equal the reference's."""
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.calculator import CCHipCalculator
    from .test_io_readers import BIGWIG
    for device_feed in (False, True):
        with D.DeviceBamReader(BAM) as bam, BigWigReader(BIGWIG) as bw:
            calc = CCHipCalculator(300, 36, bam.references, bam.lengths, bwfeeder=bw)
            # feed_bam: the kept records come back as host arrays; reader.feed: they stay in HBM (pmx_feed_reads_dev)
            fed = bam.feed(calc, 10) if device_feed else B.feed_bam(calc, bam, mapq_criteria=10)
            assert fed == 1292
            whole = calc.get_whole_result()
            calc.close()
            names = bam.references
        _check_tables(tmp_path / ("d%d" % device_feed), whole, names)


def _check_tables(tmp_path, whole, names):
    import csv
    from pymasc_amd import tables as T
    os.makedirs(tmp_path, exist_ok=True)
    for p in T.write_tables(tmp_path / "ENCFF000RMB-test.bam", whole, references=names):
        gold = os.path.join(fx.GOLDEN, p.name)
        if p.name.endswith("_nreads.tab"):
            assert open(p, "rb").read() == open(gold, "rb").read()
        else:
            g = list(csv.reader(open(gold, newline=""), dialect="excel-tab"))
            o = list(csv.reader(open(p, newline=""), dialect="excel-tab"))
            assert g[0] == o[0] and len(g) == len(o)
            np.testing.assert_almost_equal(np.array([r[1:] for r in o[1:]], dtype=float),
                                           np.array([r[1:] for r in g[1:]], dtype=float), decimal=15)


def test_device_feed_equals_host_feed(tmp_path):
    """Several chromosomes (one of them left out by the caller), duplicates at one position, mixed read lengths: the rows after
    DeviceBamReader.feed (records never leave HBM) equal the rows after feed_bam over the host reader, integer for integer."""
    from pymasc_amd.calculator import CCHipCalculator
    rng = np.random.default_rng(21)
    refs = [("c1", 300000), ("c2", 150000), ("skipme", 50000), ("c4", 90000)]
    recs, _meta = W.synth_bam_records(rng, refs, 4000)
    recs += [W.bam_record(3, 89963, 30, 0, [("M", 36)])] * 3 + [W.bam_record(3, 89963, 30, 16, [("M", 36)]), W.bam_record(3, 89963, 30, 16, [("M", 30)])]
    path = tmp_path / "f.bam"
    W.write_bam(path, refs, recs, level=1)
    want = ["c1", "c2", "c4"]
    lens = {n: l for n, l in refs}
    out = []
    for dev in (False, True):
        calc = CCHipCalculator(200, 36, want, [lens[n] for n in want])
        if dev:
            with D.DeviceBamReader(path) as r:
                fed = r.feed(calc, 5, references=want)
        else:
            with B.BamReader(path, index=False) as r:
                fed = B.feed_bam(calc, r, 5, references=want)
        res = {n: calc.get_result(n) for n in want}
        out.append((fed, {n: (int(v.chrom.forward_sum), int(v.chrom.reverse_sum), np.asarray(v.chrom.ccbins).tolist(),
                              int(v.chrom.forward_read_len_sum), int(v.chrom.reverse_read_len_sum)) for n, v in res.items()}))
        calc.close()
    assert out[0] == out[1] and out[0][0] > 8000


# ---------------------------------------------------------------- BigWig on the device --------------------------------
def _same_intervals(a, b):
    return all((x == y).all() and x.dtype == y.dtype for x, y in zip(a, b)) and len(a[0]) == len(b[0])


def test_reference_bigwig_on_the_device_matches_host_reader_and_bedgraph_twin():
    """The reference's test track: chromosome dictionary and every chromosome's intervals at three thresholds == the host reader
    (zlib), whose own test pins it to the bedGraph twin (tests/test_io_readers.py)."""
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    from .test_io_readers import BIGWIG
    with BigWigReader(BIGWIG) as h, DeviceBigWigReader(BIGWIG) as d:
        assert d.chromsizes == h.chromsizes and list(d.chromsizes) == list(h.chromsizes)
        total = 0
        for thr in (1, 0, 0.5):
            for c in h.chromsizes:
                a, b = h.fetch_arrays(thr, c), d.fetch_arrays(thr, c)
                assert _same_intervals(a, b), (thr, c)
                total += a[0].size
                assert list(d.fetch(thr, c))[:5] == list(h.fetch(thr, c))[:5]
        assert total > 1970
        with pytest.raises(KeyError):
            d.fetch_arrays(1, "chrNope")


@pytest.mark.parametrize("kind,compress", [("bedgraph", True), ("bedgraph", False), ("varstep", True), ("fixedstep", True), ("fixedstep", False)])
def test_synthetic_bigwig_sections_on_the_device(tmp_path, kind, compress):
    """Every section type, compressed and not, several R-tree levels, chromosomes sharing index entries, values around the
    threshold, an interval beyond the chromosome's end: device == host reader."""
    from pymasc_amd.bigwig import BigWigReader
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    from .test_io_readers import _tracks
    rng = np.random.default_rng(31)
    chromsizes = {"chr1": 500000, "chr2": 120000, "chrX_random_with_a_long_name": 40000, "chrM": 16571}
    span, step = (25, 40) if kind != "bedgraph" else (None, None)
    tracks = _tracks(rng, chromsizes, span=span, step=step)
    path = tmp_path / "t.bw"
    kw = {} if kind == "bedgraph" else {"span": span, "step": step}
    W.write_bigwig(path, chromsizes, tracks, kind=kind, compress=compress, items_per_block=37, rtree_block=3, bpt_block=2, **kw)
    with BigWigReader(path) as h, DeviceBigWigReader(path) as d:
        assert d.chromsizes == h.chromsizes
        for thr in (1, 0, 0.25):
            for c in chromsizes:
                assert _same_intervals(h.fetch_arrays(thr, c), d.fetch_arrays(thr, c)), (thr, c)
        _b, _e, n, in_order = d.fetch_device(1, "chr1")
        a = h.fetch_arrays(1, "chr1")
        assert n == a[0].size and in_order == bool((a[0] < a[1]).all() and (a[0][1:] >= a[1][:-1]).all())


def test_bigwig_corrupt_input_on_the_device(tmp_path):
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    from .test_io_readers import _tracks
    rng = np.random.default_rng(5)
    chromsizes = {"c1": 100000}
    path = tmp_path / "g.bw"
    W.write_bigwig(path, chromsizes, _tracks(rng, chromsizes), items_per_block=64)
    raw = bytearray(open(path, "rb").read())
    with pytest.raises(B.PmxIOError, match="magic"):
        p = tmp_path / "m.bw"
        p.write_bytes(b"\0\0\0\0" + bytes(raw[4:]))
        DeviceBigWigReader(p)
    with pytest.raises(IOError):
        DeviceBigWigReader(tmp_path / "missing.bw")
    # a flipped byte inside the first data block: the DEFLATE decoder or the Adler-32 check reports it
    with DeviceBigWigReader(path) as d:
        d.fetch_arrays(1, "c1")
    data_off = struct.unpack("<Q", raw[16:24])[0]
    bad = bytearray(raw)
    bad[data_off + 4 + 40] ^= 0x10
    p = tmp_path / "f.bw"
    p.write_bytes(bytes(bad))
    with pytest.raises(B.PmxIOError, match="inflate"):
        with DeviceBigWigReader(p) as d:
            d.fetch_arrays(1, "c1")


def test_golden_run_with_both_inputs_on_the_device(tmp_path):
    """-d 300 -q 10 -r 36 -m bigwig with the BAM file AND the mappability track decoded on the GPU, records and intervals handed to the
    calculator in HBM: the reference's golden tables."""
    from pymasc_amd.bigwig_device import DeviceBigWigReader
    from pymasc_amd.calculator import CCHipCalculator
    from .test_io_readers import BIGWIG
    with D.DeviceBamReader(BAM) as bam, DeviceBigWigReader(BIGWIG) as bw:
        calc = CCHipCalculator(300, 36, bam.references, bam.lengths, bwfeeder=bw)
        assert bam.feed(calc, 10) == 1292
        whole = calc.get_whole_result()
        calc.close()
        names = bam.references
    _check_tables(tmp_path / "both", whole, names)


def test_pipelined_open_gives_the_same_stream(tmp_path, monkeypatch):
    """PMX_DBAM_PIPELINE=1 (inflate launches behind the copies of the pieces, four streams, growing tables): same bytes, same records."""
    rng = np.random.default_rng(41)
    refs = [("c1", 3_000_000), ("c2", 1_000_000)]
    recs, meta = W.synth_bam_records(rng, refs, 20000)
    path = tmp_path / "p.bam"
    W.write_bam(path, refs, recs, block=700, level=1)         # ~ 6000 members: the member table grows
    with D.DeviceBamReader(path) as r:
        a, ra = r.inflated(), _all_reads(r, 7)
    monkeypatch.setenv("PMX_DBAM_PIPELINE", "1")
    with D.DeviceBamReader(path) as r:
        b, rb = r.inflated(), _all_reads(r, 7)
        assert r.counters()["members"] > 4096
    assert a == b and ra == rb == _expected(meta, refs, 7)


def test_unsorted_file_falls_back_to_host_arrays_and_is_refused_alike(tmp_path):
    """More than 65536 runs of one reference (records alternating between two chromosomes): DeviceBamReader.feed has no run table to
    hand over, takes the records through host arrays like feed_bam, and the calculator refuses the file as it refuses it from the host
    reader (ReadUnsortedError, mscc.pyx:351-364)."""
    from pymasc_amd.calculator import CCHipCalculator
    from pymasc_amd.exceptions import ReadUnsortedError
    refs = [("c1", 500000), ("c2", 500000)]
    recs = [W.bam_record(i & 1, 10 + i, 30, 0, [("M", 36)], b"u%d" % i) for i in range(66000)]
    path = tmp_path / "u.bam"
    W.write_bam(path, refs, recs, level=1)
    with D.DeviceBamReader(path) as r:
        assert r.decode(0) == len(recs) and r.device_runs() is None
    for dev in (False, True):
        calc = CCHipCalculator(100, 36, [n for n, _ in refs], [l for _, l in refs])
        try:
            with pytest.raises(ReadUnsortedError):
                if dev:
                    with D.DeviceBamReader(path) as r:
                        r.feed(calc, 0)
                else:
                    with B.BamReader(path, index=False) as r:
                        B.feed_bam(calc, r, 0)
        finally:
            calc.close()


def test_header_larger_than_the_first_read_back(tmp_path):
    """A reference dictionary of 2 MB (60000 contigs) spans many BGZF members and more than the first megabyte the header parser
    copies back: dictionary and records == the host reader."""
    refs = [("contig_%05d_with_a_fairly_long_name" % i, 1000 + i) for i in range(60000)]
    recs = [W.bam_record(i * 7, 5, 30, 16 * (i & 1), [("M", 20)], b"h%d" % i) for i in range(300)]
    path = tmp_path / "big_header.bam"
    W.write_bam(path, refs, recs, level=1)
    with B.BamReader(path, index=False) as h, D.DeviceBamReader(path) as d:
        assert d.references == h.references and d.lengths == h.lengths and len(d.references) == 60000
        assert d.header_text == h.header_text
        assert _all_reads(d, 0) == _all_reads(h, 0)
