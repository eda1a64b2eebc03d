"""TEST INFRASTRUCTURE: minimal writers of the two input formats, used to build synthetic BAM / BigWig files for the
native readers' tests and the ingest benchmark (pysam / pyBigWig are not installed).  They follow the published
layouts (SAM/BAM spec v1 section 4; bbi: Kent et al. 2010 supplement) and know nothing about the readers."""
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"


# ---------------------------------------------------------------- BGZF / BAM ------------------------------------
def bgzf_block(payload: bytes, level: int = 6) -> bytes:
    assert len(payload) <= 65536
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(cdata) + 8 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


BGZF_EOF = bgzf_block(b"")


def bgzf_compress(data: bytes, block: int = 0xff00, level: int = 6, eof: bool = True) -> bytes:
    out = [bgzf_block(data[i:i + block], level) for i in range(0, len(data), block)]
    if eof:
        out.append(BGZF_EOF)
    return b"".join(out)


def bam_header(refs, text=None) -> bytes:
    if text is None:
        text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:{}\tLN:{}\n".format(n, l) for n, l in refs)
    t = text.encode()
    out = [b"BAM\1", struct.pack("<i", len(t)), t, struct.pack("<i", len(refs))]
    for n, l in refs:
        nb = n.encode() + b"\0"
        out += [struct.pack("<i", len(nb)), nb, struct.pack("<i", l)]
    return b"".join(out)


def bam_record(ref, pos0, mapq, flag, cigar, name=b"r", l_seq=None, tags=b"") -> bytes:
    """cigar: list of (op_char, length)."""
    cig = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in cigar)
    if l_seq is None:
        l_seq = sum(n for op, n in cigar if op in "MIS=X")
    nm = name + b"\0"
    body = (struct.pack("<iiBBHHHiiii", ref, pos0, len(nm), mapq, 4680, len(cigar), flag, l_seq, -1, -1, 0)
            + nm + cig + b"\x11" * ((l_seq + 1) // 2) + b"\x20" * l_seq + tags)
    return struct.pack("<i", len(body)) + body


def long_cigar_record(ref, pos0, mapq, flag, cigar, name=b"long") -> bytes:
    """A record whose real CIGAR lives in the CG:B,I tag behind the <l_seq>S<ref_len>N placeholder (spec 4.2.2)."""
    l_seq = sum(n for op, n in cigar if op in "MIS=X")
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    real = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in cigar)
    tags = b"NMi" + struct.pack("<i", 0) + b"XZZhello\0" + b"CGBI" + struct.pack("<I", len(cigar)) + real
    return bam_record(ref, pos0, mapq, flag, [("S", l_seq), ("N", ref_len)], name, l_seq, tags)


def write_bam(path, refs, records, block: int = 0xff00, level: int = 6, eof: bool = True, text=None):
    """records: iterable of bytes from bam_record()."""
    data = bam_header(refs, text) + b"".join(records)
    with open(path, "wb") as fp:
        fp.write(bgzf_compress(data, block, level, eof))


def synth_bam_records(rng, refs, n_per_ref, readlen=36, mapq_lo=0, mapq_hi=60):
    """Coordinate-sorted single-end records; returns (list of record bytes, arrays ref, pos0, mapq, flag, qlen)."""
    recs, meta = [], []
    for rid, (_n, ln) in enumerate(refs):
        pos = np.sort(rng.integers(0, ln - readlen - 1, size=n_per_ref))
        mq = rng.integers(mapq_lo, mapq_hi + 1, size=n_per_ref)
        fl = np.where(rng.random(n_per_ref) < 0.5, 16, 0)
        fl = np.where(rng.random(n_per_ref) < 0.02, fl | 0x400, fl)
        fl = np.where(rng.random(n_per_ref) < 0.02, fl | 0x80 | 0x1, fl)
        ql = np.where(rng.random(n_per_ref) < 0.1, readlen - 1, readlen)
        for p, q, f, l in zip(pos.tolist(), mq.tolist(), fl.tolist(), ql.tolist()):
            recs.append(bam_record(rid, p, q, f, [("M", l)], b"read%d" % len(recs)))
            meta.append((rid, p, q, f, l))
    m = np.array(meta, dtype=np.int64).reshape(-1, 5)
    return recs, m


# ---------------------------------------------------------------- BigWig ----------------------------------------
def _bpt(chroms, block_size):
    """Chromosome B+ tree: chroms = [(name, id, size)] sorted by name.  Returns bytes (header + nodes)."""
    key = max(len(n) for n, _, _ in chroms) + 1
    hdr = struct.pack("<IIIIQQ", 0x78CA8C91, block_size, key, 8, len(chroms), 0)
    # leaves
    level = []                                                   # (first key, node bytes)
    for i in range(0, len(chroms), block_size):
        items = chroms[i:i + block_size]
        node = struct.pack("<BBH", 1, 0, len(items))
        for n, cid, sz in items:
            node += n.encode().ljust(key, b"\0") + struct.pack("<II", cid, sz)
        level.append((items[0][0], node))
    levels = [level]
    while len(levels[-1]) > 1:
        prev, cur = levels[-1], []
        for i in range(0, len(prev), block_size):
            cur.append((prev[i][0], prev[i:i + block_size]))     # children resolved below
        levels.append(cur)
    # lay out top-down, children after parents
    base = 32
    order = []                                                   # (level index, node index)
    sizes = {}
    for li in range(len(levels) - 1, -1, -1):
        for ni, (_k, node) in enumerate(levels[li]):
            sizes[(li, ni)] = len(node) if li == 0 else 4 + len(node) * (key + 8)
            order.append((li, ni))
    offs, p = {}, base
    for k in order:
        offs[k] = p
        p += sizes[k]
    out = b""
    for li, ni in order:
        if li == 0:
            out += levels[0][ni][1]
        else:
            kids = levels[li][ni][1]
            node = struct.pack("<BBH", 0, 0, len(kids))
            first_child = ni * block_size
            for j, (k, _n) in enumerate(kids):
                node += k.encode().ljust(key, b"\0") + struct.pack("<Q", offs[(li - 1, first_child + j)])
            out += node
    return hdr, out, offs


def write_bigwig(path, chromsizes, tracks, kind="bedgraph", compress=True, items_per_block=64, rtree_block=4,
                 bpt_block=3, span=1, step=1):
    """tracks: {chrom: [(begin, end, value)]} ascending.  kind: bedgraph | varstep (end = begin + span) |
    fixedstep (consecutive items at begin0 + i*step, width span; runs are split where the pattern breaks)."""
    names = sorted(chromsizes)
    chroms = [(n, i, chromsizes[n]) for i, n in enumerate(names)]
    cid = {n: i for n, i, _ in chroms}
    # data blocks
    blocks = []        # (chrom id, start, end, payload)
    for n in names:
        ivs = tracks.get(n, [])
        i = 0
        while i < len(ivs):
            chunk = ivs[i:i + items_per_block]
            if kind == "fixedstep":
                k = 1
                while k < len(chunk) and chunk[k][0] == chunk[0][0] + k * step:
                    k += 1
                chunk = chunk[:k]
            i += len(chunk)
            b0, e1 = chunk[0][0], chunk[-1][1]
            if kind == "bedgraph":
                body = b"".join(struct.pack("<IIf", b, e, v) for b, e, v in chunk)
                hdr = struct.pack("<IIIIIBBH", cid[n], b0, e1, 0, 0, 1, 0, len(chunk))
            elif kind == "varstep":
                assert all(e - b == span for b, e, _ in chunk)
                body = b"".join(struct.pack("<If", b, v) for b, _e, v in chunk)
                hdr = struct.pack("<IIIIIBBH", cid[n], b0, e1, 0, span, 2, 0, len(chunk))
            else:
                assert all(e - b == span for b, e, _ in chunk)
                body = b"".join(struct.pack("<f", v) for _b, _e, v in chunk)
                hdr = struct.pack("<IIIIIBBH", cid[n], b0, e1, step, span, 3, 0, len(chunk))
            blocks.append((cid[n], b0, e1, hdr + body))
    raw_max = max([len(b[3]) for b in blocks] + [0])
    bpt_hdr, bpt_nodes, _ = _bpt(chroms, bpt_block)
    chrom_tree_off = 64
    # the tree's child offsets are relative to the file: rebuild with the right base
    key = max(len(n) for n in names) + 1

    def shift_offsets(nodes: bytes) -> bytes:
        out, p = bytearray(nodes), 0
        while p < len(nodes):
            leaf, _r, cnt = struct.unpack_from("<BBH", nodes, p)
            p += 4
            for _ in range(cnt):
                if leaf:
                    p += key + 8
                else:
                    off, = struct.unpack_from("<Q", nodes, p + key)
                    struct.pack_into("<Q", out, p + key, off + chrom_tree_off)
                    p += key + 8
        return bytes(out)

    bpt = bpt_hdr + shift_offsets(bpt_nodes)
    data_off = chrom_tree_off + len(bpt)
    data = struct.pack("<Q", len(blocks))
    leaves = []        # (c0, s0, c1, e1, offset, size)
    for c, b0, e1, payload in blocks:
        z = zlib.compress(payload) if compress else payload
        leaves.append((c, b0, c, e1, data_off + len(data), len(z)))
        data += z
    index_off = data_off + len(data)
    # R-tree, bottom-up
    levels = [[leaves[i:i + rtree_block] for i in range(0, len(leaves), rtree_block)] or [[]]]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([list(range(i, min(i + rtree_block, len(prev)))) for i in range(0, len(prev), rtree_block)])

    def bounds(li, ni):
        if li == 0:
            items = levels[0][ni]
            if not items:
                return (0, 0, 0, 0)
            return (items[0][0], items[0][1], items[-1][2], max(x[3] for x in items if x[2] == items[-1][2]))
        kids = [bounds(li - 1, k) for k in levels[li][ni]]
        return (kids[0][0], kids[0][1], kids[-1][2], kids[-1][3])

    order, sizes = [], {}
    for li in range(len(levels) - 1, -1, -1):
        for ni, node in enumerate(levels[li]):
            sizes[(li, ni)] = 4 + len(node) * (32 if li == 0 else 24)
            order.append((li, ni))
    offs, p = {}, index_off + 48
    for k in order:
        offs[k] = p
        p += sizes[k]
    rt = b""
    for li, ni in order:
        node = levels[li][ni]
        if li == 0:
            rt += struct.pack("<BBH", 1, 0, len(node))
            for c0, s0, c1, e1, off, sz in node:
                rt += struct.pack("<IIIIQQ", c0, s0, c1, e1, off, sz)
        else:
            rt += struct.pack("<BBH", 0, 0, len(node))
            for k in node:
                c0, s0, c1, e1 = bounds(li - 1, k)
                rt += struct.pack("<IIIIQ", c0, s0, c1, e1, offs[(li - 1, k)])
    top = bounds(len(levels) - 1, 0)
    rt_hdr = struct.pack("<IIQIIIIQII", 0x2468ACE0, rtree_block, len(leaves), top[0], top[1], top[2], top[3],
                         index_off, items_per_block, 0)
    header = struct.pack("<IHHQQQHHQQIQ", 0x888FFC26, 4, 0, chrom_tree_off, data_off, index_off, 0, 0, 0, 0,
                         (raw_max if compress else 0), 0)
    assert len(header) == 64 and len(rt_hdr) == 48
    with open(path, "wb") as fp:
        fp.write(header + bpt + data + rt_hdr + rt + struct.pack("<I", 0x888FFC26))


def write_bam_indexed(path, refs, records, rec_refs, block: int = 0xff00, level: int = 6, pseudo_bin: bool = True,
                      text=None):
    """write_bam + a .bai next to it.  rec_refs[i] = reference id of records[i] (coordinate-sorted, -1 last).

    The index holds, per reference with records, one real bin (bin 0, a single chunk spanning all its records), the
    linear index left empty, and optionally the samtools pseudo-bin 37450 {ref_beg, ref_end, n_mapped, n_unmapped}."""
    head = bam_header(refs, text)
    data = head + b"".join(records)
    blocks = [bgzf_block(data[i:i + block], level) for i in range(0, len(data), block)]
    coff = [0]
    for b in blocks:
        coff.append(coff[-1] + len(b))
    with open(path, "wb") as fp:
        fp.write(b"".join(blocks) + BGZF_EOF)

    def voff(u):          # uncompressed offset -> virtual offset (a position at a block end belongs to the next block)
        k, r = divmod(u, block)
        return (coff[k] << 16) | r

    u = len(head)
    spans = {}
    for rec, rid in zip(records, rec_refs):
        if rid >= 0:
            beg, _ = spans.get(rid, (u, u))
            spans[rid] = (beg, u + len(rec))
        u += len(rec)
    out = [b"BAI\1", struct.pack("<i", len(refs))]
    for rid in range(len(refs)):
        if rid not in spans:
            out.append(struct.pack("<ii", 0, 0))
            continue
        vb, ve = voff(spans[rid][0]), voff(spans[rid][1])
        bins = [struct.pack("<Ii", 0, 1) + struct.pack("<QQ", vb, ve)]
        if pseudo_bin:
            n = sum(1 for r in rec_refs if r == rid)
            bins.append(struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", vb, ve, n, 0))
        out.append(struct.pack("<i", len(bins)) + b"".join(bins) + struct.pack("<i", 0))
    with open(str(path) + ".bai", "wb") as fp:
        fp.write(b"".join(out))
