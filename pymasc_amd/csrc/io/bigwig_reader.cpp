// BigWig (bbi) -> (begin, end, value) intervals of one chromosome for the MI355X calculator.
//
// Replaces PyMaSC/reader/bigwig.pyx:147-177 (BigWigReader.fetch: pyBigWig's intervals(chrom, 0, size) filtered by
// value >= threshold) and :129-145 (chromsizes).  pyBigWig / libBigWig are absent from this image, so this is an
// independent reader of the published bbi layout (Kent et al. 2010, Bioinformatics 26:2204, supplementary tables):
//
//   64-byte header -> chromosome B+ tree (name -> id, size) -> R-tree index over the data blocks ->
//   data blocks (zlib-compressed when uncompressBufSize > 0): 24-byte section header + bedGraph / variableStep /
//   fixedStep items.
//
// The blocks of the wanted chromosome are found by walking the R-tree in order, inflated and decoded in parallel,
// and concatenated in index order, which is ascending position for a valid file.  Parity is pinned on the
// reference's own twin files tests/data/hg19_36mer-test.{bigwig,bedGraph}.
#include "../../../include/pymasc_amd_io.h"
#include "io_common.h"

#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr uint32_t BIGWIG_MAGIC = 0x888FFC26u;
constexpr uint32_t CHROM_TREE_MAGIC = 0x78CA8C91u;
constexpr uint32_t RTREE_MAGIC = 0x2468ACE0u;

struct Span {
    uint64_t offset, size;
};

struct Interval {
    uint32_t begin, end;
    float value;
};

}  // namespace

struct pmx_bigwig {
    pmx_io::MappedFile file;
    uint16_t version = 0;
    uint64_t chrom_tree_off = 0, data_off = 0, index_off = 0;
    uint32_t uncompress_buf = 0;
    std::vector<std::string> names;     // in B+ tree order
    std::vector<uint32_t> ids;
    std::vector<int64_t> sizes;
    int nthreads = 1;
    // result of the last counting call, handed out by the filling call that follows it
    std::string cache_chrom;
    float cache_threshold = 0;
    bool cache_valid = false;
    std::vector<Interval> cache;
};

namespace {

struct Cursor {
    const pmx_bigwig &w;
    const uint8_t *at(uint64_t off, uint64_t n) const
    {
        if (off > w.file.size || n > w.file.size - off)
            throw pmx_io::Error(PMX_IO_ERR_FORMAT, "BigWig structure points past the end of the file");
        return w.file.data + off;
    }
    uint8_t u8(uint64_t off) const { return *at(off, 1); }
    uint16_t u16(uint64_t off) const { uint16_t v; memcpy(&v, at(off, 2), 2); return v; }
    uint32_t u32(uint64_t off) const { uint32_t v; memcpy(&v, at(off, 4), 4); return v; }
    uint64_t u64(uint64_t off) const { uint64_t v; memcpy(&v, at(off, 8), 8); return v; }
};

void walk_chrom_tree(pmx_bigwig &w, const Cursor &c, uint64_t node, uint32_t key_size, uint32_t val_size, int depth)
{
    if (depth > 32) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "chromosome tree too deep");
    const bool leaf = c.u8(node) != 0;
    const uint32_t count = c.u16(node + 2);
    uint64_t p = node + 4;
    for (uint32_t i = 0; i < count; i++) {
        const char *key = (const char *)c.at(p, key_size);
        if (leaf) {
            if (val_size < 8) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "chromosome tree value size < 8");
            w.names.emplace_back(key, strnlen(key, key_size));
            w.ids.push_back(c.u32(p + key_size));
            w.sizes.push_back((int64_t)c.u32(p + key_size + 4));
            p += key_size + val_size;
        } else {
            walk_chrom_tree(w, c, c.u64(p + key_size), key_size, val_size, depth + 1);
            p += key_size + 8;
        }
    }
}

// Collects, in index order, the data blocks whose chromosome range includes chrom id `cid`.
void walk_rtree(const Cursor &c, uint64_t node, uint32_t cid, std::vector<Span> &out, int depth)
{
    if (depth > 64) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "R-tree too deep");
    const bool leaf = c.u8(node) != 0;
    const uint32_t count = c.u16(node + 2);
    uint64_t p = node + 4;
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t c0 = c.u32(p), c1 = c.u32(p + 8);
        const bool hit = c0 <= cid && cid <= c1;
        if (leaf) {
            if (hit) out.push_back(Span{c.u64(p + 16), c.u64(p + 24)});
            p += 32;
        } else {
            if (hit) walk_rtree(c, c.u64(p + 16), cid, out, depth + 1);
            p += 24;
        }
    }
}

void decode_block(const pmx_bigwig &w, const Span &sp, uint32_t cid, int64_t chrom_len, float threshold,
                  std::vector<Interval> &out)
{
    const Cursor c{w};
    const uint8_t *raw = c.at(sp.offset, sp.size);
    std::vector<uint8_t> tmp;
    const uint8_t *d = raw;
    size_t n = sp.size;
    if (w.uncompress_buf > 0) {
        tmp.resize(w.uncompress_buf);
        uLongf dl = tmp.size();
        if (uncompress(tmp.data(), &dl, raw, sp.size) != Z_OK)
            throw pmx_io::Error(PMX_IO_ERR_FORMAT, "BigWig data block does not inflate");
        d = tmp.data();
        n = dl;
    }
    if (n < 24) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "BigWig data block shorter than its header");
    uint32_t chrom, start, step, span;
    memcpy(&chrom, d, 4);
    memcpy(&start, d + 4, 4);
    memcpy(&step, d + 12, 4);
    memcpy(&span, d + 16, 4);
    const uint8_t type = d[20];
    uint16_t count;
    memcpy(&count, d + 22, 2);
    if (chrom != cid) return;      // an index entry may straddle chromosomes; blocks themselves never do
    const size_t item = type == 1 ? 12 : type == 2 ? 8 : type == 3 ? 4 : 0;
    if (!item) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "BigWig data block of unknown type");
    if (24 + (size_t)count * item > n) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "BigWig data block shorter than its items");
    const uint8_t *p = d + 24;
    for (uint32_t i = 0; i < count; i++, p += item) {
        Interval iv;
        if (type == 1) {
            memcpy(&iv.begin, p, 4);
            memcpy(&iv.end, p + 4, 4);
            memcpy(&iv.value, p + 8, 4);
        } else if (type == 2) {
            memcpy(&iv.begin, p, 4);
            memcpy(&iv.value, p + 4, 4);
            iv.end = iv.begin + span;
        } else {
            iv.begin = start + i * step;
            iv.end = iv.begin + span;
            memcpy(&iv.value, p, 4);
        }
        // intervals(chrom, 0, chrom_len): entries overlapping the chromosome's extent
        if ((int64_t)iv.begin >= chrom_len || iv.end == 0) continue;
        if (threshold > 0 && !(iv.value >= threshold)) continue;
        out.push_back(iv);
    }
}

void open_impl(pmx_bigwig &w, const char *path)
{
    w.file.open(path);
    const Cursor c{w};
    const uint32_t magic = c.u32(0);
    if (magic == __builtin_bswap32(BIGWIG_MAGIC))
        throw pmx_io::Error(PMX_IO_ERR_FORMAT, "byte-swapped (big-endian) BigWig files are not supported");
    if (magic != BIGWIG_MAGIC) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "not a BigWig file (bad magic)");
    w.version = c.u16(4);
    w.chrom_tree_off = c.u64(8);
    w.data_off = c.u64(16);
    w.index_off = c.u64(24);
    w.uncompress_buf = c.u32(52);
    if (w.uncompress_buf > (1u << 30)) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "implausible uncompressBufSize");
    // chromosome B+ tree
    const uint64_t t = w.chrom_tree_off;
    if (c.u32(t) != CHROM_TREE_MAGIC) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "bad chromosome tree magic");
    const uint32_t key_size = c.u32(t + 8), val_size = c.u32(t + 12);
    const uint64_t item_count = c.u64(t + 16);
    if (key_size == 0 || key_size > 4096) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "bad chromosome tree key size");
    if (item_count) walk_chrom_tree(w, c, t + 32, key_size, val_size, 0);
    if (c.u32(w.index_off) != RTREE_MAGIC) throw pmx_io::Error(PMX_IO_ERR_FORMAT, "bad R-tree index magic");
    w.nthreads = pmx_io::pick_threads(0);
}

}  // namespace

extern "C" {

int pmx_bigwig_open(const char *path, pmx_bigwig **out)
{
    if (!path || !out) return pmx_io::fail(PMX_IO_ERR_INVALID, "pmx_bigwig_open: NULL argument");
    *out = nullptr;
    pmx_bigwig *w = new pmx_bigwig();
    try {
        open_impl(*w, path);
    } catch (const pmx_io::Error &e) {
        delete w;
        return pmx_io::fail(e.code, std::string(path) + ": " + e.msg);
    } catch (const std::exception &e) {
        delete w;
        return pmx_io::fail(PMX_IO_ERR_OPEN, std::string(path) + ": " + e.what());
    }
    *out = w;
    return PMX_IO_OK;
}

void pmx_bigwig_close(pmx_bigwig *w) { delete w; }

int32_t pmx_bigwig_nchrom(const pmx_bigwig *w) { return w ? (int32_t)w->names.size() : 0; }

const char *pmx_bigwig_chrom_name(const pmx_bigwig *w, int32_t i)
{
    if (!w || i < 0 || (size_t)i >= w->names.size()) return nullptr;
    return w->names[i].c_str();
}

int64_t pmx_bigwig_chrom_len(const pmx_bigwig *w, int32_t i)
{
    if (!w || i < 0 || (size_t)i >= w->sizes.size()) return -1;
    return w->sizes[i];
}

int64_t pmx_bigwig_fetch(pmx_bigwig *w, const char *chrom, float threshold, int64_t cap, uint32_t *begin,
                         uint32_t *end, float *value)
{
    if (!w || !chrom) return pmx_io::fail(PMX_IO_ERR_INVALID, "pmx_bigwig_fetch: NULL argument");
    if (begin && (!end || cap < 0)) return pmx_io::fail(PMX_IO_ERR_INVALID, "pmx_bigwig_fetch: end is NULL or cap < 0");
    size_t k = 0;
    while (k < w->names.size() && w->names[k] != chrom) k++;
    if (k == w->names.size()) return pmx_io::fail(PMX_IO_ERR_NOTFOUND, std::string("unknown chromosome: ") + chrom);
    try {
        if (!(w->cache_valid && w->cache_chrom == chrom && w->cache_threshold == threshold)) {
            const Cursor c{*w};
            std::vector<Span> spans;
            walk_rtree(c, w->index_off + 48, w->ids[k], spans, 0);
            std::vector<std::vector<Interval>> parts(spans.size());
            const uint32_t cid = w->ids[k];
            const int64_t clen = w->sizes[k];
            pmx_io::parallel_for(w->nthreads, spans.size(), 8, [&](size_t lo, size_t hi, size_t) {
                for (size_t i = lo; i < hi; i++) decode_block(*w, spans[i], cid, clen, threshold, parts[i]);
            });
            size_t total = 0;
            for (auto &v : parts) total += v.size();
            w->cache.clear();
            w->cache.reserve(total);
            for (auto &v : parts) w->cache.insert(w->cache.end(), v.begin(), v.end());
            w->cache_chrom = chrom;
            w->cache_threshold = threshold;
            w->cache_valid = true;
        }
        if (!begin) return (int64_t)w->cache.size();
        const size_t n = std::min<size_t>((size_t)cap, w->cache.size());
        for (size_t i = 0; i < n; i++) {
            begin[i] = w->cache[i].begin;
            end[i] = w->cache[i].end;
            if (value) value[i] = w->cache[i].value;
        }
        w->cache_valid = false;
        std::vector<Interval>().swap(w->cache);
        return (int64_t)n;
    } catch (const pmx_io::Error &e) {
        return pmx_io::fail(e.code, e.msg);
    } catch (const std::exception &e) {
        return pmx_io::fail(PMX_IO_ERR_FORMAT, e.what());
    }
}

}  // extern "C"
