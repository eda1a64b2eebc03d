/* pymasc_amd_ingest.h -- C ABI of the DEVICE-side BAM ingest.
 *
 * libpymasc_ingest.so (pymasc_amd/csrc/ingest/bam_device.hip, hipcc, gfx950) does on the GPU what libpymasc_io.so
 * (include/pymasc_amd_io.h) does with zlib on host threads: it replaces the per-read pysam loop of the reference,
 * PyMaSC/handler/calc.py:140-153 + handler/read.py:62-155, for a whole BAM file at once:
 *
 *   host    the compressed file is read into page-locked staging buffers and copied to HBM as it is; the host only
 *           hops over the BGZF member headers (SAM spec 4.1: BSIZE, CRC32, ISIZE) -- no zlib in this library;
 *   k_bgzf_inflate   one wavefront per BGZF member: DEFLATE (RFC 1951; stored / fixed / dynamic blocks), Huffman tables
 *           and a 4-KB window of recent output in LDS, matches further back read from the member's own output in HBM;
 *   k_bgzf_crc       CRC-32 of every member's output (64 slices per member combined with x^(8n) mod P), compared with
 *           the member's footer together with ISIZE;
 *   k_bam_spec / k_bam_walk   the chain of alignment records (block_size hops) of the inflated stream, found in
 *           parallel: every 16-KB piece guesses its first record, walks to its end, and the guesses are VERIFIED against
 *           the neighbour's end (pieces whose guess was wrong are walked again until the chain closes: the result is the
 *           exact chain, not a heuristic); the walk applies the reference's read filter and extracts the fields;
 *   result  (ref_id, 1-based position, query length, strand) of the reads that pass, in file order, in device memory
 *           (pmx_dbam_device_arrays) or copied to the host (pmx_dbam_fetch): the arrays pmx_bam_next_batch yields.
 *
 * Same filter and fields as pmx_bam_next_batch (pymasc_amd_io.h), bit for bit; the host reader stays as the checker
 * (tests/test_gpu_ingest.py).  All functions return 0 or a negative PMX_IO_ERR_* code (same values as pymasc_amd_io.h);
 * the message of the calling thread's last error is pmx_dbam_last_error().  A handle is not thread-safe.
 */
#ifndef PYMASC_AMD_INGEST_H
#define PYMASC_AMD_INGEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_DBAM_OK            0
#define PMX_DBAM_ERR_OPEN     -1   /* cannot open / read the file, or out of (device) memory */
#define PMX_DBAM_ERR_FORMAT   -2   /* not a BAM file, truncated or corrupt (gzip magic, DEFLATE, CRC32, ISIZE, records) */
#define PMX_DBAM_ERR_INVALID  -3   /* bad argument */
#define PMX_DBAM_ERR_NOTFOUND -4   /* unknown chromosome (pmx_dbw_fetch) */
#define PMX_DBAM_ERR_DEVICE   -5   /* a HIP call failed */

typedef struct pmx_dbam pmx_dbam;

const char *pmx_dbam_last_error(void);
int pmx_dbam_version(void);

/* Reads `path` (a BGZF-compressed BAM file), inflates ALL of it on GPU `device`, checks every member's CRC32 and ISIZE
 * and parses the BAM header.  nthreads: host threads that read the file into the staging buffers (<= 0: up to 16).
 * The inflated stream stays in device memory until pmx_dbam_close.  Replaces pysam.AlignmentFile(path)
 * (reader/bam.py:84-126). */
int pmx_dbam_open(const char *path, int device, int nthreads, pmx_dbam **out);
void pmx_dbam_close(pmx_dbam *b);

/* Reference dictionary = pysam's AlignmentFile.references / .lengths (reader/bam.py:137-153). */
int32_t pmx_dbam_nref(const pmx_dbam *b);
const char *pmx_dbam_ref_name(const pmx_dbam *b, int32_t i);
int64_t pmx_dbam_ref_len(const pmx_dbam *b, int32_t i);
const char *pmx_dbam_header_text(const pmx_dbam *b, uint32_t *len);

/* Walks every alignment record on the device and keeps those that pass the reference's filter
 * (ReadFilter.should_skip_read, handler/read.py:62-90; reference_name None, read.py:131-133; query length 0,
 * read.py:139-141) -- exactly the records, fields and order of pmx_bam_next_batch; want_ref >= 0 keeps the records of
 * that reference only (what AlignmentFile.fetch(chrom) yields a worker, handler/worker.py:106-132).
 * Returns the number of kept records or a negative error code; may be called again with another filter. */
int64_t pmx_dbam_decode(pmx_dbam *b, uint32_t mapq_min, uint32_t flag_exclude, int32_t want_ref);

/* The kept records of the last pmx_dbam_decode in device memory: int32 ref_id[n], int32 pos1[n], int32 read_len[n],
 * uint8 reverse[n] (valid until the next decode / close). */
int pmx_dbam_device_arrays(const pmx_dbam *b, const int32_t **d_ref_id, const int32_t **d_pos1, const int32_t **d_read_len,
                           const uint8_t **d_reverse);
/* ... and copied to host arrays: records [first, first + n). */
int pmx_dbam_fetch(pmx_dbam *b, int64_t first, int64_t n, int32_t *ref_id, int32_t *pos1, int32_t *read_len, uint8_t *reverse);

/* The kept records as RUNS of one reference each, in file order (a coordinate-sorted file: one run per chromosome):
 * start[r] = index of the run's first record in the arrays above, ref_id[r], and the positions of its first and last record
 * (what the calculator's order check needs on the host; everything inside a run is checked by the device feeders).
 * Two-call protocol: with start == NULL returns the number of runs, else fills up to cap entries and returns the number
 * written.  More than 65536 runs (an unsorted file): PMX_DBAM_ERR_INVALID -- take the arrays through pmx_dbam_fetch then. */
int64_t pmx_dbam_runs(pmx_dbam *b, int64_t cap, int64_t *start, int32_t *ref_id, int32_t *first_pos1, int32_t *last_pos1);

/* Counters: alignment records walked and records kept by the last decode, uncompressed / compressed bytes of the file,
 * BGZF members, and how many 16-KB pieces had to be walked again because their guessed first record was wrong. */
int pmx_dbam_counters(const pmx_dbam *b, uint64_t *records, uint64_t *kept, uint64_t *bytes_out, uint64_t *bytes_in,
                      uint64_t *members, uint64_t *rewalked);
/* Wall-clock seconds of the phases of open + the last decode: [0] file -> HBM (read + copies + member scan), [1] inflate,
 * [2] CRC32, [3] header, [4] record chain (guess + walk + verification), [5] filter + field extraction + write. */
int pmx_dbam_timings(const pmx_dbam *b, double t[6]);

/* Test hook: the inflated stream (bytes [first, first + n)) copied to the host. */
int pmx_dbam_inflated(pmx_dbam *b, uint64_t first, uint64_t n, uint8_t *dst);

/* ---- BigWig (bbi) on the device -----------------------------------------------------------
 * What pmx_bigwig_* (pymasc_amd_io.h) does with zlib on host threads, with the data blocks inflated (the BGZF kernel, for zlib
 * streams of unknown length), Adler-32-checked and decoded by HIP kernels; the host reads the header, the chromosome B+ tree and
 * the R-tree (a few KB per chromosome).  Replaces PyMaSC/reader/bigwig.pyx:129-177 (pyBigWig). */
typedef struct pmx_dbw pmx_dbw;
int pmx_dbw_open(const char *path, int device, int nthreads, pmx_dbw **out);
void pmx_dbw_close(pmx_dbw *w);
/* Chromosome dictionary = BigWigReader.chromsizes (reader/bigwig.pyx:60-75), in the file's B+ tree order. */
int32_t pmx_dbw_nchrom(const pmx_dbw *w);
const char *pmx_dbw_chrom_name(const pmx_dbw *w, int32_t i);
int64_t pmx_dbw_chrom_len(const pmx_dbw *w, int32_t i);
/* All intervals of `chrom` in index order (ascending position for a valid file) whose float32 value is >= threshold
 * (threshold <= 0: every interval), as BigWigReader.fetch yields them (bigwig.pyx:147-177) -- the intervals, values and order
 * of pmx_bigwig_fetch.  They stay in device memory (begin[n], end[n] uint32, value[n] float32; every fetch has arrays of its own,
 * valid until close).  Returns n, PMX_DBAM_ERR_NOTFOUND for an unknown chromosome, or another negative error code. */
int64_t pmx_dbw_fetch(pmx_dbw *w, const char *chrom, float threshold);
int pmx_dbw_device_arrays(const pmx_dbw *w, const uint32_t **d_begin, const uint32_t **d_end, const float **d_value);
/* 1 when the intervals of the last fetch are non-empty, ascending and disjoint (begin_i < end_i <= begin_(i+1): BigWig order) */
int pmx_dbw_sorted(const pmx_dbw *w);
int pmx_dbw_copy(pmx_dbw *w, int64_t first, int64_t n, uint32_t *begin, uint32_t *end, float *value);

#ifdef __cplusplus
}
#endif
#endif
