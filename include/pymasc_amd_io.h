/* pymasc_amd_io.h -- C ABI of the host-side readers that feed the MI355X calculator.
 *
 * libpymasc_io.so is plain host code (C++17 + zlib + threads, no HIP): it turns the two input formats of the
 * reference into the flat arrays the GPU entry points of pymasc_amd.h take --
 *
 *   BAM  -> (ref_id, 1-based position, query length, strand) of the reads that pass the reference's filter,
 *           in file order; consumed by pmx_bits_set_positions[_dev] after the calculator's duplicate rules.
 *           Replaces the per-read pysam loop of PyMaSC/handler/calc.py:140-153 + handler/read.py:62-155.
 *   BigWig -> (begin, end, value) intervals of one chromosome with value >= threshold; consumed by
 *           pmx_bits_set_regions[_dev].  Replaces PyMaSC/reader/bigwig.pyx:147-177 (BigWigReader.fetch, itself a
 *           wrapper over the absent third-party libBigWig submodule).
 *
 * All functions return PMX_IO_OK (0) or a negative error code unless stated; the message of the last error of the
 * calling thread is pmx_io_last_error().  Handles are not thread-safe; the library runs its own worker threads.
 */
#ifndef PYMASC_AMD_IO_H
#define PYMASC_AMD_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_IO_OK            0
#define PMX_IO_ERR_OPEN     -1   /* cannot open / map the file */
#define PMX_IO_ERR_FORMAT   -2   /* not the expected format, truncated or corrupt (bad magic, CRC, sizes) */
#define PMX_IO_ERR_INVALID  -3   /* bad argument */
#define PMX_IO_ERR_NOTFOUND -4   /* unknown chromosome */

const char *pmx_io_last_error(void);
int pmx_io_version(void);

/* ---- BAM (SAM spec v1 section 4; BGZF section 4.1) -------------------------------------------------------- */
typedef struct pmx_bam pmx_bam;

/* Opens and memory-maps a BGZF-compressed BAM file and parses its header.  nthreads <= 0: one per core (max 16). */
int pmx_bam_open(const char *path, int nthreads, pmx_bam **out);
void pmx_bam_close(pmx_bam *b);

/* Reference dictionary = pysam's AlignmentFile.references / .lengths (reader/bam.py:137-153). */
int32_t pmx_bam_nref(const pmx_bam *b);
const char *pmx_bam_ref_name(const pmx_bam *b, int32_t i);
int64_t pmx_bam_ref_len(const pmx_bam *b, int32_t i);
/* The SAM header text (not NUL-terminated in the file; a terminator is appended here). */
const char *pmx_bam_header_text(const pmx_bam *b, uint32_t *len);

/* SAM flag bits the reference's filter looks at (handler/read.py:62-90) */
#define PMX_BAM_FLAG_UNMAPPED  0x4u
#define PMX_BAM_FLAG_REVERSE   0x10u
#define PMX_BAM_FLAG_READ2     0x80u
#define PMX_BAM_FLAG_DUPLICATE 0x400u
#define PMX_BAM_DEFAULT_EXCLUDE (PMX_BAM_FLAG_READ2 | PMX_BAM_FLAG_UNMAPPED | PMX_BAM_FLAG_DUPLICATE)

/* Next batch of alignment records in file order that pass the filter:
 *   skipped:  flag & flag_exclude, mapq < mapq_min          (ReadFilter.should_skip_read, read.py:62-90)
 *             ref_id < 0 (reference_name is None, read.py:131-133)
 *             query length 0 (infer_query_length() is None, read.py:139-141)
 *   pos1[i]     = reference_start + 1                                            (read.py:138)
 *   read_len[i] = sum of the CIGAR operations M, I, S, =, X (pysam infer_query_length; a CIGAR moved to the CG:B,I
 *                 tag because it has > 65535 operations is followed there, as htslib does on reading)
 *   reverse[i]  = flag & 0x10 != 0                                               (read.py:144)
 * Returns the number of records written (<= cap), 0 at end of file, or a negative error code.  The filter
 * arguments must not change between calls on one handle. */
int64_t pmx_bam_next_batch(pmx_bam *b, uint32_t mapq_min, uint32_t flag_exclude, int64_t cap,
                           int32_t *ref_id, int32_t *pos1, int32_t *read_len, uint8_t *reverse);

/* .bai index (SAM spec 5.2): what the reference's multi-process mode requires (reader/bam.py:246-262) and uses
 * through pysam's fetch(chrom) (handler/worker.py:106-132).  After pmx_bam_fetch_ref, pmx_bam_next_batch yields
 * the records of that reference only (same filter, same fields) and returns 0 at the end of the reference; it
 * may be called again for another reference, or with ref_id = -1 to go back to one pass over the whole file
 * (no index needed for that).  A reference without records is an empty range, not an error. */
int pmx_bam_index_load(pmx_bam *b, const char *bai_path);
int pmx_bam_has_index(const pmx_bam *b);
int pmx_bam_fetch_ref(pmx_bam *b, int32_t ref_id);

/* Counters since open: alignment records decoded, records that passed the filter, uncompressed bytes inflated,
 * compressed bytes consumed. */
int pmx_bam_counters(const pmx_bam *b, uint64_t *records, uint64_t *kept, uint64_t *bytes_out, uint64_t *bytes_in);

/* ---- BigWig (bbi) ----------------------------------------------------------------------------------------- */
typedef struct pmx_bigwig pmx_bigwig;

int pmx_bigwig_open(const char *path, pmx_bigwig **out);
void pmx_bigwig_close(pmx_bigwig *w);

/* Chromosome dictionary = BigWigReader.chromsizes (reader/bigwig.pyx:60-75), in the file's B+ tree order. */
int32_t pmx_bigwig_nchrom(const pmx_bigwig *w);
const char *pmx_bigwig_chrom_name(const pmx_bigwig *w, int32_t i);
int64_t pmx_bigwig_chrom_len(const pmx_bigwig *w, int32_t i);

/* All intervals of `chrom` in ascending order whose float32 value is >= threshold (threshold <= 0: every interval),
 * as BigWigReader.fetch yields them (bigwig.pyx:147-177): begin 0-based inclusive, end exclusive.
 * Two-call protocol: with begin == NULL returns the number of intervals; otherwise fills up to cap entries and
 * returns the number written.  value may be NULL.  PMX_IO_ERR_NOTFOUND for an unknown chromosome (the
 * reference raises KeyError). */
int64_t pmx_bigwig_fetch(pmx_bigwig *w, const char *chrom, float threshold, int64_t cap,
                         uint32_t *begin, uint32_t *end, float *value);

#ifdef __cplusplus
}
#endif
#endif
