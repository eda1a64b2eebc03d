/*
 * pymasc_amd.h -- C ABI of the MI355X (gfx950) strand cross-correlation library (libpymasc_hip.so).
 *
 * This is the drop-in boundary for ONE path of PyMaSC: the per-chromosome shifted strand
 * cross-correlation of the BitArray calculator.  Each entry point names the reference interface it
 * replaces (paths relative to the PyMaSC source tree).  A maintainer binds these with ctypes
 * (see INTEGRATION.md); no C++/torch types cross the ABI.
 *
 * Bit-vector layout = the reference's BIT_ARRAY (PyMaSC/core/bitarray/bitarray.pxd:13-17):
 * bit i lives in words[i >> 6] at (i & 63), i is the 1-based genomic position (bit 0 unused),
 * nwords = ceil(nbits / 64), bits at index >= nbits must be zero.
 * For a chromosome of length G:  nbits = G + read_len + max_shift + 100
 * (PyMaSC/core/bitarray/mscc.pyx:117,134,165-167,340-341).
 *
 * Conventions: every function returns PMX_OK (0) or a negative PMX_ERR_* code; the message of the
 * last failure on the calling thread is returned by pmx_last_error().  Pointers named d_* are
 * DEVICE pointers (HBM) of the context's GPU, h_* are host pointers.  All work is enqueued on the
 * context's HIP stream; *_dev functions are asynchronous (call pmx_ctx_sync, or order later work
 * on the same stream), functions taking host output pointers synchronise before returning.
 * A context is used by one thread at a time (the reference's calculator is single-threaded too:
 * one calculator per worker process, PyMaSC/handler/worker.py:198-204).
 */
#ifndef PYMASC_AMD_H
#define PYMASC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_OK            0
#define PMX_ERR_INVALID  -1   /* bad argument */
#define PMX_ERR_HIP      -2   /* HIP runtime / kernel launch failure */
#define PMX_ERR_NOMEM    -3   /* host or device allocation failure */
#define PMX_ERR_NODEVICE -4   /* no usable gfx950 device */

/* Rows of the result block written by pmx_cc_dev / pmx_calc_correlation.
 * The block is PMX_NROWS rows of `stride = max_shift + 1` uint64 each, indexed by shift d. */
#define PMX_ROW_NCC_CCBINS  0  /* sum_j F[j] & R[j+d]                      mscc.pyx:314        */
#define PMX_ROW_MSCC_FSUM   1  /* sum_j F[j] & D_d[j]                      mscc.pyx:300,303    */
#define PMX_ROW_MSCC_RSUM   2  /* sum_j R[j+d] & D_d[j]                    mscc.pyx:301,304    */
#define PMX_ROW_MSCC_CCBINS 3  /* sum_j F[j] & R[j+d] & D_d[j]             mscc.pyx:305        */
#define PMX_ROW_MLEN        4  /* popcount(D_d), D_d[j]=M[j]&M[j+L-1-d]    mscc.pyx:291-298    */
#define PMX_ROW_SCALARS     5  /* [0]=popcount(F) [1]=popcount(R)          mscc.pyx:236-237
                                  [2]=popcount(M) [3]=kernel path actually used (PMX_PATH_*)    */
#define PMX_NROWS           6

/* flags for pmx_cc_dev / pmx_calc_correlation */
#define PMX_FLAG_SKIP_NCC     1u  /* skip_ncc=True (mscc.pyx:235,313): row 0 left zero (popcounts still reported) */
#define PMX_FLAG_FORCE_DENSE  2u  /* use the dense word-parallel kernels (one lane per shift)          */
#define PMX_FLAG_FORCE_SPARSE 4u  /* insist on the set-bit driven kernels (error if unsupported); no density probe:
                                   * the event kernel finds the tiles beyond its lists itself (see PMX_FLAG_EVENTS_HINT) */
#define PMX_FLAG_WINDOW_ONLY 16u  /* hint: the caller knows the vectors are dense -- per 65536 positions, forward reads +
                                   * reverse reads + 2 x mappable runs above 2416 (3304 without a track), or 2 x runs above 1536
                                   * (max_shift <= 1023: ~1.8 % read starts per strand, ~110 M reads on hg38; above 1023: 768 forward /
                                   * 1000 reverse reads / 384 run edges) --: go straight to the window kernels instead of
                                   * letting the event kernel find that out tile by tile.  Same integers either way */
#define PMX_FLAG_DEEP_LISTS  32u  /* hint: deep data, but not beyond the event kernel (max_shift <= 1023 with a track:
                                   * between ~1.5 % and ~3.2 % read starts per strand): the instantiation with the larger
                                   * list pool (4328 entries per 65536 positions, four workgroups per CU instead of
                                   * five).  Same integers either way */
#define PMX_FLAG_EVENTS_HINT 64u  /* hint: the caller knows the data is within the event kernel's lists (the ordinary case:
                                   * up to ~1.5 % read starts per strand on an ordinary track).  Without ANY of the three
                                   * hints pmx_cc_batch_dev / pmx_calc_correlation take one from a sample of the vectors
                                   * themselves (16 tiles of each of the largest chromosomes: one small launch and ONE
                                   * SYNCHRONISATION of the stream per call, ~20 us): pass a hint to stay asynchronous */
#define PMX_FLAG_SKIP_MLEN    8u  /* the caller holds mappable_len already (the *_mappability.json cache,
                                   * handler/mappability.py:239-259): no autocorrelation pass; row 4 and
                                   * scalars[2] are written as zeros */
/* default (neither): the set-bit kernels for read_len <= 1024 -- for max_shift <= 8191 the event kernel on the
 * tiles (64 Kbit) whose read / run-edge lists fit its LDS lists and the window kernel on the others (beyond 1023
 * shifts in chunks of 1024 shifts), for max_shift > 8191 the window kernel alone in chunks of 1024 shifts --, the
 * dense kernels for longer reads.  The integers are the same whichever kernel produced them.
 * Environment switches read once per process (A/B measurements, tests): PMX_CC_EVENTS=0 (window kernel on every
 * tile), PMX_CC_FUSE_MLEN=0 (mappable-length pass not fused into the event kernel), PMX_AUTOCORR_PAIRS=0,
 * PMX_AUTOCORR_FORK=0.
 * Limits of every entry point below: max_shift <= 65535 (the reference takes any -d, mscc.pyx:288, default 1000;
 * larger values return PMX_ERR_INVALID), read_len <= 65535, nbits < 2^40.
 * max_shift < 3: a row of max_shift + 1 words cannot hold the four scalars, so the scalar row is cut to its first
 * max_shift + 1 entries (popcount(F)[, popcount(R)[, popcount(M)]]); all other rows are complete. */

#define PMX_PATH_DENSE  1
#define PMX_PATH_SPARSE 2

typedef struct pmx_ctx pmx_ctx;

/* ---- library / context ------------------------------------------------------------------- */
const char *pmx_last_error(void);
int pmx_version(void);
/* sha256 (16 hex digits) of the sources this library was built from (pymasc_amd/build.py: source_hash): the profile
 * summaries under profiles/ name the build they were measured on, bench.py quotes them only for that build. */
const char *pmx_build_id(void);
int pmx_device_count(int *n);

/* One context per worker process / per GPU (replaces nothing in the reference: it owns the HIP
 * stream, scratch memory and kernel timers).  `hip_stream` may be NULL (the context creates its
 * own stream) or an existing hipStream_t, e.g. torch.cuda.current_stream().cuda_stream. */
int pmx_ctx_create(int device, void *hip_stream, pmx_ctx **out);
int pmx_ctx_destroy(pmx_ctx *ctx);
int pmx_ctx_sync(pmx_ctx *ctx);

/* ---- device bit-vectors: replace bit_array_create/free/clear_all/set_bit/set_region/num_bits_set
 *      (bitarray.pxd:20-36; bitarray.pyx:40-50,72-107) ------------------------------------ */
int pmx_bits_alloc(pmx_ctx *ctx, uint64_t nbits, uint64_t **d_words);      /* zero-filled */
int pmx_bits_free(pmx_ctx *ctx, uint64_t *d_words);
int pmx_bits_clear(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits);
int pmx_bits_upload(pmx_ctx *ctx, uint64_t *d_words, const uint64_t *h_words, uint64_t nbits);
int pmx_bits_download(pmx_ctx *ctx, const uint64_t *d_words, uint64_t *h_words, uint64_t nbits);
/* bitarray[pos] = 1 for every pos (feed_forward_read mscc.pyx:393, feed_reverse_read :416-417);
 * positions are bit indices, duplicates allowed; any pos >= nbits or < 0 -> PMX_ERR_INVALID (host variant: the range
 * check runs on the device, so the bits of the in-range positions have been set when the error is reported; dev
 * variant: out-of-range positions are dropped). */
int pmx_bits_set_positions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits,
                           const int64_t *h_pos, uint64_t n);
int pmx_bits_set_positions_dev(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits,
                               const int64_t *d_pos, uint64_t n);
/* bitarray.set(from, to), both ends inclusive (bitarray.pyx:88-95); the mappability loader calls
 * it as set(begin + 1, end) per BigWig interval (mscc.pyx:343-344).  Intervals with to < from are
 * ignored; to >= nbits -> PMX_ERR_INVALID (host variant) / clipped (dev variant). */
int pmx_bits_set_regions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits,
                         const int64_t *h_from, const int64_t *h_to, uint64_t n);
int pmx_bits_set_regions_dev(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits,
                             const int64_t *d_from, const int64_t *d_to, uint64_t n);
/* bit_array_num_bits_set (bitarray.pyx:101-107) */
int pmx_bits_count(pmx_ctx *ctx, const uint64_t *d_words, uint64_t nbits, uint64_t *h_count);

/* ---- stream-ordered feeding --------------------------------------------------------------------------
 * The calls above that take host arrays synchronise (they hand an error back); a genome fed through them costs one
 * round trip per vector.  The calls below only ENQUEUE -- copies, builder kernels and the reference's per-read rules run
 * on the context's stream, errors are recorded on the device and read later -- so the vectors of chromosome k are built
 * while the host decodes chromosome k + 1 and nothing waits until the results are fetched.
 * Host arrays: pageable memory may be reused as soon as a call returns (HIP stages it); pinned memory (pmx_host_alloc)
 * is copied asynchronously and must stay untouched until the stream has passed the call (pmx_ctx_sync, or any
 * synchronising call). */
int pmx_host_alloc(pmx_ctx *ctx, uint64_t bytes, void **h);   /* page-locked host memory */
int pmx_host_free(pmx_ctx *ctx, void *h);

/* Feed state of ONE chromosome: PMX_FEED_WORDS uint64 in device memory, cleared (pmx_bits_clear) when the chromosome
 * starts (CCBitArrayCalculator._init_buff, mscc.pyx:161-171), downloaded with pmx_bits_download when its sums are needed. */
#define PMX_FEED_FORWARD_LEN_SUM     0   /* _forward_read_len_sum of the chromosome                 mscc.pyx:392       */
#define PMX_FEED_REVERSE_LEN_SUM     1   /* _reverse_read_len_sum                                   mscc.pyx:418       */
#define PMX_FEED_FORWARD_KEPT        2   /* forward reads that were not duplicates (= forward bits set by the feed)    */
#define PMX_FEED_REVERSE_KEPT        3   /* reverse reads that found their bit clear                                   */
#define PMX_FEED_FIRST_UNSORTED      4   /* 0, or PMX_FEED_ERR_BASE - index of the first read (counted over all calls since
                                            the clear) below its predecessor: ReadUnsortedError, mscc.pyx:362-363     */
#define PMX_FEED_FIRST_OUT_OF_RANGE  5   /* 0, or PMX_FEED_ERR_BASE - index of the first read / interval whose bit(s) fall
                                            outside [0, nbits) (dropped / clipped; the reference writes unchecked)    */
#define PMX_FEED_LAST_POS            6   /* _last_pos                                               mscc.pyx:365       */
#define PMX_FEED_LAST_FORWARD_POS    7   /* _last_forward_pos                                       mscc.pyx:390       */
#define PMX_FEED_READS               8   /* reads fed since the clear                                                  */
#define PMX_FEED_MAX_REVERSE_LEN     9   /* longest reverse read seen (look-back bound of the reverse rule)            */
#define PMX_FEED_CHUNK_FORWARD_POS  10   /* internal: 1 + last forward position of the call in progress                */
#define PMX_FEED_REGIONS_UNSORTED   11   /* 0, or PMX_FEED_ERR_BASE - index of the first interval that breaks the order
                                           * PMX_REGIONS_SORTED promises (pmx_bits_set_regions_ex)                        */
#define PMX_FEED_WORDS              16
#define PMX_FEED_ERR_BASE (1ull << 62)
/* feed_forward_read / feed_reverse_read (mscc.pyx:370-418) for a run of n reads of one chromosome in FILE ORDER:
 * h_pos[i] = 1-based leftmost position, h_readlen[i] = query length, h_is_reverse[i] != 0 for the reverse strand -- or
 * h_is_reverse = NULL and the strand PACKED into the top bit of h_pos[i] (set: reverse; 4 bytes per read instead of 5) --
 * (forward bit: pos; reverse bit: pos + readlen - 1).  pos_bytes: 4 (int32) or 8 (int64); len_bytes: 2 (uint16), 4 (int32)
 * or 8 (int64), or 0: every read of the run has the same length and h_readlen points to that ONE int64.  Applies the
 * reference's rules in file order across calls: a forward read at the position of the previous forward read is a
 * duplicate; a reverse read counts only if its bit was clear; read-length sums over the reads that count.
 * `reads_before` = reads fed to this chromosome by earlier calls since its state was cleared.  Asynchronous. */
int pmx_feed_reads(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *h_pos, uint32_t pos_bytes,
                   const void *h_readlen, uint32_t len_bytes, const uint8_t *h_is_reverse, uint64_t n,
                   uint64_t reads_before, uint64_t *d_state);
/* The same with flags.  PMX_FEED_WHOLE_VECTORS: this run is the FIRST of its chromosome (reads_before = 0) and d_F / d_R
 * need not be cleared -- the call writes EVERY word of both vectors: the words are dealt to workgroups, each finds the
 * reads of its words in the sorted run and stores them whole, instead of one atomic OR per read into cleared vectors
 * (round 4: the kernels of the feed, not the PCIe copy, paced a genome).  Same rules, same state words.  Later runs of
 * the chromosome are fed without the flag. */
#define PMX_FEED_WHOLE_VECTORS 1u
int pmx_feed_reads_ex(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *h_pos, uint32_t pos_bytes,
                      const void *h_readlen, uint32_t len_bytes, const uint8_t *h_is_reverse, uint64_t n,
                      uint64_t reads_before, uint64_t *d_state, uint32_t flags);
/* The same run of reads in TWO bytes per read (round 4: the feed of a genome is a PCIe copy, and the reads of a sorted BAM
 * lie ~100 bp apart): h_words[i] = strand in bit 15 (set: reverse), distance to the position of read i - 1 in bits 0..14.
 * The run is cut into nseg SEGMENTS of 1..1024 reads: h_seg_start[s] = index of the first read of segment s
 * (h_seg_start[0] = 0, h_seg_start[nseg] = n), h_seg_base[s] = its absolute 1-based position (its distance field is 0); a
 * new segment begins at least every 1024 reads and wherever two neighbours lie 32767 bp or more apart.  Read lengths as
 * for pmx_feed_reads.  nbits < 2^31 (32-bit positions: what a BAM file can hold, SAMv1 4.2).  The device expands the
 * words to positions (one wavefront per segment) and applies the rules of feed_forward_read / feed_reverse_read
 * (mscc.pyx:370-418) exactly as pmx_feed_reads does.  flags: PMX_FEED_WHOLE_VECTORS (above) or 0.  Asynchronous. */
int pmx_feed_reads_delta16(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const uint16_t *h_words, uint64_t n,
                           const uint32_t *h_seg_start, const int32_t *h_seg_base, uint32_t nseg, const void *h_readlen,
                           uint32_t len_bytes, uint64_t reads_before, uint64_t *d_state, uint32_t flags);
/* The same run of reads already IN DEVICE MEMORY (round 4: libpymasc_ingest.so inflates and decodes a BAM file on the GPU,
 * include/pymasc_amd_ingest.h, and its kept records never visit the host): int32 1-based positions, int32 query lengths,
 * uint8 strand.  The arrays must be complete when the call is made (the producer has synchronised its stream) and stay
 * untouched until the next synchronising call on this context.  Same rules, same state words as pmx_feed_reads
 * (mscc.pyx:370-418); flags: PMX_FEED_WHOLE_VECTORS or 0.  Asynchronous. */
int pmx_feed_reads_dev(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const int32_t *d_pos, const int32_t *d_readlen,
                       const uint8_t *d_is_reverse, uint64_t n, uint64_t reads_before, uint64_t *d_state, uint32_t flags);
/* _load_mappability's loop (mscc.pyx:343-344): set(h_first[i] + first_offset, h_last[i]) for n intervals, ends inclusive
 * (BigWig (begin, end) pairs: first_offset = 1).  width_bytes: 4 (uint32) or 8 (int64).  An interval outside [0, nbits) is
 * clipped and recorded in d_state[PMX_FEED_FIRST_OUT_OF_RANGE] (d_state may be NULL).  Asynchronous. */
int pmx_bits_set_regions_async(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *h_first, const void *h_last,
                               uint32_t width_bytes, uint64_t n, int64_t first_offset, uint64_t *d_state);
/* The same with flags (round 4).  PMX_REGIONS_CLEAR: the vector is cleared first, in the same stream order (n may be 0).
 * PMX_REGIONS_SIDE: clear, copy and kernel run on a SIDE stream of the context, behind everything queued on the context's
 * stream at the time of the call and beside what is queued after it: a chromosome's mappability vector is built while the
 * next chromosome's reads are fed (the feeders pmx_feed_reads* never wait for the side stream; a genome's feed was paced by
 * one queue of small kernels, 25 x (feed + regions)).  Until the next call of any OTHER entry point on this context --
 * pmx_cc_batch_dev, pmx_bits_count, pmx_ctx_sync, ... all of which first make the context's stream wait for the side stream
 * -- only the feeders may be called, and not on d_words. */
#define PMX_REGIONS_CLEAR 1u
#define PMX_REGIONS_SIDE 2u
/* PMX_REGIONS_SORTED: the intervals are in BigWig order -- h_first[i] + first_offset <= h_last[i] < h_first[i + 1] +
 * first_offset -- and the call BUILDS the vector: every word is written once by the workgroup that owns it, zeros included
 * (no clear, no atomics; whatever the vector held is gone, PMX_REGIONS_CLEAR is implied; n may be 0).  The order is checked
 * on the device: a violation is recorded in d_state[PMX_FEED_REGIONS_UNSORTED] (d_state must not be NULL) and leaves the
 * vector undefined -- a caller that cannot vouch for the order checks it first (two comparisons per interval) and uses
 * the plain call otherwise, as pymasc_amd/calculator.py does. */
#define PMX_REGIONS_SORTED 4u
int pmx_bits_set_regions_ex(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *h_first, const void *h_last,
                            uint32_t width_bytes, uint64_t n, int64_t first_offset, uint64_t *d_state, uint32_t flags);
/* The same for intervals already IN DEVICE MEMORY (round 4: libpymasc_ingest.so decodes a BigWig file on the GPU, pmx_dbw_fetch):
 * uint32 d_first[n], d_last[n], complete when the call is made and untouched until the next synchronising call.  flags:
 * PMX_REGIONS_CLEAR, PMX_REGIONS_SORTED (not PMX_REGIONS_SIDE: nothing is copied, the kernel runs on the context's stream). */
int pmx_bits_set_regions_dev_ex(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const uint32_t *d_first, const uint32_t *d_last,
                                uint64_t n, int64_t first_offset, uint64_t *d_state, uint32_t flags);
/* (The three feeders above and below stage their host arrays in one device slot, each array padded to 16 bytes.  Arrays
 * that lie in host memory in that same layout -- back to back in argument order, each padded to 16 bytes, ideally in
 * page-locked memory from pmx_host_alloc -- are copied in ONE piece; anything else array by array.) */
/* A batch of chromosomes from bit positions and intervals in host memory -> cleared and rebuilt vectors, ONE call, no
 * synchronisation: for callers that hold bit positions already (any of d_F / d_R / d_M may be NULL).  Range errors of the
 * whole batch are reported by pmx_bits_build_status, which synchronises. */
typedef struct pmx_build_job {
    uint64_t *d_F, *d_R, *d_M;
    uint64_t nbits;
    const void *h_fpos, *h_rpos;        /* bit positions, pos_bytes wide */
    uint64_t n_f, n_r;
    const void *h_first, *h_last;       /* set(first, last), inclusive, pos_bytes wide */
    uint64_t n_iv;
} pmx_build_job;
int pmx_bits_build_batch(pmx_ctx *ctx, uint32_t njobs, const pmx_build_job *jobs, uint32_t pos_bytes);
int pmx_bits_build_status(pmx_ctx *ctx);   /* PMX_OK, or PMX_ERR_INVALID naming the first job with a position outside its vector */

/* ---- the hot path ------------------------------------------------------------------------ */
/* CCBitArrayCalculator._calc_correlation for one chromosome (mscc.pyx:217-325), inputs resident in
 * HBM.  d_M may be NULL (no mappability: rows 1-4 left zero).  d_out: PMX_NROWS * (max_shift+1)
 * uint64 in device memory, overwritten.  Inputs are NOT modified (the reference destroys R in
 * place, mscc.pyx:316; nothing reads it afterwards).  Asynchronous. */
int pmx_cc_dev(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M,
               uint64_t nbits, uint32_t max_shift, uint32_t read_len, uint32_t flags,
               uint64_t *d_out);
/* The same for a BATCH of chromosomes in one pass of the kernels -- what a worker that owns several
 * chromosomes (PyMaSC/handler/worker.py:68-104 pulls them one by one) should call: the batch shares
 * max_shift / read_len / flags; d_F, d_R, d_M, nbits, d_out are HOST arrays of njobs entries (device
 * pointers / sizes); d_M is NULL or has a non-NULL entry for every job.  Asynchronous; the arrays
 * may be reused as soon as the call returns. */
int pmx_cc_batch_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                     const uint64_t *const *d_M, const uint64_t *nbits, uint32_t max_shift,
                     uint32_t read_len, uint32_t flags, uint64_t *const *d_out);
/* The same for SHARES of chromosomes (round 4: multi-GPU runs cut the genome into equal tile ranges instead of whole
 * chromosomes): job i computes the part of its chromosome's sums that belongs to the PMX_RANGE_TILE_BITS-bit tiles
 * [tile_first[i], tile_first[i] + tile_count[i]) -- every pair, run-edge event and mappable position is owned by exactly
 * one tile (the forward read's / the reverse read's / the lower edge's), so the result blocks of jobs that cover a
 * chromosome between them ADD UP, row by row and scalar by scalar, to what pmx_cc_batch_dev writes for it (the
 * north star's all-reduce(sum) of per-shift sums); the path marker is written by the job that holds tile 0.  The
 * vectors are the WHOLE chromosome's (a tile's events reach max_shift + read_len bits beyond it).  3 <= max_shift
 * <= 1023, read_len <= 1024, not with FORCE_DENSE / WINDOW_ONLY / SKIP_MLEN: ranges are taken by the event kernel
 * (dense tiles inside them still go to the window kernels).  Asynchronous. */
#define PMX_RANGE_TILE_BITS 65536u
int pmx_cc_batch_ranges_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                            const uint64_t *const *d_M, const uint64_t *nbits, const uint32_t *tile_first,
                            const uint32_t *tile_count, uint32_t max_shift, uint32_t read_len, uint32_t flags,
                            uint64_t *const *d_out);
/* Same as pmx_cc_dev with host buffers: uploads F, R (and M), runs, downloads the result block. Synchronous. */
int pmx_calc_correlation(pmx_ctx *ctx, const uint64_t *h_F, const uint64_t *h_R, const uint64_t *h_M,
                         uint64_t nbits, uint32_t max_shift, uint32_t read_len, uint32_t flags,
                         uint64_t *h_out);
/* CCBitArrayCalculator._fill_result's loop for chromosomes without reads (mscc.pyx:207-215):
 * out[k] = sum_j M[j] & M[j+k], k = 0..max_shift.  d_out / h_out: (max_shift+1) uint64. */
int pmx_mappable_len_dev(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_shift,
                         uint32_t flags, uint64_t *d_out);
int pmx_mappable_len(pmx_ctx *ctx, const uint64_t *h_M, uint64_t nbits, uint32_t max_shift,
                     uint32_t flags, uint64_t *h_out);
/* The same for a batch of read-less chromosomes in one pass of the kernels (finishup_calculation walks every
 * reference, mscc.pyx:436-439: 86 in the reference's test BAM): d_M, nbits, d_out are HOST arrays of njobs entries. */
int pmx_mappable_len_batch_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_M, const uint64_t *nbits,
                               uint32_t max_shift, uint32_t flags, uint64_t *const *d_out);

/* ---- measurement ------------------------------------------------------------------------- */
/* on = 1: every launch of a hot-path kernel that does work is bracketed by HIP events; on = 2: also the window-kernel
 * launches behind the event kernel, which return at once unless it flagged dense tiles (a bracket costs ~6 us of
 * stream time); 0: off.
 * With profiling on, every launch of a hot-path kernel is bracketed by HIP events on the context's
 * stream.  pmx_ctx_kernel_time synchronises and returns the summed duration and launch count of
 * kernel `kernel_id` since the last reset. */
#define PMX_KERNEL_CC_DENSE     0
#define PMX_KERNEL_CC_SPARSE    1
#define PMX_KERNEL_AUTOCORR     2
#define PMX_KERNEL_CC_EVENTS    3   /* the event kernel of the set-bit path (sparse tiles; CC_SPARSE then counts its window pass) */
#define PMX_KERNEL_COUNT_       4
int pmx_ctx_set_profiling(pmx_ctx *ctx, int on);
int pmx_ctx_reset_kernel_times(pmx_ctx *ctx);
int pmx_ctx_kernel_time(pmx_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches);
const char *pmx_kernel_name(int kernel_id);

/* ---- diagnostics (tests, tools/) ----------------------------------------------------------- */
/* Fills the context's scratch buffers (mask bits 0..7: slab, pair slab, window slabs, flag arrays, scratch, staging,
 * output staging) and the LDS of every CU (bit 8) with `pattern`, so that a kernel reading memory it has not written
 * fails deterministically.  No effect on results. */
int pmx_debug_poison(pmx_ctx *ctx, uint32_t pattern, uint32_t mask);
/* Caps the number of persistent workgroups of every later launch (0: no cap), so that a small test input gives each
 * workgroup many tiles: the paths that depend on a workgroup's tile count (histogram flushes before a 16-bit cell can
 * overflow, job changes inside a range) are then checked against the oracle at sizes it finishes.  No effect on results. */
int pmx_debug_set_max_workgroups(pmx_ctx *ctx, uint32_t n);
/* Copies `bytes` from byte offset `off` of the event / window kernels' slab to host memory (phase stamps of the
 * diagnostic builds -DEV_STAMPS / -DSP_STAMPS). */
int pmx_debug_read_slab(pmx_ctx *ctx, uint64_t off, void *dst, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* PYMASC_AMD_H */
