// Internal declarations shared by the HIP translation units of libpymasc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/pymasc_amd.h"

typedef unsigned long long u64;   // what atomicAdd(unsigned long long*) wants; same size as uint64_t
typedef unsigned int u32;

void pmx_set_error(const char *fmt, ...);

#define PMX_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            pmx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                        \
            return PMX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define PMX_CHECK_LAUNCH(name)                                                              \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            pmx_set_error("launch of %s failed: %s", name, hipGetErrorString(e__));         \
            return PMX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

struct pmx_timed_launch {
    int kernel_id;
    bool active;
    hipEvent_t start, stop;
};

#define PMX_SIDE_SLOTS 4    // staging slots of the side stream's work
struct pmx_side_task {
    uint64_t *d_words;
    uint64_t nbits;
    const void *h_first, *h_last;
    uint32_t width_bytes;
    uint64_t n;
    int64_t first_offset;
    uint64_t *d_state;
    uint32_t flags;
    hipEvent_t fork;
};

#ifndef PMX_FEED_SLOTS
#define PMX_FEED_SLOTS 6
#endif
#define PMX_JOBTAB_SLOTS 8

struct pmx_ctx {
    int device;
    hipStream_t stream;          // the stream launches are enqueued on (the caller's, or our own)
    bool own_stream;
    // The mappable-length pass only reads M and writes its own row: it runs on a second stream beside the set-bit kernel
    // (fork / join events on `stream`), so its chain of small launches hides under k_cc_sparse.
    hipStream_t aux_stream;
    hipEvent_t ev_fork, ev_join;
    int num_cus;
    bool window_only;            // PMX_FLAG_WINDOW_ONLY of the call in progress (pmx_cc_batch_dev)
    bool deep_lists;             // PMX_FLAG_DEEP_LISTS of the call in progress
    uint32_t debug_max_wg;       // 0: off; tests: cap on the persistent workgroups of a launch (many tiles per workgroup on small inputs)
    int profiling;               // 0 off; 1: kernels that do work; 2: also the (usually empty) fallback launches behind the event kernel
    std::vector<pmx_timed_launch> timed;       // launches not yet folded into the totals
    std::vector<hipEvent_t> event_pool;
    double total_ms[PMX_KERNEL_COUNT_];
    uint64_t launches[PMX_KERNEL_COUNT_];
    // scratch (device): autocorrelation row + small flags
    u64 *d_scratch;
    size_t scratch_words;
    // per-workgroup partial result slabs of the set-bit kernels (u32)
    u32 *d_slab;
    size_t slab_words;
    // the pair-enumeration autocorrelation pass: its own slab, and the dense-tile flags (+ counter) it hands to the window pass
    u32 *d_slab2;
    size_t slab2_words;
    unsigned char *d_flags;
    size_t flags_bytes;
    unsigned char *d_flags_cc;   // dense-tile flags (+ counter) the event kernel hands to the window kernel
    size_t flags_cc_bytes;
    // the event pass uses two areas of d_flags_cc in turn; a pass's k_events_finish clears the other one (kernels_sparse.hip: ev_flag_area)
    uint32_t flags_cc_area;
    uint32_t *d_probe, *h_probe;   // density probe: device counters + their page-locked copy (pmx_cc_batch_dev without a hint)
    size_t flags_cc_zeroed[2], flags_cc_dirty[2];
    u32 *d_slab_ac;              // slab of the window autocorrelation kernel (separate: it may run beside k_cc_sparse)
    size_t slab_ac_words;
    u32 *d_slab_fb;              // slab of k_cc_sparse when it runs as the fallback behind the event kernel
    size_t slab_fb_words;
    // staging for the host-pointer entry points
    uint64_t *d_stage[3];
    size_t stage_words[3];
    // staging of the stream-ordered feeders (pmx_feed_reads, pmx_bits_set_regions_async, pmx_bits_build_batch): a ring of
    // FEED_SLOTS buffers; a slot is reused once the event recorded behind its consumer kernels has passed
    uint64_t *d_feed[PMX_FEED_SLOTS];
    size_t feed_words[PMX_FEED_SLOTS];
    hipEvent_t feed_done[PMX_FEED_SLOTS];
    bool feed_used[PMX_FEED_SLOTS];
    uint32_t feed_next;
    hipStream_t copy_stream;     // H2D copies of the feeders run here, one slot ahead of the kernels on `stream`
    hipStream_t copy_stream2;    // ... the small interval copies of pmx_bits_set_regions_async here, beside the read copies
    hipEvent_t feed_copied;
    // job tables in device memory (kernels_sparse.hip: SpJobTableRef): a ring of (page-locked staging buffer, device buffer)
    // pairs filled on the second copy stream, so that neither the host nor the compute stream waits for an upload
    void *d_jobtab[PMX_JOBTAB_SLOTS];
    void *h_jobtab[PMX_JOBTAB_SLOTS];
    size_t h_jobtab_bytes[PMX_JOBTAB_SLOTS];              // (capacity of both buffers of the slot)
    hipEvent_t jobtab_done[PMX_JOBTAB_SLOTS];             // the copy into the device buffer has run (copy stream)
    hipEvent_t jobtab_mark[PMX_JOBTAB_SLOTS][2];          // where the context's two streams stood when the NEXT table was uploaded
    bool jobtab_used[PMX_JOBTAB_SLOTS], jobtab_marked[PMX_JOBTAB_SLOTS];
    uint32_t jobtab_next;
    // pmx_bits_set_regions_ex(PMX_REGIONS_SIDE): clear, copy (copy_stream2, a staging ring of its own) and kernel on a stream
    // beside the context's.  side_fork: where the context's stream stood when the work was queued (the side stream waits for
    // it); side_done: the end of the side work (pmx_side_join: the context's stream waits for it -- every entry point but the
    // read feeders does that first).
    hipStream_t side_stream;
    hipEvent_t side_fork, side_done, side_copied;
    bool side_pending;
    uint64_t *d_side[PMX_SIDE_SLOTS];
    size_t side_words[PMX_SIDE_SLOTS];
    hipEvent_t side_slot_done[PMX_SIDE_SLOTS];
    bool side_slot_used[PMX_SIDE_SLOTS];
    uint32_t side_next;
    hipStream_t user_stream;     // `stream` as given at creation (launchers may point `stream` at aux_stream for a while)
    u64 *d_build_err;            // pmx_bits_build_batch: one range-error word per job since the last pmx_bits_build_status
    size_t build_err_cap, build_err_jobs;
    u64 *d_out_stage;
    size_t out_stage_words;
    // result blocks of max_shift < 3 are computed 4 columns wide here and narrowed into the caller's blocks
    u64 *d_pad_stage;
    size_t pad_stage_words;
};

// density probe (kernels_bits.hip): up to PMX_PROBE_JOBS chromosomes per launch, PMX_PROBE_SAMPLES tiles of 64 Kbit each
#define PMX_PROBE_JOBS 64u
#define PMX_PROBE_SAMPLES 16u
struct pmx_probe_jobs {
    const u32 *F[PMX_PROBE_JOBS], *R[PMX_PROBE_JOBS], *M[PMX_PROBE_JOBS];
    uint64_t nbits[PMX_PROBE_JOBS];
};
int pmx_launch_density_probe(pmx_ctx *ctx, const pmx_probe_jobs *jobs, uint32_t njobs, uint32_t *d_out);   // d_out: [4 per job] forward, reverse, edges, positions sampled

// profiling helpers (pmx_api.hip)
int pmx_prof_begin(pmx_ctx *ctx, int kernel_id, pmx_timed_launch *tl, bool fallback = false);
int pmx_prof_end(pmx_ctx *ctx, pmx_timed_launch *tl);
int pmx_side_join(pmx_ctx *ctx);   // the context's stream waits for the side stream's work (no-op when there is none)
#define PMX_JOIN_SIDE(ctx)                   \
    do {                                     \
        const int rcj_ = pmx_side_join(ctx); \
        if (rcj_) return rcj_;               \
    } while (0)

int pmx_ensure_scratch(pmx_ctx *ctx, size_t words);
// copies `bytes` of a job table to the device buffer of the context's current stream (stream-ordered; the source may be
// reused at once); *d receives the device address
int pmx_upload_jobtab(pmx_ctx *ctx, const void *src, size_t bytes, const void **d);

// ---- launchers implemented next to their kernels --------------------------------------------
// bits (kernels_bits.hip)
int pmx_launch_set_positions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_pos, uint64_t n,
                             u64 *d_bad = nullptr);   // d_bad: device u64, receives the smallest out-of-range index
int pmx_launch_set_regions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_from,
                           const int64_t *d_to, uint64_t n);
int pmx_launch_count(pmx_ctx *ctx, const uint64_t *d_words, uint64_t nbits, u64 *d_count /* += */);

// stream-ordered feeding (kernels_feed.hip): device arrays in, nothing read back
int pmx_launch_feed_reads(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *d_pos, uint32_t pos_bytes,
                          const void *d_len, uint32_t len_bytes, int64_t uniform_len, const unsigned char *d_rev, uint64_t n,
                          uint64_t base, uint64_t *d_state, uint64_t *d_partial = nullptr);   // d_partial: PMX_FEED_WHOLE_VECTORS (k_feed_build)
uint32_t pmx_feed_build_blocks(uint64_t nbits);
int pmx_launch_feed_expand16(pmx_ctx *ctx, const void *d_words, const void *d_seg_start, const void *d_seg_base, uint32_t nseg,
                             uint64_t n, void *d_pos32);
int pmx_launch_set_regions_w(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to, uint32_t width,
                             uint64_t n, int64_t offset, uint64_t *d_err);
// the whole vector from sorted, disjoint intervals (k_regions_build): no clear needed, violations recorded in d_err_order
int pmx_launch_regions_build_on(pmx_ctx *ctx, hipStream_t stream, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to,
                                uint32_t width, uint64_t n, int64_t offset, uint64_t *d_err_range, uint64_t *d_err_order);
// (the same on an explicit stream: the side work never touches ctx->stream)
int pmx_launch_set_regions_on(pmx_ctx *ctx, hipStream_t stream, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to, uint32_t width,
                             uint64_t n, int64_t offset, uint64_t *d_err);
int pmx_launch_set_positions_w(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *d_pos, uint32_t width, uint64_t n,
                               uint64_t *d_err);

// dense cross-correlation (kernels_dense.hip)
// rows NCC_CCBINS / MSCC_FSUM / MSCC_RSUM / MSCC_CCBINS of d_out (stride = out_stride) are overwritten (per-workgroup
// partial sums go through the context's slab, no atomics).
int pmx_launch_cc_dense(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M,
                        uint64_t nbits, uint32_t max_shift, uint32_t read_len, bool do_ncc,
                        u64 *d_out, uint32_t out_stride, const u64 *d_select, uint32_t select_mode);
// ---- set-bit driven kernels (kernels_sparse.hip): one launch covers a batch of chromosomes ------------------
struct pmx_job {
    const uint64_t *d_F, *d_R, *d_M;   // device bit-vectors (d_M null: no mappability; all jobs of a batch alike)
    uint64_t nbits;
    uint64_t *d_out;                   // result block [PMX_NROWS][out_stride] (or the lag row, autocorr mode 0)
    uint64_t *d_out2;                  // autocorrelation only: pmx_autocorr_scratch_words(max_lag) u64 of per-job scratch
    uint32_t tile_first, tile_count;   // pmx_cc_batch_ranges_dev: the 64-Kbit tiles of the chromosome this job takes (count 0: all)
};
// Whether the event kernel took the mappable-length pass as well (edge pairs in k_cc_events, the autocorrelation window
// kernel for the tiles it flagged, the recurrence in k_events_tail).
struct pmx_fused_mlen {
    bool done;   // row MLEN and scalar [2] of every job were written by pmx_launch_cc_sparse_batch: no autocorrelation pass
};
int pmx_sparse_supported(uint32_t max_shift, uint32_t read_len);
// the event kernel can also take the edge pairs of the mappable-length pass for this geometry (and is not disabled)
int pmx_events_can_fuse_mlen(uint32_t max_shift, uint32_t max_lag);
// the event kernel (not the window kernel in shift chunks) takes this max_shift > 1023
int pmx_events_take_big(uint32_t max_shift);
int pmx_events_used(uint32_t max_shift);      // the event kernel takes this shift range (any of its instantiations)
int pmx_events_big_subgroups(uint32_t max_shift, int has_m);
uint32_t pmx_sparse_max_jobs(void);
uint32_t pmx_cc_batch_jobs(uint32_t max_shift);   // jobs per call of pmx_launch_cc_sparse_batch (device-side job tables beyond 1023 shifts)
uint32_t pmx_autocorr_batch_jobs(void);           // ... of pmx_launch_autocorr_edges_batch
// Writes rows NCC_CCBINS / MSCC_FSUM / MSCC_CCBINS / MSCC_RSUM and the scalar row of every job's result block
// (rows the batch does not produce are written as zeros, MLEN included when there is no mappability).
int pmx_launch_cc_sparse_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_shift,
                               uint32_t read_len, bool do_ncc, uint32_t out_stride, bool zero_mlen,
                               uint32_t fused_lag = 0xffffffffu, pmx_fused_mlen *fused = nullptr);
// fused_lag / fused: let the event kernel enumerate the edge pairs up to that lag as well (njobs <= pmx_sparse_max_jobs())
// zero_mlen == false: the autocorrelation pass (possibly running concurrently) owns row MLEN and scalar [2]: both are left alone
// Run-edge autocorrelation of every job's d_M.  mode 0: d_out[k] = A(k), k <= max_lag.
// mode 1: d_out is a result block: row MLEN[d] = A(|read_len - 1 - d|), d <= max_shift; scalar [2] = popcount(M).
int pmx_launch_autocorr_edges_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_lag,
                                    uint32_t mode, uint32_t read_len, uint32_t max_shift, uint32_t out_stride);
int pmx_ensure_slab(pmx_ctx *ctx, size_t u32_words);
int pmx_ensure_slab2(pmx_ctx *ctx, size_t u32_words);
int pmx_ensure_slab_ac(pmx_ctx *ctx, size_t u32_words);
int pmx_ensure_slab_fb(pmx_ctx *ctx, size_t u32_words);
int pmx_ensure_flags(pmx_ctx *ctx, size_t bytes);
int pmx_ensure_flags_cc(pmx_ctx *ctx, size_t bytes);
size_t pmx_autocorr_scratch_words(uint32_t max_lag);
// out[k] = sum_j M[j] & M[j+k], k = 0..max_lag
int pmx_launch_autocorr_dense(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_lag, u64 *d_out);
// mlen_by_shift[d] = autocorr[|read_len - 1 - d|], d = 0..max_shift
int pmx_launch_mlen_map(pmx_ctx *ctx, const u64 *d_autocorr, uint32_t max_shift, uint32_t read_len, u64 *d_mlen);
