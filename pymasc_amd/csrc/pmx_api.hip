// extern "C" surface of libpymasc_hip.so -- see include/pymasc_amd.h for the contract and the
// reference interfaces each entry point replaces.
#include "pmx_common.h"
#include <algorithm>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void pmx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#define REQUIRE(cond, msg)              \
    do {                                \
        if (!(cond)) {                  \
            pmx_set_error("%s", msg);   \
            return PMX_ERR_INVALID;     \
        }                               \
    } while (0)

static inline uint64_t words_for(uint64_t nbits) { return (nbits + 63) / 64; }

extern "C" {

const char *pmx_last_error(void) { return g_err; }
int pmx_version(void) { return 100; }
#ifndef PMX_SRC_SHA16
#define PMX_SRC_SHA16 "unknown"
#endif
const char *pmx_build_id(void) { return PMX_SRC_SHA16; }

int pmx_device_count(int *n)
{
    REQUIRE(n, "pmx_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        c = 0;
        (void)hipGetLastError();
    }
    *n = c;
    return PMX_OK;
}

int pmx_ctx_create(int device, void *hip_stream, pmx_ctx **out)
{
    REQUIRE(out, "pmx_ctx_create: out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        pmx_set_error("pmx_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return PMX_ERR_NODEVICE;
    }
    REQUIRE(device >= 0 && device < ndev, "pmx_ctx_create: device index out of range");
    PMX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PMX_HIP(hipGetDeviceProperties(&prop, device));
    pmx_ctx *ctx = new (std::nothrow) pmx_ctx();
    if (!ctx) {
        pmx_set_error("pmx_ctx_create: out of host memory");
        return PMX_ERR_NOMEM;
    }
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->profiling = 0;
    ctx->window_only = false;
    ctx->deep_lists = false;
    ctx->debug_max_wg = 0;
    ctx->d_scratch = nullptr;
    ctx->scratch_words = 0;
    ctx->d_slab = nullptr;
    ctx->slab_words = 0;
    ctx->d_slab2 = nullptr;
    ctx->slab2_words = 0;
    ctx->d_flags = nullptr;
    ctx->flags_bytes = 0;
    ctx->d_flags_cc = nullptr;
    ctx->flags_cc_bytes = 0;
    ctx->flags_cc_area = 0;
    ctx->d_probe = nullptr;
    ctx->h_probe = nullptr;
    ctx->flags_cc_zeroed[0] = ctx->flags_cc_zeroed[1] = ctx->flags_cc_dirty[0] = ctx->flags_cc_dirty[1] = 0;
    ctx->d_slab_ac = nullptr;
    ctx->slab_ac_words = 0;
    ctx->d_slab_fb = nullptr;
    ctx->slab_fb_words = 0;
    ctx->aux_stream = nullptr;
    ctx->ev_fork = nullptr;
    ctx->ev_join = nullptr;
    ctx->d_out_stage = nullptr;
    ctx->out_stage_words = 0;
    ctx->d_pad_stage = nullptr;
    ctx->pad_stage_words = 0;
    for (int i = 0; i < 3; i++) {
        ctx->d_stage[i] = nullptr;
        ctx->stage_words[i] = 0;
    }
    for (int i = 0; i < PMX_FEED_SLOTS; i++) {
        ctx->d_feed[i] = nullptr;
        ctx->feed_words[i] = 0;
        ctx->feed_done[i] = nullptr;
        ctx->feed_used[i] = false;
    }
    ctx->feed_next = 0;
    for (int i = 0; i < PMX_JOBTAB_SLOTS; i++) {
        ctx->d_jobtab[i] = nullptr;
        ctx->h_jobtab[i] = nullptr;
        ctx->h_jobtab_bytes[i] = 0;
        ctx->jobtab_done[i] = nullptr;
        ctx->jobtab_mark[i][0] = ctx->jobtab_mark[i][1] = nullptr;
        ctx->jobtab_used[i] = false;
        ctx->jobtab_marked[i] = false;
    }
    ctx->jobtab_next = 0;
    ctx->copy_stream = nullptr;
    ctx->copy_stream2 = nullptr;
    ctx->side_stream = nullptr;
    ctx->side_done = ctx->side_copied = ctx->side_fork = nullptr;
    ctx->side_pending = false;
    for (int i = 0; i < PMX_SIDE_SLOTS; i++) {
        ctx->d_side[i] = nullptr;
        ctx->side_words[i] = 0;
        ctx->side_slot_done[i] = nullptr;
        ctx->side_slot_used[i] = false;
    }
    ctx->side_next = 0;
    ctx->feed_copied = nullptr;
    ctx->d_build_err = nullptr;
    ctx->build_err_cap = 0;
    ctx->build_err_jobs = 0;
    for (int i = 0; i < PMX_KERNEL_COUNT_; i++) {
        ctx->total_ms[i] = 0;
        ctx->launches[i] = 0;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            pmx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return PMX_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    ctx->user_stream = ctx->stream;
    bool ok = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->copy_stream2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->feed_copied, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->side_copied, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < PMX_SIDE_SLOTS; i++) ok = hipEventCreateWithFlags(&ctx->side_slot_done[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < PMX_FEED_SLOTS; i++) ok = hipEventCreateWithFlags(&ctx->feed_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok || hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        pmx_set_error("pmx_ctx_create: cannot create the auxiliary stream / events");
        (void)pmx_ctx_destroy(ctx);
        return PMX_ERR_HIP;
    }
    *out = ctx;
    return PMX_OK;
}

int pmx_ctx_sync(pmx_ctx *ctx)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx, "pmx_ctx_sync: ctx is NULL");
    PMX_JOIN_SIDE(ctx);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_ctx_destroy(pmx_ctx *ctx)
{
    if (!ctx) return PMX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->copy_stream2) (void)hipStreamSynchronize(ctx->copy_stream2);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (int i = 0; i < PMX_FEED_SLOTS; i++) {
        if (ctx->d_feed[i]) (void)hipFree(ctx->d_feed[i]);
        if (ctx->feed_done[i]) (void)hipEventDestroy(ctx->feed_done[i]);
    }
    if (ctx->feed_copied) (void)hipEventDestroy(ctx->feed_copied);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->copy_stream2) (void)hipStreamDestroy(ctx->copy_stream2);
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    for (int i = 0; i < PMX_SIDE_SLOTS; i++) {
        if (ctx->d_side[i]) (void)hipFree(ctx->d_side[i]);
        if (ctx->side_slot_done[i]) (void)hipEventDestroy(ctx->side_slot_done[i]);
    }
    if (ctx->side_copied) (void)hipEventDestroy(ctx->side_copied);
    if (ctx->side_done) (void)hipEventDestroy(ctx->side_done);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->d_build_err) (void)hipFree(ctx->d_build_err);
    for (int i = 0; i < PMX_JOBTAB_SLOTS; i++) {
        if (ctx->d_jobtab[i]) (void)hipFree(ctx->d_jobtab[i]);
        if (ctx->h_jobtab[i]) (void)hipHostFree(ctx->h_jobtab[i]);
        if (ctx->jobtab_done[i]) (void)hipEventDestroy(ctx->jobtab_done[i]);
        if (ctx->jobtab_mark[i][0]) (void)hipEventDestroy(ctx->jobtab_mark[i][0]);
        if (ctx->jobtab_mark[i][1]) (void)hipEventDestroy(ctx->jobtab_mark[i][1]);
    }
    for (auto &tl : ctx->timed) {
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    for (auto &e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_slab) (void)hipFree(ctx->d_slab);
    if (ctx->d_slab2) (void)hipFree(ctx->d_slab2);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_flags_cc) (void)hipFree(ctx->d_flags_cc);
    if (ctx->d_probe) (void)hipFree(ctx->d_probe);
    if (ctx->h_probe) (void)hipHostFree(ctx->h_probe);
    if (ctx->d_slab_ac) (void)hipFree(ctx->d_slab_ac);
    if (ctx->d_slab_fb) (void)hipFree(ctx->d_slab_fb);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->d_out_stage) (void)hipFree(ctx->d_out_stage);
    if (ctx->d_pad_stage) (void)hipFree(ctx->d_pad_stage);
    for (int i = 0; i < 3; i++)
        if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PMX_OK;
}

}   // extern "C"

// ---- internal helpers ---------------------------------------------------------------------------

int pmx_ensure_scratch(pmx_ctx *ctx, size_t words)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    if (ctx->scratch_words >= words) return PMX_OK;
    if (ctx->d_scratch) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_words = 0;
    }
    size_t want = words < 4096 ? 4096 : words;
    PMX_HIP(hipMalloc((void **)&ctx->d_scratch, want * sizeof(u64)));
    ctx->scratch_words = want;
    return PMX_OK;
}

int pmx_upload_jobtab(pmx_ctx *ctx, const void *src, size_t bytes, const void **d)
{
    // A ring of PMX_JOBTAB_SLOTS (page-locked staging buffer, device buffer) pairs.  The copy runs on the COPY stream and
    // the caller's stream only waits for its event: a table does not depend on the kernels queued before it, so it
    // travels while they run (round 4; a copy queued on the compute stream itself stalled it for ~30 us per table, five
    // tables per config-5 step).  A slot is taken again after PMX_JOBTAB_SLOTS further uploads, once the stream has
    // passed the point it had reached when the NEXT table was uploaded -- every launcher queues a table's consumers
    // before it uploads another table (jobtab_mark[]).
    const uint32_t slot = ctx->jobtab_next;
    const uint32_t prev = (slot + PMX_JOBTAB_SLOTS - 1) % PMX_JOBTAB_SLOTS;
    ctx->jobtab_next = (slot + 1) % PMX_JOBTAB_SLOTS;
    if (!ctx->jobtab_done[slot]) {
        PMX_HIP(hipEventCreateWithFlags(&ctx->jobtab_done[slot], hipEventDisableTiming));
        PMX_HIP(hipEventCreateWithFlags(&ctx->jobtab_mark[slot][0], hipEventDisableTiming));
        PMX_HIP(hipEventCreateWithFlags(&ctx->jobtab_mark[slot][1], hipEventDisableTiming));
    }
    // the consumers of the previous table are queued by now: mark BOTH streams of the context (a launcher may have been
    // pointed at the auxiliary stream: pmx_cc_batch_dev forks the mappable-length pass there)
    if (ctx->jobtab_used[prev]) {
        PMX_HIP(hipEventRecord(ctx->jobtab_mark[prev][0], ctx->user_stream));
        PMX_HIP(hipEventRecord(ctx->jobtab_mark[prev][1], ctx->aux_stream));
        ctx->jobtab_marked[prev] = true;
    }
    if (ctx->jobtab_used[slot]) {
        PMX_HIP(hipEventSynchronize(ctx->jobtab_done[slot]));           // the copy out of the staging buffer
        if (ctx->jobtab_marked[slot]) {                                 // the kernels that read the device buffer
            PMX_HIP(hipEventSynchronize(ctx->jobtab_mark[slot][0]));
            PMX_HIP(hipEventSynchronize(ctx->jobtab_mark[slot][1]));
        } else {                                                        // (not expected: a ring of one)
            PMX_HIP(hipStreamSynchronize(ctx->user_stream));
            PMX_HIP(hipStreamSynchronize(ctx->aux_stream));
        }
    }
    if (ctx->h_jobtab_bytes[slot] < bytes) {
        if (ctx->h_jobtab[slot]) PMX_HIP(hipHostFree(ctx->h_jobtab[slot]));
        if (ctx->d_jobtab[slot]) PMX_HIP(hipFree(ctx->d_jobtab[slot]));
        ctx->h_jobtab[slot] = nullptr;
        ctx->d_jobtab[slot] = nullptr;
        ctx->h_jobtab_bytes[slot] = 0;
        const size_t want = bytes * 2 + 4096;
        PMX_HIP(hipHostMalloc(&ctx->h_jobtab[slot], want, hipHostMallocDefault));
        PMX_HIP(hipMalloc(&ctx->d_jobtab[slot], want));
        ctx->h_jobtab_bytes[slot] = want;
    }
    memcpy(ctx->h_jobtab[slot], src, bytes);
    PMX_HIP(hipMemcpyAsync(ctx->d_jobtab[slot], ctx->h_jobtab[slot], bytes, hipMemcpyHostToDevice, ctx->copy_stream2));
    PMX_HIP(hipEventRecord(ctx->jobtab_done[slot], ctx->copy_stream2));
    PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->jobtab_done[slot], 0));
    ctx->jobtab_used[slot] = true;
    ctx->jobtab_marked[slot] = false;
    *d = ctx->d_jobtab[slot];
    return PMX_OK;
}

int pmx_ensure_slab(pmx_ctx *ctx, size_t u32_words)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    if (ctx->slab_words >= u32_words) return PMX_OK;
    if (ctx->d_slab) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_slab));
        ctx->d_slab = nullptr;
        ctx->slab_words = 0;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_slab, u32_words * sizeof(u32)));
    ctx->slab_words = u32_words;
    return PMX_OK;
}

int pmx_ensure_slab2(pmx_ctx *ctx, size_t u32_words)
{
    if (ctx->slab2_words >= u32_words) return PMX_OK;
    if (ctx->d_slab2) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_slab2));
        ctx->d_slab2 = nullptr;
        ctx->slab2_words = 0;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_slab2, u32_words * sizeof(u32)));
    ctx->slab2_words = u32_words;
    return PMX_OK;
}

int pmx_ensure_slab_ac(pmx_ctx *ctx, size_t u32_words)
{
    if (ctx->slab_ac_words >= u32_words) return PMX_OK;
    if (ctx->d_slab_ac) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_slab_ac));
        ctx->d_slab_ac = nullptr;
        ctx->slab_ac_words = 0;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_slab_ac, u32_words * sizeof(u32)));
    ctx->slab_ac_words = u32_words;
    return PMX_OK;
}

int pmx_ensure_slab_fb(pmx_ctx *ctx, size_t u32_words)
{
    if (ctx->slab_fb_words >= u32_words) return PMX_OK;
    if (ctx->d_slab_fb) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_slab_fb));
        ctx->d_slab_fb = nullptr;
        ctx->slab_fb_words = 0;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_slab_fb, u32_words * sizeof(u32)));
    ctx->slab_fb_words = u32_words;
    return PMX_OK;
}

int pmx_ensure_flags(pmx_ctx *ctx, size_t bytes)
{
    if (ctx->flags_bytes >= bytes) return PMX_OK;
    if (ctx->d_flags) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_flags));
        ctx->d_flags = nullptr;
        ctx->flags_bytes = 0;
    }
    const size_t want = bytes < 65536 ? 65536 : bytes * 2;
    PMX_HIP(hipMalloc((void **)&ctx->d_flags, want));
    ctx->flags_bytes = want;
    return PMX_OK;
}

int pmx_ensure_flags_cc(pmx_ctx *ctx, size_t bytes)
{
    if (ctx->flags_cc_bytes >= bytes) return PMX_OK;
    if (ctx->d_flags_cc) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_flags_cc));
        ctx->d_flags_cc = nullptr;
        ctx->flags_cc_bytes = 0;
    }
    const size_t want = bytes < 65536 ? 65536 : bytes * 2;
    PMX_HIP(hipMalloc((void **)&ctx->d_flags_cc, want));
    ctx->flags_cc_bytes = want;
    return PMX_OK;
}

static int ensure_stage(pmx_ctx *ctx, int slot, size_t words)
{
    if (ctx->stage_words[slot] >= words) return PMX_OK;
    if (ctx->d_stage[slot]) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_stage[slot]));
        ctx->d_stage[slot] = nullptr;
        ctx->stage_words[slot] = 0;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_stage[slot], (words ? words : 1) * sizeof(uint64_t)));
    ctx->stage_words[slot] = words ? words : 1;
    return PMX_OK;
}

static int ensure_out_stage(pmx_ctx *ctx, size_t words)
{
    if (ctx->out_stage_words >= words) return PMX_OK;
    if (ctx->d_out_stage) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipFree(ctx->d_out_stage));
        ctx->d_out_stage = nullptr;
    }
    PMX_HIP(hipMalloc((void **)&ctx->d_out_stage, words * sizeof(u64)));
    ctx->out_stage_words = words;
    return PMX_OK;
}

// the context's stream waits for whatever pmx_bits_set_regions_ex(PMX_REGIONS_SIDE) has queued on the side stream
int pmx_side_join(pmx_ctx *ctx)
{
    if (!ctx->side_pending) return PMX_OK;
    ctx->side_pending = false;
    PMX_HIP(hipStreamWaitEvent(ctx->user_stream, ctx->side_done, 0));
    return PMX_OK;
}

static int get_event(pmx_ctx *ctx, hipEvent_t *ev)
{
    if (!ctx->event_pool.empty()) {
        *ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return PMX_OK;
    }
    PMX_HIP(hipEventCreate(ev));
    return PMX_OK;
}

int pmx_prof_begin(pmx_ctx *ctx, int kernel_id, pmx_timed_launch *tl, bool fallback)
{
    tl->kernel_id = kernel_id;
    // a HIP-event pair costs ~6 us of stream time: the launches that return at once when the event kernel flagged
    // nothing are only bracketed at level 2
    tl->active = ctx->profiling >= (fallback ? 2 : 1);
    if (!tl->active) return PMX_OK;
    int rc = get_event(ctx, &tl->start);
    if (rc) return rc;
    rc = get_event(ctx, &tl->stop);
    if (rc) return rc;
    PMX_HIP(hipEventRecord(tl->start, ctx->stream));
    return PMX_OK;
}

int pmx_prof_end(pmx_ctx *ctx, pmx_timed_launch *tl)
{
    if (!tl->active) return PMX_OK;
    PMX_HIP(hipEventRecord(tl->stop, ctx->stream));
    ctx->timed.push_back(*tl);
    return PMX_OK;
}

static int fold_timed(pmx_ctx *ctx)
{
    if (ctx->timed.empty()) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &tl : ctx->timed) {
        float ms = 0.f;
        PMX_HIP(hipEventElapsedTime(&ms, tl.start, tl.stop));
        ctx->total_ms[tl.kernel_id] += (double)ms;
        ctx->launches[tl.kernel_id] += 1;
        ctx->event_pool.push_back(tl.start);
        ctx->event_pool.push_back(tl.stop);
    }
    ctx->timed.clear();
    return PMX_OK;
}

extern "C" {

// ---- measurement ---------------------------------------------------------------------------------

int pmx_ctx_set_profiling(pmx_ctx *ctx, int on)
{
    REQUIRE(ctx, "pmx_ctx_set_profiling: ctx is NULL");
    int rc = fold_timed(ctx);
    if (rc) return rc;
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    return PMX_OK;
}

int pmx_ctx_reset_kernel_times(pmx_ctx *ctx)
{
    REQUIRE(ctx, "pmx_ctx_reset_kernel_times: ctx is NULL");
    int rc = fold_timed(ctx);
    if (rc) return rc;
    for (int i = 0; i < PMX_KERNEL_COUNT_; i++) {
        ctx->total_ms[i] = 0;
        ctx->launches[i] = 0;
    }
    return PMX_OK;
}

int pmx_ctx_kernel_time(pmx_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx, "pmx_ctx_kernel_time: ctx is NULL");
    REQUIRE(kernel_id >= 0 && kernel_id < PMX_KERNEL_COUNT_, "pmx_ctx_kernel_time: bad kernel_id");
    int rc = fold_timed(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->total_ms[kernel_id];
    if (launches) *launches = ctx->launches[kernel_id];
    return PMX_OK;
}

const char *pmx_kernel_name(int kernel_id)
{
    switch (kernel_id) {
    case PMX_KERNEL_CC_DENSE: return "k_cc_dense";
    case PMX_KERNEL_CC_SPARSE: return "k_cc_sparse";
    case PMX_KERNEL_AUTOCORR: return "k_autocorr_pairs+edges";
    case PMX_KERNEL_CC_EVENTS: return "k_cc_events";
    default: return "?";
    }
}

// ---- device bit-vectors ---------------------------------------------------------------------------

// diagnostics only (not in the public header): copies `bytes` of the slab starting at byte offset `off`
// diagnostic: leaves a pattern in every scratch buffer of the context and in the LDS of every CU, so that a kernel that
// reads memory it has not written shows up as a deterministic mismatch (tools/repro_case.py)
__global__ void __launch_bounds__(256) k_debug_poison_lds(u32 pattern, u32 *sink)
{
    __shared__ u32 lds[16 * 1024 - 64];   // just under 64 KB
    for (u32 i = threadIdx.x; i < 16 * 1024 - 64; i += 256) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 61) % (16 * 1024 - 64)] != pattern) sink[0] = 1;   // keeps the stores alive
}

int pmx_debug_poison(pmx_ctx *ctx, uint32_t pattern, uint32_t mask)
{
    REQUIRE(ctx, "pmx_debug_poison: ctx is NULL");
    PMX_JOIN_SIDE(ctx);
    const int byte = (int)(pattern & 0xff);
    if ((mask & 1) && ctx->d_slab) PMX_HIP(hipMemsetAsync(ctx->d_slab, byte, ctx->slab_words * sizeof(u32), ctx->stream));
    if ((mask & 2) && ctx->d_slab2) PMX_HIP(hipMemsetAsync(ctx->d_slab2, byte, ctx->slab2_words * sizeof(u32), ctx->stream));
    if ((mask & 4) && ctx->d_slab_ac) PMX_HIP(hipMemsetAsync(ctx->d_slab_ac, byte, ctx->slab_ac_words * sizeof(u32), ctx->stream));
    if ((mask & 4) && ctx->d_slab_fb) PMX_HIP(hipMemsetAsync(ctx->d_slab_fb, byte, ctx->slab_fb_words * sizeof(u32), ctx->stream));
    if ((mask & 8) && ctx->d_flags) PMX_HIP(hipMemsetAsync(ctx->d_flags, byte, ctx->flags_bytes, ctx->stream));
    if ((mask & 16) && ctx->d_flags_cc) {
        PMX_HIP(hipMemsetAsync(ctx->d_flags_cc, byte, ctx->flags_cc_bytes, ctx->stream));
        ctx->flags_cc_zeroed[0] = ctx->flags_cc_zeroed[1] = ctx->flags_cc_dirty[0] = ctx->flags_cc_dirty[1] = 0;   // (neither area is clean now)
    }
    if ((mask & 32) && ctx->d_scratch) PMX_HIP(hipMemsetAsync(ctx->d_scratch, byte, ctx->scratch_words * sizeof(u64), ctx->stream));
    for (int i = 0; i < 3; i++)
        if ((mask & 64) && ctx->d_stage[i])
            PMX_HIP(hipMemsetAsync(ctx->d_stage[i], byte, ctx->stage_words[i] * sizeof(uint64_t), ctx->stream));
    if ((mask & 128) && ctx->d_out_stage) PMX_HIP(hipMemsetAsync(ctx->d_out_stage, byte, ctx->out_stage_words * sizeof(u64), ctx->stream));
    if (mask & 256) {
        int rc = pmx_ensure_scratch(ctx, 64);
        if (rc) return rc;
        hipLaunchKernelGGL(k_debug_poison_lds, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, pattern, (u32 *)ctx->d_scratch + 32);
        PMX_CHECK_LAUNCH("k_debug_poison_lds");
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_debug_set_max_workgroups(pmx_ctx *ctx, uint32_t n)
{
    REQUIRE(ctx, "pmx_debug_set_max_workgroups: ctx is NULL");
    ctx->debug_max_wg = n;
    return PMX_OK;
}

int pmx_debug_read_slab(pmx_ctx *ctx, uint64_t off, void *dst, uint64_t bytes)
{
    REQUIRE(ctx && dst && ctx->d_slab, "pmx_debug_read_slab: bad argument");
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    PMX_HIP(hipMemcpy(dst, (const char *)ctx->d_slab + off, bytes, hipMemcpyDeviceToHost));
    return PMX_OK;
}

int pmx_bits_alloc(pmx_ctx *ctx, uint64_t nbits, uint64_t **d_words)
{
    REQUIRE(ctx && d_words, "pmx_bits_alloc: NULL argument");
    PMX_HIP(hipSetDevice(ctx->device));
    const uint64_t nw = words_for(nbits);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (nw ? nw : 1) * sizeof(uint64_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pmx_set_error("pmx_bits_alloc: hipMalloc of %llu bytes failed: %s", (unsigned long long)(nw * 8),
                      hipGetErrorString(e));
        return PMX_ERR_NOMEM;
    }
    PMX_HIP(hipMemsetAsync(p, 0, (nw ? nw : 1) * sizeof(uint64_t), ctx->stream));
    *d_words = (uint64_t *)p;
    return PMX_OK;
}

int pmx_bits_free(pmx_ctx *ctx, uint64_t *d_words)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx, "pmx_bits_free: ctx is NULL");
    PMX_JOIN_SIDE(ctx);
    if (!d_words) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    PMX_HIP(hipFree(d_words));
    return PMX_OK;
}

int pmx_bits_clear(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words, "pmx_bits_clear: NULL argument");
    PMX_JOIN_SIDE(ctx);
    PMX_HIP(hipMemsetAsync(d_words, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
    return PMX_OK;
}

int pmx_bits_upload(pmx_ctx *ctx, uint64_t *d_words, const uint64_t *h_words, uint64_t nbits)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && h_words, "pmx_bits_upload: NULL argument");
    PMX_JOIN_SIDE(ctx);
    PMX_HIP(hipMemcpyAsync(d_words, h_words, words_for(nbits) * sizeof(uint64_t), hipMemcpyHostToDevice,
                           ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_bits_download(pmx_ctx *ctx, const uint64_t *d_words, uint64_t *h_words, uint64_t nbits)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && h_words, "pmx_bits_download: NULL argument");
    PMX_JOIN_SIDE(ctx);
    PMX_HIP(hipMemcpyAsync(h_words, d_words, words_for(nbits) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_bits_set_positions_dev(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_pos, uint64_t n)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && (d_pos || n == 0), "pmx_bits_set_positions_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    return pmx_launch_set_positions(ctx, d_words, nbits, d_pos, n);
}

int pmx_bits_set_positions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *h_pos, uint64_t n)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && (h_pos || n == 0), "pmx_bits_set_positions: NULL argument");
    PMX_JOIN_SIDE(ctx);
    if (n == 0) return PMX_OK;
    // positions are range-checked by the kernel itself (the smallest bad index comes back with the synchronisation this
    // entry point does anyway); on PMX_ERR_INVALID the bits of the valid positions have been set
    int rc = ensure_stage(ctx, 0, n);
    if (rc) return rc;
    rc = pmx_ensure_scratch(ctx, 4096);
    if (rc) return rc;
    u64 *d_bad = ctx->d_scratch + 8;
    const u64 none = ~0ull;
    PMX_HIP(hipMemcpyAsync(d_bad, &none, sizeof none, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(ctx->d_stage[0], h_pos, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_set_positions(ctx, d_words, nbits, (const int64_t *)ctx->d_stage[0], n, d_bad);
    if (rc) return rc;
    u64 bad = none;
    PMX_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));   // h_pos and the staging buffer may be reused by the caller
    if (bad != none) {
        pmx_set_error("pmx_bits_set_positions: position %lld at index %llu outside [0, %llu)", (long long)h_pos[bad],
                      (unsigned long long)bad, (unsigned long long)nbits);
        return PMX_ERR_INVALID;
    }
    return PMX_OK;
}

int pmx_bits_set_regions_dev(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_from,
                             const int64_t *d_to, uint64_t n)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && ((d_from && d_to) || n == 0), "pmx_bits_set_regions_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    return pmx_launch_set_regions(ctx, d_words, nbits, d_from, d_to, n);
}

int pmx_bits_set_regions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *h_from,
                         const int64_t *h_to, uint64_t n)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && ((h_from && h_to) || n == 0), "pmx_bits_set_regions: NULL argument");
    PMX_JOIN_SIDE(ctx);
    if (n == 0) return PMX_OK;
    for (uint64_t i = 0; i < n; i++) {
        if (h_to[i] < h_from[i]) continue;
        if (h_from[i] < 0 || (uint64_t)h_to[i] >= nbits) {
            pmx_set_error("pmx_bits_set_regions: interval [%lld, %lld] at index %llu outside [0, %llu)",
                          (long long)h_from[i], (long long)h_to[i], (unsigned long long)i,
                          (unsigned long long)nbits);
            return PMX_ERR_INVALID;
        }
    }
    int rc = ensure_stage(ctx, 0, n);
    if (rc) return rc;
    rc = ensure_stage(ctx, 1, n);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(ctx->d_stage[0], h_from, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(ctx->d_stage[1], h_to, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_set_regions(ctx, d_words, nbits, (const int64_t *)ctx->d_stage[0],
                                (const int64_t *)ctx->d_stage[1], n);
    if (rc) return rc;
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_bits_count(pmx_ctx *ctx, const uint64_t *d_words, uint64_t nbits, uint64_t *h_count)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_words && h_count, "pmx_bits_count: NULL argument");
    PMX_JOIN_SIDE(ctx);
    int rc = pmx_ensure_scratch(ctx, 4096);
    if (rc) return rc;
    PMX_HIP(hipMemsetAsync(ctx->d_scratch, 0, sizeof(u64), ctx->stream));
    rc = pmx_launch_count(ctx, d_words, nbits, ctx->d_scratch);
    if (rc) return rc;
    u64 v = 0;
    PMX_HIP(hipMemcpyAsync(&v, ctx->d_scratch, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    *h_count = (uint64_t)v;
    return PMX_OK;
}

// ---- stream-ordered feeding ------------------------------------------------------------------------

int pmx_host_alloc(pmx_ctx *ctx, uint64_t bytes, void **h)
{
    REQUIRE(ctx && h, "pmx_host_alloc: NULL argument");
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pmx_set_error("pmx_host_alloc: hipHostMalloc of %llu bytes failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return PMX_ERR_NOMEM;
    }
    *h = p;
    return PMX_OK;
}

int pmx_host_free(pmx_ctx *ctx, void *h)
{
    REQUIRE(ctx, "pmx_host_free: ctx is NULL");
    if (!h) return PMX_OK;
    (void)hipSetDevice(ctx->device);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->copy_stream));
    PMX_HIP(hipStreamSynchronize(ctx->copy_stream2));
    PMX_HIP(hipHostFree(h));
    return PMX_OK;
}

}   // extern "C"

// A staging slot of the feeders: device memory the copy stream fills while the kernels of the previous call still run on
// the context's stream.  A slot is taken again only after the event behind its last consumer has passed.
static int feed_acquire(pmx_ctx *ctx, size_t bytes, unsigned char **d, uint32_t *slot_out)
{
    const uint32_t slot = ctx->feed_next;
    ctx->feed_next = (slot + 1) % PMX_FEED_SLOTS;
    if (ctx->feed_used[slot]) PMX_HIP(hipEventSynchronize(ctx->feed_done[slot]));
    const size_t words = (bytes + 7) / 8 + 8;
    if (ctx->feed_words[slot] < words) {
        if (ctx->d_feed[slot]) PMX_HIP(hipFree(ctx->d_feed[slot]));
        ctx->d_feed[slot] = nullptr;
        ctx->feed_words[slot] = 0;
        const size_t want = words + words / 4;
        PMX_HIP(hipMalloc((void **)&ctx->d_feed[slot], want * sizeof(uint64_t)));
        ctx->feed_words[slot] = want;
    }
    *d = (unsigned char *)ctx->d_feed[slot];
    *slot_out = slot;
    return PMX_OK;
}

// copies queued on the copy stream -> visible to the kernels queued on the context's stream after this
static int feed_publish(pmx_ctx *ctx, hipStream_t cs = nullptr)
{
    PMX_HIP(hipEventRecord(ctx->feed_copied, cs ? cs : ctx->copy_stream));
    PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->feed_copied, 0));
    return PMX_OK;
}

static int feed_release(pmx_ctx *ctx, uint32_t slot)
{
    PMX_HIP(hipEventRecord(ctx->feed_done[slot], ctx->stream));
    ctx->feed_used[slot] = true;
    return PMX_OK;
}

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Host arrays -> one staging slot.  Arrays that lie in host memory at the offsets they take in the slot (back to back,
// each padded to 16 bytes: ffi.Context.host_packed lays them out so) go in ONE copy: a 0.1-MB copy costs ~13 us of the
// copy engine + the gap to the next one, whatever its size, and a genome is a hundred of them.
struct FeedPart {
    size_t off;        // byte offset in the slot
    const void *src;   // host array (nullptr / 0 bytes: none)
    size_t bytes;
};
static int feed_copy(pmx_ctx *ctx, unsigned char *d, const FeedPart *parts, uint32_t nparts, hipStream_t cs = nullptr)
{
    if (!cs) cs = ctx->copy_stream;

    const unsigned char *base = nullptr;
    size_t base_off = 0, end = 0;
    bool packed = true;
    uint32_t live = 0;
    for (uint32_t i = 0; i < nparts; i++) {
        if (!parts[i].src || parts[i].bytes == 0) continue;
        const unsigned char *p = (const unsigned char *)parts[i].src;
        if (!base) {
            base = p;
            base_off = parts[i].off;
        } else if (p != base + (parts[i].off - base_off)) {
            packed = false;
        }
        end = parts[i].off + parts[i].bytes;
        live++;
    }
    if (live == 0) return PMX_OK;
    if (packed && live > 1) {
        PMX_HIP(hipMemcpyAsync(d + base_off, base, end - base_off, hipMemcpyHostToDevice, cs));
        return PMX_OK;
    }
    for (uint32_t i = 0; i < nparts; i++)
        if (parts[i].src && parts[i].bytes)
            PMX_HIP(hipMemcpyAsync(d + parts[i].off, parts[i].src, parts[i].bytes, hipMemcpyHostToDevice, cs));
    return PMX_OK;
}

// ---- the side stream (pmx_bits_set_regions_ex with PMX_REGIONS_SIDE) ----
// One task = one vector: wait for the mark on the context's stream, clear, copy the intervals (copy_stream2) into a staging slot
// of the side ring, set the regions -- all on side_stream.  Nothing here touches ctx->stream, the feeders' ring or their events.
// (A worker thread issuing these calls was tried: the caller spends its time waiting for the feeders' staging slots anyway --
// the GPU paces a genome's feed --, so it bought nothing.)
static int side_run(pmx_ctx *ctx, const pmx_side_task &t)
{
    PMX_HIP(hipStreamWaitEvent(ctx->side_stream, t.fork, 0));
    const bool build = (t.flags & PMX_REGIONS_SORTED) != 0;   // (writes every word itself)
    if ((t.flags & PMX_REGIONS_CLEAR) && !build)
        PMX_HIP(hipMemsetAsync(t.d_words, 0, ((t.nbits + 63) / 64) * sizeof(uint64_t), ctx->side_stream));
    if (t.n == 0 && !build) return PMX_OK;
    const size_t o_last = align16((size_t)t.n * t.width_bytes);
    const uint32_t slot = ctx->side_next;
    ctx->side_next = (slot + 1) % PMX_SIDE_SLOTS;
    if (ctx->side_slot_used[slot]) PMX_HIP(hipEventSynchronize(ctx->side_slot_done[slot]));
    const size_t words = (2 * o_last + 7) / 8 + 8;
    if (ctx->side_words[slot] < words) {
        if (ctx->d_side[slot]) PMX_HIP(hipFree(ctx->d_side[slot]));
        ctx->d_side[slot] = nullptr;
        ctx->side_words[slot] = 0;
        const size_t want = words + words / 4;
        PMX_HIP(hipMalloc((void **)&ctx->d_side[slot], want * sizeof(uint64_t)));
        ctx->side_words[slot] = want;
    }
    unsigned char *d = (unsigned char *)ctx->d_side[slot];
    const FeedPart parts[2] = {{0, t.h_first, (size_t)t.n * t.width_bytes}, {o_last, t.h_last, (size_t)t.n * t.width_bytes}};
    int rc = feed_copy(ctx, d, parts, 2, ctx->copy_stream2);
    if (rc) return rc;
    PMX_HIP(hipEventRecord(ctx->side_copied, ctx->copy_stream2));
    PMX_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_copied, 0));
    if (build)
        rc = pmx_launch_regions_build_on(ctx, ctx->side_stream, t.d_words, t.nbits, d, d + o_last, t.width_bytes, t.n, t.first_offset,
                                         t.d_state + PMX_FEED_FIRST_OUT_OF_RANGE, t.d_state + PMX_FEED_REGIONS_UNSORTED);
    else
        rc = pmx_launch_set_regions_on(ctx, ctx->side_stream, t.d_words, t.nbits, d, d + o_last, t.width_bytes, t.n, t.first_offset,
                                       t.d_state ? t.d_state + PMX_FEED_FIRST_OUT_OF_RANGE : nullptr);
    if (rc) return rc;
    PMX_HIP(hipEventRecord(ctx->side_slot_done[slot], ctx->side_stream));
    ctx->side_slot_used[slot] = true;
    return PMX_OK;
}

static int side_submit(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *h_first, const void *h_last, uint32_t width_bytes,
                       uint64_t n, int64_t first_offset, uint64_t *d_state, uint32_t flags)
{
    pmx_side_task t;
    t.d_words = d_words; t.nbits = nbits; t.h_first = h_first; t.h_last = h_last; t.width_bytes = width_bytes;
    t.n = n; t.first_offset = first_offset; t.d_state = d_state; t.flags = flags;
    t.fork = ctx->side_fork;
    // behind everything queued on the context's stream so far (the vector's memory may come from a pool whose last user is a
    // kernel queued there), beside everything queued after
    PMX_HIP(hipEventRecord(t.fork, ctx->stream));
    const int rc = side_run(ctx, t);
    // the end of the side work so far: what pmx_side_join makes the context's stream wait for (recorded after a failure too --
    // whatever was queued is then still waited for)
    ctx->side_pending = true;
    PMX_HIP(hipEventRecord(ctx->side_done, ctx->side_stream));
    return rc;
}

extern "C" {

int pmx_feed_reads(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *h_pos, uint32_t pos_bytes,
                   const void *h_readlen, uint32_t len_bytes, const uint8_t *h_is_reverse, uint64_t n,
                   uint64_t reads_before, uint64_t *d_state)
{
    return pmx_feed_reads_ex(ctx, d_F, d_R, nbits, h_pos, pos_bytes, h_readlen, len_bytes, h_is_reverse, n, reads_before, d_state, 0);
}

int pmx_feed_reads_ex(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *h_pos, uint32_t pos_bytes,
                      const void *h_readlen, uint32_t len_bytes, const uint8_t *h_is_reverse, uint64_t n,
                      uint64_t reads_before, uint64_t *d_state, uint32_t flags)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    const bool whole = (flags & PMX_FEED_WHOLE_VECTORS) != 0;
    REQUIRE(!whole || reads_before == 0, "pmx_feed_reads_ex: PMX_FEED_WHOLE_VECTORS is for the first run of a chromosome (reads_before = 0)");
    if (whole && n == 0) {   // nothing to set: the promise is still that every word is written
        REQUIRE(ctx && d_F && d_R, "pmx_feed_reads_ex: NULL argument");
        PMX_HIP(hipMemsetAsync(d_F, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        PMX_HIP(hipMemsetAsync(d_R, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        return PMX_OK;
    }
    REQUIRE(ctx && d_F && d_R && d_state, "pmx_feed_reads: NULL argument");
    REQUIRE(n == 0 || (h_pos && h_readlen), "pmx_feed_reads: NULL read arrays");
    REQUIRE((pos_bytes == 4 || pos_bytes == 8) && (len_bytes == 0 || len_bytes == 2 || len_bytes == 4 || len_bytes == 8),
            "pmx_feed_reads: positions must be 4 or 8 bytes wide, read lengths 2, 4 or 8 (0: one int64 for all)");
    REQUIRE(nbits >= 1 && nbits < (1ull << 40), "pmx_feed_reads: nbits must be in [1, 2^40)");
    if (n == 0) return PMX_OK;
    const int64_t uniform_len = len_bytes == 0 ? *(const int64_t *)h_readlen : 0;   // (read now: the caller's word may be gone later)
    const size_t o_len = align16((size_t)n * pos_bytes), o_rev = o_len + align16((size_t)n * len_bytes);
    const size_t o_part = o_rev + align16((size_t)n);      // (PMX_FEED_WHOLE_VECTORS: per-workgroup sums of k_feed_build)
    unsigned char *d = nullptr;
    uint32_t slot = 0;
    int rc = feed_acquire(ctx, o_part + (whole ? (size_t)pmx_feed_build_blocks(nbits) * 64 : 0), &d, &slot);
    if (rc) return rc;
    const FeedPart parts[3] = {{0, h_pos, (size_t)n * pos_bytes}, {o_len, len_bytes ? h_readlen : nullptr, (size_t)n * len_bytes},
                               {o_rev, h_is_reverse, (size_t)n}};
    rc = feed_copy(ctx, d, parts, 3);
    if (rc) return rc;
    rc = feed_publish(ctx);
    if (rc) return rc;
    rc = pmx_launch_feed_reads(ctx, d_F, d_R, nbits, d, pos_bytes, d + o_len, len_bytes, uniform_len, h_is_reverse ? d + o_rev : nullptr,
                               n, reads_before, d_state, whole ? (uint64_t *)(d + o_part) : nullptr);
    if (rc) return rc;
    return feed_release(ctx, slot);
}

int pmx_feed_reads_dev(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const int32_t *d_pos, const int32_t *d_readlen,
                       const uint8_t *d_is_reverse, uint64_t n, uint64_t reads_before, uint64_t *d_state, uint32_t flags)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    const bool whole = (flags & PMX_FEED_WHOLE_VECTORS) != 0;
    REQUIRE(!whole || reads_before == 0, "pmx_feed_reads_dev: PMX_FEED_WHOLE_VECTORS is for the first run of a chromosome (reads_before = 0)");
    if (whole && n == 0) {
        REQUIRE(ctx && d_F && d_R, "pmx_feed_reads_dev: NULL argument");
        PMX_HIP(hipMemsetAsync(d_F, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        PMX_HIP(hipMemsetAsync(d_R, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        return PMX_OK;
    }
    REQUIRE(ctx && d_F && d_R && d_state, "pmx_feed_reads_dev: NULL argument");
    REQUIRE(n == 0 || (d_pos && d_readlen && d_is_reverse), "pmx_feed_reads_dev: NULL read arrays");
    REQUIRE(nbits >= 1 && nbits < (1ull << 40), "pmx_feed_reads_dev: nbits must be in [1, 2^40)");
    if (n == 0) return PMX_OK;
    // nothing to copy: the staging slot only holds the per-workgroup sums of the whole-vector builder
    unsigned char *d = nullptr;
    uint32_t slot = 0;
    int rc = PMX_OK;
    if (whole) {
        rc = feed_acquire(ctx, (size_t)pmx_feed_build_blocks(nbits) * 64, &d, &slot);
        if (rc) return rc;
    }
    rc = pmx_launch_feed_reads(ctx, d_F, d_R, nbits, d_pos, 4, d_readlen, 4, 0, d_is_reverse, n, reads_before, d_state,
                               whole ? (uint64_t *)d : nullptr);
    if (rc) return rc;
    return whole ? feed_release(ctx, slot) : PMX_OK;
}

int pmx_feed_reads_delta16(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const uint16_t *h_words, uint64_t n,
                           const uint32_t *h_seg_start, const int32_t *h_seg_base, uint32_t nseg, const void *h_readlen,
                           uint32_t len_bytes, uint64_t reads_before, uint64_t *d_state, uint32_t flags)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    const bool whole = (flags & PMX_FEED_WHOLE_VECTORS) != 0;
    REQUIRE(!whole || reads_before == 0, "pmx_feed_reads_delta16: PMX_FEED_WHOLE_VECTORS is for the first run of a chromosome (reads_before = 0)");
    if (whole && n == 0) {
        REQUIRE(ctx && d_F && d_R, "pmx_feed_reads_delta16: NULL argument");
        PMX_HIP(hipMemsetAsync(d_F, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        PMX_HIP(hipMemsetAsync(d_R, 0, words_for(nbits) * sizeof(uint64_t), ctx->stream));
        return PMX_OK;
    }
    REQUIRE(ctx && d_F && d_R && d_state, "pmx_feed_reads_delta16: NULL argument");
    REQUIRE(n == 0 || (h_words && h_seg_start && h_seg_base && h_readlen && nseg >= 1), "pmx_feed_reads_delta16: NULL read arrays");
    REQUIRE(len_bytes == 0 || len_bytes == 2 || len_bytes == 4 || len_bytes == 8,
            "pmx_feed_reads_delta16: read lengths must be 2, 4 or 8 bytes wide (0: one int64 for all)");
    REQUIRE(nbits >= 1 && nbits < (1ull << 31), "pmx_feed_reads_delta16: nbits must be in [1, 2^31) (32-bit positions: what a BAM file holds)");
    REQUIRE(n < (1ull << 32), "pmx_feed_reads_delta16: at most 2^32 - 1 reads per call");
    if (n == 0) return PMX_OK;
    // the segment table as the kernel will index it: starts increasing from 0, at most 1024 reads each, ends with n
    // (checked here, on a few hundred words: the kernel never reads beyond the run)
    REQUIRE(h_seg_start[0] == 0 && h_seg_start[nseg] == n, "pmx_feed_reads_delta16: the segment table must start at 0 and end with n");
    for (uint32_t s = 0; s < nseg; s++)
        REQUIRE(h_seg_start[s + 1] > h_seg_start[s] && h_seg_start[s + 1] - h_seg_start[s] <= 1024u,
                "pmx_feed_reads_delta16: segments hold 1..1024 reads");
    const int64_t uniform_len = len_bytes == 0 ? *(const int64_t *)h_readlen : 0;
    // slot: [words][segment starts (nseg + 1)][segment bases][read lengths][expanded positions (device only)]
    const size_t o_start = align16((size_t)n * 2), o_base = o_start + align16(((size_t)nseg + 1) * 4);
    const size_t o_len = o_base + align16((size_t)nseg * 4), o_pos = o_len + align16((size_t)n * len_bytes);
    const size_t o_part = o_pos + align16((size_t)n * 4);
    unsigned char *d = nullptr;
    uint32_t slot = 0;
    int rc = feed_acquire(ctx, o_part + (whole ? (size_t)pmx_feed_build_blocks(nbits) * 64 : 0), &d, &slot);
    if (rc) return rc;
    const FeedPart parts[4] = {{0, h_words, (size_t)n * 2}, {o_start, h_seg_start, ((size_t)nseg + 1) * 4}, {o_base, h_seg_base, (size_t)nseg * 4},
                               {o_len, len_bytes ? h_readlen : nullptr, (size_t)n * len_bytes}};
    rc = feed_copy(ctx, d, parts, 4);
    if (rc) return rc;
    rc = feed_publish(ctx);
    if (rc) return rc;
    // (the expansion stays on the context's stream: on the copy stream it held up the next copy, on a stream of its own it ran
    // beside the builder kernels and slowed them by what it gained -- the feed is paced by the GPU's work, not by the queues)
    rc = pmx_launch_feed_expand16(ctx, d, d + o_start, d + o_base, nseg, n, d + o_pos);
    if (rc) return rc;
    rc = pmx_launch_feed_reads(ctx, d_F, d_R, nbits, d + o_pos, 4, d + o_len, len_bytes, uniform_len, nullptr, n, reads_before, d_state,
                               whole ? (uint64_t *)(d + o_part) : nullptr);
    if (rc) return rc;
    return feed_release(ctx, slot);
}

int pmx_bits_set_regions_ex(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *h_first, const void *h_last,
                            uint32_t width_bytes, uint64_t n, int64_t first_offset, uint64_t *d_state, uint32_t flags)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx && d_words && ((h_first && h_last) || n == 0), "pmx_bits_set_regions_ex: NULL argument");
    REQUIRE(width_bytes == 4 || width_bytes == 8, "pmx_bits_set_regions_ex: interval ends must be 4 or 8 bytes wide");
    REQUIRE(!(flags & ~(PMX_REGIONS_CLEAR | PMX_REGIONS_SIDE | PMX_REGIONS_SORTED)), "pmx_bits_set_regions_ex: unknown flag");
    REQUIRE(!(flags & PMX_REGIONS_SORTED) || d_state, "pmx_bits_set_regions_ex: PMX_REGIONS_SORTED needs d_state (order violations are recorded there)");
    if (flags & PMX_REGIONS_SORTED) flags |= PMX_REGIONS_CLEAR;
    if (flags & PMX_REGIONS_SIDE) {
        if (n == 0 && !(flags & PMX_REGIONS_CLEAR)) return PMX_OK;
        return side_submit(ctx, d_words, nbits, h_first, h_last, width_bytes, n, first_offset, d_state, flags);
    }
    PMX_JOIN_SIDE(ctx);
    if (flags & PMX_REGIONS_SORTED) {
        const size_t o_last = align16((size_t)n * width_bytes);
        unsigned char *d = nullptr;
        uint32_t slot = 0;
        int rc = feed_acquire(ctx, 2 * o_last, &d, &slot);
        if (rc) return rc;
        const FeedPart parts[2] = {{0, h_first, (size_t)n * width_bytes}, {o_last, h_last, (size_t)n * width_bytes}};
        rc = feed_copy(ctx, d, parts, 2, ctx->copy_stream2);
        if (rc) return rc;
        rc = feed_publish(ctx, ctx->copy_stream2);
        if (rc) return rc;
        rc = pmx_launch_regions_build_on(ctx, ctx->stream, d_words, nbits, d, d + o_last, width_bytes, n, first_offset,
                                         d_state + PMX_FEED_FIRST_OUT_OF_RANGE, d_state + PMX_FEED_REGIONS_UNSORTED);
        if (rc) return rc;
        return feed_release(ctx, slot);
    }
    if (flags & PMX_REGIONS_CLEAR) PMX_HIP(hipMemsetAsync(d_words, 0, ((nbits + 63) / 64) * sizeof(uint64_t), ctx->stream));
    return pmx_bits_set_regions_async(ctx, d_words, nbits, h_first, h_last, width_bytes, n, first_offset, d_state);
}

int pmx_bits_set_regions_dev_ex(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const uint32_t *d_first, const uint32_t *d_last,
                                uint64_t n, int64_t first_offset, uint64_t *d_state, uint32_t flags)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx && d_words && ((d_first && d_last) || n == 0), "pmx_bits_set_regions_dev_ex: NULL argument");
    REQUIRE(!(flags & ~(PMX_REGIONS_CLEAR | PMX_REGIONS_SORTED)), "pmx_bits_set_regions_dev_ex: unknown flag (PMX_REGIONS_SIDE is for host arrays)");
    REQUIRE(!(flags & PMX_REGIONS_SORTED) || d_state, "pmx_bits_set_regions_dev_ex: PMX_REGIONS_SORTED needs d_state (order violations are recorded there)");
    PMX_JOIN_SIDE(ctx);
    if (flags & PMX_REGIONS_SORTED)
        return pmx_launch_regions_build_on(ctx, ctx->stream, d_words, nbits, d_first, d_last, 4, n, first_offset,
                                           d_state + PMX_FEED_FIRST_OUT_OF_RANGE, d_state + PMX_FEED_REGIONS_UNSORTED);
    if (flags & PMX_REGIONS_CLEAR) PMX_HIP(hipMemsetAsync(d_words, 0, ((nbits + 63) / 64) * sizeof(uint64_t), ctx->stream));
    if (n == 0) return PMX_OK;
    return pmx_launch_set_regions_w(ctx, d_words, nbits, d_first, d_last, 4, n, first_offset,
                                    d_state ? d_state + PMX_FEED_FIRST_OUT_OF_RANGE : nullptr);
}

int pmx_bits_set_regions_async(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *h_first, const void *h_last,
                               uint32_t width_bytes, uint64_t n, int64_t first_offset, uint64_t *d_state)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx && d_words && ((h_first && h_last) || n == 0), "pmx_bits_set_regions_async: NULL argument");
    PMX_JOIN_SIDE(ctx);
    REQUIRE(width_bytes == 4 || width_bytes == 8, "pmx_bits_set_regions_async: interval ends must be 4 or 8 bytes wide");
    if (n == 0) return PMX_OK;
    const size_t o_last = align16((size_t)n * width_bytes);
    unsigned char *d = nullptr;
    uint32_t slot = 0;
    int rc = feed_acquire(ctx, 2 * o_last, &d, &slot);
    if (rc) return rc;
    // (on the second copy stream: a chromosome's intervals are 0.1-0.2 MB between two 5-10 MB read copies -- in one queue
    // they cost the copy engine 20 us + a gap either side, per chromosome)
    const FeedPart parts[2] = {{0, h_first, (size_t)n * width_bytes}, {o_last, h_last, (size_t)n * width_bytes}};
    rc = feed_copy(ctx, d, parts, 2, ctx->copy_stream2);
    if (rc) return rc;
    rc = feed_publish(ctx, ctx->copy_stream2);
    if (rc) return rc;
    rc = pmx_launch_set_regions_w(ctx, d_words, nbits, d, d + o_last, width_bytes, n, first_offset,
                                  d_state ? d_state + PMX_FEED_FIRST_OUT_OF_RANGE : nullptr);
    if (rc) return rc;
    return feed_release(ctx, slot);
}

int pmx_bits_build_batch(pmx_ctx *ctx, uint32_t njobs, const pmx_build_job *jobs, uint32_t pos_bytes)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx && (jobs || njobs == 0), "pmx_bits_build_batch: NULL argument");
    PMX_JOIN_SIDE(ctx);
    REQUIRE(pos_bytes == 4 || pos_bytes == 8, "pmx_bits_build_batch: positions must be 4 or 8 bytes wide");
    if (njobs == 0) return PMX_OK;
    // one error word per job of the batches since the last status call
    const size_t need = (size_t)ctx->build_err_jobs + njobs;
    if (ctx->build_err_cap < need) {
        u64 *grown = nullptr;
        const size_t cap = need * 2 + 64;
        PMX_HIP(hipMalloc((void **)&grown, cap * sizeof(u64)));
        PMX_HIP(hipMemsetAsync(grown, 0, cap * sizeof(u64), ctx->stream));
        if (ctx->d_build_err) {
            if (ctx->build_err_jobs)
                PMX_HIP(hipMemcpyAsync(grown, ctx->d_build_err, ctx->build_err_jobs * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
            PMX_HIP(hipStreamSynchronize(ctx->stream));
            PMX_HIP(hipFree(ctx->d_build_err));
        }
        ctx->d_build_err = grown;
        ctx->build_err_cap = cap;
    }
    for (uint32_t i = 0; i < njobs; i++) {
        const pmx_build_job &jb = jobs[i];
        REQUIRE(jb.nbits >= 1 && jb.nbits < (1ull << 40), "pmx_bits_build_batch: nbits must be in [1, 2^40)");
        REQUIRE((jb.n_f == 0 || (jb.d_F && jb.h_fpos)) && (jb.n_r == 0 || (jb.d_R && jb.h_rpos)) &&
                    (jb.n_iv == 0 || (jb.d_M && jb.h_first && jb.h_last)),
                "pmx_bits_build_batch: a job has entries but no vector / array for them");
        const size_t nw = (size_t)words_for(jb.nbits) * sizeof(uint64_t);
        if (jb.d_F) PMX_HIP(hipMemsetAsync(jb.d_F, 0, nw, ctx->stream));
        if (jb.d_R) PMX_HIP(hipMemsetAsync(jb.d_R, 0, nw, ctx->stream));
        if (jb.d_M) PMX_HIP(hipMemsetAsync(jb.d_M, 0, nw, ctx->stream));
        const size_t o_r = align16((size_t)jb.n_f * pos_bytes), o_a = o_r + align16((size_t)jb.n_r * pos_bytes),
                     o_b = o_a + align16((size_t)jb.n_iv * pos_bytes), total = o_b + align16((size_t)jb.n_iv * pos_bytes);
        if (total == 0) continue;
        unsigned char *d = nullptr;
        uint32_t slot = 0;
        int rc = feed_acquire(ctx, total, &d, &slot);
        if (rc) return rc;
        const FeedPart parts[4] = {{0, jb.h_fpos, (size_t)jb.n_f * pos_bytes}, {o_r, jb.h_rpos, (size_t)jb.n_r * pos_bytes},
                                   {o_a, jb.h_first, (size_t)jb.n_iv * pos_bytes}, {o_b, jb.h_last, (size_t)jb.n_iv * pos_bytes}};
        rc = feed_copy(ctx, d, parts, 4);
        if (rc) return rc;
        rc = feed_publish(ctx);
        if (rc) return rc;
        uint64_t *err = (uint64_t *)(ctx->d_build_err + ctx->build_err_jobs + i);
        if (jb.n_f) rc = pmx_launch_set_positions_w(ctx, jb.d_F, jb.nbits, d, pos_bytes, jb.n_f, err);
        if (!rc && jb.n_r) rc = pmx_launch_set_positions_w(ctx, jb.d_R, jb.nbits, d + o_r, pos_bytes, jb.n_r, err);
        if (!rc && jb.n_iv) rc = pmx_launch_set_regions_w(ctx, jb.d_M, jb.nbits, d + o_a, d + o_b, pos_bytes, jb.n_iv, 0, err);
        if (rc) return rc;
        rc = feed_release(ctx, slot);
        if (rc) return rc;
    }
    ctx->build_err_jobs += njobs;
    return PMX_OK;
}

int pmx_bits_build_status(pmx_ctx *ctx)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx, "pmx_bits_build_status: ctx is NULL");
    PMX_JOIN_SIDE(ctx);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    const size_t n = ctx->build_err_jobs;
    if (n == 0) return PMX_OK;
    std::vector<u64> err(n);
    PMX_HIP(hipMemcpy(err.data(), ctx->d_build_err, n * sizeof(u64), hipMemcpyDeviceToHost));
    PMX_HIP(hipMemsetAsync(ctx->d_build_err, 0, n * sizeof(u64), ctx->stream));
    ctx->build_err_jobs = 0;
    for (size_t i = 0; i < n; i++)
        if (err[i]) {
            pmx_set_error("pmx_bits_build_batch: job %llu of the batches since the last status call has a position or interval "
                          "outside its vector (first at index %llu of its array)",
                          (unsigned long long)i, (unsigned long long)(PMX_FEED_ERR_BASE - err[i]));
            return PMX_ERR_INVALID;
        }
    return PMX_OK;
}

// ---- hot path -------------------------------------------------------------------------------------

static int check_shift_args(uint64_t nbits, uint32_t max_shift, const char *who)
{
    if (nbits == 0 || nbits >= (1ull << 40)) {
        pmx_set_error("%s: nbits must be in [1, 2^40)", who);
        return PMX_ERR_INVALID;
    }
    if (max_shift > 65535) {
        pmx_set_error("%s: max_shift must be <= 65535", who);
        return PMX_ERR_INVALID;
    }
    return PMX_OK;
}

int pmx_mappable_len_dev(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_shift, uint32_t flags,
                         uint64_t *d_out)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_M && d_out, "pmx_mappable_len_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    int rc = check_shift_args(nbits, max_shift, "pmx_mappable_len_dev");
    if (rc) return rc;
    REQUIRE(!((flags & PMX_FLAG_FORCE_DENSE) && (flags & PMX_FLAG_FORCE_SPARSE)),
            "pmx_mappable_len_dev: FORCE_DENSE and FORCE_SPARSE are exclusive");
    if (!(flags & PMX_FLAG_FORCE_DENSE) && pmx_sparse_supported(max_shift > 3 ? max_shift : 3, 1)) {
        rc = pmx_ensure_scratch(ctx, pmx_autocorr_scratch_words(max_shift));
        if (rc) return rc;
        pmx_job job = {nullptr, nullptr, d_M, nbits, d_out, (uint64_t *)ctx->d_scratch};
        return pmx_launch_autocorr_edges_batch(ctx, &job, 1, max_shift, 0, 1, max_shift, max_shift + 1);
    }
    PMX_HIP(hipMemsetAsync(d_out, 0, ((size_t)max_shift + 1) * sizeof(u64), ctx->stream));
    return pmx_launch_autocorr_dense(ctx, d_M, nbits, max_shift, (u64 *)d_out);
}

int pmx_mappable_len_batch_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_M, const uint64_t *nbits,
                               uint32_t max_shift, uint32_t flags, uint64_t *const *d_out)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    REQUIRE(ctx && (njobs == 0 || (d_M && nbits && d_out)), "pmx_mappable_len_batch_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    REQUIRE(!((flags & PMX_FLAG_FORCE_DENSE) && (flags & PMX_FLAG_FORCE_SPARSE)),
            "pmx_mappable_len_batch_dev: FORCE_DENSE and FORCE_SPARSE are exclusive");
    if (njobs == 0) return PMX_OK;
    for (uint32_t i = 0; i < njobs; i++) {
        REQUIRE(d_M[i] && d_out[i], "pmx_mappable_len_batch_dev: NULL vector or output in the batch");
        int rc = check_shift_args(nbits[i], max_shift, "pmx_mappable_len_batch_dev");
        if (rc) return rc;
    }
    if ((flags & PMX_FLAG_FORCE_DENSE) || !pmx_sparse_supported(max_shift > 3 ? max_shift : 3, 1)) {
        for (uint32_t i = 0; i < njobs; i++) {
            int rc = pmx_mappable_len_dev(ctx, d_M[i], nbits[i], max_shift, flags, d_out[i]);
            if (rc) return rc;
        }
        return PMX_OK;
    }
    const uint32_t chunk = pmx_autocorr_batch_jobs();
    const size_t ac_words = pmx_autocorr_scratch_words(max_shift);
    std::vector<pmx_job> jobs(njobs < chunk ? njobs : chunk);
    for (uint32_t lo = 0; lo < njobs; lo += chunk) {
        const uint32_t n = njobs - lo < chunk ? njobs - lo : chunk;
        int rc = pmx_ensure_scratch(ctx, (size_t)n * ac_words);
        if (rc) return rc;
        for (uint32_t i = 0; i < n; i++) {
            jobs[i].d_F = nullptr;
            jobs[i].d_R = nullptr;
            jobs[i].d_M = d_M[lo + i];
            jobs[i].nbits = nbits[lo + i];
            jobs[i].d_out = d_out[lo + i];
            jobs[i].d_out2 = (uint64_t *)(ctx->d_scratch + (size_t)i * ac_words);
        }
        rc = pmx_launch_autocorr_edges_batch(ctx, jobs.data(), n, max_shift, 0, 1, max_shift, max_shift + 1);
        if (rc) return rc;
    }
    return PMX_OK;
}

// one chromosome through the dense kernels (atomics into a zeroed block)
// The hint a caller without counts would give (include/pymasc_amd.h: PMX_FLAG_WINDOW_ONLY / PMX_FLAG_DEEP_LISTS), from a
// sample of the vectors themselves: k_density_probe on the (at most 64) largest chromosomes of the batch, one copy, one
// synchronisation.  Same rule as pymasc_amd/calculator.py (window_only_hint / deep_lists_hint: the AVERAGE tile against the
// event kernel's list capacities), weighted by chromosome length: the batch takes the choice that most of its positions ask for.
static int auto_density_hint(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                             const uint64_t *const *d_M, const uint64_t *nbits, uint32_t max_shift, bool has_m, uint32_t *hint)
{
    *hint = 0;
    if (!ctx->d_probe) {
        PMX_HIP(hipMalloc((void **)&ctx->d_probe, PMX_PROBE_JOBS * 4 * sizeof(uint32_t)));
        PMX_HIP(hipHostMalloc((void **)&ctx->h_probe, PMX_PROBE_JOBS * 4 * sizeof(uint32_t), hipHostMallocDefault));
    }
    std::vector<uint32_t> order(njobs);
    for (uint32_t i = 0; i < njobs; i++) order[i] = i;
    if (njobs > PMX_PROBE_JOBS) {
        std::partial_sort(order.begin(), order.begin() + PMX_PROBE_JOBS, order.end(),
                          [&](uint32_t a, uint32_t b) { return nbits[a] > nbits[b]; });
        order.resize(PMX_PROBE_JOBS);
    }
    pmx_probe_jobs pj;
    memset(&pj, 0, sizeof pj);
    const uint32_t n = (uint32_t)order.size();
    for (uint32_t k = 0; k < n; k++) {
        pj.F[k] = (const u32 *)d_F[order[k]];
        pj.R[k] = (const u32 *)d_R[order[k]];
        pj.M[k] = has_m ? (const u32 *)d_M[order[k]] : nullptr;
        pj.nbits[k] = nbits[order[k]];
    }
    int rc = pmx_launch_density_probe(ctx, &pj, n, ctx->d_probe);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(ctx->h_probe, ctx->d_probe, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    // list capacities per 64-Kbit tile: kernels_events.h (EV_POOL_SMALL / EV_POOL_DEEP / EV_CAPF + EV_CAPR + EV_CAPE_SMALL / EV_CAPE)
    const double POOL = 2416, POOL_NCC = 768 + 1000 + 1536, POOL_DEEP = 4328, EDGES = 1536, CAPF = 768, CAPR = 1000, EDGES_BIG = 384;
    double w_window = 0, w_deep = 0, w_all = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t *c = ctx->h_probe + 4 * k;
        if (!c[3]) continue;
        const double per_tile = 65536.0 / (double)c[3];
        const double f = c[0] * per_tile, r = c[1] * per_tile, e = has_m ? c[2] * per_tile : 0.0;
        bool window, deep = false;
        if (max_shift <= 1023) {
            window = e > EDGES || f + r + e > (has_m ? POOL_DEEP : POOL_NCC);
            deep = !window && has_m && f + r + e > 0.85 * POOL;
        } else {
            window = f > CAPF || r > CAPR || e > EDGES_BIG;
        }
        const double w = (double)pj.nbits[k];
        w_all += w;
        if (window) w_window += w;
        else if (deep) w_deep += w;
    }
    if (w_all > 0 && w_window > 0.5 * w_all) *hint = PMX_FLAG_WINDOW_ONLY;
    else if (w_all > 0 && w_window + w_deep > 0.5 * w_all) *hint = PMX_FLAG_DEEP_LISTS;
    return PMX_OK;
}

static int cc_dense_one(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M, uint64_t nbits,
                        uint32_t max_shift, uint32_t read_len, bool do_ncc, bool do_mlen, uint64_t *d_out)
{
    const uint32_t stride = max_shift + 1;
    u64 *out = (u64 *)d_out;
    PMX_HIP(hipMemsetAsync(out, 0, (size_t)PMX_NROWS * stride * sizeof(u64), ctx->stream));
    u64 *scal = out + (size_t)PMX_ROW_SCALARS * stride;
    int rc = pmx_launch_count(ctx, d_F, nbits, scal + 0);
    if (rc) return rc;
    rc = pmx_launch_count(ctx, d_R, nbits, scal + 1);
    if (rc) return rc;
    rc = pmx_launch_cc_dense(ctx, d_F, d_R, d_M, nbits, max_shift, read_len, do_ncc, out, stride, scal, 2);
    if (rc) return rc;
    if (d_M && do_mlen) {
        const uint32_t c = read_len - 1;
        const uint32_t far = max_shift > c ? max_shift - c : 0;   // lags |c - d| over d in [0, max_shift]
        const uint32_t max_lag = c > far ? c : far;
        rc = pmx_launch_count(ctx, d_M, nbits, scal + 2);
        if (rc) return rc;
        rc = pmx_ensure_scratch(ctx, (size_t)max_lag + 1 + 16);
        if (rc) return rc;
        PMX_HIP(hipMemsetAsync(ctx->d_scratch, 0, ((size_t)max_lag + 1) * sizeof(u64), ctx->stream));
        rc = pmx_launch_autocorr_dense(ctx, d_M, nbits, max_lag, ctx->d_scratch);
        if (rc) return rc;
        rc = pmx_launch_mlen_map(ctx, ctx->d_scratch, max_shift, read_len, out + (size_t)PMX_ROW_MLEN * stride);
        if (rc) return rc;
    }
    return PMX_OK;
}

static int cc_batch_impl(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                         const uint64_t *const *d_M, const uint64_t *nbits, uint32_t max_shift, uint32_t read_len,
                         uint32_t flags, uint64_t *const *d_out, const uint32_t *tile_first, const uint32_t *tile_count);

int pmx_cc_batch_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                     const uint64_t *const *d_M, const uint64_t *nbits, uint32_t max_shift, uint32_t read_len,
                     uint32_t flags, uint64_t *const *d_out)
{
    return cc_batch_impl(ctx, njobs, d_F, d_R, d_M, nbits, max_shift, read_len, flags, d_out, nullptr, nullptr);
}

int pmx_cc_batch_ranges_dev(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                            const uint64_t *const *d_M, const uint64_t *nbits, const uint32_t *tile_first,
                            const uint32_t *tile_count, uint32_t max_shift, uint32_t read_len, uint32_t flags,
                            uint64_t *const *d_out)
{
    REQUIRE(tile_first && tile_count && nbits, "pmx_cc_batch_ranges_dev: NULL argument");
    REQUIRE(max_shift >= 3 && max_shift <= 1023, "pmx_cc_batch_ranges_dev: tile ranges are for 3 <= max_shift <= 1023");
    REQUIRE(read_len >= 1 && read_len <= 1024, "pmx_cc_batch_ranges_dev: read_len must be in [1, 1024] (the set-bit kernels)");
    REQUIRE(!(flags & (PMX_FLAG_FORCE_DENSE | PMX_FLAG_WINDOW_ONLY | PMX_FLAG_SKIP_MLEN)),
            "pmx_cc_batch_ranges_dev: not with FORCE_DENSE / WINDOW_ONLY / SKIP_MLEN (the event kernel takes the ranges, mappable lengths included)");
    for (uint32_t i = 0; i < njobs; i++) {
        const uint64_t ntiles = (nbits[i] + PMX_RANGE_TILE_BITS - 1) / PMX_RANGE_TILE_BITS;
        REQUIRE(tile_count[i] >= 1 && (uint64_t)tile_first[i] + tile_count[i] <= ntiles,
                "pmx_cc_batch_ranges_dev: a range must hold 1 .. ceil(nbits / 65536) - tile_first tiles of its chromosome");
    }
    // (a hint is needed: the density probe would decide per rank; the event kernel is what takes ranges)
    return cc_batch_impl(ctx, njobs, d_F, d_R, d_M, nbits, max_shift, read_len, flags | PMX_FLAG_EVENTS_HINT, d_out, tile_first, tile_count);
}

static int cc_batch_impl(pmx_ctx *ctx, uint32_t njobs, const uint64_t *const *d_F, const uint64_t *const *d_R,
                         const uint64_t *const *d_M, const uint64_t *nbits, uint32_t max_shift, uint32_t read_len,
                         uint32_t flags, uint64_t *const *d_out, const uint32_t *tile_first, const uint32_t *tile_count)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && d_F && d_R && nbits && d_out, "pmx_cc_batch_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    REQUIRE(read_len >= 1 && read_len <= 65535, "pmx_cc_batch_dev: read_len must be in [1, 65535]");
    REQUIRE(!((flags & PMX_FLAG_FORCE_DENSE) && (flags & PMX_FLAG_FORCE_SPARSE)),
            "pmx_cc_batch_dev: FORCE_DENSE and FORCE_SPARSE are exclusive");
    if (njobs == 0) return PMX_OK;
    if (max_shift < 3) {
        // The kernels write four scalars into a row of max_shift + 1 words, so they run with at least 3 shifts: compute
        // 4 columns wide into staging blocks and copy the leading max_shift + 1 columns of every row (shifts are
        // independent, mscc.pyx:288; the scalar row is cut to max_shift + 1 entries, include/pymasc_amd.h).
        const uint32_t kstride = 4, stride = max_shift + 1;
        const size_t words = (size_t)njobs * PMX_NROWS * kstride;
        if (ctx->pad_stage_words < words) {
            if (ctx->d_pad_stage) {
                PMX_HIP(hipStreamSynchronize(ctx->stream));
                PMX_HIP(hipFree(ctx->d_pad_stage));
                ctx->d_pad_stage = nullptr;
                ctx->pad_stage_words = 0;
            }
            PMX_HIP(hipMalloc((void **)&ctx->d_pad_stage, words * sizeof(u64)));
            ctx->pad_stage_words = words;
        }
        std::vector<uint64_t *> wide(njobs);
        for (uint32_t i = 0; i < njobs; i++) {
            REQUIRE(d_out[i], "pmx_cc_batch_dev: NULL output in the batch");
            wide[i] = (uint64_t *)ctx->d_pad_stage + (size_t)i * PMX_NROWS * kstride;
        }
        int rc = cc_batch_impl(ctx, njobs, d_F, d_R, d_M, nbits, 3, read_len, flags, wide.data(), nullptr, nullptr);
        if (rc) return rc;
        for (uint32_t i = 0; i < njobs; i++)
            PMX_HIP(hipMemcpy2DAsync(d_out[i], stride * sizeof(u64), wide[i], kstride * sizeof(u64), stride * sizeof(u64),
                                     PMX_NROWS, hipMemcpyDeviceToDevice, ctx->stream));
        return PMX_OK;
    }
    const bool has_m = d_M != nullptr && d_M[0] != nullptr;
    for (uint32_t i = 0; i < njobs; i++) {
        REQUIRE(d_F[i] && d_R[i] && d_out[i], "pmx_cc_batch_dev: NULL vector or output in the batch");
        REQUIRE((d_M != nullptr && d_M[i] != nullptr) == has_m,
                "pmx_cc_batch_dev: mappability must be given for all jobs of a batch or for none");
        int rc = check_shift_args(nbits[i], max_shift, "pmx_cc_batch_dev");
        if (rc) return rc;
    }
    const bool do_ncc = !(flags & PMX_FLAG_SKIP_NCC);
    const bool do_mlen = has_m && !(flags & PMX_FLAG_SKIP_MLEN);
    const uint32_t stride = max_shift + 1;
    const uint32_t c = read_len - 1;
    const uint32_t far = max_shift > c ? max_shift - c : 0;   // lags |c - d| over d in [0, max_shift]
    const uint32_t max_lag = c > far ? c : far;

    // The set-bit kernels are correct for any input and faster than the dense ones unless the occupancy
    // vectors are pathologically dense, so they are the default wherever the geometry is supported.
    const bool sparse_ok = pmx_sparse_supported(max_shift, read_len) != 0;
    if ((flags & PMX_FLAG_FORCE_SPARSE) && !sparse_ok) {
        pmx_set_error("pmx_cc_batch_dev: PMX_FLAG_FORCE_SPARSE needs 3 <= max_shift <= 65535 and read_len <= 1024");
        return PMX_ERR_INVALID;
    }
    if (!sparse_ok || (flags & PMX_FLAG_FORCE_DENSE)) {
        for (uint32_t i = 0; i < njobs; i++) {
            int rc = cc_dense_one(ctx, d_F[i], d_R[i], has_m ? d_M[i] : nullptr, nbits[i], max_shift, read_len, do_ncc,
                                  do_mlen, d_out[i]);
            if (rc) return rc;
        }
        return PMX_OK;
    }
    // no hint from the caller: take one from a sample of the vectors (one small launch, one synchronisation)
    bool probed = false;
    static const bool probe_enabled = [] {   // PMX_DENSITY_PROBE=0: never (A/B, tests of the unhinted event path)
        const char *e = getenv("PMX_DENSITY_PROBE");
        return !(e && e[0] == '0');
    }();
    // (PMX_FLAG_FORCE_SPARSE: the set-bit path exactly as it is without a hint -- the event kernel finds dense tiles itself)
    if (probe_enabled && !(flags & (PMX_FLAG_WINDOW_ONLY | PMX_FLAG_DEEP_LISTS | PMX_FLAG_EVENTS_HINT | PMX_FLAG_FORCE_SPARSE)) &&
        pmx_events_used(max_shift)) {
        uint32_t hint = 0;
        int rc = auto_density_hint(ctx, njobs, d_F, d_R, d_M, nbits, max_shift, has_m, &hint);
        if (rc) return rc;
        flags |= hint;
        probed = true;
    }
    const uint32_t chunk = pmx_cc_batch_jobs(max_shift);
    const size_t ac_words = pmx_autocorr_scratch_words(max_lag);
    std::vector<pmx_job> jobs(njobs < chunk ? njobs : chunk);
    for (uint32_t lo = 0; lo < njobs; lo += chunk) {
        const uint32_t n = njobs - lo < chunk ? njobs - lo : chunk;
        if (do_mlen) {
            int rc = pmx_ensure_scratch(ctx, (size_t)n * ac_words);
            if (rc) return rc;
        }
        for (uint32_t i = 0; i < n; i++) {
            jobs[i].d_F = d_F[lo + i];
            jobs[i].d_R = d_R[lo + i];
            jobs[i].d_M = has_m ? d_M[lo + i] : nullptr;
            jobs[i].nbits = nbits[lo + i];
            jobs[i].d_out = d_out[lo + i];
            jobs[i].d_out2 = has_m ? (uint64_t *)(ctx->d_scratch + (size_t)i * ac_words) : nullptr;
            jobs[i].tile_first = tile_first ? tile_first[lo + i] : 0;
            jobs[i].tile_count = tile_count ? tile_count[lo + i] : 0;
        }
        // fork: the mappable-length pass on the auxiliary stream, beside the set-bit kernel (they share no output word:
        // row MLEN and scalar [2] belong to the autocorrelation, everything else to the cross-correlation)
        // PMX_AUTOCORR_FORK=0 in the environment keeps everything on the caller's stream (per-kernel profiling)
        ctx->window_only = (flags & PMX_FLAG_WINDOW_ONLY) != 0;
        ctx->deep_lists = (flags & PMX_FLAG_DEEP_LISTS) != 0;
        static const bool fork_enabled = [] {
            const char *e = getenv("PMX_AUTOCORR_FORK");
            return !(e && e[0] == '0');
        }();
        // beside the shift-chunked instantiation (max_shift > 1023) the fork does not pay (config 5 on one GPU: 17.4 ms
        // forked, 16.5 ms in sequence): the pair pass then holds 40 KB of histograms per workgroup
        static const bool fork_big = [] {   // PMX_AUTOCORR_FORK_BIG=1: also beside the event kernel of max_shift > 1023 (A/B)
            const char *e = getenv("PMX_AUTOCORR_FORK_BIG");
            return e && e[0] == '1';
        }();
        // (not behind the density probe either: its synchronisation has drained the stream, so the forked chain and the window
        // kernel would START together -- the window kernel's workgroups own fixed tile ranges, and the ones that find their CU
        // taken for the first 0.1 ms double up elsewhere for the whole launch: 6.6 -> 10.2 ms at 5 % reads per strand, measured)
        const bool fork = fork_enabled && !probed && (max_shift <= 1023 || (pmx_events_take_big(max_shift) && !ctx->window_only &&
                                                                 (fork_big || pmx_events_big_subgroups(max_shift, has_m ? 1 : 0) == 1)));
        int rc;
        if (do_mlen && !ctx->window_only && pmx_events_can_fuse_mlen(max_shift, max_lag)) {
            // the event kernel stages M and lists its run edges anyway: it takes the edge pairs of the mappable-length pass
            // too, and only the window kernel for the tiles it flagged + the recurrence remain of that pass
            pmx_fused_mlen fm;
            rc = pmx_launch_cc_sparse_batch(ctx, jobs.data(), n, max_shift, read_len, do_ncc, stride, false, max_lag, &fm);
            if (rc) return rc;
            if (!fm.done) {   // (not expected: the launcher declined the fusion)
                rc = pmx_launch_autocorr_edges_batch(ctx, jobs.data(), n, max_lag, 1, read_len, max_shift, stride);
                if (rc) return rc;
            }
            continue;
        }
        bool forked = false;
        rc = PMX_OK;
        // Behind the probe, window kernels alone: the chain is forked LATE -- queued on the auxiliary stream behind the window
        // kernel's launch (but not behind its completion: the mark is recorded in front of it).  The window kernel starts
        // alone and owns the CUs; the chain's kernels move in as its workgroups drain, under its tail.
        const bool late_fork = fork_enabled && probed && do_mlen && ctx->window_only && max_shift <= 1023;
        if (late_fork) {
            PMX_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        } else if (do_mlen && !fork) {
            rc = pmx_launch_autocorr_edges_batch(ctx, jobs.data(), n, max_lag, 1, read_len, max_shift, stride);
            if (rc) return rc;
        } else if (do_mlen) {
            PMX_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            PMX_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->aux_stream;       // the launchers enqueue on ctx->stream (a context is single-threaded)
            rc = pmx_launch_autocorr_edges_batch(ctx, jobs.data(), n, max_lag, 1, read_len, max_shift, stride);
            ctx->stream = main_stream;
            forked = true;
        }
        // Whatever was queued on the auxiliary stream (possibly a partial chain after an error) writes the context's
        // scratch: the caller's stream waits for it on EVERY exit from here on, so that no later call can race with it.
        auto join = [&]() -> hipError_t {
            hipError_t er = hipEventRecord(ctx->ev_join, ctx->aux_stream);
            if (er == hipSuccess) er = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
            if (er != hipSuccess) (void)hipStreamSynchronize(ctx->aux_stream);
            return er;
        };
        if (rc) {
            if (forked) (void)join();
            return rc;
        }
        rc = pmx_launch_cc_sparse_batch(ctx, jobs.data(), n, max_shift, read_len, do_ncc, stride, !do_mlen);
        if (late_fork && !rc) {
            PMX_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->aux_stream;
            rc = pmx_launch_autocorr_edges_batch(ctx, jobs.data(), n, max_lag, 1, read_len, max_shift, stride);
            ctx->stream = main_stream;
            forked = true;
        }
        if (forked) {
            const hipError_t er = join();
            if (!rc && er != hipSuccess) {
                pmx_set_error("join of the mappable-length pass failed: %s", hipGetErrorString(er));
                return PMX_ERR_HIP;
            }
        }
        if (rc) return rc;
    }
    return PMX_OK;
}

int pmx_cc_dev(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M, uint64_t nbits,
               uint32_t max_shift, uint32_t read_len, uint32_t flags, uint64_t *d_out)
{
    REQUIRE(ctx && d_F && d_R && d_out, "pmx_cc_dev: NULL argument");
    PMX_JOIN_SIDE(ctx);
    return pmx_cc_batch_dev(ctx, 1, &d_F, &d_R, d_M ? &d_M : nullptr, &nbits, max_shift, read_len, flags, &d_out);
}

int pmx_calc_correlation(pmx_ctx *ctx, const uint64_t *h_F, const uint64_t *h_R, const uint64_t *h_M,
                         uint64_t nbits, uint32_t max_shift, uint32_t read_len, uint32_t flags, uint64_t *h_out)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && h_F && h_R && h_out, "pmx_calc_correlation: NULL argument");
    PMX_JOIN_SIDE(ctx);
    int rc = check_shift_args(nbits, max_shift, "pmx_calc_correlation");
    if (rc) return rc;
    const uint64_t nw = words_for(nbits);
    const uint64_t *src[3] = {h_F, h_R, h_M};
    for (int i = 0; i < 3; i++) {
        if (!src[i]) continue;
        rc = ensure_stage(ctx, i, nw);
        if (rc) return rc;
        PMX_HIP(hipMemcpyAsync(ctx->d_stage[i], src[i], nw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t out_words = (size_t)PMX_NROWS * (max_shift + 1);
    rc = ensure_out_stage(ctx, out_words);
    if (rc) return rc;
    rc = pmx_cc_dev(ctx, ctx->d_stage[0], ctx->d_stage[1], h_M ? ctx->d_stage[2] : nullptr, nbits, max_shift,
                    read_len, flags, (uint64_t *)ctx->d_out_stage);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(h_out, ctx->d_out_stage, out_words * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_mappable_len(pmx_ctx *ctx, const uint64_t *h_M, uint64_t nbits, uint32_t max_shift, uint32_t flags,
                     uint64_t *h_out)
{
    if (ctx) (void)hipSetDevice(ctx->device);   // the caller may have switched devices (one context per GPU)
    REQUIRE(ctx && h_M && h_out, "pmx_mappable_len: NULL argument");
    PMX_JOIN_SIDE(ctx);
    int rc = check_shift_args(nbits, max_shift, "pmx_mappable_len");
    if (rc) return rc;
    const uint64_t nw = words_for(nbits);
    rc = ensure_stage(ctx, 2, nw);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(ctx->d_stage[2], h_M, nw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = ensure_out_stage(ctx, (size_t)max_shift + 1);
    if (rc) return rc;
    rc = pmx_mappable_len_dev(ctx, ctx->d_stage[2], nbits, max_shift, flags, (uint64_t *)ctx->d_out_stage);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(h_out, ctx->d_out_stage, ((size_t)max_shift + 1) * sizeof(u64), hipMemcpyDeviceToHost,
                           ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

}   // extern "C"
