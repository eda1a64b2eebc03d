// Device bit-vector builders: the MI355X counterparts of bit_array_set_bit / bit_array_set_region /
// bit_array_num_bits_set (PyMaSC/core/bitarray/bitarray.pxd:26-31, bitarray.pyx:72-107).
#include "pmx_common.h"

// bitarray[pos] = 1 (mscc.pyx:393, :416-417). One lane per read; 64-bit atomic OR because many
// reads share a word.  Out-of-range positions are dropped; with `bad` != null the smallest offending index is
// recorded there (the host entry point reads it back instead of scanning the positions on the CPU first: for the
// 31 M reads of the benchmark genome that scan was half of the end-to-end time).
__global__ void __launch_bounds__(256) k_set_positions(u64 *__restrict__ words, uint64_t nbits,
                                                       const int64_t *__restrict__ pos, uint64_t n,
                                                       u64 *__restrict__ bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t p = pos[i];
        if (p >= 0 && (uint64_t)p < nbits)
            atomicOr(&words[p >> 6], 1ull << (p & 63));
        else if (bad)
            atomicMin(bad, (u64)i);
    }
}

// bitarray.set(from, to), inclusive (bitarray.pyx:88-95). One wavefront per interval: the two edge
// words are OR-ed atomically, interior words are stored whole, lanes striding over the interval.
__global__ void __launch_bounds__(256) k_set_regions(u64 *__restrict__ words, uint64_t nbits,
                                                     const int64_t *__restrict__ from,
                                                     const int64_t *__restrict__ to, uint64_t n)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n; i += nwaves) {
        int64_t a = from[i], b = to[i];
        if (a < 0) a = 0;
        if (b >= (int64_t)nbits) b = (int64_t)nbits - 1;
        if (b < a) continue;
        const uint64_t wa = (uint64_t)a >> 6, wb = (uint64_t)b >> 6;
        const u64 lo_mask = ~0ull << (a & 63);
        const u64 hi_mask = ~0ull >> (63 - (b & 63));
        if (wa == wb) {
            if (lane == 0) atomicOr(&words[wa], lo_mask & hi_mask);
            continue;
        }
        if (lane == 0) atomicOr(&words[wa], lo_mask);
        if (lane == 1) atomicOr(&words[wb], hi_mask);
        for (uint64_t w = wa + 1 + lane; w < wb; w += 64) words[w] = ~0ull;
    }
}

// bit_array_num_bits_set: grid-stride popcount, wave reduce by DPP-free shuffles, one atomic per wave.
__global__ void __launch_bounds__(256) k_count(const u64 *__restrict__ words, uint64_t nwords,
                                               uint64_t nbits, u64 *__restrict__ out)
{
    u64 acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords;
         i += (uint64_t)gridDim.x * blockDim.x) {
        u64 w = words[i];
        if (i == nwords - 1 && (nbits & 63)) w &= ~0ull >> (64 - (nbits & 63));
        acc += (u64)__popcll(w);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// ---- density probe (round 4): what pmx_cc_batch_dev does for a caller that gives no hint ----
// The event kernel's lists hold a bounded number of reads and run edges per 64-Kbit tile; beyond that the window kernels
// are the right choice from the start (PMX_FLAG_WINDOW_ONLY), and in between the DEEP instantiation (PMX_FLAG_DEEP_LISTS).
// A caller that knows its counts says so; for one that does not, PMX_PROBE_SAMPLES evenly spread tiles of each of the
// largest chromosomes are counted here -- forward reads, reverse reads, run edges -- and the host applies the rule the
// calculator applies to the counts it holds (pymasc_amd/calculator.py: window_only_hint / deep_lists_hint).  One small
// launch + one 2-KB copy + one synchronisation per call, instead of an event pass that finds out tile by tile and a
// window pass behind it (round 3: 5-15 % above 1.8 % reads per strand).
__global__ void __launch_bounds__(256) k_density_probe(const pmx_probe_jobs jobs, u32 *__restrict__ out)
{
    const u32 job = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const uint64_t nbits = jobs.nbits[job];
    const uint64_t ntiles = (nbits + 65535) / 65536;
    if (ntiles == 0) return;
    const uint64_t nsamp = ntiles < PMX_PROBE_SAMPLES ? ntiles : PMX_PROBE_SAMPLES;
    if (s >= nsamp) return;
    const uint64_t tile = (2 * (uint64_t)s + 1) * ntiles / (2 * nsamp);   // the middle of the s-th of nsamp stretches
    const uint64_t nw = (nbits + 31) / 32;                                  // dwords of a vector
    const u32 *F = jobs.F[job], *R = jobs.R[job], *M = jobs.M[job];
    u32 cf = 0, cr = 0, ce = 0;
    const uint64_t j0 = tile * 2048 + 8 * (uint64_t)tid;
    u32 prev = (M && j0 > 0 && j0 - 1 < nw) ? M[j0 - 1] : 0u;
    for (u32 k = 0; k < 8; k++) {
        const uint64_t j = j0 + k;
        if (j >= nw) break;
        u32 f = F[j], r = R[j];
        if (j == nw - 1 && (nbits & 31)) {          // bits beyond the vector's length do not count
            const u32 keep = (1u << (nbits & 31)) - 1u;
            f &= keep;
            r &= keep;
        }
        cf += __popc(f);
        cr += __popc(r);
        if (M) {
            const u32 m = M[j];
            ce += __popc(m ^ ((m << 1) | (prev >> 31)));
            prev = m;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cf += __shfl_down(cf, off, 64);
        cr += __shfl_down(cr, off, 64);
        ce += __shfl_down(ce, off, 64);
    }
    if ((tid & 63) == 0) {
        if (cf) atomicAdd(&out[4 * job + 0], cf);
        if (cr) atomicAdd(&out[4 * job + 1], cr);
        if (ce) atomicAdd(&out[4 * job + 2], ce);
    }
    if (tid == 0) {
        const uint64_t b0 = tile * 65536, left = nbits - b0;
        atomicAdd(&out[4 * job + 3], (u32)(left < 65536 ? left : 65536));   // positions sampled
    }
}

int pmx_launch_density_probe(pmx_ctx *ctx, const pmx_probe_jobs *jobs, uint32_t njobs, uint32_t *d_out)
{
    if (njobs == 0) return PMX_OK;
    PMX_HIP(hipMemsetAsync(d_out, 0, (size_t)njobs * 4 * sizeof(u32), ctx->stream));
    hipLaunchKernelGGL(k_density_probe, dim3(njobs, PMX_PROBE_SAMPLES), dim3(256), 0, ctx->stream, *jobs, d_out);
    PMX_CHECK_LAUNCH("k_density_probe");
    return PMX_OK;
}

static int grid_for(pmx_ctx *ctx, uint64_t items, int per_block)
{
    uint64_t blocks = (items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int pmx_launch_set_positions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_pos, uint64_t n,
                             u64 *d_bad)
{
    if (n == 0) return PMX_OK;
    hipLaunchKernelGGL(k_set_positions, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream,
                       (u64 *)d_words, nbits, d_pos, n, d_bad);
    PMX_CHECK_LAUNCH("k_set_positions");
    return PMX_OK;
}

int pmx_launch_set_regions(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const int64_t *d_from,
                           const int64_t *d_to, uint64_t n)
{
    if (n == 0) return PMX_OK;
    hipLaunchKernelGGL(k_set_regions, dim3(grid_for(ctx, n, 4)), dim3(256), 0, ctx->stream,
                       (u64 *)d_words, nbits, d_from, d_to, n);
    PMX_CHECK_LAUNCH("k_set_regions");
    return PMX_OK;
}

int pmx_launch_count(pmx_ctx *ctx, const uint64_t *d_words, uint64_t nbits, u64 *d_count)
{
    const uint64_t nwords = (nbits + 63) / 64;
    if (nwords == 0) return PMX_OK;
    hipLaunchKernelGGL(k_count, dim3(grid_for(ctx, nwords, 256 * 8)), dim3(256), 0, ctx->stream,
                       (const u64 *)d_words, nwords, nbits, d_count);
    PMX_CHECK_LAUNCH("k_count");
    return PMX_OK;
}
