// Stream-ordered feeding (gfx950): the bulk counterparts of CCBitArrayCalculator.feed_forward_read / feed_reverse_read
// (PyMaSC/core/bitarray/mscc.pyx:370-418) and of _load_mappability's set(begin + 1, end) loop (mscc.pyx:327-349).
//
// The reference takes one Python call per read: sortedness check (mscc.pyx:362-366), forward duplicate rule ("same
// position as the previous forward read", :388-392), reverse rule ("the bit is already set", :416-418), read-length sums,
// bit set.  Here a RUN of reads of one chromosome arrives as arrays in file order (position, read length, strand) and
// three small kernels apply the same rules on the device, so that the host never walks the reads:
//   k_feed_maxlen   longest reverse read of the chunk (bounds the look-back of the reverse rule)
//   k_feed_reads    per read: sortedness against its predecessor, range check, duplicate rules, the two read-length
//                   sums; sets the FORWARD bits.  Reads R as earlier chunks left it and never writes it, so "already set"
//                   means exactly what it means in the reference: set by an earlier read in file order -- an earlier chunk
//                   (the vector) or an earlier read of this chunk (look-back over the reads whose position allows the
//                   same bit: pos_j >= bit - maxlen + 1).
//   k_feed_finish   sets the REVERSE bits of the chunk and hands the chunk's last position / last forward position on to
//                   the next chunk (the reference's _last_pos / _last_forward_pos).
// Nothing is returned to the host here: sums, counts and the first offending read (unsorted / out of range) stay in the
// chromosome's feed state in device memory (include/pymasc_amd.h: PMX_FEED_*) until the caller downloads it.
#include "pmx_common.h"

#define FEED_ERR_BASE (1ull << 62)   // error words hold FEED_ERR_BASE - index (atomicMax keeps the FIRST index); 0 = none

// position i; with the strand PACKED into the top bit of the position word (rev == nullptr: 4 instead of 5 bytes per read over
// PCIe) that bit is masked off
template <typename T>
__device__ __forceinline__ int64_t feed_ld(const T *p, uint64_t i, bool packed)
{
    const T v = p[i];
    constexpr T top = (T)((T)1 << (8 * sizeof(T) - 1));
    return (int64_t)(packed ? (T)(v & (T)~top) : v);
}
template <typename T>
__device__ __forceinline__ bool feed_rev(const T *p, const unsigned char *rev, uint64_t i)
{
    return rev ? rev[i] != 0 : p[i] < 0;   // (packed: the sign bit)
}
// read lengths: an array, or (null pointer) ONE length for every read of the run
template <typename T>
__device__ __forceinline__ int64_t feed_len(const T *p, int64_t uniform, uint64_t i) { return p ? (int64_t)p[i] : uniform; }


// ---- the two look-backs of the duplicate rules, bounded (round 4) ----
// "Was this reverse read's bit set by an earlier read of the run" (mscc.pyx:416-418) walks back over the reads whose position
// allows the same bit; "is there an earlier forward read at this position" (mscc.pyx:388-392) over the run of equal
// positions.  Ordinarily that is a step or two.  In a pile-up -- 10^5 reads inside one read length: chrM, rDNA, satellites --
// it was a chain of K dependent loads per read by ONE lane, K^2 in all (round-3 advisor finding).  Now a lane walks
// FEED_OWN_STEPS steps itself; the lanes that are still undecided are then served one after the other by the whole
// wavefront, 64 earlier reads per step (coalesced loads, one ballot): K^2 / 64 wave-steps.  Exact in both phases.
// ALL 64 lanes of a wavefront call these together (`active`: the lane has a read to decide).
#define FEED_OWN_STEPS 16
__device__ __forceinline__ uint64_t feed_lane64(uint64_t v, int src)
{
    return ((uint64_t)(u32)__shfl((int)(v >> 32), src, 64) << 32) | (u32)__shfl((int)(u32)v, src, 64);
}

// reverse read i with bit `bit`: true iff an earlier read j < i of the run is a reverse read with the same bit
template <typename PT, typename LT>
__device__ __forceinline__ bool feed_reverse_seen(const PT *__restrict__ pos, const LT *__restrict__ rlen, int64_t ulen,
                                                  const unsigned char *__restrict__ rev, bool packed, uint64_t i, int64_t bit,
                                                  int64_t maxlen, bool active, u32 lane)
{
    const int64_t lowest = bit - maxlen + 1;
    bool set = false, done = !active;
    uint64_t j = i;
    for (int step = 0; step < FEED_OWN_STEPS && !done; step++) {
        if (j == 0) {
            done = true;
            break;
        }
        const int64_t pj = feed_ld(pos, j - 1, packed);
        if (pj < lowest) {
            done = true;
            break;
        }
        set = feed_rev(pos, rev, j - 1) && pj + feed_len(rlen, ulen, j - 1) - 1 == bit;
        done = set;
        j--;
    }
    uint64_t pending = __ballot(!done);
    while (pending) {
        const int src = (int)__builtin_ctzll(pending);
        pending &= pending - 1;
        uint64_t base = feed_lane64(j, src);                         // the reads base - 1, base - 2, ... are still to be looked at
        const int64_t b = (int64_t)feed_lane64((uint64_t)bit, src), low = (int64_t)feed_lane64((uint64_t)lowest, src);
        bool found = false;
        for (;;) {
            bool hit = false, below = false;
            if (base > lane) {
                const uint64_t k = base - 1 - lane;
                const int64_t pk = feed_ld(pos, k, packed);
                below = pk < low;
                hit = !below && feed_rev(pos, rev, k) && pk + feed_len(rlen, ulen, k) - 1 == b;
            }
            const uint64_t hits = __ballot(hit), bel = __ballot(below);
            // (sorted run: the reads below the window are the higher lanes from some lane on; a hit counts in front of them only)
            const uint64_t in_front = bel ? ((1ull << __builtin_ctzll(bel)) - 1ull) : ~0ull;
            found = (hits & in_front) != 0;
            if (found || bel || base <= 64) break;
            base -= 64;
        }
        if ((int)lane == src) set = found;
    }
    return set;
}

// forward read i at position p: true iff an earlier read of the run at the same position is a forward read
template <typename PT>
__device__ __forceinline__ bool feed_forward_seen(const PT *__restrict__ pos, const unsigned char *__restrict__ rev, bool packed,
                                                  uint64_t i, int64_t p, bool active, u32 lane)
{
    bool dup = false, done = !active;
    uint64_t j = i;
    for (int step = 0; step < FEED_OWN_STEPS && !done; step++) {
        if (j == 0 || feed_ld(pos, j - 1, packed) != p) {
            done = true;
            break;
        }
        dup = !feed_rev(pos, rev, j - 1);
        done = dup;
        j--;
    }
    uint64_t pending = __ballot(!done);
    while (pending) {
        const int src = (int)__builtin_ctzll(pending);
        pending &= pending - 1;
        uint64_t base = feed_lane64(j, src);
        const int64_t q = (int64_t)feed_lane64((uint64_t)p, src);
        bool found = false;
        for (;;) {
            bool hit = false, other = false;
            if (base > lane) {
                const uint64_t k = base - 1 - lane;
                other = feed_ld(pos, k, packed) != q;              // the run of equal positions has ended
                hit = !other && !feed_rev(pos, rev, k);
            }
            const uint64_t hits = __ballot(hit), oth = __ballot(other);
            const uint64_t in_front = oth ? ((1ull << __builtin_ctzll(oth)) - 1ull) : ~0ull;
            found = (hits & in_front) != 0;
            if (found || oth || base <= 64) break;
            base -= 64;
        }
        if ((int)lane == src) dup = found;
    }
    return dup;
}

template <typename LT>
__global__ void __launch_bounds__(256) k_feed_maxlen(const LT *__restrict__ rlen, int64_t ulen, const unsigned char *__restrict__ rev,
                                                     uint64_t n, u64 *__restrict__ state)
{
    u64 m = 0;
    if (!rlen) {                       // one length for every read (the strand does not matter for the bound)
        m = ulen > 0 ? (u64)ulen : 0;
    } else {
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
            const int64_t l = feed_len(rlen, ulen, i);
            if ((!rev || rev[i]) && l > 0 && (u64)l > m) m = (u64)l;   // (packed strand: the longest read of either strand)
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    // one atomic per BLOCK: thousands of waves adding to one address serialise in L2 (96 us for 8 K waves)
    __shared__ u64 part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) m = part[w] > m ? part[w] : m;
        if (m > state[PMX_FEED_MAX_REVERSE_LEN]) atomicMax(&state[PMX_FEED_MAX_REVERSE_LEN], m);
    }
}

template <typename PT, typename LT>
__global__ void __launch_bounds__(256) k_feed_reads(const PT *__restrict__ pos, const LT *__restrict__ rlen, int64_t ulen,
                                                    const unsigned char *__restrict__ rev, uint64_t n, uint64_t base, uint64_t nbits,
                                                    u64 *__restrict__ F, const u64 *__restrict__ R, u64 *__restrict__ state)
{
    const int64_t prev_last = base ? (int64_t)state[PMX_FEED_LAST_POS] : 0;            // _last_pos (0 at a chromosome's start)
    const int64_t prev_fwd = (int64_t)state[PMX_FEED_LAST_FORWARD_POS];                 // _last_forward_pos (0 likewise)
    // (one length for every read of the run: no k_feed_maxlen launch, the bound is that length or an earlier run's;
    // k_feed_finish records it for the runs that follow)
    int64_t maxlen = (int64_t)state[PMX_FEED_MAX_REVERSE_LEN];
    if (!rlen && ulen > maxlen) maxlen = ulen;
    const bool packed = rev == nullptr;
    u64 fsum = 0, rsum = 0, nf = 0, nr = 0, maxf = 0, e_sort = 0, e_range = 0;
    bool any_f = false;
    const u32 lane = threadIdx.x & 63u;
    // (wave-uniform trips: the look-backs are wavefront-wide, see feed_reverse_seen)
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + lane;
        const bool valid = i < n;
        const int64_t p = valid ? feed_ld(pos, i, packed) : 0, l = valid ? feed_len(rlen, ulen, i) : 0;
        const bool rv = valid && feed_rev(pos, rev, i);
        if (valid) {
            const int64_t before = i ? feed_ld(pos, i - 1, packed) : prev_last;
            if (p < before) {                                                           // mscc.pyx:362-363
                const u64 code = FEED_ERR_BASE - (base + i);
                e_sort = code > e_sort ? code : e_sort;
            }
        }
        const int64_t bit = rv ? p + l - 1 : p;
        const bool in_range = valid && bit >= 0 && (uint64_t)bit < nbits;
        if (valid && !in_range) {
            const u64 code = FEED_ERR_BASE - (base + i);
            e_range = code > e_range ? code : e_range;
        }
        const bool is_f = in_range && !rv, is_r = in_range && rv;
        // duplicate: an earlier forward read at this position (sorted input: such reads are the ones right before it)
        const bool fdup = feed_forward_seen(pos, rev, packed, i, p, is_f && p != prev_fwd, lane) || (is_f && p == prev_fwd);
        // counts iff its bit is clear: set by an earlier chunk (the vector) or by an earlier read of this chunk
        const bool r_old = is_r && ((R[bit >> 6] >> (bit & 63)) & 1ull);
        const bool rdup = feed_reverse_seen(pos, rlen, ulen, rev, packed, i, bit, maxlen, is_r && !r_old, lane) || r_old;
        if (is_f) {
            any_f = true;
            if ((u64)p > maxf) maxf = (u64)p;
            if (!fdup) {
                fsum += (u64)l;
                nf++;
                atomicOr(&F[bit >> 6], 1ull << (bit & 63));
            }
        } else if (is_r && !rdup) {
            rsum += (u64)l;
            nr++;
        }
    }
    // wave sums / maxima, one atomic per wave and word
    for (int off = 32; off > 0; off >>= 1) {
        fsum += __shfl_down(fsum, off, 64);
        rsum += __shfl_down(rsum, off, 64);
        nf += __shfl_down(nf, off, 64);
        nr += __shfl_down(nr, off, 64);
        const u64 a = __shfl_down(maxf, off, 64), b = __shfl_down(e_sort, off, 64), c = __shfl_down(e_range, off, 64);
        maxf = a > maxf ? a : maxf;
        e_sort = b > e_sort ? b : e_sort;
        e_range = c > e_range ? c : e_range;
    }
    // block sums / maxima through LDS, then one atomic per block and word (per-wave atomics on a handful of addresses took
    // 0.3 ms per chromosome: they serialise in L2)
    __shared__ u64 part[4][8];
    const u32 wv = threadIdx.x >> 6;
    const bool wave_any_f = __ballot(any_f) != 0;
    if ((threadIdx.x & 63) == 0) {
        part[wv][0] = fsum;
        part[wv][1] = rsum;
        part[wv][2] = nf;
        part[wv][3] = nr;
        part[wv][4] = wave_any_f ? maxf + 1 : 0;     // (+1: 0 = no forward read in the chunk)
        part[wv][5] = e_sort;
        part[wv][6] = e_range;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const u32 k = threadIdx.x;
        u64 v = part[0][k];
        for (int w = 1; w < 4; w++) v = k < 4 ? v + part[w][k] : (part[w][k] > v ? part[w][k] : v);
        if (v) {
            if (k == 0) atomicAdd(&state[PMX_FEED_FORWARD_LEN_SUM], v);
            else if (k == 1) atomicAdd(&state[PMX_FEED_REVERSE_LEN_SUM], v);
            else if (k == 2) atomicAdd(&state[PMX_FEED_FORWARD_KEPT], v);
            else if (k == 3) atomicAdd(&state[PMX_FEED_REVERSE_KEPT], v);
            else if (k == 4) atomicMax(&state[PMX_FEED_CHUNK_FORWARD_POS], v);
            else if (k == 5) atomicMax(&state[PMX_FEED_FIRST_UNSORTED], v);
            else atomicMax(&state[PMX_FEED_FIRST_OUT_OF_RANGE], v);
        }
    }
}

template <typename PT, typename LT>
__global__ void __launch_bounds__(256) k_feed_finish(const PT *__restrict__ pos, const LT *__restrict__ rlen, int64_t ulen,
                                                     const unsigned char *__restrict__ rev, uint64_t n, uint64_t nbits,
                                                     u64 *__restrict__ R, u64 *__restrict__ state)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!feed_rev(pos, rev, i)) continue;
        const int64_t bit = feed_ld(pos, i, rev == nullptr) + feed_len(rlen, ulen, i) - 1;
        if (bit >= 0 && (uint64_t)bit < nbits) atomicOr(&R[bit >> 6], 1ull << (bit & 63));   // mscc.pyx:416-417
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t last = feed_ld(pos, n - 1, rev == nullptr);
        state[PMX_FEED_LAST_POS] = last < 0 ? 0 : (u64)last;
        const u64 cf = state[PMX_FEED_CHUNK_FORWARD_POS];
        if (cf) state[PMX_FEED_LAST_FORWARD_POS] = cf - 1;
        state[PMX_FEED_CHUNK_FORWARD_POS] = 0;
        state[PMX_FEED_READS] += n;
        if (!rlen && ulen > 0 && (u64)ulen > state[PMX_FEED_MAX_REVERSE_LEN]) state[PMX_FEED_MAX_REVERSE_LEN] = (u64)ulen;
    }
}

// bitarray.set(first + offset, last), inclusive (bitarray.pyx:88-95; the mappability loader calls set(begin + 1, end),
// mscc.pyx:343-344: offset = 1 for BigWig (begin, end) pairs).  One wavefront per interval as in k_set_regions; an
// interval outside [0, nbits) is clipped and its index recorded in err (FEED_ERR_BASE - index, atomicMax).
template <typename T>
__global__ void __launch_bounds__(256) k_set_regions_t(u64 *__restrict__ words, uint64_t nbits, const T *__restrict__ from,
                                                       const T *__restrict__ to, uint64_t n, int64_t offset, u64 *__restrict__ err)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n; i += nwaves) {
        int64_t a = (int64_t)from[i] + offset, b = (int64_t)to[i];
        if (b < a) continue;
        if (a < 0 || b >= (int64_t)nbits) {
            if (err && lane == 0) atomicMax(err, FEED_ERR_BASE - i);
            if (a < 0) a = 0;
            if (b >= (int64_t)nbits) b = (int64_t)nbits - 1;
            if (b < a) continue;
        }
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

// ---- a whole vector from SORTED, DISJOINT intervals, owner-computes (round 4; pmx_bits_set_regions_ex, PMX_REGIONS_SORTED) ----
// BigWig order: begin_i + offset <= end_i < begin_(i+1) + offset.  k_set_regions_t gives an interval to a wavefront: two atomic
// ORs and a run of scattered 8-byte stores per interval into a vector that had to be cleared first (22 us + 7 us per
// chromosome of a genome's feed, which the GPU's work paces).  Here the vector's words are dealt to the workgroups, RB_WORDS
// each; a workgroup finds the intervals that reach into its bits with two searches of the sorted ends, marks in LDS where a
// run of ones starts and where it stops (XOR: two marks on one bit -- a run that ends where the next begins -- cancel), turns
// the marks into the bits with a prefix XOR (within a word by shifts; across its 1024 words by ballots) and writes every word
// of its range once, plainly, zeros included: no clear, no atomics on the vector.  The order is CHECKED (every interval once,
// by index): a violation is recorded in err_order (FEED_ERR_BASE - index; the vector's content is then undefined, and the
// caller must use the general setter); out-of-range intervals are clipped and recorded as k_set_regions_t records them.
#define RB_WORDS 1024u
template <typename PT>
__device__ __forceinline__ uint64_t feed_lower_bound(const PT *__restrict__ pos, uint64_t n, int64_t target, bool packed, u32 lane);   // (below)
template <typename T>
__global__ void __launch_bounds__(256) k_regions_build(u64 *__restrict__ words, uint64_t nbits, const T *__restrict__ from,
                                                       const T *__restrict__ to, uint64_t n, int64_t offset, u64 *__restrict__ err_range,
                                                       u64 *__restrict__ err_order)
{
    __shared__ __align__(16) u64 marks[RB_WORDS];
    __shared__ uint64_t s_idx[2];
    __shared__ u32 s_par[4];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t nwords = (nbits + 63) / 64;
    const uint64_t w0 = (uint64_t)blockIdx.x * RB_WORDS;
    const uint64_t w1 = w0 + RB_WORDS < nwords ? w0 + RB_WORDS : nwords;
    const int64_t lo = (int64_t)(w0 * 64), hi = (int64_t)(w1 * 64);
    for (u32 i = tid; i < RB_WORDS; i += 256) marks[i] = 0;
    // the intervals that reach into [lo, hi): from the first whose end is >= lo up to the first whose begin is >= hi
    if (wv < 2) {
        const uint64_t idx = wv == 0 ? feed_lower_bound(to, n, lo, false, lane) : feed_lower_bound(from, n, hi - offset, false, lane);
        if (lane == 0) s_idx[wv] = idx;
    }
    __syncthreads();
    const uint64_t i_lo = s_idx[0], i_hi = s_idx[1];
    for (uint64_t i = i_lo + tid; i < i_hi; i += 256) {
        int64_t a = (int64_t)from[i] + offset, b = (int64_t)to[i];
        if (b < a) continue;                           // (an order violation, recorded below)
        if (a < 0) a = 0;
        if (b >= (int64_t)nbits) b = (int64_t)nbits - 1;
        const int64_t sa = a > lo ? a : lo, e1 = b + 1;   // ones on [sa, e1)
        if (sa >= hi || e1 <= sa) continue;
        atomicXor(&marks[(u32)((uint64_t)(sa - lo) >> 6)], 1ull << (sa & 63));
        if (e1 < hi) atomicXor(&marks[(u32)((uint64_t)(e1 - lo) >> 6)], 1ull << (e1 & 63));
    }
    // every interval once, by index: order against its successor, emptiness, range
    {
        const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
        const uint64_t ia = (uint64_t)blockIdx.x * per, ib = ia + per < n ? ia + per : n;
        u64 e_order = 0, e_range = 0;
        for (uint64_t i = ia + tid; i < ib; i += 256) {
            const int64_t a = (int64_t)from[i] + offset, b = (int64_t)to[i];
            const bool bad = b < a || (i + 1 < n && (int64_t)from[i + 1] + offset <= b);
            if (bad && FEED_ERR_BASE - i > e_order) e_order = FEED_ERR_BASE - i;
            if (b >= a && (a < 0 || b >= (int64_t)nbits) && FEED_ERR_BASE - i > e_range) e_range = FEED_ERR_BASE - i;
        }
        if (e_order && err_order) atomicMax(err_order, e_order);
        if (e_range && err_range) atomicMax(err_range, e_range);
    }
    __syncthreads();
    // marks -> bits: four consecutive words per thread
    u64 f[4];
    u32 par = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u64 m = marks[4 * tid + k];
        u64 x = m;
        x ^= x << 1;
        x ^= x << 2;
        x ^= x << 4;
        x ^= x << 8;
        x ^= x << 16;
        x ^= x << 32;           // bit j: parity of the marks on bits 0 .. j of the word
        f[k] = par ? ~x : x;    // ... and of the thread's words before it
        par ^= (u32)__popcll(m) & 1u;
    }
    const uint64_t bal = __ballot(par != 0);
    u32 carry = (u32)__popcll(bal & ((1ull << lane) - 1ull)) & 1u;   // the lanes before mine
    if (lane == 0) s_par[wv] = (u32)__popcll(bal) & 1u;
    __syncthreads();
    for (u32 k = 0; k < wv; k++) carry ^= s_par[k];                   // the waves before mine
    const uint64_t w = w0 + 4ull * tid;
    if (carry) {
#pragma unroll
        for (u32 k = 0; k < 4; k++) f[k] = ~f[k];
    }
    if (w + 3 < w1) {
        reinterpret_cast<ulonglong2 *>(words + w)[0] = make_ulonglong2(f[0], f[1]);
        reinterpret_cast<ulonglong2 *>(words + w)[1] = make_ulonglong2(f[2], f[3]);
    } else {
#pragma unroll
        for (u32 k = 0; k < 4; k++)
            if (w + k < w1) words[w + k] = f[k];
    }
}

// bitarray[pos] = 1 for positions of either width; out-of-range positions are dropped and recorded in err
template <typename T>
__global__ void __launch_bounds__(256) k_set_positions_t(u64 *__restrict__ words, uint64_t nbits, const T *__restrict__ pos,
                                                         uint64_t n, u64 *__restrict__ err)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t p = (int64_t)pos[i];
        if (p >= 0 && (uint64_t)p < nbits)
            atomicOr(&words[p >> 6], 1ull << (p & 63));
        else if (err)
            atomicMax(err, FEED_ERR_BASE - i);
    }
}

static int feed_grid(pmx_ctx *ctx, uint64_t items, int per_block, int per_cu = 8)
{
    uint64_t blocks = (items + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)ctx->num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- the FIRST run of a chromosome, owner-computes (round 4) ----
// k_feed_reads sets one bit per read with a 64-bit atomic OR: reads lie ~200 bp apart per strand, so every atomic is a
// memory-side operation on a line of its own -- 57 us for chromosome 1's 1.3 M reads, plus 15 us for the reverse bits
// (k_feed_finish) and 2 x 6 us to clear the two vectors first: the feed of a genome was paced by these kernels, not by the
// PCIe copy (rocprofv3, round 4).  When a run is the first of its chromosome -- the common case: a reader hands over a
// chromosome at a time -- the vectors hold nothing yet, and the work can be dealt by WORDS instead of by reads:
// a workgroup owns FB_WORDS words of F and of R, finds the reads whose bits fall into them (the run is sorted: two
// searches, 64 probes per round), applies the reference's rules to them exactly as k_feed_reads does, sets the bits in
// LDS and writes its words with plain 16-byte stores -- EVERY word of both vectors is written, so the caller does not
// clear them.  Sortedness / range errors are found by index slices (every read is looked at once whatever the order),
// per-workgroup sums go to `partial` and k_feed_build_finish adds them to the chromosome's state (thousands of
// workgroups adding to seven words would serialise).  An unsorted run leaves vectors of no use, as in the reference (it
// raises at the first such read, mscc.pyx:362-363); the error word is exact.
#define FB_WORDS 1024u
#define FB_PART 8u

// first index i in [0, n] with pos[i] >= target (sorted run); all 64 lanes of a wave call this together
template <typename PT>
__device__ __forceinline__ uint64_t feed_lower_bound(const PT *__restrict__ pos, uint64_t n, int64_t target, bool packed, u32 lane)
{
    uint64_t lo = 0, hi = n;   // the answer lies in [lo, hi]
    while (hi - lo > 64) {
        const uint64_t step = (hi - lo + 63) / 64;
        const uint64_t probe = lo + (uint64_t)lane * step;
        const bool below = probe < hi && feed_ld(pos, probe, packed) < target;
        const u32 cnt = (u32)__popcll(__ballot(below));   // (sorted: the probes below the target are the first cnt)
        if (cnt == 0) {
            hi = lo;
            break;
        }
        const uint64_t nhi = lo + (uint64_t)cnt * step;
        lo = lo + (uint64_t)(cnt - 1) * step + 1;
        hi = nhi < hi ? nhi : hi;
    }
    const uint64_t i = lo + lane;
    const bool below = i < hi && feed_ld(pos, i, packed) < target;
    return lo + (uint64_t)__popcll(__ballot(below));
}

template <typename PT, typename LT>
__global__ void __launch_bounds__(256) k_feed_build(const PT *__restrict__ pos, const LT *__restrict__ rlen, int64_t ulen,
                                                    const unsigned char *__restrict__ rev, uint64_t n, uint64_t base, uint64_t nbits,
                                                    u64 *__restrict__ F, u64 *__restrict__ R, const u64 *__restrict__ state,
                                                    u64 *__restrict__ partial)
{
    __shared__ __align__(16) u64 sF[FB_WORDS];
    __shared__ __align__(16) u64 sR[FB_WORDS];
    __shared__ uint64_t s_idx[2];
    __shared__ u64 part[4][FB_PART];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const bool packed = rev == nullptr;
    const uint64_t nwords = (nbits + 63) / 64;
    const uint64_t w0 = (uint64_t)blockIdx.x * FB_WORDS;
    const uint64_t w1 = w0 + FB_WORDS < nwords ? w0 + FB_WORDS : nwords;
    const int64_t lo = (int64_t)(w0 * 64), hi = (int64_t)(w1 * 64);
    const int64_t prev_last = base ? (int64_t)state[PMX_FEED_LAST_POS] : 0;
    const int64_t prev_fwd = (int64_t)state[PMX_FEED_LAST_FORWARD_POS];
    int64_t maxlen = (int64_t)state[PMX_FEED_MAX_REVERSE_LEN];
    if (!rlen && ulen > maxlen) maxlen = ulen;
    if (maxlen < 1) maxlen = 1;
    for (u32 i = tid; i < FB_WORDS; i += 256) {
        sF[i] = 0;
        sR[i] = 0;
    }
    // the reads whose bit can fall into [lo, hi): forward bit = pos, reverse bit = pos + len - 1, len <= maxlen
    if (wv < 2) {
        const uint64_t idx = feed_lower_bound(pos, n, wv == 0 ? lo - maxlen + 1 : hi, packed, lane);
        if (lane == 0) s_idx[wv] = idx;
    }
    __syncthreads();
    const uint64_t i_lo = s_idx[0], i_hi = s_idx[1];
    u64 fsum = 0, rsum = 0, nf = 0, nr = 0, maxf = 0, e_sort = 0, e_range = 0;
    bool any_f = false;
    for (uint64_t i0 = i_lo + (tid & ~63u); i0 < i_hi; i0 += 256) {   // (wave-uniform trips: the look-backs are wavefront-wide)
        const uint64_t i = i0 + lane;
        const bool valid = i < i_hi;
        const int64_t p = valid ? feed_ld(pos, i, packed) : 0, l = valid ? feed_len(rlen, ulen, i) : 0;
        const bool rv = valid && feed_rev(pos, rev, i);
        const int64_t bit = rv ? p + l - 1 : p;
        const bool mine = valid && bit >= lo && bit < hi && (uint64_t)bit < nbits;   // (else another workgroup's read, or out of range: dropped)
        const bool is_f = mine && !rv, is_r = mine && rv;
        // duplicate: an earlier forward read at this position (mscc.pyx:388-392; k_feed_reads)
        const bool fdup = feed_forward_seen(pos, rev, packed, i, p, is_f && p != prev_fwd, lane) || (is_f && p == prev_fwd);
        // counts iff its bit is clear: nothing was set before this run (the first of the chromosome), so only an earlier
        // read of the run can have set it (mscc.pyx:416-418)
        const bool rdup = feed_reverse_seen(pos, rlen, ulen, rev, packed, i, bit, maxlen, is_r, lane);
        const u64 mask = 1ull << (bit & 63);
        const u32 w = mine ? (u32)(((uint64_t)bit >> 6) - w0) : 0u;
        if (is_f && !fdup) {
            fsum += (u64)l;
            nf++;
            atomicOr(&sF[w], mask);
        }
        if (is_r) {
            if (!rdup) {
                rsum += (u64)l;
                nr++;
            }
            atomicOr(&sR[w], mask);
        }
    }
    // every read once, by index: order against its predecessor, range, the last forward position of the run
    {
        const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
        const uint64_t a = (uint64_t)blockIdx.x * per, b = a + per < n ? a + per : n;
        for (uint64_t i = a + tid; i < b; i += 256) {
            const int64_t p = feed_ld(pos, i, packed), l = feed_len(rlen, ulen, i);
            const bool rv = feed_rev(pos, rev, i);
            const int64_t before = i ? feed_ld(pos, i - 1, packed) : prev_last;
            if (p < before) {
                const u64 code = FEED_ERR_BASE - (base + i);
                e_sort = code > e_sort ? code : e_sort;
            }
            const int64_t bit = rv ? p + l - 1 : p;
            if (bit < 0 || (uint64_t)bit >= nbits) {
                const u64 code = FEED_ERR_BASE - (base + i);
                e_range = code > e_range ? code : e_range;
                continue;
            }
            if (!rv) {
                any_f = true;
                if ((u64)p > maxf) maxf = (u64)p;
            }
        }
    }
    __syncthreads();
    // my words, whole: the vectors need not have been cleared
    {
        const u32 cnt = (u32)(w1 - w0);
        uint4 *dF = reinterpret_cast<uint4 *>(F + w0), *dR = reinterpret_cast<uint4 *>(R + w0);
        const uint4 *qF = reinterpret_cast<const uint4 *>(sF), *qR = reinterpret_cast<const uint4 *>(sR);
        for (u32 i = tid; i < cnt / 2; i += 256) {
            dF[i] = qF[i];
            dR[i] = qR[i];
        }
        if ((cnt & 1u) && tid == 0) {
            F[w0 + cnt - 1] = sF[cnt - 1];
            R[w0 + cnt - 1] = sR[cnt - 1];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        fsum += __shfl_down(fsum, off, 64);
        rsum += __shfl_down(rsum, off, 64);
        nf += __shfl_down(nf, off, 64);
        nr += __shfl_down(nr, off, 64);
        const u64 a = __shfl_down(maxf, off, 64), b = __shfl_down(e_sort, off, 64), c = __shfl_down(e_range, off, 64);
        maxf = a > maxf ? a : maxf;
        e_sort = b > e_sort ? b : e_sort;
        e_range = c > e_range ? c : e_range;
    }
    const bool wave_any_f = __ballot(any_f) != 0;
    if (lane == 0) {
        part[wv][0] = fsum;
        part[wv][1] = rsum;
        part[wv][2] = nf;
        part[wv][3] = nr;
        part[wv][4] = wave_any_f ? maxf + 1 : 0;
        part[wv][5] = e_sort;
        part[wv][6] = e_range;
    }
    __syncthreads();
    if (tid < 7) {
        u64 v = part[0][tid];
        for (int w = 1; w < 4; w++) v = tid < 4 ? v + part[w][tid] : (part[w][tid] > v ? part[w][tid] : v);
        partial[(size_t)blockIdx.x * FB_PART + tid] = v;
    }
}

template <typename PT>
__global__ void __launch_bounds__(1024) k_feed_build_finish(const u64 *__restrict__ partial, u32 nblocks, const PT *__restrict__ pos,
                                                            bool packed, uint64_t n, int64_t ulen, bool uniform, u64 *__restrict__ state)
{
    __shared__ u64 acc[16][FB_PART];
    const u32 tid = threadIdx.x, k = tid & 7u, g = tid >> 3;   // 128 groups of eight: one word each (seven used)
    u64 v = 0;
    if (k < 7)
        for (u32 b = g; b < nblocks; b += 128) {
            const u64 x = partial[(size_t)b * FB_PART + k];
            v = k < 4 ? v + x : (x > v ? x : v);
        }
    // groups g, g + 8, ... share a lane residue: fold the 8 groups of a wave, then the 16 waves
    for (int off = 32; off >= 8; off >>= 1) {
        const u64 o = __shfl_down(v, off, 64);
        v = k < 4 ? v + o : (o > v ? o : v);
    }
    if ((tid & 63u) < 8) acc[tid >> 6][k] = v;
    __syncthreads();
    if (tid < 7) {
        u64 t = acc[0][tid];
        for (u32 w = 1; w < 16; w++) t = tid < 4 ? t + acc[w][tid] : (acc[w][tid] > t ? acc[w][tid] : t);
        if (tid == 0) state[PMX_FEED_FORWARD_LEN_SUM] += t;
        else if (tid == 1) state[PMX_FEED_REVERSE_LEN_SUM] += t;
        else if (tid == 2) state[PMX_FEED_FORWARD_KEPT] += t;
        else if (tid == 3) state[PMX_FEED_REVERSE_KEPT] += t;
        else if (tid == 4) {
            if (t) state[PMX_FEED_LAST_FORWARD_POS] = t - 1;
        } else if (tid == 5) {
            if (t > state[PMX_FEED_FIRST_UNSORTED]) state[PMX_FEED_FIRST_UNSORTED] = t;
        } else if (t > state[PMX_FEED_FIRST_OUT_OF_RANGE]) state[PMX_FEED_FIRST_OUT_OF_RANGE] = t;
    }
    if (tid == 1023) {
        const int64_t last = feed_ld(pos, n - 1, packed);
        state[PMX_FEED_LAST_POS] = last < 0 ? 0 : (u64)last;
        state[PMX_FEED_READS] += n;
        if (uniform && ulen > 0 && (u64)ulen > state[PMX_FEED_MAX_REVERSE_LEN]) state[PMX_FEED_MAX_REVERSE_LEN] = (u64)ulen;
    }
}

// workgroups of k_feed_build for a vector of nbits (and the words of `partial` it needs: FB_PART per workgroup)
uint32_t pmx_feed_build_blocks(uint64_t nbits) { return (uint32_t)(((nbits + 63) / 64 + FB_WORDS - 1) / FB_WORDS); }

template <typename PT, typename LT>
static int launch_feed(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *d_pos, const void *d_len,
                       int64_t ulen, const unsigned char *d_rev, uint64_t n, uint64_t base, u64 *d_state, u64 *d_partial)
{
    const int g = feed_grid(ctx, n, 1024, 4);   // (every block ends in a few atomics on the state words)
    if (d_len) {   // (a run of one read length carries its bound in `ulen`)
        hipLaunchKernelGGL(k_feed_maxlen<LT>, dim3(g), dim3(256), 0, ctx->stream, (const LT *)d_len, ulen, d_rev, n, d_state);
        PMX_CHECK_LAUNCH("k_feed_maxlen");
    }
    if (d_partial) {   // the first run of the chromosome: every word of F and R written by its owner (k_feed_build)
        const u32 nb = pmx_feed_build_blocks(nbits);
        hipLaunchKernelGGL((k_feed_build<PT, LT>), dim3(nb), dim3(256), 0, ctx->stream, (const PT *)d_pos, (const LT *)d_len, ulen, d_rev,
                           n, base, nbits, (u64 *)d_F, (u64 *)d_R, (const u64 *)d_state, d_partial);
        PMX_CHECK_LAUNCH("k_feed_build");
        hipLaunchKernelGGL(k_feed_build_finish<PT>, dim3(1), dim3(1024), 0, ctx->stream, (const u64 *)d_partial, nb, (const PT *)d_pos,
                           d_rev == nullptr, n, ulen, d_len == nullptr, d_state);
        PMX_CHECK_LAUNCH("k_feed_build_finish");
        return PMX_OK;
    }
    hipLaunchKernelGGL((k_feed_reads<PT, LT>), dim3(g), dim3(256), 0, ctx->stream, (const PT *)d_pos, (const LT *)d_len, ulen, d_rev, n,
                       base, nbits, (u64 *)d_F, (const u64 *)d_R, d_state);
    PMX_CHECK_LAUNCH("k_feed_reads");
    hipLaunchKernelGGL((k_feed_finish<PT, LT>), dim3(g), dim3(256), 0, ctx->stream, (const PT *)d_pos, (const LT *)d_len, ulen, d_rev, n,
                       nbits, (u64 *)d_R, d_state);
    PMX_CHECK_LAUNCH("k_feed_finish");
    return PMX_OK;
}

int pmx_launch_feed_reads(pmx_ctx *ctx, uint64_t *d_F, uint64_t *d_R, uint64_t nbits, const void *d_pos, uint32_t pos_bytes,
                          const void *d_len, uint32_t len_bytes, int64_t uniform_len, const unsigned char *d_rev, uint64_t n,
                          uint64_t base, uint64_t *d_state, uint64_t *d_partial)
{
    if (n == 0) return PMX_OK;
    u64 *st = (u64 *)d_state;
    if (len_bytes == 0) d_len = nullptr;     // one length for every read
#define FEED(PT, LT) return launch_feed<PT, LT>(ctx, d_F, d_R, nbits, d_pos, d_len, uniform_len, d_rev, n, base, st, (u64 *)d_partial)
    if (pos_bytes == 4 && (len_bytes == 4 || len_bytes == 0)) FEED(int32_t, int32_t);
    if (pos_bytes == 8 && (len_bytes == 8 || len_bytes == 0)) FEED(int64_t, int64_t);
    if (pos_bytes == 4 && len_bytes == 2) FEED(int32_t, uint16_t);
    if (pos_bytes == 8 && len_bytes == 2) FEED(int64_t, uint16_t);
    if (pos_bytes == 4 && len_bytes == 8) FEED(int32_t, int64_t);
    if (pos_bytes == 8 && len_bytes == 4) FEED(int64_t, int32_t);
#undef FEED
    pmx_set_error("pmx_feed_reads: positions must be 4 or 8 bytes wide (int32 / int64), read lengths 2, 4 or 8 (uint16 / int32 / int64) or 0 (uniform)");
    return PMX_ERR_INVALID;
}

// ---- the 2-bytes-per-read form of a run of reads (pmx_feed_reads_delta16, round 4) ----
// Reads of a sorted BAM lie ~100 bp apart: a word of 16 bits holds the strand (bit 15) and the distance to the read before
// it (bits 0..14).  The run is cut into SEGMENTS of at most FEED_SEG_READS reads -- every FEED_SEG_READS reads, and wherever
// two neighbours lie 32767 bp or more apart --, each with the absolute position of its first read in a table (that read's
// distance field is 0): one workgroup expands a segment with one block scan, nothing is carried between workgroups.
// Output: the 4-bytes-per-read form k_feed_reads takes (int32 position, strand in the top bit).
#define FEED_SEG_READS 1024u
__global__ void __launch_bounds__(64) k_feed_expand16(const unsigned short *__restrict__ words, const u32 *__restrict__ seg_start,
                                                      const int32_t *__restrict__ seg_base, uint64_t n, int32_t *__restrict__ pos)
{
    // ONE wavefront per segment (chromosome 1: 1300 segments in flight instead of 320 workgroups of four; no LDS, no barrier)
    const u32 s = blockIdx.x, lane = threadIdx.x;
    const u32 i0 = seg_start[s], i1 = seg_start[s + 1];      // (the table ends with n)
    const u32 cnt = i1 > i0 ? i1 - i0 : 0u;
    if (cnt == 0 || cnt > FEED_SEG_READS || i1 > n) return;   // (a malformed table is caught on the host; never index beyond n)
    // 16 consecutive reads per lane
    const u32 k0 = 16 * lane;
    u32 w[16], run = 0;
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        w[k] = k0 + k < cnt ? (u32)words[i0 + k0 + k] : 0u;
        run += w[k] & 0x7fffu;
    }
    u32 x = run;   // inclusive scan of the lane sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off, 64);
        if (lane >= (u32)off) x += y;
    }
    u32 p = (u32)seg_base[s] + x - run;
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        p += w[k] & 0x7fffu;
        if (k0 + k < cnt) pos[i0 + k0 + k] = (int32_t)(p | ((w[k] >> 15) << 31));
    }
}

int pmx_launch_feed_expand16(pmx_ctx *ctx, const void *d_words, const void *d_seg_start, const void *d_seg_base, uint32_t nseg,
                             uint64_t n, void *d_pos32)
{
    if (n == 0 || nseg == 0) return PMX_OK;
    hipLaunchKernelGGL(k_feed_expand16, dim3(nseg), dim3(64), 0, ctx->stream, (const unsigned short *)d_words, (const u32 *)d_seg_start,
                       (const int32_t *)d_seg_base, n, (int32_t *)d_pos32);
    PMX_CHECK_LAUNCH("k_feed_expand16");
    return PMX_OK;
}

int pmx_launch_set_regions_on(pmx_ctx *ctx, hipStream_t stream, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to, uint32_t width,
                             uint64_t n, int64_t offset, uint64_t *d_err)
{
    if (n == 0) return PMX_OK;
    const int g = feed_grid(ctx, n, 4);
    if (width == 4)
        hipLaunchKernelGGL(k_set_regions_t<uint32_t>, dim3(g), dim3(256), 0, stream, (u64 *)d_words, nbits, (const uint32_t *)d_from,
                           (const uint32_t *)d_to, n, offset, (u64 *)d_err);
    else if (width == 8)
        hipLaunchKernelGGL(k_set_regions_t<int64_t>, dim3(g), dim3(256), 0, stream, (u64 *)d_words, nbits, (const int64_t *)d_from,
                           (const int64_t *)d_to, n, offset, (u64 *)d_err);
    else {
        pmx_set_error("set_regions: interval ends must be 4 (uint32) or 8 (int64) bytes wide");
        return PMX_ERR_INVALID;
    }
    PMX_CHECK_LAUNCH("k_set_regions_t");
    return PMX_OK;
}

int pmx_launch_regions_build_on(pmx_ctx *ctx, hipStream_t stream, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to,
                                uint32_t width, uint64_t n, int64_t offset, uint64_t *d_err_range, uint64_t *d_err_order)
{
    (void)ctx;
    const uint64_t nwords = (nbits + 63) / 64;
    if (nwords == 0) return PMX_OK;
    const uint64_t g = (nwords + RB_WORDS - 1) / RB_WORDS;
    if (g > 0x7fffffffull) {
        pmx_set_error("regions_build: vector too long");
        return PMX_ERR_INVALID;
    }
    if (width == 4)
        hipLaunchKernelGGL(k_regions_build<uint32_t>, dim3((u32)g), dim3(256), 0, stream, (u64 *)d_words, nbits, (const uint32_t *)d_from,
                           (const uint32_t *)d_to, n, offset, (u64 *)d_err_range, (u64 *)d_err_order);
    else if (width == 8)
        hipLaunchKernelGGL(k_regions_build<int64_t>, dim3((u32)g), dim3(256), 0, stream, (u64 *)d_words, nbits, (const int64_t *)d_from,
                           (const int64_t *)d_to, n, offset, (u64 *)d_err_range, (u64 *)d_err_order);
    else {
        pmx_set_error("regions_build: interval ends must be 4 (uint32) or 8 (int64) bytes wide");
        return PMX_ERR_INVALID;
    }
    PMX_CHECK_LAUNCH("k_regions_build");
    return PMX_OK;
}

int pmx_launch_set_regions_w(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *d_from, const void *d_to, uint32_t width,
                             uint64_t n, int64_t offset, uint64_t *d_err)
{
    return pmx_launch_set_regions_on(ctx, ctx->stream, d_words, nbits, d_from, d_to, width, n, offset, d_err);
}

int pmx_launch_set_positions_w(pmx_ctx *ctx, uint64_t *d_words, uint64_t nbits, const void *d_pos, uint32_t width, uint64_t n,
                               uint64_t *d_err)
{
    if (n == 0) return PMX_OK;
    const int g = feed_grid(ctx, n, 256);
    if (width == 4)
        hipLaunchKernelGGL(k_set_positions_t<uint32_t>, dim3(g), dim3(256), 0, ctx->stream, (u64 *)d_words, nbits, (const uint32_t *)d_pos, n,
                           (u64 *)d_err);
    else if (width == 8)
        hipLaunchKernelGGL(k_set_positions_t<int64_t>, dim3(g), dim3(256), 0, ctx->stream, (u64 *)d_words, nbits, (const int64_t *)d_pos, n,
                           (u64 *)d_err);
    else {
        pmx_set_error("set_positions: positions must be 4 (uint32) or 8 (int64) bytes wide");
        return PMX_ERR_INVALID;
    }
    PMX_CHECK_LAUNCH("k_set_positions_t");
    return PMX_OK;
}
