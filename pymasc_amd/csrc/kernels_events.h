// Event-driven cross-correlation for sparse tiles (gfx950).  Included by kernels_sparse.hip (job table, loaders, scans).
//
// Same sums as k_cc_sparse (mscc.pyx:288-317), enumerated as EVENTS between sorted position lists instead of as
// windows: on read-occupancy vectors (0.5 % density) and a mappability track (one run edge per ~1200 bits) a 64-Kbit
// tile holds ~330 forward reads, ~330 reverse reads and ~55 run edges, and every output is a short sum over pairs:
//
//   ncc[d]       = #{(x in F, y in R) : y - x = d}                                            one event per pair
//   mscc.cc[d]   = #{(x in F&M, y in R) : y - x = d, M[x + c - d]}     (c = read_len - 1)     + one bit of M per pair
//   mscc.fsum[d] = sum_{x in F&M} M[x + c - d]: as a function of d this is a step function that changes only where
//                  x + c - d crosses a run edge.  With E[j] = M[j] - M[j-1]:
//                  M[x+c-d] = M[x+c] - sum_{j = x+c-d+1 .. x+c} E[j], so
//                  fsum[d] = Bf - sum_{t=1..d} GF[t],  Bf = sum_x M[x+c],  GF[t] = sum_x E[x+c-t+1]   (x in F&M)
//                  -> one signed event per (forward read, edge within max_shift below x + c)
//   mscc.rsum[d] = sum_{p in R} a_p(d) b_p(d),  a_p(d) = M[p-d],  b_p(d) = M[p+c-2d]:
//                  a(d)b(d) - a(d-1)b(d-1) = [a(d)-a(d-1)] b(d) + a(d-1) [b(d)-b(d-1)]
//                  a(d)-a(d-1) = -E[p-d+1]                    -> edge j = p-d+1:        GR[d] -= E[j] M[p+c-2d]   (type A)
//                  b(d)-b(d-1) = -E[p+c-2d+1] - E[p+c-2d+2]   -> d = (p+c-j+2) >> 1:    GR[d] -= E[j] M[p-d+1]    (type B)
//                  rsum[d] = R0 + sum_{t=1..d} GR[t],  R0 = sum_p M[p] M[p+c]
// All integer arithmetic; the histograms GF / GR are signed and k_events_tail takes their prefix sums.
//
// The lists are SORTED (position order comes out of block-wide scans of the per-thread popcounts), and nothing is ever
// searched: the first reverse partner of a forward read is its rank among the reverse reads = a popcount of the emitting
// thread's own registers; the reads an EDGE meets (the edge events are driven from the ~60 edges, not from the ~660
// reads) are a contiguous range of a list, entered at the first read of its 512-bit block (a u16 index per block, written
// by the emit).  Cost follows the number of events (~3000 per tile), not tile size x shift range.
//
// Tiles whose lists would overflow (dense vectors: deep data, tests, tracks with very short runs) are flagged and left to
// k_cc_sparse / k_autocorr_edges, which then process the flagged tiles only; their sums are added to what this pass wrote.
#pragma once

#define EV_NQ 2u                          // driver quads per thread and vector: a tile is EV_NQ x 32 Kbit
#define EV_TBW (EV_NQ * SP_TBW)           // 2048 dwords
#define EV_TB (EV_NQ * SP_TB)             // 65536 bits
#define EV_LO 64u                         // dwords of M staged below the tile (2 x max_shift bits), max_shift <= 1023
#define EV_LO_MAX 512u                    // ... in the BIG instantiations (max_shift <= 8191): a kernel argument
#define EV_HI 36u                         // ... and above it (read_len - 1 bits, + the partners' M bits)
// The three lists of a tile (forward reads, reverse reads of the tile + of the max_shift bits above it, run edges of
// everything staged) share ONE pool of LDS: LF at its start, LR behind the forward reads, LE behind the reverse reads
// (offsets known after the block scan).  max_shift <= 1023: a tile is dense -- left to the window kernels -- when its run
// edges alone exceed EV_CAPE_SMALL, or when reads + edges together exceed the pool (2416 entries; 3304 without M): on an
// ordinary track (~60 edges) the reads may fill all of it, ~1.8 % per strand, where the window kernel costs twice the event
// kernel; on a track with edges every 50 bases the reads get what the edges leave.  BIG (the LDS is spent on
// histograms): fixed shares, 768 / 1000 / 384.
#ifndef EV_CAPF
#define EV_CAPF 768u
#endif
#ifndef EV_CAPR
#define EV_CAPR 1000u
#endif
#ifndef EV_CAPE_SMALL
#define EV_CAPE_SMALL 1536u
#endif
#define EV_CAPE(BIG) ((BIG) ? 384u : EV_CAPE_SMALL)
#ifndef EV_POOL_SMALL
#define EV_POOL_SMALL 2416u               // max_shift <= 1023 with M: what 31 KB of LDS per workgroup (FIVE per CU: 31744 B fit,
                                          // 32256 B do not) leave the lists
#endif
#define EV_POOL_ENTRIES(HAS_M, BIG) ((BIG) ? EV_CAPF + EV_CAPR + ((HAS_M) ? 384u : 0u) : ((HAS_M) ? EV_POOL_SMALL : EV_CAPF + EV_CAPR + EV_CAPE_SMALL))
#define EV_RANK_BITS 13u                  // rank field of a forward entry (bits 17..29): index into the reverse list
#ifndef EV_POOL_DEEP
#define EV_POOL_DEEP 4328u                // the DEEP instantiations (PMX_FLAG_DEEP_LISTS: four workgroups per CU, 39.4 KB): ~3.2 % per strand
#endif
#define EV_POS 0x1ffffu                   // 17 bits of biased position (BIG: 16384 + 65536 + 8192 + 1152 staged bits at most)
#define EV_PAD 12u                        // sentinel entries behind the read lists
#define EV_RSENT 0x3fffffffu              // reverse-list sentinel: beyond every range, and (sentinel - lo) stays positive as
                                          // an int32 for range starts lo >= -1023 (an edge below the tile minus read_len - 1)
#define EV_JOBSUM(jobstat) (reinterpret_cast<unsigned long long *>((jobstat) + 4 * SP_MAXJOBS))   // [6 per job] behind the statistics words (max_shift <= 1023)
#define EV_STAT_BYTES (4 * SP_MAXJOBS * sizeof(u32) + 6 * SP_MAXJOBS * sizeof(unsigned long long))
#define EV_SEG_ROWS 6u                    // slab segment rows of `rowlen` u32: ncc, GF, cc, GR, scalars, EE
#define EV_MAX_SHIFT 8191u                // largest max_shift of the event formulation (BIG instantiations)
// s_setprio per phase (as in k_cc_sparse): with equal priorities the SIMD arbitrates by age and the co-resident workgroups
// move in lockstep through the same phase; the staging + emit phase above the event loops: 0.519 -> 0.47 ms (same-box A/B,
// every assignment with staging > events within +-2 % of each other; events above staging: 0.496; events alone raised: 0.52)
#ifndef EV_PRIO_STAGE
#define EV_PRIO_STAGE 3
#endif
// Round 4 (edge events dealt by wave role): the pair loop and the edge loops at DIFFERENT priorities -- same-box A/B, kernel ms:
// pairs / edges 1 / 1: 0.412, 1 / 0: 0.399, 1 / 2: 0.401, 2 / 1: 0.397-0.400, 0 / 2: 0.400, 2 / 0: 0.408, 0 / 1: 0.404 (staging 3
// throughout; staging 2: 0.409): again it is the co-resident workgroups falling out of step that pays, whichever way.
#ifndef EV_PRIO_EVENTS
#define EV_PRIO_EVENTS 2
#endif
#ifndef EV_PRIO_EDGES
#define EV_PRIO_EDGES 1
#endif
#ifndef EV_EDGE_ROLES
#define EV_EDGE_ROLES 1
#endif
#ifndef EV_PAIR_SLOT
#define EV_PAIR_SLOT(w) (w)     // the first block of 64 forward reads a wave takes in the pair loop (then every fourth)
#endif
// (the pair loop deals its blocks of 64 forward reads to waves 0, 1, 2, 3, 0, 1: waves 0 and 1 carry two)
#ifndef EV_ROLE_F
#define EV_ROLE_F 1
#define EV_ROLE_A 2
#define EV_ROLE_B0 3
#define EV_ROLE_B1 0
#endif
#ifndef EV_WAVES
#define EV_WAVES 5
#endif
#ifndef EV_WAVES_NCC
#define EV_WAVES_NCC 8
#endif

#ifdef EV_STAMPS   // diagnostic build: where does a tile's time go (never defined in the shipped library)
#define EV_NSTAMP 12
#define EV_STAMP(i)                                                                                    \
    {                                                                                                  \
        unsigned long long t_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        stamp_acc[i] += t_ - stamp_last;                                                               \
        stamp_last = t_;                                                                               \
    }
#else
#define EV_STAMP(i)
#endif

// ---- LDS (dwords) -------------------------------------------------------------------------------------------------
// One set of HISTOGRAMS per workgroup, shared by its NSG sub-groups of 256 threads (a sub-group = four wavefronts that
// take one 64-Kbit tile per iteration with lists and M words of their own):
//   NC [HN]      one cell per shift: ncc in the low half, mscc.ccbins in the high half -- ONE LDS atomic per pair
//                (NCC-only: plain 32-bit ncc).  Both are counts of pairs of listed forward reads, at most one pair
//                per forward read and shift: the row is flushed to the slab before 65535 listed forward reads.
//   GF, GR, EE   signed.  A dword that holds two 16-bit cells is lo + 65536 hi as an integer (the high cell is added to
//                with value << 16) and is recovered as lo = int16(w), hi = int16((w - lo) >> 16) while |lo|, |hi| < 32768.
//                !BIG: GF[t] (low) and EE[t] (high) share a row [HN] -- the cells of ONE index, like NC --, GR is [HN] i32.
//                BIG: GF and GR rows of 16-bit cells, two per dword (cell k: dword k >> 1, half k & 1), no EE.
//                |GF[t]| <= listed forward reads, |EE[k]| <= listed run edges, |GR[d]| <= 3 x listed reverse reads (type
//                A: one edge per read and shift, type B: two): the rows are flushed before a 16-bit bound reaches 32767.
//   DUMP [64]    an event that misses is not predicated away but added to a dump slot, one per lane (LDS atomics of a wave
//                to ONE address are serialised); every row reaches the slots with a row-relative index.
// 12 (BIG: 8, without EE) bytes of LDS per shift instead of round 2's 20: max_shift 2047 in the LDS of max_shift 1023,
// 8191 shifts + four sub-groups' lists (16 waves per CU) in 145 KB, and -- with the lists' pool at 2416 entries -- 31 KB
// per workgroup below 1024 shifts: FIVE workgroups per CU at 96 VGPRs (same-box A/B: 0.463 -> 0.429 ms).
template <bool HAS_M, bool BIG, bool DEEP = false>
struct EvLds {
    // per sub-group block
    static constexpr u32 LF = 0;                                    // the pool (+EV_PAD behind every list: sentinels; the
    static constexpr u32 POOL = DEEP ? EV_POOL_DEEP : EV_POOL_ENTRIES(HAS_M, BIG);   // loops read ahead of their entry)
    static_assert(POOL + 3 * EV_PAD < (1u << EV_RANK_BITS), "rank field of the forward entries");
    static constexpr u32 WT = LF + POOL + 3 * EV_PAD;               // WT: [5][4 waves] scan totals
    static constexpr u32 MISC = WT + 32;
    static constexpr u32 IDXF = MISC + 16;                          // u16 per 512-bit block of the tile (+ end): list index of
    static constexpr u32 IDXR = IDXF + (HAS_M ? 66u : 0u);          // its first forward / reverse read
    static constexpr u32 MT0 = IDXR + (HAS_M ? 66u : 0u);           // [3] = the dword below the staged range, [4..] = M
    static_assert(MT0 % 4 == 0, "alignment of the 16-byte stores");
    __host__ __device__ static constexpr u32 sg_words(u32 lo, u32 hi = EV_HI) { return MT0 + (HAS_M ? 4u + lo + EV_TBW + hi : 0u); }
    // shared block (hn: entries per row, a multiple of 128)
    __host__ __device__ static constexpr u32 row_words(u32 hn) { return BIG ? hn / 2 : hn; }
    __host__ __device__ static constexpr u32 o_gf(u32 hn) { return hn; }
    __host__ __device__ static constexpr u32 o_gr(u32 hn) { return o_gf(hn) + (HAS_M ? row_words(hn) : 0u); }
    __host__ __device__ static constexpr u32 o_ee(u32 hn) { return o_gf(hn); }   // (!BIG: the high halves of the GF row)
    __host__ __device__ static constexpr u32 o_dump(u32 hn) { return o_gr(hn) + (HAS_M ? row_words(hn) : 0u); }
    __host__ __device__ static constexpr u32 o_xch(u32 hn) { return o_dump(hn) + 64; }    // [0..7] sub-group counts, [8..13] scalars
    __host__ __device__ static constexpr u32 o_sg(u32 hn) { return o_xch(hn) + 16; }
    __host__ __device__ static constexpr u32 total(u32 hn, u32 lo, u32 nsg, u32 hi = EV_HI) { return o_sg(hn) + nsg * sg_words(lo, hi); }
};

struct EvRegs {
    uint4 f[EV_NQ], r[EV_NQ], m[EV_NQ], h;   // quad q covers dwords q*1024 + 4 tid ..; h: one halo quad (see ev_fetch)
    u32 wb[EV_NQ], hbw;                      // WAVE-UNIFORM: the dword of M below lane 0's m[q] / below lane 0's h; the other
                                             // lanes take their neighbour's last dword (ev_below).  A per-thread load of
                                             // M[j - 1] is merged with the quad load into one misaligned 16-byte load + a
                                             // register shuffle that waits for the prefetch right where it is issued.
};

// halo quads, one role per wave so that no wave carries all the extra emission:
//   !BIG: threads [64,80) M below the tile, [128,137) M above it, [192, 192+nhr) R above it
//   BIG:  threads [0, lo/4) M below the tile (up to two waves: 2 x 8191 bits), the other two as above
template <bool BIG>
__device__ __forceinline__ bool ev_role_mlo(u32 tid, u32 lo) { return BIG ? tid < (lo >> 2) : (tid >> 6) == 1 && (tid & 63u) < 16; }
template <bool BIG>
__device__ __forceinline__ u32 ev_role_mlo_index(u32 tid) { return BIG ? tid : (tid & 63u); }

// EV_BIG_NT: the tile bodies of the max_shift > 1023 instantiations are streamed with non-temporal loads -- every word is used
// once, and the lines that ARE reused (the halos, the slab segments the flushes add to) stay in the L2 longer.  Same-box A/B on
// BASELINE config 5: 3.10 -> 3.045 ms per step (0 = plain loads).
#ifndef EV_BIG_NT
#define EV_BIG_NT 1
#endif
typedef u32 ev_v4u __attribute__((ext_vector_type(4)));
template <bool GUARD, bool NT>
__device__ __forceinline__ uint4 ev_ld_body(const u32 *__restrict__ p, int64_t j, uint64_t nbits)
{
    if (GUARD || !NT) return ld_quad<GUARD>(p, j, nbits);
    const ev_v4u v = __builtin_nontemporal_load(reinterpret_cast<const ev_v4u *>(p + j));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <bool HAS_M, bool GUARD, bool BIG>
__device__ __forceinline__ void ev_fetch(EvRegs &er, const u32 *__restrict__ F, const u32 *__restrict__ R,
                                         const u32 *__restrict__ M, int64_t d0, uint64_t nbits, u32 tid, u32 nhr, u32 lo,
                                         bool skip_reads, u32 nhm = EV_HI / 4u)
{
    if (skip_reads) nhr = 0;   // (uniform) the reads of this stretch belong to the window kernel: only M is staged
#pragma unroll
    for (u32 q = 0; q < EV_NQ; q++) {
        const int64_t j = d0 + (int64_t)q * SP_TBW + 4 * (int64_t)tid;
        if (skip_reads) {
            er.f[q] = make_uint4(0, 0, 0, 0);
            er.r[q] = make_uint4(0, 0, 0, 0);
        } else {
            er.f[q] = ev_ld_body<GUARD, BIG && EV_BIG_NT>(F, j, nbits);
            er.r[q] = ev_ld_body<GUARD, BIG && EV_BIG_NT>(R, j, nbits);
        }
        if (HAS_M) {
            er.m[q] = ev_ld_body<GUARD, BIG && EV_BIG_NT>(M, j, nbits);
            const int64_t j0 = d0 + (int64_t)q * SP_TBW + 4 * (int64_t)(tid & ~63u) - 1;   // uniform over the wave
            er.wb[q] = GUARD ? ld_dword_guarded(M, j0, nbits) : M[j0];
        } else {
            er.m[q] = make_uint4(0, 0, 0, 0);
            er.wb[q] = 0;
        }
    }
    // ONE halo load for every lane (lanes without a role re-read their own quad and ignore it): loads of the roles in
    // separate branches target the same registers, and the compiler then waits for everything in flight between them
    const u32 ht = tid & 63u, hw = tid >> 6;
    const bool m_lo = HAS_M && ev_role_mlo<BIG>(tid, lo), m_hi = HAS_M && hw == 2 && ht < nhm, r_hi = hw == 3 && ht < nhr;
    const u32 *hp = (m_lo || m_hi || skip_reads) ? M : R;   // (skip_reads without M does not occur)
    int64_t jh = d0 + 4 * (int64_t)tid;
    if (m_lo) jh = d0 - (int64_t)lo + 4 * (int64_t)ev_role_mlo_index<BIG>(tid);
    if (m_hi || r_hi) jh = d0 + EV_TBW + 4 * (int64_t)ht;
    er.h = ld_quad<GUARD>(hp, jh, nbits);
    er.hbw = 0;
    if (HAS_M) {
        // the dword below lane 0's halo quad (uniform over the wave): waves with quads of M below the tile, else above it
        const bool lo_wave = BIG ? hw < 2 : hw == 1;
        const int64_t jh0 = (lo_wave ? d0 - (int64_t)lo + (BIG ? 256 * (int64_t)hw : 0) : d0 + EV_TBW) - 1;
        er.hbw = GUARD ? ld_dword_guarded(M, jh0, nbits) : M[jh0];
    }
}

// The interior tile of the max_shift <= 1023 instantiations (round 4): every address is a UNIFORM base -- vector + tile, in
// scalar registers -- plus a per-thread 32-bit byte offset, so the loads take the scalar-base form and the address
// arithmetic is a handful of scalar instructions instead of ~80 vector ones (64-bit per-thread pointer arithmetic for
// every load; the prefetch was 4.7 % of a tile).  The halo quad is loaded under a wave-uniform role (`wave` is a scalar):
// wave 1 M below the tile, wave 2 M above it, wave 3 R above it (lanes beyond the role re-read the role's last quad).
template <bool HAS_M>
__device__ __forceinline__ void ev_fetch_fast(EvRegs &er, const SpJobRegs &jb, u32 local_tile, u32 tid, u32 wave, u32 nhr)
{
    const size_t tb = (size_t)local_tile * (EV_TBW * 4u);   // (uniform) byte offset of the tile in its vectors
    const char *fb = reinterpret_cast<const char *>(jb.F) + tb;
    const char *rb = reinterpret_cast<const char *>(jb.R) + tb;
    const char *mb = reinterpret_cast<const char *>(jb.M) + tb;
    const u32 voff = 16u * tid, lane = tid & 63u;
    // (every offset is formed as ONE 32-bit per-thread value added to a scalar base: an immediate beyond +-4 KB would
    // push the compiler back to 64-bit per-thread pointers)
#pragma unroll
    for (u32 q = 0; q < EV_NQ; q++) {
        const u32 vq = voff + q * (SP_TBW * 4u);
        er.f[q] = *reinterpret_cast<const uint4 *>(fb + vq);
        er.r[q] = *reinterpret_cast<const uint4 *>(rb + vq);
        if (HAS_M) {
            er.m[q] = *reinterpret_cast<const uint4 *>(mb + vq);
            // (uniform) the dword below lane 0's quad.  SIGNED: wave 0 of quad 0 reads the dword BELOW the tile -- as an
            // unsigned 32-bit offset "0 - 4" is +4 GB (a memory fault on the first interior tile, caught in review of the
            // round-4 work in progress)
            const int32_t sq = (int32_t)(q * (SP_TBW * 4u) + wave * 1024u) - 4;
            er.wb[q] = *reinterpret_cast<const u32 *>(mb + sq);
        } else {
            er.m[q] = make_uint4(0, 0, 0, 0);
            er.wb[q] = 0;
        }
    }
    // ONE halo load for every lane, from a base the wave's role picks (scalar selects, no branch: loads of the roles in
    // separate branches target the same registers, and the compiler waits for everything in flight between them):
    // wave 1 M below the tile, wave 2 M above it, wave 3 R above it; lanes beyond the role repeat its last quad; wave 0
    // re-reads a quad of R and ignores it.  hbw: the dword below the first halo quad (waves 1 and 2).
    const char *hb = (HAS_M && wave == 1) ? mb - EV_LO * 4u : (HAS_M && wave == 2) ? mb + EV_TBW * 4u : wave == 3 ? rb + EV_TBW * 4u : rb;
    const u32 lim = (HAS_M && wave == 1) ? 63u : (HAS_M && wave == 2) ? 8u : wave == 3 ? nhr - 1u : 63u;
    const u32 vh = 16u * (lane < lim ? lane : lim);
    er.h = *reinterpret_cast<const uint4 *>(hb + vh);
    er.hbw = HAS_M ? *reinterpret_cast<const u32 *>(hb - 4) : 0u;
}

template <bool HAS_M, bool BIG>
__device__ __forceinline__ void ev_fetch_job(EvRegs &er, const SpJobRegs &jb, u32 local_tile, u32 tid, u32 nhr, u32 lo,
                                             bool skip_reads = false, u32 wave = 0, u32 hi_dwords = EV_HI)
{
    const int64_t d0 = (int64_t)local_tile * EV_TBW;
    const int64_t low = d0 - (int64_t)lo - 1;
    const u32 above = (BIG && 4 * nhr > hi_dwords) ? 4 * nhr : hi_dwords;   // dwords read above the tile (M: hi_dwords, R: 4 nhr)
    const uint64_t hi = (uint64_t)d0 + EV_TBW + above;
    const bool interior = jb.aligned16 && low >= 0 && hi + 2 <= jb.nbits / 32;
    // (tried: raw buffer loads from a scalar resource + one 32-bit lane offset for the interior tile -- 67 fewer vector
    // but 160 more scalar instructions in the kernel, 1.4 % slower with M, 9 % slower NCC-only: same-box A/B, round 3)
#ifndef EV_NO_FAST_FETCH
    if (!BIG && interior && !skip_reads)
        ev_fetch_fast<HAS_M>(er, jb, local_tile, tid, wave, nhr);
    else
#endif
    if (interior)
        ev_fetch<HAS_M, false, BIG>(er, jb.F, jb.R, jb.M, d0, jb.nbits, tid, nhr, lo, skip_reads, hi_dwords / 4u);
    else
        ev_fetch<HAS_M, true, BIG>(er, jb.F, jb.R, jb.M, d0, jb.nbits, tid, nhr, lo, skip_reads, hi_dwords / 4u);
}

// forward list entry: bits 0..16 biased position, 17..29 index of the first reverse read at or above it, 31 = M[x]
__device__ __forceinline__ void ev_emit_f(const uint4 f, const uint4 r, const uint4 m, u32 idx, u32 rank0, u32 base_bit,
                                          u32 *list)
{
    const u64 fs[2] = {(u64)f.x | ((u64)f.y << 32), (u64)f.z | ((u64)f.w << 32)};
    const u64 rs[2] = {(u64)r.x | ((u64)r.y << 32), (u64)r.z | ((u64)r.w << 32)};
    const u64 ms[2] = {(u64)m.x | ((u64)m.y << 32), (u64)m.z | ((u64)m.w << 32)};
    u32 rk = rank0;
#pragma unroll
    for (u32 k = 0; k < 2; k++) {
        u64 ww = fs[k];
        while (ww) {
            const u32 b = (u32)__builtin_ctzll(ww);
            ww &= ww - 1;
            const u64 below = (rs[k] << (63u - b)) << 1;
            list[idx] = (base_bit + 64u * k + b) | ((rk + (u32)__popcll(below)) << 17) | ((u32)((ms[k] >> b) & 1ull) << 31);
            idx++;
        }
        rk += (u32)__popcll(rs[k]);
    }
}
__device__ __forceinline__ void ev_emit_pos(const uint4 v, u32 idx, u32 base_bit, u32 *list)
{
    const u64 vs[2] = {(u64)v.x | ((u64)v.y << 32), (u64)v.z | ((u64)v.w << 32)};
#pragma unroll
    for (u32 k = 0; k < 2; k++) {
        u64 ww = vs[k];
        while (ww) {
            const u32 b = (u32)__builtin_ctzll(ww);
            ww &= ww - 1;
            list[idx] = base_bit + 64u * k + b;
            idx++;
        }
    }
}

// the dword below this lane's quad: the last dword of the lane below it (lane 0: `first`, uniform over the wave)
__device__ __forceinline__ u32 ev_below(u32 last_dword, u32 first)
{
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)last_dword, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

__device__ __forceinline__ u32 ev_mbit(const u32 *MT, u32 q) { return (MT[q >> 5] >> (q & 31u)) & 1u; }

// signed add to cell `cell` of a GF / GR / EE row (see EvLds)
template <bool BIG>
__device__ __forceinline__ void ev_add_cell(u32 *row, u32 cell, u32 val)
{
    if (BIG)
        atomicAdd(&row[cell >> 1], val << ((cell & 1u) << 4));
    else
        atomicAdd(&row[cell], val);
}

// BIG with the mappable-length pairs: row 5 of a (workgroup, job) segment takes WORKGROUP-scope atomic adds (done in the
// XCD's L2: the row is this workgroup's alone until the kernel ends); it is cleared with plain stores when the workgroup
// enters the job, and they have been performed before the workgroup's next barrier
__device__ __forceinline__ void ev_zero_row5(u32 *__restrict__ slab, u32 segment, u32 hn, u32 gt, u32 nt)
{
    u32 *row = slab + ((size_t)segment * EV_SEG_ROWS + 5u) * hn;
    for (u32 i = gt; i < hn; i += nt) row[i] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// slab segment of a (workgroup, job) pair: EV_SEG_ROWS rows of `hn` u32: ncc, GF, cc, GR, scalars (|F|, |R|, Bf, R0,
// popcount(M), runs), EE
// DO_MLEN: also the pairs of run edges (the mappable-length autocorrelation, see k_autocorr_pairs): EE[k] = sum of
// E[j] E[j + k] over the edges j of the tile, k = 1..max_lag, and popcount(M) / the runs starting in the tile.
// NSG sub-groups of 256 threads; BIG: max_shift up to EV_MAX_SHIFT, geometry in the arguments `hn_arg` (entries per
// histogram row) and `lo_arg` (dwords of M staged below a tile), histograms in dynamic LDS; !BIG: max_shift <= 1023.
// DEEP (max_shift <= 1023 with M only): a pool of EV_POOL_DEEP entries at four workgroups per CU, for callers that know their
// data is deep (PMX_FLAG_DEEP_LISTS): between ~1.5 % and ~3.2 % read starts per strand the event formulation is still twice
// as fast as the window kernels, but the five-per-CU instantiation would hand most tiles over.
template <bool HAS_M, bool DO_NCC, bool DO_MLEN, u32 NSG, bool BIG, typename JT = SpJobTable, bool DEEP = false>
__global__ void __launch_bounds__(256 * NSG, BIG ? 4 : (HAS_M ? (DEEP ? 4 : EV_WAVES) : EV_WAVES_NCC))
k_cc_events(const JT jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg, u32 c, u32 S, u32 nhr, u32 max_lag,
            u32 hn_arg, u32 lo_arg, u32 hi_arg, u32 *__restrict__ slab, unsigned char *__restrict__ tile_flags,
            unsigned char *__restrict__ tile_flags_ac, u32 *__restrict__ n_flagged, u32 *__restrict__ jobstat)
{
    typedef EvLds<HAS_M, BIG, DEEP> L;
    static_assert(!DEEP || (HAS_M && !BIG), "DEEP: the max_shift <= 1023 instantiations with a track");
    static_assert(BIG || NSG == 1, "the max_shift <= 1023 instantiations are one sub-group per workgroup");
    constexpr u32 NT = 256 * NSG;
    extern __shared__ __align__(16) u32 ev_dyn_lds[];
    __shared__ __align__(16) u32 ev_static_lds[BIG ? 4 : L::total(1024, EV_LO, 1)];
    u32 *const lds = BIG ? ev_dyn_lds : ev_static_lds;
    const u32 HN = BIG ? hn_arg : 1024u;      // entries per histogram row
    const u32 LO = BIG ? lo_arg : EV_LO;      // dwords of M staged below the tile
    const u32 HI = BIG ? (hi_arg & 0xffffu) : EV_HI;   // ... and above it (BIG with the mappable-length pairs: max_lag bits, a multiple of 4 dwords)
    // BIG with the mappable-length pairs: their row of lags lives in LDS behind the sub-groups' blocks when the launcher found
    // room for it without giving up a resident wavefront (bit 16 of hi_arg; ev_big_plan), else in the slab (atomics in the L2)
    const bool EE_LDS = BIG && DO_MLEN && (hi_arg >> 16) != 0;
    const u32 BIAS = LO * 32u;                // list positions are relative to the first staged bit of M
    const u32 sg = NSG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0u;
    u32 *const hN = lds;
    u32 *const hGF = lds + L::o_gf(HN);
    u32 *const hGR = lds + L::o_gr(HN);
    u32 *const hEE = lds + L::o_ee(HN);
    u32 *const xch = lds + L::o_xch(HN);
    u32 *const sgb = lds + L::o_sg(HN) + sg * L::sg_words(LO, HI);
    u32 *const hEEb = lds + L::total(HN, LO, NSG, HI);   // (EE_LDS: HN / 2 dwords, two signed 16-bit cells each, as rows GF and GR)
    u32 *const MT = sgb + L::MT0 + 4;
    unsigned short *const idxF = reinterpret_cast<unsigned short *>(sgb + L::IDXF);
    unsigned short *const idxR = reinterpret_cast<unsigned short *>(sgb + L::IDXR);
    u32 *const LF = sgb + L::LF;
    u32 *const wt = sgb + L::WT;

    const u32 gt = threadIdx.x;              // thread of the workgroup (histogram clears / flushes)
    const u32 tid_ = NSG > 1 ? (threadIdx.x & 255u) : threadIdx.x;   // thread of the sub-group
    const u32 wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);      // wave of the sub-group
    const u32 g0 = blockIdx.x * tiles_per_wg;
    const u32 g1 = g0 + tiles_per_wg < total_tiles ? g0 + tiles_per_wg : total_tiles;
    if (g0 >= g1) return;
    {
#pragma nounroll
        for (u32 i = gt; i < L::o_sg(HN); i += NT) lds[i] = 0;
        if (EE_LDS)
            for (u32 i = gt; i < HN / 2; i += NT) hEEb[i] = 0;
    }

    FIRST_JOB(ji, JT, jobs, njobs, g0)
    u32 jn = ji;
    if (BIG && DO_MLEN && !EE_LDS) ev_zero_row5(slab, blockIdx.x + ji, HN, gt, NT);
    EvRegs er;
    SpJobRegs pj;   // job of the tiles being prefetched (index jn), held in scalar registers
    load_job(pj, jobs.j[ji]);
    // an iteration takes the NSG tiles g .. g + NSG - 1 of ONE job (the histograms belong to a (workgroup, job) pair);
    // sub-groups beyond the end of the job / of the workgroup's range sit the iteration out (they keep the barriers)
    bool act_var = true;
    if (NSG > 1) {
        const u32 lim = pj.tile_end < g1 ? pj.tile_end : g1;
        act_var = g0 + sg < lim;
    }
    if (NSG == 1 || act_var) ev_fetch_job<HAS_M, BIG>(er, pj, g0 + sg - pj.tile0, tid_, nhr, LO, false, wave, HI);
    // Read-dense stretches (deep data: every tile far above the list capacities): after two such tiles in a row the
    // workgroup hands the REST of its tile range in this chromosome to the window kernels in one go (flags only, nothing
    // staged): the event kernel then costs two tiles per workgroup instead of a wasted pass over everything.  (NSG == 1)
    // Dense chromosomes (NSG == 1).  Every workgroup adds what it sees to three counters of the JOB (jobstat: tiles seen,
    // read-dense, edge-dense tiles), and once most tiles of a chromosome turn out dense everybody stops staging it tile by
    // tile -- job_mode: 1 = the reads of the rest of my range go to the cross-correlation window kernel unseen (the run
    // edges are still listed here when the mappable-length pass is fused: F and R are not even loaded), 2 = the whole
    // rest goes to both window kernels.  A job-wide verdict keeps the workgroups of a chromosome together: with a
    // per-workgroup rule the unlucky ones walked their whole range while the others had left (round 3, first try), and a
    // density right at a list capacity paid for both kernels (round 2: +30 % between 1.1 % and 2 % reads per strand).
    // Isolated dense tiles -- a pile-up, a repeat -- are flagged one by one whatever the verdict.
    u32 job_mode = 0;                  // (uniform)
    u32 unreported = 0;                // (uniform) tiles of this job seen since my last report to jobstat
    bool had_dense = false;            // (uniform) a verdict may have been left in LDS since the job began
    bool cur_skip = false, next_skip = false;   // (uniform) the tile in the registers / the next fetch leaves F and R out
    u32 cur_tile0 = pj.tile0, cur_flag0 = pj.flag0;   // of the job whose tiles are being processed (index ji)
    u32 cntB = 0, cnt0 = 0;                       // per-thread: Bf, R0 of the tiles taken here (|F|, |R|: uniform per tile, added
                                                  // to xch[8], xch[9] in LDS by one thread: no register lives across tiles for them)
    u32 cntM = 0, cntU = 0;                       // DO_MLEN: popcount(M), runs starting in them
    u32 accF = 0, accR = 0, accE = 0;             // (uniform) listed forward / reverse reads / run edges since the histograms' last flush
    bool seg_written = false;                     // (uniform) this (workgroup, job) segment already holds a flush
    bool seg_gr_written = false;                  // (uniform) ... and its GR row (BIG: that row is also flushed alone, below)
#ifdef EV_STAMPS
    unsigned long long stamp_acc[EV_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    for (u32 g = g0; g < g1;) {
        const bool act = NSG == 1 ? true : act_var;   // (uniform over the sub-group)
        EV_STAMP(9)
        __syncthreads();   // B0: every wave is done with the previous tile's lists and M words
        EV_STAMP(0)
        if (EV_PRIO_STAGE) __builtin_amdgcn_s_setprio(EV_PRIO_STAGE);
        // ---- phase A: from the prefetched registers ----
        // (thread-index derived addresses are recomputed per tile from an opaque copy: hoisted out of the tile loop they
        // cost ~30 registers for its whole lifetime and push the allocation into scratch)
        u32 tid = tid_;
        asm volatile("" : "+v"(tid));
        const u32 lane = tid & 63;
        u32 cF[EV_NQ] = {0, 0}, cR[EV_NQ] = {0, 0}, cE[EV_NQ] = {0, 0}, cEh = 0, cRh = 0, pendM = 0, pendU = 0;
        const bool h_below = HAS_M && ev_role_mlo<BIG>(tid, LO), h_above = HAS_M && wave == 2 && lane < HI / 4u;
        const u32 hb = ev_role_mlo_index<BIG>(tid);
        const bool h_r = wave == 3 && lane < nhr;
        u32 pF = 0, pR = 0, pE = 0, pH = 0, sF = 0, sR = 0, sH = 0, sE = 0, sX = 0;
        if (act) {
#pragma unroll
            for (u32 q = 0; q < EV_NQ; q++) {
                cF[q] = popc4(er.f[q]);
                cR[q] = popc4(er.r[q]);
                if (HAS_M) {
                    const u32 bel = ev_below(er.m[q].w, er.wb[q]);
                    cE[q] = popc4(edge_words(er.m[q], bel));   // (the edge words are recomputed when they are emitted)
                    *reinterpret_cast<uint4 *>(MT + LO + q * SP_TBW + 4 * tid) = er.m[q];
                    if (DO_MLEN) {
                        pendM += popc4(er.m[q]);
                        // run starts = rising edges; rising - falling = M[last bit of the quad] - M[bit before it]
                        pendU += (cE[q] + (er.m[q].w >> 31) - (bel >> 31)) >> 1;
                    }
                }
            }
            const u32 hbel = HAS_M ? ev_below(er.h.w, er.hbw) : 0u;   // (all lanes: DPP reads the lane below)
            if (h_below || h_above) {
                cEh = popc4(edge_words(er.h, hbel));
                *reinterpret_cast<uint4 *>(MT + (h_below ? 4 * hb : LO + EV_TBW + 4 * lane)) = er.h;
                if (h_below && hb == 0) MT[-1] = er.hbw;
            }
            if (h_r) cRh = popc4(er.h);
            // position order = (row q, thread, word, bit): exclusive offsets from packed block scans (a row holds <= 32768
            // bits and the halos <= 16384, so the 16-bit fields never carry)
            pF = cF[0] | (cF[1] << 16);
            pR = cR[0] | (cR[1] << 16);
            pE = cE[0] | (cE[1] << 16);
            pH = h_below ? cEh : cEh << 16;   // below | above
            sF = pF;
            sR = pR;
            sH = cRh;
            sE = pE;
            sX = pH;
            if (HAS_M)
                wave_inclusive_scan5(sF, sR, sH, sE, sX);
            else
                wave_inclusive_scan3(sF, sR, sH);
            if (lane == 63) {
                wt[0 + wave] = sF;
                wt[4 + wave] = sR;
                wt[8 + wave] = sH;
                if (HAS_M) {
                    wt[12 + wave] = sE;
                    wt[16 + wave] = sX;
                }
            }
        }
        EV_STAMP(1)
        __syncthreads();   // Bs
        EV_STAMP(2)
        u32 nF = 0, nRt = 0, nR = 0, nE = 0, TXb = 0, TE0 = 0, TE1 = 0;
        bool dense = false, do_edges = false;
        u32 *LR = LF, *LE = LF;
        u32 gnext = g + NSG;
        if (act) {
            u32 bF = 0, bR = 0, bH = 0, bE = 0, bX = 0, tF = 0, tR = 0, tH = 0, tE = 0, tX = 0;
#pragma unroll
            for (u32 w = 0; w < 4; w++) {
                const u32 xF = wt[w], xR = wt[4 + w], xH = wt[8 + w], xE = HAS_M ? wt[12 + w] : 0u, xX = HAS_M ? wt[16 + w] : 0u;
                if (w < wave) {
                    bF += xF;
                    bR += xR;
                    bH += xH;
                    bE += xE;
                    bX += xX;
                }
                tF += xF;
                tR += xR;
                tH += xH;
                tE += xE;
                tX += xX;
            }
            const u32 TF0 = tF & 0xffffu, TR0 = tR & 0xffffu;
            TE0 = tE & 0xffffu;
            TE1 = tE >> 16;
            TXb = tX & 0xffffu;
            nF = __builtin_amdgcn_readfirstlane(TF0 + (tF >> 16));
            nRt = __builtin_amdgcn_readfirstlane(TR0 + (tR >> 16));   // reverse reads inside the tile: the rsum drivers
            nR = __builtin_amdgcn_readfirstlane(nRt + tH);            // + the partners above it
            nE = __builtin_amdgcn_readfirstlane(TXb + TE0 + TE1 + (tX >> 16));
            // A tile is left to the window kernels when its lists would overflow.  Too many EDGES: both window kernels take it.
            // Too many READS only: the cross-correlation window kernel takes it, but its run edges are still listed here for the
            // edge pairs of the mappable-length pass (DO_MLEN), so that deep data on an ordinary track does not push that pass onto
            // its window kernel as well.
            const bool dense_e = HAS_M && nE > EV_CAPE(BIG);
            const bool dense_r = cur_skip || (BIG ? (nF > EV_CAPF || nR > EV_CAPR)
                                                  : nF + nR + ((HAS_M && !dense_e) ? nE : 0u) > L::POOL);
            dense = dense_e || dense_r;
            do_edges = HAS_M && !dense_e && (!dense || DO_MLEN);
            LR = LF + (dense ? 0u : nF) + EV_PAD;   // (uniform) this tile's share of the pool
            LE = LR + (dense ? 0u : nR) + EV_PAD;
            const u32 my_tile = g + sg;
            if (!dense) {
                const u32 eF = bF + sF - pF, eR = bR + sR - pR;   // exclusive, per row
                const u32 oR0 = eR & 0xffffu, oR1 = TR0 + (eR >> 16);
                ev_emit_f(er.f[0], er.r[0], er.m[0], eF & 0xffffu, oR0, BIAS + 0 * SP_TB + 128u * tid, LF);
                ev_emit_f(er.f[1], er.r[1], er.m[1], TF0 + (eF >> 16), oR1, BIAS + 1 * SP_TB + 128u * tid, LF);
                ev_emit_pos(er.r[0], oR0, BIAS + 0 * SP_TB + 128u * tid, LR);
                ev_emit_pos(er.r[1], oR1, BIAS + 1 * SP_TB + 128u * tid, LR);
                if (h_r) ev_emit_pos(er.h, nRt + bH + sH - cRh, BIAS + EV_TB + 128u * lane, LR);
                // sentinels behind the lists: the event loops need no index bounds (and read ahead of their entry)
                if (tid < EV_PAD) LR[nR + tid] = EV_RSENT;
                else if (HAS_M && tid < 2 * EV_PAD) LF[nF + tid - EV_PAD] = EV_POS;
                if (HAS_M) {
                    // list index of the first read of every 512-bit block of the tile (the edge-driven loops start there)
                    if ((tid & 3u) == 0) {
                        idxF[tid >> 2] = (unsigned short)(eF & 0xffffu);
                        idxF[64 + (tid >> 2)] = (unsigned short)(TF0 + (eF >> 16));
                        idxR[tid >> 2] = (unsigned short)oR0;
                        idxR[64 + (tid >> 2)] = (unsigned short)oR1;
                    }
                    if (tid == 255) {
                        idxF[128] = (unsigned short)nF;
                        idxR[128] = (unsigned short)nRt;
                    }
                }
                if (tid == 0) {
                    if (NSG == 1) {   // one thread between two barriers: plain adds (the atomic optimiser wraps an LDS atomic in a
                        xch[8 + 0] += nF;   // wave reduction of ~14 instructions, even under tid == 0)
                        xch[8 + 1] += nRt;
                    } else {          // (thread 0 of EVERY sub-group comes here)
                        atomicAdd(&xch[8 + 0], nF);
                        atomicAdd(&xch[8 + 1], nRt);
                    }
                }
            } else {
                // dense tile: left to k_cc_sparse (both of its 32-Kbit tiles; the flag array is padded per job)
                if (tid == 0) {
                    const u32 f = cur_flag0 + EV_NQ * (my_tile - cur_tile0);
                    for (u32 i = 0; i < EV_NQ; i++) tile_flags[f + i] = 1;
                    if (DO_MLEN && dense_e) tile_flags_ac[cur_flag0 + (my_tile - cur_tile0)] = 1;   // (its window tile is this tile; same padded indexing)
                    if (NSG > 1) {
                        atomicAdd(n_flagged, 1u);
                        if (DO_MLEN && dense_e) atomicAdd(n_flagged + 1, 1u);
                    }
                }
                if (NSG == 1 && tid == 0) {      // (added to n_flagged when the workgroup leaves the job)
                    xch[14] += 1;
                    if (DO_MLEN && dense_e) xch[15] += 1;
                }
            }
            if (do_edges) {
                // the run edges (+ those of the halos) in position order
                const u32 eE = bE + sE - pE, eX = bX + sX - pH;
                const u32 oE0 = TXb + (eE & 0xffffu), oE1 = TXb + TE0 + (eE >> 16);
#pragma unroll
                for (u32 q = 0; q < EV_NQ; q++) {
                    const uint4 Eq = edge_words(er.m[q], ev_below(er.m[q].w, er.wb[q]));
                    const u32 o = q ? oE1 : oE0;
                    ap_emit(Eq, er.m[q], o, BIAS + q * SP_TB + 128u * tid, LE);
                }
                const uint4 Eh = edge_words(er.h, ev_below(er.h.w, er.hbw));
                if (h_below) {
                    ap_emit(Eh, er.h, eX & 0xffffu, 128u * hb, LE);
                } else if (h_above) {
                    const u32 oa = TXb + TE0 + TE1 + (eX >> 16);
                    ap_emit(Eh, er.h, oa, BIAS + EV_TB + 128u * lane, LE);
                    // (BIG with the mappable-length pairs: the run edges further above the tile than the read length are in
                    // the list as PARTNERS of the pairs only; the edge events of fsum / rsum end in front of them)
                    if (BIG && DO_MLEN && lane == EV_HI / 4u) sgb[L::MISC + 1] = oa;
                }
                if (tid >= 2 * EV_PAD && tid < 2 * EV_PAD + 4) LE[nE + tid - 2 * EV_PAD] = EV_POS;   // sentinels
                cntM += pendM;
                cntU += pendU;
            }
            if (NSG == 1) {
                // the job's counters.  A workgroup reports when it meets a dense tile -- that tile and the tiles it has seen since
                // its last report, so every report is unbiased -- and leaves its verdict for everybody (read after B1): the
                // ordinary tile costs no global atomic, and a workgroup that never sees a dense tile never asks for a verdict
                // (a dense REGION of a chromosome is handed over by the workgroups inside it only).
                if (!dense) {
                    unreported++;
                } else {
                    if (tid == 0 && job_mode == 0) {
                        u32 *st = jobstat + 4 * ji;
                        const u32 seen = atomicAdd(st, unreported + 1) + unreported + 1;
                        const u32 nr = atomicAdd(st + 1, dense_r ? 1u : 0u) + (dense_r ? 1u : 0u);
                        const u32 ne = atomicAdd(st + 2, dense_e ? 1u : 0u) + (dense_e ? 1u : 0u);
                        u32 verdict = 0;
                        if (seen >= 16 && 10 * ne > 6 * seen) verdict = 2;
                        else if (seen >= 16 && 10 * (nr + ne) > 6 * seen) verdict = 1;
                        sgb[L::MISC] = verdict;
                    }
                    unreported = 0;
                    had_dense = true;
                }
                if (job_mode == 2 || (job_mode == 1 && !(HAS_M && DO_MLEN))) {
                    // the rest of my range in this chromosome: flags only, nothing staged
                    const u32 end = pj.tile_end < g1 ? pj.tile_end : g1;   // (pj is still the job of tile g here)
                    const bool ac_too = DO_MLEN && job_mode == 2;
                    for (u32 t = g + 1 + tid; t < end; t += 256) {
                        const u32 f = cur_flag0 + EV_NQ * (t - cur_tile0);
                        for (u32 i = 0; i < EV_NQ; i++) tile_flags[f + i] = 1;
                        if (ac_too) tile_flags_ac[cur_flag0 + (t - cur_tile0)] = 1;
                    }
                    if (tid == 0 && end > g + 1) {
                        xch[14] += end - (g + 1);
                        if (ac_too) xch[15] += end - (g + 1);
                    }
                    gnext = end;
                } else if (job_mode == 1) {
                    next_skip = true;   // from the next fetch on
                }
            }
        }
        if (NSG > 1) {
            // the listed reads of every sub-group (flush bookkeeping), and the end of the iteration's job
            if (tid == 0) {
                xch[sg] = (act && !dense) ? (nF | (nRt << 16)) : 0u;
                if (DO_MLEN) xch[4 + sg] = (act && do_edges) ? nE : 0u;   // (the bound of the row of lags in LDS, EE_LDS)
            }
            const u32 end = pj.tile_end < g1 ? pj.tile_end : g1;   // (pj is still the job of tile g here)
            if (gnext > end) gnext = end;
        }
        EV_STAMP(3)
        // ---- prefetch the next tile into the (now free) registers ----
        // (BIG with the mappable-length pairs: issued BEHIND their global atomics, after B1 -- the memory operations of a wave
        // complete in issue order, so loads queued behind the atomics wait for them by the time they are needed, a tile
        // later, at no cost; atomics queued behind the loads were waited for at every barrier of the next tile: +22 %)
        constexpr bool LATE_FETCH = BIG && DO_MLEN;
        auto prefetch_next = [&]() {
            if (gnext < g1) {
                if (gnext >= pj.tile_end) {   // rare: the next tile belongs to the next job
                    jn = ji + 1;
                    load_job(pj, jobs.j[jn]);
                    next_skip = false;
                }
                if (NSG > 1) {
                    const u32 lim = pj.tile_end < g1 ? pj.tile_end : g1;
                    act_var = gnext + sg < lim;
                }
                if (NSG == 1 || act_var) ev_fetch_job<HAS_M, BIG>(er, pj, gnext + sg - pj.tile0, tid, nhr, LO, NSG == 1 && next_skip, wave, HI);
            } else if (NSG > 1) {
                act_var = false;
            }
        };
        if (!LATE_FETCH) prefetch_next();
        EV_STAMP(4)
        if (EV_PRIO_STAGE) __builtin_amdgcn_s_setprio(0);
        __syncthreads();   // B1: lists, M words and edge ranks visible
        // (tried on top, same-box A/B: the two waves with two pair blocks one level above the others: 0.400 -> 0.412; the two
        // priorities exchanged in every other "layer" of 256 / 8 / 1 workgroups: +-0)
        if (EV_PRIO_EVENTS) __builtin_amdgcn_s_setprio(EV_PRIO_EVENTS);
        EV_STAMP(5)
        if (LATE_FETCH) {
            if (do_edges) {
                // beyond 1023 lags (round 4): no LDS is left for a row of lags (the histograms of 5000 shifts fill a CU at two
                // workgroups), and the pairs are few -- ~60 edges per tile, ~4 partners each --: they are added straight to row 5
                // of the (workgroup, job) segment in the slab with device-scope atomics (zeroed when the workgroup entered
                // the job).  This replaces the pair pass over M (k_autocorr_pairs: 0.44 ms of config 5's 3.7, M read twice).
                u32 *const gEE = slab + ((size_t)(blockIdx.x + ji) * EV_SEG_ROWS + 5u) * HN;
                const u32 nEt = TE0 + TE1;
                // (one wave per block of 64 edges.  Tried: every wave walking all edges and taking every fourth partner, as in the
                // edge events -- four times the atomic instructions with a quarter of the lanes each: 3.45 -> 3.5-3.8 ms, unstable)
                for (u32 b = (wave + 1) & 3; 64 * b < nEt; b += 4) {
                    const u32 i = 64 * b + lane;
                    const bool in = i < nEt;
                    const u32 ent = in ? LE[TXb + i] : 0u;
                    const u32 pos = ent & EV_POS, hi = pos + max_lag;
                    u32 e = in ? TXb + i + 1 : nE;
                    u32 e0 = LE[e];
                    bool h0 = (e0 & EV_POS) <= hi;
                    while (__ballot(h0)) {
                        // WORKGROUP scope: the row belongs to this workgroup alone while the kernel runs, and its waves share one
                        // CU, hence one L2 -- the add is done there.  At device scope every add is a memory-side operation on a
                        // line of its own: 37 M of them per config-5 step cost the kernel +0.65 ms (measured).
#ifndef EV_ABL_NOEE
                        // (EE_LDS, uniform: the row is in LDS -- config 5's 37 M adds per step each cost 32 bytes of write traffic
                        // between the L2 and the memory side, 1.0 GB of the step's 5.9)
                        if (EE_LDS) {
                            const u32 k = (e0 & EV_POS) - pos;
                            if (h0) atomicAdd(&hEEb[k >> 1], (u32)(((int32_t)(e0 ^ ent) >> 31) | 1) << ((k & 1u) << 4));
                        } else if (h0) {
                            __hip_atomic_fetch_add(&gEE[(e0 & EV_POS) - pos], (u32)(((int32_t)(e0 ^ ent) >> 31) | 1), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
#endif
                        e = h0 ? e + 1 : nE;
                        e0 = LE[e];
                        h0 = (e0 & EV_POS) <= hi;
                    }
                }
            }
            prefetch_next();
        }
        const bool fetched_skip = NSG == 1 && next_skip;
        if (NSG == 1 && had_dense) {
            const u32 v = __builtin_amdgcn_readfirstlane(sgb[L::MISC]);   // thread 0's verdict on the job (see job_mode)
            job_mode = v > job_mode ? v : job_mode;
        }
        if (act && !dense) {
            // Work items are blocks of 64 drivers of three kinds; kind k deals its blocks to the waves starting at a
            // different wave, so that the odd blocks of the kinds land on different waves.  The loops are bound by LDS
            // round trips, not by instruction issue: every trip takes two list entries, the next two are already in flight,
            // and the M bits of both are looked up together.
            const u32 nbF = (nF + 63) >> 6, nbR = (nRt + 63) >> 6;
            // row-relative dump cells (see EvLds)
            const u32 dumpN = L::o_dump(HN) + lane;
            const u32 dumpGF = (L::o_dump(HN) - L::o_gf(HN) + lane) << (BIG ? 1 : 0);
            const u32 dumpGR = (L::o_dump(HN) - L::o_gr(HN) + lane) << (BIG ? 1 : 0);
#ifndef EV_ABL_NOFR
            // ---- forward reads x reverse reads in [x, x + S]: ncc, mscc.cc ----
            // No predication: an event that misses is added to the lane's dump slot of the row; idle lanes carry x = 0
            // (every distance from it exceeds S: list positions start at BIAS), finished lanes park on the sentinel.
            for (u32 b = EV_PAIR_SLOT(wave); b < nbF; b += 4) {
                const u32 i = 64 * b + lane;
                const u32 ent = i < nF ? LF[i] : 0u;
                const u32 x = ent & EV_POS, xc = x + c;
                const u32 flm = HAS_M ? (ent >> 31) << 16 : 0u;   // mappable: the increment of the cc half
                if (HAS_M && flm) cntB += ev_mbit(MT, xc);   // Bf = sum of M[x + c]
                u32 r = (ent >> 17) & ((1u << EV_RANK_BITS) - 1u);
                u32 y0 = LR[r], y1 = LR[r + 1];
                u32 d0 = y0 - x, d1 = y1 - x;
                bool h0 = d0 <= S, h1 = d1 <= S;   // sorted: h1 implies h0
                // (a bottom-tested loop: one compare + one branch per trip; with the test in the middle the structurizer
                // spends six scalar instructions per trip on the exit flag)
                if (__ballot(h0)) do {
                    r = h1 ? r + 2 : nR;
                    y0 = LR[r];
                    y1 = LR[r + 1];
                    const u32 a0 = h0 ? d0 : dumpN, a1 = h1 ? d1 : dumpN;
                    if (HAS_M) {
                        const u32 q0 = xc - (h0 ? d0 : 0u), q1 = xc - (h1 ? d1 : 0u);
                        const u32 m0 = MT[q0 >> 5], m1 = MT[q1 >> 5];
                        // ncc + 1, cc + M[x + c - d] M[x]: one atomic for both
                        atomicAdd(&hN[a0], (DO_NCC ? 1u : 0u) | (((m0 >> (q0 & 31u)) << 16) & flm));
                        atomicAdd(&hN[a1], (DO_NCC ? 1u : 0u) | (((m1 >> (q1 & 31u)) << 16) & flm));
                    } else {
                        atomicAdd(&hN[a0], 1u);
                        atomicAdd(&hN[a1], 1u);
                    }
                    d0 = y0 - x;
                    d1 = y1 - x;
                    h0 = d0 <= S;
                    h1 = d1 <= S;
                } while (__ballot(h0));
            }
#endif
            EV_STAMP(6)
#ifndef EV_ABL_NOREV
            // ---- R0 = sum over the tile's reverse reads of M[p] M[p + c] ----
            if (HAS_M)
                for (u32 b = (wave + 2) & 3; b < nbR; b += 4) {
                    const u32 i = 64 * b + lane;
                    if (i < nRt) {
                        const u32 p = LR[i];
                        cnt0 += ev_mbit(MT, p) & ev_mbit(MT, p + c);
                    }
                }
#endif
            EV_STAMP(8)
            if (EV_PRIO_EDGES != EV_PRIO_EVENTS) __builtin_amdgcn_s_setprio(EV_PRIO_EDGES);
#ifndef EV_ABL_NOFE
            // ---- the edge events of mscc.fsum and mscc.rsum, driven from the EDGES (~60 per tile, one block of lanes) instead
            // of from the ~330 reads each: for an edge j (sign E[j]) the reads it meets are contiguous in the sorted lists,
            //   forward reads x in [j - c, j - c + S):      GF[x + c - j + 1] += E[j]                  (mappable x only)
            //   reverse reads p in [j, j + S)       (type A): d = p - j + 1,              GR[d] -= E[j] M[p + c - 2d]
            //   reverse reads p in [j - c, j - c + 2S) (type B): d = (p - (j - c) + 2) >> 1, GR[d] -= E[j] M[p - d + 1]
            // Every wave walks the same edges and takes every fourth read of a range (two per trip).  A range starts at the
            // first read of its 512-bit block (idxF / idxR, written by the emit): the reads below the range are misses like
            // any other, so there is no search.
            // (the edges an event can come from: all of the list, or -- see the emit -- all below the extra halo above the tile)
            const u32 nEv = (HAS_M && BIG && DO_MLEN && HI > EV_HI) ? __builtin_amdgcn_readfirstlane(sgb[L::MISC + 1]) : nE;
            if (HAS_M)
                for (u32 eb = 0; eb < nEv; eb += 64) {
                    const u32 i = eb + lane;
                    const bool in = i < nEv;
                    const u32 ee = in ? LE[i] : 0u;
                    const int32_t j = (int32_t)(ee & EV_POS);
                    const u32 sgn = (u32)(((int32_t)ee >> 31) | 1);   // E[j]: -1 falling, +1 rising
                    const int32_t tile_end = (int32_t)(BIAS + EV_TB);
                    // Which wave walks which range (round 4).  Until round 3 EVERY wave walked all three ranges of every edge and took
                    // every fourth read: three loop set-ups per wave and edge block, and runs of ~1.6 reads per lane whose longest
                    // decides the trips.  Now a range belongs to a wave -- wave 1 the forward reads, wave 2 type A, waves 3 and 0
                    // the halves of type B (twice the reads: 2 S) --: one set-up per wave, longer runs per lane (EV_EDGE_ROLES=0: the old deal).
#if EV_EDGE_ROLES
                    const bool roleF = wave == EV_ROLE_F, roleA = wave == EV_ROLE_A, roleB = wave == EV_ROLE_B0 || wave == EV_ROLE_B1;
                    const u32 woffF = 0, wstrF = 1, woffA = 0, wstrA = 1;
                    const u32 wstrB = EV_ROLE_B0 == EV_ROLE_B1 ? 1u : 2u, woffB = (EV_ROLE_B0 != EV_ROLE_B1 && wave == EV_ROLE_B1) ? 1u : 0u;
#else
                    const bool roleF = true, roleA = true, roleB = true;
                    const u32 woffF = wave, wstrF = 4, woffA = wave, wstrA = 4, woffB = wave, wstrB = 4;
#endif
                    // -- forward reads --
                    if (roleF) {
                        const int32_t lo = j - (int32_t)c;
                        const u32 span = in ? S : 0u;
                        int32_t bb = (lo - (int32_t)BIAS) >> 9;
                        bb = bb < 0 ? 0 : (bb > 128 ? 128 : bb);
                        u32 idx = (u32)idxF[bb] + woffF;
                        u32 e0 = LF[idx], e1 = LF[idx + wstrF];
                        u32 u0 = (e0 & EV_POS) - (u32)lo, u1 = (e1 & EV_POS) - (u32)lo;
                        bool more = (int32_t)u0 < (int32_t)span;   // this lane's entry is still below the end of its range
                        if (__ballot(more)) do {                   // (bottom-tested: see the pair loop)
                            const bool h0 = u0 < span && (e0 >> 31) != 0, h1 = u1 < span && (e1 >> 31) != 0;
                            ev_add_cell<BIG>(hGF, h0 ? u0 + 1 : dumpGF, sgn);
                            ev_add_cell<BIG>(hGF, h1 ? u1 + 1 : dumpGF, sgn);
                            idx = more ? idx + 2 * wstrF : idx;
                            e0 = LF[idx];
                            e1 = LF[idx + wstrF];
                            u0 = (e0 & EV_POS) - (u32)lo;
                            u1 = (e1 & EV_POS) - (u32)lo;
                            more = (int32_t)u0 < (int32_t)span;
                        } while (__ballot(more));
                    }
                    // -- reverse reads, type A and type B (the halo entries above the tile are not drivers: the span ends at the tile end) --
#pragma unroll
                    for (u32 kind = 0; kind < 2; kind++) {
                        if (!(kind == 0 ? roleA : roleB)) continue;   // (uniform)
                        const u32 woff = kind == 0 ? woffA : woffB, wstr = kind == 0 ? wstrA : wstrB;
                        const int32_t lo = kind == 0 ? j : j - (int32_t)c;
                        int32_t sp = tile_end - lo;
                        const int32_t full = kind == 0 ? (int32_t)S : 2 * (int32_t)S;
                        sp = sp < full ? sp : full;
                        const u32 span = (in && sp > 0) ? (u32)sp : 0u;
                        int32_t bb = (lo - (int32_t)BIAS) >> 9;
                        bb = bb < 0 ? 0 : (bb > 128 ? 128 : bb);
                        u32 idx = (u32)idxR[bb] + woff;
                        u32 p0 = LR[idx], p1 = LR[idx + wstr];
                        u32 u0 = p0 - (u32)lo, u1 = p1 - (u32)lo;
                        bool more = (int32_t)u0 < (int32_t)span;
                        if (__ballot(more)) do {
                            const bool h0 = u0 < span, h1 = u1 < span;
                            const u32 d0 = kind == 0 ? u0 + 1 : (u0 + 2) >> 1, d1 = kind == 0 ? u1 + 1 : (u1 + 2) >> 1;
                            const u32 q0 = h0 ? (kind == 0 ? p0 + c - 2 * d0 : p0 - d0 + 1) : BIAS;
                            const u32 q1 = h1 ? (kind == 0 ? p1 + c - 2 * d1 : p1 - d1 + 1) : BIAS;
                            const u32 m0 = MT[q0 >> 5], m1 = MT[q1 >> 5];
                            idx = more ? idx + 2 * wstr : idx;
                            p0 = LR[idx];
                            p1 = LR[idx + wstr];
                            // -E[j] M[q]
                            ev_add_cell<BIG>(hGR, h0 ? d0 : dumpGR, (0u - sgn) * ((m0 >> (q0 & 31u)) & 1u));
                            ev_add_cell<BIG>(hGR, h1 ? d1 : dumpGR, (0u - sgn) * ((m1 >> (q1 & 31u)) & 1u));
                            u0 = p0 - (u32)lo;
                            u1 = p1 - (u32)lo;
                            more = (int32_t)u0 < (int32_t)span;
                        } while (__ballot(more));
                    }
                }
#endif
        }
        {
            // ---- run edges of the tile x the edges within max_lag above them: the mappable-length autocorrelation (also
            // for a tile whose READS went to the window kernel) ----
            if (!BIG && DO_MLEN && do_edges) {
                const u32 dumpEE = L::o_dump(HN) - L::o_ee(HN) + lane;
                const u32 nEt = TE0 + TE1;   // the tile's own edges sit at list indices [TXb, TXb + nEt)
                for (u32 b = (wave + 1) & 3; 64 * b < nEt; b += 4) {
                    const u32 i = 64 * b + lane;
                    const bool in = i < nEt;
                    const u32 ent = in ? LE[TXb + i] : 0u;
                    const u32 pos = ent & EV_POS, hi = pos + max_lag;
                    u32 e = in ? TXb + i + 1 : nE;
                    u32 e0 = LE[e], e1 = LE[e + 1];
                    u32 p0 = e0 & EV_POS, p1 = e1 & EV_POS;
                    bool h0 = p0 <= hi, h1 = p1 <= hi;
                    if (__ballot(h0)) do {
                        // E[j] E[j + k] into the HIGH half of cell k of the GF row
                        atomicAdd(&hEE[h0 ? p0 - pos : dumpEE], (u32)(((int32_t)(e0 ^ ent) >> 31) | 1) << 16);
                        atomicAdd(&hEE[h1 ? p1 - pos : dumpEE], (u32)(((int32_t)(e1 ^ ent) >> 31) | 1) << 16);
                        e = h1 ? e + 2 : nE;
                        e0 = LE[e];
                        e1 = LE[e + 1];
                        p0 = e0 & EV_POS;
                        p1 = e1 & EV_POS;
                        h0 = p0 <= hi;
                        h1 = p1 <= hi;
                    } while (__ballot(h0));
                }
            }
        }
        if (EV_PRIO_EVENTS) __builtin_amdgcn_s_setprio(0);
        EV_STAMP(7)
        // listed reads since the last flush (uniform over the workgroup: xch was written before B1)
        if (NSG == 1) {
            if (!dense) {
                accF += nF;
                accR += nRt;
            }
            if (DO_MLEN && do_edges) accE += nE;
        } else {
#pragma unroll
            for (u32 k = 0; k < NSG; k++) {
                const u32 v = xch[k];
                accF += v & 0xffffu;
                accR += v >> 16;
                if (DO_MLEN) accE += xch[4 + k];
            }
            accF = __builtin_amdgcn_readfirstlane(accF);
            accR = __builtin_amdgcn_readfirstlane(accR);
            accE = __builtin_amdgcn_readfirstlane(accE);
        }
        const bool leaving = jn != ji || gnext >= g1;
        // another iteration could overflow a 16-bit cell (bounds in the comment of EvLds): flush now
        // (EE_LDS: a cell of the row of lags takes at most one pair per listed edge)
        const bool risk = HAS_M && (BIG ? (accF + NSG * EV_CAPF > 32767u || 3u * (accR + NSG * EV_CAPR) > 32767u ||
                                           (EE_LDS && accE + NSG * L::POOL > 32767u))
                                        : (accF + L::POOL > 32767u || (DO_MLEN && accE + EV_CAPE(BIG) > 32767u)));
        if (NSG == 1 && leaving && tid == 0) {
            // tiles flagged in this job: one atomic per workgroup and job, not per tile (47 k adds to one word serialise in L2
            // when every tile is dense)
            const u32 pc = xch[14], pa = xch[15];
            if (pc) atomicAdd(n_flagged, pc);
            if (pa) atomicAdd(n_flagged + 1, pa);
            xch[14] = 0;
            xch[15] = 0;
        }
        // BIG (round 4, late): the GR row reaches its bound four times as often as the others (three events per listed reverse read and
        // shift against one per forward read), and a flush of everything is 8 rows' worth of read-modify-write (config 5: 0.47 GB of
        // writes per step).  When GR alone is at risk, GR alone is flushed.
#ifndef EV_NO_GR_FLUSH
        const bool risk_gr_only = BIG && HAS_M && !leaving && risk && !(accF + NSG * EV_CAPF > 32767u) &&
                                  !(EE_LDS && accE + NSG * L::POOL > 32767u);
#else
        const bool risk_gr_only = false;
#endif
        if (risk_gr_only) {
            __syncthreads();
            u32 *seg = slab + (size_t)(blockIdx.x + ji) * EV_SEG_ROWS * HN;
            const bool add = seg_gr_written;
#pragma nounroll
            for (u32 k = gt; k < (HN >> 1); k += NT) {
                const u32 w = hGR[k];
                hGR[k] = 0;
                const int32_t lo16 = (int32_t)(short)(w & 0xffffu);
                const int32_t hi16 = (int32_t)(short)((w - (u32)lo16) >> 16);
                uint2 *dst = reinterpret_cast<uint2 *>(seg + (size_t)3 * HN + 2 * k);
                uint2 v = make_uint2((u32)lo16, (u32)hi16);
                if (add) {
                    const uint2 o = *dst;
                    v.x += o.x;
                    v.y += o.y;
                }
                *dst = v;
            }
            accR = 0;
            seg_gr_written = true;
        } else if (leaving || risk) {
            // histograms of this (workgroup, job) -> its slab segment (added to it from the second flush on); cleared
            __syncthreads();
            u32 *seg = slab + (size_t)(blockIdx.x + ji) * EV_SEG_ROWS * HN;
            const bool add = seg_written;
#pragma nounroll   // (unrolled, its index registers are hoisted out of the tile loop and spill)
            for (u32 i = gt; i < HN; i += NT) {
                const u32 w = hN[i];
                hN[i] = 0;
                if (HAS_M) {
                    const u32 lo16 = w & 0xffffu, hi16 = w >> 16;
                    seg[i] = add ? seg[i] + lo16 : lo16;
                    seg[2 * HN + i] = add ? seg[2 * HN + i] + hi16 : hi16;
                } else {
                    seg[i] = add ? seg[i] + w : w;
                }
            }
            if (HAS_M) {
                if (BIG) {
#pragma nounroll
                    for (u32 i = gt; i < HN; i += NT) {   // rows GF (i < HN / 2) and GR: two cells per dword
                        const u32 half = HN >> 1, row = i >= half ? 1u : 0u, k = i - row * half;
                        u32 *src = (row ? hGR : hGF) + k;
                        const u32 w = *src;
                        *src = 0;
                        const int32_t lo16 = (int32_t)(short)(w & 0xffffu);
                        const int32_t hi16 = (int32_t)(short)((w - (u32)lo16) >> 16);
                        uint2 *dst = reinterpret_cast<uint2 *>(seg + (size_t)(row ? 3 : 1) * HN + 2 * k);
                        uint2 v = make_uint2((u32)lo16, (u32)hi16);
                        if (row ? seg_gr_written : add) {
                            const uint2 o = *dst;
                            v.x += o.x;
                            v.y += o.y;
                        }
                        *dst = v;
                    }
                    if (DO_MLEN && EE_LDS) {
#pragma nounroll
                        for (u32 k = gt; k < HN / 2; k += NT) {   // the pairs of run edges by lag: two signed cells per dword
                            const u32 w = hEEb[k];
                            hEEb[k] = 0;
                            const int32_t lo16 = (int32_t)(short)(w & 0xffffu);
                            const int32_t hi16 = (int32_t)(short)((w - (u32)lo16) >> 16);
                            uint2 *dst = reinterpret_cast<uint2 *>(seg + (size_t)5 * HN + 2 * k);
                            uint2 v = make_uint2((u32)lo16, (u32)hi16);
                            if (add) {
                                const uint2 o = *dst;
                                v.x += o.x;
                                v.y += o.y;
                            }
                            *dst = v;
                        }
                    }
                } else {
                    // max_shift <= 1023 (round 4): the rows leave the workgroup TRANSFORMED -- the inclusive prefix sums of GF and GR,
                    // and for the edge pairs the recurrence's A(|c - d|) of THIS flush's (popcount(M), runs, EE) --, because all of
                    // them are linear: the sum over the workgroups of the transformed rows is the transform of the sums.  What
                    // remains for k_events_finish is a column sum, which any number of workgroups can share (one workgroup per row
                    // read chromosome 1's 103 segments at ~30 GB/s: 17 us).  Arithmetic modulo 2^32: every total is below 2^32
                    // (the launcher keeps vectors of 2^32 bits and more off this path).
                    // (a) the scalars of this flush: the recurrence starts from them
                    {
                        u32 v[4] = {cntB, cnt0, cntM, cntU};
#pragma unroll
                        for (u32 k = 0; k < 4; k++)
                            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
                        if ((gt & 63u) == 0) {
#pragma unroll
                            for (u32 k = 0; k < 4; k++)
                                if (k < 2 || DO_MLEN) atomicAdd(&xch[8 + 2 + k], v[k]);
                        }
                        cntB = cnt0 = cntM = cntU = 0;
                    }
                    __syncthreads();
                    const u32 a0 = xch[8 + 4], runs = xch[8 + 5];
                    // (b) cells 4 t .. 4 t + 3 of the three rows
                    const u32 k0 = 4 * gt;
                    const uint4 wgf = *reinterpret_cast<const uint4 *>(hGF + k0), wgr = *reinterpret_cast<const uint4 *>(hGF + 1024 + k0);
                    const u32 cgf[4] = {wgf.x, wgf.y, wgf.z, wgf.w}, cgr[4] = {wgr.x, wgr.y, wgr.z, wgr.w};
                    u32 pgf[4], pgr[4], dl[4], tGF = 0, tGR = 0, tX = 0;
#pragma unroll
                    for (u32 j = 0; j < 4; j++) {
                        const int32_t lo16 = (int32_t)(short)(cgf[j] & 0xffffu);
                        const int32_t ee = (int32_t)(short)((cgf[j] - (u32)lo16) >> 16);
                        tGF += (u32)lo16;
                        tGR += cgr[j];
                        tX += (k0 + j == 0) ? 0u - runs : 0u - (u32)ee;    // x(0) = -runs, x(k) = -EE(k)
                        pgf[j] = tGF;
                        pgr[j] = tGR;
                        dl[j] = tX;
                    }
                    u32 sGF = tGF, sGR = tGR, sX = tX;
                    wave_inclusive_scan3(sGF, sGR, sX);
                    if ((gt & 63u) == 63u) {
                        wt[0 + (gt >> 6)] = sGF;
                        wt[4 + (gt >> 6)] = sGR;
                        wt[8 + (gt >> 6)] = sX;
                    }
                    __syncthreads();
                    u32 bGF = sGF - tGF, bGR = sGR - tGR, bX = sX - tX;
                    for (u32 w = 0; w < (gt >> 6); w++) {
                        bGF += wt[w];
                        bGR += wt[4 + w];
                        bX += wt[8 + w];
                    }
                    u32 tD = 0, ex[4];
#pragma unroll
                    for (u32 j = 0; j < 4; j++) {
                        pgf[j] += bGF;
                        pgr[j] += bGR;
                        dl[j] += bX;          // Delta(k), inclusive
                        ex[j] = tD;           // ... and the sum of the Deltas below k inside the thread
                        tD += dl[j];
                    }
                    {
                        uint4 *d1 = reinterpret_cast<uint4 *>(seg + 1 * 1024 + k0), *d3 = reinterpret_cast<uint4 *>(seg + 3 * 1024 + k0);
                        uint4 v1 = make_uint4(pgf[0], pgf[1], pgf[2], pgf[3]), v3 = make_uint4(pgr[0], pgr[1], pgr[2], pgr[3]);
                        if (add) {
                            const uint4 o1 = *d1, o3 = *d3;
                            v1.x += o1.x; v1.y += o1.y; v1.z += o1.z; v1.w += o1.w;
                            v3.x += o3.x; v3.y += o3.y; v3.z += o3.z; v3.w += o3.w;
                        }
                        *d1 = v1;
                        *d3 = v3;
                    }
                    if (DO_MLEN) {
                        // A(k) = a0 + the sum of Delta(t) over t < k, then row 5 holds A(|c - d|) for the shifts d = 0 .. S
                        u32 sD = wave_inclusive_scan(tD);
                        __syncthreads();                    // (wt was read above)
                        if ((gt & 63u) == 63u) wt[12 + (gt >> 6)] = sD;
                        __syncthreads();
                        u32 bD = sD - tD;
                        for (u32 w = 0; w < (gt >> 6); w++) bD += wt[12 + w];
                        *reinterpret_cast<uint4 *>(hGF + k0) = make_uint4(a0 + bD + ex[0], a0 + bD + ex[1], a0 + bD + ex[2], a0 + bD + ex[3]);
                        __syncthreads();
#pragma unroll
                        for (u32 j = 0; j < 4; j++) {
                            const u32 d = k0 + j;
                            if (d <= S) {
                                const int32_t k = (int32_t)c - (int32_t)d;
                                const u32 av = hGF[k < 0 ? -k : k];
                                u32 *de = seg + (size_t)5 * 1024 + d;
                                *de = add ? *de + av : av;
                            }
                        }
                        __syncthreads();
                    }
                    *reinterpret_cast<uint4 *>(hGF + k0) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4 *>(hGF + 1024 + k0) = make_uint4(0, 0, 0, 0);
                    // (c) the scalar row of the segment and the chromosome's totals (k_events_finish reads six words per job)
                    if (gt < 6) {
                        const u32 v = xch[8 + gt];
                        u32 *ds = seg + 4 * HN + gt;
                        *ds = add ? *ds + v : v;
                        if (v) atomicAdd(EV_JOBSUM(jobstat) + 6 * ji + gt, (unsigned long long)v);
                    }
                    __syncthreads();
                    if (gt < 6) xch[8 + gt] = 0;
                }
            }
            accF = 0;
            accR = 0;
            accE = 0;
            seg_written = true;
            seg_gr_written = true;
            if (leaving && !(!BIG && HAS_M)) {   // (max_shift <= 1023 with a track: written with every flush, above)
                // scalars: |F|, |R|, Bf, R0, popcount(M), runs -> row 4 of the segment
                u32 v[6] = {0, 0, cntB, cnt0, cntM, cntU};
#pragma unroll
                for (u32 k = 2; k < 6; k++)
                    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
                if ((gt & 63u) == 0) {
#pragma unroll
                    for (u32 k = 2; k < 6; k++)
                        if (k < 4 || DO_MLEN) atomicAdd(&xch[8 + k], v[k]);
                }

                __syncthreads();
                if (gt < 6) {
                    const u32 v = xch[8 + gt];
                    seg[4 * HN + gt] = v;
                    // max_shift <= 1023: the chromosome's totals are added up here (k_events_finish reads six words per job
                    // instead of six words per segment)
                    if (!BIG && v) atomicAdd(EV_JOBSUM(jobstat) + 6 * ji + gt, (unsigned long long)v);
                    xch[8 + gt] = 0;
                }
                cntB = 0;
                cnt0 = 0;
                cntM = 0;
                cntU = 0;
            }
            if (leaving) {
                seg_written = false;
                seg_gr_written = false;
            }
        }
        if (jn != ji) {
            if (BIG && DO_MLEN && !EE_LDS) ev_zero_row5(slab, blockIdx.x + jn, HN, gt, NT);   // (the barrier at the top of the loop follows)
            job_mode = 0;
            next_skip = false;
            unreported = 0;
            if (had_dense && tid_ == 0) sgb[L::MISC] = 0;   // (read again after the next tile's B1 at the earliest)
            had_dense = false;
        }
        cur_skip = fetched_skip;
        ji = jn;
        cur_tile0 = pj.tile0;
        cur_flag0 = pj.flag0;
        g = gnext;
    }
#ifdef EV_STAMPS
    if ((gt & 63) == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(slab + (size_t)(gridDim.x + njobs) * EV_SEG_ROWS * HN);
        for (int i = 0; i < EV_NSTAMP; i++) dbg[((size_t)blockIdx.x * 4 * NSG + (gt >> 6)) * EV_NSTAMP + i] = stamp_acc[i];
    }
#endif
}

// ---- k_events_finish: everything between the event kernel and the (normally empty) window launches, in ONE launch ----
// Round 4.  Until round 3 the step behind k_cc_events was k_reduce_segments2 (16 us), k_plan_flagged (6 us), two window
// launches that return at once (2 x 6 us), k_events_tail (8 us) and a memset in front (5 us): 0.057 ms of a 0.49-ms step,
// and as much as a rank's whole share of the kernel at 8 GPUs.  (Folding the sums into the event kernel's epilogue -- the
// last workgroup of a chromosome -- was priced and not built: ONE workgroup reads another CU's fresh data at 30-70 GB/s,
// chromosome 1's 103 segments of 24 KB would take 35-60 us at the end of the kernel, where every workgroup finishes at the
// same time.)  The event kernel hands its rows over TRANSFORMED (prefix sums taken, recurrence run, see its flush), so that
// nothing but column sums is left -- and those are cut into EVF_CHUNKS x rows x chromosomes workgroups (a first version with
// one workgroup per row, prefix sums here, took 17 us: a workgroup reads fresh data of other CUs at ~30 GB/s).
// grid (EVF_CHUNKS, EVF_TASKS, njobs + 1), 1024 threads; block (chunk, task, job), 128 columns per chunk:
//   task 0  (chunk 0) the scalar row: |F|, |R|, popcount(M) with the fused mappable-length pass, path marker, zero fill
//   task 1  ncc           sum of row 0
//   task 2  mscc.ccbins   sum of row 2
//   task 3  mscc.fsum     Bf - sum of row 1 (prefix sums of GF)
//   task 4  mscc.rsum     R0 + sum of row 3 (prefix sums of GR)
//   task 5  mappable_len  sum of row 5 (A(|c - d|) per flush)
//   rows the batch does not produce are written as zeros by the blocks of their task.  Sums modulo 2^32 (all totals are
//   below: the launcher keeps vectors of 2^32 bits and more off this path).
// slice njobs of the grid: blocks 0 / 1 plan the two window launches (returns at once when nothing was flagged); the
// others clear the OTHER flag area of the context for the next call (see ev_flag_area).
#define EVF_TASKS 6u
#define EVF_CHUNKS 8u
struct EvFinishArgs {
    const u32 *slab;
    const unsigned long long *jobsum;   // [6 per job] |F|, |R|, Bf, R0, popcount(M), runs: added up by k_cc_events (EV_JOBSUM)
    u32 S, out_stride, has_m, do_ncc, fused_mlen, zero_mlen, keep_scalar2;
    uint4 *zero_area;     // the other flag area (16-byte units), cleared for the next call
    u32 zero_quads;
};

__global__ void __launch_bounds__(1024)
k_events_finish(const SpJobTable jobs, u32 njobs, const EvFinishArgs a, const PlanLaunch pcc, const PlanLaunch pac)
{
    __shared__ u32 acc[128];
    const u32 tid = threadIdx.x, chunk = blockIdx.x, task = blockIdx.y, job = blockIdx.z;
    if (job == njobs) {
        const u32 id = task * EVF_CHUNKS + chunk, nid = EVF_TASKS * EVF_CHUNKS;
        if (id == 0) plan_flagged(pcc);
        else if (id == 1) plan_flagged(pac);
        else
            for (u32 i = (id - 2) * 1024 + tid; i < a.zero_quads; i += (nid - 2) * 1024) a.zero_area[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    const SpJobDev &jb = jobs.j[job];
    const u32 S = a.S, n = S + 1;
    u64 *const out = jb.out;
    const size_t os = a.out_stride;
    const unsigned long long *js = a.jobsum + 6 * job;
    if (task == 0) {
        if (chunk != 0) return;
        u64 *dst = out + (size_t)PMX_ROW_SCALARS * os;
        for (u32 k = tid; k < a.out_stride; k += 1024) {
            if (k == 2 && a.keep_scalar2 && !a.fused_mlen) continue;   // (popcount(M): a pass of its own writes it)
            // (the path marker of a chromosome shared by several tile-range jobs -- ranks -- is written by the one that holds
            // its first tile, so that the shares add up to it)
            dst[k] = k < 2 ? js[k] : (k == 2 ? (a.fused_mlen ? js[4] : 0ull) : (k == 3 && jb.tile_first == 0 ? (u64)PMX_PATH_SPARSE : 0ull));
        }
        return;
    }
    const u32 dst_row = task == 1 ? PMX_ROW_NCC_CCBINS : task == 2 ? PMX_ROW_MSCC_CCBINS : task == 3 ? PMX_ROW_MSCC_FSUM
                        : task == 4 ? PMX_ROW_MSCC_RSUM : PMX_ROW_MLEN;
    const bool produced = task == 1 ? a.do_ncc != 0 : task == 5 ? a.fused_mlen != 0 : a.has_m != 0;
    const u32 col0 = 128 * chunk;
    u64 *dst = out + (size_t)dst_row * os;
    if (!produced) {
        // a row this batch leaves empty (the mappable-length row only if no pass of its own will write it)
        if (task != 5 || !a.has_m || a.zero_mlen)
            for (u32 k = col0 + tid; k < a.out_stride && k < col0 + 128; k += 1024) dst[k] = 0;
        if (chunk == EVF_CHUNKS - 1 && (task != 5 || !a.has_m || a.zero_mlen))
            for (u32 k = 128 * EVF_CHUNKS + tid; k < a.out_stride; k += 1024) dst[k] = 0;
        return;
    }
    if (col0 >= n) return;
    const u32 src_row = task == 1 ? 0u : task == 2 ? 2u : task == 3 ? 1u : task == 4 ? 3u : 5u;
    if (tid < 128) acc[tid] = 0;
    __syncthreads();
    // 32 lanes x 16 bytes cover the chunk's 128 columns of one segment row; the 32 groups of 32 lanes take every 32nd segment
    const u32 q = tid & 31u, g = tid >> 5;
    const size_t stride = (size_t)EV_SEG_ROWS * 1024;
    const u32 *p = a.slab + (size_t)job * stride + (size_t)src_row * 1024 + col0 + 4 * q;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (u32 w = jb.wg_first + g; w <= jb.wg_last; w += 32) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)w * stride);
        s0 += v.x;
        s1 += v.y;
        s2 += v.z;
        s3 += v.w;
    }
    atomicAdd(&acc[4 * q + 0], s0);
    atomicAdd(&acc[4 * q + 1], s1);
    atomicAdd(&acc[4 * q + 2], s2);
    atomicAdd(&acc[4 * q + 3], s3);
    __syncthreads();
    if (tid < 128 && col0 + tid < n) {
        const u32 t = acc[tid];
        dst[col0 + tid] = task == 3 ? (u64)(u32)((u32)js[2] - t) : task == 4 ? (u64)(u32)((u32)js[3] + t) : (u64)t;
    }
}

// The tail of the event pass, ONE launch per batch (grid: jobs x 4; y = 1 only with the mappable-length fusion, y = 2, 3 take
// the ncc / cc rows of the slow path and return at once when nothing was flagged):
//   y = 0: fsum[d] = Bf - sum_{t<=d} GF[t], rsum[d] = R0 + sum_{t<=d} GR[t] (k_reduce_segments left the signed sums of GF /
//          GR over the workgroups in rows MSCC_FSUM / MSCC_RSUM of the result block; Bf and R0 are summed from the slab
//          here), then -- only if the event kernel flagged dense tiles -- the sums of the cross-correlation window kernel
//          (its private slab) are added to the four rows and the two read counts;
//   y = 1: the same for the autocorrelation window kernel (P, N, popcount, runs in the per-job scratch), then the
//          recurrence of k_autocorr_finish.
// The additions are a slow path (one block sums every workgroup segment of its job) that costs nothing when nothing was
// flagged; it replaces two gated reduce launches and a finish launch of ~6 us each in every step.
struct EvTailPlan {
    u32 cc_first[SP_MAXJOBS], cc_last[SP_MAXJOBS];   // workgroup range of each job in the cc window launch
    u32 ac_first[SP_MAXJOBS], ac_last[SP_MAXJOBS];   // ... in the autocorrelation window launch
};

#define EV_TAIL_THREADS 1024u   // (the additions of the slow path are one block per chromosome: as many threads as shifts)
template <typename JT = SpJobTable>
__global__ void __launch_bounds__(EV_TAIL_THREADS)
k_events_tail(const u32 *__restrict__ slab, const JT jobs, const EvTailPlan plan, const u32 *__restrict__ slab_cc,
              const u32 *__restrict__ slab_ac, const u32 *__restrict__ n_flagged, u32 S, u32 out_stride, u32 has_m, u32 do_ncc,
              u32 max_lag, u32 lagcap, int32_t c, u32 fused, u32 rowlen, u32 slow_path, const u32 *__restrict__ plan_cc,
              const u32 *__restrict__ plan_ac, u32 prefix_done)
{
    __shared__ long long part[256];
    __shared__ long long tot[2];
    const u32 job = blockIdx.x, tid = threadIdx.x;
    const SpJobDev &jb = jobs.j[job];
    // blockIdx.y: 0 prefix sums (+ slow path: fsum, rsum), 1 autocorrelation, 2 / 3 slow path only: ncc + read counts / cc
    // (slow_path == 0: max_shift > 1023, the window kernel runs in shift chunks behind this kernel and a gated reduce adds its sums)
    const bool flagged = slow_path && n_flagged[blockIdx.y == 1 ? 1 : 0] != 0;   // [0]: tiles flagged for k_cc_sparse, [1]: for k_autocorr_edges
    if ((blockIdx.y >= 2 && !flagged) || (blockIdx.y == 1 && !fused)) return;
    // prefix_done (max_shift <= 1023, round 4): k_events_finish has taken the prefix sums and the recurrence already; only the
    // additions of the window kernels' sums remain here, and the whole launch returns at once when nothing was flagged
    if (prefix_done && !flagged) return;
    if (blockIdx.y == 1) {
        if (flagged) {
            u64 *P = jb.out2, *N = jb.out2 + lagcap, *scal = jb.out2 + 2 * (size_t)lagcap;
            const size_t stride = (size_t)AC_SEG_ROWS * 1024;
            // (the workgroups that wrote a segment for this job: entries [i0, i1) of the device-side plan's list of non-empty
            // ranges (k_plan_flagged) when there is one, else the host's static range)
            const u32 i0 = plan_ac ? plan_ac[PLAN_JOBWG + 2 * job] : plan.ac_first[job];
            const u32 i1 = plan_ac ? plan_ac[PLAN_JOBWG + 2 * job + 1] : plan.ac_last[job] + 1;
            // (prefix_done: the event pass's share of the row is written already -- k_events_finish --; the window kernel's
            // sums are put into the scratch on their own and their recurrence is ADDED to the row, mode 2)
            for (u32 k = tid; k <= max_lag; k += EV_TAIL_THREADS) {
                u64 sp = 0, sn = 0;
#pragma unroll 4
                for (u32 i = i0; i < i1; i++) {
                    const u32 w = plan_ac ? plan_ac[PLAN_LIST + i] : i;
                    const u32 *seg = slab_ac + (size_t)(w + job) * stride;
                    sp += seg[k];
                    sn += seg[1024 + k];
                }
                P[k] = prefix_done ? sp : P[k] + sp;
                N[k] = prefix_done ? sn : N[k] + sn;
            }
            if (tid < 2) {
                u64 sc = 0;
                for (u32 i = i0; i < i1; i++) {
                    const u32 w = plan_ac ? plan_ac[PLAN_LIST + i] : i;
                    sc += slab_ac[(size_t)(w + job) * stride + 2 * 1024 + tid];
                }
                scal[tid] = prefix_done ? sc : scal[tid] + sc;
            }
            __threadfence_block();
            __syncthreads();
        }
        autocorr_finish_job(jb, part, max_lag, lagcap, prefix_done ? 2u : 1u, c, S, out_stride, EV_TAIL_THREADS);
        return;
    }
    if (has_m && blockIdx.y == 0 && !prefix_done) {
        long long bf = 0, r0 = 0;
        for (u32 w = jb.wg_first + tid; w <= jb.wg_last; w += EV_TAIL_THREADS) {
            const u32 *sc = slab + ((size_t)(w + job) * EV_SEG_ROWS + 4) * rowlen;
            bf += sc[2];
            r0 += sc[3];
        }
        const long long eb = block_exclusive_offset(bf, part, tid);   // (ends with a barrier)
        if (tid == EV_TAIL_THREADS - 1) tot[0] = eb + bf;
        const long long e0 = block_exclusive_offset(r0, part, tid);
        if (tid == EV_TAIL_THREADS - 1) tot[1] = e0 + r0;
        __syncthreads();
        const long long Bf = tot[0], R0 = tot[1];
        const u32 seg = (S + 1 + EV_TAIL_THREADS - 1) / EV_TAIL_THREADS;
        const u32 k0 = tid * seg, k1 = (k0 + seg < S + 1) ? k0 + seg : S + 1;
#pragma unroll
        for (u32 row = 0; row < 2; row++) {
            long long *v = reinterpret_cast<long long *>(jb.out + (size_t)(row ? PMX_ROW_MSCC_RSUM : PMX_ROW_MSCC_FSUM) * out_stride);
            long long sum = 0;
            for (u32 k = k0; k < k1; k++) sum += v[k];
            long long run = block_exclusive_offset(sum, part, tid);
            for (u32 k = k0; k < k1; k++) {
                run += v[k];
                v[k] = row ? R0 + run : Bf - run;
            }
            __syncthreads();
        }
    }
    if (flagged) {
        // rows of the window kernel's segments: 0 ncc, 1 fsum, 2 cc, 3 rsum, 4 scalars (|F|, |R|)
        const u32 dst_row[4] = {PMX_ROW_NCC_CCBINS, PMX_ROW_MSCC_FSUM, PMX_ROW_MSCC_CCBINS, PMX_ROW_MSCC_RSUM};
        const size_t stride = (size_t)SP_SEG_ROWS * 1024;
        const u32 i0 = plan_cc ? plan_cc[PLAN_JOBWG + 2 * job] : plan.cc_first[job];       // (see the autocorrelation branch)
        const u32 i1 = plan_cc ? plan_cc[PLAN_JOBWG + 2 * job + 1] : plan.cc_last[job] + 1;
        for (u32 r = 0; r < 4; r++) {
            if (r == 0 ? !do_ncc : !has_m) continue;
            // the rows with prefix sums stay with the block that took them (y = 0); ncc -> y = 2, cc -> y = 3
            if (blockIdx.y != (r == 0 ? 2u : (r == 2 ? 3u : 0u))) continue;
            u64 *dst = jb.out + (size_t)dst_row[r] * out_stride;
            for (u32 d = tid; d <= S; d += EV_TAIL_THREADS) {
                u64 sum = 0;
#pragma unroll 4
                for (u32 i = i0; i < i1; i++) {
                    const u32 w = plan_cc ? plan_cc[PLAN_LIST + i] : i;
                    sum += slab_cc[(size_t)(w + job) * stride + r * 1024 + d];
                }
                dst[d] += sum;
            }
        }
        if (blockIdx.y == 2 && tid < 2) {
            u64 sc = 0;
            for (u32 i = i0; i < i1; i++) {
                const u32 w = plan_cc ? plan_cc[PLAN_LIST + i] : i;
                sc += slab_cc[(size_t)(w + job) * stride + 4 * 1024 + tid];
            }
            jb.out[(size_t)PMX_ROW_SCALARS * out_stride + tid] += sc;
        }
    }
}
