// Set-bit driven cross-correlation kernels (gfx950) for read-occupancy vectors, which are sparse by nature (at most
// one bit per position and strand; ~0.5 % density for 15 M reads / 3.1 Gbp).  k_cc_sparse below is the WINDOW kernel:
// every tile when max_shift > 1023, otherwise the tiles the event kernel (kernels_events.h, included further down)
// flags as too dense for its lists.
//
// Same outputs as the reference's per-shift loop (PyMaSC/core/bitarray/mscc.pyx:288-317), different algorithm:
// instead of sliding the whole R vector past F once per shift (N/64 words x (S+1) shifts), every SET BIT x of
// a driver vector contributes a whole (S+1)-bit WINDOW of the other vector(s):
//     ncc[d]        += R[x + d]                                   for x in F            (mscc.pyx:314)
//     mscc.fsum[d]  += M[x + c - d]                               for x in F & M, c=L-1 (mscc.pyx:300,303)
//     mscc.cc[d]    += R[x + d] & M[x + c - d]                    for x in F & M        (mscc.pyx:305)
//     mscc.rsum[d]  += M[p - d] & M[p + c - 2d]                   for p in R            (mscc.pyx:301,304)
// (D_d[j] = M[j] & M[j+c-d], mscc.pyx:291; the last line substitutes p = j + d.)  Windows are summed
// position-wise with bit-sliced carry-save counters: lane l of a slot owns shifts 32l..32l+31 as ONE 32-bit
// word per counter bit plane, so adding a window costs ~2 VALU ops (v_bitop3) per 32 shifts, not 32 adds.
//
// Work layout.  One launch covers a BATCH of chromosomes (jobs): the tiles (32 Kbit) of all jobs form one
// global sequence that is cut into equal contiguous ranges, one per persistent workgroup of 4 wavefronts.
// Per tile: the F / R / M words (+ halos of S bits) are fetched with 16-byte coalesced loads into REGISTERS
// one tile ahead (software pipeline); R and M go to LDS, M also as even/odd decimated copies (the stride-2
// window of rsum); the set bits of F and R are turned into 16-byte RECORDS in LDS lists (LDS atomic cursor;
// order is irrelevant for a sum) that already hold the LDS byte addresses and shift amounts of the windows,
// so the 32 lanes of a slot add only their lane offset; entries that must contribute nothing (padding,
// unmappable forward reads) point at a zeroed LDS region instead of being masked.  Each SLOT (G lanes,
// G = 2^k >= (S+1)/32; 64/G slots per wave) consumes its share of the records four at a time: 3 static
// carry-save adders + a binary-counter style carry insertion behind wave-uniform branches.
// Counters are hierarchical: 10 planes + parked carries in registers per slot -> (rarely) folded, combined
// across slots with bit-sliced adders and added into 24-plane workgroup accumulators in LDS -> converted to
// integers only when the workgroup leaves a job, into a private slab segment that k_reduce_segments sums.
// No global atomics anywhere.
#include "pmx_common.h"

#include <stdio.h>
#include <string.h>
#include <type_traits>
#include <vector>

#define SP_TB 32768u                 // driver bits per tile
#define SP_TBW 1024u                 // dwords per tile
#define SP_RHI 36u                   // R dwords staged above the tile (32 G / 32 + 2, rounded to 16 B)
#define SP_MLO 68u                   // M dwords staged below the tile (64 G + 64 bits, rounded to 16 B)
#define SP_MHI 36u                   // M dwords staged above the tile (read_len - 1 + 64 bits)
#define SP_MW (SP_MLO + SP_TBW + SP_MHI)   // 1128 dwords of M per tile
#define SP_HALO_M (SP_MLO * 32u)     // 2176 bits
#define SP_CAP 512u                  // records per list round
#define SP_NP 11                     // register counter: planes P[0..10] + two parked carries (Q2, Q3)
#define SP_NQ 11                     // planes of a folded per-slot number
#define SP_QLIMIT 511u               // quads a register counter may absorb between flushes (count < 2048)
#ifndef SP_QSOFT
#define SP_QSOFT 256u                // flush at a tile boundary once this many quads are pending
#endif
#define SP_NS 16                     // planes of a slot-combined number handed to the LDS accumulator
#define SP_NL 24                     // planes of the workgroup accumulator in LDS
#define SP_L2LIMIT (1u << 21)        // counts folded into the LDS accumulator before it is converted (< 2^24)
#define SP_MAXJOBS 32u               // jobs per launch (job table travels in the kernel arguments)
#define SP_SEG_ROWS 5u               // slab segment rows of 1024 u32: ncc, fsum, ccbins(mscc), rsum, scalars

// A job is one chromosome x one CHUNK of 1024 shifts.  Shifts d = d_off + d' (d' < 1024) are obtained by staging
// the WINDOW tiles from dword-shifted positions of the same vectors (d_off is a multiple of 1024 bits = 32 dwords):
// R windows from +off, M windows from -off, the decimated M copies from -2 off; drivers and flags stay unshifted.
struct SpJobDev {
    const u32 *F, *R, *M;
    u64 nbits;
    u32 tile0, ntiles;     // position in the global tile sequence
    u32 aligned16, wg_first, wg_last;
    u32 flags;             // bit 0: this job writes the scalar row / zero rows of its result block (chunk 0)
    u32 d_off, d_n;        // first shift and number of shifts of this chunk
    u64 *out;              // result block of the job
    u64 *out2;             // autocorrelation: per-job scratch (P, N, scalars, A)
    u32 flag0;             // index of the job's first tile in the dense-tile flag array
    u32 tile_first;        // event kernel only: the job is the tiles [tile_first, tile_first + ntiles) of its chromosome (a rank's
                           // share of it, pmx_cc_batch_ranges_dev; 0 with ntiles = all tiles otherwise)
};

struct SpJobTable {
    SpJobDev j[SP_MAXJOBS];
};
// The same table in DEVICE memory, any number of jobs (uploaded per launch on the launch's stream): the launches of
// max_shift > 1023 -- the event kernel over hundreds of chromosomes, the window kernel over (chromosome x shift chunk)
// jobs -- take all their jobs at once instead of 32 per launch (BASELINE config 5: 120 launches per step became 10).
struct SpJobTableRef {
    const SpJobDev *j;
};

// ji = the job of global tile g0 (tile0 is increasing): a linear scan of the table in the kernel arguments (at most 32
// jobs), a binary search of a device-side table.  A macro on purpose: with the table passed BY REFERENCE to a helper, the
// compiler dropped the scan of the kernel-argument table altogether (every workgroup started at job 0: a memory fault on
// the first launch with a workgroup range beyond job 0's tiles... and on launches with one job, harmlessly right).
#define FIRST_JOB(ji, JT, jobs, njobs, g0)                                         \
    u32 ji = 0;                                                                    \
    if constexpr (std::is_same<JT, SpJobTableRef>::value) {                        \
        u32 lo_ = 0, hi_ = (njobs) - 1; /* the last job with tile0 <= g0 */        \
        while (lo_ < hi_) {                                                        \
            const u32 mid_ = (lo_ + hi_ + 1) >> 1;                                 \
            if ((jobs).j[mid_].tile0 <= (g0)) lo_ = mid_;                          \
            else hi_ = mid_ - 1;                                                   \
        }                                                                          \
        ji = lo_;                                                                  \
    } else {                                                                       \
        while (ji + 1 < (njobs) && (jobs).j[ji + 1].tile0 <= (g0)) ji++;           \
    }

// The fields of the job whose tiles are being prefetched, held in (scalar) registers: the table lives in the kernel
// argument segment, and indexing it per tile cost a chain of four dependent scalar loads (~0.5 us) in every tile.
struct SpJobRegs {
    const u32 *F, *R, *M;
    u64 nbits;
    u32 tile0, tile_end, aligned16, d_off, flag0;
};

__device__ __forceinline__ void load_job(SpJobRegs &r, const SpJobDev &j)
{
    r.F = j.F;
    r.R = j.R;
    r.M = j.M;
    r.nbits = j.nbits;
    // (g - tile0 is the tile's index IN THE CHROMOSOME -- addresses, flags --: a job that starts at tile_first of its chromosome
    // shifts tile0 down by it, modulo 2^32; tile_end stays the end in the launch's tile sequence)
    r.tile0 = j.tile0 - j.tile_first;
    r.tile_end = j.tile0 + j.ntiles;
    r.aligned16 = j.aligned16;
    r.d_off = j.d_off;
    r.flag0 = j.flag0;
}

// ---- LDS layouts (dwords) ----------------------------------------------------------------------------------
template <bool HAS_M>
struct SpLds {
    static constexpr u32 ZERO = 0;                                   // 40 zero dwords: target of no-op windows
    static constexpr u32 R = 40;
    static constexpr u32 M = R + SP_TBW + SP_RHI;
    static constexpr u32 E = M + (HAS_M ? SP_MW : 0);
    static constexpr u32 O = E + (HAS_M ? SP_MW / 2 : 0);
    static constexpr u32 PLF = O + (HAS_M ? SP_MW / 2 : 0);         // positions of the set bits of F (+ flag)
    static constexpr u32 PLR = PLF + SP_CAP;                        // positions of the set bits of R
    static constexpr u32 REC = PLR + (HAS_M ? SP_CAP : 0);          // per-slot record staging (4 dwords each)
    static constexpr u32 ACC = REC + 4 * SP_CAP;                    // [counter][plane][32]
    static constexpr u32 NCOUNTERS = HAS_M ? 4 : 1;
    // fold staging [wave][SP_NS planes][32] ALIASES the record region: counter_to_lds runs at the top of a
    // (tile, round) pass, after B0 retired the previous pass's records and before this pass builds its own
    static constexpr u32 STAGE = REC;
    static constexpr u32 MISC = ACC + NCOUNTERS * SP_NL * 32;
    static constexpr u32 TOTAL = MISC + 16;
};
static_assert(4 * SP_NS * 32 <= 4 * SP_CAP, "the fold staging must fit in the record region it aliases");

struct Planes {
    u32 P[SP_NP];
    u32 Q2, Q3;   // parked carries of weight 4 / 8: valid iff bit 0 / bit 1 of the quad count
};

__device__ __forceinline__ void planes_zero(Planes &c)
{
#pragma unroll
    for (int l = 0; l < SP_NP; l++) c.P[l] = 0;
    c.Q2 = 0;
    c.Q3 = 0;
}

// carry-save adder on gfx950's 3-input boolean op: acc = a ^ b ^ acc, carry = maj(a, b, acc)
// (both truth tables are symmetric in their operands: 0x96 = odd parity, 0xE8 = majority)
__device__ __forceinline__ void csa(u32 &acc, u32 a, u32 b, u32 &carry)
{
    const u32 s = __builtin_amdgcn_bitop3_b32(acc, a, b, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(acc, a, b, 0xE8);
    acc = s;
}

// Add four 32-shift window words; quadcnt = quads this counter absorbed since its last flush (wave-uniform).
// A radix-16 Harley-Seal counter made resumable: every quad costs 3 carry-save adders and produces one carry of
// weight 4; carries of weight 4 and 8 are parked (Q2, Q3) until their partner arrives, i.e. quad 4k+1 pairs the
// weight-4 carries, quad 4k+3 pairs both levels and ripples the weight-16 carry through the upper planes with
// half adders.  One wave-uniform branch on (quadcnt & 3); ~11 VALU ops per quad on average, 13 registers.
__device__ __forceinline__ void add_quad(Planes &c, u32 w0, u32 w1, u32 w2, u32 w3, u32 quadcnt)
{
    u32 c1a, c1b, c2;
    csa(c.P[0], w0, w1, c1a);
    csa(c.P[0], w2, w3, c1b);
    csa(c.P[1], c1a, c1b, c2);
    if ((quadcnt & 1u) == 0u) {
        c.Q2 = c2;
    } else {
        u32 c3;
        csa(c.P[2], c.Q2, c2, c3);
        if ((quadcnt & 2u) == 0u) {
            c.Q3 = c3;
        } else {
            u32 cy;
            csa(c.P[3], c.Q3, c3, cy);
#pragma unroll
            for (int l = 4; l < SP_NP; l++) {   // half adders
                const u32 t = c.P[l] & cy;
                c.P[l] ^= cy;
                cy = t;
            }
        }
    }
}

// Two quads at once (8 windows): the two weight-4 carries pair with EACH OTHER (no parity at that level: a parked Q2
// stays parked), so only the weight-8 carry goes through the parked-carry branch -- one wave-uniform branch per 8
// windows and counter instead of three.  quadcnt as in add_quad; the caller advances it by 2.
__device__ __forceinline__ u32 quad_carry(Planes &c, u32 w0, u32 w1, u32 w2, u32 w3)
{
    u32 c1a, c1b, c2;
    csa(c.P[0], w0, w1, c1a);
    csa(c.P[0], w2, w3, c1b);
    csa(c.P[1], c1a, c1b, c2);
    return c2;
}

__device__ __forceinline__ void add_carry_pair(Planes &c, u32 c2a, u32 c2b, u32 quadcnt)
{
    u32 c3;
    csa(c.P[2], c2a, c2b, c3);
    if ((quadcnt & 2u) == 0u) {
        c.Q3 = c3;
    } else {
        u32 cy;
        csa(c.P[3], c.Q3, c3, cy);
#pragma unroll
        for (int l = 4; l < SP_NP; l++) {   // half adders
            const u32 t = c.P[l] & cy;
            c.P[l] ^= cy;
            cy = t;
        }
    }
}

// full adder on bit planes
__device__ __forceinline__ void fa(u32 a, u32 b, u32 cin, u32 &s, u32 &cout)
{
    s = __builtin_amdgcn_bitop3_b32(a, b, cin, 0x96);
    cout = __builtin_amdgcn_bitop3_b32(a, b, cin, 0xE8);
}

// Register counter -> workgroup accumulator in LDS (all 256 threads call this together):
//   1. fold the two parked carries into the planes (bit-sliced add inside the lane)              -> 11 planes
//   2. add the slots of a wave together with cross-lane bit-sliced adds (xor shuffles)            -> <= 15 planes
//   3. lanes 0..G-1 of every wave stage their number in LDS, barrier
//   4. lanes 0..G-1 of wave 0 add the 4 staged numbers into acc[plane][l] (24 planes held in registers), barrier
// The counter is cleared.  Inlined at exactly ONE place per kernel (the loops below are built around that): its
// temporaries (n[16], a[24]) are live where few other registers are, so it does not raise the kernel's allocation,
// whereas a non-inlined callee's registers ADD to the caller's on this target.
__device__ __forceinline__ void counter_to_lds(Planes &c, u32 quadcnt, u32 lgG, u32 tid, u32 *stage, u32 *acc,
                                               bool active = true)
{
    // active == false (wave-uniform; role-split builds): this wave does not carry the counter being folded: it stages
    // zeros and keeps `c`, which belongs to another accumulator
    u32 n[SP_NS];
    if (!active) {
#pragma unroll
        for (int k = 0; k < SP_NS; k++) n[k] = 0;
    } else {
        n[0] = c.P[0];
        n[1] = c.P[1];
        const u32 q2v = (quadcnt & 1u) ? c.Q2 : 0u;
        const u32 q3v = (quadcnt & 2u) ? c.Q3 : 0u;
        u32 cy;
        fa(c.P[2], q2v, 0u, n[2], cy);
        fa(c.P[3], q3v, cy, n[3], cy);
#pragma unroll
        for (int k = 4; k < SP_NP; k++) {
            n[k] = c.P[k] ^ cy;   // count < 2048: no carry out of plane 10
            cy &= c.P[k];
        }
#pragma unroll
        for (int k = SP_NP; k < SP_NS; k++) n[k] = 0;
        planes_zero(c);
    }
    for (u32 step = 1u << lgG; step < 64; step <<= 1) {   // lanes ^ G, ^ 2G, ...: the same shifts of other slots
        u32 carry = 0;
#pragma unroll
        for (int k = 0; k < SP_NS; k++) {
            const u32 o = __shfl_xor(n[k], step, 64);
            u32 s_, co;
            fa(n[k], o, carry, s_, co);
            n[k] = s_;
            carry = co;
        }
    }
    const u32 lane = tid & 63, wave = tid >> 6;
    const u32 G = 1u << lgG;
    if (lane < G) {
#pragma unroll
        for (int k = 0; k < SP_NS; k++) stage[(wave * SP_NS + k) * 32 + lane] = n[k];
    }
    __syncthreads();
    if (tid < G) {
        u32 a[SP_NL];
#pragma unroll
        for (int k = 0; k < SP_NL; k++) a[k] = acc[k * 32 + tid];
#pragma unroll 1
        for (u32 w = 0; w < 4; w++) {
            u32 carry = 0;
#pragma unroll
            for (int k = 0; k < SP_NL; k++) {
                const u32 o = k < SP_NS ? stage[(w * SP_NS + k) * 32 + tid] : 0u;
                u32 s_, co;
                fa(a[k], o, carry, s_, co);
                a[k] = s_;
                carry = co;
            }
        }
#pragma unroll
        for (int k = 0; k < SP_NL; k++) acc[k * 32 + tid] = a[k];
    }
    __syncthreads();
}

// LDS accumulators (ncounters x 24 planes x 32 words) -> integers added into rows of a slab segment; cleared.
// `seg_written`: the segment already holds an earlier partial conversion of this job (same thread owns d).
// `rev_mask`: counters kept in bit-reversed order inside each 32-shift word (bit i <-> shift 32 l + 31 - i).
__device__ __forceinline__ void acc_to_segment(u32 *acc, u32 ncounters, u32 rev_mask, u32 *__restrict__ seg, bool seg_written,
                                               u32 tid)
{
    __syncthreads();
#pragma unroll 1
    for (u32 q = 0; q < ncounters; q++) {
        const u32 *a = acc + q * SP_NL * 32;
        u32 *row = seg + q * 1024;
        const u32 flip = ((rev_mask >> q) & 1u) ? 31u : 0u;
#pragma unroll 1
        for (u32 d = tid; d < 1024; d += 256) {
            const u32 l = d >> 5, i = (d & 31) ^ flip;
            u32 v = seg_written ? row[d] : 0u;
#pragma unroll
            for (int k = 0; k < SP_NL; k++) v += ((a[k * 32 + l] >> i) & 1u) << k;
            row[d] = v;
        }
    }
    __syncthreads();
    for (u32 i = tid; i < ncounters * SP_NL * 32; i += 256) acc[i] = 0;
}

// ---- tile staging --------------------------------------------------------------------------------------

// dword j of a bit-vector with everything outside [0, nbits) read as zero
__device__ __forceinline__ u32 ld_dword_guarded(const u32 *__restrict__ p, int64_t j, uint64_t nbits)
{
    if (j < 0) return 0;
    const uint64_t bit0 = (uint64_t)j * 32;
    if (bit0 >= nbits) return 0;
    u32 w = p[j];
    const uint64_t left = nbits - bit0;
    if (left < 32) w &= (1u << left) - 1u;
    return w;
}

template <bool GUARD>
__device__ __forceinline__ uint4 ld_quad(const u32 *__restrict__ p, int64_t j, uint64_t nbits)
{
    if (!GUARD) return *reinterpret_cast<const uint4 *>(p + j);
    uint4 v;
    v.x = ld_dword_guarded(p, j, nbits);
    v.y = ld_dword_guarded(p, j + 1, nbits);
    v.z = ld_dword_guarded(p, j + 2, nbits);
    v.w = ld_dword_guarded(p, j + 3, nbits);
    return v;
}

// Behind the event / pair kernel: the next tile at or after `g` (below `lim`) whose flag is set, or lim.  64 flags per trip
// (one byte per lane + a ballot; every wave of the workgroup runs it and gets the same answer): with the flagged tiles in
// equal shares (k_plan_flagged) a range can hold thousands of unflagged tiles, and one dependent byte load per tile was
// 0.5 us each.
__device__ __forceinline__ u32 next_flagged_tile(const unsigned char *__restrict__ flags, u32 g, u32 lim)
{
    const u32 lane = threadIdx.x & 63u;
    while (g < lim) {
        const u32 t = g + lane;
        const bool set = t < lim && flags[t] != 0;
        const unsigned long long b = __ballot(set);
        if (b) return g + (u32)__builtin_ctzll(b);
        g += 64;
    }
    return lim;
}

struct TileRegs {
    uint4 f, r, m, h;   // main F / R / M quads of this thread + one halo quad (threads 0..34)
};
struct TileRegsX {      // shift chunks > 0 only: window tiles come from shifted positions
    uint4 rw, mw, md;   // main quads of the R window tile, the M window tile, the M tile to decimate
    uint4 hw, hd;       // halo quads of the window tiles / of the tile to decimate
};

// thread t: main dwords 4t..4t+3 of each vector; halo quads: t in [0,17) M below, [17,26) M above, [26,35) R above
template <bool HAS_M, bool CH, bool GUARD>
__device__ __forceinline__ void tile_fetch(TileRegs &tr, TileRegsX &tx, const u32 *__restrict__ F, const u32 *__restrict__ R,
                                           const u32 *__restrict__ M, int64_t d0, int64_t off, uint64_t nbits, u32 tid)
{
    const int64_t j = d0 + 4 * (int64_t)tid;
    tr.f = ld_quad<GUARD>(F, j, nbits);
    tr.r = ld_quad<GUARD>(R, j, nbits);
    if (HAS_M) tr.m = ld_quad<GUARD>(M, j, nbits);
    tr.h = make_uint4(0, 0, 0, 0);
    if (!CH) {
        if (HAS_M && tid < 17)
            tr.h = ld_quad<GUARD>(M, d0 - (int64_t)SP_MLO + 4 * (int64_t)tid, nbits);
        else if (HAS_M && tid < 26)
            tr.h = ld_quad<GUARD>(M, d0 + SP_TBW + 4 * (int64_t)(tid - 17), nbits);
        else if (tid >= 26 && tid < 35)
            tr.h = ld_quad<GUARD>(R, d0 + SP_TBW + 4 * (int64_t)(tid - 26), nbits);
    } else {
        tx.rw = ld_quad<GUARD>(R, j + off, nbits);
        tx.hw = make_uint4(0, 0, 0, 0);
        tx.hd = make_uint4(0, 0, 0, 0);
        if (HAS_M) {
            tx.mw = ld_quad<GUARD>(M, j - off, nbits);
            tx.md = ld_quad<GUARD>(M, j - 2 * off, nbits);
            if (tid < 17) {
                tx.hw = ld_quad<GUARD>(M, d0 - off - (int64_t)SP_MLO + 4 * (int64_t)tid, nbits);
                tx.hd = ld_quad<GUARD>(M, d0 - 2 * off - (int64_t)SP_MLO + 4 * (int64_t)tid, nbits);
            } else if (tid < 26) {
                tx.hw = ld_quad<GUARD>(M, d0 - off + SP_TBW + 4 * (int64_t)(tid - 17), nbits);
                tx.hd = ld_quad<GUARD>(M, d0 - 2 * off + SP_TBW + 4 * (int64_t)(tid - 17), nbits);
            }
        }
        if (tid >= 26 && tid < 35) tx.hw = ld_quad<GUARD>(R, d0 + off + SP_TBW + 4 * (int64_t)(tid - 26), nbits);
    }
}

template <bool HAS_M, bool CH>
__device__ __forceinline__ void tile_fetch_job(TileRegs &tr, TileRegsX &tx, const SpJobRegs &jb, u32 local_tile, u32 tid,
                                               const unsigned char *__restrict__ flags = nullptr)
{
    // behind the event kernel only the tiles it flagged as dense are processed: the others read as empty
    if (flags && !flags[jb.flag0 + local_tile]) {
        tr.f = make_uint4(0, 0, 0, 0);
        tr.r = tr.f;
        tr.m = tr.f;
        tr.h = tr.f;
        if (CH) {
            tx.rw = tr.f;
            tx.mw = tr.f;
            tx.md = tr.f;
            tx.hw = tr.f;
            tx.hd = tr.f;
        }
        return;
    }
    // edge tiles (or unaligned vectors) take the guarded loader
    const int64_t d0 = (int64_t)local_tile * SP_TBW;
    const int64_t off = CH ? (int64_t)(jb.d_off / 32) : 0;
    const int64_t lo = d0 - 2 * off - (int64_t)SP_MLO;                        // lowest dword touched
    const uint64_t hi = (uint64_t)(d0 + off) + SP_TBW + SP_MHI;                // one past the highest
    const bool interior = jb.aligned16 && lo >= 0 && hi + 2 <= jb.nbits / 32;
    if (interior)
        tile_fetch<HAS_M, CH, false>(tr, tx, jb.F, jb.R, jb.M, d0, off, jb.nbits, tid);
    else
        tile_fetch<HAS_M, CH, true>(tr, tx, jb.F, jb.R, jb.M, d0, off, jb.nbits, tid);
}

// bit unshuffle: even bits of x to the low half, odd bits to the high half (4 swap steps, 2 bitop3 each)
__device__ __forceinline__ u32 unshuffle32(u32 x)
{
    u32 t;
    t = (x ^ (x >> 1)) & 0x22222222u; x = x ^ t ^ (t << 1);
    t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x = x ^ t ^ (t << 2);
    t = (x ^ (x >> 4)) & 0x00f000f0u; x = x ^ t ^ (t << 4);
    t = (x ^ (x >> 8)) & 0x0000ff00u; x = x ^ t ^ (t << 8);
    return x;
}

// even / odd decimated copies of four consecutive M dwords: sE[off/2 .. +1], sO[off/2 .. +1]
__device__ __forceinline__ void decimate_quad(const uint4 m, u32 *sE, u32 *sO, u32 dword_off)
{
    const u32 x0 = unshuffle32(m.x), x1 = unshuffle32(m.y), x2 = unshuffle32(m.z), x3 = unshuffle32(m.w);
    const u32 e0 = (x0 & 0xffffu) | (x1 << 16), e1 = (x2 & 0xffffu) | (x3 << 16);
    const u32 o0 = (x0 >> 16) | (x1 & 0xffff0000u), o1 = (x2 >> 16) | (x3 & 0xffff0000u);
    *reinterpret_cast<uint2 *>(sE + dword_off / 2) = make_uint2(e0, e1);
    *reinterpret_cast<uint2 *>(sO + dword_off / 2) = make_uint2(o0, o1);
}

template <bool HAS_M, bool CH>
__device__ __forceinline__ void tile_store(const TileRegs &tr, const TileRegsX &tx, u32 *lds, u32 tid)
{
    typedef SpLds<HAS_M> L;
    const uint4 rq = CH ? tx.rw : tr.r;     // R window tile
    const uint4 mq = CH ? tx.mw : tr.m;     // M window tile
    const uint4 dq = CH ? tx.md : tr.m;     // M tile that is decimated
    const uint4 hq = CH ? tx.hw : tr.h;     // halo of the window tiles
    const uint4 hdq = CH ? tx.hd : tr.h;    // halo of the decimated tile
    reinterpret_cast<uint4 *>(lds + L::R)[tid] = rq;
    if (HAS_M) {
        reinterpret_cast<uint4 *>(lds + L::M + SP_MLO)[tid] = mq;
#ifndef SP_ABL_NODEC
        decimate_quad(dq, lds + L::E, lds + L::O, SP_MLO + 4 * tid);
#endif
        if (tid < 17) {
            reinterpret_cast<uint4 *>(lds + L::M)[tid] = hq;
            decimate_quad(hdq, lds + L::E, lds + L::O, 4 * tid);
        } else if (tid < 26) {
            reinterpret_cast<uint4 *>(lds + L::M + SP_MLO + SP_TBW)[tid - 17] = hq;
            decimate_quad(hdq, lds + L::E, lds + L::O, SP_MLO + SP_TBW + 4 * (tid - 17));
        }
    }
    if (tid >= 26 && tid < 35) reinterpret_cast<uint4 *>(lds + L::R + SP_TBW)[tid - 26] = hq;
}

// ---- set-bit records ---------------------------------------------------------------------------------------

struct SlotGeom {
    u32 total_slots, slot, l4, quad_span, lg_span, lg_region;   // quad_span = 4 * total_slots = 2^lg_span records
};

__device__ __forceinline__ SlotGeom slot_geom(u32 lgG, u32 tid)
{
    SlotGeom g;
    const u32 lane = tid & 63, wave = tid >> 6;
    const u32 spw = 64u >> lgG;
    g.total_slots = 4 * spw;
    g.slot = wave * spw + (lane >> lgG);
    g.l4 = 4u * (lane & ((1u << lgG) - 1));
    g.quad_span = g.total_slots * 4;
    g.lg_span = 10 - lgG;                 // 4 * 4 * (64 >> lgG)
    g.lg_region = lgG + 1;                // records per slot region: SP_CAP / total_slots = 512 / (256 >> lgG)
    return g;
}

// reserves list slots for this thread's set bits with one LDS atomic (order across threads is irrelevant)
__device__ __forceinline__ u32 emit_reserve(const uint4 w, u32 *cursor)
{
    const u32 n = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
    return n ? atomicAdd(cursor, n) : 0u;
}

// Stage 1 of the compaction (phase A, divergent loops, so the body is kept minimal): position of every set bit,
// with one flag bit (forward read mappable / falling edge), into an LDS list.
// CHECKED = false: the caller knows that all of this thread's entries fit the list (id + popcount <= SP_CAP): the
// capacity test leaves the loop body
template <bool WIDE = false, bool CHECKED = true>
__device__ __forceinline__ void emit_positions(const uint4 w, const uint4 flag_bits, u32 idx0, u32 round_lo, u32 *list, u32 tid,
                                               u32 base = 0)
{
    u32 id = idx0 - round_lo;   // unsigned: entries before the round wrap to huge values
    if (!WIDE) {   // one loop per dword
    const u32 ws[4] = {w.x, w.y, w.z, w.w};
    const u32 fs[4] = {flag_bits.x, flag_bits.y, flag_bits.z, flag_bits.w};
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        u32 ww = ws[k];
        while (ww) {
            const u32 b = __builtin_ctz(ww);
            ww &= ww - 1;
            if (!CHECKED || id < SP_CAP) list[id] = (base + 128u * tid + 32u * k + b) | (((fs[k] >> b) & 1u) << 16);
            id++;
        }
    }
    } else {
    // the quad is walked as two 64-bit words: every loop costs its ~25 instructions of control flow per trip whether or
    // not one of the 64 lanes has a bit, and at read densities a 64-bit word rarely needs more trips than a dword.
    // Measured on config 4: NCC+MSCC -5.3 % (0.936 -> 0.886 ms); the NCC-only instantiation (80 VGPRs, 6 waves) +7 %,
    // so it keeps the dword loops.
    const u64 ws[2] = {(u64)w.x | ((u64)w.y << 32), (u64)w.z | ((u64)w.w << 32)};
    const u64 fs[2] = {(u64)flag_bits.x | ((u64)flag_bits.y << 32), (u64)flag_bits.z | ((u64)flag_bits.w << 32)};
#pragma unroll
    for (u32 k = 0; k < 2; k++) {
        u64 ww = ws[k];
        while (ww) {
            const u32 b = (u32)__builtin_ctzll(ww);
            ww &= ww - 1;
            if (!CHECKED || id < SP_CAP) list[id] = (base + 128u * tid + 64u * k + b) | ((u32)((fs[k] >> b) & 1ull) << 16);
            id++;
        }
    }
    }
}

// round 0 of a tile: threads whose entries all fit (every thread, unless a vector has more than SP_CAP set bits in the
// tile) take the loop without the capacity test
template <bool WIDE>
__device__ __forceinline__ void emit_positions_round0(const uint4 w, const uint4 flag_bits, u32 idx0, u32 *list, u32 tid)
{
#ifdef SP_EMIT_CHECKED
    emit_positions<WIDE, true>(w, flag_bits, idx0, 0, list, tid);
#else
    const u32 n = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
    if (idx0 + n <= SP_CAP)
        emit_positions<WIDE, false>(w, flag_bits, idx0, 0, list, tid);
    else
        emit_positions<WIDE, true>(w, flag_bits, idx0, 0, list, tid);
#endif
}

// Stage 2 (after B1, no divergence): every slot turns ITS share of the positions into 16-byte records in its own
// staging region (same wave writes then reads: LDS keeps a wave's accesses in order, so no barrier).  A record
// holds the LDS byte addresses and shift words of the entry's windows; positions beyond the list end, and
// unmappable forward reads, get the address of the zero region instead of a mask.
//   forward: {shift word of the R window, R window address (lane adds 4l), M window address (lane subtracts 4l), M shift}
template <bool HAS_M>
__device__ __forceinline__ void build_forward_records(u32 *lds, uint4 *recs, u32 qbase, u32 qstride, u32 lane_in_slot,
                                                      u32 G, u32 nq, u32 n, int32_t c)
{
    typedef SpLds<HAS_M> L;
    const u32 *pl = lds + L::PLF;
    for (u32 j = lane_in_slot; j < 4 * nq; j += G) {
        uint4 rec = make_uint4(0u, L::ZERO * 4u, (L::ZERO + 32u) * 4u, 0u);
        const u32 idx = 4u * (qbase + (j >> 2) * qstride) + (j & 3u);   // the slot's k-th quad is quad qbase + k qstride
        if (idx < n) {
            const u32 e = pl[idx];
            const u32 pos = e & 0x7fffu;
            rec.x = pos;
            rec.y = (L::R + (pos >> 5)) * 4u;
            if (HAS_M) {
                const u32 a0 = pos + (u32)c + SP_HALO_M - 31u;
                rec.w = a0;
                if (e >> 16) rec.z = (L::M + (a0 >> 5)) * 4u;
            }
        }
        recs[j] = rec;
    }
}

//   reverse: {shift word of M[p-d], its address (lane subtracts 4l), address of the decimated copy holding M[p+c-2d]
//             (lane subtracts 4l), its shift word}
__device__ __forceinline__ void build_reverse_records(u32 *lds, uint4 *recs, u32 qbase, u32 qstride, u32 lane_in_slot, u32 G,
                                                      u32 nq, u32 n, int32_t c)
{
    typedef SpLds<true> L;
    const u32 *pl = lds + L::PLR;
    for (u32 j = lane_in_slot; j < 4 * nq; j += G) {
        uint4 rec = make_uint4(0u, (L::ZERO + 32u) * 4u, (L::ZERO + 32u) * 4u, 0u);
        const u32 idx = 4u * (qbase + (j >> 2) * qstride) + (j & 3u);
        if (idx < n) {
            const u32 p = pl[idx] & 0x7fffu;
            const u32 a1 = p + SP_HALO_M - 31u;
            const u32 bb = p + SP_HALO_M + (u32)c;
            const u32 a2 = (bb >> 1) - 31u;
            rec.x = a1;
            rec.y = (L::M + (a1 >> 5)) * 4u;
            rec.z = (((bb & 1u) ? L::O : L::E) + (a2 >> 5)) * 4u;
            rec.w = a2;
        }
        recs[j] = rec;
    }
}

__device__ __forceinline__ u32 lds_window(const u32 *lds, u32 byte_addr, u32 shift_word)
{
    const u32 *p = reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(lds) + byte_addr);
    return __builtin_amdgcn_alignbit(p[1], p[0], shift_word);
}

// ---- the cross-correlation kernel ---------------------------------------------------------------------------
#ifdef SP_STAMPS   // diagnostic build: where does a tile's time go (never defined in the shipped library)
#define SP_NSTAMP 12
#define SP_STAMP(i)                                                                                    \
    {                                                                                                  \
        unsigned long long t_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        stamp_acc[i] += t_ - stamp_last;                                                               \
        stamp_last = t_;                                                                               \
    }
#else
#define SP_STAMP(i)
#endif

#ifndef SP_WAVES
#define SP_WAVES 3
#endif
#ifndef SP_WAVES_CH
#define SP_WAVES_CH 3              // shift-chunked instantiations (max_shift > 1023) with / without mappability; measured on
#endif                             // config 5 (10 Gbp, max_shift 5000): 2 -> 3 waves 21.5 -> 16.9 ms; NCC-only 2: 9.05,
#ifndef SP_WAVES_CH_NCC            // 4: 5.70, 5: 5.49, 6: 5.19, 7: 5.82 ms
#define SP_WAVES_CH_NCC 6
#endif
#ifndef SP_WAVES_NCC
#define SP_WAVES_NCC 6             // waves per SIMD of the NCC-only instantiation (1 counter; measured 4: 0.334, 5: 0.309,
                                   // 6: 0.300, 7: 0.311, 8: 0.473 ms on the benchmark genome)
#endif
// s_setprio per phase.  Any phase of the tile loop raised above the bookkeeping / barrier code (priority 0) is worth
// 5-6 % on config 4 (k_cc_sparse alone, same box: 0.851 -> 0.799 ms with staging 3 / record loops 2; 0.80-0.81 for every
// other assignment tried, including staging only or record loops only): with equal priorities the SIMD arbitrates by
// age, the waves of the oldest workgroup win every cycle they can issue, and the co-resident workgroups move in lockstep
// through the same phases; with phase priorities a wave that is ready to do tile work is never behind one that is
// spinning towards a barrier.
#ifndef SP_PRIO_STAGE
#define SP_PRIO_STAGE 3
#endif
#ifndef SP_PRIO_PROCESS
#define SP_PRIO_PROCESS 2
#endif
#ifndef SP_ROLES
#define SP_ROLES 0                 // 1: waves 0-1 carry (ncc, mscc.ccbins), waves 2-3 (mscc.fsum, mscc.rsum) -- A/B build
#endif
// The body of k_cc_sparse as a device function (lds: SpLds<HAS_M>::TOTAL dwords; bx: the workgroup's index in ITS launch
// plan), so that the launch behind the event pass can carry it next to the autocorrelation window kernel's body in ONE grid
// (k_windows_flagged, round 4); k_cc_sparse itself is the wrapper below the body.
template <bool HAS_M, bool DO_NCC, bool CH, typename JT>
__device__ __forceinline__ void cc_sparse_body(u32 *const lds, const u32 bx, const JT &jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg,
                                               int32_t c, u32 lgG, u32 *__restrict__ slab, const unsigned char *__restrict__ tile_flags,
                                               const u32 *__restrict__ plan, u32 add_stride)
{
    typedef SpLds<HAS_M> L;
    u32 *const cursor = lds + L::MISC;   // [0] = F records, [1] = R records of the current tile
    u32 *const acc = lds + L::ACC;
    u32 *const stage = lds + L::STAGE;

    const u32 tid = threadIdx.x;
    const SlotGeom sg = slot_geom(lgG, tid);

    // tile range of this workgroup: equal shares of the tile sequence, or -- behind the event kernel -- the range
    // k_plan_flagged cut for it (equal shares of the FLAGGED tiles, see there)
    const u32 g0 = plan ? plan[2 * bx] : bx * tiles_per_wg;
    const u32 g1 = plan ? plan[2 * bx + 1] : (g0 + tiles_per_wg < total_tiles ? g0 + tiles_per_wg : total_tiles);
    if (g0 >= g1) return;

    for (u32 i = tid; i < 40; i += 256) lds[L::ZERO + i] = 0;
    for (u32 i = tid; i < L::NCOUNTERS * SP_NL * 32; i += 256) acc[i] = 0;

    constexpr bool ROLES = SP_ROLES && HAS_M && DO_NCC && !CH;
#ifdef SP_NOPAIRS
    constexpr bool QUAD_PAIRS = false;
#else
    constexpr bool QUAD_PAIRS = !HAS_M;
#endif
    constexpr u32 RCAP = ROLES ? SP_CAP / 2 : SP_CAP;   // records of one list a round can take (a role has half the slots)
    Planes cN, cF, cC, cR;
    planes_zero(cN);
    planes_zero(cF);
    planes_zero(cC);
    planes_zero(cR);
    Planes c0, c1;   // ROLES: role A (waves 0-1): c0 = ncc, c1 = mscc.ccbins; role B (waves 2-3): c0 = mscc.fsum, c1 = mscc.rsum
    planes_zero(c0);
    planes_zero(c1);
    u32 qF = 0, qR = 0;           // upper bound of the quads in the register counters (workgroup-uniform: fold decisions)
    u32 qFw = 0, qRw = 0;         // quads in THIS wave's register counters (wave-uniform: carry parity)
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 q2 = 0;                   // upper bound of the counts folded into the LDS accumulators since conversion
    u32 totF = 0, totR = 0;       // set bits of the current job seen by this workgroup (uniform)
    u32 cntR_thread = 0;          // NCC-only mode: popcount of R accumulated per thread
    bool seg_written = false;     // this (workgroup, job) slab segment already holds a partial conversion

    // job of the first tile
    FIRST_JOB(ji, JT, jobs, njobs, g0)

#ifdef SP_STAMPS
    unsigned long long stamp_acc[SP_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    TileRegs tr;
    TileRegsX tx;
    SpJobRegs pj;   // job of the tile being prefetched (index jn)
    load_job(pj, jobs.j[ji]);
    tile_fetch_job<HAS_M, CH>(tr, tx, pj, g0 - pj.tile0, tid, tile_flags);
    if (tid < 4) cursor[tid] = 0;   // [2 par + 0] = F records, [2 par + 1] = R records; parity alternates per tile
    u32 par = 0;
    const u32 G = 1u << lgG;
    const u32 lane_in_slot = sg.l4 >> 2;

    // The loop runs over (tile, round) pairs plus one final drain pass, so that the fold / convert code below is
    // inlined exactly once (one round per tile unless a vector has more than SP_CAP set bits in the tile).
    u32 g = g0, gnext = g0, round_lo = 0, jn = ji;
    u32 nF = 0, nR = 0, iF = 0, iR = 0, nmax = 0, pendR = 0;
    bool leaving = false;   // the previous tile was the last of its job (for this workgroup): convert before going on
    u32 leave_job = 0;
    for (;;) {
        const bool have_tile = g < g1;
        SP_STAMP(7)
        __syncthreads();   // B0: everyone is done with the previous tile's (round's) LDS
        SP_STAMP(0)
        if (have_tile && round_lo == 0) {
            if (SP_PRIO_STAGE) __builtin_amdgcn_s_setprio(SP_PRIO_STAGE);
            // phase A: consume the prefetched registers -- tile to LDS, set bits to records (needs no LDS input:
            // the mappability flag of a forward read is a bit of this thread's own M quad)
#ifdef SP_RESERVE_LATE
            tile_store<HAS_M, CH>(tr, tx, lds, tid);
            SP_STAMP(2)
#endif
            // the two cursor atomics go first: their LDS round trips (~400 cycles each under load) run under the
            // tile stores and the decimation instead of in front of the position loops
            if (!HAS_M) pendR = __popc(tr.r.x) + __popc(tr.r.y) + __popc(tr.r.z) + __popc(tr.r.w);
            iF = emit_reserve(tr.f, &cursor[2 * par]);
            SP_STAMP(8)
            if (HAS_M) iR = emit_reserve(tr.r, &cursor[2 * par + 1]);
            SP_STAMP(9)
#ifndef SP_RESERVE_LATE
            tile_store<HAS_M, CH>(tr, tx, lds, tid);
            SP_STAMP(2)
#endif
#ifndef SP_ABL_NOEMIT
            emit_positions_round0<HAS_M>(tr.f, HAS_M ? tr.m : tr.f, iF, lds + L::PLF, tid);
            SP_STAMP(10)
            if (HAS_M) emit_positions_round0<HAS_M>(tr.r, tr.r, iR, lds + L::PLR, tid);
#endif
            if (tid < 2) cursor[2 * (par ^ 1) + tid] = 0;   // next tile's cursors (last read before B0)
            SP_STAMP(3)
            // fetch the next tile into the (now free) registers; consumed after the next tile's B0.  Behind the event kernel
            // the unflagged tiles of this chromosome are skipped (pj is still the job of tile g here); the first tile of every
            // chromosome in this workgroup's range is visited whatever its flag, so that its slab segment gets written.
            gnext = g + 1;
            if (tile_flags) {
                const u32 lim = pj.tile_end < g1 ? pj.tile_end : g1;
                gnext = pj.tile0 + next_flagged_tile(tile_flags + pj.flag0, gnext - pj.tile0, lim - pj.tile0);
            }
            if (gnext < g1) {
                if (gnext >= pj.tile_end) {   // rare: the next tile belongs to the next job
                    jn = ji + 1;
                    load_job(pj, jobs.j[jn]);
                }
                tile_fetch_job<HAS_M, CH>(tr, tx, pj, gnext - pj.tile0, tid, tile_flags);
            }
            SP_STAMP(4)
            if (SP_PRIO_STAGE) __builtin_amdgcn_s_setprio(0);
            __syncthreads();   // B1: tile and records visible
            SP_STAMP(5)
#ifdef SP_ABL_NOPROC
            nF = 0;
            nR = 0;
#else
            nF = cursor[2 * par];
            nR = HAS_M ? cursor[2 * par + 1] : 0u;
#endif
            par ^= 1;
            nmax = nF > nR ? nF : nR;
        } else if (have_tile) {
            // rare: a tile with more than SP_CAP set bits in one vector -> re-read its words, emit the next round
            TileRegs cur;
            TileRegsX curx;
            tile_fetch<HAS_M, false, true>(cur, curx, jobs.j[ji].F, jobs.j[ji].R, jobs.j[ji].M,
                                           (int64_t)(g - jobs.j[ji].tile0) * SP_TBW, 0, jobs.j[ji].nbits, tid);
            emit_positions(cur.f, HAS_M ? cur.m : cur.f, iF, round_lo, lds + L::PLF, tid);
            if (HAS_M) emit_positions(cur.r, cur.r, iR, round_lo, lds + L::PLR, tid);
            __syncthreads();
        }
        u32 nFr = 0, nRr = 0;
        if (have_tile) {
            nFr = nF > round_lo ? (nF - round_lo < RCAP ? nF - round_lo : RCAP) : 0u;
            nRr = nR > round_lo ? (nR - round_lo < RCAP ? nR - round_lo : RCAP) : 0u;
        }
        // Quads (4 records) are dealt to the waves one wave-iteration (one quad per slot of the wave) at a time: wave w
        // takes iterations w, w + 4, ...  The waves of a workgroup differ by at most one iteration, and no wave pads
        // its share up to the workgroup's maximum (164 records in 8 slots: 6 + 5 + 5 + 5 iterations instead of 4 x 6).
        const u32 lg_spw = 6 - lgG;   // slots per wave = 2^lg_spw
        const u32 wiF = (((nFr + 3) >> 2) + (1u << lg_spw) - 1) >> lg_spw, wiR = (((nRr + 3) >> 2) + (1u << lg_spw) - 1) >> lg_spw;
        // ROLES: each role (2 waves) takes ALL forward quads; role B takes all reverse quads as well
        const bool roleA = wave < 2;
        const u32 wr = ROLES ? (wave & 1u) : wave;                          // wave index among the waves that share a list
        const u32 nqF = ROLES ? (wiF + 1) >> 1 : (wiF + 3) >> 2;            // the most any wave has: workgroup-uniform bounds
        const u32 nqR = ROLES ? (wiR + 1) >> 1 : (wiR + 3) >> 2;
#ifdef SP_UNBALANCED   // A/B: every wave pads up to the workgroup's maximum, as before
        const u32 nqFw = nqF, nqRw = nqR, wrR = wr;
#else
        const u32 nqFw = wr < wiF ? (ROLES ? (wiF - wr + 1) >> 1 : (wiF - wr + 3) >> 2) : 0u;   // this wave's quads per slot
        // the reverse quads are dealt in the opposite wave order (wave 3 first), so the odd iteration of each list
        // lands on a different wave: the tile's critical wave carries 6 + 5 iterations instead of 6 + 6
#ifdef SP_RDEAL_SAME
        const u32 wrR = wr;
#else
        const u32 wrR = ROLES ? wr : 3u - wave;
#endif
        const u32 nqRw = (ROLES && roleA) ? 0u : (wrR < wiR ? (ROLES ? (wiR - wrR + 1) >> 1 : (wiR - wrR + 3) >> 2) : 0u);
#endif
        const u32 rslotR = (wrR << lg_spw) + (sg.slot & ((1u << lg_spw) - 1));   // this slot's place in the reverse dealing

        // ---- the one fold / convert site ----
        // registers -> LDS accumulators: when leaving a job, at a tile boundary once enough quads are pending,
        // or (dense tiles only) before a round that would overflow a register counter
        if (leaving || (round_lo == 0 && (qF >= SP_QSOFT || qR >= SP_QSOFT)) || qF + nqF > SP_QLIMIT ||
            qR + nqR > SP_QLIMIT) {
            if (ROLES) {
                counter_to_lds(c0, qFw, lgG, tid, stage, acc, roleA);
                counter_to_lds(c0, qFw, lgG, tid, stage, acc + 1 * SP_NL * 32, !roleA);
                counter_to_lds(c1, qFw, lgG, tid, stage, acc + 2 * SP_NL * 32, roleA);
                counter_to_lds(c1, qRw, lgG, tid, stage, acc + 3 * SP_NL * 32, !roleA);
            } else {
                if (DO_NCC) counter_to_lds(cN, qFw, lgG, tid, stage, acc);
                if (HAS_M) {
                    counter_to_lds(cF, qFw, lgG, tid, stage, acc + 1 * SP_NL * 32);
                    counter_to_lds(cC, qFw, lgG, tid, stage, acc + 2 * SP_NL * 32);
                    counter_to_lds(cR, qRw, lgG, tid, stage, acc + 3 * SP_NL * 32);
                }
            }
            q2 += (qF > qR ? qF : qR) * 4u * sg.total_slots;
            qF = 0;
            qR = 0;
            qFw = 0;
            qRw = 0;
        }
        // LDS accumulators -> integers in this workgroup's slab segment of the job (when leaving it, or before the
        // 24-plane accumulators could overflow)
        if (leaving || q2 >= SP_L2LIMIT) {
            const u32 j = leaving ? leave_job : ji;
            u32 *seg = slab + (size_t)(bx + j) * SP_SEG_ROWS * 1024;
            acc_to_segment(acc, L::NCOUNTERS, HAS_M ? 8u : 0u, seg, seg_written, tid);   // counter 3 (rsum) is bit-reversed
            q2 = 0;
            seg_written = true;
            if (leaving) {
                if (!HAS_M) {   // popcount(R) was counted per thread
                    u32 v = cntR_thread;
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                    if ((tid & 63) == 0) cursor[4 + (tid >> 6)] = v;
                    __syncthreads();
                    totR = cursor[4] + cursor[5] + cursor[6] + cursor[7];
                    cntR_thread = 0;
                }
                if (tid == 0) {   // popcount(F), popcount(R): bit_array_num_bits_set of mscc.pyx:236-237
                    seg[4 * 1024 + 0] = totF;
                    seg[4 * 1024 + 1] = totR;
                }
                if (add_stride) {
                    // Behind the event pass (max_shift <= 1023, round 4): the rows of the chromosome are written already
                    // (k_events_finish) and this workgroup ADDS its share of the flagged tiles to them with 64-bit atomics --
                    // every thread the entries it has just written to the segment itself --, so that no launch is needed
                    // behind the window kernels (k_events_tail summed their segments until round 3: a launch in every
                    // step for tiles that the ordinary workload does not have).
                    const SpJobDev &jd = jobs.j[j];
                    unsigned long long *out = reinterpret_cast<unsigned long long *>(jd.out);
                    const u32 dst_row[4] = {PMX_ROW_NCC_CCBINS, PMX_ROW_MSCC_FSUM, PMX_ROW_MSCC_CCBINS, PMX_ROW_MSCC_RSUM};
#pragma unroll 1
                    for (u32 r = 0; r < 4; r++) {
                        if (r == 0 ? !DO_NCC : !HAS_M) continue;
#pragma unroll 1
                        for (u32 d = tid; d < jd.d_n; d += 256) {
                            const u32 v = seg[r * 1024 + d];
                            if (v) atomicAdd(&out[(size_t)dst_row[r] * add_stride + d], (unsigned long long)v);
                        }
                    }
                    if (tid == 0) {
                        if (totF) atomicAdd(&out[(size_t)PMX_ROW_SCALARS * add_stride + 0], (unsigned long long)totF);
                        if (totR) atomicAdd(&out[(size_t)PMX_ROW_SCALARS * add_stride + 1], (unsigned long long)totR);
                    }
                }
                totF = 0;
                totR = 0;
                seg_written = false;
                leaving = false;
            }
            __syncthreads();
        }
        SP_STAMP(1)
        if (!have_tile) break;

        if (round_lo == 0) {
            totF += nF;
            totR += nR;
            cntR_thread += pendR;
        }
        // this slot's record staging region (fixed size, shared by the forward and the reverse pass)
        uint4 *const recs = reinterpret_cast<uint4 *>(lds + L::REC) + (sg.slot << sg.lg_region);
        if (SP_PRIO_PROCESS) __builtin_amdgcn_s_setprio(SP_PRIO_PROCESS);

        if (ROLES) {
            // ---- role-split build: every wave carries two counters ----
            const u32 rslot = (wr << lg_spw) + (sg.slot & ((1u << lg_spw) - 1)), rstride = 2u << lg_spw;
            build_forward_records<HAS_M>(lds, recs, rslot, rstride, lane_in_slot, G, nqFw, nFr, c);
            if (roleA) {
                for (u32 q = 0; q < nqFw; q++) {
                    u32 wN[4], wC[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const uint4 rec = recs[q * 4 + k];
                        const u32 rw = lds_window(lds, rec.y + sg.l4, rec.x);
                        wN[k] = rw;
                        wC[k] = __builtin_bitreverse32(lds_window(lds, rec.z - sg.l4, rec.w)) & rw;
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qFw + q);
                    add_quad(c0, wN[0], wN[1], wN[2], wN[3], qc);
                    add_quad(c1, wC[0], wC[1], wC[2], wC[3], qc);
                }
            } else {
                for (u32 q = 0; q < nqFw; q++) {
                    u32 wF[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const uint4 rec = recs[q * 4 + k];
                        wF[k] = __builtin_bitreverse32(lds_window(lds, rec.z - sg.l4, rec.w));
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qFw + q);
                    add_quad(c0, wF[0], wF[1], wF[2], wF[3], qc);
                }
                build_reverse_records(lds, recs, rslot, rstride, lane_in_slot, G, nqRw, nRr, c);
                for (u32 q = 0; q < nqRw; q++) {
                    u32 wR[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const uint4 rec = recs[q * 4 + k];
                        wR[k] = lds_window(lds, rec.y - sg.l4, rec.x) & lds_window(lds, rec.z - sg.l4, rec.w);
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qRw + q);
                    add_quad(c1, wR[0], wR[1], wR[2], wR[3], qc);
                }
            }
            qF += nqF;
            qFw += nqFw;
            qR += nqR;
            qRw += nqRw;
        } else {
        // ---- forward reads drive: ncc, mscc.fsum, mscc.ccbins ----
        {
            build_forward_records<HAS_M>(lds, recs, sg.slot, sg.total_slots, lane_in_slot, G, nqFw, nFr, c);
            u32 q = 0;
            // two quads per trip: one parked-carry branch per counter (add_carry_pair).  NCC-only: -3 % (0.296 -> 0.286 ms);
            // with mappability the three extra live carries push the allocation into scratch and nothing is gained, so
            // that instantiation keeps one quad per trip
            for (; QUAD_PAIRS && q + 1 < nqFw; q += 2) {
                u32 c2N[2], c2F[2], c2C[2];
#pragma unroll
                for (u32 h = 0; h < 2; h++) {
                    u32 wN[4], wF[4], wC[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const uint4 rec = recs[(q + h) * 4 + k];
                        const u32 rw = lds_window(lds, rec.y + sg.l4, rec.x);
                        wN[k] = rw;
                        if (HAS_M) {
                            const u32 mw = __builtin_bitreverse32(lds_window(lds, rec.z - sg.l4, rec.w));
                            wF[k] = mw;
                            wC[k] = mw & rw;
                        }
                    }
                    if (DO_NCC) c2N[h] = quad_carry(cN, wN[0], wN[1], wN[2], wN[3]);
                    if (HAS_M) {
                        c2F[h] = quad_carry(cF, wF[0], wF[1], wF[2], wF[3]);
                        c2C[h] = quad_carry(cC, wC[0], wC[1], wC[2], wC[3]);
                    }
                }
                const u32 qc = __builtin_amdgcn_readfirstlane(qFw + q);
                if (DO_NCC) add_carry_pair(cN, c2N[0], c2N[1], qc);
                if (HAS_M) {
                    add_carry_pair(cF, c2F[0], c2F[1], qc);
                    add_carry_pair(cC, c2C[0], c2C[1], qc);
                }
            }
            for (; q < nqFw; q++) {
                u32 wN[4], wF[4], wC[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const uint4 rec = recs[q * 4 + k];
                    const u32 rw = lds_window(lds, rec.y + sg.l4, rec.x);
                    wN[k] = rw;
                    if (HAS_M) {
                        const u32 mw = __builtin_bitreverse32(lds_window(lds, rec.z - sg.l4, rec.w));
                        wF[k] = mw;
                        wC[k] = mw & rw;
                    }
                }
                const u32 qc = __builtin_amdgcn_readfirstlane(qFw + q);
                if (DO_NCC) add_quad(cN, wN[0], wN[1], wN[2], wN[3], qc);
                if (HAS_M) {
                    add_quad(cF, wF[0], wF[1], wF[2], wF[3], qc);
                    add_quad(cC, wC[0], wC[1], wC[2], wC[3], qc);
                }
            }
            qF += nqF;
            qFw += nqFw;
        }
        // ---- reverse reads drive: mscc.rsum ----
        if (HAS_M) {
            build_reverse_records(lds, recs, rslotR, sg.total_slots, lane_in_slot, G, nqRw, nRr, c);
            u32 q = 0;
            for (; QUAD_PAIRS && q + 1 < nqRw; q += 2) {
                u32 c2R[2];
#pragma unroll
                for (u32 h = 0; h < 2; h++) {
                    u32 wR[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const uint4 rec = recs[(q + h) * 4 + k];
                        wR[k] = lds_window(lds, rec.y - sg.l4, rec.x) & lds_window(lds, rec.z - sg.l4, rec.w);
                    }
                    c2R[h] = quad_carry(cR, wR[0], wR[1], wR[2], wR[3]);
                }
                add_carry_pair(cR, c2R[0], c2R[1], __builtin_amdgcn_readfirstlane(qRw + q));
            }
            for (; q < nqRw; q++) {
                u32 wR[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const uint4 rec = recs[q * 4 + k];
                    const u32 w1 = lds_window(lds, rec.y - sg.l4, rec.x);
                    const u32 w2 = lds_window(lds, rec.z - sg.l4, rec.w);
                    wR[k] = w1 & w2;   // bit i <-> shift 32 l + 31 - i: this counter is kept bit-reversed (see convert)
                }
                const u32 qc = __builtin_amdgcn_readfirstlane(qRw + q);
                add_quad(cR, wR[0], wR[1], wR[2], wR[3], qc);
            }
            qR += nqR;
            qRw += nqRw;
        }
        }
        if (SP_PRIO_PROCESS) __builtin_amdgcn_s_setprio(0);
        SP_STAMP(6)

        // next round of this tile, or next tile (and remember to convert when the job ends here)
        round_lo += RCAP;
        if (round_lo >= nmax) {
            round_lo = 0;
            if (jn != ji || gnext >= g1) {
                leaving = true;
                leave_job = ji;
            }
            ji = jn;
            g = gnext;
        }
    }
#ifdef SP_STAMPS
    if ((tid & 63) == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(slab + (size_t)(gridDim.x + njobs) * SP_SEG_ROWS * 1024);
        for (int i = 0; i < SP_NSTAMP; i++) dbg[((size_t)bx * 4 + (tid >> 6)) * SP_NSTAMP + i] = stamp_acc[i];
    }
#endif
}

template <bool HAS_M, bool DO_NCC, bool CH, typename JT = SpJobTable>
__global__ void __launch_bounds__(256, CH ? (HAS_M ? SP_WAVES_CH : SP_WAVES_CH_NCC) : (HAS_M ? SP_WAVES : SP_WAVES_NCC))
k_cc_sparse(const JT jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg, int32_t c, u32 lgG,
            u32 *__restrict__ slab, const unsigned char *__restrict__ tile_flags, const u32 *__restrict__ n_flagged,
            const u32 *__restrict__ plan, u32 add_stride = 0)
{
    if (n_flagged && *n_flagged == 0) return;   // the event kernel took every tile (uniform over the whole grid)
    __shared__ __align__(16) u32 lds[SpLds<HAS_M>::TOTAL];
    cc_sparse_body<HAS_M, DO_NCC, CH, JT>(lds, blockIdx.x, jobs, njobs, total_tiles, tiles_per_wg, c, lgG, slab, tile_flags, plan, add_stride);
}

// dst[job][row][d_off + i] = sum over the workgroups that touched the job of slab[(wg + job)][src_row][i], i < d_n
struct ReduceSpec {
    u32 nrows;
    u32 src_row[5];
    u32 dst_row[5];     // row of the result block (PMX_ROW_*); with use_out2: offset (u64 units) into the job's out2
    u32 is_scalar[5];   // the scalar row: 2 sums, the rest of the row zero-filled, [3] = path marker; chunk 0 only
    u32 out_stride;
    u32 use_out2;       // autocorrelation: rows go to out2 (per-job scratch) instead of the result block
    u32 accumulate;     // sums are ADDED to what an earlier pass (pair / event kernel) left in the destination
    u32 is_signed[5];   // the slab row holds i32 (event histograms of edge signs)
    u32 scalar_off;     // first of the two scalars in the slab's scalar row
    u32 n_override;     // entries per row (0: the job's d_n)
    u32 keep_scalar2;   // scalar [2] (popcount(M)) belongs to the autocorrelation pass, which may run concurrently
    u32 nzero;          // rows of the result block this batch does not produce: written as zeros (chunk 0 only)
    u32 zero_row[5];
    u32 rowlen;         // u32 per slab row (0: 1024; the event kernel beyond 1023 shifts: its histogram length)
};

#define RS_NBX 32u   // chunks of 32 elements per row of 1024 shifts (rowlen / 32 in general); a block takes the chunks blockIdx.x, + gridDim.x, ...
__device__ __forceinline__ u32 rs_rowlen(const ReduceSpec &rs) { return rs.rowlen ? rs.rowlen : 1024u; }
template <typename JT>
__device__ __forceinline__ void reduce_segments_chunk(const u32 *__restrict__ slab, const JT &jobs, u32 seg_rows,
                                                      const ReduceSpec &rs, const u32 *__restrict__ gate, u32 r, u32 bx,
                                                      u64 (*part)[32])   // part[blockDim.x / 32][32]
{
    const bool live = !gate || *gate != 0;   // gate == 0: the producing kernel did not run, every sum is zero
    if (!live && rs.accumulate) return;      // ... and there is nothing to add
    // 32 consecutive elements x (blockDim.x / 32) workgroup phases per block: every load is a full 128-B line
    const u32 job = blockIdx.z, ng = blockDim.x >> 5;
    const SpJobDev &jb = jobs.j[job];
    const u32 e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const u32 i = bx * 32 + e;
    const u32 rowlen = rs_rowlen(rs);
    if (r >= rs.nrows) {   // a row this batch leaves empty
        if ((jb.flags & 1u) && g == 0 && !rs.accumulate) {
            u64 *dst = rs.use_out2 ? jb.out2 + rs.zero_row[r - rs.nrows] : jb.out + (size_t)rs.zero_row[r - rs.nrows] * rs.out_stride;
            const u32 cnt = rs.use_out2 ? (rs.n_override ? rs.n_override : jb.d_n) : rs.out_stride;
            for (u32 k = i; k < cnt; k += rowlen) dst[k] = 0;
        }
        return;
    }
    const bool scalar = rs.is_scalar[r] != 0;
    if (scalar && !(jb.flags & 1u)) return;
    const u32 n = scalar ? 2u : (rs.n_override ? rs.n_override : jb.d_n);
    u64 sum = 0;
    if (i < n && live) {
        const size_t stride = (size_t)seg_rows * rowlen;
        const u32 *p = slab + (size_t)rs.src_row[r] * rowlen + (scalar ? rs.scalar_off : 0u) + i;
        if (rs.is_signed[r])
            for (u32 w = jb.wg_first + g; w <= jb.wg_last; w += ng) sum += (u64)(long long)(int32_t)p[(size_t)(w + job) * stride];
        else
            for (u32 w = jb.wg_first + g; w <= jb.wg_last; w += ng) sum += p[(size_t)(w + job) * stride];
    }
    part[g][e] = sum;
    __syncthreads();
    if (g != 0) return;
    u64 t = 0;
    for (u32 k = 0; k < ng; k++) t += part[k][e];
    if (rs.use_out2) {
        if (i < n) {
            u64 *dst = jb.out2 + (size_t)rs.dst_row[r] + (scalar ? 0u : jb.d_off) + i;
            *dst = rs.accumulate ? *dst + t : t;
        }
    } else if (scalar) {
        u64 *dst = jb.out + (size_t)rs.dst_row[r] * rs.out_stride;
        for (u32 k = i; k < rs.out_stride; k += rowlen) {   // [0],[1] sums, [3] path, everything else zero
            if (k == 2 && rs.keep_scalar2) continue;
            if (rs.accumulate) {
                if (k < 2) dst[k] += t;
                continue;
            }
            dst[k] = k < 2 ? t : (k == 3 ? (u64)PMX_PATH_SPARSE : 0ull);
        }
    } else if (i < n) {
        u64 *dst = jb.out + (size_t)rs.dst_row[r] * rs.out_stride + jb.d_off + i;
        *dst = rs.accumulate ? *dst + t : t;
    }
}

// A job that few workgroups touched (many short chromosomes, max_shift > 1023: two or three segments of 8192-entry rows):
// every thread sums the segments of one element, a block takes blockDim.x consecutive elements per trip -- no LDS, no
// barrier, every lane busy.  (The phased form below keeps 3 of 8 phases busy there: 147 us for config 5's 57 MB.)
template <typename JT>
__device__ __forceinline__ bool reduce_segments_row_wide(const u32 *__restrict__ slab, const JT &jobs, u32 seg_rows,
                                                         const ReduceSpec &rs, u32 r)
{
    const u32 job = blockIdx.z;
    const SpJobDev &jb = jobs.j[job];
    if (r >= rs.nrows || rs.is_scalar[r] || jb.wg_last - jb.wg_first >= 8u) return false;   // (uniform over the block)
    const u32 rowlen = rs_rowlen(rs);
    const u32 n = rs.n_override ? rs.n_override : jb.d_n;
    const size_t stride = (size_t)seg_rows * rowlen;
    const u32 *p = slab + (size_t)rs.src_row[r] * rowlen;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u64 sum = 0;
        if (rs.is_signed[r])
            for (u32 w = jb.wg_first; w <= jb.wg_last; w++) sum += (u64)(long long)(int32_t)p[(size_t)(w + job) * stride + i];
        else
            for (u32 w = jb.wg_first; w <= jb.wg_last; w++) sum += p[(size_t)(w + job) * stride + i];
        u64 *dst = rs.use_out2 ? jb.out2 + (size_t)rs.dst_row[r] + jb.d_off + i
                               : jb.out + (size_t)rs.dst_row[r] * rs.out_stride + jb.d_off + i;
        *dst = rs.accumulate ? *dst + sum : sum;
    }
    return true;
}

template <typename JT>
__device__ __forceinline__ void reduce_segments_row(const u32 *__restrict__ slab, const JT &jobs, u32 seg_rows,
                                                    const ReduceSpec &rs, const u32 *__restrict__ gate, u32 r)
{
    __shared__ u64 part[32][32];   // (blocks of 256 or 1024 threads)
    if (gate && *gate == 0 && rs.accumulate) return;   // the producing kernel did not run: nothing to add
    if ((!gate || *gate != 0) && reduce_segments_row_wide(slab, jobs, seg_rows, rs, r)) return;
    for (u32 bx = blockIdx.x; bx < rs_rowlen(rs) / 32; bx += gridDim.x) {   // (uniform over the block)
        reduce_segments_chunk(slab, jobs, seg_rows, rs, gate, r, bx, part);
        __syncthreads();
    }
}

template <typename JT = SpJobTable>
__global__ void __launch_bounds__(256)
k_reduce_segments(const u32 *__restrict__ slab, const JT jobs, u32 seg_rows, ReduceSpec rs,
                  const u32 *__restrict__ gate)
{
    reduce_segments_row(slab, jobs, seg_rows, rs, gate, blockIdx.y);
}

// two specifications over the same slab in one launch (grid.y = rows of A + rows of B)
__global__ void __launch_bounds__(1024)
k_reduce_segments2(const u32 *__restrict__ slab, const SpJobTable jobs, u32 seg_rows, ReduceSpec rsA, ReduceSpec rsB)
{
    const u32 rowsA = rsA.nrows + rsA.nzero;   // (uniform over the block)
    if (blockIdx.y < rowsA)
        reduce_segments_row(slab, jobs, seg_rows, rsA, nullptr, blockIdx.y);
    else
        reduce_segments_row(slab, jobs, seg_rows, rsB, nullptr, blockIdx.y - rowsA);
}

// ---------------------------------------------------------------------------------------------------
// Mappability autocorrelation A(k) = sum_j M[j] & M[j+k] (mappable_len: mscc.pyx:291-298 by symmetry, and the
// read-less loop mscc.pyx:207-215) from RUN EDGES instead of dense popcounts.  With E[j] = M[j] - M[j-1]
// (+1 at run starts U, -1 one past run ends D; E is defined on [0, nbits]):
//     (E*E)(k) = sum_j E[j] E[j+k] = 2 A(k) - A(k-1) - A(k+1)
// so A(k+1) = 2 A(k) - A(k-1) - EE(k), A(0) = popcount(M), A(1) = A(0) - #runs, and
//     EE(k) = [U*U + D*D](k) - [U*D + D*U](k) = P(k) - N(k)
// are window sums driven by the edges only (two edges per mappable run), computed with the same set-bit
// machinery as k_cc_sparse.  k_autocorr_finish runs the integer recurrence.
#ifndef AC_NQ
#define AC_NQ 2                      // driver quads per thread: the autocorrelation tile is AC_NQ x 32 Kbit (the pass is all
#endif                               // per-tile overhead on sparse-edge tracks, so its tile is larger than k_cc_sparse's)
#define AC_TBW (AC_NQ * SP_TBW)      // dwords per autocorrelation tile
#define AC_TB (AC_NQ * SP_TB)
#define AC_W (AC_TBW + SP_RHI)       // dwords of U / D per tile
struct AcLds {
    static constexpr u32 ZERO = 0;
    static constexpr u32 U = 40;
    static constexpr u32 D = U + AC_W;
    static constexpr u32 PL = D + AC_W;                  // positions of the edges (+ falling flag)
    static constexpr u32 REC = PL + SP_CAP;              // per-slot record staging (4 dwords each)
    static constexpr u32 ACC = REC + 4 * SP_CAP;         // [2][plane][32]
    static constexpr u32 STAGE = REC;                    // aliases the records (same argument as in SpLds)
    static constexpr u32 MISC = ACC + 2 * SP_NL * 32;
    static constexpr u32 TOTAL = MISC + 16;
};
#define AC_SEG_ROWS 3u               // P, N, scalars
#ifndef AC_WAVES
#define AC_WAVES 4
#endif
#ifndef AC_WAVES_CH
#define AC_WAVES_CH 4                // lag-chunked instantiation (max lag > 1023)
#endif

struct AcRegs {
    uint4 m[AC_NQ], h;   // driver quads (unshifted; quad q covers dwords q*1024 + 4 tid ..) + (threads 0..8) the quad
                         // above the WINDOW tile
    u32 below[AC_NQ];    // dword just below each driver quad (for M[j-1])
    u32 hbelow;
    uint4 mw[AC_NQ];     // lag chunks > 0 only: main quads of the window tile, staged from +off dwords
    u32 wbelow[AC_NQ];
};

template <bool GUARD, bool CH>
__device__ __forceinline__ void ac_fetch(AcRegs &ar, const u32 *__restrict__ M, int64_t d0, int64_t off, uint64_t nbits, u32 tid)
{
#pragma unroll
    for (int q = 0; q < AC_NQ; q++) {
        const int64_t j = d0 + (int64_t)q * SP_TBW + 4 * (int64_t)tid;
        ar.m[q] = ld_quad<GUARD>(M, j, nbits);
        ar.below[q] = GUARD ? ld_dword_guarded(M, j - 1, nbits) : M[j - 1];
        if (CH) {
            ar.mw[q] = ld_quad<GUARD>(M, j + off, nbits);
            ar.wbelow[q] = GUARD ? ld_dword_guarded(M, j + off - 1, nbits) : M[j + off - 1];
        }
    }
    ar.h = make_uint4(0, 0, 0, 0);
    ar.hbelow = 0;
    if (tid < 9) {
        const int64_t jh = d0 + off + AC_TBW + 4 * (int64_t)tid;
        ar.h = ld_quad<GUARD>(M, jh, nbits);
        ar.hbelow = GUARD ? ld_dword_guarded(M, jh - 1, nbits) : M[jh - 1];
    }
}

// flags != null: only the tiles the pair kernel flagged as dense are processed; the others read as all-zero words
// (no edges, no set bits: they contribute nothing) without touching memory
template <bool CH>
__device__ __forceinline__ void ac_fetch_job(AcRegs &ar, const SpJobDev &jb, u32 local_tile, u32 tid,
                                             const unsigned char *__restrict__ flags)
{
    if (flags && !flags[jb.flag0 + local_tile]) {
#pragma unroll
        for (int q = 0; q < AC_NQ; q++) {
            ar.m[q] = make_uint4(0, 0, 0, 0);
            ar.below[q] = 0;
            if (CH) {
                ar.mw[q] = make_uint4(0, 0, 0, 0);
                ar.wbelow[q] = 0;
            }
        }
        ar.h = make_uint4(0, 0, 0, 0);
        ar.hbelow = 0;
        return;
    }
    const int64_t d0 = (int64_t)local_tile * AC_TBW;
    const int64_t off = CH ? (int64_t)(jb.d_off / 32) : 0;
    const uint64_t hi = (uint64_t)(d0 + off) + AC_TBW + SP_RHI;
    const bool interior = jb.aligned16 && local_tile > 0 && hi + 2 <= jb.nbits / 32;
    if (interior)
        ac_fetch<false, CH>(ar, jb.M, d0, off, jb.nbits, tid);
    else
        ac_fetch<true, CH>(ar, jb.M, d0, off, jb.nbits, tid);
}

// rising (U) and falling (D) edge words of a quad; `below` = the dword preceding m.x
__device__ __forceinline__ void edge_quad(const uint4 m, u32 below, uint4 &U, uint4 &D)
{
    const u32 s0 = (m.x << 1) | (below >> 31), s1 = (m.y << 1) | (m.x >> 31);
    const u32 s2 = (m.z << 1) | (m.y >> 31), s3 = (m.w << 1) | (m.z >> 31);
    U = make_uint4(m.x & ~s0, m.y & ~s1, m.z & ~s2, m.w & ~s3);
    D = make_uint4(~m.x & s0, ~m.y & s1, ~m.z & s2, ~m.w & s3);
}

// Edge records: {shift word, byte address of the SAME-sign edge vector window (lane adds 4l),
//                byte address of the OPPOSITE-sign edge vector window, 0}; built per slot from the position list
__device__ __forceinline__ void build_edge_records(u32 *lds, uint4 *recs, u32 first, u32 lane_in_slot, u32 G, u32 nq, u32 n)
{
    const u32 *pl = lds + AcLds::PL;
    for (u32 j = lane_in_slot; j < 4 * nq; j += G) {
        uint4 rec = make_uint4(0u, AcLds::ZERO * 4u, AcLds::ZERO * 4u, 0u);
        if (first + j < n) {
            const u32 e = pl[first + j];
            const u32 pos = e & 0xffffu;
            const u32 au = (AcLds::U + (pos >> 5)) * 4u, ad = (AcLds::D + (pos >> 5)) * 4u;
            const bool fall = (e >> 16) != 0;
            rec = make_uint4(pos, fall ? ad : au, fall ? au : ad, 0u);
        }
        recs[j] = rec;
    }
}

__device__ __forceinline__ long long block_exclusive_offset(long long local_sum, long long *part, u32 tid);   // (below)

// (body / wrapper: see cc_sparse_body)
template <bool CH, typename JT>
__device__ __forceinline__ void autocorr_edges_body(u32 *const lds, const u32 bx, const JT &jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg,
                                                    u32 lgG, u32 *__restrict__ slab, const unsigned char *__restrict__ tile_flags,
                                                    const u32 *__restrict__ plan, u32 add_stride, int32_t add_c, u32 add_shift, u32 add_lag)
{
    typedef AcLds L;
    u32 *const cursor = lds + L::MISC;
    u32 *const acc = lds + L::ACC;
    u32 *const stage = lds + L::STAGE;

    const u32 tid = threadIdx.x;
    const SlotGeom sg = slot_geom(lgG, tid);

    const u32 g0 = plan ? plan[2 * bx] : bx * tiles_per_wg;      // (see k_cc_sparse)
    const u32 g1 = plan ? plan[2 * bx + 1] : (g0 + tiles_per_wg < total_tiles ? g0 + tiles_per_wg : total_tiles);
    if (g0 >= g1) return;

    for (u32 i = tid; i < 40; i += 256) lds[L::ZERO + i] = 0;
    for (u32 i = tid; i < 2 * SP_NL * 32; i += 256) acc[i] = 0;

    Planes cP, cN;
    planes_zero(cP);
    planes_zero(cN);
    u32 qc = 0, q2 = 0;
    u32 cntM = 0, cntU = 0;
    bool seg_written = false;

    FIRST_JOB(ji, JT, jobs, njobs, g0)

    AcRegs ar;
    ac_fetch_job<CH>(ar, jobs.j[ji], g0 - jobs.j[ji].tile0, tid, tile_flags);
    if (tid < 2) cursor[tid] = 0;   // record cursors, parity alternates per tile
    u32 par = 0;
    const u32 G = 1u << lgG;
    const u32 lane_in_slot = sg.l4 >> 2;

    // (tile, round) pairs + one drain pass; the fold / convert code is inlined once (see k_cc_sparse)
    u32 g = g0, gnext = g0, round_lo = 0, jn = ji, n = 0, i0 = 0, pendM = 0, pendU = 0;
    bool leaving = false;
    u32 leave_job = 0;
    for (;;) {
        const bool have_tile = g < g1;
        __syncthreads();   // B0: everyone is done with the previous tile's (round's) LDS
        if (have_tile && round_lo == 0) {
            // phase A: edges from the prefetched registers -> LDS tiles and records
            uint4 E[AC_NQ], Dq[AC_NQ];
            pendM = 0;
            pendU = 0;
            u32 nE = 0;
#pragma unroll
            for (int q = 0; q < AC_NQ; q++) {
                uint4 U, D;
                edge_quad(ar.m[q], ar.below[q], U, D);   // the drivers: edges inside this tile
                if (CH) {                                // the windows start d_off bits above
                    uint4 Uw, Dw;
                    edge_quad(ar.mw[q], ar.wbelow[q], Uw, Dw);
                    reinterpret_cast<uint4 *>(lds + L::U + q * SP_TBW)[tid] = Uw;
                    reinterpret_cast<uint4 *>(lds + L::D + q * SP_TBW)[tid] = Dw;
                } else {
                    reinterpret_cast<uint4 *>(lds + L::U + q * SP_TBW)[tid] = U;
                    reinterpret_cast<uint4 *>(lds + L::D + q * SP_TBW)[tid] = D;
                }
                E[q] = make_uint4(U.x | D.x, U.y | D.y, U.z | D.z, U.w | D.w);
                Dq[q] = D;
                pendM += __popc(ar.m[q].x) + __popc(ar.m[q].y) + __popc(ar.m[q].z) + __popc(ar.m[q].w);
                pendU += __popc(U.x) + __popc(U.y) + __popc(U.z) + __popc(U.w);
                nE += __popc(E[q].x) + __popc(E[q].y) + __popc(E[q].z) + __popc(E[q].w);
            }
            if (tid < 9) {
                uint4 Uh, Dh;
                edge_quad(ar.h, ar.hbelow, Uh, Dh);
                reinterpret_cast<uint4 *>(lds + L::U + AC_TBW)[tid] = Uh;
                reinterpret_cast<uint4 *>(lds + L::D + AC_TBW)[tid] = Dh;
            }
            i0 = nE ? atomicAdd(&cursor[par], nE) : 0u;
            {
                u32 at = i0;
#pragma unroll
                for (int q = 0; q < AC_NQ; q++) {   // flag = falling edge
                    emit_positions(E[q], Dq[q], at, 0, lds + L::PL, tid, q * SP_TB);
                    at += __popc(E[q].x) + __popc(E[q].y) + __popc(E[q].z) + __popc(E[q].w);
                }
            }
            if (tid == 0) cursor[par ^ 1] = 0;
            jn = ji;
            // behind the pair / event kernel the unflagged tiles of this chromosome are skipped; the first tile of every
            // chromosome in this workgroup's range is visited whatever its flag, so that its slab segment gets written
            gnext = g + 1;
            if (!CH && tile_flags) {
                const u32 je = jobs.j[ji].tile0 + jobs.j[ji].ntiles, lim = je < g1 ? je : g1;
                gnext = jobs.j[ji].tile0 + next_flagged_tile(tile_flags + jobs.j[ji].flag0, gnext - jobs.j[ji].tile0, lim - jobs.j[ji].tile0);
            }
            if (gnext < g1) {
                if (gnext >= jobs.j[ji].tile0 + jobs.j[ji].ntiles) jn = ji + 1;
                ac_fetch_job<CH>(ar, jobs.j[jn], gnext - jobs.j[jn].tile0, tid, tile_flags);
            }
            __syncthreads();   // B1: tiles and records visible
            n = cursor[par];
            par ^= 1;
        } else if (have_tile) {
            // rare: more than SP_CAP edges in one tile -> recompute its edge words, emit the next round
            AcRegs cur;
            ac_fetch<true, false>(cur, jobs.j[ji].M, (int64_t)(g - jobs.j[ji].tile0) * AC_TBW, 0, jobs.j[ji].nbits, tid);
            u32 at = i0;
#pragma unroll
            for (int q = 0; q < AC_NQ; q++) {
                uint4 U2, D2;
                edge_quad(cur.m[q], cur.below[q], U2, D2);
                const uint4 E2 = make_uint4(U2.x | D2.x, U2.y | D2.y, U2.z | D2.z, U2.w | D2.w);
                emit_positions(E2, D2, at, round_lo, lds + L::PL, tid, q * SP_TB);
                at += __popc(E2.x) + __popc(E2.y) + __popc(E2.z) + __popc(E2.w);
            }
            __syncthreads();
        }
        u32 nr = 0;
        if (have_tile) nr = n > round_lo ? (n - round_lo < SP_CAP ? n - round_lo : SP_CAP) : 0u;
        const u32 nq = (nr + sg.quad_span - 1) >> sg.lg_span;

        // ---- the one fold / convert site ----
        if (leaving || (round_lo == 0 && qc >= SP_QSOFT) || qc + nq > SP_QLIMIT) {
            counter_to_lds(cP, qc, lgG, tid, stage, acc);
            counter_to_lds(cN, qc, lgG, tid, stage, acc + SP_NL * 32);
            q2 += qc * 4u * sg.total_slots;
            qc = 0;
        }
        if (leaving || q2 >= SP_L2LIMIT) {
            const u32 j = leaving ? leave_job : ji;
            u32 *seg = slab + (size_t)(bx + j) * AC_SEG_ROWS * 1024;
            acc_to_segment(acc, 2, 0u, seg, seg_written, tid);
            q2 = 0;
            seg_written = true;
            if (leaving) {
                u32 vm = cntM, vu = cntU;
                for (int off = 32; off > 0; off >>= 1) {
                    vm += __shfl_down(vm, off, 64);
                    vu += __shfl_down(vu, off, 64);
                }
                if ((tid & 63) == 0) {
                    cursor[4 + (tid >> 6)] = vm;
                    cursor[8 + (tid >> 6)] = vu;
                }
                __syncthreads();
                if (tid == 0) {
                    seg[2048 + 0] = cursor[4] + cursor[5] + cursor[6] + cursor[7];     // popcount(M) of my tiles
                    seg[2048 + 1] = cursor[8] + cursor[9] + cursor[10] + cursor[11];   // runs starting in my tiles
                }
                if (!CH && add_stride) {
                    // Behind the event pass with the fused mappable-length pairs (max_lag <= 1023, round 4): the recurrence
                    // A(k+1) = 2 A(k) - A(k-1) - EE(k) is LINEAR in (popcount(M), runs, EE), so this workgroup runs it on its
                    // own share -- the flagged tiles it took -- and adds A(|c - d|) to the row the event pass has written
                    // (k_events_finish); no launch behind the window kernels.  64-bit, two's complement: a share may be negative.
                    const long long a0 = (long long)(cursor[4] + cursor[5] + cursor[6] + cursor[7]);
                    const long long runs = (long long)(cursor[8] + cursor[9] + cursor[10] + cursor[11]);
                    long long *const A = reinterpret_cast<long long *>(lds + L::U);     // (tile staging: free between two tiles)
                    long long *const part = reinterpret_cast<long long *>(lds + L::REC);
                    __syncthreads();
                    const u32 k0 = 4 * tid;
                    long long x[4], dl[4], tX = 0;
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        const u32 k = k0 + q;   // (each thread reads the entries it wrote to the segment itself)
                        x[q] = k == 0 ? -runs : (k <= add_lag ? (long long)seg[1024 + k] - (long long)seg[k] : 0ll);
                        tX += x[q];
                        dl[q] = tX;
                    }
                    const long long bX = block_exclusive_offset(tX, part, tid);   // (ends with a barrier)
                    long long tD = 0, ex[4];
#pragma unroll
                    for (u32 q = 0; q < 4; q++) {
                        dl[q] += bX;      // Delta(k), inclusive
                        ex[q] = tD;
                        tD += dl[q];
                    }
                    const long long bD = block_exclusive_offset(tD, part, tid);
#pragma unroll
                    for (u32 q = 0; q < 4; q++) A[k0 + q] = a0 + bD + ex[q];
                    __syncthreads();
                    const SpJobDev &jd = jobs.j[j];
                    unsigned long long *out = reinterpret_cast<unsigned long long *>(jd.out);
                    for (u32 d = tid; d <= add_shift; d += 256) {
                        const int32_t k = add_c - (int32_t)d;
                        const long long v = A[k < 0 ? -k : k];
                        if (v) atomicAdd(&out[(size_t)PMX_ROW_MLEN * add_stride + d], (unsigned long long)v);
                    }
                    if (tid == 0 && a0) atomicAdd(&out[(size_t)PMX_ROW_SCALARS * add_stride + 2], (unsigned long long)a0);
                    __syncthreads();
                }
                cntM = 0;
                cntU = 0;
                seg_written = false;
                leaving = false;
            }
            __syncthreads();
        }
        if (!have_tile) break;

        if (round_lo == 0) {
            cntM += pendM;
            cntU += pendU;
        }
        {
            uint4 *const recs = reinterpret_cast<uint4 *>(lds + L::REC) + (sg.slot << sg.lg_region);
            build_edge_records(lds, recs, sg.slot * 4 * nq, lane_in_slot, G, nq, nr);
            for (u32 q = 0; q < nq; q++) {
                u32 wp[4], wn[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const uint4 rec = recs[q * 4 + k];
                    wp[k] = lds_window(lds, rec.y + sg.l4, rec.x);   // same-sign pairs: U*U, D*D
                    wn[k] = lds_window(lds, rec.z + sg.l4, rec.x);   // opposite-sign pairs: U*D, D*U
                }
                const u32 qq = __builtin_amdgcn_readfirstlane(qc + q);
                add_quad(cP, wp[0], wp[1], wp[2], wp[3], qq);
                add_quad(cN, wn[0], wn[1], wn[2], wn[3], qq);
            }
            qc += nq;
        }

        round_lo += SP_CAP;
        if (round_lo >= n) {
            round_lo = 0;
            if (jn != ji || gnext >= g1) {
                leaving = true;
                leave_job = ji;
            }
            ji = jn;
            g = gnext;
        }
    }
}

template <bool CH, typename JT = SpJobTable>
__global__ void __launch_bounds__(256, CH ? AC_WAVES_CH : AC_WAVES)
k_autocorr_edges(const JT jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg, u32 lgG, u32 *__restrict__ slab,
                 const unsigned char *__restrict__ tile_flags, const u32 *__restrict__ n_flagged, const u32 *__restrict__ plan = nullptr,
                 u32 add_stride = 0, int32_t add_c = 0, u32 add_shift = 0, u32 add_lag = 0)
{
    if (n_flagged && *n_flagged == 0) return;   // the pair kernel took every tile (uniform over the whole grid)
    __shared__ __align__(16) u32 lds[AcLds::TOTAL];
    autocorr_edges_body<CH, JT>(lds, blockIdx.x, jobs, njobs, total_tiles, tiles_per_wg, lgG, slab, tile_flags, plan, add_stride, add_c,
                                add_shift, add_lag);
}

// Both window kernels behind the event pass (max_shift <= 1023, mappable-length pairs fused) in ONE launch: workgroups
// [0, nwg_cc) run the cross-correlation windows over the tiles flagged for them, the others the autocorrelation windows over
// theirs; each half returns at once when its counter of flagged tiles is zero (the usual case: the launch is then one empty
// grid instead of two -- a dependent launch costs 3-6 us of stream time even when it does nothing, tools/launch_cost.hip).
// Registers and LDS are the larger of the two bodies'; the occupancy is k_cc_sparse's (SP_WAVES <= AC_WAVES).
struct WinCcArgs {
    u32 total_tiles, tiles_per_wg, lgG;
    int32_t c;
    u32 *slab;
    const unsigned char *tile_flags;
    const u32 *plan;
};
struct WinAcArgs {
    u32 total_tiles, tiles_per_wg, lgG;
    u32 add_shift, add_lag;
    u32 *slab;
    const unsigned char *tile_flags;
    const u32 *plan;
};
template <bool DO_NCC>
__global__ void __launch_bounds__(256, SP_WAVES < AC_WAVES ? SP_WAVES : AC_WAVES)
k_windows_flagged(const SpJobTable jobs_cc, const SpJobTable jobs_ac, u32 njobs, u32 nwg_cc, const WinCcArgs w, const WinAcArgs a,
                  const u32 *__restrict__ n_flagged, u32 add_stride)
{
    const bool cc_half = blockIdx.x < nwg_cc;   // (uniform over the workgroup)
    if (n_flagged[cc_half ? 0 : 1] == 0) return;
    constexpr u32 LDS_DWORDS = SpLds<true>::TOTAL > AcLds::TOTAL ? SpLds<true>::TOTAL : AcLds::TOTAL;
    __shared__ __align__(16) u32 lds[LDS_DWORDS];
    if (cc_half)
        cc_sparse_body<true, DO_NCC, false, SpJobTable>(lds, blockIdx.x, jobs_cc, njobs, w.total_tiles, w.tiles_per_wg, w.c, w.lgG, w.slab,
                                                        w.tile_flags, w.plan, add_stride);
    else
        autocorr_edges_body<false, SpJobTable>(lds, blockIdx.x - nwg_cc, jobs_ac, njobs, a.total_tiles, a.tiles_per_wg, a.lgG, a.slab,
                                               a.tile_flags, a.plan, add_stride, w.c, a.add_shift, a.add_lag);
}

// ---------------------------------------------------------------------------------------------------
// Sparse-edge tiles: PAIR enumeration instead of windows.  A mappability track with long runs has a few dozen
// run edges per 64-Kbit tile; for those the window machinery above is all per-tile overhead (tiles of U / D in LDS,
// records, bit-sliced counters for ~50 entries).  EE(k) = P(k) - N(k) is just a histogram over edge PAIRS:
//     P(k) = #{(e1, e2): same sign, e2 - e1 = k},  N(k) likewise for opposite signs,  0 <= k <= max_lag
// (k = 0 pairs every edge with itself).  Per tile: edge words from the prefetched registers -> positions into ONE LDS
// list IN POSITION ORDER (block-wide scans of the per-thread edge counts; drivers = edges inside the tile, partners =
// drivers + the edges of the next max_lag bits) -> for every driver the following entries until the distance exceeds
// max_lag, one LDS atomic per hit, into histograms that live in LDS for the workgroup's whole tile range.
// Nothing else is staged, so six workgroups fit a CU and the pass runs close to the rate M streams from HBM.
// A tile with more than AP_CAP edges (dense-edge regions of real tracks) is NOT processed here: its flags are set and
// k_autocorr_edges, launched behind this kernel over the same tiles, picks up exactly the flagged ones.
#ifndef AP_NQ
#define AP_NQ 4                      // driver quads per thread: a pair tile is AP_NQ x 32 Kbit = AP_NQ / AC_NQ window-kernel tiles
#endif
#define AP_TBW (AP_NQ * SP_TBW)
#define AP_TB (AP_NQ * SP_TB)
#define AP_CAP 256u                  // list capacity = dense-tile threshold (edges of one pair tile + the lags above it)
#define AP_MAX_LAGS 8192u            // histograms of 2 x (max_lag + 1) u32 must fit in LDS beside the other workgroups' share
#ifndef AP_WAVES
#define AP_WAVES 5
#endif

struct ApRegs {
    uint4 m[AP_NQ], h;   // driver quads (quad q covers dwords q*1024 + 4 tid ..) + (threads 0..nh-1) a quad above the tile
    u32 below[AP_NQ], hbelow;
};

template <bool GUARD>
__device__ __forceinline__ void ap_fetch(ApRegs &ar, const u32 *__restrict__ M, int64_t d0, uint64_t nbits, u32 tid, u32 nh)
{
#pragma unroll
    for (int q = 0; q < AP_NQ; q++) {
        const int64_t j = d0 + (int64_t)q * SP_TBW + 4 * (int64_t)tid;
        ar.m[q] = ld_quad<GUARD>(M, j, nbits);
        ar.below[q] = GUARD ? ld_dword_guarded(M, j - 1, nbits) : M[j - 1];
    }
    ar.h = make_uint4(0, 0, 0, 0);
    ar.hbelow = 0;
    if (tid < nh) {
        const int64_t jh = d0 + AP_TBW + 4 * (int64_t)tid;
        ar.h = ld_quad<GUARD>(M, jh, nbits);
        ar.hbelow = GUARD ? ld_dword_guarded(M, jh - 1, nbits) : M[jh - 1];
    }
}

__device__ __forceinline__ void ap_fetch_job(ApRegs &ar, const SpJobDev &jb, u32 local_tile, u32 tid, u32 nh)
{
    const int64_t d0 = (int64_t)local_tile * AP_TBW;
    const uint64_t hi = (uint64_t)d0 + AP_TBW + 4 * (uint64_t)nh;
    const bool interior = jb.aligned16 && local_tile > 0 && hi + 2 <= jb.nbits / 32;
    if (interior)
        ap_fetch<false>(ar, jb.M, d0, jb.nbits, tid, nh);
    else
        ap_fetch<true>(ar, jb.M, d0, jb.nbits, tid, nh);
}

// edge word of each dword of a quad: bit i set iff M[i] != M[i-1] (`below` = the dword preceding m.x)
__device__ __forceinline__ uint4 edge_words(const uint4 m, u32 below)
{
    return make_uint4(m.x ^ __builtin_amdgcn_alignbit(m.x, below, 31), m.y ^ __builtin_amdgcn_alignbit(m.y, m.x, 31),
                      m.z ^ __builtin_amdgcn_alignbit(m.z, m.y, 31), m.w ^ __builtin_amdgcn_alignbit(m.w, m.z, 31));
}

__device__ __forceinline__ u32 popc4(const uint4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

// inclusive add-scan over the 64 lanes of a wave (DPP: four row shifts, two row broadcasts; bound_ctrl on the shifts lets
// the compiler fold each step into one v_add_u32_dpp instead of mov 0 / mov_dpp / add)
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Several scans at once, one v_add_u32_dpp per step and scan (the compiler leaves mov_dpp + add pairs in the event kernel).
// A DPP operand written by the previous VALU instruction needs two wait states: the steps of the other scans stand between
// the dependent ones; the s_nop in front covers an input written (or EXEC changed) by the instruction just before the block.
#define PMX_SCAN_STEP(ctl)                                                       \
    "v_add_u32_dpp %0, %0, %0 " ctl "\n\tv_add_u32_dpp %1, %1, %1 " ctl "\n\tv_add_u32_dpp %2, %2, %2 " ctl "\n\t"
#define PMX_SCAN_STEP2(ctl) "v_add_u32_dpp %3, %3, %3 " ctl "\n\tv_add_u32_dpp %4, %4, %4 " ctl "\n\t"
__device__ __forceinline__ void wave_inclusive_scan3(u32 &a, u32 &b, u32 &c)
{
    asm volatile("s_nop 4\n\t" PMX_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 PMX_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void wave_inclusive_scan5(u32 &a, u32 &b, u32 &c, u32 &d, u32 &e)
{
    asm volatile("s_nop 4\n\t" PMX_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") PMX_SCAN_STEP2("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1") PMX_SCAN_STEP2("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1") PMX_SCAN_STEP2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1") PMX_SCAN_STEP2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 PMX_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") PMX_SCAN_STEP2("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 PMX_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf") PMX_SCAN_STEP2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}

// list entry: bits 0..17 tile-relative position, bit 31 = falling edge (M is 0 at the edge position).  The four
// edge words of a quad are walked as two 64-bit words: two loops per quad instead of four (on sparse-edge tiles
// every loop costs its ~25 instructions of control flow whether or not one of the 64 lanes has a bit).
__device__ __forceinline__ void ap_emit(const uint4 e, const uint4 m, u32 idx, u32 base_bit, u32 *list)
{
    const u64 es[2] = {(u64)e.x | ((u64)e.y << 32), (u64)e.z | ((u64)e.w << 32)};
    const u64 ms[2] = {(u64)m.x | ((u64)m.y << 32), (u64)m.z | ((u64)m.w << 32)};
#pragma unroll
    for (u32 k = 0; k < 2; k++) {
        u64 ww = es[k];
        while (ww) {
            const u32 b = (u32)__builtin_ctzll(ww);
            ww &= ww - 1;
            list[idx] = (base_bit + 64u * k + b) | ((u32)((~ms[k] >> b) & 1ull) << 31);
            idx++;
        }
    }
}

// segment of a (workgroup, job) pair in the pair slab: [P: nl][N: nl][scalars: 16] u32
template <typename JT = SpJobTable>
__global__ void __launch_bounds__(256, AP_WAVES)
k_autocorr_pairs(const JT jobs, u32 njobs, u32 total_tiles, u32 tiles_per_wg, u32 max_lag, u32 nl, u32 nh,
                 u32 *__restrict__ slab, unsigned char *__restrict__ tile_flags, u32 *__restrict__ n_flagged)
{
    extern __shared__ __align__(16) u32 ap_lds[];
    u32 *const hP = ap_lds;                 // [nl]
    u32 *const hN = ap_lds + nl;            // [nl]
    u32 *const lists = hN + nl;             // [2][AP_CAP]: the tile's edges in POSITION order, parity alternates per tile
    u32 *const wtot = lists + 2 * AP_CAP;   // [2][4 waves][4]: per-wave edge counts (rows 0|1, rows 2|3, above the tile)
    u32 *const misc = wtot + 32;            // [0..7] scalar reduction when leaving a job

    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 g0 = blockIdx.x * tiles_per_wg;
    const u32 g1 = g0 + tiles_per_wg < total_tiles ? g0 + tiles_per_wg : total_tiles;
    if (g0 >= g1) return;
    for (u32 i = tid; i < 2 * nl; i += 256) ap_lds[i] = 0;

    FIRST_JOB(ji, JT, jobs, njobs, g0)
    u32 jn = ji;
    ApRegs ar;
    ap_fetch_job(ar, jobs.j[ji], g0 - jobs.j[ji].tile0, tid, nh);
    const size_t seg_stride = 2 * (size_t)nl + 16;
    u32 cntM = 0, cntU = 0, par = 0;
    __syncthreads();

    for (u32 g = g0; g < g1; g++) {
        // ---- edge words of this tile (and of the max_lag bits above it) from the prefetched registers ----
        uint4 E[AP_NQ], Eh;
        u32 c[AP_NQ], pendM = 0, pendU = 0;
#pragma unroll
        for (int q = 0; q < AP_NQ; q++) {
            E[q] = edge_words(ar.m[q], ar.below[q]);
            c[q] = popc4(E[q]);
            pendM += popc4(ar.m[q]);
            // run starts = rising edges; rising - falling = M[last bit of the quad] - M[bit before it]
            pendU += (c[q] + (ar.m[q].w >> 31) - (ar.below[q] >> 31)) >> 1;
        }
        Eh = edge_words(ar.h, ar.hbelow);   // zero registers (threads >= nh) give no edges
        const u32 ch = popc4(Eh);
        // ---- position order = (row q, thread, word, bit): exclusive offsets from three packed block scans ----
        const u32 s01 = wave_inclusive_scan(c[0] | (c[1] << 16));   // a row holds <= 32768 edges: 16-bit fields do not carry
        const u32 s23 = AP_NQ > 2 ? wave_inclusive_scan(c[AP_NQ > 2 ? 2 : 0] | (c[AP_NQ > 2 ? 3 : 0] << 16)) : 0u;
        const u32 sh = wave_inclusive_scan(ch);
        u32 *const wt = wtot + par * 16;
        if (lane == 63) {
            wt[wave * 4 + 0] = s01;
            wt[wave * 4 + 1] = s23;
            wt[wave * 4 + 2] = sh;
        }
        __syncthreads();   // Bs: per-wave totals visible (and every wave is done with the previous tile's pair search)
        u32 b01 = 0, b23 = 0, bh = 0, t01 = 0, t23 = 0, th = 0;
#pragma unroll
        for (u32 w = 0; w < 4; w++) {
            const u32 x01 = wt[w * 4 + 0], x23 = wt[w * 4 + 1], xh = wt[w * 4 + 2];
            if (w < wave) {
                b01 += x01;
                b23 += x23;
                bh += xh;
            }
            t01 += x01;
            t23 += x23;
            th += xh;
        }
        const u32 T0 = t01 & 0xffffu, T1 = t01 >> 16, T2 = AP_NQ > 2 ? t23 & 0xffffu : 0u, T3 = AP_NQ > 2 ? t23 >> 16 : 0u;
        const u32 nd = __builtin_amdgcn_readfirstlane(T0 + T1 + T2 + T3);   // drivers: edges inside the tile
        const u32 n = __builtin_amdgcn_readfirstlane(nd + th);             // + partners above it
        const bool dense = n > AP_CAP;
        u32 *const list = lists + par * AP_CAP;
        if (!dense) {
#ifdef AP_ABL_NOEMIT
            if (max_lag == 0xffffffffu) {
#else
            {
#endif
            const u32 e01 = b01 + s01 - (c[0] | (c[1] << 16));   // exclusive, per row
            ap_emit(E[0], ar.m[0], e01 & 0xffffu, 0 * SP_TB + 128u * tid, list);
            ap_emit(E[1], ar.m[1], T0 + (e01 >> 16), 1 * SP_TB + 128u * tid, list);
            if (AP_NQ > 2) {
                const u32 e23 = b23 + s23 - (c[AP_NQ > 2 ? 2 : 0] | (c[AP_NQ > 2 ? 3 : 0] << 16));
                ap_emit(E[AP_NQ > 2 ? 2 : 0], ar.m[AP_NQ > 2 ? 2 : 0], T0 + T1 + (e23 & 0xffffu), 2 * SP_TB + 128u * tid, list);
                ap_emit(E[AP_NQ > 2 ? 3 : 0], ar.m[AP_NQ > 2 ? 3 : 0], T0 + T1 + T2 + (e23 >> 16), 3 * SP_TB + 128u * tid, list);
            }
            ap_emit(Eh, ar.h, nd + bh + sh - ch, AP_TB + 128u * tid, list);
            }
            cntM += pendM;
            cntU += pendU;
        } else if (tid == 0) {
            // dense tile: left to the window kernel (which also counts its set bits and run starts): both of its tiles
            const u32 f = jobs.j[ji].flag0 + (AP_NQ / AC_NQ) * (g - jobs.j[ji].tile0);
            for (u32 i = 0; i < AP_NQ / AC_NQ; i++) tile_flags[f + i] = 1;   // (the flag array is padded per job)
            atomicAdd(n_flagged, 1u);
        }
        // ---- prefetch the next tile into the (now free) registers ----
        if (g + 1 < g1) {
            if (g + 1 >= jobs.j[ji].tile0 + jobs.j[ji].ntiles) jn = ji + 1;
            ap_fetch_job(ar, jobs.j[jn], g + 1 - jobs.j[jn].tile0, tid, nh);
        }
        __syncthreads();   // B1: the list is complete
        par ^= 1;
#ifdef AP_ABL_NOPAIRS
        if (max_lag == 0xffffffffu)
#else
        if (!dense)
#endif
        {
            // ---- pairs: the list is sorted, so the partners of entry i are i, i+1, ... until the distance exceeds
            // max_lag.  Wave w takes the offsets o = w, w+4, ...; lanes take the drivers; a wave stops at the first
            // offset without a hit (larger offsets only increase every distance).
            u32 pi[AP_CAP / 64];
#pragma unroll
            for (u32 h = 0; h < AP_CAP / 64; h++) pi[h] = lane + 64 * h < nd ? list[lane + 64 * h] : 0u;
            for (u32 o = wave; o < n; o += 4) {
                bool any = false;
#pragma unroll
                for (u32 h = 0; h < AP_CAP / 64; h++) {
                    const u32 i = lane + 64 * h;
                    if (64 * h < nd && i < nd && i + o < n) {
                        const u32 ej = list[i + o];
                        const u32 k = (ej & 0x3ffffu) - (pi[h] & 0x3ffffu);
                        if (k <= max_lag) {
                            atomicAdd(((pi[h] ^ ej) >> 31) ? &hN[k] : &hP[k], 1u);
                            any = true;
                        }
                    }
                }
                if (!__builtin_amdgcn_readfirstlane(__ballot(any) != 0)) break;
            }
        }
        const bool leaving = jn != ji || g + 1 == g1;
        if (leaving) {
            // histograms + scalars of this (workgroup, job) -> its slab segment; cleared for the next job
            __syncthreads();
            u32 *seg = slab + (size_t)(blockIdx.x + ji) * seg_stride;
            for (u32 i = tid; i < 2 * nl; i += 256) {
                seg[i] = ap_lds[i];
                ap_lds[i] = 0;
            }
            u32 vm = cntM, vu = cntU;
            for (int off = 32; off > 0; off >>= 1) {
                vm += __shfl_down(vm, off, 64);
                vu += __shfl_down(vu, off, 64);
            }
            if (lane == 0) {
                misc[wave] = vm;
                misc[4 + wave] = vu;
            }
            __syncthreads();
            if (tid == 0) {
                seg[2 * nl + 0] = misc[0] + misc[1] + misc[2] + misc[3];   // popcount(M) of my (non-dense) tiles
                seg[2 * nl + 1] = misc[4] + misc[5] + misc[6] + misc[7];   // runs starting in them
            }
            cntM = 0;
            cntU = 0;
            __syncthreads();
        }
        ji = jn;
    }
}

// out2[row][i] (+)= sum over the workgroups that touched the job of their pair-slab segments (see k_reduce_segments):
// 32 consecutive elements x 8 workgroup phases per block
template <typename JT = SpJobTable>
__global__ void __launch_bounds__(256)
k_reduce_pairs(const u32 *__restrict__ slab, const JT jobs, u32 nl, u32 max_lag, u32 lagcap, u32 accumulate)
{
    __shared__ u64 part[8][32];
    const u32 job = blockIdx.z, row = blockIdx.y;          // row 0: P, 1: N, 2: scalars
    const SpJobDev &jb = jobs.j[job];
    const u32 e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const u32 n = row == 2 ? 2u : max_lag + 1;
    const size_t seg_stride = 2 * (size_t)nl + 16;
    for (u32 i0 = blockIdx.x * 32; i0 < n; i0 += gridDim.x * 32) {   // uniform trip count over the block
        const u32 i = i0 + e;
        u64 sum = 0;
        if (i < n) {
            const u32 *p = slab + (row == 2 ? 2 * (size_t)nl : (size_t)row * nl) + i;
            for (u32 w = jb.wg_first + g; w <= jb.wg_last; w += 8) sum += p[(size_t)(w + job) * seg_stride];
        }
        part[g][e] = sum;
        __syncthreads();
        if (g == 0 && i < n) {
            u64 t = 0;
#pragma unroll
            for (u32 k = 0; k < 8; k++) t += part[k][e];
            u64 *dst = jb.out2 + (row == 2 ? 2 * (size_t)lagcap : (size_t)row * lagcap) + i;
            *dst = accumulate ? *dst + t : t;
        }
        __syncthreads();
    }
}

// A(k) from EE(k) = P(k) - N(k), one block per chromosome.  A(k+1) = 2 A(k) - A(k-1) - EE(k) is a double prefix sum:
// with Delta(k) = A(k+1) - A(k):  Delta(0) = -#runs, Delta(k) = Delta(k-1) - EE(k);  A(k) = A(0) + sum_{i<k} Delta(i).
// Per-job scratch out2 (u64): P[lagcap], N[lagcap], scalars[16], A[lagcap].  All arithmetic in signed 64-bit.
// mode 0: out[k] = A(k), k = 0..max_lag.  mode 1: out is a result block: row MLEN[d] = A(|c - d|), d = 0..max_shift,
// and scalar [2] = popcount(M).
__device__ __forceinline__ long long block_exclusive_offset(long long local_sum, long long *part, u32 tid)
{
    // inclusive scan over the wave (shuffles), wave totals through LDS; ends with a barrier (part may be reused)
    long long x = local_sum;
    const u32 lane = tid & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long y = __shfl_up(x, off, 64);
        if (lane >= (u32)off) x += y;
    }
    if (lane == 63) part[tid >> 6] = x;
    __syncthreads();
    long long base = 0;
    for (u32 w = 0; w < (tid >> 6); w++) base += part[w];
    __syncthreads();
    return base + x - local_sum;
}

__device__ __forceinline__ void autocorr_finish_job(const SpJobDev &jb, long long *part, u32 max_lag, u32 lagcap, u32 mode,
                                                    int32_t c, u32 max_shift, u32 out_stride, u32 nthreads = 256)
{
    const u64 *P = jb.out2, *N = jb.out2 + lagcap, *scal = jb.out2 + 2 * (size_t)lagcap;
    long long *A = reinterpret_cast<long long *>(jb.out2 + 2 * (size_t)lagcap + 16);
    const u32 tid = threadIdx.x;
    const long long a0 = (long long)scal[0], runs = (long long)scal[1];
    const u32 seg = (max_lag + 1 + nthreads - 1) / nthreads;
    const u32 k0 = tid * seg, k1 = (k0 + seg < max_lag + 1) ? k0 + seg : max_lag + 1;
    // pass 1: Delta(k) = inclusive prefix of x(k), x(0) = -runs, x(k) = -(P[k] - N[k])
    long long sum = 0;
    for (u32 k = k0; k < k1; k++) sum += k == 0 ? -runs : (long long)N[k] - (long long)P[k];
    long long run = block_exclusive_offset(sum, part, tid);
    for (u32 k = k0; k < k1; k++) {
        run += k == 0 ? -runs : (long long)N[k] - (long long)P[k];
        A[k] = run;   // Delta(k)
    }
    __syncthreads();
    // pass 2: A(k) = a0 + exclusive prefix of Delta (in place: each thread owns its segment)
    sum = 0;
    for (u32 k = k0; k < k1; k++) sum += A[k];
    run = a0 + block_exclusive_offset(sum, part, tid);
    for (u32 k = k0; k < k1; k++) {
        const long long d = A[k];
        A[k] = run;
        run += d;
    }
    __syncthreads();
    if (mode == 0) {
        for (u32 k = tid; k <= max_lag; k += nthreads) jb.out[k] = (u64)A[k];
    } else {
        // mode 2: ADDED to the row (the recurrence is linear in (popcount(M), runs, EE): the window kernel's share of a
        // chromosome on top of what the event pass has written, k_events_tail)
        u64 *sc2 = jb.out + (size_t)PMX_ROW_SCALARS * out_stride + 2;
        if (tid == 0) *sc2 = mode == 2 ? *sc2 + (u64)a0 : (u64)a0;
        u64 *dst = jb.out + (size_t)PMX_ROW_MLEN * out_stride;
        for (u32 d = tid; d <= max_shift; d += nthreads) {
            const int32_t k = c - (int32_t)d;
            const u64 v = (u64)A[k < 0 ? -k : k];
            dst[d] = mode == 2 ? dst[d] + v : v;
        }
    }
}

#define AC_FINISH_THREADS 1024u   // (256 until round 4: 42 us for config 5's 200 chromosomes x 4901 lags -- two dependent passes of 20 entries per thread)
template <typename JT = SpJobTable>
__global__ void __launch_bounds__(AC_FINISH_THREADS)
k_autocorr_finish(const JT jobs, u32 max_lag, u32 lagcap, u32 mode, int32_t c, u32 max_shift, u32 out_stride)
{
    __shared__ long long part[256];
    const SpJobDev &jb = jobs.j[blockIdx.x];
    if (!(jb.flags & 1u)) return;   // one finish per chromosome (its chunk-0 job)
    autocorr_finish_job(jb, part, max_lag, lagcap, mode, c, max_shift, out_stride, AC_FINISH_THREADS);
}

// Work split of a window kernel that runs BEHIND the event / pair kernel and only sees the tiles it flagged.  Equal shares
// of the tile sequence would leave the flagged stretches (a read-dense or edge-dense region of a chromosome; a range a
// workgroup of the event kernel handed over) to the one or two workgroups whose share they fall into, with everybody else
// idle: the window pass then took 70 % of its full-genome time for 10 % of the tiles.  This kernel (one block) cuts the
// tile sequence into nwg contiguous ranges with equal numbers of FLAGGED tiles instead:
//   plan[2 w], plan[2 w + 1]    range [start, end) of workgroup w in the launch's global tile sequence
//   plan[PLAN_LIST + i]         the workgroups with a non-empty range, in order
//   plan[PLAN_JOBWG + 2 job..]  entries [first, end) of that list whose ranges meet the job's tiles (k_events_tail sums their
//                               slab segments; a workgroup with an empty range wrote none)
// Contiguous ranges keep the window kernels' per-(workgroup, job) bookkeeping as it is.
#define PLAN_JOBWG 4096u                      // [2 x SP_MAXJOBS] per job: first / last index into the list below
#define PLAN_LIST (PLAN_JOBWG + 2 * SP_MAXJOBS)   // the workgroups with a non-empty range, in order
#define PLAN_WORDS (PLAN_LIST + 2048u)
// The flag array is scanned in its RAW layout (job i's tiles at [flag0_i, flag0_i + ntiles_i) + padding): 16 flags per thread
// and pass with one 16-byte load, a block scan per pass.
__device__ __forceinline__ u32 plan_block_scan(u32 v, u32 *part, u32 tid, u32 *total)   // exclusive; ends with a barrier
{
    u32 x = v;
    const u32 lane = tid & 63, wave = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off, 64);
        if (lane >= (u32)off) x += y;
    }
    if (lane == 63) part[wave] = x;
    __syncthreads();
    u32 base = x - v, tot = 0;
    for (u32 w = 0; w < 16; w++) {
        if (w < wave) base += part[w];
        tot += part[w];
    }
    *total = tot;
    __syncthreads();
    return base;
}

// what the planner needs of a launch: three words per job + the launch's shape (two of these fit the kernel arguments of
// ONE launch: block 0 plans the cross-correlation window launch, block 1 the autocorrelation one)
struct PlanLaunch {
    u32 flag0[SP_MAXJOBS], tile0[SP_MAXJOBS], ntiles[SP_MAXJOBS];
    u32 njobs, total_tiles, nwg, raw_lo, raw_hi, flags_per_count;
    const unsigned char *flags;
    const u32 *n_flagged;
    u32 *plan;   // nullptr: no such launch
};

__device__ __forceinline__ void plan_flagged(const PlanLaunch &pl)
{
    const u32 njobs = pl.njobs, total_tiles = pl.total_tiles, nwg = pl.nwg, raw_lo = pl.raw_lo, raw_hi = pl.raw_hi;
    const u32 flags_per_count = pl.flags_per_count;
    const unsigned char *__restrict__ flags = pl.flags;
    const u32 *__restrict__ n_flagged = pl.n_flagged;
    u32 *__restrict__ plan = pl.plan;
    if (!plan) return;
    // [raw_lo, raw_hi): the flag entries of THIS launch's jobs (a batch of more than SP_MAXJOBS chromosomes is several
    // launches over one flag array; n_flagged is zeroed per launch)
    // flags set in the array: the producer counts ITS tiles, each of which sets flags_per_count entries (an event tile is two
    // tiles of k_cc_sparse; the second may be a padding entry behind the job's last tile: it maps to the next job's first tile,
    // a valid cut point like any other)
    const u32 F = *n_flagged * flags_per_count;
    if (F == 0) return;            // nothing flagged: the window kernel returns at once and nobody reads the plan
    __shared__ u32 part[16];
    __shared__ u32 jflag0[SP_MAXJOBS + 1], jtile0[SP_MAXJOBS];
    __shared__ u32 sb[2049];       // first flagged tile of every workgroup (nwg <= 2048)
    __shared__ u32 wl[2048];       // the workgroups with a non-empty range ...
    __shared__ u32 se[2048];       // ... and where their ranges end
    const u32 tid = threadIdx.x;
    if (tid < njobs) {
        jflag0[tid] = pl.flag0[tid];
        jtile0[tid] = pl.tile0[tid];
    }
    if (tid == 0) jflag0[njobs] = 0xffffffffu;
    for (u32 w = tid; w <= nwg; w += 1024) sb[w] = total_tiles;   // (a start nobody writes -- fewer flags set than counted -- is an empty range)
    __syncthreads();
    u32 running = 0;               // flagged entries before this pass (uniform)
    for (u32 f0 = raw_lo & ~15u; f0 < raw_hi; f0 += 1024 * 16) {
        const u32 f = f0 + 16 * tid;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (f < raw_hi) q = *reinterpret_cast<const uint4 *>(flags + f);   // (the array is padded to 16 bytes)
        if (f < raw_lo || f + 16 > raw_hi) {                               // entries of other launches' jobs in my 16 bytes
            u32 qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (u32 k = 0; k < 16; k++)
                if (f + k < raw_lo || f + k >= raw_hi) qq[k >> 2] &= ~(0xffu << (8 * (k & 3)));
            q = make_uint4(qq[0], qq[1], qq[2], qq[3]);
        }
        // (flags are 0 / 1) one bit per flag: the four low bits of the bytes of a dword gathered by a multiplication
        const u32 m16 = (((q.x & 0x01010101u) * 0x01020408u) >> 24) | ((((q.y & 0x01010101u) * 0x01020408u) >> 24) << 4) |
                        ((((q.z & 0x01010101u) * 0x01020408u) >> 24) << 8) | ((((q.w & 0x01010101u) * 0x01020408u) >> 24) << 12);
        const u32 cnt = __popc(m16);
        u32 tot;
        const u32 rank = running + plan_block_scan(cnt, part, tid, &tot);
        running += tot;
        if (cnt && F < nwg) {
            // fewer flagged tiles than workgroups: the flagged tile of rank r is workgroup r's, alone (see the ranges below)
            u32 m = m16, r = rank;
            while (m) {
                const u32 k = (u32)__builtin_ctz(m);
                m &= m - 1;
                u32 ji = 0;
                while (jflag0[ji + 1] <= f + k) ji++;
                if (r < nwg) sb[r] = jtile0[ji] + (f + k - jflag0[ji]);   // raw flag index -> tile of the launch's sequence
                r++;
            }
        } else if (cnt) {
            // workgroup w takes the flagged tiles of rank [w F / nwg, (w + 1) F / nwg): its range starts AT the first of them.
            // The workgroups whose first tile is one of my cnt entries (ranks [rank, rank + cnt)): three divisions per thread,
            // then floor(w F / nwg) step by step (a division per entry took 100 us: 64-bit divisions are long dependent VALU
            // sequences and one block does all of this)
            const u32 w_lo = (u32)(((u64)rank * nwg + F - 1) / F), w_hi = (u32)(((u64)(rank + cnt) * nwg + F - 1) / F);
            u64 prod = (u64)w_lo * F;
            u32 quo = (u32)(prod / nwg), rem = (u32)(prod - (u64)quo * nwg);
            for (u32 w = w_lo; w < w_hi && w < nwg; w++) {
                u32 m = m16;             // my (quo - rank)-th set flag (0-based): drop the lower ones
                for (u32 r = quo - rank; r > 0 && m; r--) m &= m - 1;
                if (m) {                                       // (always, while F is the number of flags set)
                    const u32 k = (u32)__builtin_ctz(m);
                    u32 ji = 0;
                    while (jflag0[ji + 1] <= f + k) ji++;
                    sb[w] = jtile0[ji] + (f + k - jflag0[ji]);
                }
                u64 nr = (u64)rem + F;                         // (w + 1) F = quo' nwg + rem'
                while (nr >= nwg) {
                    nr -= nwg;
                    quo++;
                }
                rem = (u32)nr;
            }
        }
    }
    __syncthreads();
    if (tid == 0) sb[nwg] = total_tiles;
    __syncthreads();
    // ranges [start, end) per workgroup.  F >= nwg: a range runs up to the next one's start.  Fewer flagged tiles than
    // workgroups: a workgroup has ONE tile or none, and its range is that tile alone -- otherwise it would walk the flags of
    // thousands of tiles up to the next flagged one (0.3 ms for half a dozen flagged tiles).  Then the workgroups that have
    // tiles, compacted (an empty range wrote no slab segment).
    u32 nlive = 0;
    for (u32 w0 = 0; w0 < nwg; w0 += 1024) {
        const u32 w = w0 + tid;
        u32 start = 0, end = 0, live = 0;
        if (w < nwg) {
            start = sb[w];
            live = start < sb[w + 1] ? 1u : 0u;
            end = F < nwg ? start + live : sb[w + 1];
            plan[2 * w] = start;
            plan[2 * w + 1] = end;
        }
        u32 tot;
        const u32 idx = nlive + plan_block_scan(live, part, tid, &tot);
        if (live) {
            wl[idx] = w;
            se[idx] = end;
            plan[PLAN_LIST + idx] = w;
        }
        nlive += tot;
    }
    __syncthreads();
    if (tid < njobs) {
        // list entries whose range meets the job's tiles [a, b): starts and ends increase along the list
        const u32 a = pl.tile0[tid], b = a + pl.ntiles[tid];
        u32 lo = 0, hi = nlive;                 // first i with end_i > a
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (se[mid] > a) hi = mid;
            else lo = mid + 1;
        }
        const u32 first = lo;
        hi = nlive;                             // first i with start_i >= b
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (sb[wl[mid]] >= b) hi = mid;
            else lo = mid + 1;
        }
        plan[PLAN_JOBWG + 2 * tid] = first;     // entries [first, end) of the list
        plan[PLAN_JOBWG + 2 * tid + 1] = lo;
    }
}

__global__ void __launch_bounds__(1024) k_plan_flagged(const PlanLaunch a, const PlanLaunch b)
{
    if (blockIdx.x == 0) plan_flagged(a);
    else plan_flagged(b);
}

#include "kernels_events.h"

// ---- host side ----------------------------------------------------------------------------------------------

static int launch_autocorr_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_lag, uint32_t mode, uint32_t read_len,
                                 uint32_t max_shift, uint32_t out_stride, const uint32_t *pre_flag0, const unsigned char *pre_flags,
                                 const u32 *pre_nflagged);

static uint32_t lg_slot_lanes(uint32_t nshifts)   // nshifts = shifts handled per slot
{
    const u32 need = (nshifts + 31) / 32;
    u32 lg = 2;   // G >= 4 keeps the record padding granule (16 * 64 / G records) within one workgroup pass
    while ((1u << lg) < need) lg++;
    return lg;
}

static inline bool is_aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

int pmx_sparse_supported(uint32_t max_shift, uint32_t read_len)
{
    return max_shift >= 3 && max_shift <= 65535 && read_len >= 1 && read_len <= 1024;
}

uint32_t pmx_sparse_max_jobs(void) { return SP_MAXJOBS; }

// one launch's worth of (chromosome, shift chunk) jobs
struct VJob {
    const pmx_job *job;
    u32 d_off, d_n;
    u32 flag0;   // autocorrelation: index of the chromosome's first tile in the dense-tile flag array
    bool ranged; // the event launch of a tile-range job: take job->tile_first / tile_count (every other launch sees the whole chromosome)
};

static void fill_plan_launch(PlanLaunch &p, const SpJobTable &tab, u32 n, u32 total, u32 nwg, u32 raw_lo, u32 raw_hi,
                             u32 flags_per_count, const unsigned char *flags, const u32 *n_flagged, u32 *plan)
{
    for (u32 i = 0; i < n; i++) {
        p.flag0[i] = tab.j[i].flag0;
        p.tile0[i] = tab.j[i].tile0;
        p.ntiles[i] = tab.j[i].ntiles;
    }
    p.njobs = n; p.total_tiles = total; p.nwg = nwg; p.raw_lo = raw_lo; p.raw_hi = raw_hi;
    p.flags_per_count = flags_per_count; p.flags = flags; p.n_flagged = n_flagged; p.plan = plan;
}

// Cuts the global tile sequence of a launch into per-workgroup ranges and fills the device job table.
static void plan_launch(pmx_ctx *ctx, const VJob *vj, uint32_t n, bool autocorr, uint32_t wg_per_cu, SpJobDev *out,
                        uint32_t *total_tiles, uint32_t *tiles_per_wg, uint32_t *nwg, uint64_t tile_bits_override = 0)
{
    uint32_t t = 0;
    for (uint32_t i = 0; i < n; i++) {
        SpJobDev &d = out[i];
        const pmx_job &jb = *vj[i].job;
        const uint64_t bits = jb.nbits + (autocorr ? 1 : 0);   // edges live on [0, nbits]
        const uint64_t tile_bits = tile_bits_override ? tile_bits_override : (autocorr ? AC_TB : SP_TB);
        uint64_t nt = (bits + tile_bits - 1) / tile_bits;
        if (nt < 1) nt = 1;
        uint32_t first = 0;
        if (vj[i].ranged && jb.tile_count) {   // (validated by the caller: inside the chromosome, event tiles)
            first = jb.tile_first;
            nt = jb.tile_count;
        }
        d.F = (const u32 *)jb.d_F;
        d.R = (const u32 *)jb.d_R;
        d.M = (const u32 *)jb.d_M;
        d.nbits = jb.nbits;
        d.tile0 = t;
        d.ntiles = (u32)nt;
        d.aligned16 = autocorr ? is_aligned16(jb.d_M)
                               : (is_aligned16(jb.d_F) && is_aligned16(jb.d_R) && (!jb.d_M || is_aligned16(jb.d_M)));
        d.flags = vj[i].d_off == 0 ? 1u : 0u;
        d.d_off = vj[i].d_off;
        d.d_n = vj[i].d_n;
        d.out = (u64 *)jb.d_out;
        d.out2 = (u64 *)jb.d_out2;
        d.flag0 = vj[i].flag0;
        d.tile_first = first;
        t += (u32)nt;
    }
    uint64_t want = (uint64_t)ctx->num_cus * wg_per_cu;
    if (ctx->debug_max_wg && want > ctx->debug_max_wg) want = ctx->debug_max_wg;   // (tests: pmx_debug_set_max_workgroups)
    if (want > t) want = t;
    if (want < 1) want = 1;
    const uint32_t tpw = (uint32_t)((t + want - 1) / want);
    const uint32_t nw = (t + tpw - 1) / tpw;
    for (uint32_t i = 0; i < n; i++) {
        SpJobDev &d = out[i];
        d.wg_first = d.tile0 / tpw;
        d.wg_last = (d.tile0 + d.ntiles - 1) / tpw;
    }
    *total_tiles = t;
    *tiles_per_wg = tpw;
    *nwg = nw;
}

static void plan_launch(pmx_ctx *ctx, const VJob *vj, uint32_t n, bool autocorr, uint32_t wg_per_cu, SpJobTable *tab,
                        uint32_t *total_tiles, uint32_t *tiles_per_wg, uint32_t *nwg, uint64_t tile_bits_override = 0)
{
    plan_launch(ctx, vj, n, autocorr, wg_per_cu, tab->j, total_tiles, tiles_per_wg, nwg, tile_bits_override);
}

// a launch's job table in device memory (any number of jobs)
static int upload_table(pmx_ctx *ctx, const std::vector<SpJobDev> &v, SpJobTableRef *ref)
{
    const void *d = nullptr;
    int rc = pmx_upload_jobtab(ctx, v.data(), v.size() * sizeof(SpJobDev), &d);
    ref->j = (const SpJobDev *)d;
    return rc;
}
#define SP_MAXJOBS_REF 2048u   // jobs per launch with a device-side table (bounds the slab: one segment per job beside the workgroups')

// (chromosome x chunk-of-1024-shifts) jobs of a batch, in launches of at most SP_MAXJOBS
static void expand_chunks(const pmx_job *jobs, uint32_t njobs, uint32_t nshifts, std::vector<VJob> &out,
                          const uint32_t *flag0 = nullptr)
{
    for (uint32_t i = 0; i < njobs; i++)
        for (uint32_t off = 0; off < nshifts; off += 1024) {
            VJob v;
            v.ranged = false;
            v.job = &jobs[i];
            v.d_off = off;
            v.d_n = nshifts - off < 1024 ? nshifts - off : 1024;
            v.flag0 = flag0 ? flag0[i] : 0;
            out.push_back(v);
        }
}

static bool events_enabled()
{
    // PMX_CC_EVENTS=0 in the environment keeps everything on the window kernel (A/B measurements, tests)
    static const bool on = [] {
        const char *e = getenv("PMX_CC_EVENTS");
        return !(e && e[0] == '0');
    }();
    return on;
}

// ---- the event kernel beyond 1023 shifts (BIG instantiations) ----
struct EvBigPlan {
    u32 hn, lo, nsg, lds_bytes, wg_per_cu, hi;
    u32 ee_lds;   // the row of lags of the fused mappable-length pairs fits into LDS behind the sub-groups' blocks
};

static bool events_big_enabled()
{
    // PMX_CC_EVENTS_BIG=0 in the environment keeps max_shift > 1023 on the window kernel in shift chunks (A/B, tests)
    static const bool on = [] {
        const char *e = getenv("PMX_CC_EVENTS_BIG");
        return !(e && e[0] == '0');
    }();
    return on;
}

// Geometry of a BIG launch: histogram length, staged M below a tile, and the number of sub-groups per workgroup that puts
// the most wavefronts on a CU (16 at 128 VGPRs; ties go to the smaller workgroup: more independent phase groups).
// PMX_EV_NSG=1|2|4 in the environment forces the sub-group count (A/B).
// fused_lag != 0: the mappable-length pairs are taken by this launch too, so the run edges up to fused_lag bits ABOVE a tile
// must be in its list: that many more dwords of M are staged above it (EV_HI otherwise); one wavefront loads them (<= 64 quads)
template <bool HAS_M>
static bool ev_big_plan(uint32_t max_shift, EvBigPlan *p, uint32_t fused_lag = 0)
{
    typedef EvLds<HAS_M, true> L;
    const u32 hn = (max_shift + 1 + 127) / 128 * 128;
    const u32 lo = ((2 * max_shift + 31) / 32 + 3) / 4 * 4;
    u32 hi = EV_HI;
    if (fused_lag) {
        const u32 need = ((fused_lag + 31) / 32 + 2 + 3) / 4 * 4;
        hi = need > EV_HI ? need : EV_HI;
        if (hi > 256u) return false;
    }
    static const int forced = [] {
        const char *e = getenv("PMX_EV_NSG");
        return e ? atoi(e) : 0;
    }();
    const u32 lds_cu = 160u * 1024u;
    u32 best_waves = 0;
    for (u32 nsg = 1; nsg <= 4; nsg *= 2) {
        if (forced && (u32)forced != nsg) continue;
        const u32 bytes = L::total(hn, lo, nsg, hi) * 4 + 16;   // + the static stub
        if (bytes > lds_cu) continue;
        u32 per_cu = lds_cu / (bytes + 256);                // (allocation granularity)
        if (per_cu * nsg > 4) per_cu = 4 / nsg;
        if (per_cu < 1) continue;
        const u32 waves = per_cu * nsg * 4;
        if (waves > best_waves) {
            best_waves = waves;
            p->hn = hn;
            p->lo = lo;
            p->hi = hi;
            p->nsg = nsg;
            p->lds_bytes = bytes - 16;
            p->wg_per_cu = per_cu;
            p->ee_lds = 0;
        }
    }
    // the row of lags of the fused mappable-length pairs (hn 16-bit cells) in LDS, if that costs no resident workgroup
    // (PMX_EV_EE_LDS=0 in the environment: always in the slab -- A/B, tests)
    static const bool ee_ok = [] {
        const char *e = getenv("PMX_EV_EE_LDS");
        return !(e && e[0] == '0');
    }();
    if (best_waves && fused_lag && ee_ok) {
        const u32 bytes = p->lds_bytes + 16 + hn * 2;
        u32 per_cu = bytes <= lds_cu ? lds_cu / (bytes + 256) : 0;
        if (per_cu * p->nsg > 4) per_cu = 4 / p->nsg;
        if (per_cu >= p->wg_per_cu) {
            p->lds_bytes = bytes - 16;
            p->ee_lds = 1;
        }
    }
    static const bool dbg = getenv("PMX_DEBUG_PLAN") != nullptr;
    if (dbg && best_waves)
        fprintf(stderr, "[ev_big_plan] max_shift %u fused_lag %u: hn %u lo %u hi %u nsg %u lds %u B x %u per CU, row of lags in LDS: %u\n", max_shift,
                fused_lag, p->hn, p->lo, p->hi, p->nsg, p->lds_bytes, p->wg_per_cu, p->ee_lds);
    return best_waves != 0;
}

// resident workgroups per CU of a max_shift <= 1023 instantiation on this device, at most `built_for` (asked once per instantiation)
// (cached per DEVICE: a process may hold one context per GPU, and both the occupancy answer and the function attribute of
// ev_big_launch belong to the current device)
#define EV_MAX_DEVICES 64
template <bool HAS_M, bool DO_NCC, bool DO_MLEN, bool DEEP>
static u32 ev_resident_per_cu(pmx_ctx *ctx, u32 built_for)
{
    static int cached[EV_MAX_DEVICES];
    static bool known[EV_MAX_DEVICES];
    const int dev = ctx->device >= 0 && ctx->device < EV_MAX_DEVICES ? ctx->device : -1;
    int n = dev >= 0 && known[dev] ? cached[dev] : -1;
    if (n < 0) {
        auto kern = k_cc_events<HAS_M, DO_NCC, DO_MLEN, 1, false, SpJobTable, DEEP>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kern), 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = (int)built_for;
        }
        if (dev >= 0) {
            cached[dev] = n;
            known[dev] = true;
        }
    }
    return (u32)n < built_for ? (u32)n : built_for;
}

template <bool HAS_M, bool DO_NCC, u32 NSG, bool DO_MLEN = false>
static int ev_big_launch(pmx_ctx *ctx, const EvBigPlan &pl, const SpJobTableRef &tab, u32 n, u32 total, u32 tpw, u32 nwg, u32 c,
                         u32 max_shift, u32 nhr, unsigned char *d_flags, u32 *d_nflagged, u32 *d_jobstat, u32 fused_lag = 0,
                         unsigned char *d_flags_ac = nullptr)
{
    auto kern = k_cc_events<HAS_M, DO_NCC, DO_MLEN, NSG, true, SpJobTableRef>;
    static bool attr_set[EV_MAX_DEVICES];   // (per device function AND device: set once per GPU of the process, see ev_resident_per_cu)
    const int dev = ctx->device >= 0 && ctx->device < EV_MAX_DEVICES ? ctx->device : -1;
    if (dev < 0 || !attr_set[dev]) {
        PMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 64));
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256 * NSG), pl.lds_bytes, ctx->stream, tab, n, total, tpw, c, max_shift, nhr, fused_lag,
                       pl.hn, pl.lo, pl.hi | (DO_MLEN && pl.ee_lds ? 1u << 16 : 0u), ctx->d_slab, d_flags, d_flags_ac ? d_flags_ac : d_flags, d_nflagged,
                       d_jobstat);
    PMX_CHECK_LAUNCH("k_cc_events (max_shift > 1023)");
    return PMX_OK;
}

// The window kernel in chunks of 1024 shifts over (chromosome x chunk) jobs with a device-side job table: every job of the
// batch in launches of SP_MAXJOBS_REF (BASELINE config 5: 1000 jobs, one launch).  d_flags / d_nflagged: only the tiles the
// event kernel flagged (the launch returns at once when there are none) and the sums are ADDED by a gated reduce;
// null: every tile, rows written.
static int launch_cc_window_chunks(pmx_ctx *ctx, const std::vector<VJob> &vjobs, bool has_m, bool do_ncc, int32_t c,
                                   const ReduceSpec &rs, u32 nr, u32 nz, const unsigned char *d_flags, const u32 *d_nflagged)
{
    const bool behind_events = d_flags != nullptr;
    for (size_t lo = 0; lo < vjobs.size(); lo += SP_MAXJOBS_REF) {
        const uint32_t n = (uint32_t)(vjobs.size() - lo < SP_MAXJOBS_REF ? vjobs.size() - lo : SP_MAXJOBS_REF);
        std::vector<SpJobDev> tab(n);
        memset(tab.data(), 0, n * sizeof(SpJobDev));
        uint32_t total, tpw, nwg;
        plan_launch(ctx, &vjobs[lo], n, false, has_m ? SP_WAVES_CH : SP_WAVES_CH_NCC, tab.data(), &total, &tpw, &nwg);
        const size_t wwords = (size_t)(nwg + n) * SP_SEG_ROWS * 1024 + (size_t)nwg * 4 * 12 * 2 + 64;
        int rc = behind_events ? pmx_ensure_slab_fb(ctx, wwords) : pmx_ensure_slab(ctx, wwords);
        if (rc) return rc;
        u32 *const wslab = behind_events ? ctx->d_slab_fb : ctx->d_slab;
        SpJobTableRef ref;
        rc = upload_table(ctx, tab, &ref);
        if (rc) return rc;
        pmx_timed_launch tl;
        rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_SPARSE, &tl, behind_events);
        if (rc) return rc;
#define SP_LAUNCH(HM, NC)                                                                                                   \
    hipLaunchKernelGGL((k_cc_sparse<HM, NC, true, SpJobTableRef>), dim3(nwg), dim3(256), 0, ctx->stream, ref, n, total, tpw, c, 5u, \
                       wslab, d_flags, d_nflagged, (const u32 *)nullptr)
        if (has_m && do_ncc) SP_LAUNCH(true, true);
        else if (has_m) SP_LAUNCH(true, false);
        else SP_LAUNCH(false, true);
#undef SP_LAUNCH
        PMX_CHECK_LAUNCH("k_cc_sparse");
        rc = pmx_prof_end(ctx, &tl);
        if (rc) return rc;
        // (behind the event pass the launch is gated: normally 25 k blocks that return at once, 34 us of dispatch at 32 per row)
        hipLaunchKernelGGL(k_reduce_segments<SpJobTableRef>, dim3(behind_events ? 8 : 32, nr + nz, n), dim3(256), 0, ctx->stream,
                           (const u32 *)wslab, ref, (u32)SP_SEG_ROWS, rs, d_nflagged);
        PMX_CHECK_LAUNCH("k_reduce_segments");
    }
    return PMX_OK;
}

// max_shift in [1024, EV_MAX_SHIFT]: the event kernel over every chromosome (one pass over the vectors whatever the
// shift range), then -- for the tiles it flagged as dense only -- the window kernel in chunks of 1024 shifts, whose sums a
// gated reduce adds to the rows.  The mappable-length pass is not fused here (the caller runs k_autocorr_pairs).
static int launch_cc_events_big(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_shift, uint32_t read_len,
                                bool do_ncc, uint32_t out_stride, const ReduceSpec &rs, u32 nr, u32 nz, uint32_t fused_lag,
                                pmx_fused_mlen *fused)
{
    const bool has_m = jobs[0].d_M != nullptr;
    const u32 c = read_len - 1;
    EvBigPlan pl;
    // the mappable-length pairs in the same launch (round 4), when the caller asks for it and the geometry allows it
    bool fuse_mlen = has_m && fused && fused_lag && pmx_events_can_fuse_mlen(max_shift, fused_lag) && ev_big_plan<true>(max_shift, &pl, fused_lag);
    if (!fuse_mlen && !(has_m ? ev_big_plan<true>(max_shift, &pl) : ev_big_plan<false>(max_shift, &pl))) {
        pmx_set_error("k_cc_events: no launch geometry for max_shift %u", max_shift);
        return PMX_ERR_INVALID;
    }
    if (fuse_mlen) fused->done = true;   // row MLEN and scalar [2] are written by this call
    std::vector<uint32_t> flag0(njobs);
    uint64_t total_flags = 0;
    for (uint32_t i = 0; i < njobs; i++) {
        flag0[i] = (uint32_t)total_flags;
        total_flags += (jobs[i].nbits + SP_TB - 1) / SP_TB + EV_NQ;   // padding: an event tile flags EV_NQ window tiles
    }
    const size_t flag_bytes = (size_t)((total_flags + 15) / 16 * 16);
    const uint32_t per_launch = njobs < SP_MAXJOBS_REF ? njobs : SP_MAXJOBS_REF;
    const size_t stat_bytes = 4 * (size_t)per_launch * sizeof(u32);
    // [flags of the cc window kernel][flags of the autocorrelation window kernel (fused pairs only)][counters][job statistics]
    const size_t nflag_arrays = fuse_mlen ? 2 : 1;
    int rc = pmx_ensure_flags_cc(ctx, nflag_arrays * flag_bytes + 16 + stat_bytes);
    if (rc) return rc;
    unsigned char *d_flags = ctx->d_flags_cc;
    unsigned char *d_flags_ac = fuse_mlen ? ctx->d_flags_cc + flag_bytes : nullptr;
    u32 *d_nflagged = (u32 *)(ctx->d_flags_cc + nflag_arrays * flag_bytes);
    u32 *d_jobstat = d_nflagged + 4;     // per job: tiles seen / read-dense / edge-dense (k_cc_events, one sub-group)
    PMX_HIP(hipMemsetAsync(ctx->d_flags_cc, 0, nflag_arrays * flag_bytes + 16 + stat_bytes, ctx->stream));
    ctx->flags_cc_zeroed[0] = ctx->flags_cc_zeroed[1] = ctx->flags_cc_dirty[0] = ctx->flags_cc_dirty[1] = 0;   // (see ev_flag_area)

    ReduceSpec rs_ev = rs;
    for (u32 i = 0; i < nr; i++)
        if (rs.dst_row[i] == PMX_ROW_MSCC_FSUM || rs.dst_row[i] == PMX_ROW_MSCC_RSUM) rs_ev.is_signed[i] = 1;
    rs_ev.rowlen = pl.hn;
    const u32 nhr = (max_shift + 1 + 127) / 128;   // quads of R above a tile that hold partners of its forward reads
    for (uint32_t lo = 0; lo < njobs; lo += SP_MAXJOBS_REF) {
        const uint32_t n = njobs - lo < SP_MAXJOBS_REF ? njobs - lo : SP_MAXJOBS_REF;
        std::vector<VJob> ev(n);
        for (auto &v : ev) v.ranged = false;
        for (uint32_t i = 0; i < n; i++) {
            ev[i].job = &jobs[lo + i];
            ev[i].d_off = 0;
            ev[i].d_n = max_shift + 1;
            ev[i].flag0 = flag0[lo + i];
        }
        std::vector<SpJobDev> tab(n);
        memset(tab.data(), 0, n * sizeof(SpJobDev));
        uint32_t total, tpw, nwg;
        if (lo) PMX_HIP(hipMemsetAsync(d_jobstat, 0, stat_bytes, ctx->stream));   // (the next launch's jobs)
        plan_launch(ctx, ev.data(), n, false, pl.wg_per_cu, tab.data(), &total, &tpw, &nwg, EV_TB);
        rc = pmx_ensure_slab(ctx, (size_t)(nwg + n) * EV_SEG_ROWS * pl.hn + (size_t)nwg * 4 * pl.nsg * 12 * 2 + 64);
        if (rc) return rc;
        SpJobTableRef ref;
        rc = upload_table(ctx, tab, &ref);
        if (rc) return rc;
        pmx_timed_launch tl;
        rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_EVENTS, &tl);
        if (rc) return rc;
#define EVB(HM, NC, ML)                                                                                                                 \
    (pl.nsg == 1   ? ev_big_launch<HM, NC, 1, ML>(ctx, pl, ref, n, total, tpw, nwg, c, max_shift, nhr, d_flags, d_nflagged, d_jobstat,  \
                                                  ML ? fused_lag : 0u, d_flags_ac)                                                      \
     : pl.nsg == 2 ? ev_big_launch<HM, NC, 2, ML>(ctx, pl, ref, n, total, tpw, nwg, c, max_shift, nhr, d_flags, d_nflagged, d_jobstat,  \
                                                  ML ? fused_lag : 0u, d_flags_ac)                                                      \
                   : ev_big_launch<HM, NC, 4, ML>(ctx, pl, ref, n, total, tpw, nwg, c, max_shift, nhr, d_flags, d_nflagged, d_jobstat,  \
                                                  ML ? fused_lag : 0u, d_flags_ac))
        if (fuse_mlen) rc = do_ncc ? EVB(true, true, true) : EVB(true, false, true);
        else rc = has_m ? (do_ncc ? EVB(true, true, false) : EVB(true, false, false)) : EVB(false, true, false);
#undef EVB
        if (rc) return rc;
        rc = pmx_prof_end(ctx, &tl);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reduce_segments<SpJobTableRef>, dim3(32, nr + nz, n), dim3(256), 0, ctx->stream,
                           (const u32 *)ctx->d_slab, ref, (u32)EV_SEG_ROWS, rs_ev, (const u32 *)nullptr);
        PMX_CHECK_LAUNCH("k_reduce_segments");
        if (fuse_mlen) {
            // the pair sums EE = P - N (row 5, signed) -> the P row of the jobs' scratch, N cleared, popcount(M) and the run
            // count (scalars 4, 5) -> its scalars: what k_autocorr_pairs + k_reduce_pairs leave there
            const u32 lagcap = (u32)(((size_t)fused_lag + 1 + 1023) / 1024 * 1024);
            ReduceSpec r2;
            memset(&r2, 0, sizeof r2);
            r2.nrows = 2;
            r2.src_row[0] = 5; r2.dst_row[0] = 0; r2.is_signed[0] = 1;
            r2.src_row[1] = 4; r2.dst_row[1] = 2 * lagcap; r2.is_scalar[1] = 1; r2.scalar_off = 4;
            r2.nzero = 1;
            r2.zero_row[0] = lagcap;
            r2.use_out2 = 1;
            r2.n_override = fused_lag + 1;
            r2.out_stride = out_stride;
            r2.rowlen = pl.hn;
            hipLaunchKernelGGL(k_reduce_segments<SpJobTableRef>, dim3(32, 3, n), dim3(256), 0, ctx->stream, (const u32 *)ctx->d_slab, ref,
                               (u32)EV_SEG_ROWS, r2, (const u32 *)nullptr);
            PMX_CHECK_LAUNCH("k_reduce_segments (mappable-length pairs)");
        }
        if (has_m) {
            EvTailPlan tp;
            memset(&tp, 0, sizeof tp);
            hipLaunchKernelGGL(k_events_tail<SpJobTableRef>, dim3(n, 1), dim3(EV_TAIL_THREADS), 0, ctx->stream, (const u32 *)ctx->d_slab,
                               ref, tp, (const u32 *)nullptr, (const u32 *)nullptr, (const u32 *)d_nflagged, max_shift, out_stride, 1u,
                               do_ncc ? 1u : 0u, 0u, 1024u, (int32_t)c, 0u, pl.hn, 0u, (const u32 *)nullptr, (const u32 *)nullptr, 0u);
            PMX_CHECK_LAUNCH("k_events_tail");
        }
    }
    // the flagged tiles: window kernel per (chromosome, chunk of 1024 shifts); nothing flagged: the launch returns at once
    std::vector<VJob> vjobs;
    expand_chunks(jobs, njobs, max_shift + 1, vjobs, flag0.data());
    ReduceSpec rs_w = rs;
    rs_w.accumulate = 1;
    rc = launch_cc_window_chunks(ctx, vjobs, has_m, do_ncc, (int32_t)c, rs_w, nr, nz, d_flags, d_nflagged);
    if (rc || !fuse_mlen) return rc;
    // the window kernel of the mappable-length pass for the tiles whose run edges were not listed (normally none) and the
    // recurrence A(k+1) = 2 A(k) - A(k-1) - EE(k) -> row MLEN and scalar [2]
    return launch_autocorr_batch(ctx, jobs, njobs, fused_lag, 1, read_len, max_shift, out_stride, flag0.data(), d_flags_ac, d_nflagged + 1);
}

// jobs one call of pmx_launch_cc_sparse_batch takes: launches with a device-side job table take the whole batch
uint32_t pmx_cc_batch_jobs(uint32_t max_shift) { return max_shift > 1023 ? SP_MAXJOBS_REF : SP_MAXJOBS; }
uint32_t pmx_autocorr_batch_jobs(void) { return SP_MAXJOBS_REF; }

int pmx_events_take_big(uint32_t max_shift)
{
    return max_shift > 1023 && max_shift <= EV_MAX_SHIFT && events_enabled() && events_big_enabled();
}

int pmx_events_can_fuse_mlen(uint32_t max_shift, uint32_t max_lag)
{
    static const bool fuse = [] {   // PMX_CC_FUSE_MLEN=0: the mappable-length pass stays a pass of its own
        const char *e = getenv("PMX_CC_FUSE_MLEN");
        return !(e && e[0] == '0');
    }();
    if (!(events_enabled() && fuse)) return 0;
    if (max_shift <= 1023) return max_lag <= 1023;   // one histogram row of 1024 lags
    // beyond: the pairs go to the slab with global atomics (k_cc_events, BIG + DO_MLEN); the run edges up to max_lag bits above a
    // tile are staged by one wavefront (PMX_CC_FUSE_MLEN_BIG=0: the pair pass over M stays a pass of its own, A/B)
    static const bool fuse_big = [] {
        const char *e = getenv("PMX_CC_FUSE_MLEN_BIG");
        return !(e && e[0] == '0');
    }();
    EvBigPlan pl;
    return fuse_big && pmx_events_take_big(max_shift) && ev_big_plan<true>(max_shift, &pl, max_lag);
}

// does the event kernel take this shift range at all (the density probe of pmx_cc_batch_dev only matters then)
int pmx_events_used(uint32_t max_shift) { return (max_shift <= 1023 && events_enabled()) || pmx_events_take_big(max_shift); }

// The sub-group count the BIG launch of this shift range will use (0: none).  With ONE sub-group per workgroup (max_shift <=
// 2047: 35 KB of LDS, four workgroups per CU) a CU has LDS to spare and the mappable-length pair pass runs beside the event
// kernel on the auxiliary stream (hg38, -d 1024: 0.74 -> 0.68 ms; -d 2047: 0.85 -> 0.80); with 2 or 4 sub-groups it loses.
int pmx_events_big_subgroups(uint32_t max_shift, int has_m)
{
    if (!pmx_events_take_big(max_shift)) return 0;
    EvBigPlan pl;
    if (!(has_m ? ev_big_plan<true>(max_shift, &pl) : ev_big_plan<false>(max_shift, &pl))) return 0;
    return (int)pl.nsg;
}

// The dense-tile flags, the flagged-tile counters and the job statistics of the event pass must be ZERO when k_cc_events
// starts.  Until round 3 a memset in front of every pass did that (5 us of stream time per step); now the context keeps TWO
// areas and uses them in turn: the pass in area A has its k_events_finish launch clear what the previous pass dirtied in area B
// (nothing of the running pass touches B), so a pass finds its area clean without a launch of its own.  zeroed[]: prefix of
// an idle area known to be zero; dirty[]: prefix a pass has used and nobody has cleared yet (cleared here if the pass that
// should have done it never launched its k_events_finish: an error exit).
static int ev_flag_area(pmx_ctx *ctx, size_t area_bytes, size_t zero_bytes, unsigned char **cur, unsigned char **other,
                        size_t *other_dirty)
{
    const size_t half = (area_bytes + 255) & ~(size_t)255;
    const size_t before = ctx->flags_cc_bytes;
    int rc = pmx_ensure_flags_cc(ctx, 2 * half);
    if (rc) return rc;
    if (ctx->flags_cc_bytes != before) ctx->flags_cc_zeroed[0] = ctx->flags_cc_zeroed[1] = ctx->flags_cc_dirty[0] = ctx->flags_cc_dirty[1] = 0;
    const size_t stride = (ctx->flags_cc_bytes / 2) & ~(size_t)255;
    const u32 a = ctx->flags_cc_area & 1u;
    ctx->flags_cc_area = a ^ 1u;
    *cur = ctx->d_flags_cc + a * stride;
    *other = ctx->d_flags_cc + (a ^ 1u) * stride;
    if (ctx->flags_cc_dirty[a]) {
        PMX_HIP(hipMemsetAsync(*cur, 0, ctx->flags_cc_dirty[a], ctx->stream));
        ctx->flags_cc_dirty[a] = 0;
    }
    if (ctx->flags_cc_zeroed[a] < zero_bytes) {
        PMX_HIP(hipMemsetAsync(*cur, 0, zero_bytes, ctx->stream));
        ctx->flags_cc_zeroed[a] = zero_bytes;
    }
    ctx->flags_cc_dirty[a] = zero_bytes;
    *other_dirty = ctx->flags_cc_dirty[a ^ 1u];
    return PMX_OK;
}

int pmx_launch_cc_sparse_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_shift,
                               uint32_t read_len, bool do_ncc, uint32_t out_stride, bool zero_mlen,
                               uint32_t fused_lag, pmx_fused_mlen *fused)
{
    if (fused) fused->done = false;
    if (njobs == 0) return PMX_OK;
    const bool has_m = jobs[0].d_M != nullptr;
    if (!has_m && !do_ncc) return PMX_OK;
    const bool chunked = max_shift > 1023;
    const int32_t c = (int32_t)read_len - 1;
    const u32 lgG = chunked ? 5u : lg_slot_lanes(max_shift + 1);

    ReduceSpec rs;
    memset(&rs, 0, sizeof rs);
    u32 nr = 0, nz = 0;
    if (do_ncc) { rs.src_row[nr] = 0; rs.dst_row[nr] = PMX_ROW_NCC_CCBINS; nr++; }
    else rs.zero_row[nz++] = PMX_ROW_NCC_CCBINS;
    if (has_m) {
        rs.src_row[nr] = 1; rs.dst_row[nr] = PMX_ROW_MSCC_FSUM; nr++;
        rs.src_row[nr] = 2; rs.dst_row[nr] = PMX_ROW_MSCC_CCBINS; nr++;
        rs.src_row[nr] = 3; rs.dst_row[nr] = PMX_ROW_MSCC_RSUM; nr++;
        if (zero_mlen) rs.zero_row[nz++] = PMX_ROW_MLEN;   // no autocorrelation pass will write it
    } else {
        rs.zero_row[nz++] = PMX_ROW_MSCC_FSUM;
        rs.zero_row[nz++] = PMX_ROW_MSCC_CCBINS;
        rs.zero_row[nz++] = PMX_ROW_MSCC_RSUM;
        rs.zero_row[nz++] = PMX_ROW_MLEN;
    }
    rs.src_row[nr] = 4; rs.dst_row[nr] = PMX_ROW_SCALARS; rs.is_scalar[nr] = 1; nr++;
    rs.nrows = nr;
    rs.nzero = nz;
    rs.out_stride = out_stride;
    rs.use_out2 = 0;
    rs.keep_scalar2 = (has_m && !zero_mlen) ? 1 : 0;

    if (!ctx->window_only && pmx_events_take_big(max_shift))
        return launch_cc_events_big(ctx, jobs, njobs, max_shift, read_len, do_ncc, out_stride, rs, nr, nz, fused_lag, fused);

    std::vector<VJob> vjobs;
    expand_chunks(jobs, njobs, max_shift + 1, vjobs);
    if (chunked)   // the window kernel alone over (chromosome x shift chunk) jobs (max_shift > 8191, or the event kernel disabled)
        return launch_cc_window_chunks(ctx, vjobs, has_m, do_ncc, c, rs, nr, nz, nullptr, nullptr);

    // Pass 1 (sparse tiles): the event kernel over every chromosome; tiles whose lists would overflow are flagged.
    // (the event pass sums modulo 2^32 -- kernels_events.h, the transformed flush --: vectors of 2^32 bits and more, which no BAM
    // file can describe, stay on the window kernels)
    bool small_vectors = true;
    for (uint32_t i = 0; i < njobs; i++) small_vectors = small_vectors && jobs[i].nbits < (1ull << 32);
    const bool use_events = events_enabled() && !ctx->window_only && !chunked && small_vectors;
    const bool fuse_mlen = use_events && has_m && fused && njobs <= SP_MAXJOBS && pmx_events_can_fuse_mlen(max_shift, fused_lag);
    for (uint32_t i = 0; i < njobs; i++)
        if (jobs[i].tile_count && (!use_events || (has_m && fused && !fuse_mlen))) {
            pmx_set_error("pmx_cc_batch_ranges_dev: tile ranges are taken by the event kernel of max_shift <= 1023 only (not with "
                          "PMX_FLAG_WINDOW_ONLY, PMX_CC_EVENTS=0 or vectors of 2^32 bits)");
            return PMX_ERR_INVALID;
        }
    unsigned char *d_flags = nullptr, *d_flags_ac = nullptr;
    u32 *d_nflagged = nullptr, *d_plan_cc = nullptr, *d_plan_ac = nullptr, *d_jobstat = nullptr;
    size_t flag_bytes_all = 0;     // raw length of a flag array (multiple of 16)
    unsigned char *d_other_area = nullptr;   // the context's other flag area: cleared by this pass's k_events_finish
    size_t other_dirty = 0;
    if (use_events) {
        uint64_t total_flags = 0;
        for (size_t i = 0; i < vjobs.size(); i++) {
            vjobs[i].flag0 = (uint32_t)total_flags;
            total_flags += (vjobs[i].job->nbits + SP_TB - 1) / SP_TB + EV_NQ;   // padding: an event tile flags EV_NQ window tiles
        }
        const size_t flag_bytes = (size_t)((total_flags + 15) / 16 * 16);
        const size_t stat_bytes = EV_STAT_BYTES;   // job statistics + the chromosomes' scalar totals
        const size_t zero_bytes = 2 * flag_bytes + 16 + stat_bytes;   // (a multiple of 16)
        unsigned char *area = nullptr;
        int rc = ev_flag_area(ctx, zero_bytes + 2 * PLAN_WORDS * sizeof(u32), zero_bytes, &area, &d_other_area, &other_dirty);
        if (rc) return rc;
        d_flags = area;                              // tiles of the cross-correlation window kernel (32 Kbit)
        d_flags_ac = area + flag_bytes;              // tiles of the autocorrelation window kernel (64 Kbit), same flag0 per job
        d_nflagged = (u32 *)(area + 2 * flag_bytes);
        d_jobstat = (u32 *)(area + 2 * flag_bytes + 16);   // per job: tiles seen / read-dense / edge-dense (k_cc_events)
        flag_bytes_all = flag_bytes;
        d_plan_cc = (u32 *)((unsigned char *)d_jobstat + stat_bytes);   // work split of the two window launches (k_events_finish)
        d_plan_ac = d_plan_cc + PLAN_WORDS;
        if (fuse_mlen) fused->done = true;   // row MLEN and scalar [2] are written by this call (k_events_finish)
    }
    ReduceSpec rs_ev = rs;
    for (u32 i = 0; i < nr; i++)
        if (rs.dst_row[i] == PMX_ROW_MSCC_FSUM || rs.dst_row[i] == PMX_ROW_MSCC_RSUM) rs_ev.is_signed[i] = 1;

    for (size_t lo = 0; lo < vjobs.size(); lo += SP_MAXJOBS) {
        const uint32_t n = (uint32_t)(vjobs.size() - lo < SP_MAXJOBS ? vjobs.size() - lo : SP_MAXJOBS);
        SpJobTable tab;
        uint32_t total, tpw, nwg;
        int rc;
        pmx_timed_launch tl;
        // the flag entries of this launch's jobs
        const u32 raw_lo = use_events ? vjobs[lo].flag0 : 0u;
        const u32 raw_hi = use_events ? (lo + n < vjobs.size() ? vjobs[lo + n].flag0 : (u32)flag_bytes_all) : 0u;
        if (use_events) {
            memset(&tab, 0, sizeof tab);
            // (the next 32 jobs: their own flagged-tile counters and job statistics)
            if (lo) PMX_HIP(hipMemsetAsync(d_nflagged, 0, 16 + EV_STAT_BYTES, ctx->stream));
            const bool deep = has_m && ctx->deep_lists;   // PMX_FLAG_DEEP_LISTS: the larger list pool at four workgroups per CU
            // workgroups per CU: what the instantiation was built for (5 / 4 / 8), or fewer if this device takes fewer (the
            // grid is one round of resident workgroups: a sixth that does not fit would run as a tail behind the others)
            u32 per_cu = has_m ? (deep ? 4 : EV_WAVES) : EV_WAVES_NCC;
            {
#define EV_OCC(HM, NC, ML, DP) ev_resident_per_cu<HM, NC, ML, DP>(ctx, per_cu)
                if (has_m && do_ncc && fuse_mlen) per_cu = deep ? EV_OCC(true, true, true, true) : EV_OCC(true, true, true, false);
                else if (has_m && do_ncc) per_cu = deep ? EV_OCC(true, true, false, true) : EV_OCC(true, true, false, false);
                else if (has_m && fuse_mlen) per_cu = deep ? EV_OCC(true, false, true, true) : EV_OCC(true, false, true, false);
                else if (has_m) per_cu = deep ? EV_OCC(true, false, false, true) : EV_OCC(true, false, false, false);
                else per_cu = EV_OCC(false, true, false, false);
#undef EV_OCC
            }
            // (tile-range jobs -- a rank's share of a chromosome, pmx_cc_batch_ranges_dev -- exist for this launch only: the window
            // launches behind it see whole chromosomes and take the tiles it flagged, which lie inside the ranges)
            for (uint32_t i = 0; i < n; i++) vjobs[lo + i].ranged = true;
            plan_launch(ctx, &vjobs[lo], n, false, per_cu, &tab, &total, &tpw, &nwg, EV_TB);
            for (uint32_t i = 0; i < n; i++) vjobs[lo + i].ranged = false;
            rc = pmx_ensure_slab(ctx, (size_t)(nwg + n) * EV_SEG_ROWS * 1024 + (size_t)nwg * 4 * 12 * 2 + 64);
            if (rc) return rc;
            rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_EVENTS, &tl);
            if (rc) return rc;
            const u32 nhr = (max_shift + 1 + 127) / 128;   // quads of R above a tile that hold partners of its forward reads
#define EV_LAUNCH_(HM, NC, ML, DP)                                                                                     \
    hipLaunchKernelGGL((k_cc_events<HM, NC, ML, 1, false, SpJobTable, DP>), dim3(nwg), dim3(256), 0, ctx->stream, tab, n, total, tpw, \
                       (u32)c, max_shift, nhr, fused_lag, 1024u, (u32)EV_LO, (u32)EV_HI, ctx->d_slab, d_flags, d_flags_ac, d_nflagged, d_jobstat)
#define EV_LAUNCH(HM, NC, ML)                \
    do {                                     \
        if (deep) EV_LAUNCH_(HM, NC, ML, HM); \
        else EV_LAUNCH_(HM, NC, ML, false);  \
    } while (0)
            if (has_m && do_ncc && fuse_mlen) EV_LAUNCH(true, true, true);
            else if (has_m && do_ncc) EV_LAUNCH(true, true, false);
            else if (has_m && fuse_mlen) EV_LAUNCH(true, false, true);
            else if (has_m) EV_LAUNCH(true, false, false);
            else EV_LAUNCH_(false, true, false, false);
#undef EV_LAUNCH
#undef EV_LAUNCH_
            PMX_CHECK_LAUNCH("k_cc_events");
            rc = pmx_prof_end(ctx, &tl);
            if (rc) return rc;
        }
        // Pass 2 (window kernel): every tile, or -- behind the event pass -- only the tiles it flagged (the whole grid returns
        // at once when there are none).  Behind the event pass it adds its sums to the rows k_events_finish has written.
        SpJobTable tabW, tabA;
        std::vector<VJob> va(n);
        for (auto &v : va) v.ranged = false;
        uint32_t totalA = 0, tpwA = 0, nwgA = 0;
        memset(&tabW, 0, sizeof tabW);
        plan_launch(ctx, &vjobs[lo], n, false, chunked ? (has_m ? SP_WAVES_CH : SP_WAVES_CH_NCC) : (has_m ? SP_WAVES : SP_WAVES_NCC), &tabW, &total, &tpw, &nwg);
        const size_t wwords = (size_t)(nwg + n) * SP_SEG_ROWS * 1024 + (size_t)nwg * 4 * 12 * 2 + 64;
        rc = use_events ? pmx_ensure_slab_fb(ctx, wwords) : pmx_ensure_slab(ctx, wwords);
        if (rc) return rc;
        u32 *const wslab = use_events ? ctx->d_slab_fb : ctx->d_slab;
        if (use_events) {
            // the flagged tiles in equal shares (one small block; returns at once when nothing was flagged)
            if (nwg > 2048) {
                pmx_set_error("k_events_finish: %u workgroups exceed the planner's tables", nwg);
                return PMX_ERR_INVALID;
            }
            // (both window launches of this batch are planned in the same launch: the autocorrelation launch's shape is known here)
            PlanLaunch pcc, pac;
            memset(&pcc, 0, sizeof pcc);
            memset(&pac, 0, sizeof pac);
            fill_plan_launch(pcc, tabW, n, total, nwg, raw_lo, raw_hi, (u32)EV_NQ, (const unsigned char *)d_flags, d_nflagged, d_plan_cc);
            if (fuse_mlen) {
                for (uint32_t i = 0; i < n; i++) {
                    va[i].job = vjobs[lo + i].job;
                    va[i].d_off = 0;
                    va[i].d_n = fused_lag + 1;
                    va[i].flag0 = vjobs[lo + i].flag0;
                }
                memset(&tabA, 0, sizeof tabA);
                plan_launch(ctx, va.data(), n, true, AC_WAVES, &tabA, &totalA, &tpwA, &nwgA);
                if (nwgA > 2048) {
                    pmx_set_error("k_events_finish: %u workgroups exceed the planner's tables", nwgA);
                    return PMX_ERR_INVALID;
                }
                fill_plan_launch(pac, tabA, n, totalA, nwgA, raw_lo, raw_hi, 1u, (const unsigned char *)d_flags_ac, d_nflagged + 1, d_plan_ac);
            }
            // sums over the workgroups' segments, prefix sums, the mappable-length recurrence, the plan of the window launches
            // and the clearing of the other flag area: ONE launch (k_reduce_segments2 + k_plan_flagged + half of k_events_tail + a
            // memset until round 3)
            EvFinishArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.slab = ctx->d_slab;
            fa.jobsum = EV_JOBSUM(d_jobstat);
            fa.S = max_shift;
            fa.out_stride = out_stride;
            fa.has_m = has_m ? 1u : 0u;
            fa.do_ncc = do_ncc ? 1u : 0u;
            fa.fused_mlen = fuse_mlen ? 1u : 0u;
            fa.zero_mlen = zero_mlen ? 1u : 0u;
            fa.keep_scalar2 = rs.keep_scalar2;
            fa.zero_area = reinterpret_cast<uint4 *>(d_other_area);
            fa.zero_quads = (u32)(other_dirty / 16);
            hipLaunchKernelGGL(k_events_finish, dim3(EVF_CHUNKS, EVF_TASKS, n + 1), dim3(1024), 0, ctx->stream, tab, n, fa, pcc, pac);
            PMX_CHECK_LAUNCH("k_events_finish");
            ctx->flags_cc_dirty[(ctx->flags_cc_area & 1u)] = 0;   // (flags_cc_area already points at the other area: the next pass's)
        }
        if (use_events && fuse_mlen) {
            // both window kernels for the flagged tiles in ONE launch (k_windows_flagged); each adds its shares to the rows
            // itself, nothing is launched behind it (k_events_tail until round 3).  Timed as the window kernel.
            rc = pmx_ensure_slab_ac(ctx, (size_t)(nwgA + n) * AC_SEG_ROWS * 1024);
            if (rc) return rc;
            rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_SPARSE, &tl, true);
            if (rc) return rc;
            WinCcArgs wa;
            wa.total_tiles = total; wa.tiles_per_wg = tpw; wa.lgG = lgG; wa.c = c;
            wa.slab = wslab; wa.tile_flags = d_flags; wa.plan = d_plan_cc;
            WinAcArgs aa;
            aa.total_tiles = totalA; aa.tiles_per_wg = tpwA; aa.lgG = lg_slot_lanes(fused_lag + 1);
            aa.add_shift = max_shift; aa.add_lag = fused_lag;
            aa.slab = ctx->d_slab_ac; aa.tile_flags = d_flags_ac; aa.plan = d_plan_ac;
            if (do_ncc)
                hipLaunchKernelGGL(k_windows_flagged<true>, dim3(nwg + nwgA), dim3(256), 0, ctx->stream, tabW, tabA, n, nwg, wa, aa,
                                   (const u32 *)d_nflagged, out_stride);
            else
                hipLaunchKernelGGL(k_windows_flagged<false>, dim3(nwg + nwgA), dim3(256), 0, ctx->stream, tabW, tabA, n, nwg, wa, aa,
                                   (const u32 *)d_nflagged, out_stride);
            PMX_CHECK_LAUNCH("k_windows_flagged");
            rc = pmx_prof_end(ctx, &tl);
            if (rc) return rc;
            continue;
        }
        rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_SPARSE, &tl, use_events);
        if (rc) return rc;
#define SP_LAUNCH(HM, NC, CK)                                                                                       \
    hipLaunchKernelGGL((k_cc_sparse<HM, NC, CK>), dim3(nwg), dim3(256), 0, ctx->stream, tabW, n, total, tpw, c, lgG, \
                       wslab, (const unsigned char *)d_flags, (const u32 *)d_nflagged, (const u32 *)(use_events ? d_plan_cc : nullptr), \
                       use_events ? out_stride : 0u)
        if (chunked) {
            if (has_m && do_ncc) SP_LAUNCH(true, true, true);
            else if (has_m) SP_LAUNCH(true, false, true);
            else SP_LAUNCH(false, true, true);
        } else {
            if (has_m && do_ncc) SP_LAUNCH(true, true, false);
            else if (has_m) SP_LAUNCH(true, false, false);
            else SP_LAUNCH(false, true, false);
        }
#undef SP_LAUNCH
        PMX_CHECK_LAUNCH("k_cc_sparse");
        rc = pmx_prof_end(ctx, &tl);
        if (rc) return rc;
        if (!use_events) {
            // sum the per-workgroup slab segments into the result blocks
            hipLaunchKernelGGL(k_reduce_segments, dim3(32, nr + nz, n), dim3(256), 0, ctx->stream, (const u32 *)ctx->d_slab, tabW,
                               (u32)SP_SEG_ROWS, rs, (const u32 *)nullptr);
            PMX_CHECK_LAUNCH("k_reduce_segments");
            continue;
        }
    }
    return PMX_OK;
}

size_t pmx_autocorr_scratch_words(uint32_t max_lag)
{
    const size_t lagcap = ((size_t)max_lag + 1 + 1023) / 1024 * 1024;
    return 3 * lagcap + 16;
}

// jobs[i].d_M / nbits: the vector; jobs[i].d_out2: pmx_autocorr_scratch_words(max_lag) u64 of scratch per job;
// mode 0: jobs[i].d_out[k] = A(k), k <= max_lag.  mode 1: jobs[i].d_out is a result block: row MLEN[d] = A(|L-1-d|),
// scalar [2] = popcount(M).
int pmx_launch_autocorr_edges_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_lag, uint32_t mode,
                                    uint32_t read_len, uint32_t max_shift, uint32_t out_stride)
{
    return launch_autocorr_batch(ctx, jobs, njobs, max_lag, mode, read_len, max_shift, out_stride, nullptr, nullptr, nullptr);
}

// pre_flags != nullptr: the event kernel (max_shift > 1023, fused mappable-length pairs) has left the pair sums, popcount(M)
// and the run count in the jobs' scratch and flagged the tiles whose run edges it could not list (pre_flag0[job]: index of
// the job's first tile in pre_flags): only the window kernel for those tiles + the recurrence remain.
static int launch_autocorr_batch(pmx_ctx *ctx, const pmx_job *jobs, uint32_t njobs, uint32_t max_lag, uint32_t mode,
                                 uint32_t read_len, uint32_t max_shift, uint32_t out_stride, const uint32_t *pre_flag0,
                                 const unsigned char *pre_flags, const u32 *pre_nflagged)
{
    if (njobs == 0) return PMX_OK;
    const bool chunked = max_lag > 1023;
    const u32 lagcap = (u32)(((size_t)max_lag + 1 + 1023) / 1024 * 1024);
    const u32 lgG = chunked ? 5u : lg_slot_lanes(max_lag + 1);
    typedef SpJobTableRef JT;   // job tables in device memory: every chromosome of the batch in one launch of each kernel

    // Pass 1 (sparse-edge tiles): pair enumeration over every chromosome; tiles with more than AP_CAP edges are flagged.
    // PMX_AUTOCORR_PAIRS=0 in the environment keeps everything on the window kernel (A/B measurements, tests).
    static const bool pairs_enabled = [] {
        const char *e = getenv("PMX_AUTOCORR_PAIRS");
        return !(e && e[0] == '0');
    }();
    const bool use_pairs = !pre_flags && pairs_enabled && max_lag + 1 <= AP_MAX_LAGS;
    std::vector<uint32_t> flag0(njobs, 0);
    unsigned char *d_flags = const_cast<unsigned char *>(pre_flags);
    u32 *d_nflagged = const_cast<u32 *>(pre_nflagged);
    if (pre_flags)
        for (uint32_t i = 0; i < njobs; i++) flag0[i] = pre_flag0[i];
    const u32 nl = (max_lag + 1 + 63) / 64 * 64;
    int rc;
    if (use_pairs) {
        uint64_t total_flags = 0;
        for (uint32_t i = 0; i < njobs; i++) {
            flag0[i] = (uint32_t)total_flags;
            total_flags += (jobs[i].nbits + 1 + AC_TB - 1) / AC_TB + AP_NQ / AC_NQ;   // padding: a pair tile flags AP_NQ / AC_NQ window tiles
        }
        const size_t flag_bytes = (size_t)((total_flags + 15) / 16 * 16);
        rc = pmx_ensure_flags(ctx, flag_bytes + 16);
        if (rc) return rc;
        d_flags = ctx->d_flags;
        d_nflagged = (u32 *)(ctx->d_flags + flag_bytes);
        PMX_HIP(hipMemsetAsync(ctx->d_flags, 0, flag_bytes + 16, ctx->stream));
        const u32 nh = (max_lag + 1 + 127) / 128;   // quads above a tile that hold partners of its drivers
        const size_t lds_bytes = (2 * (size_t)nl + 2 * AP_CAP + 32 + 16) * sizeof(u32);
        const size_t seg_stride = 2 * (size_t)nl + 16;
        // workgroups per CU: 8 by registers; the histograms may allow fewer
        u32 per_cu = (u32)((160u * 1024u) / (lds_bytes + 512));
        if (per_cu > AP_WAVES) per_cu = AP_WAVES;
        if (per_cu < 1) per_cu = 1;
        for (uint32_t lo = 0; lo < njobs; lo += SP_MAXJOBS_REF) {
            const uint32_t n = njobs - lo < SP_MAXJOBS_REF ? njobs - lo : SP_MAXJOBS_REF;
            std::vector<VJob> vj(n);
            for (auto &v : vj) v.ranged = false;
            for (uint32_t i = 0; i < n; i++) {
                vj[i].job = &jobs[lo + i];
                vj[i].d_off = 0;
                vj[i].d_n = max_lag + 1;
                vj[i].flag0 = flag0[lo + i];
            }
            std::vector<SpJobDev> tab(n);
            memset(tab.data(), 0, n * sizeof(SpJobDev));
            uint32_t total, tpw, nwg;
            plan_launch(ctx, vj.data(), n, true, per_cu, tab.data(), &total, &tpw, &nwg, AP_TB);
            rc = pmx_ensure_slab2(ctx, (size_t)(nwg + n) * seg_stride);
            if (rc) return rc;
            JT ref;
            rc = upload_table(ctx, tab, &ref);
            if (rc) return rc;
            if (lds_bytes > 64 * 1024) {   // (max_lag above ~7870: the histograms alone pass 64 KB)
                static bool attr_set[EV_MAX_DEVICES];   // (per device, see ev_big_launch)
                const int dev = ctx->device >= 0 && ctx->device < EV_MAX_DEVICES ? ctx->device : -1;
                if (dev < 0 || !attr_set[dev]) {
                    PMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_autocorr_pairs<JT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
                    if (dev >= 0) attr_set[dev] = true;
                }
            }
            pmx_timed_launch tl;
            rc = pmx_prof_begin(ctx, PMX_KERNEL_AUTOCORR, &tl);
            if (rc) return rc;
            hipLaunchKernelGGL(k_autocorr_pairs<JT>, dim3(nwg), dim3(256), lds_bytes, ctx->stream, ref, n, total, tpw, max_lag, nl,
                               nh, ctx->d_slab2, d_flags, d_nflagged);
            PMX_CHECK_LAUNCH("k_autocorr_pairs");
            rc = pmx_prof_end(ctx, &tl);
            if (rc) return rc;
            hipLaunchKernelGGL(k_reduce_pairs<JT>, dim3((max_lag + 1 + 31) / 32 < 64 ? (max_lag + 1 + 31) / 32 : 64, 3, n), dim3(256), 0,
                               ctx->stream, (const u32 *)ctx->d_slab2, ref, nl, max_lag, lagcap, 0u);
            PMX_CHECK_LAUNCH("k_reduce_pairs");
        }
    }

    // Pass 2 (window kernel): every tile, or -- behind the pair pass -- only the tiles it flagged as dense (the whole
    // grid returns at once when there are none); its sums are ADDED to what the pair pass left in the per-job scratch.
    std::vector<VJob> vjobs;
    expand_chunks(jobs, njobs, max_lag + 1, vjobs, flag0.data());

    ReduceSpec rs;
    memset(&rs, 0, sizeof rs);
    rs.nrows = 3;
    rs.src_row[0] = 0; rs.dst_row[0] = 0;            // P
    rs.src_row[1] = 1; rs.dst_row[1] = lagcap;       // N
    rs.src_row[2] = 2; rs.dst_row[2] = 2 * lagcap; rs.is_scalar[2] = 1;
    rs.use_out2 = 1;
    rs.accumulate = (use_pairs || pre_flags) ? 1 : 0;
    rs.out_stride = out_stride;

    for (size_t lo = 0; lo < vjobs.size(); lo += SP_MAXJOBS_REF) {
        const uint32_t n = (uint32_t)(vjobs.size() - lo < SP_MAXJOBS_REF ? vjobs.size() - lo : SP_MAXJOBS_REF);
        std::vector<SpJobDev> tab(n);
        memset(tab.data(), 0, n * sizeof(SpJobDev));
        uint32_t total, tpw, nwg;
        plan_launch(ctx, &vjobs[lo], n, true, chunked ? AC_WAVES_CH : AC_WAVES, tab.data(), &total, &tpw, &nwg);
        rc = pmx_ensure_slab_ac(ctx, (size_t)(nwg + n) * AC_SEG_ROWS * 1024);
        if (rc) return rc;
        JT ref;
        rc = upload_table(ctx, tab, &ref);
        if (rc) return rc;
        pmx_timed_launch tl;
        rc = pmx_prof_begin(ctx, PMX_KERNEL_AUTOCORR, &tl);
        if (rc) return rc;
        if (chunked)
            hipLaunchKernelGGL((k_autocorr_edges<true, JT>), dim3(nwg), dim3(256), 0, ctx->stream, ref, n, total, tpw, lgG,
                               ctx->d_slab_ac, (const unsigned char *)d_flags, (const u32 *)d_nflagged, (const u32 *)nullptr);
        else
            hipLaunchKernelGGL((k_autocorr_edges<false, JT>), dim3(nwg), dim3(256), 0, ctx->stream, ref, n, total, tpw, lgG,
                               ctx->d_slab_ac, (const unsigned char *)d_flags, (const u32 *)d_nflagged, (const u32 *)nullptr);
        PMX_CHECK_LAUNCH("k_autocorr_edges");
        rc = pmx_prof_end(ctx, &tl);
        if (rc) return rc;
        hipLaunchKernelGGL(k_reduce_segments<JT>, dim3(32, 3, n), dim3(256), 0, ctx->stream, (const u32 *)ctx->d_slab_ac, ref,
                           (u32)AC_SEG_ROWS, rs, (const u32 *)d_nflagged);
        PMX_CHECK_LAUNCH("k_reduce_segments");
    }
    // the recurrence needs every chunk of a chromosome: run it once all launches are queued (same stream)
    for (uint32_t lo = 0; lo < njobs; lo += SP_MAXJOBS_REF) {
        const uint32_t n = njobs - lo < SP_MAXJOBS_REF ? njobs - lo : SP_MAXJOBS_REF;
        std::vector<SpJobDev> tab(n);
        memset(tab.data(), 0, n * sizeof(SpJobDev));
        for (uint32_t i = 0; i < n; i++) {
            tab[i].flags = 1;
            tab[i].out = (u64 *)jobs[lo + i].d_out;
            tab[i].out2 = (u64 *)jobs[lo + i].d_out2;
        }
        JT ref;
        rc = upload_table(ctx, tab, &ref);
        if (rc) return rc;
        hipLaunchKernelGGL(k_autocorr_finish<JT>, dim3(n), dim3(AC_FINISH_THREADS), 0, ctx->stream, ref, max_lag, lagcap, mode,
                           (int32_t)read_len - 1, max_shift, out_stride);
        PMX_CHECK_LAUNCH("k_autocorr_finish");
    }
    return PMX_OK;
}
