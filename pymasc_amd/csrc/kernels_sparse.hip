// Set-bit driven cross-correlation kernels (gfx950): the default path for read-occupancy vectors, which are
// sparse by nature (at most one bit per position and strand; ~0.5 % density for 15 M reads / 3.1 Gbp).
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
// Work layout: persistent workgroups of 4 wavefronts walk 32-Kbit tiles.  Per tile the F / R / M words (+ halos
// of S bits) are fetched with 16-byte coalesced loads into REGISTERS one tile ahead (software pipeline) and
// then stored to LDS; the tile's set bits are compacted into LDS lists with LDS atomic cursors (order is
// irrelevant for a sum); each SLOT (G lanes, G = 2^k >= (S+1)/32; 64/G slots per wave) pulls list entries,
// fetches its window words from LDS (ds_read2_b32 + v_alignbit, v_bfrev for descending windows) and feeds them
// to its counters four at a time: 3 static carry-save adders, then a binary-counter style insertion selected
// by one wave-uniform switch.  The stride-2 window of rsum reads even/odd decimated copies of the M tile built
// once per tile.  Counters become integers only when they could overflow or when the workgroup retires; they
// go to a PRIVATE per-workgroup slab (no cross-workgroup atomics) that k_reduce_slab sums afterwards.
#include "pmx_common.h"

#define SP_TB 32768u                 // driver bits per tile
#define SP_TBW 1024u                 // dwords per tile
#define SP_RHI 36u                   // R dwords staged above the tile (32 G / 32 + 2, rounded to 16 B)
#define SP_MLO 68u                   // M dwords staged below the tile (64 G + 64 bits, rounded to 16 B)
#define SP_MHI 36u                   // M dwords staged above the tile (read_len - 1 + 64 bits)
#define SP_MW (SP_MLO + SP_TBW + SP_MHI)   // 1128 dwords of M per tile
#define SP_HALO_M (SP_MLO * 32u)     // 2176 bits
#define SP_CAP 1024u                 // list entries per round
#define SP_NP 10                     // counter bit planes P[0..9]; parked carries Q[2..10]
#define SP_NQ 11
#define SP_QLIMIT 511u               // quads a counter may absorb between flushes (count < 2048)
#define SP_QSOFT 256u                // flush at a tile boundary once this many quads are pending
#define SP_VALID 0x20000u            // list entry: bit 17 = real entry, bit 16 = flag, bits 0..14 = position
#define SP_SLAB_ROWS 5               // per-workgroup partial rows of 1024 u32: ncc, fsum, ccbins(mscc), rsum, scalars

// LDS layout (dwords).  The flush accumulators (4 x 1024 u32) alias the tile buffers between tiles.
#define SP_OFF_F 0u
#define SP_OFF_R (SP_OFF_F + SP_TBW)
#define SP_OFF_M (SP_OFF_R + SP_TBW + SP_RHI)
#define SP_OFF_E (SP_OFF_M + SP_MW)
#define SP_OFF_O (SP_OFF_E + SP_MW / 2)
#define SP_OFF_LF (SP_OFF_O + SP_MW / 2)
#define SP_OFF_LR (SP_OFF_LF + SP_CAP + 256)
#define SP_OFF_MISC (SP_OFF_LR + SP_CAP + 256)
#define SP_LDS_DWORDS (SP_OFF_MISC + 16)

struct Planes {
    u32 P[SP_NP];
    u32 Q[SP_NQ];   // Q[l], l >= 2: carry of weight 2^l parked at level l, valid iff bit (l-2) of the quad count
};

__device__ __forceinline__ void planes_zero(Planes &c)
{
#pragma unroll
    for (int l = 0; l < SP_NP; l++) c.P[l] = 0;
#pragma unroll
    for (int l = 0; l < SP_NQ; l++) c.Q[l] = 0;
}

// carry-save adder on gfx950's 3-input boolean op: acc = a ^ b ^ acc, carry = maj(a, b, acc)
// (both truth tables are symmetric in their operands: 0x96 = odd parity, 0xE8 = majority)
__device__ __forceinline__ void csa(u32 &acc, u32 a, u32 b, u32 &carry)
{
    const u32 s = __builtin_amdgcn_bitop3_b32(acc, a, b, 0x96);
    carry = __builtin_amdgcn_bitop3_b32(acc, a, b, 0xE8);
    acc = s;
}

// add four 32-shift window words; quadcnt = quads this counter absorbed since its last flush (wave-uniform).
// The weight-4 carry is inserted like an increment of a binary counter: levels whose quadcnt bit is set hold a
// parked carry -> CSA and pass the carry up; the first level with a clear bit parks it.  One wave-uniform
// switch on the number of trailing one bits selects straight-line code (indices must stay compile-time
// constants or the planes leave the register file).
__device__ __forceinline__ void add_quad(Planes &c, u32 w0, u32 w1, u32 w2, u32 w3, u32 quadcnt)
{
    u32 c1a, c1b, c2;
    csa(c.P[0], w0, w1, c1a);
    csa(c.P[0], w2, w3, c1b);
    csa(c.P[1], c1a, c1b, c2);
#define SP_CSA(L)                              \
    {                                          \
        u32 nx_;                               \
        csa(c.P[L], c.Q[L], c2, nx_);          \
        c2 = nx_;                              \
    }
    const u32 tz = __builtin_ctz(~quadcnt);
    if (tz > 0u) {
        SP_CSA(2);
        if (tz > 1u) {
            SP_CSA(3);
            if (tz > 2u) {
                SP_CSA(4);
                if (tz > 3u) {
                    SP_CSA(5);
                    if (tz > 4u) {
                        SP_CSA(6);
                        if (tz > 5u) {
                            SP_CSA(7);
                            if (tz > 6u) {
                                SP_CSA(8);
                                if (tz > 7u) {
                                    SP_CSA(9);
                                    c.Q[10] = c2;
                                } else {
                                    c.Q[9] = c2;
                                    asm volatile("; park 9");   // keeps the per-level stores from being merged into one indexed store
                                }
                            } else {
                                c.Q[8] = c2;
                                asm volatile("; park 8");   // keeps the per-level stores from being merged into one indexed store
                            }
                        } else {
                            c.Q[7] = c2;
                            asm volatile("; park 7");   // keeps the per-level stores from being merged into one indexed store
                        }
                    } else {
                        c.Q[6] = c2;
                        asm volatile("; park 6");   // keeps the per-level stores from being merged into one indexed store
                    }
                } else {
                    c.Q[5] = c2;
                    asm volatile("; park 5");   // keeps the per-level stores from being merged into one indexed store
                }
            } else {
                c.Q[4] = c2;
                asm volatile("; park 4");   // keeps the per-level stores from being merged into one indexed store
            }
        } else {
            c.Q[3] = c2;
            asm volatile("; park 3");   // keeps the per-level stores from being merged into one indexed store
        }
    } else {
        c.Q[2] = c2;
        asm volatile("; park 2");   // keeps the per-level stores from being merged into one indexed store
    }
#undef SP_CSA
}

// integer count at bit i of this lane's 32 shifts
__device__ __forceinline__ u32 planes_value(const Planes &c, u32 quadcnt, u32 i)
{
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < SP_NP; k++) v += ((c.P[k] >> i) & 1u) << k;
#pragma unroll
    for (int k = 2; k < SP_NQ; k++)
        if ((quadcnt >> (k - 2)) & 1u) v += ((c.Q[k] >> i) & 1u) << k;
    return v;
}

// counters -> integers added into acc[i * 32 + l] (LDS), then cleared
__device__ __forceinline__ void planes_flush_lds(Planes &c, u32 quadcnt, u32 *acc, u32 l)
{
#ifdef SP_ABL_NOFLUSH
    planes_zero(c);
    return;
#endif
#pragma unroll 1
    for (u32 i = 0; i < 32; i++) {
        const u32 v = planes_value(c, quadcnt, i);
        if (v) atomicAdd(&acc[i * 32 + l], v);
    }
    planes_zero(c);
}

// same, straight into this workgroup's slab row with (uncontended) global atomics: used mid-tile, when the LDS
// tile buffers are live and a counter is about to overflow (only tiles with thousands of set bits get here)
__device__ __forceinline__ void planes_flush_slab(Planes &c, u32 quadcnt, u32 *__restrict__ row, u32 l)
{
#pragma unroll 1
    for (u32 i = 0; i < 32; i++) {
        const u32 v = planes_value(c, quadcnt, i);
        if (v) atomicAdd(&row[32 * l + i], v);
    }
    planes_zero(c);
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

struct TileRegs {
    uint4 f, r, m, h;   // main F / R / M quads of this thread + one halo quad (threads 0..34)
};

// thread t: main dwords 4t..4t+3 of each vector; halo quads: t in [0,17) M below, [17,26) M above, [26,35) R above
template <bool HAS_M, bool GUARD>
__device__ __forceinline__ void tile_fetch(TileRegs &tr, const u32 *__restrict__ F, const u32 *__restrict__ R,
                                           const u32 *__restrict__ M, int64_t d0, uint64_t nbits, u32 tid)
{
    const int64_t j = d0 + 4 * (int64_t)tid;
    tr.f = ld_quad<GUARD>(F, j, nbits);
    tr.r = ld_quad<GUARD>(R, j, nbits);
    if (HAS_M) tr.m = ld_quad<GUARD>(M, j, nbits);
    tr.h = make_uint4(0, 0, 0, 0);
    if (HAS_M && tid < 17)
        tr.h = ld_quad<GUARD>(M, d0 - (int64_t)SP_MLO + 4 * (int64_t)tid, nbits);
    else if (HAS_M && tid < 26)
        tr.h = ld_quad<GUARD>(M, d0 + SP_TBW + 4 * (int64_t)(tid - 17), nbits);
    else if (tid >= 26 && tid < 35)
        tr.h = ld_quad<GUARD>(R, d0 + SP_TBW + 4 * (int64_t)(tid - 26), nbits);
}

// 16 even bits of each of two dwords -> one dword (lo's bits in the low half)
__device__ __forceinline__ u32 even16x2(u32 lo, u32 hi)
{
    u32 a = lo & 0x55555555u, b = hi & 0x55555555u;
    a = (a | (a >> 1)) & 0x33333333u;
    b = (b | (b >> 1)) & 0x33333333u;
    a = (a | (a >> 2)) & 0x0f0f0f0fu;
    b = (b | (b >> 2)) & 0x0f0f0f0fu;
    a = a | (a >> 4);   // bytes 0 and 2 now hold 8 packed bits each
    b = b | (b >> 4);
    return (a & 0xffu) | ((a >> 8) & 0xff00u) | ((b & 0xffu) << 16) | ((b << 8) & 0xff000000u);
}

__device__ __forceinline__ void decimate_quad(const uint4 m, u32 *sE, u32 *sO, u32 dword_off)
{
    const u32 e0 = even16x2(m.x, m.y), e1 = even16x2(m.z, m.w);
    const u32 o0 = even16x2(m.x >> 1, m.y >> 1), o1 = even16x2(m.z >> 1, m.w >> 1);
    *reinterpret_cast<uint2 *>(sE + dword_off / 2) = make_uint2(e0, e1);
    *reinterpret_cast<uint2 *>(sO + dword_off / 2) = make_uint2(o0, o1);
}

template <bool HAS_M>
__device__ __forceinline__ void tile_store(const TileRegs &tr, u32 *lds, u32 tid)
{
    reinterpret_cast<uint4 *>(lds + SP_OFF_F)[tid] = tr.f;
    reinterpret_cast<uint4 *>(lds + SP_OFF_R)[tid] = tr.r;
    if (HAS_M) {
        reinterpret_cast<uint4 *>(lds + SP_OFF_M + SP_MLO)[tid] = tr.m;
#ifndef SP_ABL_NODEC
        decimate_quad(tr.m, lds + SP_OFF_E, lds + SP_OFF_O, SP_MLO + 4 * tid);
#endif
        if (tid < 17) {
            reinterpret_cast<uint4 *>(lds + SP_OFF_M)[tid] = tr.h;
            decimate_quad(tr.h, lds + SP_OFF_E, lds + SP_OFF_O, 4 * tid);
        } else if (tid < 26) {
            reinterpret_cast<uint4 *>(lds + SP_OFF_M + SP_MLO + SP_TBW)[tid - 17] = tr.h;
            decimate_quad(tr.h, lds + SP_OFF_E, lds + SP_OFF_O, SP_MLO + SP_TBW + 4 * (tid - 17));
        }
    }
    if (tid >= 26 && tid < 35) reinterpret_cast<uint4 *>(lds + SP_OFF_R + SP_TBW)[tid - 26] = tr.h;
}

// ---- set-bit lists -----------------------------------------------------------------------------------------

struct EmitState {
    uint4 w;      // this thread's 4 driver dwords
    u32 idx0;     // first list index of this thread (whole tile, not per round)
};

// reserves list slots for this thread's set bits with one LDS atomic (order across threads is irrelevant)
__device__ __forceinline__ EmitState emit_reserve(const uint4 w, u32 *cursor)
{
    EmitState st;
    st.w = w;
    const u32 n = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
    st.idx0 = n ? atomicAdd(cursor, n) : 0u;
    return st;
}

// writes the entries whose index falls in [round_lo, round_lo + SP_CAP)
template <bool WITH_FLAG>
__device__ __forceinline__ void emit_write(const EmitState &st, u32 round_lo, u32 *list, const u32 *sFlag, u32 flag_off,
                                           u32 tid)
{
    u32 id = st.idx0 - round_lo;   // unsigned: entries before the round wrap to huge values
    const u32 ws[4] = {st.w.x, st.w.y, st.w.z, st.w.w};
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        u32 ww = ws[k];
        while (ww) {
            const u32 b = __builtin_ctz(ww);
            ww &= ww - 1;
            if (id < SP_CAP) {
                const u32 pos = 128u * tid + 32u * k + b;
                u32 e = pos | SP_VALID;
                if (WITH_FLAG) {
                    const u32 fa = flag_off + pos;
                    e |= ((sFlag[fa >> 5] >> (fa & 31)) & 1u) << 16;
                }
                list[id] = e;
            }
            id++;
        }
    }
}

// pads the list to a multiple of pad_to (<= 256) entries with invalid entries (position 0, no flags)
__device__ __forceinline__ void emit_pad(u32 *list, u32 nround, u32 pad_to, u32 tid)
{
    const u32 npad = (nround + pad_to - 1) / pad_to * pad_to;
    if (nround + tid < npad) list[nround + tid] = 0u;
}

struct SlotGeom {
    u32 total_slots, slot, l, quad_span;
};

__device__ __forceinline__ SlotGeom slot_geom(u32 lgG, u32 tid)
{
    SlotGeom g;
    const u32 lane = tid & 63, wave = tid >> 6;
    const u32 spw = 64u >> lgG;
    g.total_slots = 4 * spw;
    g.slot = wave * spw + (lane >> lgG);
    g.l = lane & ((1u << lgG) - 1);
    g.quad_span = g.total_slots * 4;
    return g;
}

// ---- the cross-correlation kernel ---------------------------------------------------------------------------

template <bool HAS_M, bool DO_NCC>
__device__ __forceinline__ void cc_flush_tile_boundary(Planes &cN, Planes &cF, Planes &cC, Planes &cR, u32 &qF, u32 &qR,
                                                       u32 *lds, u32 l, u32 tid, u32 *__restrict__ slab)
{
    u32 *acc = lds;   // 4 x 1024 u32 over the (dead) tile buffers
    __syncthreads();
    for (u32 i = tid; i < 4096; i += 256) acc[i] = 0;
    __syncthreads();
    if (DO_NCC) planes_flush_lds(cN, qF, acc, l);
    if (HAS_M) {
        planes_flush_lds(cF, qF, acc + 1024, l);
        planes_flush_lds(cC, qF, acc + 2048, l);
        planes_flush_lds(cR, qR, acc + 3072, l);
    }
    qF = 0;
    qR = 0;
    __syncthreads();
    // every slab update is an L2 atomic (uncontended: the slab is private), so mid-tile spills and these adds
    // can never read each other's stale L1 lines
    for (u32 d = tid; d < 1024; d += 256) {
        const u32 a = (d & 31) * 32 + (d >> 5);
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            const u32 v = acc[q * 1024 + a];
            if (v) atomicAdd(&slab[q * 1024 + d], v);
        }
    }
    __syncthreads();
}

template <bool HAS_M, bool DO_NCC>
__global__ void __launch_bounds__(256, 2)
k_cc_sparse(const u32 *__restrict__ F, const u32 *__restrict__ R, const u32 *__restrict__ M, uint64_t nbits, int32_t c,
            u32 lgG, u32 ntiles, u32 aligned16, u32 *__restrict__ slab_all)
{
    __shared__ __align__(16) u32 lds[SP_LDS_DWORDS];
    u32 *const listF = lds + SP_OFF_LF;
    u32 *const listR = lds + SP_OFF_LR;
    u32 *const cursor = lds + SP_OFF_MISC;   // [0] = F entries, [1] = R entries of the current tile
    const u32 *const sR = lds + SP_OFF_R;
    const u32 *const sM = lds + SP_OFF_M;
    const u32 *const sE = lds + SP_OFF_E;

    const u32 tid = threadIdx.x;
    const SlotGeom sg = slot_geom(lgG, tid);
    const u32 l = sg.l;
    u32 *const slab = slab_all + (size_t)blockIdx.x * SP_SLAB_ROWS * 1024;

    // the slab is private to this workgroup for the whole launch; the host zeroes it before the launch

    Planes cN, cF, cC, cR;
    planes_zero(cN);
    planes_zero(cF);
    planes_zero(cC);
    planes_zero(cR);
    u32 qF = 0, qR = 0;           // quads absorbed since the last flush (workgroup-uniform)
    u32 totF = 0, totR = 0;       // set bits seen by this workgroup (uniform)
    u32 cntR_thread = 0;          // NCC-only mode: popcount of R accumulated per thread

    // per-lane window offsets
    const u32 kM = (u32)c + SP_HALO_M - 32u * l - 31u;   // M[x + c - d], descending
    const u32 k1 = SP_HALO_M - 32u * l - 31u;            // M[p - d], descending
    const u32 kB = SP_HALO_M + (u32)c;                   // M[p + c - 2d]: position in the decimated copies
    const u32 k2 = 32u * l + 31u;

    // edge tiles (or unaligned vectors) take the guarded loader
    const uint64_t full_dw = nbits / 32;
    auto interior = [&](u32 t) -> bool {
        if (!aligned16 || t == 0) return false;
        const uint64_t hi = (uint64_t)t * SP_TBW + SP_TBW + SP_MHI;   // one past the highest dword touched
        return hi + 2 <= full_dw;
    };

    TileRegs tr;
    u32 tile = blockIdx.x;
    if (tile < ntiles) {
        if (interior(tile))
            tile_fetch<HAS_M, false>(tr, F, R, M, (int64_t)tile * SP_TBW, nbits, tid);
        else
            tile_fetch<HAS_M, true>(tr, F, R, M, (int64_t)tile * SP_TBW, nbits, tid);
    }

    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();   // everyone is done with the previous tile's LDS
        if (qF >= SP_QSOFT || qR >= SP_QSOFT)
            cc_flush_tile_boundary<HAS_M, DO_NCC>(cN, cF, cC, cR, qF, qR, lds, l, tid, slab);

        tile_store<HAS_M>(tr, lds, tid);
        if (tid < 2) cursor[tid] = 0;
        if (!HAS_M) cntR_thread += __popc(tr.r.x) + __popc(tr.r.y) + __popc(tr.r.z) + __popc(tr.r.w);
        const uint4 wf = tr.f, wr = tr.r;
        __syncthreads();   // tile visible

        // fetch the next tile into registers now; it is consumed after the next loop-top barrier
        {
            const u32 nt = tile + gridDim.x;
            if (nt < ntiles) {
                if (interior(nt))
                    tile_fetch<HAS_M, false>(tr, F, R, M, (int64_t)nt * SP_TBW, nbits, tid);
                else
                    tile_fetch<HAS_M, true>(tr, F, R, M, (int64_t)nt * SP_TBW, nbits, tid);
            }
        }

        // compact both driver vectors
        const EmitState eF = emit_reserve(wf, &cursor[0]);
        EmitState eR;
        if (HAS_M) eR = emit_reserve(wr, &cursor[1]);
#ifndef SP_ABL_NOEMIT
        emit_write<HAS_M>(eF, 0, listF, sM, SP_HALO_M, tid);
        if (HAS_M) emit_write<false>(eR, 0, listR, nullptr, 0, tid);
#endif
        __syncthreads();   // cursors final, first round written
#ifdef SP_ABL_NOPROC
        const u32 nF = 0, nR = 0;
#else
        const u32 nF = cursor[0], nR = HAS_M ? cursor[1] : 0u;
#endif
        totF += nF;
        totR += nR;
        const u32 nmax = nF > nR ? nF : nR;

        for (u32 round_lo = 0; round_lo < nmax; round_lo += SP_CAP) {
            if (round_lo) {   // rare: a tile with more than SP_CAP set bits in one vector
                __syncthreads();
                emit_write<HAS_M>(eF, round_lo, listF, sM, SP_HALO_M, tid);
                if (HAS_M) emit_write<false>(eR, round_lo, listR, nullptr, 0, tid);
            }
            const u32 nFr = nF > round_lo ? (nF - round_lo < SP_CAP ? nF - round_lo : SP_CAP) : 0u;
            const u32 nRr = nR > round_lo ? (nR - round_lo < SP_CAP ? nR - round_lo : SP_CAP) : 0u;
            emit_pad(listF, nFr, sg.quad_span, tid);
            if (HAS_M) emit_pad(listR, nRr, sg.quad_span, tid);
            __syncthreads();
            const u32 nqF = (nFr + sg.quad_span - 1) / sg.quad_span;
            const u32 nqR = (nRr + sg.quad_span - 1) / sg.quad_span;

            // a counter must never absorb more than SP_QLIMIT quads: spill to the slab first (dense tiles only)
            if (qF + nqF > SP_QLIMIT) {
                if (DO_NCC) planes_flush_slab(cN, qF, slab, l);
                if (HAS_M) {
                    planes_flush_slab(cF, qF, slab + 1024, l);
                    planes_flush_slab(cC, qF, slab + 2048, l);
                }
                qF = 0;
            }
            if (HAS_M && qR + nqR > SP_QLIMIT) {
                planes_flush_slab(cR, qR, slab + 3072, l);
                qR = 0;
            }

            // ---- forward reads drive: ncc, mscc.fsum, mscc.ccbins ----
            for (u32 q = 0; q < nqF; q++) {
                u32 wN[4], wF[4], wC[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 e = listF[(q * 4 + k) * sg.total_slots + sg.slot];
                    const u32 vmask = (u32)__builtin_amdgcn_sbfe(e, 17, 1);   // all ones for a real entry
                    const u32 ri = ((e >> 5) & 1023u) + l;
                    const u32 rw = __builtin_amdgcn_alignbit(sR[ri + 1], sR[ri], e) & vmask;
                    wN[k] = rw;
                    if (HAS_M) {
                        const u32 fmask = (u32)__builtin_amdgcn_sbfe(e, 16, 1);   // forward read is mappable
                        const u32 a = (e & 0x7fffu) + kM;
                        const u32 mw = __builtin_bitreverse32(__builtin_amdgcn_alignbit(sM[(a >> 5) + 1], sM[a >> 5], a));
                        wF[k] = mw & fmask;
                        wC[k] = __builtin_amdgcn_bitop3_b32(mw, rw, fmask, 0x80);   // three-way AND
                    }
                }
                const u32 qc = __builtin_amdgcn_readfirstlane(qF + q);
                if (DO_NCC) add_quad(cN, wN[0], wN[1], wN[2], wN[3], qc);
                if (HAS_M) {
                    add_quad(cF, wF[0], wF[1], wF[2], wF[3], qc);
                    add_quad(cC, wC[0], wC[1], wC[2], wC[3], qc);
                }
            }
            qF += nqF;

            // ---- reverse reads drive: mscc.rsum ----
            if (HAS_M) {
                for (u32 q = 0; q < nqR; q++) {
                    u32 wR[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const u32 e = listR[(q * 4 + k) * sg.total_slots + sg.slot];
                        const u32 vmask = (u32)__builtin_amdgcn_sbfe(e, 17, 1);
                        const u32 p = e & 0x7fffu;
                        const u32 a1 = p + k1;
                        const u32 w1 = __builtin_amdgcn_alignbit(sM[(a1 >> 5) + 1], sM[a1 >> 5], a1);
                        const u32 b = p + kB;
                        const u32 a2 = (b >> 1) - k2;
                        const u32 *dec = sE + (b & 1u) * (SP_MW / 2);   // even or odd decimated copy
                        const u32 w2 = __builtin_amdgcn_alignbit(dec[(a2 >> 5) + 1], dec[a2 >> 5], a2);
                        wR[k] = __builtin_bitreverse32(w1 & w2) & vmask;
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qR + q);
                    add_quad(cR, wR[0], wR[1], wR[2], wR[3], qc);
                }
                qR += nqR;
            }
        }
    }

    __syncthreads();
    cc_flush_tile_boundary<HAS_M, DO_NCC>(cN, cF, cC, cR, qF, qR, lds, l, tid, slab);
    // popcount(F), popcount(R): bit_array_num_bits_set of mscc.pyx:236-237, for free from the compaction
    if (!HAS_M) {
        u32 v = cntR_thread;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) cursor[4 + (tid >> 6)] = v;
        __syncthreads();
        if (tid == 0) totR = cursor[4] + cursor[5] + cursor[6] + cursor[7];
    }
    if (tid == 0) {
        slab[4 * 1024 + 0] = totF;
        slab[4 * 1024 + 1] = totR;
    }
}

// dst[r][i] = sum over workgroups of slab[wg][src_row[r]][i], i < n[r]; blockIdx.y = r
struct ReduceRows {
    u64 *dst[5];
    u32 src_row[5];
    u32 n[5];
};

__global__ void __launch_bounds__(256) k_reduce_slab(const u32 *__restrict__ slab, u32 nwg, u32 rows_per_wg, ReduceRows rr)
{
    // 32 consecutive elements x 8 workgroup phases per block: every load is a full 128-B line and each thread
    // keeps 8 independent loads in flight
    __shared__ u64 part[8][32];
    const u32 r = blockIdx.y;
    const u32 e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const u32 i = blockIdx.x * 32 + e;
    const u32 n = rr.n[r];
    u64 sum = 0;
    if (i < n) {
        const size_t stride = (size_t)rows_per_wg * 1024;
        const u32 *p = slab + (size_t)rr.src_row[r] * 1024 + i;
        u32 w = g;
        for (; w + 56 < nwg; w += 64) {
            u32 v[8];
#pragma unroll
            for (u32 k = 0; k < 8; k++) v[k] = p[(size_t)(w + 8 * k) * stride];
#pragma unroll
            for (u32 k = 0; k < 8; k++) sum += v[k];
        }
        for (; w < nwg; w += 8) sum += p[(size_t)w * stride];
    }
    part[g][e] = sum;
    __syncthreads();
    if (g == 0 && i < n) {
        u64 t = 0;
#pragma unroll
        for (u32 k = 0; k < 8; k++) t += part[k][e];
        rr.dst[r][i] = t;
    }
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
#define AC_W (SP_TBW + SP_RHI)       // dwords of U / D per tile
#define AC_OFF_U 0u
#define AC_OFF_D (AC_OFF_U + AC_W)
#define AC_OFF_L (AC_OFF_D + AC_W)
#define AC_OFF_MISC (AC_OFF_L + SP_CAP + 256)
#define AC_LDS_DWORDS (AC_OFF_MISC + 16)

struct AcRegs {
    uint4 m, h;   // main quad + (threads 0..8) the quad above the tile
    u32 below;    // dword just below this thread's main quad (for M[j-1])
    u32 hbelow;
};

template <bool GUARD>
__device__ __forceinline__ void ac_fetch(AcRegs &ar, const u32 *__restrict__ M, int64_t d0, uint64_t nbits, u32 tid)
{
    const int64_t j = d0 + 4 * (int64_t)tid;
    ar.m = ld_quad<GUARD>(M, j, nbits);
    ar.below = GUARD ? ld_dword_guarded(M, j - 1, nbits) : M[j - 1];
    ar.h = make_uint4(0, 0, 0, 0);
    ar.hbelow = 0;
    if (tid < 9) {
        const int64_t jh = d0 + SP_TBW + 4 * (int64_t)tid;
        ar.h = ld_quad<GUARD>(M, jh, nbits);
        ar.hbelow = GUARD ? ld_dword_guarded(M, jh - 1, nbits) : M[jh - 1];
    }
}

// rising (U) and falling (D) edge words of a quad; `below` = the dword preceding m.x
__device__ __forceinline__ void edge_quad(const uint4 m, u32 below, uint4 &U, uint4 &D)
{
    const u32 s0 = (m.x << 1) | (below >> 31), s1 = (m.y << 1) | (m.x >> 31);
    const u32 s2 = (m.z << 1) | (m.y >> 31), s3 = (m.w << 1) | (m.z >> 31);
    U = make_uint4(m.x & ~s0, m.y & ~s1, m.z & ~s2, m.w & ~s3);
    D = make_uint4(~m.x & s0, ~m.y & s1, ~m.z & s2, ~m.w & s3);
}

__device__ __forceinline__ void ac_flush_boundary(Planes &cP, Planes &cN, u32 &qc, u32 *lds, u32 l, u32 tid,
                                                  u32 *__restrict__ slab)
{
    u32 *acc = lds;
    __syncthreads();
    for (u32 i = tid; i < 2048; i += 256) acc[i] = 0;
    __syncthreads();
    planes_flush_lds(cP, qc, acc, l);
    planes_flush_lds(cN, qc, acc + 1024, l);
    qc = 0;
    __syncthreads();
    for (u32 k = tid; k < 1024; k += 256) {
        const u32 a = (k & 31) * 32 + (k >> 5);
        const u32 vp = acc[a], vn = acc[1024 + a];
        if (vp) atomicAdd(&slab[k], vp);
        if (vn) atomicAdd(&slab[1024 + k], vn);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256, 3)
k_autocorr_edges(const u32 *__restrict__ M, uint64_t nbits, u32 lgG, u32 ntiles, u32 aligned16,
                 u32 *__restrict__ slab_all)
{
    __shared__ __align__(16) u32 lds[AC_LDS_DWORDS];
    const u32 *const sU = lds + AC_OFF_U;
    const u32 *const sD = lds + AC_OFF_D;
    u32 *const list = lds + AC_OFF_L;
    u32 *const cursor = lds + AC_OFF_MISC;

    const u32 tid = threadIdx.x;
    const SlotGeom sg = slot_geom(lgG, tid);
    const u32 l = sg.l;
    u32 *const slab = slab_all + (size_t)blockIdx.x * 3 * 1024;   // rows: P, N, scalars

    Planes cP, cN;
    planes_zero(cP);
    planes_zero(cN);
    u32 qc = 0;
    u32 cntM = 0, cntU = 0;

    const uint64_t full_dw = nbits / 32;
    auto interior = [&](u32 t) -> bool {
        if (!aligned16 || t == 0) return false;
        const uint64_t hi = (uint64_t)t * SP_TBW + SP_TBW + SP_RHI;
        return hi + 2 <= full_dw;
    };

    AcRegs ar;
    u32 tile = blockIdx.x;
    if (tile < ntiles) {
        if (interior(tile))
            ac_fetch<false>(ar, M, (int64_t)tile * SP_TBW, nbits, tid);
        else
            ac_fetch<true>(ar, M, (int64_t)tile * SP_TBW, nbits, tid);
    }

    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        if (qc >= SP_QSOFT) ac_flush_boundary(cP, cN, qc, lds, l, tid, slab);

        uint4 U, D;
        edge_quad(ar.m, ar.below, U, D);
        reinterpret_cast<uint4 *>(lds + AC_OFF_U)[tid] = U;
        reinterpret_cast<uint4 *>(lds + AC_OFF_D)[tid] = D;
        const uint4 E = make_uint4(U.x | D.x, U.y | D.y, U.z | D.z, U.w | D.w);
        cntM += __popc(ar.m.x) + __popc(ar.m.y) + __popc(ar.m.z) + __popc(ar.m.w);
        cntU += __popc(U.x) + __popc(U.y) + __popc(U.z) + __popc(U.w);
        if (tid < 9) {
            uint4 Uh, Dh;
            edge_quad(ar.h, ar.hbelow, Uh, Dh);
            reinterpret_cast<uint4 *>(lds + AC_OFF_U + SP_TBW)[tid] = Uh;
            reinterpret_cast<uint4 *>(lds + AC_OFF_D + SP_TBW)[tid] = Dh;
        }
        if (tid == 0) cursor[0] = 0;
        __syncthreads();

        {
            const u32 nt = tile + gridDim.x;
            if (nt < ntiles) {
                if (interior(nt))
                    ac_fetch<false>(ar, M, (int64_t)nt * SP_TBW, nbits, tid);
                else
                    ac_fetch<true>(ar, M, (int64_t)nt * SP_TBW, nbits, tid);
            }
        }

        const EmitState es = emit_reserve(E, &cursor[0]);
        emit_write<true>(es, 0, list, sD, 0, tid);   // flag = falling edge
        __syncthreads();
        const u32 n = cursor[0];
        for (u32 round_lo = 0; round_lo < n; round_lo += SP_CAP) {
            if (round_lo) {
                __syncthreads();
                emit_write<true>(es, round_lo, list, sD, 0, tid);
            }
            const u32 nr = n - round_lo < SP_CAP ? n - round_lo : SP_CAP;
            emit_pad(list, nr, sg.quad_span, tid);
            __syncthreads();
            const u32 nq = (nr + sg.quad_span - 1) / sg.quad_span;
            if (qc + nq > SP_QLIMIT) {
                planes_flush_slab(cP, qc, slab, l);
                planes_flush_slab(cN, qc, slab + 1024, l);
                qc = 0;
            }
            for (u32 q = 0; q < nq; q++) {
                u32 wp[4], wn[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 e = list[(q * 4 + k) * sg.total_slots + sg.slot];
                    const u32 vmask = (u32)__builtin_amdgcn_sbfe(e, 17, 1);
                    const u32 fall = (u32)__builtin_amdgcn_sbfe(e, 16, 1);
                    const u32 wi = ((e >> 5) & 1023u) + l;
                    const u32 wu = __builtin_amdgcn_alignbit(sU[wi + 1], sU[wi], e) & vmask;
                    const u32 wd = __builtin_amdgcn_alignbit(sD[wi + 1], sD[wi], e) & vmask;
                    wp[k] = (fall & wd) | (~fall & wu);   // same-sign pairs: U*U, D*D
                    wn[k] = (fall & wu) | (~fall & wd);   // opposite-sign pairs: U*D, D*U
                }
                const u32 qq = __builtin_amdgcn_readfirstlane(qc + q);
                add_quad(cP, wp[0], wp[1], wp[2], wp[3], qq);
                add_quad(cN, wn[0], wn[1], wn[2], wn[3], qq);
            }
            qc += nq;
        }
    }

    // retire: counters -> slab rows, popcount(M) and #runs -> scalar row
    __syncthreads();
    ac_flush_boundary(cP, cN, qc, lds, l, tid, slab);
    for (int off = 32; off > 0; off >>= 1) {
        cntM += __shfl_down(cntM, off, 64);
        cntU += __shfl_down(cntU, off, 64);
    }
    if ((tid & 63) == 0) {
        cursor[4 + (tid >> 6)] = cntM;
        cursor[8 + (tid >> 6)] = cntU;
    }
    __syncthreads();
    if (tid == 0) {
        slab[2048 + 0] = cursor[4] + cursor[5] + cursor[6] + cursor[7];
        slab[2048 + 1] = cursor[8] + cursor[9] + cursor[10] + cursor[11];
    }
}

// A(k) recurrence + output.  mode 0: out[k] = A(k), k = 0..max_lag.  mode 1: out[d] = A(|c - d|), d = 0..max_shift.
__global__ void __launch_bounds__(256)
k_autocorr_finish(const u64 *__restrict__ P, const u64 *__restrict__ N, const u64 *__restrict__ scal, u32 max_lag,
                  u32 mode, int32_t c, u32 max_shift, u64 *__restrict__ out, u64 *__restrict__ popcount_out)
{
    __shared__ long long A[1025];
    if (threadIdx.x == 0) {
        long long a_prev = (long long)scal[0];            // A(0) = popcount(M)
        A[0] = a_prev;
        if (max_lag >= 1) {
            long long a = a_prev - (long long)scal[1];    // A(1) = A(0) - #runs
            A[1] = a;
            for (u32 k = 1; k < max_lag; k++) {
                const long long ee = (long long)P[k] - (long long)N[k];
                const long long nxt = 2 * a - a_prev - ee;
                a_prev = a;
                a = nxt;
                A[k + 1] = a;
            }
        }
        if (popcount_out) *popcount_out = scal[0];
    }
    __syncthreads();
    if (mode == 0) {
        for (u32 k = threadIdx.x; k <= max_lag; k += 256) out[k] = (u64)A[k];
    } else {
        for (u32 d = threadIdx.x; d <= max_shift; d += 256) {
            const int32_t k = c - (int32_t)d;
            out[d] = (u64)A[k < 0 ? -k : k];
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------

static uint32_t lg_slot_lanes(uint32_t max_shift)
{
    const u32 need = (max_shift + 1 + 31) / 32;
    u32 lg = 2;   // G >= 4 keeps the list padding granule (16 * 64 / G entries) within one workgroup pass
    while ((1u << lg) < need) lg++;
    return lg;
}

static uint32_t sparse_grid(pmx_ctx *ctx, uint64_t ntiles, uint32_t wg_per_cu)
{
    uint64_t gx = (uint64_t)ctx->num_cus * wg_per_cu;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    return (uint32_t)gx;
}

static inline bool is_aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// d_tmp: 2 * 1024 + 16 u64 of scratch (P, N, scalars).
int pmx_launch_autocorr_edges(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_lag, u64 *d_tmp,
                              uint32_t mode, uint32_t read_len, uint32_t max_shift, u64 *d_out, u64 *d_popcount_out)
{
    const uint64_t ntiles = (nbits + 1 + SP_TB - 1) / SP_TB;     // E lives on [0, nbits]
    const uint32_t gx = sparse_grid(ctx, ntiles, 3);
    int rc = pmx_ensure_slab(ctx, (size_t)gx * 3 * 1024);
    if (rc) return rc;
    u64 *P = d_tmp, *N = d_tmp + 1024, *scal = d_tmp + 2048;
    PMX_HIP(hipMemsetAsync(ctx->d_slab, 0, (size_t)gx * 3 * 1024 * sizeof(u32), ctx->stream));
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_AUTOCORR, &tl);
    if (rc) return rc;
    hipLaunchKernelGGL(k_autocorr_edges, dim3(gx), dim3(256), 0, ctx->stream, (const u32 *)d_M, nbits,
                       lg_slot_lanes(max_lag), (u32)ntiles, (u32)is_aligned16(d_M), ctx->d_slab);
    PMX_CHECK_LAUNCH("k_autocorr_edges");
    rc = pmx_prof_end(ctx, &tl);
    if (rc) return rc;
    ReduceRows rr = {};
    rr.dst[0] = P; rr.src_row[0] = 0; rr.n[0] = max_lag + 1;
    rr.dst[1] = N; rr.src_row[1] = 1; rr.n[1] = max_lag + 1;
    rr.dst[2] = scal; rr.src_row[2] = 2; rr.n[2] = 2;
    hipLaunchKernelGGL(k_reduce_slab, dim3((max_lag + 32) / 32, 3), dim3(256), 0, ctx->stream,
                       (const u32 *)ctx->d_slab, gx, 3u, rr);
    PMX_CHECK_LAUNCH("k_reduce_slab");
    hipLaunchKernelGGL(k_autocorr_finish, dim3(1), dim3(256), 0, ctx->stream, (const u64 *)P, (const u64 *)N,
                       (const u64 *)scal, max_lag, mode, (int32_t)read_len - 1, max_shift, d_out, d_popcount_out);
    PMX_CHECK_LAUNCH("k_autocorr_finish");
    return PMX_OK;
}

int pmx_sparse_supported(uint32_t max_shift, uint32_t read_len)
{
    return max_shift >= 3 && max_shift <= 1023 && read_len >= 1 && read_len <= 1024;
}

int pmx_launch_cc_sparse(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M,
                         uint64_t nbits, uint32_t max_shift, uint32_t read_len, bool do_ncc, u64 *d_out,
                         uint32_t out_stride)
{
    if (!d_M && !do_ncc) return PMX_OK;
    uint64_t ntiles = (nbits + SP_TB - 1) / SP_TB;
    if (ntiles < 1) ntiles = 1;
    const int32_t c = (int32_t)read_len - 1;
    const uint32_t gx = sparse_grid(ctx, ntiles, d_M ? 2 : 4);
    int rc = pmx_ensure_slab(ctx, (size_t)gx * SP_SLAB_ROWS * 1024);
    if (rc) return rc;
    const dim3 grid(gx), block(256);
    const u32 *F = (const u32 *)d_F, *R = (const u32 *)d_R, *M = (const u32 *)d_M;
    const u32 al = is_aligned16(d_F) && is_aligned16(d_R) && (!d_M || is_aligned16(d_M));
    const u32 lgG = lg_slot_lanes(max_shift);
    PMX_HIP(hipMemsetAsync(ctx->d_slab, 0, (size_t)gx * SP_SLAB_ROWS * 1024 * sizeof(u32), ctx->stream));
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_SPARSE, &tl);
    if (rc) return rc;
#define SP_LAUNCH(HM, NC)                                                                                          \
    hipLaunchKernelGGL((k_cc_sparse<HM, NC>), grid, block, 0, ctx->stream, F, R, M, nbits, c, lgG, (u32)ntiles, al, \
                       ctx->d_slab)
    if (d_M && do_ncc)
        SP_LAUNCH(true, true);
    else if (d_M)
        SP_LAUNCH(true, false);
    else
        SP_LAUNCH(false, true);
#undef SP_LAUNCH
    PMX_CHECK_LAUNCH("k_cc_sparse");
    rc = pmx_prof_end(ctx, &tl);
    if (rc) return rc;
    // sum the per-workgroup slabs into the result block
    u64 *scal = d_out + (size_t)PMX_ROW_SCALARS * out_stride;
    ReduceRows rr = {};
    u32 nr = 0;
    if (do_ncc) { rr.dst[nr] = d_out + (size_t)PMX_ROW_NCC_CCBINS * out_stride; rr.src_row[nr] = 0; rr.n[nr] = max_shift + 1; nr++; }
    if (d_M) {
        rr.dst[nr] = d_out + (size_t)PMX_ROW_MSCC_FSUM * out_stride; rr.src_row[nr] = 1; rr.n[nr] = max_shift + 1; nr++;
        rr.dst[nr] = d_out + (size_t)PMX_ROW_MSCC_CCBINS * out_stride; rr.src_row[nr] = 2; rr.n[nr] = max_shift + 1; nr++;
        rr.dst[nr] = d_out + (size_t)PMX_ROW_MSCC_RSUM * out_stride; rr.src_row[nr] = 3; rr.n[nr] = max_shift + 1; nr++;
    }
    rr.dst[nr] = scal; rr.src_row[nr] = 4; rr.n[nr] = 2; nr++;
    hipLaunchKernelGGL(k_reduce_slab, dim3((max_shift + 32) / 32, nr), dim3(256), 0, ctx->stream,
                       (const u32 *)ctx->d_slab, gx, (u32)SP_SLAB_ROWS, rr);
    PMX_CHECK_LAUNCH("k_reduce_slab");
    PMX_HIP(hipMemsetD32Async((hipDeviceptr_t)(scal + 3), PMX_PATH_SPARSE, 1, ctx->stream));
    return PMX_OK;
}
