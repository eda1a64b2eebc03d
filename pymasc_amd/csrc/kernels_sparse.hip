// Set-bit driven cross-correlation kernel (gfx950): the fast path for read-occupancy vectors, which
// are sparse by nature (at most one bit per position and strand; ~0.5 % density for 15 M reads / 3.1 Gbp).
//
// Same outputs as the reference's per-shift loop (PyMaSC/core/bitarray/mscc.pyx:288-317), different
// algorithm: instead of sliding the whole R vector past F once per shift (N/64 words x (S+1) shifts), every
// SET BIT x of the driver vector contributes a whole (S+1)-bit WINDOW of the other vector(s):
//     ncc[d]        += R[x + d]                                   for x in F            (mscc.pyx:314)
//     mscc.fsum[d]  += M[x + c - d]                               for x in F & M, c=L-1 (mscc.pyx:300,303)
//     mscc.cc[d]    += R[x + d] & M[x + c - d]                    for x in F & M        (mscc.pyx:305)
//     mscc.rsum[d]  += M[p - d] & M[p + c - 2d]                   for p in R            (mscc.pyx:301,304)
// (D_d[j] = M[j] & M[j+c-d], mscc.pyx:291; the last line substitutes p = j + d.)  Windows are summed
// position-wise with bit-sliced carry-save counters: lane l of a slot owns shifts 32l..32l+31 as ONE 32-bit
// word per counter bit plane, so adding a window costs ~3 VALU ops per 32 shifts instead of 32 adds.
//
// Work layout: persistent workgroups of 4 wavefronts walk 32-Kbit tiles.  Per tile the F / R / M words
// (+ halos of S bits) are staged in LDS with coalesced loads, the tile's set bits are compacted into an LDS
// list (wave prefix sums), and each SLOT (G lanes, G = 2^k >= (S+1)/32; 64/G slots per wave) pulls list
// entries, fetches its window words from LDS (two dwords + v_alignbit, v_bfrev for descending windows) and
// feeds them to its counters four at a time (3 static carry-save adders + a binary-counter style insertion
// whose branch is wave-uniform).  The stride-2 window of rsum reads from even/odd decimated copies of the M
// tile built once per tile.  Counters are converted to integers only when they could overflow or when the
// workgroup retires: LDS atomic adds, then one 64-bit global atomic per shift and row.
#include "pmx_common.h"

#define SP_TB 32768u                 // driver bits per tile
#define SP_TBW (SP_TB / 32u)         // dwords per tile
#define SP_TBW64 (SP_TB / 64u)
#define SP_CAP 2048u                 // list entries per round
#define SP_K 13                      // counter bit planes
#define SP_QLIMIT ((1u << (SP_K - 2)) - 1u)   // max quads between flushes
#define SP_INVALID 0x80000000u

struct SparseGeom {
    u32 G;            // lanes per slot
    u32 halo_m;       // bits of M staged below the tile start (multiple of 64)
    u32 rlen64;       // 64-bit words in the R tile
    u32 mlen64;       // 64-bit words in the M tile
    u32 lds_bytes;
};

static SparseGeom sparse_geom(u32 max_shift, u32 read_len, bool has_m)
{
    SparseGeom g;
    u32 need = (max_shift + 1 + 31) / 32;
    u32 G = 1;
    while (G < need) G <<= 1;
    if (G < 4) G = 4;   // keeps the list padding granule (16 * 64 / G entries) within one workgroup pass
    g.G = G;
    g.halo_m = 64 * G + 64;
    g.rlen64 = (SP_TB + 32 * G + 64 + 63) / 64;
    g.mlen64 = has_m ? (SP_TB + g.halo_m + (read_len - 1) + 64 + 63) / 64 : 0;
    u32 bytes = SP_TBW64 * 8 + g.rlen64 * 8 + g.mlen64 * 8 /*sM*/ + g.mlen64 * 8 /*sME+sMO*/ + (SP_CAP + 64) * 4 + 64;
    if (bytes < 4 * 1024 * 4 + 64) bytes = 4 * 1024 * 4 + 64;   // the flush accumulators alias the tile buffers
    g.lds_bytes = bytes;
    return g;
}

struct Planes {
    u32 P[SP_K];
    u32 Q[SP_K];   // pending carry of weight 2^l parked at level l (valid iff bit (l-2) of the quad count)
};

__device__ __forceinline__ void planes_zero(Planes &c)
{
#pragma unroll
    for (int l = 0; l < SP_K; l++) {
        c.P[l] = 0;
        c.Q[l] = 0;
    }
}

// carry-save adder: (acc, a, b) -> acc = parity, carry = majority
__device__ __forceinline__ void csa(u32 &acc, u32 a, u32 b, u32 &carry)
{
    const u32 t = acc ^ a;
    carry = (t & b) | (~t & acc);
    acc = t ^ b;
}

// add four 32-shift window words; quadcnt = quads added to this counter before (wave-uniform)
__device__ __forceinline__ void add_quad(Planes &c, u32 w0, u32 w1, u32 w2, u32 w3, u32 quadcnt)
{
    u32 c1a, c1b, c2;
    csa(c.P[0], w0, w1, c1a);
    csa(c.P[0], w2, w3, c1b);
    csa(c.P[1], c1a, c1b, c2);
    // binary-counter insertion: levels below the first zero bit of quadcnt hold a parked carry -> CSA and
    // pass the carry up; the level at the first zero bit parks the carry.  All branches are wave-uniform.
    const u32 tz = __builtin_ctz(~quadcnt);
#pragma unroll
    for (int l = 2; l < SP_K; l++) {
        if ((u32)(l - 2) < tz) {
            u32 cy;
            csa(c.P[l], c.Q[l], c2, cy);
            c2 = cy;
        } else if ((u32)(l - 2) == tz) {
            c.Q[l] = c2;
        }
    }
}

// counters -> integers, added into acc[i * 32 + l] (LDS), then cleared
__device__ __forceinline__ void planes_flush(Planes &c, u32 quadcnt, u32 *acc, u32 l)
{
#pragma unroll 1
    for (u32 i = 0; i < 32; i++) {
        u32 v = 0;
#pragma unroll
        for (int k = 0; k < SP_K; k++) v += ((c.P[k] >> i) & 1u) << k;
#pragma unroll
        for (int k = 2; k < SP_K; k++)
            if ((quadcnt >> (k - 2)) & 1u) v += ((c.Q[k] >> i) & 1u) << k;
        if (v) atomicAdd(&acc[i * 32 + l], v);
    }
    planes_zero(c);
}

__device__ __forceinline__ u64 sp_ld_word(const u64 *__restrict__ p, int64_t idx, uint64_t nwords, uint64_t nbits)
{
    if (idx < 0 || (uint64_t)idx >= nwords) return 0;
    u64 w = p[idx];
    if ((uint64_t)idx == nwords - 1 && (nbits & 63)) w &= ~0ull >> (64 - (nbits & 63));
    return w;
}

// the 32 even bits of m, packed
__device__ __forceinline__ u32 even_bits(u64 m)
{
    u64 x = m & 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return (u32)x;
}

// Compacts the set bits of the 1024-dword driver tile into `list` rounds of SP_CAP entries.
// Entry = position in tile | (flag << 16); flag = bit of sFlag32 at (flag_off + position).
// Returns the total number of set bits (workgroup-uniform).  Thread-private scan state stays in `w`/`idx0`.
struct CompactState {
    uint4 w;
    u32 idx0;     // exclusive prefix of this thread
    u32 total;
};

__device__ __forceinline__ CompactState compact_scan(const u32 *sDrv32, u32 *sTot, u32 tid)
{
    CompactState st;
    st.w = reinterpret_cast<const uint4 *>(sDrv32)[tid];
    const u32 n = __popc(st.w.x) + __popc(st.w.y) + __popc(st.w.z) + __popc(st.w.w);
    const u32 lane = tid & 63, wave = tid >> 6;
    u32 inc = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_up(inc, off, 64);
        if (lane >= (u32)off) inc += v;
    }
    __syncthreads();   // previous readers of sTot are done
    if (lane == 63) sTot[wave] = inc;
    __syncthreads();
    u32 base = 0, total = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        const u32 t = sTot[k];
        if (k < wave) base += t;
        total += t;
    }
    st.idx0 = base + inc - n;
    st.total = total;
    return st;
}

template <bool WITH_FLAG>
__device__ __forceinline__ void compact_emit(const CompactState &st, u32 round_lo, u32 *list, const u32 *sFlag32,
                                             u32 flag_off, u32 tid, u32 pad_to)
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
                u32 e = pos;
                if (WITH_FLAG) {
                    const u32 fa = flag_off + pos;
                    e |= ((sFlag32[fa >> 5] >> (fa & 31)) & 1u) << 16;
                }
                list[id] = e;
            }
            id++;
        }
    }
    const u32 left = st.total - round_lo;
    const u32 nround = left < SP_CAP ? left : SP_CAP;
    const u32 npad = (nround + pad_to - 1) / pad_to * pad_to;
    if (nround + tid < npad) list[nround + tid] = SP_INVALID;   // pad_to <= 256
}

#define SP_SLAB_ROWS 5   // per-workgroup partial rows of 1024 u32: ncc, fsum, ccbins(mscc), rsum, scalars

// Counters -> this workgroup's PRIVATE slab rows (plain stores; a reduce kernel sums the slabs).  Thousands of
// workgroups adding into the same 4 x (S+1) global words with atomics serialise on those addresses; private
// slabs + one reduction pass do not.
template <bool HAS_M, bool DO_NCC>
__device__ __forceinline__ void flush_all(Planes &cN, Planes &cF, Planes &cC, Planes &cR, u32 &qF, u32 &qR, u32 *acc,
                                          u32 l, u32 tid, u32 *__restrict__ slab, bool &first)
{
    __syncthreads();
    for (u32 i = tid; i < 4096; i += 256) acc[i] = 0;
    __syncthreads();
    if (DO_NCC) planes_flush(cN, qF, acc, l);
    if (HAS_M) {
        planes_flush(cF, qF, acc + 1024, l);
        planes_flush(cC, qF, acc + 2048, l);
        planes_flush(cR, qR, acc + 3072, l);
    }
    qF = 0;
    qR = 0;
    __syncthreads();
    for (u32 d = tid; d < 1024; d += 256) {
        const u32 a = (d & 31) * 32 + (d >> 5);
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            const u32 v = acc[q * 1024 + a];
            slab[q * 1024 + d] = first ? v : slab[q * 1024 + d] + v;
        }
    }
    first = false;
    __syncthreads();
}

template <bool HAS_M, bool DO_NCC>
__global__ void __launch_bounds__(256)
k_cc_sparse(const u64 *__restrict__ F, const u64 *__restrict__ R, const u64 *__restrict__ M, uint64_t nbits,
            uint64_t nwords, u32 max_shift, int32_t c, u32 G, u32 halo_m, u32 rlen64, u32 mlen64, u32 ntiles,
            u32 *__restrict__ slab_all)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u32 *slab = slab_all + (size_t)blockIdx.x * SP_SLAB_ROWS * 1024;
    bool first_flush = true;

    u64 *sF64 = reinterpret_cast<u64 *>(smem_raw);
    u64 *sR64 = sF64 + SP_TBW64;
    u64 *sM64 = sR64 + rlen64;
    u32 *sME32 = reinterpret_cast<u32 *>(sM64 + mlen64);
    u32 *sMO32 = sME32 + mlen64;
    u32 *list = sMO32 + mlen64;
    u32 *sTot = list + SP_CAP + 64;
    const u32 *sF32 = reinterpret_cast<const u32 *>(sF64);
    const u32 *sR32 = reinterpret_cast<const u32 *>(sR64);
    const u32 *sM32 = reinterpret_cast<const u32 *>(sM64);
    u32 *acc = reinterpret_cast<u32 *>(smem_raw);   // 4 x 1024 u32, aliases the tile buffers between tiles

    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63, wave = tid >> 6;
    const u32 slots_per_wave = 64 / G;
    const u32 total_slots = 4 * slots_per_wave;
    const u32 slot = wave * slots_per_wave + lane / G;
    const u32 l = lane % G;
    const u32 quad_span = total_slots * 4;                 // list entries consumed per quad step
    const u32 maxq_tile = SP_TB / quad_span;               // quads per slot if every bit of a tile is set

    u64 totF = 0, totR = 0;   // set bits seen (workgroup-uniform); NCC-only mode counts R per thread instead
    u32 cntR_thread = 0;
    Planes cN, cF, cC, cR;
    planes_zero(cN);
    planes_zero(cF);
    planes_zero(cC);
    planes_zero(cR);
    u32 qF = 0, qR = 0;   // quads added since the last flush (workgroup-uniform)


    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (qF + maxq_tile + 1 > SP_QLIMIT || qR + maxq_tile + 1 > SP_QLIMIT) flush_all<HAS_M, DO_NCC>(cN, cF, cC, cR, qF, qR, acc, l, tid, slab, first_flush);

        // ---- stage the tile ----------------------------------------------------------------------
        const int64_t w0 = (int64_t)tile * SP_TBW64;
        __syncthreads();
        for (u32 i = tid; i < SP_TBW64; i += 256) sF64[i] = sp_ld_word(F, w0 + i, nwords, nbits);
        for (u32 i = tid; i < rlen64; i += 256) sR64[i] = sp_ld_word(R, w0 + i, nwords, nbits);
        if (HAS_M) {
            const int64_t m0 = w0 - (int64_t)(halo_m / 64);
            for (u32 i = tid; i < mlen64; i += 256) {
                const u64 m = sp_ld_word(M, m0 + i, nwords, nbits);
                sM64[i] = m;
                sME32[i] = even_bits(m);
                sMO32[i] = even_bits(m >> 1);
            }
        }
        __syncthreads();

        // ---- forward reads drive: ncc, mscc.fsum, mscc.ccbins ----------------------------------------
        {
            const CompactState st = compact_scan(sF32, sTot, tid);
            totF += st.total;
            if (!HAS_M) {
                const uint4 r = reinterpret_cast<const uint4 *>(sR32)[tid];
                cntR_thread += __popc(r.x) + __popc(r.y) + __popc(r.z) + __popc(r.w);
            }
            for (u32 round_lo = 0; round_lo < st.total; round_lo += SP_CAP) {
                compact_emit<HAS_M>(st, round_lo, list, sM32, halo_m, tid, quad_span);
                __syncthreads();
                const u32 left = st.total - round_lo;
                const u32 nround = left < SP_CAP ? left : SP_CAP;
                const u32 nq = (nround + quad_span - 1) / quad_span;
                for (u32 q = 0; q < nq; q++) {
                    u32 wN[4], wF[4], wC[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const u32 e = list[(q * 4 + k) * total_slots + slot];
                        const bool valid = (e & SP_INVALID) == 0;
                        const u32 x = e & 0xffffu;
                        const u32 ri = (x >> 5) + l;
                        u32 rw = __builtin_amdgcn_alignbit(sR32[ri + 1], sR32[ri], x & 31u);
                        rw = valid ? rw : 0u;
                        wN[k] = rw;
                        if (HAS_M) {
                            const u32 a = x + (u32)c + halo_m - 32u * l - 31u;
                            u32 mw = __builtin_amdgcn_alignbit(sM32[(a >> 5) + 1], sM32[a >> 5], a & 31u);
                            mw = __builtin_bitreverse32(mw);
                            mw = (valid && ((e >> 16) & 1u)) ? mw : 0u;
                            wF[k] = mw;
                            wC[k] = mw & rw;
                        }
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qF + q);
                    if (DO_NCC) add_quad(cN, wN[0], wN[1], wN[2], wN[3], qc);
                    if (HAS_M) {
                        add_quad(cF, wF[0], wF[1], wF[2], wF[3], qc);
                        add_quad(cC, wC[0], wC[1], wC[2], wC[3], qc);
                    }
                }
                qF += nq;
                __syncthreads();
            }
        }

        // ---- reverse reads drive: mscc.rsum -----------------------------------------------------------
        if (HAS_M) {
            const CompactState st = compact_scan(sR32, sTot, tid);
            totR += st.total;
            for (u32 round_lo = 0; round_lo < st.total; round_lo += SP_CAP) {
                compact_emit<false>(st, round_lo, list, nullptr, 0, tid, quad_span);
                __syncthreads();
                const u32 left = st.total - round_lo;
                const u32 nround = left < SP_CAP ? left : SP_CAP;
                const u32 nq = (nround + quad_span - 1) / quad_span;
                for (u32 q = 0; q < nq; q++) {
                    u32 wR[4];
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        const u32 e = list[(q * 4 + k) * total_slots + slot];
                        const bool valid = (e & SP_INVALID) == 0;
                        const u32 p = e & 0xffffu;
                        const u32 a1 = p + halo_m - 32u * l - 31u;
                        const u32 w1 = __builtin_amdgcn_alignbit(sM32[(a1 >> 5) + 1], sM32[a1 >> 5], a1 & 31u);
                        const u32 b = halo_m + p + (u32)c;
                        const u32 a2 = (b >> 1) - 32u * l - 31u;
                        const u32 *dec = sME32 + (b & 1u) * mlen64;     // even or odd decimated copy
                        const u32 w2 = __builtin_amdgcn_alignbit(dec[(a2 >> 5) + 1], dec[a2 >> 5], a2 & 31u);
                        const u32 w = __builtin_bitreverse32(w1 & w2);
                        wR[k] = valid ? w : 0u;
                    }
                    const u32 qc = __builtin_amdgcn_readfirstlane(qR + q);
                    add_quad(cR, wR[0], wR[1], wR[2], wR[3], qc);
                }
                qR += nq;
                __syncthreads();
            }
        }
    }
    flush_all<HAS_M, DO_NCC>(cN, cF, cC, cR, qF, qR, acc, l, tid, slab, first_flush);
    // popcount(F), popcount(R): bit_array_num_bits_set of mscc.pyx:236-237, for free from the compaction
    if (!HAS_M) {
        u32 v = cntR_thread;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) sTot[wave] = v;
        __syncthreads();
        if (tid == 0) totR = (u64)sTot[0] + sTot[1] + sTot[2] + sTot[3];
    }
    if (tid == 0) {
        slab[4 * 1024 + 0] = (u32)totF;
        slab[4 * 1024 + 1] = (u32)totR;
    }
}

// dst[r][i] = sum over workgroups of slab[wg][src_row[r]][i], i < n; blockIdx.y = r
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
__global__ void __launch_bounds__(256)
k_autocorr_edges(const u64 *__restrict__ M, uint64_t nbits, uint64_t nwords, u32 max_lag, u32 G, u32 wlen64,
                 u32 ntiles, u32 *__restrict__ slab_all)
{
    u32 *slab = slab_all + (size_t)blockIdx.x * 3 * 1024;   // rows: P, N, scalars
    bool first = true;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u64 *sU64 = reinterpret_cast<u64 *>(smem_raw);
    u64 *sD64 = sU64 + wlen64;
    u64 *sE64 = sD64 + wlen64;
    u32 *list = reinterpret_cast<u32 *>(sE64 + SP_TBW64);
    u32 *sTot = list + SP_CAP + 64;
    const u32 *sU32 = reinterpret_cast<const u32 *>(sU64);
    const u32 *sD32 = reinterpret_cast<const u32 *>(sD64);
    const u32 *sE32 = reinterpret_cast<const u32 *>(sE64);
    u32 *acc = reinterpret_cast<u32 *>(smem_raw);

    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63, wave = tid >> 6;
    const u32 slots_per_wave = 64 / G;
    const u32 total_slots = 4 * slots_per_wave;
    const u32 slot = wave * slots_per_wave + lane / G;
    const u32 l = lane % G;
    const u32 quad_span = total_slots * 4;
    const u32 maxq_tile = SP_TB / quad_span;

    Planes cP, cN;
    planes_zero(cP);
    planes_zero(cN);
    u32 qc = 0;
    u32 cntM = 0, cntU = 0;

    auto flush = [&]() {
        __syncthreads();
        for (u32 i = tid; i < 2048; i += 256) acc[i] = 0;
        __syncthreads();
        planes_flush(cP, qc, acc, l);
        planes_flush(cN, qc, acc + 1024, l);
        qc = 0;
        __syncthreads();
        for (u32 k = tid; k < 1024; k += 256) {
            const u32 a = (k & 31) * 32 + (k >> 5);
            const u32 vp = acc[a], vn = acc[1024 + a];
            slab[k] = first ? vp : slab[k] + vp;
            slab[1024 + k] = first ? vn : slab[1024 + k] + vn;
        }
        first = false;
        __syncthreads();
    };

    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (qc + maxq_tile + 1 > SP_QLIMIT) flush();
        const int64_t w0 = (int64_t)tile * SP_TBW64;
        __syncthreads();
        for (u32 i = tid; i < wlen64; i += 256) {
            const u64 m = sp_ld_word(M, w0 + i, nwords, nbits);
            const u64 mprev = sp_ld_word(M, w0 + i - 1, nwords, nbits);
            const u64 sh = (m << 1) | (mprev >> 63);     // M[j-1] aligned with M[j]
            const u64 U = m & ~sh, D = ~m & sh;
            sU64[i] = U;
            sD64[i] = D;
            if (i < SP_TBW64) {
                sE64[i] = U | D;
                cntM += __popcll(m);
                cntU += __popcll(U);
            }
        }
        __syncthreads();
        const CompactState st = compact_scan(sE32, sTot, tid);
        for (u32 round_lo = 0; round_lo < st.total; round_lo += SP_CAP) {
            compact_emit<true>(st, round_lo, list, sD32, 0, tid, quad_span);
            __syncthreads();
            const u32 left = st.total - round_lo;
            const u32 nround = left < SP_CAP ? left : SP_CAP;
            const u32 nq = (nround + quad_span - 1) / quad_span;
            for (u32 q = 0; q < nq; q++) {
                u32 wp[4], wn[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    const u32 e = list[(q * 4 + k) * total_slots + slot];
                    const bool valid = (e & SP_INVALID) == 0;
                    const bool falling = (e >> 16) & 1u;
                    const u32 x = e & 0xffffu;
                    const u32 wi = (x >> 5) + l;
                    const u32 wu = __builtin_amdgcn_alignbit(sU32[wi + 1], sU32[wi], x & 31u);
                    const u32 wd = __builtin_amdgcn_alignbit(sD32[wi + 1], sD32[wi], x & 31u);
                    wp[k] = valid ? (falling ? wd : wu) : 0u;   // same-sign pairs: U*U, D*D
                    wn[k] = valid ? (falling ? wu : wd) : 0u;   // opposite-sign pairs: U*D, D*U
                }
                const u32 qq = __builtin_amdgcn_readfirstlane(qc + q);
                add_quad(cP, wp[0], wp[1], wp[2], wp[3], qq);
                add_quad(cN, wn[0], wn[1], wn[2], wn[3], qq);
            }
            qc += nq;
            __syncthreads();
        }
    }
    flush();
    for (int off = 32; off > 0; off >>= 1) {
        cntM += __shfl_down(cntM, off, 64);
        cntU += __shfl_down(cntU, off, 64);
    }
    if (lane == 0) {
        sTot[wave] = cntM;
        sTot[4 + wave] = cntU;
    }
    __syncthreads();
    if (tid == 0) {
        slab[2048 + 0] = sTot[0] + sTot[1] + sTot[2] + sTot[3];
        slab[2048 + 1] = sTot[4] + sTot[5] + sTot[6] + sTot[7];
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

static uint32_t sparse_grid(pmx_ctx *ctx, uint32_t lds_bytes, uint64_t ntiles, uint32_t wg_per_cu)
{
    uint32_t per_cu = (160u * 1024u) / lds_bytes;
    if (per_cu > wg_per_cu) per_cu = wg_per_cu;
    if (per_cu < 1) per_cu = 1;
    uint64_t gx = (uint64_t)ctx->num_cus * per_cu;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    return (uint32_t)gx;
}

// d_tmp: 2 * 1024 + 16 u64 of scratch (P, N, scalars).
int pmx_launch_autocorr_edges(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_lag, u64 *d_tmp,
                              uint32_t mode, uint32_t read_len, uint32_t max_shift, u64 *d_out, u64 *d_popcount_out)
{
    const uint64_t nwords = (nbits + 63) / 64;
    u32 need = (max_lag + 1 + 31) / 32, G = 1;
    while (G < need) G <<= 1;
    if (G < 4) G = 4;
    const u32 wlen64 = (SP_TB + 32 * G + 64 + 63) / 64;
    u32 lds = wlen64 * 16 + SP_TBW64 * 8 + (SP_CAP + 64) * 4 + 64;
    if (lds < 2 * 1024 * 4 + 64) lds = 2 * 1024 * 4 + 64;
    const uint64_t ntiles = (nbits + 1 + SP_TB - 1) / SP_TB;     // E lives on [0, nbits]
    const uint32_t gx = sparse_grid(ctx, lds, ntiles, 4);
    int rc = pmx_ensure_slab(ctx, (size_t)gx * 3 * 1024);
    if (rc) return rc;
    u64 *P = d_tmp, *N = d_tmp + 1024, *scal = d_tmp + 2048;
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_AUTOCORR, &tl);
    if (rc) return rc;
    hipLaunchKernelGGL(k_autocorr_edges, dim3(gx), dim3(256), lds, ctx->stream, (const u64 *)d_M, nbits, nwords,
                       max_lag, G, wlen64, (u32)ntiles, ctx->d_slab);
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
    return max_shift >= 3 && max_shift <= 1023 && read_len >= 1 && read_len <= 4096;
}

int pmx_launch_cc_sparse(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M,
                         uint64_t nbits, uint32_t max_shift, uint32_t read_len, bool do_ncc, u64 *d_out,
                         uint32_t out_stride)
{
    if (!d_M && !do_ncc) return PMX_OK;
    const uint64_t nwords = (nbits + 63) / 64;
    const SparseGeom g = sparse_geom(max_shift, read_len, d_M != nullptr);
    uint64_t ntiles = (nbits + SP_TB - 1) / SP_TB;
    if (ntiles < 1) ntiles = 1;
    const int32_t c = (int32_t)read_len - 1;
    const uint32_t gx = sparse_grid(ctx, g.lds_bytes, ntiles, d_M ? 2 : 4);
    int rc = pmx_ensure_slab(ctx, (size_t)gx * SP_SLAB_ROWS * 1024);
    if (rc) return rc;
    const dim3 grid(gx), block(256);
    const u64 *F = (const u64 *)d_F, *R = (const u64 *)d_R, *M = (const u64 *)d_M;
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_SPARSE, &tl);
    if (rc) return rc;
#define SP_LAUNCH(HM, NC)                                                                                    \
    hipLaunchKernelGGL((k_cc_sparse<HM, NC>), grid, block, g.lds_bytes, ctx->stream, F, R, M, nbits, nwords, \
                       max_shift, c, g.G, g.halo_m, g.rlen64, g.mlen64, (u32)ntiles, ctx->d_slab)
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
