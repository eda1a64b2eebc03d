// Device-side BAM ingest for gfx950 (include/pymasc_amd_ingest.h).
//
// What the reference does one read at a time through pysam (PyMaSC/handler/calc.py:140-153, handler/read.py:62-155)
// and libpymasc_io.so does with zlib on host threads happens here on the GPU for the whole file:
//
//   k_bgzf_inflate  one wavefront (= one workgroup) per BGZF member.  DEFLATE is a serial bit stream, so the wavefront
//                   decodes UNIFORMLY -- the bit buffer, the Huffman lookups and the output cursor live in scalar
//                   registers; the 64 lanes hold 256 bytes of compressed input as one dword each (v_readlane feeds the
//                   bit buffer), build the Huffman tables together, and copy matches 64 bytes per step.  Tables and the
//                   last 4 KB of output are in LDS (7.9 KB per wavefront: 20 wavefronts per CU); a match that reaches
//                   further back reads the member's own output from HBM (written 256 bytes at a time, in order, by
//                   this wavefront).  Members are independent, so a file is ~15 members per MB of wavefronts.
//   k_bgzf_crc      CRC-32 of a member's output: 64 slices, one per lane, combined with x^(8n) mod P.
//   k_bam_spec / k_bam_walk / k_bam_scan   the record chain and the filter, see the comment above k_bam_spec.
//
// No zlib, no host inflate: a host that cannot launch these kernels gets an error, not a fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../../include/pymasc_amd_ingest.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define RFL(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))

// ---------------------------------------------------------------------------------------------------------------
// BGZF members
struct DMember {
    u64 in_off;    // byte offset of the raw DEFLATE stream in the device copy of the file
    u64 out_off;   // byte offset of the member's output in the inflated stream
    u32 clen, isize, crc;
    u32 open_size;   // 1: `isize` is only a bound (a zlib stream of unknown length: BigWig data blocks); the kernel reports the length
};

// status word of a member (0 = fine)
enum {
    INF_OK = 0,
    INF_ERR_BTYPE = 1,      // reserved block type
    INF_ERR_STORED = 2,     // LEN / NLEN of a stored block
    INF_ERR_TABLE = 3,      // over-subscribed or malformed Huffman code lengths
    INF_ERR_CODE = 4,       // a bit pattern that is no code of the block's tables
    INF_ERR_DIST = 5,       // distance beyond the start of the member / too-large symbol
    INF_ERR_OUTPUT = 6,     // more output than ISIZE
    INF_ERR_INPUT = 7,      // the stream runs past the member's compressed bytes
    INF_ERR_ISIZE = 8,      // less output than ISIZE
    INF_ERR_CRC = 9,
};

#define INF_RING 4096u                      // bytes of recent output kept in LDS
#define INF_NEAR (INF_RING - 258u - 64u)    // distances up to here are served from the ring
#define INF_LP 11u                          // primary bits of the literal / length table (level-1 BAM: the 256 values of a packed
                                            // sequence byte get 10- and 11-bit codes; at 10 bits 8 % of the symbols took the slow path)
#define INF_DP 8u                           // ... of the distance table (and of the code-length code: 7 used)

struct InfLds {
    u8 ring[INF_RING];
    u8 dump[64];                            // where the lanes beyond the end of a match write (no branch around the store)
    u16 litT[1u << INF_LP];                 // a literal: byte | code length << 8.  Anything else has bit 15 set: a length / end-of-block
                                            // symbol as (symbol - 256) | code length << 8, or 0x8000 = no code of <= INF_LP bits
    u16 distT[1u << INF_DP];                // entry: symbol | code length << 9; 0 = not a code of <= INF_DP bits
    u16 symL[320];                          // symbols in canonical order (the slow path for longer codes)
    u16 symD[32];
    u32 cntL[16], cntD[16];                 // codes per length
    u32 fcL[16], fcD[16];                   // first code of every length
    u32 ixL[16], ixD[16];                   // ... and its index in the canonical order
    u8 lens[352];                           // code lengths of the block (literal / length, then distance); [351]: dump
    u8 cl[32];                              // lengths of the code-length code
};

// The bit reader: `cur` / `nxt` are per-lane dwords of the compressed stream (lane i: dword i of a 256-byte piece),
// everything else is uniform.  Bits are consumed LSB first (RFC 1951 3.1.1).
struct BitRd {
    const u32 *base;
    u32 cur, nxt, lane;
    u32 ci, widx;   // piece, next dword of it
    u64 bb;
    u32 bc;
};
__device__ __forceinline__ void br_adv(BitRd &r)
{
    if (++r.widx == 64u) {
        r.widx = 0;
        r.ci++;
        r.cur = r.nxt;
        r.nxt = r.base[(size_t)(r.ci + 1u) * 64u + r.lane];
    }
}
__device__ __forceinline__ void br_seek(BitRd &r, u32 bytepos)
{
    r.ci = bytepos >> 8;
    r.widx = (bytepos & 255u) >> 2;
    r.cur = r.base[(size_t)r.ci * 64u + r.lane];
    r.nxt = r.base[(size_t)(r.ci + 1u) * 64u + r.lane];
    const u32 w = (u32)__builtin_amdgcn_readlane((int)r.cur, (int)r.widx);
    const u32 sh = 8u * (bytepos & 3u);
    r.bb = (u64)(w >> sh);
    r.bc = 32u - sh;
    br_adv(r);
}
__device__ __forceinline__ void br_refill(BitRd &r)   // afterwards at least 32 bits are buffered
{
    if (r.bc < 32u) {
        const u32 w = (u32)__builtin_amdgcn_readlane((int)r.cur, (int)r.widx);
        r.bb |= (u64)w << r.bc;
        r.bc += 32u;
        br_adv(r);
    }
}
__device__ __forceinline__ u32 br_take(BitRd &r, u32 n)
{
    const u32 v = (u32)r.bb & ((1u << n) - 1u);
    r.bb >>= n;
    r.bc -= n;
    return v;
}

// 256 bytes of the ring -> the member's output.  Positions are relative to `ob`, the member's first output byte rounded down to
// 256.  The first and the last group of a member are stored byte by byte under a per-lane test -- in a function of its own:
// ONE divergent branch anywhere inside the symbol loop makes the compiler structurize the whole loop (every uniform branch of
// the decoder becomes exec-mask and flag handling, 52 scalar instructions per symbol instead of ~25).
__device__ __noinline__ void inf_flush_edge(const u8 *ring, u8 *ob, u32 g0, u32 pstart, u32 pend, u32 lane)
{
    const u32 g = g0 + 4u * lane;
    const u32 v = *reinterpret_cast<const u32 *>(ring + (g & (INF_RING - 1u)));
    for (u32 b = 0; b < 4; b++)
        if (g + b >= pstart && g + b < pend) ob[g + b] = (u8)(v >> (8u * b));
}
__device__ __forceinline__ void inf_flush(const u8 *ring, u8 *__restrict__ ob, u32 g0, u32 pstart, u32 pend, u32 lane)
{
    __syncthreads();
    if (g0 >= pstart && g0 + 256u <= pend) {   // (uniform)
        const u32 g = g0 + 4u * lane;
        *reinterpret_cast<u32 *>(ob + g) = *reinterpret_cast<const u32 *>(ring + (g & (INF_RING - 1u)));
    } else {
        inf_flush_edge(ring, ob, g0, pstart, pend, lane);
    }
}

// Canonical Huffman tables from code lengths (RFC 1951 3.2.2), built by the whole wavefront: counts per length with LDS
// atomics, then every lane places its symbols -- the rank of a symbol among those of its length is a ballot + popcount.
template <u32 P, bool LIT>
__device__ __forceinline__ bool inf_build(const u8 *lens, u32 n, u16 *T, u32 *cnt, u32 *fc, u32 *ix, u16 *syms, u32 lane)
{
    for (u32 i = lane; i < (1u << P); i += 64u) T[i] = LIT ? 0x8000 : 0;
    if (lane < 16u) cnt[lane] = 0;
    __syncthreads();
    for (u32 s = lane; s < n; s += 64u) {
        const u32 l = lens[s];
        if (l) atomicAdd(&cnt[l], 1u);
    }
    __syncthreads();
    u32 run[16], nc[16];
    u32 code = 0, idx = 0;
    int left = 1;
    bool over = false;
#pragma unroll
    for (u32 L = 1; L <= 15u; L++) {
        const u32 c = RFL(cnt[L]);
        nc[L] = code;
        run[L] = idx;
        fc[L] = code;   // (every lane the same word)
        ix[L] = idx;
        code = (code + c) << 1;
        idx += c;
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
    }
    if (over) return false;
    const u64 below = (1ull << lane) - 1ull;
    for (u32 b = 0; b < n; b += 64u) {
        const u32 s = b + lane;
        const u32 l = s < n ? lens[s] : 0u;
#pragma unroll
        for (u32 L = 1; L <= 15u; L++) {
            const u64 m = __ballot(l == L);
            if (m) {
                if (l == L) {
                    const u32 rk = (u32)__popcll(m & below);
                    syms[run[L] + rk] = (u16)s;
                    if (L <= P) {
                        const u32 rev = __brev(nc[L] + rk) >> (32u - L);
                        const u32 ent = !LIT ? (s | (L << 9)) : s < 256u ? (s | (L << 8)) : (0x8000u | (s - 256u) | (L << 8));
                        for (u32 k = rev; k < (1u << P); k += (1u << L)) T[k] = (u16)ent;
                    }
                }
                const u32 pc = (u32)__popcll(m);
                run[L] += pc;
                nc[L] += pc;
            }
        }
    }
    __syncthreads();
    return true;
}

// a code longer than the primary table: its first L bits, as a number, lie in [first code of length L, + codes of length L)
// for its own length L and above that range for every shorter one (canonical codes)
template <u32 P>
__device__ __forceinline__ bool inf_slow(u64 bb, const u32 *cnt, const u32 *fc, const u32 *ix, const u16 *syms, u32 &sym, u32 &len)
{
    const u32 rb = __brev((u32)bb);
    for (u32 L = P + 1u; L <= 15u; L++) {
        const u32 off = (rb >> (32u - L)) - RFL(fc[L]);
        if (off < RFL(cnt[L])) {
            sym = RFL(syms[RFL(ix[L]) + off]);
            len = L;
            return true;
        }
    }
    return false;
}

__constant__ u8 INF_CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// (every lane stores the same word: an `if (lane == 0)` in front of a return makes the compiler treat everything that
// reaches the function's single exit as divergent, and the whole decoder state moves to vector registers)
#define INF_FAIL(code)        \
    {                         \
        status[m] = (code);   \
        return;               \
    }

// The decoder is bound by SCALAR instruction issue (one per cycle and CU, shared by every resident wavefront), not by
// latency: what counts is the number of scalar instructions per output byte.  Hence: no divergent branch in the symbol loop
// (lanes beyond a match's end write to a dump slot; literal bytes are written by all lanes to one address), one exit from it
// (errors leave through `err`), the input-limit test only where the reader takes a dword, no output-limit test per literal
// (the ring takes the byte anyway, a flush stores nothing beyond ISIZE, and the count is checked at every flush and at the
// end), 32-bit positions relative to the member's 256-byte aligned start.
__global__ void __launch_bounds__(64)
k_bgzf_inflate(const u8 *__restrict__ in, u8 *__restrict__ out, const DMember *__restrict__ mem, u32 nmem, u32 *__restrict__ status,
               u32 *__restrict__ out_size)
{
    __shared__ __align__(16) InfLds S;
    const u32 m = blockIdx.x;
    if (m >= nmem) return;
    const u32 lane = threadIdx.x;
    const u64 in_off = mem[m].in_off;
    const u64 gstart = mem[m].out_off;
    const u32 clen = mem[m].clen, isize = mem[m].isize;
    u8 *__restrict__ ob = out + (gstart & ~255ull);
    const u32 pstart = (u32)(gstart & 255ull), pend = pstart + isize;   // positions relative to ob
    u32 pos = pstart;   // (what lies below pos & ~255 has been stored)
    BitRd r;
    r.base = reinterpret_cast<const u32 *>(in + (in_off & ~3ull));
    r.lane = lane;
    const u32 skew = (u32)(in_off & 3ull);
    br_seek(r, skew);
    const u32 wlimit = (skew + clen + 3u) / 4u + 2u;   // dwords the reader may have taken (it looks ahead up to 8 bytes)
    constexpr u32 M = INF_RING - 1u;
#define INF_REFILL()                                              \
    if (r.bc < 32u) {                                             \
        br_refill(r);                                             \
        if (r.ci * 64u + r.widx > wlimit) INF_FAIL(INF_ERR_INPUT) \
    }

    for (;;) {
        INF_REFILL()
        const u32 bfinal = br_take(r, 1), btype = br_take(r, 2);
        if (btype == 3u) INF_FAIL(INF_ERR_BTYPE)
        if (btype == 0u) {
            // stored block: to the next byte boundary, LEN, NLEN, LEN bytes
            br_take(r, r.bc & 7u);
            INF_REFILL()
            const u32 LEN = (u32)r.bb & 0xffffu, NLEN = (u32)(r.bb >> 16) & 0xffffu;
            r.bb >>= 32;
            r.bc -= 32u;
            if ((LEN ^ NLEN) != 0xffffu) INF_FAIL(INF_ERR_STORED)
            const u32 bp = (r.ci * 64u + r.widx) * 4u - (r.bc >> 3);   // the next unread byte (relative to r.base)
            if (bp + LEN > skew + clen) INF_FAIL(INF_ERR_INPUT)
            if (pos + LEN > pend) INF_FAIL(INF_ERR_OUTPUT)
            const u8 *ib = reinterpret_cast<const u8 *>(r.base);
            for (u32 k = 0; k < LEN; k += 64u) {
                const u32 n = LEN - k < 64u ? LEN - k : 64u;
                if (lane < n) S.ring[(pos + lane) & M] = ib[bp + k + lane];
                const u32 g = pos & ~255u;
                pos += n;
                if ((pos & ~255u) != g) inf_flush(S.ring, ob, g, pstart, pend, lane);
            }
            __syncthreads();
            br_seek(r, bp + LEN);
        } else {
            u32 hlit = 288u, hdist = 30u;
            if (btype == 1u) {
                for (u32 s = lane; s < 288u; s += 64u) S.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8;
                if (lane < 30u) S.lens[288u + lane] = 5;
                __syncthreads();
            } else {
                INF_REFILL()
                hlit = br_take(r, 5) + 257u;
                hdist = br_take(r, 5) + 1u;
                const u32 hclen = br_take(r, 4) + 4u;
                if (hlit > 286u || hdist > 30u) INF_FAIL(INF_ERR_TABLE)
                if (lane < 19u) S.cl[lane] = 0;
                __syncthreads();
                for (u32 i = 0; i < hclen; i++) {
                    INF_REFILL()
                    S.cl[INF_CL_ORDER[i]] = (u8)br_take(r, 3);
                }
                __syncthreads();
                if (!inf_build<7, false>(S.cl, 19u, S.distT, S.cntD, S.fcD, S.ixD, S.symD, lane)) INF_FAIL(INF_ERR_TABLE)
                const u32 total = hlit + hdist;
                u32 i = 0, prev = 0;
                while (i < total) {
                    INF_REFILL()
                    const u32 e = RFL(S.distT[(u32)r.bb & 127u]);
                    if (!e) INF_FAIL(INF_ERR_CODE)
                    const u32 s = e & 511u;
                    br_take(r, e >> 9);
                    if (s < 16u) {
                        S.lens[i] = (u8)s;
                        prev = s;
                        i++;
                    } else {
                        u32 rep, val = 0;
                        if (s == 16u) {
                            if (i == 0) INF_FAIL(INF_ERR_TABLE)
                            rep = 3u + br_take(r, 2);
                            val = prev;
                        } else if (s == 17u) {
                            rep = 3u + br_take(r, 3);
                        } else {
                            rep = 11u + br_take(r, 7);
                        }
                        if (i + rep > total) INF_FAIL(INF_ERR_TABLE)
                        for (u32 k = 0; k < rep; k += 64u) S.lens[k + lane < rep ? i + k + lane : 351u] = (u8)val;
                        i += rep;
                        prev = val;
                    }
                }
                __syncthreads();
                if (RFL(S.lens[256]) == 0u) INF_FAIL(INF_ERR_TABLE)   // no end-of-block code
            }
            if (!inf_build<INF_LP, true>(S.lens, hlit, S.litT, S.cntL, S.fcL, S.ixL, S.symL, lane)) INF_FAIL(INF_ERR_TABLE)
            if (!inf_build<INF_DP, false>(S.lens + hlit, hdist, S.distT, S.cntD, S.fcD, S.ixD, S.symD, lane)) INF_FAIL(INF_ERR_TABLE)

            u32 err = 0;
            // (the ring addresses are formed on the vector unit, from a copy of pos the compiler takes for a per-lane value)
            u32 vpos = pos;
            asm volatile("" : "+v"(vpos));
            for (;;) {
                if (r.bc < 32u) {
                    br_refill(r);
                    if (r.ci * 64u + r.widx > wlimit) {
                        err = INF_ERR_INPUT;
                        break;
                    }
                }
                // A run of literals, as tight as the compiler makes it: every instruction of a wavefront's stream costs the same,
                // whatever its type, and literals are most of a BAM file's symbols.  The byte goes to the ring straight from the
                // register the lookup filled; the run ends at a symbol that is not a plain literal or when fewer than 15 bits are
                // buffered (the lookups never see more than 15) -- at most 48 literals, so the 256-byte groups the run completed
                // are stored behind it (the ring holds 4096 bytes).
                const u32 g0 = pos & ~255u;
                u32 e;
#ifndef INF_NO_ASM_LITERALS
                {
                    // ... written out: the compiler's version of this loop is 23 instructions per literal (it keeps the two exit
                    // conditions as lane masks and forms the LDS address on the scalar unit); here the bit buffer lives in vector
                    // registers for the length of the run (every lane holds the same words), so the table index, the 64-bit shift
                    // (v_alignbit_b32) and the count are vector instructions and nothing goes through the scalar unit: 15 per literal.
                    // All LDS traffic of the block has landed when it ends (s_waitcnt), the compiler's own counting is not disturbed.
                    u32 lo = (u32)r.bb, hi = (u32)(r.bb >> 32), bc = r.bc, ev, a, l;
                    const u32 vlit = (u32)(uintptr_t)S.litT, vring = (u32)(uintptr_t)S.ring;   // (low halves of the flat addresses: LDS offsets)
                    asm volatile("1:\n\t"
                                 "v_and_b32 %[a], 0x7ff, %[lo]\n\t"
                                 "v_lshl_add_u32 %[a], %[a], 1, %[vlit]\n\t"
                                 "ds_read_i16 %[ev], %[a]\n\t"
                                 "s_waitcnt lgkmcnt(0)\n\t"
                                 "v_cmp_gt_i32 vcc, 0, %[ev]\n\t"
                                 "s_cbranch_vccnz 2f\n\t"
                                 "v_and_b32 %[a], 0xfff, %[vpos]\n\t"
                                 "v_add_u32 %[a], %[a], %[vring]\n\t"
                                 "ds_write_b8 %[a], %[ev]\n\t"
                                 "v_add_u32 %[vpos], 1, %[vpos]\n\t"
                                 "v_lshrrev_b32 %[l], 8, %[ev]\n\t"
                                 "v_alignbit_b32 %[lo], %[hi], %[lo], %[l]\n\t"
                                 "v_lshrrev_b32 %[hi], %[l], %[hi]\n\t"
                                 "v_sub_u32 %[bc], %[bc], %[l]\n\t"
                                 "v_cmp_lt_u32 vcc, 14, %[bc]\n\t"
                                 "s_cbranch_vccnz 1b\n"
                                 "2:\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : [lo] "+v"(lo), [hi] "+v"(hi), [bc] "+v"(bc), [vpos] "+v"(vpos), [ev] "=&v"(ev), [a] "=&v"(a), [l] "=&v"(l)
                                 : [vlit] "v"(vlit), [vring] "v"(vring)
                                 : "vcc", "memory");
                    static_assert(INF_LP == 11u && INF_RING == 4096u, "the masks of the literal loop");
                    r.bb = ((u64)RFL(hi) << 32) | RFL(lo);
                    r.bc = RFL(bc);
                    e = RFL(ev) & 0xffffu;   // (the entry that ended the run, or the last literal's: bit 15 tells)
                    pos = RFL(vpos);
                }
#else
                for (;;) {
                    const u32 ev = S.litT[(u32)r.bb & ((1u << INF_LP) - 1u)];   // (the same word in every lane)
                    e = RFL(ev);
                    if (e & 0x8000u) break;
                    S.ring[vpos & M] = (u8)ev;
                    vpos++;
                    pos++;
                    const u32 l = e >> 8;
                    r.bb >>= l;
                    r.bc -= l;
                    if (r.bc < 15u) break;
                }
#endif
                if ((pos & ~255u) != g0) {   // (one group at most)
                    if (g0 >= pend) {
                        err = INF_ERR_OUTPUT;
                        break;
                    }
                    inf_flush(S.ring, ob, g0, pstart, pend, lane);
                }
                if (!(e & 0x8000u)) continue;   // out of bits: the next round refills
                u32 sym = 256u + (e & 31u), l = (e >> 8) & 15u;
                if (!l && !inf_slow<INF_LP>(r.bb, S.cntL, S.fcL, S.ixL, S.symL, sym, l)) {
                    err = INF_ERR_CODE;
                    break;
                }
                r.bb >>= l;
                r.bc -= l;
                if (r.bc < 32u) br_refill(r);   // (the input limit is tested at the next symbol)
                if (sym < 256u) {               // a literal whose code is longer than the table's index
                    S.ring[vpos & M] = (u8)sym;
                    vpos++;
                    pos++;
                    if ((pos & 255u) == 0u) {
                        if (pos - 256u >= pend) {
                            err = INF_ERR_OUTPUT;
                            break;
                        }
                        inf_flush(S.ring, ob, pos - 256u, pstart, pend, lane);
                    }
                } else if (sym == 256u) {
                    break;
                } else {
                    sym -= 257u;
                    u32 len;
                    if (sym < 8u) {
                        len = 3u + sym;
                    } else if (sym >= 28u) {
                        len = 258u;
                        if (sym > 28u) {
                            err = INF_ERR_DIST;
                            break;
                        }
                    } else {
                        const u32 eb = (sym >> 2) - 1u;
                        len = 3u + ((4u + (sym & 3u)) << eb) + br_take(r, eb);
                    }
                    if (r.bc < 32u) br_refill(r);
                    const u32 de = RFL(S.distT[(u32)r.bb & ((1u << INF_DP) - 1u)]);
                    u32 ds = de & 511u, dl = de >> 9;
                    if (!de && !inf_slow<INF_DP>(r.bb, S.cntD, S.fcD, S.ixD, S.symD, ds, dl)) {
                        err = INF_ERR_CODE;
                        break;
                    }
                    r.bb >>= dl;
                    r.bc -= dl;
                    u32 dist;
                    if (ds < 4u) {
                        dist = 1u + ds;
                    } else {
                        const u32 eb = (ds >> 1) - 1u;
                        dist = 1u + ((2u + (ds & 1u)) << eb) + br_take(r, eb);
                    }
                    if (ds > 29u || dist > pos - pstart) {
                        err = INF_ERR_DIST;
                        break;
                    }
                    if (pos + len > pend) {
                        err = INF_ERR_OUTPUT;
                        break;
                    }
                    // the copy: lane i takes byte i of the match (a distance shorter than the match repeats with period dist);
                    // every source byte lies below pos, every destination at or above it
                    __syncthreads();
                    const bool wrap = dist < len, near = dist <= INF_NEAR;
                    // (further back than the ring: from the member's output in HBM, flushed up to the last 256-byte boundary by
                    // this wavefront's own, earlier stores; read past the first-level cache)
                    const volatile u8 *src = ob + (pos - dist);
                    for (u32 k = 0; k < len; k += 64u) {
                        const u32 i = k + lane;
                        const bool act = i < len;
                        u32 j = i;
                        if (wrap) j = i % dist;
                        u32 byte;
                        if (near) byte = S.ring[(pos - dist + j) & M];
                        else byte = src[act ? i : 0u];
                        S.ring[act ? ((pos + i) & M) : (INF_RING + lane)] = (u8)byte;
                    }
                    const u32 g = pos & ~255u;
                    pos += len;
                    vpos += len;
                    for (u32 gg = g; gg != (pos & ~255u); gg += 256u) inf_flush(S.ring, ob, gg, pstart, pend, lane);
                }
            }
            if (err) INF_FAIL(err)
        }
        if (bfinal) break;
    }
    // the stream must end inside the member's bytes and fill ISIZE exactly
    const u64 used_bits = (u64)(r.ci * 64u + r.widx) * 32u - r.bc - 8u * skew;
    if (used_bits > 8ull * clen) INF_FAIL(INF_ERR_INPUT)
    const bool open_size = mem[m].open_size != 0;
    if (pos > pend || (!open_size && pos != pend)) INF_FAIL(pos > pend ? INF_ERR_OUTPUT : INF_ERR_ISIZE)
    if (open_size) out_size[m] = pos - pstart;
    for (u32 g = pos & ~255u; g < pos; g += 256u) inf_flush(S.ring, ob, g, pstart, pos, lane);
#undef INF_REFILL
}

// ---------------------------------------------------------------------------------------------------------------
// CRC-32 (the gzip polynomial, reflected): a member's output in 64 slices, one per lane, byte-wise through a table in LDS;
// the slices are joined as crc(A | B) = crc(A) x^(8 |B|) mod P  xor  crc(B)   (polynomial arithmetic over GF(2)).
#define CRC_POLY 0xedb88320u
__device__ __forceinline__ u32 crc_mulmod(u32 a, u32 b)
{
    u32 p = 0;
    for (u32 i = 0; i < 32u; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ __forceinline__ u32 crc_xpow8(u32 nbytes)   // x^(8 nbytes) mod P
{
    u32 p = 0x80000000u, sq = 0x40000000u;   // 1, x
    sq = crc_mulmod(sq, sq);                 // x^2
    sq = crc_mulmod(sq, sq);                 // x^4
    sq = crc_mulmod(sq, sq);                 // x^8
    while (nbytes) {
        if (nbytes & 1u) p = crc_mulmod(sq, p);
        sq = crc_mulmod(sq, sq);
        nbytes >>= 1;
    }
    return p;
}

// (first version: one byte load per step and lane, 1 KB apart between lanes: 34 ms for 2.1 GB.  Now 16-byte aligned loads
// and four table lookups per dword -- slicing by 4, tables in LDS.)
__global__ void __launch_bounds__(256)
k_bgzf_crc(const u8 *__restrict__ out, const DMember *__restrict__ mem, u32 nmem, u32 *__restrict__ status)
{
    __shared__ u32 T[4][256];
    {
        u32 c = threadIdx.x;
        for (u32 k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        T[0][threadIdx.x] = c;
        __syncthreads();
        for (u32 k = 1; k < 4; k++) {
            c = (c >> 8) ^ T[0][c & 255u];
            T[k][threadIdx.x] = c;
        }
    }
    __syncthreads();
    const u32 m = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (m >= nmem) return;
    const u64 A = mem[m].out_off, B = A + mem[m].isize, a0 = A & ~15ull;
    const u32 nch = (u32)((B - a0 + 15u) >> 4), cpl = (nch + 63u) / 64u;
    const u32 c0 = lane * cpl < nch ? lane * cpl : nch, c1 = c0 + cpl < nch ? c0 + cpl : nch;
    u64 lo = a0 + 16ull * c0, hi = a0 + 16ull * c1;
    lo = lo < A ? A : lo;
    hi = hi > B ? B : hi;
    if (c0 >= c1) lo = hi = B;
    u32 c = 0xffffffffu;
    for (u32 ch = c0; ch < c1; ch++) {
        const u64 base = a0 + 16ull * ch;
        const uint4 v = *reinterpret_cast<const uint4 *>(out + base);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        if (base >= A && base + 16u <= B) {
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                c ^= w[k];
                c = T[3][c & 255u] ^ T[2][(c >> 8) & 255u] ^ T[1][(c >> 16) & 255u] ^ T[0][c >> 24];
            }
        } else {
#pragma unroll
            for (u32 k = 0; k < 16; k++) {
                const u64 at = base + k;
                if (at >= A && at < B) c = T[0][(c ^ (w[k >> 2] >> (8u * (k & 3u)))) & 255u] ^ (c >> 8);
            }
        }
    }
    c = (lo < hi) ? ~c : 0u;
    u32 x = crc_mulmod(crc_xpow8((u32)(B - hi)), c);
    for (u32 o = 32; o; o >>= 1) x ^= (u32)__shfl_xor((int)x, (int)o, 64);
    if (lane == 0 && x != mem[m].crc && status[m] == 0) status[m] = INF_ERR_CRC;
}

// ---------------------------------------------------------------------------------------------------------------
// The record chain.  Alignment records are a linked list through the inflated stream (a 4-byte block_size in front of
// each), which only a serial walk from the first record finds for certain.  Here the stream (from the first record on) is
// cut into 16-KB pieces; k_bam_spec GUESSES the first record of every piece (the first offset where two records in a row
// look like records), k_bam_walk walks every piece from its guess to the piece's end, and the guess of piece c is then
// compared with where piece c - 1 ended.  Piece 0 starts at the first record, so by induction every piece whose guess
// equals its neighbour's end IS on the true chain; a piece whose guess was wrong takes the neighbour's end as its start and
// is walked again, until nothing changes.  The result is the exact chain.  Each walk also applies the filter and counts,
// a prefix sum over the pieces gives every piece its place in the output, and a last walk writes the kept records.
#define WALK_PIECE 16384ull
#define WALK_NONE (~0ull)        // no guess / no record starts in this piece
#define WALK_BAD (~0ull - 1ull)  // the walk met something that is not a record

__device__ __forceinline__ u32 ld16u(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8); }
__device__ __forceinline__ u32 ld32u(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// does a record plausibly start at offset o of D[0, N)?  (only used for guesses: never decides anything by itself)
__device__ bool rec_plausible(const u8 *__restrict__ D, u64 o, u64 N, int nref)
{
    if (o + 36u > N) return false;
    const u8 *p = D + o;
    const u32 bs = ld32u(p);
    if (bs < 32u || bs > (1u << 28) || o + 4u + bs > N) return false;
    const int ref = (int)ld32u(p + 4), pos = (int)ld32u(p + 8);
    const u32 l_name = p[12], n_cig = ld16u(p + 16);
    const int l_seq = (int)ld32u(p + 20), mref = (int)ld32u(p + 24), mpos = (int)ld32u(p + 28);
    if (ref < -1 || ref >= nref || pos < -1 || mref < -1 || mref >= nref || mpos < -1 || l_name == 0 || l_seq < 0) return false;
    const u64 need = 32ull + l_name + 4ull * n_cig + ((u64)l_seq + 1u) / 2u + (u64)l_seq;
    if (need > bs) return false;
    return p[36u + l_name - 1u] == 0;   // the read name is NUL-terminated (36 + l_name <= 4 + need <= o + 4 + bs <= N)
}

__global__ void __launch_bounds__(256)
k_bam_spec(const u8 *__restrict__ D, u64 N, int nref, u64 npieces, u64 *__restrict__ spec)
{
    const u64 c = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63u;
    if (c >= npieces) return;
    if (c == 0) {
        if (lane == 0) spec[0] = 0;
        return;
    }
    const u64 a = c * WALK_PIECE, b = a + WALK_PIECE < N ? a + WALK_PIECE : N;
    u64 found = WALK_NONE;
    for (u64 o = a; o < b; o += 64u) {
        const u64 me = o + lane;
        bool ok = me < b && rec_plausible(D, me, N, nref);
        if (ok) {
            const u64 nx = me + 4u + ld32u(D + me);
            ok = nx == N || rec_plausible(D, nx, N, nref);
        }
        const u64 mk = __ballot(ok);
        if (mk) {
            found = o + (u64)__builtin_ctzll(mk);
            break;
        }
    }
    if (lane == 0) spec[c] = found;
}

struct WalkArgs {
    const u8 *D;
    u64 N;
    int nref;
    u32 mapq_min, flag_exclude;
    int want_ref;
    u64 npieces;
    u64 *spec, *end;
    u32 *cnt, *kept;
    const u64 *kept_base;            // (write pass)
    int *o_ref, *o_pos, *o_len;      // (write pass)
    u8 *o_rev;
    unsigned long long *first_error; // (write pass) min over (offset << 4 | code)
    u32 *nmis;                       // (repair pass) pieces walked again
};
enum { REC_ERR_BS = 1, REC_ERR_EOF = 2, REC_ERR_SHORT = 3, REC_ERR_REF = 4 };

// query length as pysam's infer_query_length(): CIGAR operations that consume the query (M, I, S, =, X)
__device__ __forceinline__ u32 cigar_qlen(const u8 *cig, u32 n)
{
    u32 q = 0;
    for (u32 i = 0; i < n; i++) {
        const u32 v = ld32u(cig + 4u * i);
        if ((0x193u >> (v & 15u)) & 1u) q += v >> 4;
    }
    return q;
}
// a CIGAR of more than 65535 operations lives in the CG:B,I tag behind the placeholder <l_seq>S<ref_len>N (SAM spec 4.2.2)
__device__ bool long_cigar(const u8 *rec, u32 rec_len, u32 l_name, u32 n_cig, u32 l_seq, const u8 *&cig, u32 &n)
{
    if (n_cig != 2u) return false;
    const u8 *c = rec + 32u + l_name;
    const u32 c0 = ld32u(c), c1 = ld32u(c + 4);
    if ((c0 & 15u) != 4u || (c0 >> 4) != l_seq || (c1 & 15u) != 3u) return false;
    u64 p = 32ull + l_name + 8u + ((u64)l_seq + 1u) / 2u + l_seq;
    while (p + 3u <= rec_len) {
        const u8 t0 = rec[p], t1 = rec[p + 1], ty = rec[p + 2];
        p += 3;
        u64 sz;
        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
        else if (ty == 's' || ty == 'S') sz = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
        else if (ty == 'Z' || ty == 'H') {
            u64 q = p;
            while (q < rec_len && rec[q]) q++;
            if (q >= rec_len) return false;
            sz = q - p + 1u;
        } else if (ty == 'B') {
            if (p + 5u > rec_len) return false;
            const u8 sub = rec[p];
            const u32 cntv = ld32u(rec + p + 1);
            const u64 es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                if (p + 5u + (u64)cntv * 4u > rec_len) return false;
                cig = rec + p + 5;
                n = cntv;
                return true;
            }
            sz = 5u + (u64)cntv * es;
        } else {
            return false;
        }
        p += sz;
    }
    return false;
}

// MODE 0: walk from the guess and count; 1: compare the guess with the neighbour's end, walk again when it was wrong;
// 2: walk the verified chain, report malformed records, write the kept ones
template <int MODE>
__global__ void __launch_bounds__(64) k_bam_walk(const WalkArgs A)
{
    const u64 c = (u64)blockIdx.x * 64u + threadIdx.x;
    if (c >= A.npieces) return;
    u64 s = A.spec[c];
    if (MODE == 1) {
        if (c == 0) return;
        const u64 e = A.end[c - 1];
        // (a neighbour without an end yet -- a piece inside a record longer than a piece, before ITS neighbour's end reached it -- says
        // nothing: adopting its "nowhere" would replace a right guess by a wrong one and send a wave of repairs to the end of the file,
        // one piece per round; found by tools/fuzz_ingest.py: 6.1 M walks for a 4104-piece file of 45-KB records, 3 rounds now)
        if (e == WALK_BAD || e == WALK_NONE || e == s) return;
        s = e;
        A.spec[c] = e;
        atomicAdd(A.nmis, 1u);
    }
    const u64 cend = (c + 1u) * WALK_PIECE;
    const u8 *__restrict__ D = A.D;
    u32 nrec = 0, nkept = 0;
    u64 w = MODE == 2 ? A.kept_base[c] : 0;
    while (s < cend && s < A.N) {
        u32 err = 0, bs = 0;
        if (s + 4u > A.N) {
            err = REC_ERR_EOF;
        } else {
            bs = ld32u(D + s);
            if (bs < 32u) err = REC_ERR_BS;
            else if (s + 4u + bs > A.N) err = REC_ERR_EOF;
        }
        const u8 *rec = D + s + 4;
        u32 l_name = 0, n_cig = 0;
        int ref = -1;
        if (!err) {
            ref = (int)ld32u(rec);
            l_name = rec[8];
            n_cig = ld16u(rec + 12);
            if (32ull + l_name + 4ull * n_cig > bs) err = REC_ERR_SHORT;
            else if (ref >= A.nref) err = REC_ERR_REF;
        }
        if (err) {
            if (MODE == 2) atomicMin(A.first_error, (unsigned long long)((s << 4) | err));
            s = WALK_BAD;
            break;
        }
        const u32 mapq = rec[9], flag = ld16u(rec + 14);
        if (!((flag & A.flag_exclude) || mapq < A.mapq_min || ref < 0 || (A.want_ref >= 0 && ref != A.want_ref))) {
            const u32 l_seq = ld32u(rec + 16);
            const u8 *cig = rec + 32u + l_name;
            u32 n = n_cig;
            long_cigar(rec, bs, l_name, n_cig, l_seq, cig, n);
            const u32 q = cigar_qlen(cig, n);
            if (q) {
                if (MODE == 2) {
                    A.o_ref[w] = ref;
                    A.o_pos[w] = (int)ld32u(rec + 4) + 1;
                    A.o_len[w] = (int)q;
                    A.o_rev[w] = (flag & 0x10u) ? 1 : 0;
                    w++;
                }
                nkept++;
            }
        }
        nrec++;
        s += 4ull + bs;
    }
    if (MODE != 2) {
        A.end[c] = s;
        A.cnt[c] = nrec;
        A.kept[c] = nkept;
    }
}

// exclusive prefix sums of kept[] (one workgroup: a few hundred thousand pieces at most per GB), totals of kept[] and cnt[]
__global__ void __launch_bounds__(1024) k_bam_scan(const u32 *__restrict__ kept, const u32 *__restrict__ cnt, u64 n,
                                                  u64 *__restrict__ kept_base, u64 *__restrict__ totals)
{
    __shared__ u64 sk[1024], sc[1024];
    const u32 t = threadIdx.x;
    const u64 per = (n + 1023u) / 1024u, lo = (u64)t * per < n ? (u64)t * per : n, hi = lo + per < n ? lo + per : n;
    u64 a = 0, b = 0;
    for (u64 i = lo; i < hi; i++) {
        a += kept[i];
        b += cnt[i];
    }
    sk[t] = a;
    sc[t] = b;
    __syncthreads();
    for (u32 o = 1; o < 1024u; o <<= 1) {
        const u64 x = t >= o ? sk[t - o] : 0, y = t >= o ? sc[t - o] : 0;
        __syncthreads();
        sk[t] += x;
        sc[t] += y;
        __syncthreads();
    }
    u64 run = sk[t] - a;
    for (u64 i = lo; i < hi; i++) {
        kept_base[i] = run;
        run += kept[i];
    }
    if (t == 1023u) {
        totals[0] = sk[t];
        totals[1] = sc[t];
    }
}

// runs of one reference among the kept records: one entry per place where the reference id changes
struct RunEntry {
    unsigned long long start;
    int ref, first_pos, prev_pos, pad;
};
#define RUNS_MAX 65536u
__global__ void __launch_bounds__(256) k_ref_runs(const int *__restrict__ ref, const int *__restrict__ pos, u64 n, RunEntry *__restrict__ runs,
                                                 u32 *__restrict__ nruns)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int r = ref[i];
    if (i == 0 || ref[i - 1] != r) {
        const u32 slot = atomicAdd(nruns, 1u);
        if (slot < RUNS_MAX) {
            RunEntry e;
            e.start = i;
            e.ref = r;
            e.first_pos = pos[i];
            e.prev_pos = i ? pos[i - 1] : 0;
            e.pad = 0;
            runs[slot] = e;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPOK(call)                                                                                              \
    {                                                                                                            \
        const hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail(PMX_DBAM_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    }
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline u32 h16(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8); }
inline u32 h32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

constexpr size_t STAGE_PAYLOAD = 32u << 20;   // bytes of the file per staging buffer
constexpr size_t STAGE_HEAD = 65536;          // the tail of the previous piece in front of it: every member lies whole in one buffer
constexpr int NSTAGE = 3;
constexpr size_t IN_PAD = 1024;               // the bit reader looks up to two 256-byte pieces ahead

// page-locked staging buffers, kept for the life of the process (pinning them costs more than a small file's ingest)
std::mutex g_stage_mu;   // (one open at a time goes through the staging buffers)
struct Staging {
    u8 *buf[NSTAGE] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NSTAGE];
    bool used[NSTAGE] = {false, false, false};
    bool ready = false;
};
Staging g_stage;

const char *inf_err_text(u32 code)
{
    switch (code) {
    case INF_ERR_BTYPE: return "corrupt DEFLATE stream in a BGZF block: reserved block type";
    case INF_ERR_STORED: return "corrupt DEFLATE stream in a BGZF block: stored block with LEN != ~NLEN";
    case INF_ERR_TABLE: return "corrupt DEFLATE stream in a BGZF block: malformed Huffman code lengths";
    case INF_ERR_CODE: return "corrupt DEFLATE stream in a BGZF block: bit pattern that is no Huffman code of its block";
    case INF_ERR_DIST: return "corrupt DEFLATE stream in a BGZF block: match distance beyond the start of the block";
    case INF_ERR_OUTPUT: return "BGZF block does not inflate to its recorded size (more output than ISIZE)";
    case INF_ERR_INPUT: return "DEFLATE stream runs past the BGZF block";
    case INF_ERR_ISIZE: return "BGZF block does not inflate to its recorded size";
    case INF_ERR_CRC: return "BGZF block CRC32 mismatch";
    }
    return "unknown inflate error";
}

}  // namespace

struct pmx_dbam {
    int device = 0;
    hipStream_t stream = nullptr;    // copies (and, after open, everything else)
    hipStream_t kstream = nullptr;   // open: the inflate / CRC launches of the pieces already in HBM, beside the copies of the next ones
    hipStream_t kmore[3] = {nullptr, nullptr, nullptr};   // ... dealt over four streams: a piece is ~1000 members, a quarter of what
    u32 kturn = 0;                                        // fills the GPU, and a member takes milliseconds whatever runs beside it
    DMember *d_hmem = nullptr;       // (its address on the device)
    DMember *h_mem = nullptr;        // the member table the kernels of open read: page-locked host memory, filled while they run
    u64 mem_cap = 0, dout_cap = 0;   // entries of h_mem / d_status, bytes of d_out
    u32 launched = 0;                // members handed to the kernels so far
    bool pipelined = true;
    u64 fsize = 0, N = 0, data_beg = 0;
    u8 *d_in = nullptr, *d_out = nullptr;
    u32 *d_status = nullptr;
    std::vector<DMember> members;
    std::string text;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    // record chain + results
    u64 npieces = 0;
    u64 *d_spec = nullptr, *d_end = nullptr, *d_kept_base = nullptr, *d_totals = nullptr;
    u32 *d_cnt = nullptr, *d_kept = nullptr, *d_nmis = nullptr;
    unsigned long long *d_first_error = nullptr;
    bool chain_ready = false;
    u32 chain_mapq = 0, chain_flags = 0;
    int chain_ref = -1;
    int *d_ref = nullptr, *d_pos = nullptr, *d_len = nullptr;
    u8 *d_rev = nullptr;
    u64 out_cap = 0, n_kept = 0, n_records = 0, n_rewalked = 0;
    double t[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

int sync_kernels(pmx_dbam &b)
{
    HIPOK(hipStreamSynchronize(b.kstream));
    for (hipStream_t x : b.kmore)
        if (x) HIPOK(hipStreamSynchronize(x));
    return 0;
}

// Room for `need` members and `need_out` bytes of output.  Growing waits for the kernels in flight (they read the old table
// and write the old buffers), copies what exists, and carries on: files of ordinary members (>= 8 KB compressed, <= 6x) never grow.
int ensure_room(pmx_dbam &b, u64 need, u64 need_out)
{
    if (need > b.mem_cap) {
        const u64 cap = std::max<u64>(need + need / 2, b.fsize / 8192 + 4096);
        DMember *h = nullptr;
        u32 *st = nullptr;
        HIPOK(hipHostMalloc((void **)&h, sizeof(DMember) * cap, hipHostMallocMapped));
        HIPOK(hipMalloc((void **)&st, sizeof(u32) * cap));
        { const int rc_ = sync_kernels(b); if (rc_) return rc_; }
        HIPOK(hipMemsetAsync(st, 0, sizeof(u32) * cap, b.kstream));
        if (b.h_mem) {
            memcpy(h, b.h_mem, sizeof(DMember) * b.launched);   // (what the table holds: the members already launched)
            HIPOK(hipMemcpyAsync(st, b.d_status, sizeof(u32) * b.mem_cap, hipMemcpyDeviceToDevice, b.kstream));
            { const int rc_ = sync_kernels(b); if (rc_) return rc_; }
            HIPOK(hipHostFree(b.h_mem));
            HIPOK(hipFree(b.d_status));
        }
        b.h_mem = h;
        HIPOK(hipHostGetDevicePointer((void **)&b.d_hmem, h, 0));
        b.d_status = st;
        b.mem_cap = cap;
    }
    if (need_out + 64 > b.dout_cap) {
        const u64 cap = std::max<u64>(need_out + need_out / 2 + (1u << 20), 6 * b.fsize + (1u << 20));
        u8 *o = nullptr;
        HIPOK(hipMalloc((void **)&o, cap));
        if (b.d_out) {
            { const int rc_ = sync_kernels(b); if (rc_) return rc_; }
            HIPOK(hipMemcpyAsync(o, b.d_out, b.dout_cap, hipMemcpyDeviceToDevice, b.kstream));
            { const int rc_ = sync_kernels(b); if (rc_) return rc_; }
            HIPOK(hipFree(b.d_out));
        }
        b.d_out = o;
        b.dout_cap = cap;
    }
    return 0;
}

// inflate + CRC of the members scanned so far and not yet launched, behind the copy that `after` marks
int launch_members(pmx_dbam &b, hipEvent_t after, u64 out_end)
{
    const u32 m1 = (u32)b.members.size(), m0 = b.launched;
    if (m1 == m0) return 0;
    int rc = ensure_room(b, m1, out_end);
    if (rc) return rc;
    memcpy(b.h_mem + m0, b.members.data() + m0, sizeof(DMember) * (m1 - m0));
    const u32 turn = b.kturn++ & 3u;
    hipStream_t ks = (turn == 0 || !b.kmore[turn - 1]) ? b.kstream : b.kmore[turn - 1];
    HIPOK(hipStreamWaitEvent(ks, after, 0));
    const u32 n = m1 - m0;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(n), dim3(64), 0, ks, b.d_in, b.d_out, b.d_hmem + m0, n, b.d_status + m0, (u32 *)nullptr);
    HIPOK(hipGetLastError());
    hipLaunchKernelGGL(k_bgzf_crc, dim3((n + 3) / 4), dim3(256), 0, ks, b.d_out, b.d_hmem + m0, n, b.d_status + m0);
    HIPOK(hipGetLastError());
    b.launched = m1;
    return 0;
}

int read_and_upload(pmx_dbam &b, const char *path, int nthreads)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(PMX_DBAM_ERR_OPEN, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return fail(PMX_DBAM_ERR_OPEN, std::string("not a regular file: ") + path);
    }
    b.fsize = (u64)st.st_size;
    struct Closer {
        int fd;
        ~Closer() { close(fd); }
    } closer{fd};
    if (b.fsize < 28) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF block header");
    std::lock_guard<std::mutex> stage_guard(g_stage_mu);
    // the buffers' events are recorded on THIS handle's stream, which does not outlive the handle: whatever way this function is
    // left, the copies are waited for and the buffers are marked free (an event of a destroyed stream cannot be waited on)
    struct StageReset {
        hipStream_t st;
        ~StageReset()
        {
            (void)hipStreamSynchronize(st);
            for (bool &u : g_stage.used) u = false;
        }
    } stage_reset{b.stream};
    if (!g_stage.ready) {
        for (int i = 0; i < NSTAGE; i++) {
            HIPOK(hipHostMalloc((void **)&g_stage.buf[i], STAGE_HEAD + STAGE_PAYLOAD, hipHostMallocDefault));
            HIPOK(hipEventCreateWithFlags(&g_stage.ev[i], hipEventDisableTiming));
        }
        g_stage.ready = true;
    }
    const double tq0 = now_s();
    HIPOK(hipMalloc((void **)&b.d_in, b.fsize + IN_PAD));
    HIPOK(hipMemsetAsync(b.d_in + b.fsize, 0, IN_PAD, b.stream));
    const double t_alloc_in = now_s() - tq0;
    const size_t npieces = (b.fsize + STAGE_PAYLOAD - 1) / STAGE_PAYLOAD;
    double tw = 0, tr = 0, tc = 0, ts = 0, tl = 0, tq;
    (void)tl;   // PMX_DBAM_TIMING=1: waits for a staging buffer, reads, copy calls, scan, launches
    u64 next_off = 0, out_off = 0;
    int prev = -1;
    size_t prev_len = 0, prev_head = 0;
    u64 prev_a = 0;
    double tq2;
    // hop over the members that END in a piece (they lie whole in [lo, hi), the piece and the 64 KB in front of it at `base`);
    // runs on this thread while the NEXT piece is being read
    auto scan_piece = [&](const u8 *base, u64 lo, u64 hi, bool last) -> int {
        for (;;) {
            if (next_off == b.fsize) break;
            if (next_off + 18 > hi) {
                if (last) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF block header");
                break;
            }
            const u8 *p = base + (next_off - lo);
            if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4))
                return fail(PMX_DBAM_ERR_FORMAT, "not a BGZF block (bad gzip magic / no extra field)");
            const u32 xlen = h16(p + 10);
            if (next_off + 12 + xlen + 8 > hi) {
                if (last) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF extra field");
                if (12 + (u64)xlen + 8 > STAGE_HEAD) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF extra field");
                break;
            }
            u32 bsize = 0;
            bool found = false;
            for (u32 x = 0; x + 4 <= xlen;) {
                const u8 *s = p + 12 + x;
                const u32 slen = h16(s + 2);
                if (s[0] == 'B' && s[1] == 'C' && slen == 2 && x + 6 <= xlen) {
                    bsize = h16(s + 4);
                    found = true;
                    break;
                }
                x += 4 + slen;
            }
            if (!found) return fail(PMX_DBAM_ERR_FORMAT, "gzip member without the BGZF 'BC' subfield");
            const u64 total = (u64)bsize + 1;
            if (total < 12 + (u64)xlen + 8) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF block");
            if (next_off + total > hi) {
                if (last) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF block");
                break;
            }
            DMember me;
            me.in_off = next_off + 12 + xlen;
            me.clen = (u32)(total - 12 - xlen - 8);
            me.crc = h32(p + total - 8);
            me.isize = h32(p + total - 4);
            me.out_off = out_off;
            me.open_size = 0;
            if (me.isize > 65536) return fail(PMX_DBAM_ERR_FORMAT, "BGZF block larger than 64 KiB");
            out_off += me.isize;
            b.members.push_back(me);
            next_off += total;
        }
        return 0;
    };
    for (size_t k = 0; k < npieces; k++) {
        const int j = (int)(k % NSTAGE);
        tq = now_s();
        if (g_stage.used[j]) HIPOK(hipEventSynchronize(g_stage.ev[j]));
        tw += now_s() - tq;
        tq = now_s();
        u8 *buf = g_stage.buf[j];
        const u64 a = (u64)k * STAGE_PAYLOAD;
        const size_t len = (size_t)std::min<u64>(STAGE_PAYLOAD, b.fsize - a);
        // the file -> the buffer, on several threads (a pread from the page cache is a kernel memcpy)
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)nthreads, len >> 20));
        std::vector<std::thread> th;
        std::vector<int> bad((size_t)T, 0);
        for (int ti = 0; ti < T; ti++)
            th.emplace_back([&, ti] {
                const size_t lo = len * (size_t)ti / (size_t)T, hi = len * (size_t)(ti + 1) / (size_t)T;
                size_t done = lo;
                while (done < hi) {
                    const ssize_t r = pread(fd, buf + STAGE_HEAD + done, hi - done, (off_t)(a + done));
                    if (r <= 0) {
                        bad[(size_t)ti] = 1;
                        return;
                    }
                    done += (size_t)r;
                }
            });
        size_t head = 0;
        if (prev >= 0) {   // the last 64 KB of the previous piece in front of this one (meanwhile, on this thread)
            head = std::min(prev_len, STAGE_HEAD);
            memcpy(buf + STAGE_HEAD - head, g_stage.buf[prev] + STAGE_HEAD + prev_len - head, head);
        }
        if (prev >= 0) {   // the previous piece's members, while this piece is read
            tq2 = now_s();
            const size_t phead = prev_head;
            const int rc = scan_piece(g_stage.buf[prev] + STAGE_HEAD - phead, prev_a - phead, prev_a + prev_len, false);
            ts += now_s() - tq2;
            if (rc) {
                for (auto &x : th) x.join();
                return rc;
            }
            if (b.pipelined) {
                const int rc2 = launch_members(b, g_stage.ev[prev], out_off);
                if (rc2) {
                    for (auto &x : th) x.join();
                    return rc2;
                }
            }
        }
        for (auto &x : th) x.join();
        for (int v : bad)
            if (v) return fail(PMX_DBAM_ERR_OPEN, std::string("read error on ") + path);
        tr += now_s() - tq;
        tq = now_s();
        HIPOK(hipMemcpyAsync(b.d_in + a, buf + STAGE_HEAD, len, hipMemcpyHostToDevice, b.stream));
        HIPOK(hipEventRecord(g_stage.ev[j], b.stream));
        g_stage.used[j] = true;
        tc += now_s() - tq;
        tq = now_s();
        prev = j;
        prev_len = len;
        prev_a = a;
        prev_head = head;
    }
    if (prev >= 0) {   // the last piece
        tq = now_s();
        const int rc = scan_piece(g_stage.buf[prev] + STAGE_HEAD - prev_head, prev_a - prev_head, prev_a + prev_len, true);
        ts += now_s() - tq;
        if (rc) return rc;
        if (b.pipelined) {
            const int rc2 = launch_members(b, g_stage.ev[prev], out_off);
            if (rc2) return rc2;
        }
    }
    if (next_off != b.fsize) return fail(PMX_DBAM_ERR_FORMAT, "truncated BGZF block");
    b.N = out_off;
    tq = now_s();
    HIPOK(hipStreamSynchronize(b.stream));
    if (getenv("PMX_DBAM_TIMING"))
        fprintf(stderr, "[pmx_dbam] %zu pieces: buffer waits %.1f ms, reads %.1f, copy calls %.1f, member scan %.1f, last copy %.1f, hipMalloc of the file's copy %.1f\n",
                npieces, tw * 1e3, tr * 1e3, tc * 1e3, ts * 1e3, (now_s() - tq) * 1e3, t_alloc_in * 1e3);
    return 0;
}

int check_members(pmx_dbam &b)
{
    const u32 nmem = (u32)b.members.size();
    std::vector<u32> status(nmem);
    if (nmem) HIPOK(hipMemcpy(status.data(), b.d_status, sizeof(u32) * nmem, hipMemcpyDeviceToHost));
    for (u32 i = 0; i < nmem; i++)
        if (status[i]) {
            char where[96];
            snprintf(where, sizeof where, " (member %u at file offset %llu)", i, (unsigned long long)b.members[i].in_off);
            return fail(PMX_DBAM_ERR_FORMAT, std::string(inf_err_text(status[i])) + where);
        }
    // the compressed copy is not needed any more
    const double tf = now_s();
    HIPOK(hipFree(b.d_in));
    b.d_in = nullptr;
    if (getenv("PMX_DBAM_TIMING")) fprintf(stderr, "[pmx_dbam] hipFree of the file's copy %.1f ms\n", (now_s() - tf) * 1e3);
    return 0;
}

// the pipelined open: wait for the kernels of the last pieces, give back what d_out was allocated beyond the stream
int finish_pipeline(pmx_dbam &b)
{
    double t0 = now_s();
    {
        const int rc = sync_kernels(b);
        if (rc) return rc;
    }
    b.t[1] = now_s() - t0;
    if (!b.d_out) HIPOK(hipMalloc((void **)&b.d_out, 64));
    if (b.dout_cap > b.N + b.N / 4 + (64u << 20)) {
        u8 *o = nullptr;
        HIPOK(hipMalloc((void **)&o, b.N + 64));
        HIPOK(hipMemcpy(o, b.d_out, b.N, hipMemcpyDeviceToDevice));
        HIPOK(hipFree(b.d_out));
        b.d_out = o;
        b.dout_cap = b.N + 64;
    }
    return check_members(b);
}

// the default order: everything copied first, then ONE inflate launch and ONE CRC launch, each timed
int inflate_all(pmx_dbam &b)
{
    const u32 nmem = (u32)b.members.size();
    const double ta = now_s();
    int rc = ensure_room(b, std::max<u32>(nmem, 1), b.N);
    if (rc) return rc;
    if (getenv("PMX_DBAM_TIMING")) fprintf(stderr, "[pmx_dbam] member table + output buffer allocated in %.1f ms\n", (now_s() - ta) * 1e3);
    if (!nmem) return check_members(b);
    memcpy(b.h_mem, b.members.data(), sizeof(DMember) * nmem);
    double t0 = now_s();
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(nmem), dim3(64), 0, b.kstream, b.d_in, b.d_out, b.d_hmem, nmem, b.d_status, (u32 *)nullptr);
    HIPOK(hipGetLastError());
    HIPOK(hipStreamSynchronize(b.kstream));
    double t1 = now_s();
    b.t[1] = t1 - t0;
    hipLaunchKernelGGL(k_bgzf_crc, dim3((nmem + 3) / 4), dim3(256), 0, b.kstream, b.d_out, b.d_hmem, nmem, b.d_status);
    HIPOK(hipGetLastError());
    HIPOK(hipStreamSynchronize(b.kstream));
    b.t[2] = now_s() - t1;
    return check_members(b);
}

int parse_header(pmx_dbam &b)
{
    std::vector<u8> h;
    auto need = [&](u64 upto) -> int {
        if (upto > b.N) return fail(PMX_DBAM_ERR_FORMAT, "file ends inside the BAM header");
        if (h.size() >= upto) return 0;
        const u64 want = std::min<u64>(b.N, std::max<u64>(upto, std::max<u64>(1u << 20, 2 * h.size())));
        h.resize(want);
        HIPOK(hipMemcpy(h.data(), b.d_out, want, hipMemcpyDeviceToHost));
        return 0;
    };
    int rc;
    if ((rc = need(12))) return rc;
    if (memcmp(h.data(), "BAM\1", 4) != 0) return fail(PMX_DBAM_ERR_FORMAT, "not a BAM file (bad magic)");
    const u32 l_text = h32(h.data() + 4);
    if ((rc = need(12 + (u64)l_text))) return rc;
    b.text.assign((const char *)h.data() + 8, l_text);
    while (!b.text.empty() && b.text.back() == '\0') b.text.pop_back();
    u64 p = 8 + (u64)l_text;
    const u32 n_ref = h32(h.data() + p);
    p += 4;
    for (u32 i = 0; i < n_ref; i++) {
        if ((rc = need(p + 4))) return rc;
        const u32 l_name = h32(h.data() + p);
        if (l_name == 0 || l_name > (1u << 20)) return fail(PMX_DBAM_ERR_FORMAT, "bad reference name length");
        if ((rc = need(p + 4 + l_name + 4))) return rc;
        const char *nm = (const char *)h.data() + p + 4;
        b.ref_names.emplace_back(nm, strnlen(nm, l_name));
        b.ref_lens.push_back((int64_t)h32(h.data() + p + 4 + l_name));
        p += 4 + (u64)l_name + 4;
    }
    b.data_beg = p;
    return 0;
}

int free_chain(pmx_dbam &b)
{
    for (void *p : {(void *)b.d_spec, (void *)b.d_end, (void *)b.d_kept_base, (void *)b.d_totals, (void *)b.d_cnt, (void *)b.d_kept,
                    (void *)b.d_nmis, (void *)b.d_first_error, (void *)b.d_ref, (void *)b.d_pos, (void *)b.d_len, (void *)b.d_rev})
        if (p) (void)hipFree(p);
    b.d_spec = b.d_end = b.d_kept_base = b.d_totals = nullptr;
    b.d_cnt = b.d_kept = b.d_nmis = nullptr;
    b.d_first_error = nullptr;
    b.d_ref = b.d_pos = b.d_len = nullptr;
    b.d_rev = nullptr;
    b.out_cap = 0;
    return 0;
}

}  // namespace

extern "C" {

const char *pmx_dbam_last_error(void) { return g_err.c_str(); }
int pmx_dbam_version(void) { return 1; }

static int dbam_open_impl(const char *path, int device, int nthreads, pmx_dbam **out);
int pmx_dbam_open(const char *path, int device, int nthreads, pmx_dbam **out)
{
    try {   // (std::bad_alloc, std::system_error from a thread that cannot be started: an exception must not leave the C ABI)
        return dbam_open_impl(path, device, nthreads, out);
    } catch (const std::exception &e) {
        return fail(PMX_DBAM_ERR_OPEN, std::string("pmx_dbam_open: ") + e.what());
    }
}
static int dbam_open_impl(const char *path, int device, int nthreads, pmx_dbam **out)
{
    if (!path || !out) return fail(PMX_DBAM_ERR_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PMX_DBAM_ERR_DEVICE, "no HIP device: the device ingest needs a GPU");
    if (device < 0 || device >= ndev) return fail(PMX_DBAM_ERR_INVALID, "no such device");
    HIPOK(hipSetDevice(device));
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    pmx_dbam *b = new pmx_dbam;
    b->device = device;
    int rc = 0;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&b->kstream, hipStreamNonBlocking) != hipSuccess) {
        if (b->stream) (void)hipStreamDestroy(b->stream);
        delete b;
        return fail(PMX_DBAM_ERR_DEVICE, "hipStreamCreate failed");
    }
    for (hipStream_t &x : b->kmore)
        if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) x = nullptr;
    double t0 = now_s();
    {
        // PMX_DBAM_PIPELINE=1: inflate the pieces already copied while the next ones are read and copied.  Off by default: on the
        // boxes measured the copies then wait for the inflate wavefronts (which fill every SIMD's registers) and the sum is the
        // same 0.118 s for a 1.1-GB file, while the serial order gives one timed launch per phase.
        const char *pl = getenv("PMX_DBAM_PIPELINE");
        b->pipelined = pl && pl[0] == '1';
    }
    rc = read_and_upload(*b, path, nthreads);
    b->t[0] = now_s() - t0;
    if (!rc) rc = b->pipelined ? finish_pipeline(*b) : inflate_all(*b);
    if (!rc) {
        t0 = now_s();
        rc = parse_header(*b);
        b->t[3] = now_s() - t0;
    }
    if (rc) {
        const std::string keep = g_err;
        pmx_dbam_close(b);
        g_err = keep;
        return rc;
    }
    b->npieces = b->N > b->data_beg ? (b->N - b->data_beg + WALK_PIECE - 1) / WALK_PIECE : 0;
    *out = b;
    return 0;
}

void pmx_dbam_close(pmx_dbam *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->kstream) (void)sync_kernels(*b);
    free_chain(*b);
    for (hipStream_t x : b->kmore)
        if (x) (void)hipStreamDestroy(x);
    if (b->h_mem) (void)hipHostFree(b->h_mem);
    if (b->kstream) (void)hipStreamDestroy(b->kstream);
    if (b->d_in) (void)hipFree(b->d_in);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->d_status) (void)hipFree(b->d_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int32_t pmx_dbam_nref(const pmx_dbam *b) { return b ? (int32_t)b->ref_names.size() : 0; }
const char *pmx_dbam_ref_name(const pmx_dbam *b, int32_t i)
{
    return (b && i >= 0 && (size_t)i < b->ref_names.size()) ? b->ref_names[(size_t)i].c_str() : nullptr;
}
int64_t pmx_dbam_ref_len(const pmx_dbam *b, int32_t i)
{
    return (b && i >= 0 && (size_t)i < b->ref_lens.size()) ? b->ref_lens[(size_t)i] : -1;
}
const char *pmx_dbam_header_text(const pmx_dbam *b, uint32_t *len)
{
    if (!b) return nullptr;
    if (len) *len = (uint32_t)b->text.size();
    return b->text.c_str();
}

static int64_t dbam_decode_impl(pmx_dbam *b, uint32_t mapq_min, uint32_t flag_exclude, int32_t want_ref);
int64_t pmx_dbam_decode(pmx_dbam *b, uint32_t mapq_min, uint32_t flag_exclude, int32_t want_ref)
{
    try {
        return dbam_decode_impl(b, mapq_min, flag_exclude, want_ref);
    } catch (const std::exception &e) {
        return fail(PMX_DBAM_ERR_OPEN, std::string("pmx_dbam_decode: ") + e.what());
    }
}
static int64_t dbam_decode_impl(pmx_dbam *b, uint32_t mapq_min, uint32_t flag_exclude, int32_t want_ref)
{
    if (!b) return fail(PMX_DBAM_ERR_INVALID, "null handle");
    HIPOK(hipSetDevice(b->device));
    b->n_kept = b->n_records = 0;
    if (b->npieces == 0) return 0;
    const u64 np = b->npieces;
    const u64 N = b->N - b->data_beg;
    const u8 *D = b->d_out + b->data_beg;
    double t0 = now_s();
    if (!b->d_spec) {
        // (all of the chain's tables or none: a failed allocation must not leave a handle that skips this block next time)
        struct Undo {
            pmx_dbam *b;
            bool keep = false;
            ~Undo()
            {
                if (!keep) free_chain(*b);
            }
        } undo{b};
        HIPOK(hipMalloc((void **)&b->d_end, 8 * np));
        HIPOK(hipMalloc((void **)&b->d_kept_base, 8 * np));
        HIPOK(hipMalloc((void **)&b->d_totals, 16));
        HIPOK(hipMalloc((void **)&b->d_cnt, 4 * np));
        HIPOK(hipMalloc((void **)&b->d_kept, 4 * np));
        HIPOK(hipMalloc((void **)&b->d_nmis, 4));
        HIPOK(hipMalloc((void **)&b->d_first_error, 8));
        HIPOK(hipMalloc((void **)&b->d_spec, 8 * np));
        undo.keep = true;
        hipLaunchKernelGGL(k_bam_spec, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, b->stream, D, N, (int)b->ref_names.size(), np, b->d_spec);
        HIPOK(hipGetLastError());
        b->chain_ready = false;
        b->n_rewalked = 0;
    }
    WalkArgs A;
    A.D = D;
    A.N = N;
    A.nref = (int)b->ref_names.size();
    A.mapq_min = mapq_min;
    A.flag_exclude = flag_exclude;
    A.want_ref = want_ref;
    A.npieces = np;
    A.spec = b->d_spec;
    A.end = b->d_end;
    A.cnt = b->d_cnt;
    A.kept = b->d_kept;
    A.kept_base = b->d_kept_base;
    A.o_ref = A.o_pos = A.o_len = nullptr;
    A.o_rev = nullptr;
    A.first_error = b->d_first_error;
    A.nmis = b->d_nmis;
    const dim3 wg((unsigned)((np + 63) / 64));
    // walk every piece from its (guessed, or already verified) start and count with this filter
    hipLaunchKernelGGL(k_bam_walk<0>, wg, dim3(64), 0, b->stream, A);
    HIPOK(hipGetLastError());
    // close the chain: pieces whose start is not where their neighbour ended are walked again
    for (u64 round = 0; !b->chain_ready; round++) {
        if (round > np + 1) return fail(PMX_DBAM_ERR_DEVICE, "record chain did not close");
        HIPOK(hipMemsetAsync(b->d_nmis, 0, 4, b->stream));
        hipLaunchKernelGGL(k_bam_walk<1>, wg, dim3(64), 0, b->stream, A);
        HIPOK(hipGetLastError());
        u32 nmis = 0;
        HIPOK(hipMemcpyAsync(&nmis, b->d_nmis, 4, hipMemcpyDeviceToHost, b->stream));
        HIPOK(hipStreamSynchronize(b->stream));
        b->n_rewalked += nmis;
        if (nmis == 0) b->chain_ready = true;
    }
    hipLaunchKernelGGL(k_bam_scan, dim3(1), dim3(1024), 0, b->stream, b->d_kept, b->d_cnt, np, b->d_kept_base, b->d_totals);
    HIPOK(hipGetLastError());
    u64 totals[2] = {0, 0};
    HIPOK(hipMemcpyAsync(totals, b->d_totals, 16, hipMemcpyDeviceToHost, b->stream));
    HIPOK(hipStreamSynchronize(b->stream));
    double t1 = now_s();
    b->t[4] = t1 - t0;
    if (totals[0] > b->out_cap) {
        for (void *p : {(void *)b->d_ref, (void *)b->d_pos, (void *)b->d_len, (void *)b->d_rev})
            if (p) (void)hipFree(p);
        b->d_ref = b->d_pos = b->d_len = nullptr;
        b->d_rev = nullptr;
        b->out_cap = 0;
        HIPOK(hipMalloc((void **)&b->d_ref, 4 * totals[0]));
        HIPOK(hipMalloc((void **)&b->d_pos, 4 * totals[0]));
        HIPOK(hipMalloc((void **)&b->d_len, 4 * totals[0]));
        HIPOK(hipMalloc((void **)&b->d_rev, totals[0]));
        b->out_cap = totals[0];
    }
    A.o_ref = b->d_ref;
    A.o_pos = b->d_pos;
    A.o_len = b->d_len;
    A.o_rev = b->d_rev;
    HIPOK(hipMemsetAsync(b->d_first_error, 0xff, 8, b->stream));
    hipLaunchKernelGGL(k_bam_walk<2>, wg, dim3(64), 0, b->stream, A);
    HIPOK(hipGetLastError());
    unsigned long long fe = 0;
    HIPOK(hipMemcpyAsync(&fe, b->d_first_error, 8, hipMemcpyDeviceToHost, b->stream));
    HIPOK(hipStreamSynchronize(b->stream));
    b->t[5] = now_s() - t1;
    if (fe != ~0ull) {
        switch (fe & 15ull) {
        case REC_ERR_BS: return fail(PMX_DBAM_ERR_FORMAT, "BAM record with block_size < 32");
        case REC_ERR_EOF: return fail(PMX_DBAM_ERR_FORMAT, "file ends inside an alignment record");
        case REC_ERR_SHORT: return fail(PMX_DBAM_ERR_FORMAT, "BAM record shorter than its name and CIGAR");
        default: return fail(PMX_DBAM_ERR_FORMAT, "BAM record refers to an unknown reference id");
        }
    }
    b->n_kept = totals[0];
    b->n_records = totals[1];
    b->chain_mapq = mapq_min;
    b->chain_flags = flag_exclude;
    b->chain_ref = want_ref;
    return (int64_t)b->n_kept;
}

int pmx_dbam_device_arrays(const pmx_dbam *b, const int32_t **d_ref_id, const int32_t **d_pos1, const int32_t **d_read_len,
                           const uint8_t **d_reverse)
{
    if (!b) return fail(PMX_DBAM_ERR_INVALID, "null handle");
    if (d_ref_id) *d_ref_id = b->d_ref;
    if (d_pos1) *d_pos1 = b->d_pos;
    if (d_read_len) *d_read_len = b->d_len;
    if (d_reverse) *d_reverse = b->d_rev;
    return 0;
}

int pmx_dbam_fetch(pmx_dbam *b, int64_t first, int64_t n, int32_t *ref_id, int32_t *pos1, int32_t *read_len, uint8_t *reverse)
{
    if (!b || first < 0 || n < 0 || (u64)(first + n) > b->n_kept) return fail(PMX_DBAM_ERR_INVALID, "range outside the kept records");
    if (n == 0) return 0;
    HIPOK(hipSetDevice(b->device));
    if (ref_id) HIPOK(hipMemcpyAsync(ref_id, b->d_ref + first, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
    if (pos1) HIPOK(hipMemcpyAsync(pos1, b->d_pos + first, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
    if (read_len) HIPOK(hipMemcpyAsync(read_len, b->d_len + first, 4 * (size_t)n, hipMemcpyDeviceToHost, b->stream));
    if (reverse) HIPOK(hipMemcpyAsync(reverse, b->d_rev + first, (size_t)n, hipMemcpyDeviceToHost, b->stream));
    HIPOK(hipStreamSynchronize(b->stream));
    return 0;
}

static int64_t dbam_runs_impl(pmx_dbam *b, int64_t cap, int64_t *start, int32_t *ref_id, int32_t *first_pos1, int32_t *last_pos1);
int64_t pmx_dbam_runs(pmx_dbam *b, int64_t cap, int64_t *start, int32_t *ref_id, int32_t *first_pos1, int32_t *last_pos1)
{
    try {
        return dbam_runs_impl(b, cap, start, ref_id, first_pos1, last_pos1);
    } catch (const std::exception &e) {
        return fail(PMX_DBAM_ERR_OPEN, std::string("pmx_dbam_runs: ") + e.what());
    }
}
static int64_t dbam_runs_impl(pmx_dbam *b, int64_t cap, int64_t *start, int32_t *ref_id, int32_t *first_pos1, int32_t *last_pos1)
{
    if (!b) return fail(PMX_DBAM_ERR_INVALID, "null handle");
    if (b->n_kept == 0) return 0;
    HIPOK(hipSetDevice(b->device));
    RunEntry *d_runs = nullptr;
    u32 *d_n = nullptr;
    HIPOK(hipMalloc((void **)&d_runs, sizeof(RunEntry) * RUNS_MAX));
    HIPOK(hipMalloc((void **)&d_n, 4));
    HIPOK(hipMemsetAsync(d_n, 0, 4, b->stream));
    hipLaunchKernelGGL(k_ref_runs, dim3((unsigned)((b->n_kept + 255) / 256)), dim3(256), 0, b->stream, b->d_ref, b->d_pos, b->n_kept, d_runs, d_n);
    u32 n = 0;
    int last = 0;
    std::vector<RunEntry> runs;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&last, b->d_pos + (b->n_kept - 1), 4, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess && n <= RUNS_MAX) {
        runs.resize(n);
        e = hipMemcpy(runs.data(), d_runs, sizeof(RunEntry) * n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_runs);
    (void)hipFree(d_n);
    if (e != hipSuccess) return fail(PMX_DBAM_ERR_DEVICE, std::string("pmx_dbam_runs: ") + hipGetErrorString(e));
    if (n > RUNS_MAX) return fail(PMX_DBAM_ERR_INVALID, "more than 65536 runs of one reference (unsorted file): use pmx_dbam_fetch");
    if (!start) return (int64_t)n;
    std::sort(runs.begin(), runs.end(), [](const RunEntry &x, const RunEntry &y) { return x.start < y.start; });
    const int64_t m = std::min<int64_t>(cap, (int64_t)n);
    for (int64_t r = 0; r < m; r++) {
        start[r] = (int64_t)runs[(size_t)r].start;
        if (ref_id) ref_id[r] = runs[(size_t)r].ref;
        if (first_pos1) first_pos1[r] = runs[(size_t)r].first_pos;
        if (last_pos1) last_pos1[r] = (size_t)r + 1 < runs.size() ? runs[(size_t)r + 1].prev_pos : last;
    }
    return m;
}

int pmx_dbam_counters(const pmx_dbam *b, uint64_t *records, uint64_t *kept, uint64_t *bytes_out, uint64_t *bytes_in,
                      uint64_t *members, uint64_t *rewalked)
{
    if (!b) return fail(PMX_DBAM_ERR_INVALID, "null handle");
    if (records) *records = b->n_records;
    if (kept) *kept = b->n_kept;
    if (bytes_out) *bytes_out = b->N;
    if (bytes_in) *bytes_in = b->fsize;
    if (members) *members = b->members.size();
    if (rewalked) *rewalked = b->n_rewalked;
    return 0;
}

int pmx_dbam_timings(const pmx_dbam *b, double t[6])
{
    if (!b || !t) return fail(PMX_DBAM_ERR_INVALID, "null argument");
    for (int i = 0; i < 6; i++) t[i] = b->t[i];
    return 0;
}

int pmx_dbam_inflated(pmx_dbam *b, uint64_t first, uint64_t n, uint8_t *dst)
{
    if (!b || !dst || first + n > b->N) return fail(PMX_DBAM_ERR_INVALID, "range outside the inflated stream");
    if (n == 0) return 0;
    HIPOK(hipSetDevice(b->device));
    HIPOK(hipMemcpy(dst, b->d_out + first, n, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

#include "bigwig_device.inc"
