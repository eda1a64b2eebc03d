// Dense word-parallel cross-correlation kernels (gfx950).
//
// Replaces the reference's per-shift full-vector passes (PyMaSC/core/bitarray/mscc.pyx:288-317:
// alloc_and x3, count x3, acount x2, lshift, rshift -- ~20 passes over N/64 words per shift) with ONE
// pass over the vectors: a workgroup stages a tile of F / R / M words in LDS once and every lane
// evaluates its own shift d against it with funnel shifts, so HBM sees each word once per shift
// BLOCK (256 shifts), not once per shift per statement.
//
// Mapping: blockIdx.y selects 256 consecutive shifts, lane <-> one shift d.  A wavefront owns 64
// consecutive shifts aligned to 64, so (d >> 6) is wave-uniform and the R tile reads broadcast.
//   Rs_d[i]  = bits [64 i + d, 64 i + d + 63] of R          (R.rshift(1) per shift, mscc.pyx:316)
//   M2_d[i]  = bits [64 i + c - d, ... + 63] of M, c = L-1  (the RM clone: mscc.pyx:279-282,307-310)
//   D_d      = M & M2_d                                      (mscc.pyx:291)
// and the four per-shift sums of mscc.pyx:300-305,314 are popcounts of ANDs of those words.
// popcount(D_d) (mappable_len, mscc.pyx:292-298) equals the autocorrelation of M at lag |c - d|, so it
// is produced by the same kernel run as autocorr (F = R = M) and mapped by k_mlen_map.
#include "pmx_common.h"

#define DENSE_TW 512      // 64-bit words of F per tile (32 Kbit, 4 KiB per vector)
#define DENSE_PAD 8       // halo words staged beyond the tile

__device__ __forceinline__ u64 ld_word(const u64 *__restrict__ p, int64_t idx, uint64_t nwords, uint64_t nbits)
{
    if (idx < 0 || (uint64_t)idx >= nwords) return 0;
    u64 w = p[idx];
    if ((uint64_t)idx == nwords - 1 && (nbits & 63)) w &= ~0ull >> (64 - (nbits & 63));
    return w;
}

// bits [s, s+63] of the 128-bit value hi:lo, s in [0, 63]
__device__ __forceinline__ u64 funnel64(u64 lo, u64 hi, u32 s)
{
    return (lo >> s) | ((hi << 1) << (63 - s));
}

__device__ __forceinline__ int64_t floordiv64(int64_t a)
{
    return a >> 6;   // arithmetic shift: floor for negatives
}

template <bool HAS_M, bool DO_NCC>
__global__ void __launch_bounds__(256) k_cc_dense(const u64 *__restrict__ F, const u64 *__restrict__ R,
                                                  const u64 *__restrict__ M, uint64_t nbits, uint64_t nwords,
                                                  u32 max_shift, int32_t c, u32 ntiles, u64 *__restrict__ out,
                                                  u32 out_stride, const u64 *__restrict__ select, u32 select_mode,
                                                  u32 *__restrict__ slab)
{
    // select_mode != 0: mark the path taken in the scalar row of the result block
    if (select_mode && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        out[(size_t)PMX_ROW_SCALARS * out_stride + 3] = PMX_PATH_DENSE;

    __shared__ u64 sF[DENSE_TW];
    __shared__ u64 sM[HAS_M ? DENSE_TW : 1];
    __shared__ u64 sR[DENSE_TW + DENSE_PAD];
    __shared__ u64 sM2[HAS_M ? DENSE_TW + DENSE_PAD : 1];

    const u32 tid = threadIdx.x;
    const u32 d0 = blockIdx.y * 256u;
    const u32 d = d0 + tid;
    const u32 dq = d0 >> 6;
    const u32 r_wo = (d >> 6) - dq;
    const u32 r_sh = d & 63;

    u32 a_ncc = 0, a_fs = 0, a_rs = 0, a_mc = 0;

    for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t w0 = (int64_t)t * DENSE_TW;
        const int64_t B0 = 64 * w0 + c - (int64_t)d0 - 255;
        const int64_t wb = floordiv64(B0);
        __syncthreads();
        for (u32 i = tid; i < DENSE_TW; i += 256) {
            sF[i] = ld_word(F, w0 + i, nwords, nbits);
            if (HAS_M) sM[i] = ld_word(M, w0 + i, nwords, nbits);
        }
        for (u32 i = tid; i < DENSE_TW + DENSE_PAD; i += 256) {
            sR[i] = ld_word(R, w0 + dq + i, nwords, nbits);
            if (HAS_M) sM2[i] = ld_word(M, wb + i, nwords, nbits);
        }
        __syncthreads();

        const int64_t moff = 64 * w0 + c - (int64_t)d - 64 * wb;   // >= 0, <= 318
        const u32 m_wo = (u32)(moff >> 6);
        const u32 m_sh = (u32)(moff & 63);

        u64 r_lo = sR[r_wo];
        u64 m_lo = HAS_M ? sM2[m_wo] : 0;
#pragma unroll 4
        for (u32 i = 0; i < DENSE_TW; i++) {
            const u64 f = sF[i];
            const u64 r_hi = sR[i + r_wo + 1];
            const u64 rs = funnel64(r_lo, r_hi, r_sh);
            r_lo = r_hi;
            if (DO_NCC) a_ncc += (u32)__popcll(f & rs);
            if (HAS_M) {
                const u64 m_hi = sM2[i + m_wo + 1];
                const u64 D = sM[i] & funnel64(m_lo, m_hi, m_sh);
                m_lo = m_hi;
                const u64 fd = f & D;
                a_fs += (u32)__popcll(fd);
                a_rs += (u32)__popcll(rs & D);
                a_mc += (u32)__popcll(fd & rs);
            }
        }
    }

    // per-workgroup partial sums -> this workgroup's slab segment [4 rows][gridDim.y * 256 shifts] (no atomics:
    // k_reduce_dense sums the gridDim.x segments of every shift block)
    const u32 ncols = gridDim.y * 256u;
    u32 *seg = slab + (size_t)blockIdx.x * 4 * ncols + d;
    seg[0 * (size_t)ncols] = a_ncc;
    if (HAS_M) {
        seg[1 * (size_t)ncols] = a_fs;
        seg[2 * (size_t)ncols] = a_rs;
        seg[3 * (size_t)ncols] = a_mc;
    }
}

// out[row][d] = sum over the gx workgroup segments; rows: NCC_CCBINS, MSCC_FSUM, MSCC_RSUM, MSCC_CCBINS (= 0..3)
__global__ void __launch_bounds__(256)
k_reduce_dense(const u32 *__restrict__ slab, u32 gx, u32 ncols, u32 max_shift, u32 row_mask, u64 *__restrict__ out,
               u32 out_stride)
{
    const u32 d = blockIdx.x * 256u + threadIdx.x, row = blockIdx.y;
    if (d > max_shift || !((row_mask >> row) & 1u)) return;
    u64 sum = 0;
    const u32 *p = slab + (size_t)row * ncols + d;
    for (u32 x = 0; x < gx; x++) sum += p[(size_t)x * 4 * ncols];
    out[(size_t)row * out_stride + d] = sum;
}

__global__ void k_mlen_map(const u64 *__restrict__ autocorr, u32 max_shift, int32_t c, u64 *__restrict__ mlen)
{
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > max_shift) return;
    const int32_t k = c - (int32_t)d;
    mlen[d] = autocorr[k < 0 ? -k : k];
}

static void dense_grid(pmx_ctx *ctx, uint64_t nwords, u32 max_shift, u32 *ntiles, dim3 *grid)
{
    uint64_t nt = (nwords + DENSE_TW - 1) / DENSE_TW;
    if (nt < 1) nt = 1;
    const u32 gy = max_shift / 256 + 1;
    // enough workgroups to fill 256 CUs x 8 resident blocks, but every block keeps < 2^31 bits of work
    uint64_t gx = ((uint64_t)ctx->num_cus * 8 + gy - 1) / gy;
    if (gx > nt) gx = nt;
    const uint64_t min_gx = (nwords * 64) / (1ull << 31) + 1;
    if (gx < min_gx) gx = min_gx;
    *ntiles = (u32)nt;
    *grid = dim3((u32)gx, gy, 1);
}

int pmx_launch_cc_dense(pmx_ctx *ctx, const uint64_t *d_F, const uint64_t *d_R, const uint64_t *d_M,
                        uint64_t nbits, uint32_t max_shift, uint32_t read_len, bool do_ncc,
                        u64 *d_out, uint32_t out_stride, const u64 *d_select, uint32_t select_mode)
{
    const uint64_t nwords = (nbits + 63) / 64;
    u32 ntiles;
    dim3 grid;
    dense_grid(ctx, nwords, max_shift, &ntiles, &grid);
    const int32_t c = (int32_t)read_len - 1;
    const u64 *F = (const u64 *)d_F, *R = (const u64 *)d_R, *M = (const u64 *)d_M;
    if (!d_M && !do_ncc) return PMX_OK;
    const u32 ncols = grid.y * 256u;
    int rc = pmx_ensure_slab(ctx, (size_t)grid.x * 4 * ncols);
    if (rc) return rc;
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_CC_DENSE, &tl);
    if (rc) return rc;
    if (d_M && do_ncc)
        hipLaunchKernelGGL((k_cc_dense<true, true>), grid, dim3(256), 0, ctx->stream, F, R, M, nbits, nwords,
                           max_shift, c, ntiles, d_out, out_stride, d_select, select_mode, ctx->d_slab);
    else if (d_M)
        hipLaunchKernelGGL((k_cc_dense<true, false>), grid, dim3(256), 0, ctx->stream, F, R, M, nbits, nwords,
                           max_shift, c, ntiles, d_out, out_stride, d_select, select_mode, ctx->d_slab);
    else
        hipLaunchKernelGGL((k_cc_dense<false, true>), grid, dim3(256), 0, ctx->stream, F, R, M, nbits, nwords,
                           max_shift, c, ntiles, d_out, out_stride, d_select, select_mode, ctx->d_slab);
    PMX_CHECK_LAUNCH("k_cc_dense");
    rc = pmx_prof_end(ctx, &tl);
    if (rc) return rc;
    const u32 row_mask = (do_ncc ? 1u : 0u) | (d_M ? 14u : 0u);
    hipLaunchKernelGGL(k_reduce_dense, dim3(grid.y, 4), dim3(256), 0, ctx->stream, (const u32 *)ctx->d_slab, grid.x, ncols,
                       max_shift, row_mask, d_out, out_stride);
    PMX_CHECK_LAUNCH("k_reduce_dense");
    return PMX_OK;
}

int pmx_launch_autocorr_dense(pmx_ctx *ctx, const uint64_t *d_M, uint64_t nbits, uint32_t max_lag, u64 *d_out)
{
    const uint64_t nwords = (nbits + 63) / 64;
    u32 ntiles;
    dim3 grid;
    dense_grid(ctx, nwords, max_lag, &ntiles, &grid);
    const u64 *M = (const u64 *)d_M;
    const u32 ncols = grid.y * 256u;
    int rc = pmx_ensure_slab(ctx, (size_t)grid.x * 4 * ncols);
    if (rc) return rc;
    pmx_timed_launch tl;
    rc = pmx_prof_begin(ctx, PMX_KERNEL_AUTOCORR, &tl);
    if (rc) return rc;
    // row 0 of the slab segments: out[k] = sum_j M[j] & M[j+k]
    hipLaunchKernelGGL((k_cc_dense<false, true>), grid, dim3(256), 0, ctx->stream, M, M, (const u64 *)nullptr,
                       nbits, nwords, max_lag, 0, ntiles, d_out, max_lag + 1, (const u64 *)nullptr, 0u, ctx->d_slab);
    PMX_CHECK_LAUNCH("k_cc_dense(autocorr)");
    rc = pmx_prof_end(ctx, &tl);
    if (rc) return rc;
    hipLaunchKernelGGL(k_reduce_dense, dim3(grid.y, 1), dim3(256), 0, ctx->stream, (const u32 *)ctx->d_slab, grid.x, ncols,
                       max_lag, 1u, d_out, max_lag + 1);
    PMX_CHECK_LAUNCH("k_reduce_dense");
    return PMX_OK;
}

int pmx_launch_mlen_map(pmx_ctx *ctx, const u64 *d_autocorr, uint32_t max_shift, uint32_t read_len, u64 *d_mlen)
{
    hipLaunchKernelGGL(k_mlen_map, dim3(max_shift / 256 + 1), dim3(256), 0, ctx->stream, d_autocorr, max_shift,
                       (int32_t)read_len - 1, d_mlen);
    PMX_CHECK_LAUNCH("k_mlen_map");
    return PMX_OK;
}
