// microbenchmark (GPU box): issue cost of the 64-bit integer ops in the event kernel's emit loops vs their 32-bit counterparts
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters)
{
    unsigned a = threadIdx.x + 1, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
    u64 x = ((u64)a << 32) | b, y = ((u64)c << 32) | d, z = x ^ 0x123456789abcdefull, w = y + 77;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {          // 4 independent 64-bit variable left shifts
                x = x << (a & 63); y = y << (b & 63); z = z << (c & 63); w = w << (d & 63);
                x |= 1; y |= 3; z |= 5; w |= 7;             // (+4 x 2 32-bit ors, subtract below)
            } else if (OP == 1) {   // 4 independent 64-bit adds (v_lshl_add_u64)
                x += y; y += z; z += w; w += x;
            } else if (OP == 2) {   // ffbl + bcnt pairs (32-bit)
                a = __builtin_ctz(b | 1) + __popc(c); b = __builtin_ctz(c | 1) + __popc(d);
                c = __builtin_ctz(d | 1) + __popc(a); d = __builtin_ctz(a | 1) + __popc(b);
            } else if (OP == 3) {   // 32-bit variable shifts
                a = a << (b & 31); b = b >> (c & 31); c = c << (d & 31); d = d >> (a & 31);
                a |= 1; b |= 3; c |= 5; d |= 7;
            } else {                // 32-bit or only (baseline for the ors above)
                a |= b; b |= c; c |= d; d |= a; a ^= 1; b ^= 3; c ^= 5; d ^= 7;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)(x ^ y ^ z ^ w) ^ (unsigned)((x ^ y ^ z ^ w) >> 32);
}
template <int OP> void run(const char *name, int ops_per_iter)
{
    unsigned *d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 4) {
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 8 * wg_per_cu;   // unrolled groups per SIMD
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.1f cycles per group of %d source ops per SIMD\n", name, wg_per_cu, ms,
               ms * 1e-3 * 2.4e9 / per_simd, ops_per_iter);
    }
    hipFree(d);
}
int main()
{
    run<0>("4x shl64(var) + 8 or32", 12); run<1>("4x add64", 4); run<2>("4x (ctz32 + popc32 + add)", 12);
    run<3>("4x shift32(var) + 4 or32", 8); run<4>("8x or/xor32", 8);
    return 0;
}
