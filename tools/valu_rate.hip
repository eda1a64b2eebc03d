// microbenchmark (GPU box): VALU issue rate of the ops the sparse kernel lives on, vs waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters)
{
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a ^ 9, g = a + 11, h = a * 13;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {   // 8 independent bitop3
                a = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0xE8);
                c = __builtin_amdgcn_bitop3_b32(c, d, e, 0x96); d = __builtin_amdgcn_bitop3_b32(d, e, f, 0xE8);
                e = __builtin_amdgcn_bitop3_b32(e, f, g, 0x96); f = __builtin_amdgcn_bitop3_b32(f, g, h, 0xE8);
                g = __builtin_amdgcn_bitop3_b32(g, h, a, 0x96); h = __builtin_amdgcn_bitop3_b32(h, a, b, 0xE8);
            } else if (OP == 1) {   // alignbit
                a = __builtin_amdgcn_alignbit(a, b, c); b = __builtin_amdgcn_alignbit(b, c, d);
                c = __builtin_amdgcn_alignbit(c, d, e); d = __builtin_amdgcn_alignbit(d, e, f);
                e = __builtin_amdgcn_alignbit(e, f, g); f = __builtin_amdgcn_alignbit(f, g, h);
                g = __builtin_amdgcn_alignbit(g, h, a); h = __builtin_amdgcn_alignbit(h, a, b);
            } else {   // plain add / xor
                a += b; b ^= c; c += d; d ^= e; e += f; f ^= g; g += h; h ^= a;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int OP> void run(const char *name)
{
    unsigned *d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {   // 4 waves per WG -> wg_per_cu waves per SIMD
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 64 * wg_per_cu;   // wave-instructions per SIMD
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, wg_per_cu, ms,
               ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    hipFree(d);
}
int main() { run<0>("bitop3"); run<1>("alignbit"); run<2>("add/xor"); return 0; }
