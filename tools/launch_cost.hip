// GPU box microbenchmark: what does a launch that returns at once cost on the stream, by shape?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_cost tools/launch_cost.hip && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
struct Big { unsigned w[704]; };   // 2816 bytes: a job table by value
__global__ void k_small(const unsigned *gate) { if (*gate == 0) return; }
__global__ void k_bigarg(const Big b, const unsigned *gate) { if (*gate == 0) return; if (b.w[threadIdx.x] == 7) __builtin_trap(); }
__global__ void __launch_bounds__(256) k_lds(const unsigned *gate, unsigned *out)
{
    __shared__ unsigned s[9000];
    if (*gate == 0) return;
    s[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = s[255 - threadIdx.x];
}
__global__ void __launch_bounds__(1024) k_1024(const unsigned *gate) { if (*gate == 0) return; }
__global__ void k_nogate() {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    unsigned *gate, *out;
    CK(hipMalloc(&gate, 64)); CK(hipMalloc(&out, 4096)); CK(hipMemset(gate, 0, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    Big big; memset(&big, 0, sizeof big);
    const int N = 200;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; i++) launch();
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < N; i++) launch();
        hipEventRecord(b, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %7.2f us per launch\n", name, 1e3 * ms / N);
        return 0;
    };
    run("nogate 1 x 64", [&] { hipLaunchKernelGGL(k_nogate, dim3(1), dim3(64), 0, st); });
    run("nogate 1280 x 256", [&] { hipLaunchKernelGGL(k_nogate, dim3(1280), dim3(256), 0, st); });
    run("gate small 1 x 256", [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, st, gate); });
    run("gate small 1280 x 256", [&] { hipLaunchKernelGGL(k_small, dim3(1280), dim3(256), 0, st, gate); });
    run("gate small 768 x 256", [&] { hipLaunchKernelGGL(k_small, dim3(768), dim3(256), 0, st, gate); });
    run("gate 2.8 KB kernarg 1280 x 256", [&] { hipLaunchKernelGGL(k_bigarg, dim3(1280), dim3(256), 0, st, big, gate); });
    run("gate 2.8 KB kernarg 2 x 256", [&] { hipLaunchKernelGGL(k_bigarg, dim3(2), dim3(256), 0, st, big, gate); });
    run("gate 36 KB LDS 768 x 256", [&] { hipLaunchKernelGGL(k_lds, dim3(768), dim3(256), 0, st, gate, out); });
    run("gate 2 x 1024", [&] { hipLaunchKernelGGL(k_1024, dim3(2), dim3(1024), 0, st, gate); });
    run("gate 96 x 1024", [&] { hipLaunchKernelGGL(k_1024, dim3(96), dim3(1024), 0, st, gate); });
    run("memsetAsync 100 KB", [&] { hipMemsetAsync(out, 0, 4096, st); });
    return 0;
}
