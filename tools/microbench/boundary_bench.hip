// Microbenchmark (diagnostic, not part of the product):  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o boundary_bench boundary_bench.hip
// Per-kernel time of a chain of 300 dependent tiny kernels in a replayed hipGraph: struct kernarg read with s_load vs flat arguments
// preloaded into SGPRs vs an empty kernel.  MI355X: 1.76 / 1.88 / 1.58 us per kernel -- a dependent kernel that reads and writes memory costs
// 1.8 us all in, reading its arguments costs nothing measurable, kernarg preloading does not help.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct Args { const float* in; float* out; int n; int m; long s; int pad[100]; };
__global__ void k_struct(Args x) { int i = blockIdx.x * 256 + threadIdx.x; if (i < x.n) x.out[i] = x.in[i] + (float)x.s + (float)x.pad[x.m]; }
__global__ void k_flat(const float* in, float* out, int n, int m, long s) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = in[i] + (float)s + (float)m; }
__global__ void k_empty() {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 256 * 256, CH = 300, REP = 30;
    float *a, *b;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CH; ++i) {
            float* src = (i & 1) ? b : a; float* dst = (i & 1) ? a : b;
            if (variant == 0) { Args x{}; x.in = src; x.out = dst; x.n = N; x.m = 3; x.s = 1; hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, st, x); }
            else if (variant == 1) hipLaunchKernelGGL(k_flat, dim3(256), dim3(256), 0, st, (const float*)src, dst, N, 3, 1L);
            else hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double best = 1e9;
        for (int r = 0; r < REP; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("variant %d (%s): %.3f us per kernel (chain of %d)\n", variant, variant == 0 ? "struct kernarg, s_load" : variant == 1 ? "flat args, preloaded" : "empty", best / CH, CH);
    }
    return 0;
}
