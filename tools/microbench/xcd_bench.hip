// Microbenchmark (diagnostic, not part of the product):  hipcc --offload-arch=gfx950 -O3 -o xcd_bench xcd_bench.hip
// MI355X, us per kernel (read + write of the given bytes): 1 MB 1.97 same XCD / 2.49 other XCD; 4 MB 2.70 / 4.73; 16 MB 7.96 / 9.41; 64 MB 26.4 / 27.8.
// The per-XCD L2 keeps its lines across kernel boundaries: a consumer on the producer's XCD reads them there, any other XCD goes through the fabric
// (+2 us at 4 MB).  This is what gemm_panel / row_affine (api.hip) use.
// Chain of dependent kernels in a replayed graph; kernel i+1's workgroup b reads the 16 KB that workgroup (b + shift) % G of kernel i wrote.
// shift 0 = same XCD (and CU slot), shift 8 = other CU of the same XCD, shift 1 = the next XCD.  Rows: bytes per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(256) void k_hop(const float4* in, float4* out, int shift, int per_wg4) {
    const int G = gridDim.x;
    const int src = (blockIdx.x + shift) % G;
    float4 acc[8];
    for (int base = 0; base < per_wg4; base += 256 * 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { int i = base + j * 256 + threadIdx.x; acc[j] = i < per_wg4 ? in[(long)src * per_wg4 + i] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { int i = base + j * 256 + threadIdx.x; if (i < per_wg4) { acc[j].x += 1.f; out[(long)blockIdx.x * per_wg4 + i] = acc[j]; } }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int G = 256, CH = 200, REP = 20;
    const int sizes4[] = {256, 1024, 4096, 16384};   // float4 per workgroup: 4 KB, 16 KB, 64 KB, 256 KB
    float4 *a, *b;
    const size_t bytes = (size_t)G * 16384 * 16;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int per : sizes4) for (int shift : {0, 8, 1, 3}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_hop, dim3(G), dim3(256), 0, st, (const float4*)((i & 1) ? b : a), (i & 1) ? a : b, shift, per);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double best = 1e9;
        for (int r = 0; r < REP; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("%4d KB per workgroup (%5.1f MB per kernel), shift %d: %.3f us per kernel\n", per * 16 / 1024, G * per * 16 / 1048576.0, shift, best / CH);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
