// Microbenchmark (diagnostic, not part of the product):  hipcc --offload-arch=gfx950 -O3 -o ingest_bench ingest_bench.hip
// MI355X: 125-147 GB/s per CU (32-37 TB/s chip) with LDS-DMA, 90-137 with register staging, for 4-12 waves and 0.5-8 MB regions per XCD: one
// 1-KB LDS-DMA piece per ~17 cycles per CU (64 B/clk).  The GEMM K loops ingest 41-65 GB/s per CU: they are not bound by this path.
// Per-CU global -> LDS ingest rate from L2-resident data, LDS-DMA (global_load_lds 16 B) vs register staging
// (global_load_dwordx4 + ds_write_b128), one workgroup per CU, NW waves, each wave keeps DEPTH 1-KB loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void k_ingest(const char* __restrict__ src, size_t region, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* base = src + (size_t)(blockIdx.x & 7) * region;   // one region per XCD, shared by its workgroups
    size_t off = ((size_t)(blockIdx.x >> 3) * nw + wave_u) * 1024 * DEPTH % (region - DEPTH * 1024);
    char* lds = smem + wave_u * DEPTH * 1024;
    float acc = 0.f;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + d * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(lds + d * 1024), 16, 0, 0);
            }
            off += (size_t)nw * 32 * DEPTH * 1024; if (off + DEPTH * 1024 > region) off = (size_t)wave_u * DEPTH * 1024;
            if (it >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");   // previous batch landed
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<float*>(lds + lane * 4);
    } else {
        uint4 r[DEPTH];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = *reinterpret_cast<const uint4*>(base + off + d * 1024 + lane * 16);
            off += (size_t)nw * 32 * DEPTH * 1024; if (off + DEPTH * 1024 > region) off = (size_t)wave_u * DEPTH * 1024;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<uint4*>(lds + d * 1024 + lane * 16) = r[d];
        }
        acc = *reinterpret_cast<float*>(lds + lane * 4);
    }
    if (acc == 123.456f) sink[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE, int DEPTH>
int run(const char* src, size_t region, int nw, float* sink, hipStream_t st) {
    const int iters = 400;
    const size_t smem = (size_t)nw * DEPTH * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ingest<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_ingest<MODE, DEPTH>), dim3(256), dim3(64 * nw), smem, st, src, region, iters, sink);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_ingest<MODE, DEPTH>), dim3(256), dim3(64 * nw), smem, st, src, region, iters, sink);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * nw * DEPTH * 1024.0 * iters;
    printf("%s depth %d, %2d waves, region %4zu KB/XCD: %6.1f GB/s per CU (%5.2f TB/s chip), %.1f us\n", MODE == 0 ? "LDS-DMA " : "reg+ds_w", DEPTH, nw, region >> 10,
           bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
    return 0;
}
int main() {
    char* src; float* sink;
    const size_t total = (size_t)64 << 20;
    CK(hipMalloc(&src, total + (1 << 20))); CK(hipMemset(src, 1, total + (1 << 20))); CK(hipMalloc(&sink, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (size_t region : {(size_t)512 << 10, (size_t)2 << 20, (size_t)8 << 20}) {
        for (int nw : {4, 8, 12}) {
            if (run<0, 4>(src, region, nw, sink, st)) return 1;
            if (run<0, 8>(src, region, nw, sink, st)) return 1;
            if (run<1, 4>(src, region, nw, sink, st)) return 1;
            if (run<1, 8>(src, region, nw, sink, st)) return 1;
        }
    }
    return 0;
}
