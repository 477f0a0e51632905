// microbenchmark: per-CU global->LDS (LDS-DMA) and global->VGPR ingest rate from an L2-RESIDENT window (1 MB per XCD),
// as a function of waves per workgroup, workgroups per CU and loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int INFLIGHT, int NT>
__global__ __launch_bounds__(NT) void k(const char* __restrict__ src, size_t win, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const char* base = src + (size_t)(blockIdx.x & 7) * (2u << 20);   // XCD-private window
    size_t off = ((size_t)(blockIdx.x >> 3) * 40960) % win;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const char* p = base + (off + (size_t)j * (NT * 16) + tid * 16) % win;
                char* dst = lds + ((j * (NT / 64) + wave) * 1024);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j)
                v[j] = *reinterpret_cast<const float4*>(base + (off + (size_t)j * (NT * 16) + tid * 16) % win);
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) acc += v[j].x;
        }
        off = (off + (size_t)INFLIGHT * NT * 16) % win;
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <int MODE, int INFLIGHT, int NT>
void run(const char* d, size_t win, float* sink, int blocks_per_cu) {
    const int iters = 400;
    dim3 grid(256 * blocks_per_cu);
    const int smem = INFLIGHT * NT * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE, INFLIGHT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, INFLIGHT, NT>), grid, dim3(NT), smem, 0, d, win, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, INFLIGHT, NT>), grid, dim3(NT), smem, 0, d, win, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tot = (double)grid.x * iters * INFLIGHT * NT * 16.0;
    printf("%s waves/wg %2d wg/CU %d inflight/wave %2d (%3d KB/CU) win %4zu KB: %6.2f TB/s  %6.1f GB/s per CU\n", MODE ? "vgpr" : "glds", NT / 64,
           blocks_per_cu, INFLIGHT, blocks_per_cu * smem / 1024, win >> 10, tot / ms / 1e9, tot / ms / 1e6 / 256);
}
int main() {
    char* d; hipMalloc(&d, 32u << 20); hipMemset(d, 1, 32u << 20);
    float* sink; hipMalloc(&sink, 4);
    for (size_t win : {(size_t)1 << 20, (size_t)256 << 10}) {
        run<0, 2, 256>(d, win, sink, 1); run<0, 6, 256>(d, win, sink, 1); run<0, 12, 256>(d, win, sink, 1); run<0, 24, 256>(d, win, sink, 1);
        run<0, 6, 256>(d, win, sink, 2); run<0, 12, 256>(d, win, sink, 2); run<0, 6, 256>(d, win, sink, 4);
        run<0, 3, 512>(d, win, sink, 1); run<0, 6, 512>(d, win, sink, 1); run<0, 12, 512>(d, win, sink, 1);
        run<0, 3, 1024>(d, win, sink, 1); run<0, 6, 1024>(d, win, sink, 1);
        run<1, 6, 256>(d, win, sink, 1); run<1, 12, 256>(d, win, sink, 1); run<1, 6, 256>(d, win, sink, 2); run<1, 6, 512>(d, win, sink, 1);
        run<1, 6, 1024>(d, win, sink, 1); run<1, 12, 512>(d, win, sink, 1);
    }
    return 0;
}
