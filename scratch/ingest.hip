// microbenchmark: per-CU global->LDS / global->VGPR ingest rate from an L2-resident buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    // each block streams its own window (L2 resident: windows wrap inside `bytes`)
    size_t base = ((size_t)blockIdx.x * 65536) % bytes;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) {
                const char* p = src + (base + (size_t)(it * INFLIGHT + j) * 4096 + tid * 16) % bytes;
                char* dst = lds + (j * 256 + (tid & ~63)) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j)
                v[j] = *reinterpret_cast<const float4*>(src + (base + (size_t)(it * INFLIGHT + j) * 4096 + tid * 16) % bytes);
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) acc += v[j].x;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <int MODE, int INFLIGHT>
void run(const char* d, size_t bytes, float* sink, int blocks_per_cu) {
    const int iters = 200;
    dim3 grid(256 * blocks_per_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, INFLIGHT * 4096);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, INFLIGHT>), grid, dim3(256), INFLIGHT * 4096, 0, d, bytes, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, INFLIGHT>), grid, dim3(256), INFLIGHT * 4096, 0, d, bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tot = (double)grid.x * iters * INFLIGHT * 4096.0;
    printf("mode %s inflight %2d x4KB/blk, %d blk/CU, buf %4zu MB: %7.2f TB/s  (%6.1f GB/s per CU)\n", MODE ? "vgpr" : "glds", INFLIGHT,
           blocks_per_cu, bytes >> 20, tot / ms / 1e9, tot / ms / 1e6 / 256);
}
int main() {
    for (size_t mb : {16, 2048}) {
        size_t bytes = mb << 20;
        char* d; hipMalloc(&d, bytes + (1 << 20)); hipMemset(d, 1, bytes + (1 << 20));
        float* sink; hipMalloc(&sink, 4);
        for (int bpc : {1, 2, 4}) {
            run<0, 2>(d, bytes, sink, bpc); run<0, 6>(d, bytes, sink, bpc); run<0, 12>(d, bytes, sink, bpc);
            run<1, 2>(d, bytes, sink, bpc); run<1, 6>(d, bytes, sink, bpc); run<1, 12>(d, bytes, sink, bpc);
        }
        hipFree(d);
    }
    return 0;
}
