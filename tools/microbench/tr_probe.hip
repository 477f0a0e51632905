// What does ds_read_b64_tr_b16 deliver?  LDS word i (16 bit) holds the value i; lane l reads from byte address 8 l (the lane-linear image), so the
// 16-bit value v it receives in element e tells which (lane, element) of the un-transposed access it came from: lane v / 4, element v % 4.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/tr_probe.hip -o tools/_run/tr_probe && tools/_run/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(uint16_t* out, int stride_bytes) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)lds + threadIdx.x * stride_bytes;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v[0] & 0xffff; out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xffff; out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32}) {
        k<<<1, 64>>>(d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane: lane -> 4 x (source lane, source element)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int e = 0; e < 4; ++e) { const int w = h[l * 4 + e]; const int byte = w * 2; printf("  (L%2d,e%d)", byte / stride, (byte % stride) / 2); }
            printf("\n");
        }
    }
    return 0;
}
