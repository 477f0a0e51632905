// bf16 MFMA GEMM for the DiT projections:  C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear layout, both K-contiguous)
//
// Restates the `aten::linear` calls of the reference block (src/models/utils/attention.py:127-129,148;
// src/models/utils/modules.py:263-277,341-374; src/models/blocks.py:124-128) as one kernel family.
//
// gfx950 design:
//   * v_mfma_f32_32x32x16_bf16, 4 waves (2x2) per workgroup, wave tile (BM/2)x(BN/2), fp32 accumulate
//   * BK = 64: one LDS row = 128 B = 8 chunks of 16 B.  Tiles are staged with global_load_lds (16 B per
//     lane, no VGPR round trip).  The LDS image is lane-linear, so the bank swizzle is applied to the
//     SOURCE address: chunk c of row r is fetched into slot c ^ ((r>>1)&7), and fragment reads apply the
//     same involution.  With 128-B rows two consecutive rows span the 64 banks, hence (r>>1): the 16 lanes
//     of a ds_read_b128 group (rows distinct mod 16) then hit 16 distinct 16-B slots -> conflict free.
//   * double-buffered LDS, one barrier per K tile: tile t+1 streams in while tile t feeds the MFMAs
//   * workgroup -> tile map is XCD aware: the 8 workgroups that the dispatcher puts on one XCD
//     (block b -> XCD b % 8) walk the M tiles of ONE weight panel, so each weight tile is pulled from
//     HBM into exactly one L2.
//   * split-K (blockIdx.z) writes raw fp32 slabs; the row kernel that follows reduces them
//     (launch-boundary reduce, see rowops.hip), so no atomics and bitwise-deterministic results.
#include "common.h"

namespace {

constexpr int BK = 64;

template <int ROWS>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int ld, int row0, int max_row, int k0,
                                           char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 3;
        const int c = q & 7;
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row;
        const int gc = c ^ ((row >> 1) & 7);
        const bf16_t* src = G + (long)grow * ld + k0 + gc * 8;
        char* dst = lds + (i * 256 + (tid & ~63)) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs a) {
    constexpr int TM = BM / 2, TN = BN / 2;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile map
    const int tilesM = (a.M + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7;
    const int idx = blockIdx.x >> 3;
    const int tn = (idx / tilesM) * 8 + xcd;
    const int tm = idx % tilesM;
    if (tn >= tilesN) return;
    const int row0 = tm * BM, col0 = tn * BN;

    const int nk = a.K / BK;
    const int z = blockIdx.z;
    const int kb = nk * z / a.splitk;
    const int ke = nk * (z + 1) / a.splitk;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // stage s: [A tile | B tile] at smem + s * STAGE_BYTES

    if (kb < ke) {
        stage_tile<BM>(a.A, a.lda, row0, a.M - 1, kb * BK, smem, tid);
        stage_tile<BN>(a.W, a.ldw, col0, 0x7fffffff, kb * BK, smem + A_BYTES, tid);
    }
    const int r32 = lane & 31, hi = lane >> 5;
    int cur = 0;
    for (int kt = kb; kt < ke; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ke) {
            char* nxt = smem + (cur ^ 1) * STAGE_BYTES;
            stage_tile<BM>(a.A, a.lda, row0, a.M - 1, (kt + 1) * BK, nxt, tid);
            stage_tile<BN>(a.W, a.ldw, col0, 0x7fffffff, (kt + 1) * BK, nxt + A_BYTES, tid);
        }
        const char* cA = smem + cur * STAGE_BYTES;
        const char* cB = cA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = lds_frag(cA, wm * TM + i * 32 + r32, 2 * ks + hi);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = lds_frag(cB, wn * TN + j * 32 + r32, 2 * ks + hi);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        cur ^= 1;
    }

    // ---- epilogue: C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    if constexpr (EPI == EPI_GEGLU) {
        static_assert(FN % 2 == 0, "GEGLU epilogue pairs value/gate fragments");
        bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; j += 2) {
                const int cv = col0 + wn * TN + j * 32 + r32;  // interleaved column of the value
                const int cg = cv + 32;
                const int oc = (col0 + wn * TN + j * 32) / 2 + r32;  // output (inner) index
                if (cv >= a.N) continue;
                const float bv = a.bias ? a.bias[cv] : 0.f;
                const float bg = a.bias ? a.bias[cg] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < a.M) {
                        const float v = acc[i][j][r] + bv;
                        const float g = acc[i][j + 1][r] + bg;
                        out[(long)row * a.ldo + oc] = f2bf(v * gelu_erf(g));
                    }
                }
            }
    } else {
        float* out = reinterpret_cast<float*>(a.out);
        if constexpr (EPI == EPI_PARTIAL) out += (long)z * a.slab_stride;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int col = col0 + wn * TN + j * 32 + r32;
                if (col >= a.N) continue;
                float bias = 0.f;
                if constexpr (EPI == EPI_F32) bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < a.M) out[(long)row * a.ldo + col] = acc[i][j][r] + bias;
                }
            }
    }
}

template <int BM, int BN, int EPI>
void launch_t(const GemmArgs& a, hipStream_t st) {
    const int tilesM = (a.M + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    dim3 grid(8 * tilesM * ((tilesN + 7) / 8), 1, a.splitk);
    hipLaunchKernelGGL((k_gemm<BM, BN, EPI>), grid, dim3(256), 0, st, a);
}

}  // namespace

void launch_gemm(const GemmArgs& a, hipStream_t st) {
    if (a.epi == EPI_GEGLU) {
        launch_t<128, 128, EPI_GEGLU>(a, st);
    } else if (a.epi == EPI_PARTIAL) {
        if (a.tile == 0) launch_t<128, 128, EPI_PARTIAL>(a, st);
        else launch_t<128, 64, EPI_PARTIAL>(a, st);
    } else {
        if (a.tile == 0) launch_t<128, 128, EPI_F32>(a, st);
        else launch_t<128, 64, EPI_F32>(a, st);
    }
}
