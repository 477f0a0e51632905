// Flash-style attention for the DiT block: softmax(q k^T / sqrt(dh) [+ key mask]) v
// (F.scaled_dot_product_attention call of the reference, src/models/utils/attention.py:106-110; the boolean
// key mask of cross-attention is built at attention.py:30-37,131-135.)
//
// gfx950 design (v_mfma_f32_32x32x16_bf16 everywhere, fp32 softmax state):
//   * one wave owns 32 query rows; 4 waves per workgroup (128 query rows), waves are independent.
//   * scores are computed TRANSPOSED: S^T[key, q] = K . Q^T, so in the 32x32 C layout a lane owns ONE query
//     column (q = lane & 31) and 16 of the 32 keys of the tile.  The softmax row reductions are then
//     in-lane plus a single exchange with lane ^ 32 -- no LDS, no 5-step butterflies.
//   * the output is accumulated transposed as well: O^T[d, q] = V^T . P^T, so the online-softmax rescale
//     factor (per q) is lane-local for the accumulators, and P^T (B operand: lane = q, 8 keys per lane) is
//     exactly the S^T registers converted to bf16 -- P never leaves registers.  The MFMA contraction order
//     over keys is permuted accordingly (k-slot (hi, j) <-> key 16*step + 8*(j>>2) + 4*hi + (j&3)); the
//     producer stores V TRANSPOSED ([dh][keys], keys contiguous) so the matching A fragment is two 8-byte loads.
//   * head_dim 72 (EzAudio-XL) is zero padded to 80 for the QK^T contraction (5 k-steps of 16) and to 96
//     output rows (3 tiles of 32) for P.V; head_dim 64 needs no padding.
//   * L = 500 keys: K/V of one head are 80 KB each and L2/L1 resident, so fragments are read straight
//     from global memory (no LDS staging, no barriers).
#include "common.h"

namespace {

template <int DH>
struct HeadGeom;
template <>
struct HeadGeom<64> { static constexpr int DQK = 64, DV = 64; };
template <>
struct HeadGeom<72> { static constexpr int DQK = 80, DV = 96; };

template <int DH>
__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
    constexpr int DQK = HeadGeom<DH>::DQK;
    constexpr int DV = HeadGeom<DH>::DV;
    constexpr int NKS = DQK / 16;  // k-steps of the QK^T contraction
    constexpr int NDT = DV / 32;   // 32-row tiles of O^T
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    if (q0 >= a.Lq) return;
    const long bh = (long)b * a.H + h;

    const bf16_t* Q = a.q + (bh * a.Lqp + q0 + r32) * DQK + 8 * hi;
    const bf16_t* K = a.k + (bh * a.Lkp + r32) * DQK + 8 * hi;
    const bf16_t* VT = a.vt + (bh * DV + r32) * (long)a.Lkp + 4 * hi;
    const uint8_t* km = a.kmask ? a.kmask + (long)b * a.Lk : nullptr;

    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + 16 * ks);

    f32x16 o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -1e30f, lsum = 0.f;
    const float c = 1.4426950408889634f * rsqrtf((float)DH);  // log2(e) / sqrt(dh)

    const int ntiles = (a.Lk + 31) / 32;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(K + (long)key0 * DQK + 16 * ks);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        // lane holds S^T[key0 + (r&3) + 8*(r>>2) + 4*hi][q0 + r32]
        float tmax = -1e30f;
        bool valid[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            bool v = key < a.Lk;
            if (km) v = v && (km[key < a.Lk ? key : 0] != 0);
            valid[r] = v;
            s[r] = v ? s[r] * c : -1e30f;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = exp2f(m - mnew);
        m = mnew;
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = valid[r] ? exp2f(s[r] - mnew) : 0.f;
            psum += p[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf2(p[8 * step + 2 * e], p[8 * step + 2 * e + 1]);
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                // A fragment: V^T[32t + r32][key0 + 16*step + 4*hi + {0..3}] and [... + 8 + {0..3}]
                const bf16_t* vp = VT + (long)(32 * t) * a.Lkp + key0 + 16 * step;
                union { bf16x8 v; uint2 h2[2]; } vf;
                vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
                vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 8);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[t], 0, 0, 0);
            }
        }
    }
    // lane holds O^T[d = 32t + (r&3) + 8*(r>>2) + 4*hi][q = q0 + r32]
    const int qrow = q0 + r32;
    if (qrow >= a.Lq) return;
    const float inv = 1.f / lsum;
    bf16_t* orow = a.out + ((long)b * a.Lq + qrow) * a.ldo + h * DH;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * t + 8 * g + 4 * hi;
            if (d < DH) {
                uint2 v;
                v.x = pack_bf2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
                v.y = pack_bf2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + d) = v;
            }
        }
}

}  // namespace

void launch_attention(const AttnArgs& a, hipStream_t st) {
    dim3 grid((a.Lq + 127) / 128, a.H, a.B);
    if (a.dh == 64) hipLaunchKernelGGL((k_attn<64>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_attn<72>), grid, dim3(256), 0, st, a);
}
