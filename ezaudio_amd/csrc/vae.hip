// Oobleck VAE decoder building blocks (reference:
// src/modules/stable_vae/models/autoencoders.py:38-61,82-113,149-190, models/blocks.py:317-358, nn/layers.py:9-14).
//
// Every Conv1d / ConvTranspose1d of the decoder is run on the SAME bf16 MFMA GEMM as the DiT projections:
//   * activations are kept token-major [L][C] (C = GEMM K), bf16, with zero "halo" rows before and after the sequence, so a
//     k-tap (dilated) convolution is one GEMM whose K dimension is taps x C: K tile t reads the activation rows shifted by
//     tap(t) * dilation rows (GemmArgs.conv_*), weights are pre-arranged [Cout][tap][Cin];
//   * ConvTranspose1d(kernel 2s, stride s, padding ceil(s/2)) is one GEMM with K = 2 Cin (x[q], x[q-1]) and N = s * Cout: the
//     output [q][r * Cout + co] IS the up-sampled sequence [(q s + r)][co], read back with a row offset of `padding`;
//   * SnakeBeta (x + sin^2(alpha x) / beta, log-scale parameters) is fused with the fp32 -> bf16 cast that feeds the next conv;
//   * residual adds ride in the GEMM epilogue (fp32).
// The layer sequence itself is host code (ezaudio_amd/vae.py): it runs once per call, not per denoising step.
#include "../../include/ezdit.h"
#include "common.h"
#include <cstring>

namespace {

__global__ __launch_bounds__(256) void k_snake_bf16(const float* __restrict__ x, int ldx, const float* __restrict__ alpha,
                                                    const float* __restrict__ inv_beta, bf16_t* __restrict__ out, int ldo,
                                                    long L, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per 4 channels
    const int c4n = C >> 2;
    if (idx >= L * c4n) return;
    const long l = idx / c4n;
    const int c = (int)(idx % c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + l * ldx + c);
    float r[4] = {v.x, v.y, v.z, v.w};
    if (alpha) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = sinf(r[e] * alpha[c + e]);
            r[e] = r[e] + inv_beta[c + e] * s * s;   // blocks.py:317-318 snake_beta
        }
    }
    uint2 o;
    o.x = pack_bf2(r[0], r[1]);
    o.y = pack_bf2(r[2], r[3]);
    *reinterpret_cast<uint2*>(out + l * ldo + c) = o;
}

// final WNConv1d(C -> 1, k = 7, padding 3, no bias) on a haloed bf16 sequence (3 zero rows each side): one wave per 64 outputs
__global__ __launch_bounds__(256) void k_conv_out1(const bf16_t* __restrict__ xb /* row 0 = position -3 */, int ldx,
                                                   const float* __restrict__ w /* [7][C] */, float* __restrict__ out, long L, int C) {
    const long l = (long)blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k) {
        const bf16_t* xr = xb + (l + k) * ldx;
        const float* wr = w + k * C;
        for (int c = 0; c < C; c += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += __uint_as_float(u[e] << 16) * wr[c + 2 * e];
                acc += __uint_as_float(u[e] & 0xffff0000u) * wr[c + 2 * e + 1];
            }
        }
    }
    out[l] = acc;
}

// encoder input WNConv1d(1 -> C, k = 7, padding 3): wav fp32 [T] -> x fp32 [T][C]; w fp32 [7][C]
__global__ __launch_bounds__(256) void k_conv_in1(const float* __restrict__ wav, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out, long T, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = C >> 2;
    if (idx >= T * c4n) return;
    const long t = idx / c4n;
    const int c = (int)(idx % c4n) * 4;
    float4 acc = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const long j = t + k - 3;
        const float x = (j >= 0 && j < T) ? wav[j] : 0.f;
        const float4 wk = *reinterpret_cast<const float4*>(w + k * C + c);
        acc.x = fmaf(x, wk.x, acc.x); acc.y = fmaf(x, wk.y, acc.y); acc.z = fmaf(x, wk.z, acc.z); acc.w = fmaf(x, wk.w, acc.w);
    }
    *reinterpret_cast<float4*>(out + t * C + c) = acc;
}

// VAE bottleneck (models/bottleneck.py:67-71): enc fp32 [L][2 lat] token-major (mean | scale) + noise [lat][L] -> z [lat][L]
__global__ __launch_bounds__(256) void k_vae_sample(const float* __restrict__ enc, const float* __restrict__ noise,
                                                    float* __restrict__ z, int L, int lat) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L * lat) return;
    const int c = idx / L, l = idx % L;
    const float mean = enc[(long)l * 2 * lat + c];
    const float sc = enc[(long)l * 2 * lat + lat + c];
    const float softplus = sc > 20.f ? sc : log1pf(expf(sc));   // torch softplus, threshold 20
    z[idx] = (noise ? noise[idx] : 0.f) * (softplus + 1e-4f) + mean;
}

int launch_status(const char* what) {   // a failed launch must surface as an error code, not as stale output
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ez_fail(EZDIT_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return EZDIT_OK;
}

}  // namespace

extern "C" {

// out fp32 [M][ldo] = A . W^T (+ bias) (+ resid); conv_cpb / conv_tap_bytes as in GemmArgs.  N multiple of 4.
int ezvae_gemm(const void* A, int lda, const void* W, int ldw, int wrows, const float* bias, const float* resid, int ldr,
               float* out, int ldo, int M, int N, int K, int conv_cpb, long conv_tap_bytes, int tile, ezdit_stream stream) {
    if (K % 64 || N % 4) return ez_fail(EZDIT_E_INVALID, "ezvae_gemm: K=%d must be a multiple of 64 and N=%d of 4", K, N);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.wrows = wrows; g.bias = bias;
    g.out = out; g.ldo = ldo; g.slab_stride = 0; g.M = M; g.N = N; g.K = K; g.splitk = 1; g.epi = EPI_F32; g.tile = tile;
    g.debug = 0; g.conv_cpb = conv_cpb; g.conv_tap_bytes = conv_tap_bytes; g.resid = resid; g.ldr = ldr; g.xcd_map = 1; g.part_bf16 = 0; g.wt = 0; memset(&g.hn, 0, sizeof g.hn);
    g.gate = nullptr; g.gate_slot_stride = 0; g.cur_step = nullptr; g.row_slot = nullptr; g.rows_per_b = 1;
    (void)hipGetLastError();
    if (launch_gemm(g, (hipStream_t)stream)) return ez_fail(EZDIT_E_UNSUPPORTED, "ezvae_gemm: tile %d / shape not supported", tile);
    return launch_status("k_gemm (vae)");
}

int ezvae_snake_bf16(const float* x, int ldx, const float* alpha, const float* inv_beta, void* out, int ldo, long L, int C,
                     ezdit_stream stream) {
    if (C % 4) return ez_fail(EZDIT_E_INVALID, "C=%d must be a multiple of 4", C);
    const long total = L * (C / 4);
    hipLaunchKernelGGL(k_snake_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, alpha, inv_beta,
                       (bf16_t*)out, ldo, L, C);
    return launch_status("k_snake_bf16");
}

int ezvae_conv_out1(const void* xb, int ldx, const float* w, float* out, long L, int C, ezdit_stream stream) {
    if (C % 8) return ez_fail(EZDIT_E_INVALID, "C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(k_conv_out1, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xb, ldx, w, out, L, C);
    return launch_status("k_conv_out1");
}

int ezvae_conv_in1(const float* wav, const float* w, const float* bias, float* out, long T, int C, ezdit_stream stream) {
    if (C % 4) return ez_fail(EZDIT_E_INVALID, "C=%d must be a multiple of 4", C);
    const long total = T * (C / 4);
    hipLaunchKernelGGL(k_conv_in1, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, w, bias, out, T, C);
    return launch_status("k_conv_in1");
}

int ezvae_sample(const float* enc, const float* noise, float* z, int L, int latent_dim, ezdit_stream stream) {
    const int total = L * latent_dim;
    hipLaunchKernelGGL(k_vae_sample, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, enc, noise, z, L, latent_dim);
    return launch_status("k_vae_sample");
}

}  // extern "C"
