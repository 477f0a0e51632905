// Row-wise, HBM/L2-bound kernels of the denoising step (everything that is not a GEMM or attention).
//   k_row          residual / split-K reduce + LayerNorm (+FiLM) -> bf16 GEMM operand   blocks.py:124-156
//   k_headnorm     per-head LayerNorm(q,k) + RoPE -> attention layouts                  attention.py:137-142, rotary.py:6-18
//   (k_headnorm, second part)  V -> bf16 [keys][DV] for the P.V MFMA
//   k_assemble     MaskDiT input assembly [x | gt' | m] -> token-major bf16             conditioners.py:151-176
//   k_final_conv   unpatchify + Conv1d(C,C,3,pad 1)                                     blocks.py:209-210
//   k_linear_f32   tiny fp32 linears of the time path                                   modules.py:19-61, udit.py:305-316
//   k_mod_finalize AdaLN-SOLA combine, folded with the LayerNorm affine                  blocks.py:39-45,132-139
//   k_cfg_ddim     CFG + rescale + DDIM v-prediction update                             inference.py:12-23,88-100
// All are pure streaming kernels: 16-byte vector accesses, one wave per row, no LDS except block reductions.
#include "common.h"
#include "rowbody.h"

namespace {

constexpr int RJ = 2;  // float4 chunks per thread of the 256-thread row kernel: D <= 2048

__device__ __forceinline__ float block_sum4(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// one 256-thread workgroup per row: <= 2 float4 chunks per thread (D <= 2048), 4 waves per row keep enough loads
// in flight to stream h + the split-K slabs (a one-wave-per-row version measured 13 us for 25 MB: latency bound)
__global__ __launch_bounds__(256) void k_row(RowArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const int D = a.D;
    const int nc = D >> 2;
    const int b = row / a.L;
    const int slot = (a.cur_step ? *a.cur_step : 0) + (a.row_slot ? a.row_slot[b] : 0);
    const float* gate = a.gate ? a.gate + (long)slot * a.gate_slot_stride : nullptr;

    // all global loads of a chunk are issued back to back BEFORE the first use (a runtime-trip-count loop over the split-K
    // slabs serialised one L2/MALL round trip per slab: 7 us per launch instead of ~4)
    constexpr int MAXS = 8;
    float4 x[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int c = tid + 256 * j;
        x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nc) {
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = zero, s = zero, g = make_float4(1.f, 1.f, 1.f, 1.f), p[MAXS];
            if (a.mode != 2) v = ld4(a.h_in + (long)row * D + c * 4);
            if (a.mode != 0) {
                if (a.bias) s = ld4(a.bias + c * 4);
                if (gate) g = ld4(gate + c * 4);
#pragma unroll
                for (int sp = 0; sp < MAXS; ++sp)
                    if (sp < a.nsplit) {
                        const long e = sp * a.part_stride + (long)row * a.ld_part + c * 4;
                        if (a.part_bf16) {
                            const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.part) + e);
                            p[sp] = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                                                __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
                        } else {
                            p[sp] = ld4(a.part + e);
                        }
                    } else {
                        p[sp] = zero;
                    }
#pragma unroll
                for (int sp = 0; sp < MAXS; ++sp) { s.x += p[sp].x; s.y += p[sp].y; s.z += p[sp].z; s.w += p[sp].w; }
                if (a.mode == 1) {
                    v.x += g.x * s.x; v.y += g.y * s.y; v.z += g.z * s.z; v.w += g.w * s.w;
                } else {
                    v = s;
                }
            }
            x[j] = v;
            if (a.h_out) { if (a.wt) st16_wt(a.h_out + (long)row * D + c * 4, v); else *reinterpret_cast<float4*>(a.h_out + (long)row * D + c * 4) = v; }
        }
    }
    if (!a.u) return;

    const float* lg = a.ln_g + (long)slot * a.ln_slot_stride;
    const float* lc = a.ln_c + (long)slot * a.ln_slot_stride;
    bf16_t* urow = a.u + (long)row * a.ld_u;

    if (!a.skip) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RJ; ++j) s += x[j].x + x[j].y + x[j].z + x[j].w;  // invalid chunks are zero
        const float mean = block_sum4(s, red) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < RJ; ++j)
            if (tid + 256 * j < nc) {
                const float dx = x[j].x - mean, dy = x[j].y - mean, dz = x[j].z - mean, dw = x[j].w - mean;
                q += dx * dx + dy * dy + dz * dz + dw * dw;
            }
        const float rstd = rsqrtf(block_sum4(q, red) / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int c = tid + 256 * j;
            if (c < nc) {
                const float4 g = ld4(lg + c * 4), cc = ld4(lc + c * 4);
                st_bf4(urow + c * 4, (x[j].x - mean) * rstd * g.x + cc.x, (x[j].y - mean) * rstd * g.y + cc.y,
                       (x[j].z - mean) * rstd * g.z + cc.z, (x[j].w - mean) * rstd * g.w + cc.w, a.wt);
            }
        }
        for (int i = D + tid; i < a.ld_u; i += 256) urow[i] = 0;
    } else {
        // LayerNorm over the concatenation [h_new | skip (+ controlnet residual)], width 2D  (blocks.py:124-127)
        float4 y[RJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int c = tid + 256 * j;
            y[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nc) {
                float4 v = ld4(a.skip + (long)row * D + c * 4);
                if (a.cn) {
                    const float4 w = ld4(a.cn + (long)row * D + c * 4);
                    v.x += a.cn_scale * w.x; v.y += a.cn_scale * w.y; v.z += a.cn_scale * w.z; v.w += a.cn_scale * w.w;
                }
                y[j] = v;
            }
            s += x[j].x + x[j].y + x[j].z + x[j].w + y[j].x + y[j].y + y[j].z + y[j].w;
        }
        const float inv = 1.f / (float)(2 * D);
        const float mean = block_sum4(s, red) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < RJ; ++j)
            if (tid + 256 * j < nc) {
                float d;
                d = x[j].x - mean; q += d * d; d = x[j].y - mean; q += d * d;
                d = x[j].z - mean; q += d * d; d = x[j].w - mean; q += d * d;
                d = y[j].x - mean; q += d * d; d = y[j].y - mean; q += d * d;
                d = y[j].z - mean; q += d * d; d = y[j].w - mean; q += d * d;
            }
        const float rstd = rsqrtf(block_sum4(q, red) * inv + 1e-5f);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int c = tid + 256 * j;
            if (c < nc) {
                float4 g = ld4(lg + c * 4), cc = ld4(lc + c * 4);
                st_bf4(urow + c * 4, (x[j].x - mean) * rstd * g.x + cc.x, (x[j].y - mean) * rstd * g.y + cc.y,
                       (x[j].z - mean) * rstd * g.z + cc.z, (x[j].w - mean) * rstd * g.w + cc.w);
                g = ld4(lg + D + c * 4); cc = ld4(lc + D + c * 4);
                st_bf4(urow + D + c * 4, (y[j].x - mean) * rstd * g.x + cc.x, (y[j].y - mean) * rstd * g.y + cc.y,
                       (y[j].z - mean) * rstd * g.z + cc.z, (y[j].w - mean) * rstd * g.w + cc.w);
            }
        }
        for (int i = 2 * D + tid; i < a.ld_u; i += 256) urow[i] = 0;
    }
}

template <bool CONCAT>
__global__ __launch_bounds__(256) void k_row_w(RowArgs a) {   // rowbody.h: one wave per row, 4 rows per workgroup
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a.affine) {
        // row panel p (128 rows) is processed on XCD p % 8 (hardware deals workgroup b to XCD b % 8): the same placement as the
        // panel form of the residual GEMMs, so slabs, residual stream and LayerNorm output of a panel stay in ONE XCD's L2
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        row = 128 * (xcd + 8 * (j >> 5)) + 4 * (j & 31) + (threadIdx.x >> 6);
    }
    if (row >= a.M) return;   // wave-uniform
    row_wave<CONCAT>(a, row, threadIdx.x & 63);
}


// ---------------------------------------------------------------------------------------------------
// 4 lanes per (row, head, q|k): each lane owns DH/4 contiguous channels; the LayerNorm reductions are two
// xor-shuffles, and the RoPE partner channel (i +- DH/2, rotary.py:6-8 half split) lives in lane ^ 2 at the same
// local index.  Consecutive 4-lane groups walk the heads of one row: fully coalesced fp32 reads.
__device__ __forceinline__ void vcopy_body(const HeadNormArgs& a, int DV, int idx);

template <int DH, int DQK, int DV>
__global__ __launch_bounds__(256) void k_headnorm(HeadNormArgs a, int nb_qk) {
    constexpr int E = DH / 4;
    if ((int)blockIdx.x >= nb_qk) {  // second part of the fused launch: V -> bf16 attention layout
        vcopy_body(a, DV, ((int)blockIdx.x - nb_qk) * 256 + threadIdx.x);
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int sub = idx & 3;
    const int g = idx >> 2;
    const int nparts = (a.q_col >= 0 ? 1 : 0) + (a.k_col >= 0 ? 1 : 0);
    const int M = a.B * a.L;
    if (g >= M * a.H * nparts) return;
    const int h = g % a.H;
    const int part = (g / a.H) % nparts;
    const int m = g / (a.H * nparts);
    const bool is_q = (a.q_col >= 0) && part == 0;
    const int col = is_q ? a.q_col : a.k_col;
    const float* w = is_q ? a.qn_w : a.kn_w;
    const float* bb = is_q ? a.qn_b : a.kn_b;
    bf16_t* dstbase = is_q ? a.q : a.k;
    const int b = m / a.L, l = m % a.L;
    const float* src = a.x + (long)m * a.ldx + col + h * DH + sub * E;
    float v[E];
#pragma unroll
    for (int i = 0; i < E / 2; ++i) {
        const float2 t = *reinterpret_cast<const float2*>(src + 2 * i);
        v[2 * i] = t.x; v[2 * i + 1] = t.y;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) s += v[i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s * (1.f / DH);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) { const float d = v[i] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    const float rstd = rsqrtf(q * (1.f / DH) + 1e-5f);
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = (v[i] - mean) * rstd * w[sub * E + i] + bb[sub * E + i];
    if (a.rope_cos) {
        const float* cs = a.rope_cos + (long)l * (DH / 2) + (sub & 1) * E;
        const float* sn = a.rope_sin + (long)l * (DH / 2) + (sub & 1) * E;
        const float sign = (sub & 2) ? 1.f : -1.f;  // first half: x1*c - x2*s ; second half: x2*c + x1*s
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float other = __shfl_xor(v[i], 2, 64);
            v[i] = v[i] * cs[i] + sign * other * sn[i];
        }
    }
    bf16_t* dst = dstbase + (((long)b * a.H + h) * a.Lp + l) * DQK + sub * E;
#pragma unroll
    for (int i = 0; i < E / 2; ++i) *reinterpret_cast<uint32_t*>(dst + 2 * i) = pack_bf2(v[2 * i], v[2 * i + 1]);
}

// V fp32 [M][ldx] (columns v_col + h dh + d) -> bf16 [B][H][Lp][DV]: one 16-byte chunk (8 channels) per thread; rows l >= L and channels d >= dh stay zero
__device__ __forceinline__ void vcopy_body(const HeadNormArgs& a, int DV, int idx) {
    const int c8n = a.dh >> 3;
    const int total = a.B * a.L * a.H * c8n;
    if (idx >= total) return;
    const int c8 = idx % c8n;
    const int h = (idx / c8n) % a.H;
    const int m = idx / (c8n * a.H);
    const int b = m / a.L, l = m % a.L;
    const float* src = a.x + (long)m * a.ldx + a.v_col + h * a.dh + c8 * 8;
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    uint4 o;
    o.x = pack_bf2(lo.x, lo.y); o.y = pack_bf2(lo.z, lo.w); o.z = pack_bf2(hi.x, hi.y); o.w = pack_bf2(hi.z, hi.w);
    *reinterpret_cast<uint4*>(a.v + (((long)b * a.H + h) * a.Lp + l) * DV + c8 * 8) = o;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(AssembleArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.L * a.ldo;
    if (idx >= total) return;
    const int c = (int)(idx % a.ldo);
    const int m = (int)(idx / a.ldo);
    const int b = m / a.L, l = m % a.L;
    const int C = a.C;
    float v = 0.f;
    if (a.in_ch == C) {
        const int xb = b % a.x_rows;
        if (c < C) {
            v = a.x[((long)xb * C + c) * a.L + l];
        } else if (c < 2 * C) {
            const int cc = c - C;
            if (a.gt && !a.gt_mask[((long)xb * C + cc) * a.L + l]) v = a.gt[((long)xb * C + cc) * a.L + l];
            else v = a.mask_embed[cc];
        } else if (c == 2 * C) {
            v = a.gt ? (a.gt_mask[((long)xb * C) * a.L + l] ? 1.f : 0.f) : 1.f;
        }
    } else if (c < a.in_ch) {
        v = a.x[((long)b * a.in_ch + c) * a.L + l];
    }
    a.out[idx] = f2bf(v);
}

// out[b][co][l] = bias[co] + sum_{ci,k} w[co][ci][k] * y[b][l+k-1][ci]      (fp32 FMA: this is the model output)
// workgroup = 4 frames x all C (<= 128) output channels of one batch element; thread = one output channel x 2 frames: its
// weight row [C][3] is read with 16-byte loads (12 floats = 4 input channels x 3 taps per step), activations are LDS
// broadcasts (every lane of a wave reads the same word).
__global__ __launch_bounds__(256) void k_final_conv(FinalConvArgs a) {
    // One workgroup = 4 output positions x all C output channels of one batch element.  A thread owns ONE output channel and HALF of the input channels (threads 0 - 127 the lower
    // half, 128 - 255 the upper) for all four positions: its weight row segment (C / 2 x 3 floats, its own 16-byte loads -- 64 rows per load instruction, nothing to coalesce) is the
    // kernel's critical path, a chain of dependent load batches; with four accumulators per thread and half a row each the chain is a quarter as long as in the round-1 form
    // (one channel, two positions, the whole row: 24 batches of 12 loads, 19.4 us), the two halves meet in the LDS.
    constexpr int TL = 4;
    extern __shared__ float sy[];  // [(TL+2)][C] input rows, then [C][TL] partial sums of the upper half
    const int C = a.C;
    float* red = sy + (TL + 2) * C;
    const int ltiles = (a.L + TL - 1) / TL;
    const int b = blockIdx.x / ltiles;
    const int l0 = (blockIdx.x % ltiles) * TL;
    bool staged = false;
    auto stage = [&]() {   // the TL + 2 input rows of this tile -> LDS.  Called BEHIND the first weight block's requests: one round trip instead of two in front of the first FMA
        if (staged) return;
        for (int i = threadIdx.x; i < (TL + 2) * C; i += 256) {
            const int r = i / C, ci = i % C;
            const int l = l0 + r - 1;
            sy[r * C + ci] = (l >= 0 && l < a.L) ? a.y[((long)b * a.L + l) * a.ldy + ci] : 0.f;
        }
        __syncthreads();
        staged = true;
    };
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);   // wave-uniform
    const bool split = (C & 7) == 0;             // (otherwise the lower half walks the whole row; C % 4 == 0 as ever)
    const int ci_begin = split ? half * (C >> 1) : 0, ci_end = split ? (half + 1) * (C >> 1) : (half == 0 ? C : 0);
    for (int co0 = 0; co0 < C; co0 += 128) {
        const int co = co0 + (threadIdx.x & 127);
        const bool co_ok = co < C;
        float acc[TL] = {0.f, 0.f, 0.f, 0.f};
        const float* wr = a.w + (long)(co_ok ? co : 0) * C * 3;
        auto fma4 = [&](int ci, const float4& w0, const float4& w1, const float4& w2) {   // 4 input channels x 3 taps x TL positions
            const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y[TL + 2];
#pragma unroll
                for (int r = 0; r < TL + 2; ++r) y[r] = sy[r * C + ci + e];   // (broadcast reads: the address does not depend on the lane)
#pragma unroll
                for (int t = 0; t < TL; ++t) {
                    acc[t] = fmaf(w[3 * e], y[t], acc[t]); acc[t] = fmaf(w[3 * e + 1], y[t + 1], acc[t]); acc[t] = fmaf(w[3 * e + 2], y[t + 2], acc[t]);
                }
            }
        };
        if (((ci_end - ci_begin) & 15) == 0) {
            // blocks of 16 input channels = 12 16-byte loads; the NEXT block's loads are all issued before the current block's arithmetic (pinned: left to itself hipcc
            // interleaves them three at a time with the FMAs, i.e. one L2 round trip per 4 channels again)
            // Every workgroup reads the SAME 196 KB of weights; the workgroups of one position tile start at a block of their own (rotation by the tile index, the same for every
            // batch element: a sample's bits do not depend on its place in the batch) so that 250 workgroups do not walk the same L2 lines in the same order at the same time
            const int nblk = (ci_end - ci_begin) >> 4;
            const int rot = (blockIdx.x % ltiles) % nblk;
            float4 wc[12], wn[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) wc[q] = ld4(wr + (ci_begin + 16 * rot) * 3 + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
            stage();
            for (int ib = 0; ib < nblk; ++ib) {
                int kb = ib + rot; kb = kb >= nblk ? kb - nblk : kb;
                int kn = kb + 1; kn = kn >= nblk ? kn - nblk : kn;   // (behind the last block: a harmless re-load instead of a branch around the loads)
                const int ci = ci_begin + 16 * kb, cn = ci_begin + 16 * kn;
#pragma unroll
                for (int q = 0; q < 12; ++q) wn[q] = ld4(wr + cn * 3 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) fma4(ci + 4 * q, wc[3 * q], wc[3 * q + 1], wc[3 * q + 2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 12; ++q) wc[q] = wn[q];
            }
        } else {
            stage();
            for (int ci = ci_begin; ci < ci_end; ci += 4) fma4(ci, ld4(wr + ci * 3), ld4(wr + ci * 3 + 4), ld4(wr + ci * 3 + 8));
        }
        if (half == 1 && co_ok) *reinterpret_cast<float4*>(red + co * TL) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (half == 0 && co_ok) {
            const float4 o = *reinterpret_cast<const float4*>(red + co * TL);
            const float bb = a.b[co];
            const float4 v = make_float4((acc[0] + o.x) + bb, (acc[1] + o.y) + bb, (acc[2] + o.z) + bb, (acc[3] + o.w) + bb);
            float* dst = a.out + ((long)b * C + co) * a.L + l0;
            if (l0 + TL <= a.L && (a.L & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                if (l0 < a.L) dst[0] = v.x;
                if (l0 + 1 < a.L) dst[1] = v.y;
                if (l0 + 2 < a.L) dst[2] = v.z;
                if (l0 + 3 < a.L) dst[3] = v.w;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// y[s][j] = act(x[s][:] . W[j][:] + b[j]); one wave per output feature j and per chunk of LIN_SC slots (blockIdx.y): the
// slots of a call are the denoising timesteps (50..100), so chunking them is what fills the GPU for the small layers.
constexpr int LIN_SC = 4;
__global__ __launch_bounds__(256) void k_linear_f32(const float* __restrict__ x, const int* __restrict__ ts, int x_mode,
                                                    const float* __restrict__ W, const float* __restrict__ bias,
                                                    float* __restrict__ y, int n, int N, int K, int act, long y_stride) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    const float* wr = W + (long)j * K;
    const float bj = bias ? bias[j] : 0.f;
    const int s_end = min(n, ((int)blockIdx.y + 1) * LIN_SC);
    for (int s = blockIdx.y * LIN_SC; s < s_end; ++s) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) {
            float xv;
            if (x_mode == 1) {  // timestep_embedding(t, 256): [cos(t f) | sin(t f)], f_k = exp(-ln(1e4) k / 128)
                const int half = K >> 1;
                const int kk = k < half ? k : k - half;
                const float f = expf(-9.210340371976184f * (float)kk / (float)half);
                const float arg = (float)ts[s] * f;
                xv = k < half ? cosf(arg) : sinf(arg);
            } else {
                xv = x[(long)s * K + k];
            }
            acc += xv * wr[k];
        }
        acc = wave_sum(acc) + bj;
        if (act == 1) acc = acc / (1.f + expf(-acc));
        if (lane == 0) y[(long)s * y_stride + j] = acc;
    }
}

__global__ __launch_bounds__(256) void k_mod_finalize(ModFinalizeArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int D = a.D;
    const long per_slot = (long)(a.nblk + a.has_final) * D;  // nblk block entries (+ 1 final entry)
    if (idx >= per_slot * a.n) return;
    const int s = (int)(idx / per_slot);
    const int r = (int)(idx % per_slot);
    const int blk = r / D, d = r % D;
    if (blk < a.nblk) {
        float six[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            six[i] = a.ada[(long)s * 6 * D + i * D + d] +
                     a.scaling * a.lora[((long)s * a.nblk + blk) * 6 * D + i * D + d] +
                     a.table[(long)blk * a.table_stride + i * D + d];
        // chunk order (blocks.py:132-133): shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        const float w1 = a.n1w[(long)blk * a.norm_stride + d], b1 = a.n1b[(long)blk * a.norm_stride + d];
        const float w3 = a.n3w[(long)blk * a.norm_stride + d], b3 = a.n3b[(long)blk * a.norm_stride + d];
        float* o = a.mod + (((long)s * a.nblk + blk) * 6) * D + d;
        o[0 * D] = w1 * (1.f + six[1]);
        o[1 * D] = b1 * (1.f + six[1]) + six[0];
        o[2 * D] = 1.f - six[2];
        o[3 * D] = w3 * (1.f + six[4]);
        o[4 * D] = b3 * (1.f + six[4]) + six[3];
        o[5 * D] = 1.f - six[5];
    } else {
        const float shift = a.ada_final[(long)s * 2 * D + d];   // blocks.py:203-204: shift first
        const float scale = a.ada_final[(long)s * 2 * D + D + d];
        float* o = a.mod_final + (long)s * 2 * D + d;
        o[0] = a.nfw[d] * (1.f + scale);
        o[D] = a.nfb[d] * (1.f + scale) + shift;
    }
}

__global__ void k_rope_table(float* cosT, float* sinT, int max_len, int dh) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int half = dh / 2;
    if (idx >= max_len * half) return;
    const int p = idx / half, i = idx % half;
    // rotary.py:42: inv_freq = 1 / (10000 ** (arange(0, dim, 2) / dim)); angles p * inv_freq in fp32
    const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * i) / (float)dh);
    const float ang = (float)p * inv_freq;
    cosT[idx] = cosf(ang);
    sinT[idx] = sinf(ang);
}

// ---------------------------------------------------------------------------------------------------
// CFG + guidance rescale + DDIM update in two grid-wide passes (a single workgroup per sample measured 118 us):
//   k_cfg_stats : per (sample, chunk) partial sums  S(c), S(c^2), S(g), S(g^2)   with g = u + s (c - u)
//   k_cfg_apply : every workgroup re-reduces the NB partials in double (deterministic, no atomics), forms
//                 std(c)/std(g) with torch.std's unbiased normalisation, and updates its chunk of the latent.
constexpr int CFG_NB = 64;

__global__ __launch_bounds__(256) void k_cfg_stats(CfgDdimArgs a, float* partial) {
    __shared__ float red[4];
    const int p = blockIdx.y, blk = blockIdx.x;
    const int n = a.n;
    const float* pc = a.pred + (long)p * n;
    const float* pu = a.pred + (long)(a.P + p) * n;
    const float gs = a.guidance_scale;
    float s1 = 0.f, q1 = 0.f, s2 = 0.f, q2 = 0.f;
    for (int i = blk * 256 + threadIdx.x; i < n; i += CFG_NB * 256) {
        const float c = pc[i], u = pu[i];
        const float g = u + gs * (c - u);
        s1 += c; q1 += c * c; s2 += g; q2 += g * g;
    }
    s1 = block_sum4(s1, red); q1 = block_sum4(q1, red); s2 = block_sum4(s2, red); q2 = block_sum4(q2, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((long)p * CFG_NB + blk) * 4;
        o[0] = s1; o[1] = q1; o[2] = s2; o[3] = q2;
    }
}

__global__ __launch_bounds__(256) void k_cfg_apply(CfgDdimArgs a, const float* partial) {
    const int p = blockIdx.y, blk = blockIdx.x;
    const int n = a.n;
    const int step = a.cur_step ? *a.cur_step : 0;
    const float* cf = a.cur_step ? a.coef + step * 8 : a.hc;
    const float sa = cf[0], sb = cf[1], cx0 = cf[2], cdir = cf[3], sigma = cf[4];
    const bool cfg = a.guidance_scale > 0.f;
    const bool rescale = cfg && a.guidance_rescale > 0.f;
    float ratio = 1.f;
    if (rescale) {   // wave-uniform
        // the CFG_NB partial sums of this sample: one 16-byte load per thread into LDS, then every thread adds them up in the SAME order as before
        // (broadcast LDS reads).  The former loop of CFG_NB dependent scalar loads was 10 of the kernel's 12.5 us.
        __shared__ float4 part_l[CFG_NB];
        if (threadIdx.x < CFG_NB) part_l[threadIdx.x] = reinterpret_cast<const float4*>(partial)[(long)p * CFG_NB + threadIdx.x];
        __syncthreads();
        double s1 = 0, q1 = 0, s2 = 0, q2 = 0;
#pragma unroll 8
        for (int i = 0; i < CFG_NB; ++i) {
            const float4 o = part_l[i];
            s1 += o.x; q1 += o.y; s2 += o.z; q2 += o.w;
        }
        const double v1 = (q1 - s1 * s1 / n) / (n - 1);  // torch.std default: unbiased
        const double v2 = (q2 - s2 * s2 / n) / (n - 1);
        ratio = (float)(sqrt(v1) / sqrt(v2));
    }
    const float* pc = a.pred + (long)p * n;
    const float* pu = cfg ? a.pred + (long)(a.P + p) * n : nullptr;
    float* lat = a.latents + (long)p * n;
    const float* z = a.noise ? a.noise + ((long)step * a.P + p) * n : nullptr;
    const float gs = a.guidance_scale, phi = a.guidance_rescale;
    for (int i = blk * 256 + threadIdx.x; i < n; i += CFG_NB * 256) {
        float v = pc[i];
        if (cfg) {
            const float u = pu[i];
            v = u + gs * (v - u);
            if (rescale) v = phi * (v * ratio) + (1.f - phi) * v;
        }
        const float x = lat[i];
        const float x0 = sa * x - sb * v;
        const float eps = sa * v + sb * x;
        float prev = cx0 * x0 + cdir * eps;
        if (z) prev += sigma * z[i];
        lat[i] = prev;
    }
    // Every workgroup has read *cur_step above.  The last one to arrive here advances it for the next step (device-scope
    // arrival counter, reset by that same workgroup; both stores are made visible by the end-of-kernel release).
    if (a.step_inc) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = gridDim.x * gridDim.y;
            const unsigned prev = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == total - 1) {
                __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.step_inc, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void k_set_int(int* p, int v, int add) { *p = add ? *p + v : v; }

// small direct Conv1d in fp32 for the ControlNet condition embed (controlnet.py:15-39,65-84): 4 tiny layers, once per call
__global__ __launch_bounds__(256) void k_conv1d(Conv1dArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.Cout * a.Lout;
    if (idx >= total) return;
    const int lo = (int)(idx % a.Lout);
    const int co = (int)((idx / a.Lout) % a.Cout);
    const int b = (int)(idx / ((long)a.Lout * a.Cout));
    float acc = a.b ? a.b[co] : 0.f;
    for (int ci = 0; ci < a.cin_valid; ++ci) {
        const float* xr = a.x + ((long)b * a.cin_valid + ci) * a.Lin;
        const float* wr = a.w + ((long)co * a.Cin + ci) * a.ksize;
        for (int k = 0; k < a.ksize; ++k) {
            const int li = lo * a.stride + k - a.pad;
            if (li >= 0 && li < a.Lin) acc += wr[k] * xr[li];
        }
    }
    if (a.act == 1) acc = acc / (1.f + expf(-acc));
    if (a.out_token_major) a.out[((long)b * a.Lout + lo) * a.Cout + co] = acc;
    else a.out[((long)b * a.Cout + co) * a.Lout + lo] = acc;
}

// out bf16 [M][ldo] = act(x fp32 [M][ldx]) for cols < N, zero for N <= col < ldo
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ x, int ldx, bf16_t* __restrict__ out,
                                                   int ldo, int M, int N, int act) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * ldo) return;
    const int c = (int)(idx % ldo);
    const int m = (int)(idx / ldo);
    float v = 0.f;
    if (c < N) {
        v = x[(long)m * ldx + c];
        if (act == 1) v = v / (1.f + expf(-v));
    }
    out[idx] = f2bf(v);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
void launch_row(const RowArgs& a, hipStream_t st) {
    const bool wave_form = a.variant == 1 && a.D <= RW * 256 && (a.D & 3) == 0 &&
                           (a.mode == 0 || (a.part_bf16 ? a.nsplit <= RW_MAXS : a.nsplit <= 1));
    if (wave_form) {
        const dim3 grid(a.affine ? 8 * 32 * (((a.M + 127) / 128 + 7) / 8) : (a.M + 3) / 4);
        if (a.skip) hipLaunchKernelGGL(k_row_w<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_row_w<false>, grid, dim3(256), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(k_row, dim3(a.M), dim3(256), 0, st, a);
}

void launch_headnorm(const HeadNormArgs& a, hipStream_t st) {
    // ONE launch: blocks [0, nb_qk) do the per-head LayerNorm (+RoPE) of q / k, blocks [nb_qk, nb_qk + nb_v) cast V
    const int M = a.B * a.L;
    const int nparts = (a.q_col >= 0 ? 1 : 0) + (a.k_col >= 0 ? 1 : 0);
    const int nb_qk = (M * a.H * 4 * nparts + 255) / 256;
    const int nb_v = a.v_col >= 0 ? (a.B * a.L * a.H * (a.dh / 8) + 255) / 256 : 0;
    if (nb_qk + nb_v == 0) return;
    if (a.dh == 64) hipLaunchKernelGGL((k_headnorm<64, 64, 64>), dim3(nb_qk + nb_v), dim3(256), 0, st, a, nb_qk);
    else hipLaunchKernelGGL((k_headnorm<72, 80, 96>), dim3(nb_qk + nb_v), dim3(256), 0, st, a, nb_qk);
}

void launch_assemble(const AssembleArgs& a, hipStream_t st) {
    const long total = (long)a.B * a.L * a.ldo;
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

void launch_final_conv(const FinalConvArgs& a, hipStream_t st) {
    const int ltiles = (a.L + 3) / 4;
    const size_t sh = (size_t)(6 + 4) * a.C * sizeof(float);   // input rows + the upper half's partial sums
    hipLaunchKernelGGL(k_final_conv, dim3(a.B * ltiles), dim3(256), sh, st, a);
}

void launch_linear_f32(const float* x, const int* ts, int x_mode, const float* W, const float* b, float* y,
                       int n, int N, int K, int act, long y_stride, hipStream_t st) {
    hipLaunchKernelGGL(k_linear_f32, dim3((N + 3) / 4, (n + LIN_SC - 1) / LIN_SC), dim3(256), 0, st, x, ts, x_mode, W, b, y, n, N, K, act, y_stride);
}

void launch_mod_finalize(const ModFinalizeArgs& a, hipStream_t st) {
    const long total = (long)(a.nblk + a.has_final) * a.D * a.n;
    hipLaunchKernelGGL(k_mod_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

void launch_rope_table(float* cosT, float* sinT, int max_len, int dh, hipStream_t st) {
    const int total = max_len * (dh / 2);
    hipLaunchKernelGGL(k_rope_table, dim3((total + 255) / 256), dim3(256), 0, st, cosT, sinT, max_len, dh);
}

void launch_cfg_ddim(const CfgDdimArgs& a, float* partial, hipStream_t st) {
    if (a.guidance_scale > 0.f && a.guidance_rescale > 0.f)
        hipLaunchKernelGGL(k_cfg_stats, dim3(CFG_NB, a.P), dim3(256), 0, st, a, partial);
    hipLaunchKernelGGL(k_cfg_apply, dim3(CFG_NB, a.P), dim3(256), 0, st, a, partial);
}

// ---- LayerNorm algebra tables (common.h, GemmArgs.z*): G' = g W^T and C' = c W^T (+ bias) for every modulation slot.  The gain / shift
// vectors enter the bf16 MFMA GEMM as EXACT hi + lo bf16 pairs (g = hi + lo up to 2^-17 relative), four operand rows per slot:
// (g hi, g lo, c hi, c lo); k_z_combine adds the pairs back.  Runs once per call (ezdit_prepare_timesteps), not per step.
__global__ void k_z_hilo(const float* g, const float* c, long slot_stride, bf16_t* out, int ldo, int n_slots, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_slots * ldo) return;
    const int s = (int)(i / ldo), k = (int)(i % ldo);
    float gv = 0.f, cv = 0.f;
    if (k < D) { gv = g[(long)s * slot_stride + k]; cv = c[(long)s * slot_stride + k]; }
    const uint16_t gh = f2bf(gv), ch = f2bf(cv);
    bf16_t* o = out + (long)(4 * s) * ldo + k;
    o[0] = gh;
    o[ldo] = f2bf(gv - bf2f(gh));
    o[2 * (long)ldo] = ch;
    o[3 * (long)ldo] = f2bf(cv - bf2f(ch));
}
__global__ void k_z_combine(const float* tmp, int ld_tmp, const float* bias, float* zG, float* zC, long slot_stride, int n_slots, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_slots * N) return;
    const int s = (int)(i / N), n = (int)(i % N);
    const float* t = tmp + (long)(4 * s) * ld_tmp + n;
    zG[(long)s * slot_stride + n] = t[0] + t[ld_tmp];
    zC[(long)s * slot_stride + n] = t[2 * (long)ld_tmp] + t[3 * (long)ld_tmp] + (bias ? bias[n] : 0.f);
}
void launch_z_hilo(const float* g, const float* c, long slot_stride, bf16_t* out, int ldo, int n_slots, int D, hipStream_t st) {
    const long total = (long)n_slots * ldo;
    hipLaunchKernelGGL(k_z_hilo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g, c, slot_stride, out, ldo, n_slots, D);
}
void launch_z_combine(const float* tmp, int ld_tmp, const float* bias, float* zG, float* zC, long slot_stride, int n_slots, int N, hipStream_t st) {
    const long total = (long)n_slots * N;
    hipLaunchKernelGGL(k_z_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, ld_tmp, bias, zG, zC, slot_stride, n_slots, N);
}

// one wave per output: y[i][n] = bias[n] + x[xrow[i]] . W[n]  (the constant cross-attention-out vectors of the single-key batch elements of ONE block,
// ezdit_prepare_context: up to GEMV_MAXB of them per launch, blockIdx.y = i)
__global__ __launch_bounds__(256) void k_gemv_bf16w(GemvBatch g, int x_bf16, const bf16_t* __restrict__ W, int ldw, const float* __restrict__ bias, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* x = g.x + g.xoff[blockIdx.y];
    const bf16_t* w = W + (long)n * ldw;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float xv = x_bf16 ? bf2f(f2bf(x[k])) : x[k];
        acc = fmaf(xv, bf2f(w[k]), acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) g.y[g.yoff[blockIdx.y] + n] = acc + (bias ? bias[n] : 0.f);
}
void launch_gemv_bf16w(const GemvBatch& g, int x_bf16, const bf16_t* W, int ldw, const float* bias, int N, int K, hipStream_t st) {
    if (g.n <= 0) return;
    hipLaunchKernelGGL(k_gemv_bf16w, dim3((unsigned)((N + 3) / 4), (unsigned)g.n), dim3(256), 0, st, g, x_bf16, W, ldw, bias, N, K);
}

void launch_cast_bf16(const float* x, int ldx, bf16_t* out, int ldo, int M, int N, int act, hipStream_t st) {
    const long total = (long)M * ldo;
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, out, ldo, M, N, act);
}

void launch_conv1d(const Conv1dArgs& a, hipStream_t st) {
    const long total = (long)a.B * a.Cout * a.Lout;
    hipLaunchKernelGGL(k_conv1d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

void launch_set_int(int* p, int v, int add, hipStream_t st) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, p, v, add);
}
