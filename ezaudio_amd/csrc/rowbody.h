// Wave-per-row form of the row operator (split-K reduce + bias + gated residual + LayerNorm -> bf16 GEMM operand) of the row kernel
// (rowops.hip).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st_bf4(bf16_t* p, float a, float b, float c, float d, int wt = 0) {
    uint2 v;
    v.x = pack_bf2(a, b);
    v.y = pack_bf2(c, d);
    if (wt) st8_wt(p, v); else *reinterpret_cast<uint2*>(p) = v;
}

// Same operator, ONE WAVE PER ROW (4 rows per 256-thread workgroup): a lane owns up to RW float4 chunks (D <= 1280), every
// global load of the row -- residual stream, the bf16 split-K slabs, bias, gate AND the LayerNorm scale / shift vectors -- is
// issued before the first use, and both LayerNorm reductions are wave shuffles: no LDS, no barrier, no second round trip
// for the modulation vectors after the statistics.  (The workgroup-per-row form above pays 4 barriers and a dependent
// L2 round trip per row and measured 7.75 us per launch at XL for 18 MB.)
constexpr int RW = 5;
constexpr int RW_MAXS = 4;   // bf16 slabs summed by this form (more: the workgroup form)

__device__ __forceinline__ float4 bf4_to_f4(uint2 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}

template <bool CONCAT>
__device__ __forceinline__ void row_wave(const RowArgs& a, int row, int lane) {
    const int D = a.D;
    const int nc = D >> 2;
    const int b = row / a.L;
    const int slot = (a.cur_step ? *a.cur_step : 0) + (a.row_slot ? a.row_slot[b] : 0);
    const float* gate = a.gate ? a.gate + (long)slot * a.gate_slot_stride : nullptr;
    const float* lg = a.u ? a.ln_g + (long)slot * a.ln_slot_stride : nullptr;
    const float* lc = a.u ? a.ln_c + (long)slot * a.ln_slot_stride : nullptr;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);

    float4 v[RW], sb[RW], g[RW], G0[RW], C0[RW], y[CONCAT ? RW : 1], w[CONCAT ? RW : 1], G1[CONCAT ? RW : 1], C1[CONCAT ? RW : 1];
    uint2 pb[RW][RW_MAXS];
    float4 pf[RW];
    // ---- issue everything ----
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int c = lane + 64 * j;
        const bool ok = c < nc;
        const int cc = ok ? c : 0;
        v[j] = (ok && a.mode != 2) ? ld4(a.h_in + (long)row * D + cc * 4) : zero;
        sb[j] = (ok && a.mode != 0 && a.bias) ? ld4(a.bias + cc * 4) : zero;
        g[j] = (ok && a.mode != 0 && gate) ? ld4(gate + cc * 4) : one;
        pf[j] = zero;
#pragma unroll
        for (int sp = 0; sp < RW_MAXS; ++sp) pb[j][sp] = make_uint2(0u, 0u);
        if (ok && a.mode != 0) {
            if (a.part_bf16) {
#pragma unroll
                for (int sp = 0; sp < RW_MAXS; ++sp)
                    if (sp < a.nsplit) {
                        const bf16_t* sp_ptr = reinterpret_cast<const bf16_t*>(a.part) + sp * a.part_stride + (long)row * a.ld_part + cc * 4;
                        pb[j][sp] = *reinterpret_cast<const uint2*>(sp_ptr);
                    }
            } else if (a.nsplit > 0) {
                pf[j] = ld4(a.part + (long)row * a.ld_part + cc * 4);
            }
        }
        if (a.u) {
            G0[j] = ok ? ld4(lg + cc * 4) : zero;
            C0[j] = ok ? ld4(lc + cc * 4) : zero;
        }
        if constexpr (CONCAT) {
            y[j] = ok ? ld4(a.skip + (long)row * D + cc * 4) : zero;
            w[j] = (ok && a.cn) ? ld4(a.cn + (long)row * D + cc * 4) : zero;
            G1[j] = ok ? ld4(lg + D + cc * 4) : zero;
            C1[j] = ok ? ld4(lc + D + cc * 4) : zero;
        }
    }
    // ---- combine: h_new = (SET) sum + bias | (RES) h + gate * (sum + bias) | (COPY) h ----
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int c = lane + 64 * j;
        float4 s = sb[j];
        s.x += pf[j].x; s.y += pf[j].y; s.z += pf[j].z; s.w += pf[j].w;
#pragma unroll
        for (int sp = 0; sp < RW_MAXS; ++sp) {   // absent slabs hold +0.0
            const float4 t = bf4_to_f4(pb[j][sp]);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (a.mode == 1) {
            v[j].x += g[j].x * s.x; v[j].y += g[j].y * s.y; v[j].z += g[j].z * s.z; v[j].w += g[j].w * s.w;
        } else if (a.mode == 2) {
            v[j] = s;
        }
        if (c < nc) {
            if (a.h_out) {
                float* dst = a.h_out + (long)row * D + c * 4;
                if (a.wt) st16_wt(dst, v[j]); else *reinterpret_cast<float4*>(dst) = v[j];
            }
        } else {
            v[j] = zero;
        }
        s1 += v[j].x + v[j].y + v[j].z + v[j].w;
        if constexpr (CONCAT) {
            if (c < nc) {
                y[j].x += a.cn_scale * w[j].x; y[j].y += a.cn_scale * w[j].y; y[j].z += a.cn_scale * w[j].z; y[j].w += a.cn_scale * w[j].w;
            }
            s1 += y[j].x + y[j].y + y[j].z + y[j].w;
        }
    }
    if (!a.u) return;
    const float inv = 1.f / (float)(CONCAT ? 2 * D : D);
    const float mean = wave_sum(s1) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < RW; ++j)
        if (lane + 64 * j < nc) {
            float d;
            d = v[j].x - mean; q += d * d; d = v[j].y - mean; q += d * d;
            d = v[j].z - mean; q += d * d; d = v[j].w - mean; q += d * d;
            if constexpr (CONCAT) {
                d = y[j].x - mean; q += d * d; d = y[j].y - mean; q += d * d;
                d = y[j].z - mean; q += d * d; d = y[j].w - mean; q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) * inv + 1e-5f);
    bf16_t* urow = a.u + (long)row * a.ld_u;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int c = lane + 64 * j;
        if (c < nc) {
            st_bf4(urow + c * 4, (v[j].x - mean) * rstd * G0[j].x + C0[j].x, (v[j].y - mean) * rstd * G0[j].y + C0[j].y,
                   (v[j].z - mean) * rstd * G0[j].z + C0[j].z, (v[j].w - mean) * rstd * G0[j].w + C0[j].w, a.wt);
            if constexpr (CONCAT)
                st_bf4(urow + D + c * 4, (y[j].x - mean) * rstd * G1[j].x + C1[j].x, (y[j].y - mean) * rstd * G1[j].y + C1[j].y,
                       (y[j].z - mean) * rstd * G1[j].z + C1[j].z, (y[j].w - mean) * rstd * G1[j].w + C1[j].w, a.wt);
        }
    }
    for (int i = (CONCAT ? 2 * D : D) + lane; i < a.ld_u; i += 64) urow[i] = 0;
}

}  // namespace
