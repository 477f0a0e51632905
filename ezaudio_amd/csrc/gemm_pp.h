// Ping-pong bf16 MFMA GEMM for M <= ~2000 rows:  C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear layout, both K-contiguous)
//
// Same math as k_gemm (gemm.hip; reference: the aten::linear calls of src/models/utils/attention.py:127-129,148,
// src/models/utils/modules.py:263-277,341-374, src/models/blocks.py:124-128), different K loop.
//
// Why: in k_gemm all waves of a workgroup do the same thing at the same time (refill -> fragment reads -> MFMAs -> barrier), so
// the parts of a K tile add up (measured 1.23 us per 128x288 tile against 0.51 us of MFMAs).  Here a
// workgroup is 8 waves = TWO GROUPS of 4 (wave w and w + 4 share a SIMD) that run one barrier interval apart: in every
// interval one group is in its LOAD phase (all fragments of a K tile -> registers, its share of the LDS-DMA refill) and the
// other in its MFMA phase (36 MFMAs straight out of registers), so each SIMD always has one wave feeding the matrix pipe and one
// wave feeding the memory pipes.  v_mfma_f32_16x16x32_bf16 (wave tiles that are multiples of 16, e.g. 32 x 144).
//
//   SCHED 1 (both groups walk every K tile; wave tile = 1/8 of the block):      interval  2t    2t+1   2t+2
//       group 0                                                                           LOAD(t) MFMA(t) LOAD(t+1)
//       group 1                                                                           MFMA(t-1) LOAD(t) MFMA(t)
//   SCHED 2 (k-split: group g takes the K tiles t = g mod 2; wave tile = 1/4 of the block, the two partial sums are
//       exchanged through LDS once, after the loop, each group keeping half of the wave tile's rows): interval i = LOAD(i) by
//       group i & 1, MFMA(i - 1) by the other group.  A ring slot is read in ONE interval, which leaves NS - 1 tiles in flight.
//
// LDS image of a K tile (BK = 64): rows [A tile | W tile], 128 B each, filled by global_load_lds (16 B per lane) with the bank
// swizzle on the SOURCE address (chunk c of row r lands in slot c ^ ((r >> 1) & 7), as in k_gemm); a stage is a whole number
// of 4-KB pieces (one piece = one LDS-DMA instruction of every thread of a group = 32 rows); a ragged tail piece re-fetches a
// clamped row into padding so that every wave issues the same number of loads (uniform vmcnt accounting).
//
// Hazards (all waits are counted vmcnt on the issuing wave followed by a barrier the readers pass; reads are drained with
// lgkmcnt(0) before the barrier that precedes the next LDS-DMA into the slot):
//   SCHED 1: tile u's pieces are issued in LOAD(u - PD) of each group (PD = NS - 1), waited for by group 0 at the end of
//            MFMA(u - 1) and by group 1 at the end of LOAD(u - 1) -- both before the barrier that opens group 0's LOAD(u).
//            Slot (u % NS) was last read in group 1's LOAD(u - NS), one barrier before group 0's LOAD(u - PD) issues into it.
//   SCHED 2: tile u is issued in interval u - PD by that interval's LOAD group, which waits for it at the end of interval u - 1.
#pragma once
#include "common.h"

#include <type_traits>

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `younger` x PER (+ EXTRA) loads are in flight (younger in [0, MAXY])
template <int PER, int MAXY, int EXTRA = 0>
__device__ __forceinline__ void wait_younger(int younger) {
    if constexpr (MAXY >= 3) { if (younger >= 3) { wait_vmcnt<3 * PER + EXTRA>(); return; } }
    if constexpr (MAXY >= 2) { if (younger >= 2) { wait_vmcnt<2 * PER + EXTRA>(); return; } }
    if constexpr (MAXY >= 1) { if (younger >= 1) { wait_vmcnt<PER + EXTRA>(); return; } }
    wait_vmcnt<EXTRA>();
}

// "Blind" vector loads (inline asm: hipcc's waitcnt bookkeeping neither counts them nor waits for them): the row statistics of the LayerNorm algebra (z_lane_load below).
// History of why: as C++ loads, merged per workgroup by `z_finish` between the prologue's LDS-DMA and the K loop, they cost every consumer ~3K cycles (2.4 % of the XL step,
// profiles/r06_experiments.txt r06r): hipcc put `s_waitcnt vmcnt(0)` in front of the merge (its bookkeeping is path-insensitive behind `tid <` branches) and in front of each
// LDS store (a store next to an LDS-DMA it cannot prove finished), so the loop started when EVERY prologue tile had landed instead of the first.  Rules that keep an asm output
// safe (the compiler believes it defined at once): every lane issues every load (clamped address, no branch: a phi at a join may become a copy of a register whose data
// has not arrived), the outputs are first named by an empty asm BEHIND the wait (z_lane_finish), and tests/test_host.py checks on the generated code that nothing reads them in between.
__device__ __forceinline__ f32x2 ld8_blind(const void* p) {   // into an AGPR pair: the value rides through the K loop next to the accumulators, where registers are free
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(v) : "v"(p) : "memory");
    return v;
}

// The per-column vectors of an epilogue (G' | C' of a consumer, bias | gate | gain of a producer) go STRAIGHT into their LDS table by one 16-byte LDS-DMA per thread of the
// first waves: no register, no park, and the K loop does not wait for them at all -- they are older than every K tile but the first, so whatever wait covers tile 1 covers them.
// `extra` = this wave issued that one DMA between tile 0 and the younger tiles (wave-uniform): the counted wait in front of the loop leaves it in flight.
template <int PER, int MAXY, int BASE>
__device__ __forceinline__ void wait_younger_x(int younger, int extra /* 0 .. 2, wave-uniform */) {
    if (extra >= 2) wait_younger<PER, MAXY, BASE + 2>(younger);
    else if (extra == 1) wait_younger<PER, MAXY, BASE + 1>(younger);
    else wait_younger<PER, MAXY, BASE>(younger);
}

// exact-erf GELU (F.gelu default, modules.py:268-272) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e.
// fp32-rounding class and far below the bf16 rounding of the result): ~14 VALU instead of ~40 for erff
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = 1.0f - p * t * e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// the same on a pair of values with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane and issue slot; the two
// reciprocals and the two exponentials stay scalar -- the transcendental unit has no packed form): the GEGLU epilogue evaluates 72 GELUs per
// lane, ~16 VALU instructions each in the scalar form, on the two waves that share a SIMD (in-situ stamps: 6.3K of the 10K-cycle epilogue)
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 z = __builtin_elementwise_abs(x) * f32x2{0.70710678118654752440f, 0.70710678118654752440f};
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    const f32x2 a = (z * f32x2{-1.4426950408889634f, -1.4426950408889634f}) * z;
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 erf_abs = __builtin_elementwise_fma(-(p * t), e, f32x2{1.0f, 1.0f});
    const f32x2 s = {copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])};
    return (x * f32x2{0.5f, 0.5f}) * (s + f32x2{1.0f, 1.0f});
}

// XCD-aware tile map shared by the GEMM kernels: workgroup b runs on XCD b % 8; XCD x owns box (xm, xn, xz) of the
// (M tiles x N tiles x K splits) grid, M tiles fastest inside.  Returns false for a padding slot of a ragged box.
__device__ __forceinline__ bool tile_of_block(const GemmArgs& a, int tilesM, int tilesN, int& tm, int& tn, int& z) {
    // division-free: pm, pn, pz are powers of two (pick_boxes), l / bm and (l / bm) / bn by magic numbers (common.h ez_div)
    const int xcd = blockIdx.x & 7;
    const int l = blockIdx.x >> 3;
    const int lpm = __builtin_ctz(a.pm), lpn = __builtin_ctz(a.pn);
    const int xm = xcd & (a.pm - 1), xn = (xcd >> lpm) & (a.pn - 1), xz = xcd >> (lpm + lpn);
    const int t1 = ez_div(l, a.mbm), lm = l - t1 * a.bm;
    const int lz = ez_div(t1, a.mbn), ln = t1 - lz * a.bn;
    tm = xm * a.bm + lm;
    tn = xn * a.bn + ln;
    z = xz * a.bz + lz;
    return tm < tilesM && tn < tilesN && z < a.splitk;
}
// panel placement of a split-K GEMM: ALL workgroups of an M tile (N tiles x K splits) are dealt to ONE XCD (M tile tm -> XCD tm % 8)
__device__ __forceinline__ void panel_of_block(const GemmArgs& a, int tilesN, int& tm, int& tn, int& z) {
    const int G = tilesN * a.splitk, l = blockIdx.x >> 3;
    const int lg = ez_div(l, a.mG), r = l - lg * G;
    tm = (blockIdx.x & 7) + 8 * lg;
    tn = ez_div(r, a.msplit);
    z = r - tn * a.splitk;
}
// K tiles [kb, ke) of split z
__device__ __forceinline__ void ksplit_range(const GemmArgs& a, int nk, int z, int& kb, int& ke) {
    if (a.splitk == 1) { kb = 0; ke = nk; return; }
    kb = ez_div(nk * z, a.msplit);
    ke = ez_div(nk * (z + 1), a.msplit);
}

// ---- copy a bf16 tile parked in LDS ([rows][pitch]) out to global memory as whole 16-byte row chunks ----
template <int NT>
__device__ __forceinline__ void copy_out_bf16(const bf16_t* tile, int pitch, int rows, int oc /* columns of the tile */, bf16_t* out, int ldo,
                                              int grow0, int gcol0, int M, int ncols, int wt, int tid) {
    const int cpr = oc / 8;
    for (int q = tid; q < rows * cpr; q += NT) {
        const int r = q / cpr, c = (q % cpr) * 8;
        const int grow = grow0 + r, gcol = gcol0 + c;
        if (grow < M && gcol < ncols) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + r * pitch + c);
            bf16_t* dst = out + (long)grow * ldo + gcol;
            if (gcol + 8 <= ncols) {
                if (wt) st16_wt(dst, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
                else *reinterpret_cast<uint4*>(dst) = v;
            } else {   // ragged last chunk (N is a multiple of 4 for every caller)
                const bf16_t* src = tile + r * pitch + c;
                for (int e = 0; e < ncols - gcol; ++e) dst[e] = src[e];
            }
        }
    }
}

// m / L and m % L for a token row m < 2^22 and a batch-element length 1 <= L <= 2048 without the ~40-instruction integer division (twice per
// 16-byte chunk in the fused-QKV copy-out loops): the quotient of (m + 0.5) / L sits at least 0.5 / L away from an integer, far more than the
// fp32 error of the product with the approximate reciprocal (<= 3 ulp of a value <= 240 x 2048 / L)
__device__ __forceinline__ void divmod_rows(int m, int L, float rcpL, int& b, int& l) {
    b = (int)(((float)m + 0.5f) * rcpL);
    l = m - b * L;
}

// modulation slot of a row: the device step counter (slot0, read once per kernel) + the row's batch element offset (per-row timesteps only)
__device__ __forceinline__ int z_slot(const GemmArgs& a, int slot0, int rowc) {
    return slot0 + (a.row_slot ? a.row_slot[rowc / a.rows_per_b] : 0);
}

// ---- epilogues for the 16x16 C layout.  The MFMAs compute the TRANSPOSED tile (W fragment as the A operand), so a lane owns ONE
// output row m = lane & 15 of a 16 x 16 fragment and FOUR CONSECUTIVE columns 4 * (lane >> 4) + {0..3}.
// acc[i][j]: i-th 16-row x j-th 16-column fragment of the wave tile at (wm * TM, wn * TN) of the workgroup tile.
template <int FM, int FN, int TM, int TN, int EPI>
__device__ __forceinline__ void pp_store_direct(const GemmArgs& a, f32x4 (&acc)[FM][FN], int row0, int col0, int wm, int wn, int lane, int z) {
    static_assert(EPI == EPI_F32 || EPI == EPI_PARTIAL, "fp32 outputs");
    const int m_in = lane & 15, cg = lane >> 4;
    float* out = reinterpret_cast<float*>(a.out);
    if constexpr (EPI == EPI_PARTIAL) out += (long)z * a.slab_stride;
    // every global operand of the epilogue is requested up front, unconditionally (clamped addresses): a load inside the bounds
    // check makes hipcc wait vmcnt(0) once per fragment -- 18 dependent L2 round trips in the first build of this kernel
    float4 b4[FN], g4[FN];
    const int ncl = a.N - 4;   // N is a multiple of 4 for every caller
    if constexpr (EPI == EPI_F32) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            int col = col0 + wn * TN + j * 16 + 4 * cg;
            col = col < ncl ? col : ncl;
            b4[j] = a.bias ? *reinterpret_cast<const float4*>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = row0 + wm * TM + i * 16 + m_in;
        const int rowc = row < a.M ? row : a.M - 1;
        float4 r4[FN];
        if constexpr (EPI == EPI_F32) {
            if (a.resid) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    int col = col0 + wn * TN + j * 16 + 4 * cg;
                    col = col < ncl ? col : ncl;
                    r4[j] = *reinterpret_cast<const float4*>(a.resid + (long)rowc * a.ldr + col);
                }
                if (a.gate) {
                    const int slot = (a.cur_step ? *a.cur_step : 0) + (a.row_slot ? a.row_slot[rowc / a.rows_per_b] : 0);
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        int col = col0 + wn * TN + j * 16 + 4 * cg;
                        col = col < ncl ? col : ncl;
                        g4[j] = *reinterpret_cast<const float4*>(a.gate + (long)slot * a.gate_slot_stride + col);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int col = col0 + wn * TN + j * 16 + 4 * cg;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if constexpr (EPI == EPI_F32) {
                v.x += b4[j].x; v.y += b4[j].y; v.z += b4[j].z; v.w += b4[j].w;
                if (a.resid) {
                    if (a.gate) { v.x *= g4[j].x; v.y *= g4[j].y; v.z *= g4[j].z; v.w *= g4[j].w; }
                    v.x += r4[j].x; v.y += r4[j].y; v.z += r4[j].z; v.w += r4[j].w;
                }
            }
            if (row < a.M && col < a.N) {
                float* dst = out + (long)row * a.ldo + col;
                if (a.wt) st16_wt(dst, v); else *reinterpret_cast<float4*>(dst) = v;
            }
        }
    }
}

// bf16 outputs (GEGLU activations, bf16 split-K slabs) through LDS: the tile is parked in the dead ring and leaves as whole rows, 16
// bytes per lane (a direct store would scatter 8 bytes into each of 16 lines per instruction).
// GEGLU: W rows are interleaved 8 value / 8 gate, so a 16-column fragment holds inner indices 8 j' .. 8 j' + 7: values in the lanes
// with cg = lane >> 4 in {0, 1}, their gates in the lanes cg + 2 (= lane ^ 32).  Each lane of a pair finishes two of the four outputs
// (branch-free: both lanes evaluate  mul * gelu(arg)  with their own selection of mul / arg).
template <int BM, int BN, int FM, int FN, int TM, int TN, int NT, int EPI, bool ZC>
__device__ __forceinline__ void pp_store_lds(const GemmArgs& a, f32x4 (&acc)[FM][FN], char* smem, int row0, int col0, int wm, int wn, int lane, int tid, int z,
                                             const float2* zmr /* (mu, r) of this lane's row fragments: z_lane_finish */, const float* zgc, int slot0, unsigned long long* ts = nullptr) {
    static_assert(EPI == EPI_GEGLU || EPI == EPI_PARTIAL, "bf16 outputs only");
    constexpr int OC = EPI == EPI_GEGLU ? BN / 2 : BN;   // output columns of the tile
    static_assert(OC % 8 == 0, "16-byte row chunks");
    constexpr int PITCH = OC + 8;
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
    const int m_in = lane & 15, cg = lane >> 4;
    uint32_t pk[FM][FN];
    if constexpr (EPI == EPI_GEGLU) {
        // the math runs out of registers BEFORE the barrier that frees the ring: the group that finished its MFMAs one interval earlier
        // overlaps its GELUs with the other group's last MFMA phase.  Exchange: v_permlane32_swap swaps the upper 32 lanes of its first
        // operand with the lower 32 lanes of its second; with (x0, x2) [and (x1, x3)] as operands the value lane ends up with
        // (v0, g0), the gate lane with (v2, g2): BOTH evaluate  first * gelu(second), no selects, no LDS.
        // Every global operand is requested up front, unconditionally (clamped addresses): a load inside a bounds check makes hipcc wait
        // vmcnt(0) once per fragment (18 dependent L2 round trips in the first build of this kernel).
        const int ncl = a.N - 4;
        // per-column vectors of the epilogue, requested once, up front: the bias (plain), or G' and C' of the modulation slot (LayerNorm algebra;
        // C' includes the bias).  With per-row timesteps (row_slot) the slot differs between rows: they are re-read per row fragment then.
        float4 c4[FN], g4[FN];
        const bool shared_slot = !ZC || !a.row_slot;
        if (shared_slot) {
            const long so = ZC ? (long)slot0 * a.zt_slot_stride : 0;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                int cp = col0 + wn * TN + j * 16 + 4 * cg;
                cp = cp < ncl ? cp : ncl;
                if constexpr (ZC) {   // parked behind the ring by LDS-DMA (k_gemm_pp, z_late_load)
                    (void)so;
                    g4[j] = *reinterpret_cast<const float4*>(zgc + (wn * TN + j * 16 + 4 * cg));
                    c4[j] = *reinterpret_cast<const float4*>(zgc + BN + (wn * TN + j * 16 + 4 * cg));
                } else {
                    g4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    c4[j] = a.bias ? *reinterpret_cast<const float4*>(a.bias + cp) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float mu = 0.f, r = 1.f;
            if constexpr (ZC) {   // LayerNorm algebra: x = r (acc - mu G') + C'
                const int rl = wm * TM + i * 16 + m_in;
                if (!shared_slot) {
                    int row = row0 + rl;
                    row = row < a.M ? row : a.M - 1;
                    const long so = (long)z_slot(a, slot0, row) * a.zt_slot_stride;
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        int cp = col0 + wn * TN + j * 16 + 4 * cg;
                        cp = cp < ncl ? cp : ncl;
                        g4[j] = *reinterpret_cast<const float4*>(a.zG + so + cp);
                        c4[j] = *reinterpret_cast<const float4*>(a.zC + so + cp);
                    }
                }
                const float2 mr = zmr[i];
                mu = mr.x; r = mr.y;
            }
            const float rm = r * mu;
            const f32x2 r2 = {r, r}, nrm2 = {-rm, -rm};
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                // packed fp32: (x0, x1) and (x2, x3) in one FMA each (the same two IEEE roundings per element as fmaf(r, acc, fmaf(-rm, g, c)))
                const f32x2 x01 = __builtin_elementwise_fma(r2, f32x2{acc[i][j][0], acc[i][j][1]}, __builtin_elementwise_fma(nrm2, f32x2{g4[j].x, g4[j].y}, f32x2{c4[j].x, c4[j].y}));
                const f32x2 x23 = __builtin_elementwise_fma(r2, f32x2{acc[i][j][2], acc[i][j][3]}, __builtin_elementwise_fma(nrm2, f32x2{g4[j].z, g4[j].w}, f32x2{c4[j].z, c4[j].w}));
                const auto e0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x01[0]), __float_as_uint(x23[0]), false, false);
                const auto e1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x01[1]), __float_as_uint(x23[1]), false, false);
                const f32x2 o = f32x2{__uint_as_float(e0[0]), __uint_as_float(e1[0])} * gelu_erf2(f32x2{__uint_as_float(e0[1]), __uint_as_float(e1[1])});
                pk[i][j] = pack_bf2(o[0], o[1]);
            }
        }
    }
    __syncthreads();   // every wave is done with the last K tile: the ring is dead
    if (ts && lane == 0) ts[4] = __builtin_readcyclecounter();
    const bool is_val = cg < 2;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r = wm * TM + i * 16 + m_in;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (EPI == EPI_GEGLU) {
                // value lane: outputs 0, 1 of its four columns; gate lane (cg - 2 names the same columns): outputs 2, 3
                const int oc = (wn * TN + j * 16) / 2 + 4 * (cg & 1) + (is_val ? 0 : 2);   // inner index within the tile
                *reinterpret_cast<uint32_t*>(tile + r * PITCH + oc) = pk[i][j];
            } else {
                const int c = wn * TN + j * 16 + 4 * cg;
                uint2 o;
                o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
                o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
                *reinterpret_cast<uint2*>(tile + r * PITCH + c) = o;
            }
        }
    }
    if (ts && lane == 0) ts[5] = __builtin_readcyclecounter();
    __syncthreads();
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + (EPI == EPI_PARTIAL ? (long)z * a.slab_stride : 0);
    copy_out_bf16<NT>(tile, PITCH, BM, OC, out, a.ldo, row0, EPI == EPI_GEGLU ? col0 / 2 : col0, a.M, EPI == EPI_GEGLU ? a.N / 2 : a.N, a.wt, tid);
}

// EPI_RESID of the k-split schedule (batched prompts, M > 2048: 128 x 144 tiles fill the 256 CUs in one round at M = 4000; at M = 1000 the
// K-split-inside-the-workgroup kernel k_gemm_ks is the producer): the un-split residual projection.  h_new = resid + gate * (acc + bias) (fp32,
// stored), its (sum, sum of squares) over the tile's BN columns (part-major table, GemmArgs.zstat_out) and A' = bf16(h_new * zg), the operand
// of the NEXT GEMM, whose epilogue finishes the LayerNorm (GemmArgs.z*).  After the k-split exchange a wave holds 16 rows x ALL BN columns
// of the tile (WN == 1): a row's statistics are an in-lane sum over FN fragments plus two xor-shuffles.  The per-column vectors (bias, gate,
// LayerNorm gain) were parked in LDS behind the ring (`vec`: [3][BN]); the residual rows `r4` were requested right after the K loop.
// DUAL (GemmArgs.zd, see k_gemm_ks): the rows outside [act_row0, act_row1) also get the constant cross-attention-out vector of their batch element
// (`d4`, requested with the residual rows) and the gain of the GEGLU GEMM's LayerNorm (`vec` row 3) instead of the q projection's.
// COPY2 / ZIN (round 6; the forms of k_gemm_ks, gemm_ks.h): COPY2 -- a second operand bf16(h_new * zg2) -> zu2 (`vec` row 3 holds zg2): the in-blocks' MLP-out writes the skip half of the
// out-block's [x | skip] operand; ZIN -- the launch is skip_linear and finishes LN_2D([x | skip]) first: acc := r (acc - mu G') + C' with G' in `vec` row 1 (the gate's), C' in row 0
// (the bias's) and (mu, r) from `zst`: parts cg and cg + 4 of the two statistics sets for each of this lane's rows, requested with the residual rows
template <int BM, int BN, int FM, int FN, int TM, int TN, int NT, bool GATE, bool RES, bool DUAL, bool COPY2 = false, bool ZIN = false>
__device__ __forceinline__ void pp_store_resid(const GemmArgs& a, f32x4 (&acc)[FM][FN], char* smem, int row0, int col0, int wm, int lane, int tid,
                                               const float* vec, const float4 (&r4)[FM][FN], const float4 (&d4)[FM][FN], const float2 (&zst)[FM][4]) {
    static_assert(!ZIN || (!GATE && !RES && !DUAL), "ZIN: the gate's row of `vec` carries G'");
    static_assert(TN == BN, "one wave holds whole tile rows");
    constexpr int PITCH = BN + 8;
    static_assert(BN % 8 == 0, "16-byte row chunks");
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
    const int m_in = lane & 15, cg = lane >> 4;
    const int tn = col0 / BN;
    float* out = reinterpret_cast<float*>(a.out);
    uint2 pk[FM][FN], pk2[COPY2 ? FM : 1][COPY2 ? FN : 1];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = row0 + wm * TM + i * 16 + m_in;
        const bool alt = DUAL && (row < a.act_row0 || row >= a.act_row1);
        float zr = 1.f, zrm = 0.f;
        if constexpr (ZIN) {   // the row's statistics over the zD columns of both sets: the 4 lanes (cg) of a row sit 16 lanes apart; fixed order, bit-reproducible
            const bool one = cg < a.zparts, two = cg + 4 < a.zparts;   // (fewer than 4 parts: the lanes beyond them contribute nothing)
            float zs = ((one ? zst[i][0].x : 0.f) + (two ? zst[i][1].x : 0.f)) + ((one ? zst[i][2].x : 0.f) + (two ? zst[i][3].x : 0.f));
            float zq = ((one ? zst[i][0].y : 0.f) + (two ? zst[i][1].y : 0.f)) + ((one ? zst[i][2].y : 0.f) + (two ? zst[i][3].y : 0.f));
            zs += __shfl_xor(zs, 16, 64); zs += __shfl_xor(zs, 32, 64);
            zq += __shfl_xor(zq, 16, 64); zq += __shfl_xor(zq, 32, 64);
            const float inv_d = __builtin_amdgcn_rcpf((float)a.zD);
            const float mu = zs * inv_d;
            zr = rsqrtf(fmaxf(fmaf(zq, inv_d, -mu * mu), 0.f) + a.zeps);
            zrm = zr * mu;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int cl = j * 16 + 4 * cg, col = col0 + cl;
            const bool ok = col < a.N;
            const float4 b4 = *reinterpret_cast<const float4*>(vec + cl);
            // h_new = resid + gate * (acc + bias): the same two roundings per element as the row kernel (rowbody.h)
            float4 x = make_float4(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
            if constexpr (ZIN) {   // r (acc - mu G') + C'
                const float4 g4 = *reinterpret_cast<const float4*>(vec + BN + cl);
                x = make_float4(fmaf(zr, acc[i][j][0], fmaf(-zrm, g4.x, b4.x)), fmaf(zr, acc[i][j][1], fmaf(-zrm, g4.y, b4.y)),
                                fmaf(zr, acc[i][j][2], fmaf(-zrm, g4.z, b4.z)), fmaf(zr, acc[i][j][3], fmaf(-zrm, g4.w, b4.w)));
            }
            if constexpr (GATE) {
                const float4 g4 = *reinterpret_cast<const float4*>(vec + BN + cl);
                x.x *= g4.x; x.y *= g4.y; x.z *= g4.z; x.w *= g4.w;
            }
            if constexpr (RES) { x.x += r4[i][j].x; x.y += r4[i][j].y; x.z += r4[i][j].z; x.w += r4[i][j].w; }
            if constexpr (DUAL) { x.x += alt ? d4[i][j].x : 0.f; x.y += alt ? d4[i][j].y : 0.f; x.z += alt ? d4[i][j].z : 0.f; x.w += alt ? d4[i][j].w : 0.f; }
            if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
            s1 += (x.x + x.y) + (x.z + x.w);
            s2 = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, s2))));
            if (ok && row < a.M && out) {   // (null out: nothing reads the fp32 stream behind this launch -- the MLP-out in front of an out-block)
                float* dst = out + (long)row * a.ldo + col;
                if (a.wt) st16_wt(dst, x); else *reinterpret_cast<float4*>(dst) = x;
            }
            const float4 z4 = *reinterpret_cast<const float4*>(vec + (alt ? 3 : 2) * BN + cl);
            pk[i][j].x = pack_bf2(x.x * z4.x, x.y * z4.y);
            pk[i][j].y = pack_bf2(x.z * z4.z, x.w * z4.w);
            if constexpr (COPY2) {
                const float4 y4 = *reinterpret_cast<const float4*>(vec + 3 * BN + cl);
                pk2[i][j].x = pack_bf2(x.x * y4.x, x.y * y4.y);
                pk2[i][j].y = pack_bf2(x.z * y4.z, x.w * y4.w);
            }
        }
        // the 4 lanes (cg) of a row sit 16 lanes apart
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (cg == 0 && row < a.M) a.zstat_out[(long)tn * a.zs_stride + row] = make_float2(s1, s2);
    }
    __syncthreads();   // the exchange area of the k-split schedule is dead: park A'
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
            *reinterpret_cast<uint2*>(tile + (wm * TM + i * 16 + m_in) * PITCH + j * 16 + 4 * cg) = pk[i][j];
    __syncthreads();
    copy_out_bf16<NT>(tile, PITCH, BM, BN, a.zu, a.ld_zu, row0, col0, a.M, a.N, a.wt, tid);
    if constexpr (COPY2) {   // the second operand through the same staging tile
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<uint2*>(tile + (wm * TM + i * 16 + m_in) * PITCH + j * 16 + 4 * cg) = pk2[i][j];
        __syncthreads();
        copy_out_bf16<NT>(tile, PITCH, BM, BN, a.zu2, a.ld_zu2, row0, col0, a.M, a.N, a.wt, tid);
    }
}

// q / k epilogue through an fp32 park (what k_headnorm does on the fp32 projection; attention.py:137-142, rotary.py:6-18): the tile holds NH = BN / head_dim
// WHOLE heads of q or of k in NATURAL channel order, parked in LDS as fp32 so that head boundaries need not coincide with MFMA fragments; per-head
// LayerNorm (+ RoPE) -> [B][H][Lp][DQK].  Since round 6 only the stand-alone cross-attention q projection of large grids takes it (api.hip q2_pp: its
// weight rows are in natural order, like the context K it meets); the fused q | k | v projection runs pp_store_qkv_reg below.
template <int BM, int BN, int DH, int FM, int FN, int TM, int TN, int NT, bool ZC>
__device__ __forceinline__ void pp_store_qkv(const GemmArgs& a, f32x4 (&acc)[FM][FN], char* smem, int row0, int col0, int wm, int wn, int lane, int tid,
                                             const float2* zmr, const float* zgc, int slot0) {
    constexpr int NH = BN / DH, DQK = DH == 72 ? 80 : 64, PITCH = BN + 4;
    static_assert(NH * DH == BN && DH % 4 == 0 && (DH * 2) % 16 == 0, "whole heads");
    float* tile = reinterpret_cast<float*>(smem);                        // [BM][PITCH] fp32, reuses the ring
    bf16_t* qk_st = reinterpret_cast<bf16_t*>(smem + BM * PITCH * 4);    // [BM][NH][DH] bf16: normalised q / k heads on their way out
    static_assert((BM * PITCH * 4) % 16 == 0, "staging alignment");
    const int m_in = lane & 15, cg = lane >> 4;
    if constexpr (ZC) {   // LayerNorm algebra: the projection of LN(x) g + c is  r (acc - mu G') + C'
        const int ncl = a.N - 4;
        float4 c4[FN], g4[FN];
        // shared modulation slot: G' / C' of the tile's columns were parked behind the ring by LDS-DMA (k_gemm_pp, z_late_load).  The LDS and the global
        // variant are two separate loops on purpose: as one loop with a per-element choice hipcc merged them into FLAT loads of a selected
        // address (18 flat_load_dwordx4 in the epilogue: +2.5 us per launch)
        if (!a.row_slot) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                g4[j] = *reinterpret_cast<const float4*>(zgc + (wn * TN + j * 16 + 4 * cg));
                c4[j] = *reinterpret_cast<const float4*>(zgc + BN + (wn * TN + j * 16 + 4 * cg));
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int rl = wm * TM + i * 16 + m_in;
            if (a.row_slot) {   // per-row timesteps: the slot differs between rows
                int row = row0 + rl;
                row = row < a.M ? row : a.M - 1;
                const long so = (long)z_slot(a, slot0, row) * a.zt_slot_stride;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    int cp = col0 + wn * TN + j * 16 + 4 * cg;
                    cp = cp < ncl ? cp : ncl;
                    g4[j] = *reinterpret_cast<const float4*>(a.zG + so + cp);
                    c4[j] = *reinterpret_cast<const float4*>(a.zC + so + cp);
                }
            }
            const float2 mr = zmr[i];
            const float r = mr.y, rm = mr.y * mr.x;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc[i][j][0] = fmaf(r, acc[i][j][0], fmaf(-rm, g4[j].x, c4[j].x));
                acc[i][j][1] = fmaf(r, acc[i][j][1], fmaf(-rm, g4[j].y, c4[j].y));
                acc[i][j][2] = fmaf(r, acc[i][j][2], fmaf(-rm, g4[j].z, c4[j].z));
                acc[i][j][3] = fmaf(r, acc[i][j][3], fmaf(-rm, g4[j].w, c4[j].w));
            }
        }
    }
    const HeadNormArgs& hn = a.hn;
    const float rcpL = __builtin_amdgcn_rcpf((float)hn.L);
    const int D = hn.H * DH;
    const int part = col0 / D;                 // 0 q, 1 k, 2 v
    const int head0 = (col0 % D) / DH;         // first head of this tile
    // q / k tiles: 4 lanes per (row, head), each owns E = DH / 4 contiguous channels; a thread's items are it = tid + NT k (same sub-lane every time).
    // Its LayerNorm weights and the RoPE rows of its first item are requested HERE, in front of the two barriers and the fp32 park, the other items'
    // rows right behind the park -- not inside the item loop behind the LayerNorm arithmetic: there every item waited a full L2 round trip for 2 E table values (in-situ stamps: the q / k
    // tiles' epilogue took 17K cycles against 10K of the v tiles, and the kernel ends with its slowest workgroup)
    constexpr int E = DH / 4, ITEMS = BM * NH * 4 / NT;
    static_assert(ITEMS * NT == BM * NH * 4 && E % 2 == 0, "whole items per thread, float2 pieces");
    const int sub = tid & 3;
    const bool rope = part < 2 && hn.rope_cos != nullptr;
    float2 wv[E / 2], bv[E / 2], csv[ITEMS][E / 2], snv[ITEMS][E / 2];
    // behind the LayerNorm-algebra arithmetic above, not in front of it: with per-row timesteps that arithmetic waits for its own global loads, and
    // the (static) vmcnt in front of it would drain these requests too on every path.  hipcc sinks the arithmetic towards the park below (past a
    // sched_barrier as well): the empty asm "modifies" the accumulators here, and its memory clobber keeps the loads below it
    if constexpr (ZC) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]) : : "memory");
    }
    auto park = [&]() {   // fp32 tile -> LDS (the ring is dead behind the barrier)
        __syncthreads();                                                  // every wave is done with the ring
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<f32x4*>(tile + (wm * TM + i * 16 + m_in) * PITCH + wn * TN + j * 16 + 4 * cg) = acc[i][j];
    };
    // ONE region for the q / k tiles (barriers and park duplicated in the v branch, `part` is uniform over the workgroup): hipcc's vmcnt bookkeeping is
    // path-insensitive -- with the requests and their uses in separate `if (part < 2)` blocks it assumed the later requests might not have been
    // issued and drained all of them at the first use
    if (part < 2) {
        const float2* w2 = reinterpret_cast<const float2*>((part == 0 ? hn.qn_w : hn.kn_w) + sub * E);
        const float2* b2 = reinterpret_cast<const float2*>((part == 0 ? hn.qn_b : hn.kn_b) + sub * E);
#pragma unroll
        for (int i = 0; i < E / 2; ++i) { wv[i] = w2[i]; bv[i] = b2[i]; }
        // item k's cos / sin pieces.  Without RoPE (the cross-attention q projection) the same loads read the LayerNorm weight instead and go unused
        // (a conditional load between a request and its use has the same effect on the vmcnt bookkeeping)
        auto rope_rows = [&](int k) {
            const int m = row0 + ((tid + k * NT) >> 2) / NH;
            int b_, l;
            divmod_rows(m < a.M ? m : a.M - 1, hn.L, rcpL, b_, l);
            const float* w = part == 0 ? hn.qn_w : hn.kn_w;
            const float2* c2 = reinterpret_cast<const float2*>((rope ? hn.rope_cos + (long)l * (DH / 2) : w) + (sub & 1) * E);
            const float2* s2 = reinterpret_cast<const float2*>((rope ? hn.rope_sin + (long)l * (DH / 2) : w) + (sub & 1) * E);
#pragma unroll
            for (int i = 0; i < E / 2; ++i) { csv[k][i] = c2[i]; snv[k][i] = s2[i]; }
        };
        rope_rows(0);
        park();
        // the accumulators are parked: their registers take the later items' rows (all of them up front did not fit: scratch spills)
#pragma unroll
        for (int k = 1; k < ITEMS; ++k) rope_rows(k);
        __syncthreads();
        // LN via two xor-shuffles inside the quad, RoPE partner (i +- DH/2) in lane ^ 2: DPP, not __shfl_xor (ds_bpermute: 22 LDS-pipe round trips per item)
        bf16_t* dstbase = part == 0 ? hn.q : hn.k;
        const float sign = (sub & 2) ? 1.f : -1.f;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int it = tid + k * NT;
            const int hh = (it >> 2) % NH, r = (it >> 2) / NH;
            const float* src = tile + r * PITCH + hh * DH + sub * E;
            float v[E];
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                const float2 t2 = *reinterpret_cast<const float2*>(src + 2 * i);
                v[2 * i] = t2.x; v[2 * i + 1] = t2.y;
            }
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < E; ++i) s1 += v[i];
            s1 += quad_xor1(s1);
            s1 += quad_xor2(s1);
            const float mean = s1 * (1.f / DH);
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < E; ++i) { const float d = v[i] - mean; q2 += d * d; }
            q2 += quad_xor1(q2);
            q2 += quad_xor2(q2);
            const float rstd = rsqrtf(q2 * (1.f / DH) + 1e-5f);
#pragma unroll
            for (int i = 0; i < E / 2; ++i) {
                v[2 * i] = (v[2 * i] - mean) * rstd * wv[i].x + bv[i].x;
                v[2 * i + 1] = (v[2 * i + 1] - mean) * rstd * wv[i].y + bv[i].y;
            }
            if (rope) {
#pragma unroll
                for (int i = 0; i < E / 2; ++i) {
                    const float o0 = quad_xor2(v[2 * i]), o1 = quad_xor2(v[2 * i + 1]);
                    v[2 * i] = v[2 * i] * csv[k][i].x + sign * o0 * snv[k][i].x;
                    v[2 * i + 1] = v[2 * i + 1] * csv[k][i].y + sign * o1 * snv[k][i].y;
                }
            }
            bf16_t* dst = qk_st + (r * NH + hh) * DH + sub * E;   // written out below as whole 16-byte chunks (a head is 9 or 8 of them)
#pragma unroll
            for (int i = 0; i < E / 2; ++i) *reinterpret_cast<uint32_t*>(dst + 2 * i) = pack_bf2(v[2 * i], v[2 * i + 1]);
        }
        __syncthreads();
        constexpr int CP = DH * 2 / 16;
        for (int q = tid; q < BM * NH * CP; q += NT) {
            const int c8 = q % CP, hh = (q / CP) % NH, r = q / (CP * NH);
            const int m = row0 + r;
            if (m < a.M) {
                int b, l;
                divmod_rows(m, hn.L, rcpL, b, l);
                bf16_t* dst = dstbase + (((long)b * hn.H + head0 + hh) * hn.Lp + l) * DQK + c8 * 8;
                const uint4 v = *reinterpret_cast<const uint4*>(qk_st + (r * NH + hh) * DH + c8 * 8);
                if (a.wt) st16_wt(dst, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
                else *reinterpret_cast<uint4*>(dst) = v;
            }
        }
    }
}

// channel permutation of the fused q | k projection (EZDIT_T_QKROPE, include/ezdit.h; the packer applies it to the q and k rows of `wqkv`).  Column c of a
// two-head tile (c = 16 j + 4 cg + 2 e + s: fragment j, lane group cg, pair e, half s) holds channel  f + (dh / 2) s  of head hh, where the RoPE pair
// index f = 8 jj + 2 cg + e runs over a head's full fragments jj and, for head_dim 72 (4.5 fragments per head), the middle fragment is split between the
// heads by lane group: cg 0, 1 -> head 0, cg 2, 3 -> head 1, f = 32 + 2 (cg & 1) + e.  So in the 16x16 MFMA C layout (lane = row m_in + 16 cg, four consecutive
// columns) a lane holds, per fragment, TWO COMPLETE RoPE PAIRS (c, c + 1) of ONE head: LayerNorm statistics are in-lane sums plus two lane-swap steps over cg,
// RoPE is in-lane -- no fp32 park, no column-wise LDS reads.  Head hh owns the tile columns [dh hh, dh hh + dh): q and k are stored in that column order
// ([B][H][Lp][DQK]; q . k^T does not care as long as both use the same order, and head h sits at the same tile position h % 2 in q and in k).
__host__ __device__ inline void qkrope_col(int dh, int c, int& hh, int& ch) {
    const int j = c >> 4, cg = (c >> 2) & 3, e = (c >> 1) & 1, s = c & 1, FH = dh / 16;
    int f;
    if (dh % 16 == 0) { hh = j / FH; f = 8 * (j % FH) + 2 * cg + e; }
    else if (j < FH) { hh = 0; f = 8 * j + 2 * cg + e; }
    else if (j == FH) { hh = cg >> 1; f = 8 * FH + 2 * (cg & 1) + e; }
    else { hh = 1; f = 8 * (j - FH - 1) + 2 * cg + e; }
    ch = f + (dh / 2) * s;
}

// sums over the four lanes m_in + 16 cg that share an output row: v_permlane16_swap / v_permlane32_swap (VALU; __shfl_xor would be two LDS-pipe round trips)
__device__ __forceinline__ float row4_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// ---- LayerNorm algebra, consumer side: (mu, r) of the rows a LANE finishes in the 16x16 C layout (lane = row m_in + 16 cg of every 16-row fragment), computed by the lane itself.
// The four lanes cg = 0 .. 3 of a row each fetch its parts cg, cg + 4, cg + 8 of the part-major statistics table (blind loads: ld8_blind) and add them up in the order of
// z_row_stats_finish (common.h): own parts in index order, then the lane pairs (cg, cg ^ 1), then the two pairs -- bit-identical to the table-per-workgroup form it replaces.
// Why per lane: the epilogue is the only reader of (mu, r).  Merged per workgroup in front of the K loop (rounds 4 - 6a) they sat between the prologue's LDS-DMA and the loop's
// first barrier -- the statistics' cold fetch, ~40 VALU instructions and an LDS store that every wave waited for: 2.3K cycles per consumer launch (profiles/r06_experiments.txt,
// r06r: bound -2.4 % of the step).  Now they are requested when the K loop enters its last PD tiles (the counted waits of those steps leave them in flight) and merged behind the loop.
template <int FMR>
struct ZLaneRegs { f32x2 v[FMR][Z_PT]; };
template <int FMR>
__device__ __forceinline__ void z_lane_load(const float2* zstat_in, long stride, int parts, int first_row /* of fragment 0, this lane */, int max_row, int cg, ZLaneRegs<FMR>& z) {
#pragma unroll
    for (int i = 0; i < FMR; ++i) {
        int row = first_row + 16 * i;
        row = row < max_row ? row : max_row;
#pragma unroll
        for (int k = 0; k < Z_PT; ++k) {
            const int p = cg + 4 * k;
            z.v[i][k] = ld8_blind(zstat_in + row + (p < parts ? p : parts - 1) * stride);
        }
    }
}
// behind `s_waitcnt vmcnt(0)` (the caller's): the registers are first named here
template <int FMR>
__device__ __forceinline__ void z_lane_finish(ZLaneRegs<FMR>& z, int parts, int cg, int D, float eps, float2 (&mr)[FMR]) {
    const float inv_d = __builtin_amdgcn_rcpf((float)D);
#pragma unroll
    for (int i = 0; i < FMR; ++i) {
#pragma unroll
        for (int k = 0; k < Z_PT; ++k) asm volatile("" : "+a"(z.v[i][k]));
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < Z_PT; ++k) {   // fixed order: bit-reproducible
            const bool on = cg + 4 * k < parts;
            s += on ? z.v[i][k][0] : 0.f;
            q += on ? z.v[i][k][1] : 0.f;
        }
        s = row4_sum(s);
        q = row4_sum(q);
        const float mu = s * inv_d;
        const float var = fmaxf(fmaf(q, inv_d, -mu * mu), 0.f);
        mr[i] = make_float2(mu, rsqrtf(var + eps));
    }
}

// fused q | k | v epilogue IN REGISTERS (round 6; attention.py:137-142, rotary.py:6-18): the tile holds two whole heads of q, of k (columns permuted as above) or
// of v (natural order).  q / k: [LayerNorm algebra] -> per-head LayerNorm -> RoPE on the accumulators; everything leaves as bf16 through a staging tile
// (whole 16-byte row chunks): q, k -> [B][H][Lp][DQK], v -> [B][H][Lp][DV].
// per-row global operands of the register epilogue of ONE lane: the RoPE table values of its FM rows at its pair indices (f = 8 jj + 2 cg + e for the full
// fragments jj and 8 FH + 2 (cg & 1) + e for the split middle one; the same for both heads).  Requested by qkv_request() right behind the K loop --
// unconditionally, clamped, v tiles included (one code path, no conditional requests) -- so that they land under the k-split exchange; the LayerNorm affine
// of the lane's channels (L2-hot: one vector per block) is requested at the top of the epilogue proper and lands under its reduction passes
template <int DH, int FM>
struct QkvOperands {
    static constexpr int FH = DH / 16, MID = (DH % 16) != 0, NF = FH + MID;
    f32x2 cs[FM][NF], sn[FM][NF];   // (x, y) = the table values of the pair indices f, f + 1
    float aff;                      // this THREAD's element of the workgroup's LayerNorm-affine table (threads < 32 NF; qkv_aff_*)
};
// LayerNorm affine of a q / k tile as the lanes need it, [4 lane groups cg][NF pieces q][8]: (w[f], w[f + 1], w[H + f], w[H + f + 1], b[f], b[f + 1], b[H + f], b[H + f + 1])
// with f the pair index of (cg, q) and H = dh / 2 -- 32 NF floats per workgroup, parked in LDS behind the per-column vectors: held in registers they were 8 NF = 40 per lane
template <int DH>
__device__ __forceinline__ int qkv_aff_index(int t /* 0 .. 32 NF - 1 */) {   // element of the [dh] affine vector that table entry t holds (e >= 4: of the bias)
    constexpr int FH = DH / 16, NF = FH + ((DH % 16) != 0);
    const int e = t & 7, q = (t >> 3) % NF, cg = t / (8 * NF);
    const int f = q < FH ? 8 * q + 2 * cg : 8 * FH + 2 * (cg & 1);
    return ((e & 2) ? DH / 2 : 0) + f + (e & 1);
}
template <int DH, int FM>
__device__ __forceinline__ void qkv_request(const GemmArgs& a, int col0, int first_row /* of this lane's fragment 0 */, int lane, int tid, QkvOperands<DH, FM>& op) {
    constexpr int HALF = DH / 2, FH = DH / 16, NF = QkvOperands<DH, FM>::NF;
    const HeadNormArgs& hn = a.hn;
    const int cg = lane >> 4;
    const int part = col0 / (hn.H * DH);
    const bool rope = part < 2 && hn.rope_cos != nullptr;
    const float* w = part == 1 ? hn.kn_w : hn.qn_w;
    {
        const int t = tid < 32 * NF ? tid : 0;
        op.aff = ((t & 4) ? (part == 1 ? hn.kn_b : hn.qn_b) : w)[qkv_aff_index<DH>(t)];
    }
    const float rcpL = __builtin_amdgcn_rcpf((float)hn.L);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = first_row + i * 16;
        m = m < a.M ? m : a.M - 1;
        int b_, l;
        divmod_rows(m, hn.L, rcpL, b_, l);
        const float* ct = (rope ? hn.rope_cos + (long)l * HALF : w);
        const float* st = (rope ? hn.rope_sin + (long)l * HALF : w);
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const int f = q < FH ? 8 * q + 2 * cg : 8 * FH + 2 * (cg & 1);
            // (plain loads: hipcc tracks them; inline-asm loads into AGPRs were tried and are NOT safe here -- the allocator inserted v_accvgpr_mov copies of the
            // destinations between issue and wait (stale data, xs64 golden off by 16 %): the hazard class ADVICE r05 warned about)
            op.cs[i][q] = *reinterpret_cast<const f32x2*>(ct + f);
            op.sn[i][q] = *reinterpret_cast<const f32x2*>(st + f);
        }
    }
}

template <int BM, int BN, int DH, int FM, int FN, int TM, int TN, int NT, bool ZC>
__device__ __forceinline__ void pp_store_qkv_reg(const GemmArgs& a, f32x4 (&acc)[FM][FN], char* smem, int row0, int col0, int wm, int lane, int tid,
                                                 const float2* zmr, const float* zgc, float* aff_l, int slot0, const QkvOperands<DH, FM>& op, unsigned long long* ts = nullptr) {
    constexpr int NH = 2, DQK = DH == 72 ? 80 : 64, DV = DH == 72 ? 96 : 64, FH = DH / 16, MID = (DH % 16) != 0;
    static_assert(NH * DH == BN && TN == BN && FN == 2 * FH + MID && (DH % 16 == 0 || DH % 16 == 8), "two whole heads per tile, a wave holds whole rows");
    constexpr int PITCH = BN + 8;                                       // bf16 elements per staging row
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);                     // [BM][PITCH] bf16, reuses the ring
    if (ts && lane == 0) ts[4] = __builtin_readcyclecounter();          // exchange done
    const int m_in = lane & 15, cg = lane >> 4;
    const HeadNormArgs& hn = a.hn;
    const float rcpL = __builtin_amdgcn_rcpf((float)hn.L);
    const int D = hn.H * DH;
    const int part = col0 / D;                 // 0 q, 1 k, 2 v   (uniform over the workgroup)
    const int head0 = (col0 % D) / DH;         // first head of this tile
    const bool rope = part < 2 && hn.rope_cos != nullptr;
    const auto& cs = op.cs; const auto& sn = op.sn;
    constexpr int NF = QkvOperands<DH, FM>::NF;
    static_assert(32 * NF <= NT, "one table element per thread");
    if (tid < 32 * NF) aff_l[tid] = op.aff;   // (visible behind the barrier below)
    const float4* aff4 = reinterpret_cast<const float4*>(aff_l) + cg * NF * 2;
    // every wave is done with the ring (4-wave form) / has read its partner's partial sums (k-split form): the ring becomes the bf16 staging tile, and each
    // fragment is stored there as soon as it is finished (no register array of packed results)
    __syncthreads();
    const bool mid0 = cg < 2;              // the split fragment's columns of this lane belong to head 0 (else head 1)
    // ONE pass per 16-row fragment i (LayerNorm algebra -> per-head LayerNorm -> RoPE -> staging): only that fragment's FN accumulators are in VGPRs at a time
    // (the 4-wave form has FM = 2: 72 accumulators + the 80 operand registers of both fragments spilled when the phases ran tile-wide)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if constexpr (ZC) {   // LayerNorm algebra: the projection of LN(x) g + c is  r (acc - mu G') + C'
            // G' / C' of the lane's four columns are re-read per fragment (LDS: parked behind the ring by z_late_load; per-row timesteps: global, the slot differs
            // between rows) instead of being held for the whole tile (2 x FN float4 = 72 registers)
            const int ncl = a.N - 4;
            const int rl = wm * TM + i * 16 + m_in;
            const float2 mr = zmr[i];
            const float r = mr.y, rm = mr.y * mr.x;
            if (!a.row_slot) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const float4 g4 = *reinterpret_cast<const float4*>(zgc + (j * 16 + 4 * cg)), c4 = *reinterpret_cast<const float4*>(zgc + BN + (j * 16 + 4 * cg));
                    acc[i][j][0] = fmaf(r, acc[i][j][0], fmaf(-rm, g4.x, c4.x));
                    acc[i][j][1] = fmaf(r, acc[i][j][1], fmaf(-rm, g4.y, c4.y));
                    acc[i][j][2] = fmaf(r, acc[i][j][2], fmaf(-rm, g4.z, c4.z));
                    acc[i][j][3] = fmaf(r, acc[i][j][3], fmaf(-rm, g4.w, c4.w));
                }
            } else {
                int row = row0 + rl;
                row = row < a.M ? row : a.M - 1;
                const long so = (long)z_slot(a, slot0, row) * a.zt_slot_stride;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    int cp = col0 + j * 16 + 4 * cg;
                    cp = cp < ncl ? cp : ncl;
                    const float4 g4 = *reinterpret_cast<const float4*>(a.zG + so + cp), c4 = *reinterpret_cast<const float4*>(a.zC + so + cp);
                    acc[i][j][0] = fmaf(r, acc[i][j][0], fmaf(-rm, g4.x, c4.x));
                    acc[i][j][1] = fmaf(r, acc[i][j][1], fmaf(-rm, g4.y, c4.y));
                    acc[i][j][2] = fmaf(r, acc[i][j][2], fmaf(-rm, g4.z, c4.z));
                    acc[i][j][3] = fmaf(r, acc[i][j][3], fmaf(-rm, g4.w, c4.w));
                }
            }
        }
        bf16_t* trow = tile + (wm * TM + i * 16 + m_in) * PITCH + 4 * cg;
        if (part < 2) {
            // per-head LayerNorm: two passes (mean, then the centred second moment, as the row kernels do), each an in-lane sum + row4_sum
            float s0 = 0.f, s1 = 0.f, sm = 0.f;
#pragma unroll
            for (int j = 0; j < FH; ++j) {
                s0 += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                s1 += (acc[i][FH + MID + j][0] + acc[i][FH + MID + j][1]) + (acc[i][FH + MID + j][2] + acc[i][FH + MID + j][3]);
            }
            if constexpr (MID) sm = (acc[i][FH][0] + acc[i][FH][1]) + (acc[i][FH][2] + acc[i][FH][3]);
            const float mean0 = row4_sum(s0 + (mid0 ? sm : 0.f)) * (1.f / DH), mean1 = row4_sum(s1 + (mid0 ? 0.f : sm)) * (1.f / DH);
            float q0 = 0.f, q1 = 0.f, qm = 0.f;
            const float meanm = mid0 ? mean0 : mean1;
#pragma unroll
            for (int j = 0; j < FH; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float d0 = acc[i][j][x] - mean0, d1 = acc[i][FH + MID + j][x] - mean1;
                    q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1);
                }
            if constexpr (MID) {
#pragma unroll
                for (int x = 0; x < 4; ++x) { const float d = acc[i][FH][x] - meanm; qm = fmaf(d, d, qm); }
            }
            const float rstd0 = rsqrtf(row4_sum(q0 + (mid0 ? qm : 0.f)) * (1.f / DH) + 1e-5f), rstd1 = rsqrtf(row4_sum(q1 + (mid0 ? 0.f : qm)) * (1.f / DH) + 1e-5f);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bool is_mid = MID && j == FH;
                const int q = is_mid ? FH : (j < FH ? j : j - FH - MID);          // index of the lane's affine / table piece (compile-time)
                const float mean = is_mid ? meanm : (j < FH ? mean0 : mean1);
                const float rstd = is_mid ? (mid0 ? rstd0 : rstd1) : (j < FH ? rstd0 : rstd1);
                // columns 4 cg + {0, 1, 2, 3} = (pair f: first half, second half), (pair f + 1: first half, second half)
                const float4 w4 = aff4[2 * q], b4 = aff4[2 * q + 1];   // (w[f], w[f + 1], w[H + f], w[H + f + 1]), the same of the bias
                float y0 = (acc[i][j][0] - mean) * rstd * w4.x + b4.x, y1 = (acc[i][j][1] - mean) * rstd * w4.z + b4.z;
                float y2 = (acc[i][j][2] - mean) * rstd * w4.y + b4.y, y3 = (acc[i][j][3] - mean) * rstd * w4.w + b4.w;
                if (rope) {   // rotary.py:6-8 (half split): first half x1 c - x2 s, second half x2 c + x1 s
                    const float o0 = y0 * cs[i][q][0] - y1 * sn[i][q][0], o1 = y1 * cs[i][q][0] + y0 * sn[i][q][0];
                    const float o2 = y2 * cs[i][q][1] - y3 * sn[i][q][1], o3 = y3 * cs[i][q][1] + y2 * sn[i][q][1];
                    y0 = o0; y1 = o1; y2 = o2; y3 = o3;
                }
                *reinterpret_cast<uint2*>(trow + j * 16) = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
            }
        } else {
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<uint2*>(trow + j * 16) = make_uint2(pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3]));
        }
    }
    if (ts && lane == 0) ts[5] = __builtin_readcyclecounter();   // arithmetic done, tile staged
    __syncthreads();

    // whole 16-byte chunks: head hh of row r = the DH / 8 chunks at tile columns [DH hh, DH hh + DH) -> one row of q / k ([.][DQK]) or of v ([.][DV])
    constexpr int CP = DH / 8;
    bf16_t* dstbase = part == 0 ? hn.q : part == 1 ? hn.k : hn.v;
    const int pitch = part < 2 ? DQK : DV;
    for (int q = tid; q < BM * NH * CP; q += NT) {
        const int c8 = q % CP, hh = (q / CP) % NH, r = q / (CP * NH);
        const int m = row0 + r;
        if (m < a.M) {
            int b, l;
            divmod_rows(m, hn.L, rcpL, b, l);
            bf16_t* dst = dstbase + (((long)b * hn.H + head0 + hh) * hn.Lp + l) * pitch + c8 * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(tile + r * PITCH + hh * DH + c8 * 8);
            if (a.wt) st16_wt(dst, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
            else *reinterpret_cast<uint4*>(dst) = v;
        }
    }
}

// VAR bits: 64 = LayerNorm algebra in the epilogue (EPI_GEGLU / EPI_QKV consumers) -- a template
// switch, not a run-time test, so that the epilogue's loads are straight-line;
// timing ablations (results are garbage; EZ_ABLATE builds): 8 = no MFMAs, 16 = no fragment reads, 32 = no LDS-DMA refill inside the loop.  Measured and dropped (MI355X, 128x288 tile, cycles per K tile): s_setprio(1) over the MFMA phase 1809 vs 1793,
// over the LOAD phase 1806, static priority for group 1 1819 -- priorities do not move this loop.
// GemmArgs.ts (test hook, nullable): wave 0 of every workgroup records s_memtime at kernel start, loop start, loop end, kernel end
// dynamic LDS of k_gemm_pp: ring | (mu, r) per row | per-column vectors | EPI_QKV: 256 bytes per wave that the RoPE-table warm-up DMA lands in
template <int BM, int BN, int NS, int EPI>
constexpr int pp_smem_bytes() {
    return NS * ((BM + BN + 31) / 32) * 4096 + BM * 8 + (EPI == EPI_RESID ? 4 : 2) * BN * 4 + (EPI == EPI_QKV ? 8 * 256 + 1024 : 0);   // EPI_QKV: + the RoPE warm-up sink + the LayerNorm-affine table (qkv_aff_*)
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, int SCHED, int VAR>
__global__ __launch_bounds__(512) void k_gemm_pp(GemmArgs a) {
    static_assert(SCHED == 1 || SCHED == 2, "schedule");
    static_assert(WM * WN == (SCHED == 1 ? 8 : 4), "SCHED 1: 8 waves tile the block; SCHED 2: each group of 4 tiles it");
    constexpr int NT = 512, NTG = 256;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(BM % 32 == 0 && TM % 16 == 0 && TN % 16 == 0 && TM * WM == BM && TN * WN == BN, "tile geometry");
    static_assert(SCHED == 1 || FM % 2 == 0, "SCHED 2 splits the wave tile's row fragments between the groups after the loop");
    constexpr int NP = (BM + BN + 31) / 32;    // 4-KB pieces (32 rows) per stage
    constexpr int PA = BM / 32;                // pieces [0, PA) come from A, [PA, NP) from W
    constexpr int STAGE = NP * 4096;
    constexpr int PD = NS - 1;                 // prefetch distance in K tiles
    static_assert(NS >= 3 && NS <= 6, "ring depth");
    static_assert(pp_smem_bytes<BM, BN, NS, EPI>() <= 160 * 1024, "LDS budget of a CU (ring + per-row LayerNorm statistics + G' / C' (or bias / gate / gain) of the tile's columns [+ EPI_QKV: sink of the table warm-up])");
    constexpr int P0 = SCHED == 1 ? (NP + 1) / 2 : NP;   // group 0's pieces of a tile: [0, P0); group 1: [P0, NP) (SCHED 2: the issuing group takes all)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // the kernel arguments the prologue needs, requested in ONE batch in front of the tile map (common.h "Kernel-argument batch"), then the device step
    // counter as a plain (scalar) load: nothing in front of the first LDS-DMA waits on the scalar-memory counter again, so its round trip runs under the
    // tile map, the address arithmetic and the first K tile
    int M_ = a.M;
    asm("" : "+s"(M_) : "s"(a.A), "s"(a.W), "s"(a.lda), "s"(a.ldw), "s"(a.wrows), "s"(a.N), "s"(a.K), "s"(a.splitk), "s"(a.pm), "s"(a.pn), "s"(a.bm), "s"(a.bn), "s"(a.bz),
        "s"(a.cur_step), "s"(a.ts), "s"(a.xcd_panel), "s"(a.row_slot), "s"(a.mbm), "s"(a.mbn), "s"(a.msplit), "s"(a.mG));
    const int slot0 = a.cur_step ? *a.cur_step : 0;   // device step counter: needed from z_late_load on
    // G' / C' table bases PINNED in SGPRs: the per-thread choice between them (z_late_load) must be a select on two scalars -- left to hipcc it became a vector load of the chosen
    // pointer from the argument segment, and its `s_waitcnt vmcnt(0)` drained the first K tile's LDS-DMA in front of the G' / C' request (ISA of the first blind-load build of k_gemm_co)
    const float *zG_ = a.zG, *zC_ = a.zC;
    asm("" : "+s"(M_), "+s"(zG_), "+s"(zC_));
    const float *rb_ = a.bias, *rg_ = a.gate, *rz_ = a.zg, *rz2_ = a.zg2;   // the same for the producer's per-column vectors (EPI_RESID)
    if constexpr (EPI == EPI_RESID) asm("" : "+s"(M_), "+s"(rb_), "+s"(rg_), "+s"(rz_), "+s"(rz2_));
    const float *rc_ = a.hn.rope_cos, *rs_ = a.hn.rope_sin;                 // ... and for the RoPE tables the fused-QKV warm-up chooses between per thread
    if constexpr (EPI == EPI_QKV) asm("" : "+s"(M_), "+s"(rc_), "+s"(rs_));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    const int tg = tid & (NTG - 1);
    const int wt = SCHED == 1 ? wave : wg;     // position among the waves that tile the block
    const int wm = wt / WN, wn = wt % WN;

    const int tilesM = (M_ + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    int tm, tn, z;
    if (EPI == EPI_PARTIAL && a.xcd_panel) {
        // panel placement: the tile's slabs and the row kernel that reduces them (row panel p on XCD p % 8) stay inside one XCD's L2
        panel_of_block(a, tilesN, tm, tn, z);
        if (tm >= tilesM) return;
    } else if (!tile_of_block(a, tilesM, tilesN, tm, tn, z)) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = a.K / BK;
    int kb, ke;
    ksplit_range(a, nk, z, kb, ke);
    const int nt = ke - kb;

    // VAR & 64 ("ZM"): EPI_GEGLU / EPI_QKV finish a LayerNorm in their epilogue (GemmArgs.z*, consumer side)
    constexpr bool ZM = (VAR & 64) != 0;
    // LayerNorm algebra, consumer side: the G' / C' slices of the tile's columns go straight into the LDS table `zgc` by LDS-DMA right BEHIND the first K tile (z_late_load) and are not
    // waited for in front of the loop; the partial statistics of the rows a lane finishes are requested by that lane when the loop enters its last PD tiles and merged behind the
    // loop (z_lane_load / z_lane_finish above): nothing of the algebra sits between the prologue and the loop's first barrier any more, no register rides through the steady-state
    // loop, and the epilogue needs no extra barrier.  Ledger of the alternatives: profiles/r04_experiments.txt, r06_experiments.txt (r06o, r06r - r06v).
    float* zgc = reinterpret_cast<float*>(smem + NS * STAGE + BM * 8);   // [2][BN]: G' | C' of this tile's columns (shared modulation slot only); EPI_RESID: [4][BN] bias | gate | LayerNorm gain | DUAL: the alternative gain
    constexpr bool ZMC = ZM && (EPI == EPI_GEGLU || EPI == EPI_QKV);     // this instantiation finishes a LayerNorm in its epilogue
    constexpr int FMR = SCHED == 1 ? FM : FM / 2;                        // 16-row fragments a lane finishes in the epilogue (SCHED 2: after the k-split exchange)
    constexpr int NZT = ZMC ? FMR * Z_PT : 0;                            // blind statistics loads per lane (into AGPRs), issued by z_late_load
    ZLaneRegs<FMR> zlr;
    const bool z_shared_slot = a.row_slot == nullptr;
    // threads that fetch one float4 of the epilogue's per-column vectors (LDS-DMA in z_late_load), and whether THIS wave is one of theirs: its counted wait in front of the
    // K loop then leaves that one DMA in flight (nt >= 2: the wait for K tile 1, which is younger, covers it)
    constexpr int ZNV = EPI == EPI_RESID ? ((VAR & (256 | 512)) ? 4 : 3) * (BN / 4) : 2 * (BN / 4);
    // zx = LDS-DMA instructions THIS wave issues in z_late_load on top of the NZT statistics loads every wave issues there (0 .. 2, wave-uniform): the per-column vectors'
    // (the first waves only) and, fused QKV, the RoPE-table warm-up (every wave of a q / k tile).  The counted wait in front of the K loop leaves all of them in flight;
    // they are older than every K tile but the first, so the wait for K tile 1 (nt >= 2) covers them
    const bool z_warm = EPI == EPI_QKV && rc_ && col0 < 2 * a.hn.H * a.hn.dh;
    const int zx = nt >= 2 ? ((((ZMC && z_shared_slot) || EPI == EPI_RESID) && wave * 64 < ZNV) ? 1 : 0) + (z_warm ? 1 : 0) : -1;   // (-1: a single K tile -- wait for everything)
    // G' / C' live in the table of the CURRENT modulation slot: their address needs the device step counter (slot0, a scalar load issued at
    // the top of the kernel).  Requested right AFTER the prologue's LDS-DMA went out, so that the counter's round trip does not sit in front
    // of the first tile (it did: +1 us per consumer launch)
    constexpr bool RGATE = EPI == EPI_RESID && (VAR & 64) != 0, RRES = EPI == EPI_RESID && (VAR & 128) != 0, RDUAL = EPI == EPI_RESID && (VAR & 256) != 0;
    constexpr bool RCOPY2 = EPI == EPI_RESID && (VAR & 512) != 0, RZIN = EPI == EPI_RESID && (VAR & 1024) != 0;   // pp_store_resid: second operand / LayerNorm-finishing skip_linear
    static_assert(!RDUAL || (RGATE && RRES), "DUAL: the gated residual projection only");
    static_assert(!(RDUAL && RCOPY2) && !(RZIN && (RGATE || RRES || RDUAL || RCOPY2)), "forms of the residual epilogue");
    auto z_late_load = [&]() {
        if constexpr (EPI == EPI_QKV) {
            // q / k tiles: the RoPE rows of the tile's 128 tokens (2 x 18 KB of the cos / sin tables) are read in the epilogue, by every workgroup
            // at the same time, and no longer in the L2 by then (the kernels in between stream through it): each thread pulls one end of one row
            // of one table towards its L2 now -- a 4-byte LDS-DMA into a sink, nothing waits for it, no register is held
            const HeadNormArgs& hn = a.hn;
            if (z_warm) {
                const int r = tid >> 2, which = tid & 3;
                int m = row0 + r;
                m = m < a.M ? m : a.M - 1;
                int b_, l_;
                divmod_rows(m, hn.L, __builtin_amdgcn_rcpf((float)hn.L), b_, l_);
                const char* src = reinterpret_cast<const char*>((which & 2) ? rs_ : rc_) + ((long)l_ * (hn.dh / 2) + ((which & 1) ? hn.dh / 2 - 1 : 0)) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + NS * STAGE + BM * 8 + 2 * BN * 4 + wave * 256), 4, 0, 0);
            }
        }
        if constexpr (EPI == EPI_RESID) {   // producer side: bias | gate | gain of the tile's columns, one float4 per thread (the modulation slot is shared: launch_gemm checks)
            static_assert(4 * (BN / 4) <= NT, "one float4 per thread");
            if (tid < ZNV) {   // one 16-byte LDS-DMA per thread, straight into zgc[tid] (wave w's 64 lanes fill the KB at zgc + 1024 w)
                const int which = tid / (BN / 4), t4 = tid - which * (BN / 4);
                int cp = col0 + 4 * t4;
                cp = cp < a.N - 4 ? cp : a.N - 4;
                const float* src = which == 0 ? rb_ : which == 1 ? (RGATE ? rg_ + (long)slot0 * a.gate_slot_stride : RZIN ? zG_ : rb_) : which == 2 ? rz_ + (long)slot0 * a.zg_slot_stride
                                   : ((RDUAL || RCOPY2) ? rz2_ + (long)slot0 * a.zg2_slot_stride : rb_);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + cp),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(zgc) + wave * 1024), 16, 0, 0);
            }
        }
        if constexpr (ZM && (EPI == EPI_GEGLU || EPI == EPI_QKV)) {
            static_assert(2 * (BN / 4) <= NT, "one float4 of G' or C' per thread");
            if (z_shared_slot && tid < ZNV) {   // G' | C' of the tile's columns: one 16-byte LDS-DMA per thread, straight into zgc[tid] (per-row timesteps: the epilogue reads them per row from global memory)
                const int which = tid >= BN / 4, t4 = tid - which * (BN / 4);
                int cp = col0 + 4 * t4;
                cp = cp < a.N - 4 ? cp : a.N - 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((which ? zC_ : zG_) + (long)slot0 * a.zt_slot_stride + cp),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(zgc) + wave * 1024), 16, 0, 0);
            }
            // the partial statistics of the rows this lane finishes in the epilogue: blind loads into AGPRs (every lane, no branch), in flight through the first K tile
            const int frow = SCHED == 1 ? row0 + wm * TM + (lane & 15) : row0 + (wm * 2 + grp) * (TM / 2) + (lane & 15);
            z_lane_load<FMR>(a.zstat_in, a.zs_stride, a.zparts, frow, a.M - 1, lane >> 4, zlr);
        }
    };
    unsigned long long* ts = (a.ts && wave == 0) ? a.ts + 8 * (long)blockIdx.x : nullptr;
    if (ts && lane == 0) { ts[0] = __builtin_readcyclecounter(); ts[6] = ez_stamp_start(); }   // [6], [7]: the 100 MHz device-wide clock (cycle counters are not comparable between workgroups)
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- loop-invariant addressing: byte offset of this thread's chunk of piece p (K offset excluded) ----
    constexpr int PMAX = P0;
    uint32_t poff[PMAX];
    const int pbeg = (SCHED == 1 && grp == 1) ? P0 : 0;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int p = pbeg + i;
        const int q = p * NTG + tg;
        const int row = q >> 3, c = q & 7;
        if (p < PA) {
            int grow = row0 + row;
            grow = grow < a.M ? grow : a.M - 1;
            poff[i] = (uint32_t)(grow * a.lda + ((c ^ ((row >> 1) & 7)) << 3)) * 2u;
        } else {
            const int r2 = row - BM;
            int gr = col0 + r2;
            gr = gr < a.wrows ? gr : a.wrows - 1;
            poff[i] = (uint32_t)(gr * a.ldw + ((c ^ ((r2 >> 1) & 7)) << 3)) * 2u;
        }
    }
    const char* gA = reinterpret_cast<const char*>(a.A) + (long)kb * (BK * 2);
    const char* gW = reinterpret_cast<const char*>(a.W) + (long)kb * (BK * 2);

    // issue this wave's LDS-DMA instructions of K tile t: pieces [PB, PE) of the tile (compile-time range)
    auto issue = [&](int t, auto PB_, auto PE_) {
        constexpr int PB = decltype(PB_)::value, PE = decltype(PE_)::value;
        char* dst = smem + (t % NS) * STAGE + wg * 1024;
        const long koff = (long)t * (BK * 2);
#pragma unroll
        for (int p = PB; p < PE; ++p) {
            const char* src = (p < PA ? gA : gW) + koff + poff[p - PB];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + p * 4096), 16, 0, 0);
        }
    };

    // fragment read offsets: row (lane & 15) of a 16-row fragment, k-step ks (32 of K) -> 16-byte slot (4 ks + (lane >> 4)) ^ ((row >> 1) & 7);
    // fragment bases are multiples of 16 rows, so two per-lane offsets serve every A and W fragment
    const int r16 = lane & 15, kq = lane >> 4;
    uint32_t foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = r16 * 128 + (((4 * ks + kq) ^ (r16 >> 1)) << 4);
    const int a_base = wm * TM * 128, b_base = BM * 128 + wn * TN * 128;

    // one LDS-DMA instruction: piece p (index within the tile) of K tile t; pi = its index in this thread's poff[]
    auto issue_piece = [&](int t, int p, int pi) {
        char* dst = smem + (t % NS) * STAGE + wg * 1024;
        const char* src = (p < PA ? gA : gW) + (long)t * (BK * 2) + poff[pi];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + p * 4096), 16, 0, 0);
    };
    bf16x8 af[FM][2], bfr[FN][2];
    if constexpr (VAR & 16) {
        const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(smem + lane * 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i) { af[i][ks] = f0; asm volatile("" : "+v"(af[i][ks])); }
#pragma unroll
            for (int j = 0; j < FN; ++j) { bfr[j][ks] = f0; asm volatile("" : "+v"(bfr[j][ks])); }
        }
    }
    // LOAD phase: all fragments of tile t -> registers, with the refill pieces [PB, PE) of tile tr issued BETWEEN the reads (one piece
    // every few reads): a wave blocked at the vector-memory port still has LDS reads in flight (measured 1793 -> 1594 cycles per K
    // tile on the 128x288 tile against pieces-behind-reads; pieces-before-reads: 1882)
    auto load_phase = [&](int t, int tr, bool refill, auto PB_, auto PE_) {
        constexpr int PB = decltype(PB_)::value, PE = decltype(PE_)::value, PG = PE - PB;
        constexpr int NR = 2 * (FM + FN), STRIDE = NR / (PG > 0 ? PG : 1) > 0 ? NR / (PG > 0 ? PG : 1) : 1;
        const char* cT = smem + (t % NS) * STAGE;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int ks = r / (FM + FN), idx = r % (FM + FN);
            if constexpr (!(VAR & 16)) {
                if (idx < FM) af[idx][ks] = *reinterpret_cast<const bf16x8*>(cT + foff[ks] + a_base + idx * 2048);
                else bfr[idx - FM][ks] = *reinterpret_cast<const bf16x8*>(cT + foff[ks] + b_base + (idx - FM) * 2048);
            }
            if constexpr (!(VAR & 32)) {
                if ((r + 1) % STRIDE == 0 && (r + 1) / STRIDE - 1 < PG) {
                    if (refill) issue_piece(tr, PB + (r + 1) / STRIDE - 1, (r + 1) / STRIDE - 1);
                }
            }
        }
        if constexpr (!(VAR & 32)) {
#pragma unroll
            for (int q = NR / STRIDE; q < PG; ++q)
                if (refill) issue_piece(tr, PB + q, q);
        }
    };
    auto mfmas = [&]() {
        if constexpr (VAR & 8) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(af[i][ks]));
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(bfr[j][ks]));
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bfr[j][ks]), "v"(af[i][ks]));
    };
    auto fence_lds = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (SCHED == 1) {
        auto run = [&](auto G_) {
            constexpr int G = decltype(G_)::value;
            constexpr int PB = G == 0 ? 0 : P0, PE = G == 0 ? P0 : NP, PG = PE - PB;   // this group's pieces of every tile
            static_assert(PG * (PD - 1) < 64, "vmcnt range");
            using IB = std::integral_constant<int, PB>;
            using IE = std::integral_constant<int, PE>;
            // prologue: tiles 0 .. PD-1; tile 0 must be complete (every wave's share) before interval 0
#pragma unroll
            for (int t = 0; t < PD; ++t) {
                if (t < nt) issue(t, IB{}, IE{});
                if (t == 0) z_late_load();   // right behind tile 0 (one LDS-DMA, left in flight by the wait below: zx)
            }
            if (zx < 0) wait_vmcnt<0>(); else wait_younger_x<PG, PD - 1, NZT>((nt < PD ? nt : PD) - 1, zx);
            barrier();
            if constexpr (G == 1) barrier();   // interval 0: group 1 has nothing to do yet
            auto step = [&](int t, auto RF_) {
                constexpr bool rf = decltype(RF_)::value;   // steady state: tile t + PD exists and is issued here
                // ---- LOAD(t)
                load_phase(t, t + PD, rf, IB{}, IE{});
                if constexpr (G == 1) {
                    // own share of tile t + 1 landed (group 0 reads it right after the barrier below)
                    if constexpr (rf) wait_vmcnt<PG * (PD - 1)>();
                    else if (t + 1 < nt) wait_younger<PG, PD - 1>(nt - 2 - t);
                }
                fence_lds();
                barrier();
                // ---- MFMA(t)
                mfmas();
                if constexpr (G == 0) {
                    if constexpr (rf) wait_vmcnt<PG * (PD - 1)>();
                    else if (t + 1 < nt) wait_younger<PG, PD - 1>(nt - 2 - t);
                    barrier();
                } else {
                    if (rf || t + 1 < nt) barrier();
                }
            };
            if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
            int t = 0;
            for (; t + PD < nt; ++t) step(t, std::true_type{});
            for (; t < nt; ++t) step(t, std::false_type{});
            if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();
        };
        if (grp == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    } else {
        // SCHED 2: tile u is issued (all NP pieces) by group (u + PD) & 1, in interval u - PD (prologue: before interval 0)
        using IB = std::integral_constant<int, 0>;
        using IE = std::integral_constant<int, NP>;
        static_assert(NP * ((PD - 1) / 2 + 1) < 64, "vmcnt range");
        // own tiles that are younger than tile u and already issued at the end of interval u - 1: u + 2, u + 4, ... <= min(u - 1 + PD, nt - 1)
        auto own_younger = [&](int u) {
            const int last = u - 1 + PD < nt - 1 ? u - 1 + PD : nt - 1;
            return last >= u + 2 ? (last - u) / 2 : 0;
        };
        auto run = [&](auto G_) {
            constexpr int G = decltype(G_)::value;
            // z_late_load (the per-column vectors' LDS-DMA; fused QKV: the RoPE-table warm-up) goes out right behind tile 0 in the group that owns it, in front of its first
            // tile in the other group; nothing in front of the loop waits for it in the other group
            if constexpr ((PD & 1) != G) z_late_load();
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                if (((u + PD) & 1) == G && u < nt) issue(u, IB{}, IE{});
                if (u == 0 && (PD & 1) == G) z_late_load();
            }
            {
                const int last = PD - 1 < nt - 1 ? PD - 1 : nt - 1;
                if constexpr ((PD & 1) == G) { if (zx < 0) wait_vmcnt<0>(); else wait_younger_x<NP, (PD - 1) / 2, NZT>(last >= 2 ? last / 2 : 0, zx); }   // owner of tile 0: own younger tiles 2, 4, ...
            }
            barrier();
            // end of interval i: the owner of tile i + 1 -- group (i + 1 + PD) & 1 -- makes sure it has landed.  Group G's LOAD intervals are
            // i = G mod 2, so it owns tile i + 1 at the end of its LOAD intervals iff PD is odd, at the end of its MFMA (and idle) intervals
            // iff PD is even: a compile-time fact, no parity test in the loop
            auto end_wait = [&](int i, auto RF_) {
                constexpr bool rf = decltype(RF_)::value;
                if constexpr (rf) {
                    // steady state (i + PD <= nt - 1): own younger tiles are i + 3 .. i + PD (PD odd) or .. i - 1 + PD (PD even)
                    wait_vmcnt<NP * ((PD - 1) / 2)>();
                } else {
                    if (i + 1 < nt) wait_younger<NP, (PD - 1) / 2>(own_younger(i + 1));
                }
            };
            if constexpr (G == 1) {   // interval 0
                if constexpr (PD % 2 == 0) end_wait(0, std::false_type{});
                barrier();
            }
            auto step = [&](int t, auto RF_) {
                constexpr bool rf = decltype(RF_)::value;   // t + 1 + PD <= nt - 1: both intervals of this step are steady
                // ---- interval t: LOAD(t)
                load_phase(t, t + PD, rf || t + PD < nt, IB{}, IE{});
                if constexpr (PD % 2 == 1) end_wait(t, RF_);
                fence_lds();
                barrier();
                // ---- interval t + 1: MFMA(t)
                mfmas();
                if (rf || t + 1 < nt) {
                    if constexpr (PD % 2 == 0) end_wait(t + 1, RF_);
                    barrier();
                }
            };
            if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
            int t = G;
            for (; t + 1 + PD < nt; t += 2) step(t, std::true_type{});
            for (; t < nt; t += 2) step(t, std::false_type{});
            if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();
        };
        if (grp == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    }

    // ---- epilogue ----
    // the last MFMA's result is not interlocked against the VALU reads below (inline asm, see mfmas): 20 wait states tied to the accumulators
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
    // LayerNorm algebra: (mu, r) of the rows this lane finishes (the statistics were requested behind the first K tile and landed under the loop)
    float2 zmr[FMR];
    if constexpr (ZMC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        z_lane_finish<FMR>(zlr, a.zparts, lane >> 4, a.zD, a.zeps, zmr);
    }
    if constexpr (SCHED == 2) {
        // exchange the two groups' partial sums through the (dead) ring: group g keeps the row fragments [g * FM/2, (g + 1) * FM/2) of its
        // wave tile and parks the others for its partner wave (same wg, other group); lane-linear 16-byte accesses
        constexpr int HF = FM / 2;
        static_assert(8 * HF * FN * 1024 <= NS * STAGE, "exchange area must fit the ring");
        // RAW barriers around the exchange (LDS traffic ordered by explicit lgkmcnt waits): __syncthreads() makes hipcc drain vmcnt(0) first, i.e. wait for the
        // epilogue operands requested just above (residual rows / LayerNorm affine + RoPE rows) BEFORE the exchange instead of under it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();   // every wave is past its last fragment read: the ring is dead
        f32x4* xw = reinterpret_cast<f32x4*>(smem) + (long)(wave ^ 4) * (HF * FN * 64);   // what the partner will read
        f32x4* xr = reinterpret_cast<f32x4*>(smem) + (long)wave * (HF * FN * 64);
        // (compile-time register indices on both sides of a uniform branch: a runtime-indexed accumulator array would live in scratch)
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) xw[(i * FN + j) * 64 + lane] = acc[HF + i][j];
        } else {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) xw[(i * FN + j) * 64 + lane] = acc[i][j];
        }
        // The epilogue's global operands are requested HERE -- behind the exchange's LDS writes, in front of its second barrier -- and land under the barrier,
        // the partner's partial sums and the first arithmetic.  Not earlier: hipcc drains vmcnt(0) in front of the first LDS write that follows an LDS-DMA it
        // cannot prove finished (the K loop's), so anything requested above the writes is waited for before the exchange even starts (round 5 had the residual
        // rows there: the ISA shows `s_waitcnt vmcnt(0)` in front of the first barrier)
        // EPI_RESID: the residual rows this wave finishes after the exchange (16 rows x the tile's columns)
        constexpr int RF = EPI == EPI_RESID ? FM / 2 : 1, RN = EPI == EPI_RESID ? FN : 1;
        float4 rres[RF][RN], dres[RDUAL ? RF : 1][RDUAL ? RN : 1];
        float2 zst[RF][4];
        if constexpr (RZIN) {   // partial statistics of the rows this wave finishes: parts cg, cg + 4 of both sets (zparts <= 8; clamped, weight 0 beyond: pp_store_resid)
            const int m_in = lane & 15, cg = lane >> 4;
            const int p0 = cg < a.zparts ? cg : a.zparts - 1;
            const int p1 = cg + 4 < a.zparts ? cg + 4 : a.zparts - 1;
#pragma unroll
            for (int i = 0; i < RF; ++i) {
                int row = row0 + (wm * 2 + grp) * (TM / 2) + i * 16 + m_in;
                row = row < a.M ? row : a.M - 1;
                zst[i][0] = a.zstat_in[(long)p0 * a.zs_stride + row];
                zst[i][1] = a.zstat_in[(long)p1 * a.zs_stride + row];
                zst[i][2] = a.zstat_in2[(long)p0 * a.zs_stride + row];
                zst[i][3] = a.zstat_in2[(long)p1 * a.zs_stride + row];
            }
        }
        if constexpr (RRES) {
            const int m_in = lane & 15, cg = lane >> 4;
#pragma unroll
            for (int i = 0; i < RF; ++i) {
                int row = row0 + (wm * 2 + grp) * (TM / 2) + i * 16 + m_in;
                row = row < a.M ? row : a.M - 1;
                const float* dsrc = nullptr;
                if constexpr (RDUAL) dsrc = a.zd + (long)(int)(((float)row + 0.5f) * __builtin_amdgcn_rcpf((float)a.rows_per_b)) * a.zd_stride;   // row / rows_per_b
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    int col = col0 + wn * TN + j * 16 + 4 * cg;
                    col = col < a.N - 4 ? col : a.N - 4;
                    rres[i][j] = *reinterpret_cast<const float4*>(a.resid + (long)row * a.ldr + col);
                    if constexpr (RDUAL) dres[i][j] = *reinterpret_cast<const float4*>(dsrc + col);
                }
            }
        }
        // EPI_QKV, register epilogue: RoPE rows of the 16 rows this wave finishes
        constexpr int QDH = EPI == EPI_QKV ? BN / 2 : 64;
        QkvOperands<QDH, EPI == EPI_QKV ? FM / 2 : 1> qop;
        if constexpr (EPI == EPI_QKV) {
            if (a.hn.perm) qkv_request<QDH, FM / 2>(a, col0, row0 + (wm * 2 + grp) * (TM / 2) + (lane & 15), lane, tid, qop);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
        f32x4 half[HF][FN];
        // group 0 (even K tiles) + group 1 (odd K tiles): IEEE addition is commutative, so the sum does not depend on which wave adds
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) half[i][j] = acc[i][j] + xr[(i * FN + j) * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < HF; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) half[i][j] = acc[HF + i][j] + xr[(i * FN + j) * 64 + lane];
        }
        // from here on: 2 WM x WN waves, wave tile TM/2 x TN
        const int ewm = wm * 2 + grp;
        if constexpr (EPI == EPI_RESID) {
            static_assert(WN == 1, "EPI_RESID: a wave holds whole tile rows after the exchange");
            static_assert(BM * (BN + 8) * 2 <= NS * STAGE, "A' tile must fit the ring");
            if constexpr (RDUAL) pp_store_resid<BM, BN, HF, FN, TM / 2, TN, NT, RGATE, RRES, true>(a, half, smem, row0, col0, ewm, lane, tid, zgc, rres, dres, zst);
            else pp_store_resid<BM, BN, HF, FN, TM / 2, TN, NT, RGATE, RRES, false, RCOPY2, RZIN>(a, half, smem, row0, col0, ewm, lane, tid, zgc, rres, rres, zst);
        }
        if constexpr (EPI == EPI_QKV) {
            static_assert(BN == 144 || BN == 128, "EPI_QKV tiles hold two whole heads (head_dim 72 / 64)");
            static_assert(BM * (BN + 4) * 4 + BM * BN * 2 <= NS * STAGE, "epilogue tile + staging must fit the ring");
            static_assert(BM * (BN + 8) * 2 <= NS * STAGE, "bf16 staging tile must fit the ring");
            if (a.hn.perm) pp_store_qkv_reg<BM, BN, BN / 2, HF, FN, TM / 2, TN, NT, ZM>(a, half, smem, row0, col0, ewm, lane, tid, zmr, zgc, reinterpret_cast<float*>(smem + NS * STAGE + BM * 8 + 2 * BN * 4 + 8 * 256), slot0, qop, ts);
            else pp_store_qkv<BM, BN, BN / 2, HF, FN, TM / 2, TN, NT, ZM>(a, half, smem, row0, col0, ewm, wn, lane, tid, zmr, zgc, slot0);
        }
        if constexpr (EPI == EPI_GEGLU || EPI == EPI_PARTIAL) {
            constexpr bool lds_ok = BM * ((EPI == EPI_GEGLU ? BN / 2 : BN) + 8) * 2 <= NS * STAGE;
            if constexpr (lds_ok) {
                if (EPI == EPI_GEGLU || a.part_bf16) {
                    pp_store_lds<BM, BN, HF, FN, TM / 2, TN, NT, EPI, ZM>(a, half, smem, row0, col0, ewm, wn, lane, tid, z, zmr, zgc, slot0, ts);
                    if (ts && lane == 0) { ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
                    return;
                }
            }
        }
        if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) pp_store_direct<HF, FN, TM / 2, TN, EPI>(a, half, row0, col0, ewm, wn, lane, z);
    } else {
        if constexpr (EPI == EPI_GEGLU || EPI == EPI_PARTIAL) {
            constexpr bool lds_ok = BM * ((EPI == EPI_GEGLU ? BN / 2 : BN) + 8) * 2 <= NS * STAGE;
            if constexpr (lds_ok) {
                if (EPI == EPI_GEGLU || a.part_bf16) {
                    pp_store_lds<BM, BN, FM, FN, TM, TN, NT, EPI, ZM>(a, acc, smem, row0, col0, wm, wn, lane, tid, z, zmr, zgc, slot0, ts);
                    if (ts && lane == 0) { ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
                    return;
                }
            }
        }
        if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) pp_store_direct<FM, FN, TM, TN, EPI>(a, acc, row0, col0, wm, wn, lane, z);
    }
    if (ts && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
}

}  // namespace
