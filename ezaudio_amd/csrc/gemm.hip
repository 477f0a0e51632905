// bf16 MFMA GEMM for the DiT projections:  C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear layout, both K-contiguous)
//
// Restates the `aten::linear` calls of the reference block (src/models/utils/attention.py:127-129,148;
// src/models/utils/modules.py:263-277,341-374; src/models/blocks.py:124-128) as one kernel family.
//
// gfx950 design:
//   * v_mfma_f32_32x32x16_bf16, WM x WN waves per workgroup, wave tile (BM/WM)x(BN/WN), fp32 accumulate
//   * BK = 64: one LDS row = 128 B = 8 chunks of 16 B.  Tiles are staged with global_load_lds (16 B per
//     lane, no VGPR round trip).  The LDS image is lane-linear, so the bank swizzle is applied to the
//     SOURCE address: chunk c of row r is fetched into slot c ^ ((r>>1)&7), and fragment reads apply the
//     same involution.  With 128-B rows two consecutive rows span the 64 banks, hence (r>>1): the 16 lanes
//     of a ds_read_b128 group (rows distinct mod 16) then hit 16 distinct 16-B slots -> conflict free.
//   * NS-deep LDS ring (NS = 4 for the 128-wide tiles), ONE raw s_barrier per K tile and COUNTED vmcnt waits:
//     three K tiles of LDS-DMA stay in flight across the barriers, so the ~1 us HBM/L2 latency of a weight tile
//     is covered by three tiles of MFMA work instead of one (a 2-deep ring measured 300-400 TF on these shapes:
//     latency bound, one workgroup per CU)
//   * workgroup -> tile map is XCD aware: the 8 workgroups that the dispatcher puts on one XCD
//     (block b -> XCD b % 8) walk the M tiles of ONE weight panel, so each weight tile is pulled from
//     HBM into exactly one L2.
//   * split-K (blockIdx.z) writes raw fp32 slabs; the row kernel that follows reduces them
//     (launch-boundary reduce, see rowops.hip), so no atomics and bitwise-deterministic results.
#include "common.h"
#include "gemm_pp.h"
#include "gemm_ks.h"
#include "gemm_co.h"

#include <atomic>
#include <type_traits>

namespace {

// wait until at most `younger` (0 .. MAXY) whole tiles of PER loads each are still in flight
template <int PER, int MAXY>
__device__ __forceinline__ void wait_tiles(int younger) {
    if constexpr (MAXY >= 5) { if (younger >= 5) { wait_vmcnt<5 * PER>(); return; } }
    if constexpr (MAXY >= 4) { if (younger >= 4) { wait_vmcnt<4 * PER>(); return; } }
    if constexpr (MAXY >= 3) { if (younger >= 3) { wait_vmcnt<3 * PER>(); return; } }
    if constexpr (MAXY >= 2) { if (younger >= 2) { wait_vmcnt<2 * PER>(); return; } }
    if constexpr (MAXY >= 1) { if (younger >= 1) { wait_vmcnt<PER>(); return; } }
    wait_vmcnt<0>();
}

// ---- generic epilogues.  The MFMAs compute the TRANSPOSED tile (W fragment as the A operand), so in the 32x32 C layout a
// lane owns ONE output row (m = lane & 31) and, per register group g, FOUR CONSECUTIVE output columns
// n = 32j + 8g + 4*(lane>>5) + {0..3}: every store is 16 bytes (fp32) or 8 bytes (bf16) per lane instead of 4.
// acc[i][j]: i-th 32-row fragment x j-th 32-column fragment of the wave tile at (wm * TM, wn * TN) of the workgroup tile.
template <int FM, int FN, int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile(const GemmArgs& a, f32x16 (&acc)[FM][FN], int row0, int col0, int wm, int wn, int lane, int z) {
    const int row_in = lane & 31, hi = lane >> 5;
    if constexpr (EPI == EPI_GEGLU) {
        // W rows are interleaved in groups of 8 (8 value rows, then their 8 gate rows): in the C layout above the
        // value of inner index c sits in register group g (even) and its gate in group g + 1 of the SAME lane.
        bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = row0 + wm * TM + i * 32 + row_in;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const int cv = col0 + wn * TN + j * 32 + 8 * g + 4 * hi;       // packed column of the value
                    const int oc = (col0 + wn * TN + j * 32 + 8 * g) / 2 + 4 * hi;  // output (inner) index
                    if (row < a.M && cv < a.N) {
                        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
                        if (a.bias) {
                            bv = *reinterpret_cast<const float4*>(a.bias + cv);
                            bg = *reinterpret_cast<const float4*>(a.bias + cv + 8);
                        }
                        const float v0 = acc[i][j][4 * g + 0] + bv.x, g0 = acc[i][j][4 * g + 4] + bg.x;
                        const float v1 = acc[i][j][4 * g + 1] + bv.y, g1 = acc[i][j][4 * g + 5] + bg.y;
                        const float v2 = acc[i][j][4 * g + 2] + bv.z, g2 = acc[i][j][4 * g + 6] + bg.z;
                        const float v3 = acc[i][j][4 * g + 3] + bv.w, g3 = acc[i][j][4 * g + 7] + bg.w;
                        uint2 o;
                        o.x = pack_bf2(v0 * gelu_erf(g0), v1 * gelu_erf(g1));
                        o.y = pack_bf2(v2 * gelu_erf(g2), v3 * gelu_erf(g3));
                        if (a.wt) st8_wt(out + (long)row * a.ldo + oc, o); else *reinterpret_cast<uint2*>(out + (long)row * a.ldo + oc) = o;
                    }
                }
        }
    } else {
        float* out = reinterpret_cast<float*>(a.out);
        if constexpr (EPI == EPI_PARTIAL) out += (long)z * a.slab_stride;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = row0 + wm * TM + i * 32 + row_in;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = col0 + wn * TN + j * 32 + 8 * g + 4 * hi;
                    if (row < a.M && col < a.N) {  // N is a multiple of 4 for every caller
                        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                        if constexpr (EPI == EPI_F32) {
                            if (a.bias) {
                                const float4 b = *reinterpret_cast<const float4*>(a.bias + col);
                                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                            }
                            if (a.resid) {
                                const float4 r = *reinterpret_cast<const float4*>(a.resid + (long)row * a.ldr + col);
                                if (a.gate) {
                                    const int slot = (a.cur_step ? *a.cur_step : 0) + (a.row_slot ? a.row_slot[row / a.rows_per_b] : 0);
                                    const float4 g4 = *reinterpret_cast<const float4*>(a.gate + (long)slot * a.gate_slot_stride + col);
                                    v.x *= g4.x; v.y *= g4.y; v.z *= g4.z; v.w *= g4.w;
                                }
                                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                            }
                        }
                        if (EPI == EPI_PARTIAL && a.part_bf16) {
                            uint2 pk;
                            pk.x = pack_bf2(v.x, v.y);
                            pk.y = pack_bf2(v.z, v.w);
                            bf16_t* dst = reinterpret_cast<bf16_t*>(a.out) + (long)z * a.slab_stride + (long)row * a.ldo + col;
                            if (a.wt) st8_wt(dst, pk); else *reinterpret_cast<uint2*>(dst) = pk;
                        } else {
                            float* dst = out + (long)row * a.ldo + col;
                            if (a.wt) st16_wt(dst, v); else *reinterpret_cast<float4*>(dst) = v;
                        }
                    }
                }
        }
    }
}

// bf16 epilogues (GEGLU output, bf16 split-K slabs) through LDS: in the C layout a lane owns ONE row and 4 consecutive columns, so a
// direct store instruction writes 8 bytes into each of 32 different 128-byte lines (4608 line requests for a 128 x 144 tile).  Parking the
// tile in the (dead) ring and writing rows with 16 bytes per lane needs ~300.  Same values, same rounding.
// NPASS = 2: the tile goes out in two row halves (one wave row each) where it does not fit the LDS in one piece (256 x 256 slabs).
template <int BM, int BN, int FM, int FN, int TM, int TN, int NT, int EPI, int NPASS = 1>
__device__ __forceinline__ void store_tile_lds(const GemmArgs& a, f32x16 (&acc)[FM][FN], char* smem, int row0, int col0, int wm, int wn, int lane, int tid, int z) {
    static_assert(EPI == EPI_GEGLU || EPI == EPI_PARTIAL, "bf16 outputs only");
    constexpr int OC = EPI == EPI_GEGLU ? BN / 2 : BN;   // output columns of the tile
    static_assert(OC % 8 == 0, "16-byte row chunks");
    constexpr int PITCH = OC + 8;                         // bf16 elements per LDS row
    constexpr int RP = BM / NPASS;                        // rows per pass
    static_assert(NPASS == 1 || RP == TM, "a pass is one wave row");
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
    const int row_in = lane & 31, hi = lane >> 5;
    __syncthreads();   // every wave is done with the last K tile: the ring is dead
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (pass > 0) __syncthreads();   // the previous half has been copied out
    if (NPASS == 1 || wm == pass) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r = (NPASS == 1 ? wm * TM : 0) + i * 32 + row_in;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (EPI == EPI_GEGLU) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const int cv = col0 + wn * TN + j * 32 + 8 * g + 4 * hi;     // packed column of the value (its gate: + 8)
                    const int oc = (wn * TN + j * 32 + 8 * g) / 2 + 4 * hi;       // inner index within the tile
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
                    if (a.bias && cv < a.N) {
                        bv = *reinterpret_cast<const float4*>(a.bias + cv);
                        bg = *reinterpret_cast<const float4*>(a.bias + cv + 8);
                    }
                    const float v0 = acc[i][j][4 * g + 0] + bv.x, g0 = acc[i][j][4 * g + 4] + bg.x;
                    const float v1 = acc[i][j][4 * g + 1] + bv.y, g1 = acc[i][j][4 * g + 5] + bg.y;
                    const float v2 = acc[i][j][4 * g + 2] + bv.z, g2 = acc[i][j][4 * g + 6] + bg.z;
                    const float v3 = acc[i][j][4 * g + 3] + bv.w, g3 = acc[i][j][4 * g + 7] + bg.w;
                    uint2 o;
                    o.x = pack_bf2(v0 * gelu_erf(g0), v1 * gelu_erf(g1));
                    o.y = pack_bf2(v2 * gelu_erf(g2), v3 * gelu_erf(g3));
                    *reinterpret_cast<uint2*>(tile + r * PITCH + oc) = o;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = wn * TN + j * 32 + 8 * g + 4 * hi;
                    uint2 o;
                    o.x = pack_bf2(acc[i][j][4 * g], acc[i][j][4 * g + 1]);
                    o.y = pack_bf2(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    *reinterpret_cast<uint2*>(tile + r * PITCH + c) = o;
                }
            }
        }
    }
    }
    __syncthreads();
    constexpr int CPR = OC / 8;   // 16-byte chunks per row
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out) + (EPI == EPI_PARTIAL ? (long)z * a.slab_stride : 0);
    const int ocol0 = EPI == EPI_GEGLU ? col0 / 2 : col0;
    const int ncols = EPI == EPI_GEGLU ? a.N / 2 : a.N;
    for (int q = tid; q < RP * CPR; q += NT) {
        const int r = q / CPR, c = (q % CPR) * 8;
        const int grow = row0 + pass * RP + r, gcol = ocol0 + c;
        if (grow < a.M && gcol < ncols) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + r * PITCH + c);
            bf16_t* dst = out + (long)grow * a.ldo + gcol;
            if (gcol + 8 <= ncols) {
                if (a.wt) st16_wt(dst, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
                else *reinterpret_cast<uint4*>(dst) = v;
            } else {   // ragged last chunk (N is a multiple of 4 for every caller)
                const bf16_t* src = tile + r * PITCH + c;
                for (int e = 0; e < ncols - gcol; ++e) dst[e] = src[e];
            }
        }
    }
  }
}

// Lockstep kernel: every wave refills (burst of LDS-DMA pieces behind the barrier), reads its fragments and issues its MFMAs in the same
// order, one raw s_barrier per K tile, counted vmcnt.  Used where the ping-pong kernel (gemm_pp.h) has no configuration: the split-K
// residual GEMMs (tile 9), the small fp32-output GEMMs (tile 25), the VAE convolutions (tile 6), the GEGLU / fused-QKV fall-backs.
// (Round 2's A/B variants of this loop -- refill spread over the k-steps, rotating wave-group phases, in-launch split-K reduce + row
// operator, timing ablations -- measured equal or slower and were removed in round 3; DESIGN.md keeps their numbers.)
template <int BM, int BN, int WM, int WN, int NS, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm(GemmArgs a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int LPT = (BM + BN) * 8 / NT;          // LDS-DMA instructions per thread per tile
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // stage s: [A tile | B tile] at smem + s * STAGE_BYTES
    // NS = ring depth; prefetch distance NS - 1 tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile map (gemm_pp.h tile_of_block / panel_of_block: division-free): workgroup b runs on XCD b % 8; XCD x owns box (xm, xn, xz) of the tile grid
    const int tilesM = (a.M + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    int tm, tn, z;
    if (EPI == EPI_PARTIAL && a.xcd_panel) {
        panel_of_block(a, tilesN, tm, tn, z);
        if (tm >= tilesM) return;
    } else if (!tile_of_block(a, tilesM, tilesN, tm, tn, z)) return;
    const int row0 = tm * BM, col0 = tn * BN;

    const int nk = a.K / BK;
    int kb, ke;
    ksplit_range(a, nk, z, kb, ke);
    const int nt = ke - kb;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- loop-invariant addressing ----
    uint32_t aoff[(BM * 8 + NT - 1) / NT], boff[(BN * 8 + NT - 1) / NT];
    stage_offsets<BM, NT>(aoff, a.lda, row0, a.M - 1, tid);
    stage_offsets<BN, NT>(boff, a.ldw, col0, a.wrows - 1, tid);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // provably wave-uniform -> SGPR
    const char* gA = reinterpret_cast<const char*>(a.A);
    const char* gW = reinterpret_cast<const char*>(a.W) + (long)kb * BK * 2;
    auto stage = [&](int t) {  // K tile t (relative) -> ring slot t % NS
        char* dst = smem + (t % NS) * STAGE_BYTES + wave_u * 1024;
        const long a_off = a.conv_cpb ? (long)((kb + t) / a.conv_cpb) * a.conv_tap_bytes + (long)((kb + t) % a.conv_cpb) * (BK * 2)
                                      : (long)(kb + t) * (BK * 2);
        stage_tile<BM, NT>(gA + a_off, aoff, dst, tid);
        stage_tile<BN, NT>(gW + t * (BK * 2), boff, dst + A_BYTES, tid);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nt) stage(t);

    const int r32 = lane & 31, hi = lane >> 5;
    // fragment read offsets: row r32 of a 32-row fragment, k-step ks -> 16-byte slot (2ks + hi) ^ ((r32>>1)&7).  Fragment
    // bases are multiples of 32 rows, so one set of four per-lane offsets serves every A and W fragment (the rest of the
    // address is wave-uniform and folds into the ds_read immediate).
    uint32_t foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = r32 * 128 + (((2 * ks + hi) ^ ((r32 >> 1) & 7)) << 4);
    const int a_base = wm * TM * 128, b_base = A_BYTES + wn * TN * 128;

    // top of a K tile: its LDS-DMA has landed for this wave (counted vmcnt: younger tiles keep flying), then for every wave
    auto wait_landed = [&](int t) {
        // tile t has landed once at most (tiles still allowed in flight) * LPT loads are outstanding
        const int younger = nt - 1 - t;  // tiles issued after t (capped by the prefetch distance NS - 2 here)
        if constexpr (((BM + BN) * 8) % NT != 0) {
            // ragged staging (768-thread configs): waves below the remainder issue one more LDS-DMA per tile
            constexpr int REM_WAVES = (((BM * 8) % NT) + ((BN * 8) % NT)) / 64;
            static_assert(((BM * 8) % NT == 0) || ((BN * 8) % NT == 0), "at most one ragged operand");
            constexpr int LO = (BM * 8) / NT + (BN * 8) / NT;
            static_assert(NS <= 3, "ragged staging supports rings up to 3");
            if (NS >= 3 && younger >= 1) {
                if (wave_u < REM_WAVES) wait_vmcnt<LO + 1>(); else wait_vmcnt<LO>();
            } else {
                wait_vmcnt<0>();
            }
        } else {
            constexpr int MAXY = NS - 2 < 5 ? NS - 2 : 5;   // tiles t+1 .. t+NS-2 are in flight here (t+NS-1 is issued below)
            static_assert(MAXY * LPT < 64, "ring too deep for the 6-bit vmcnt");  // deeper rings wait conservatively (at most 5 tiles in flight)
            wait_tiles<LPT, MAXY>(younger);
        }
    };
    auto tile_top = [&](int t) {
        wait_landed(t);
        __builtin_amdgcn_s_barrier();  // every wave's part of tile t is in LDS; everyone is done with tile t-1
    };
    // one K tile: 4 k-steps of MFMAs with fragment double buffering (the ds_reads of k-step ks+1 are issued before the MFMAs
    // of k-step ks); RF: tile t + NS - 1 is staged into the slot of tile t-1 on the way
    auto ktile = [&](int t, auto RF) {
        constexpr bool rf = decltype(RF)::value;
        f32x16 (&acc_r)[FM][FN] = acc;
        const char* cT = smem + (t % NS) * STAGE_BYTES;
        if constexpr (rf) stage(t + NS - 1);
        bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(cT + foff[0] + a_base + i * 4096);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(cT + foff[0] + b_base + j * 4096);
        __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(cT + foff[ks + 1] + a_base + i * 4096);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(cT + foff[ks + 1] + b_base + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
            // pin the order: next k-step's LDS reads first, then this k-step's MFMAs (hides the ds_read latency)
            if (ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0);
        }
        // keep the accumulators resident in AGPRs across the back edge: without this hipcc copies all of them to VGPRs
        // and back around every barrier (64+ v_accvgpr moves per K tile, and the copy-out waits for the MFMAs to drain)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc_r[i][j]));
    };
    // steady state (every tile refills the ring) and drain (the last NS - 1 tiles): two loops, each with ONE straight-line body
    const int nt_refill = nt - (NS - 1) > 0 ? nt - (NS - 1) : 0;
    {
        int t = 0;
        for (; t < nt_refill; ++t) { tile_top(t); ktile(t, std::true_type{}); }
        for (; t < nt; ++t) { tile_top(t); ktile(t, std::false_type{}); }
    }

    // ---- epilogue (lane <-> output element mapping: see store_tile) ----
    {
        constexpr bool lds_ok = (EPI == EPI_GEGLU || EPI == EPI_PARTIAL) && BM * ((EPI == EPI_GEGLU ? BN / 2 : BN) + 8) * 2 <= NS * STAGE_BYTES;
        if constexpr (lds_ok) {
            if (a.epi_lds && (EPI == EPI_GEGLU || a.part_bf16)) {
                store_tile_lds<BM, BN, FM, FN, TM, TN, NT, EPI>(a, acc, smem, row0, col0, wm, wn, lane, tid, z);
                return;
            }
        }
        store_tile<FM, FN, TM, TN, EPI>(a, acc, row0, col0, wm, wn, lane, z);
    }
}

// choose the 8-box partition (pm x pn x pz boxes of bm x bn x bz tiles, one box per XCD) with the smallest per-XCD operand footprint
// (bytes of A + W one XCD touches); returns the grid size
int pick_boxes(GemmArgs& a, int BM, int BN) {
    const int tilesM = (a.M + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    const int S = a.splitk;
    double best = 1e30;
    for (int pm = 1; pm <= 8; pm *= 2)
        for (int pn = 1; pm * pn <= 8; pn *= 2) {
            const int pz = 8 / (pm * pn);
            if (pz > S && pz != 1) continue;
            if (a.xcd_map == 0 && !(pm == 1 && pn == 8)) continue;
            const int bm = (tilesM + pm - 1) / pm, bn = (tilesN + pn - 1) / pn, bz = (S + pz - 1) / pz;
            const double rows = (double)(bm * BM < a.M ? bm * BM : a.M) + (double)(bn * BN < a.N ? bn * BN : a.N);
            double fp = rows * ((double)a.K * bz / S) * 2.0;
            const int slots = bm * bn * bz * 8, work = tilesM * tilesN * S;
            fp *= (double)slots / work;                       // ragged boxes waste launch slots and unbalance XCDs
            if (pm == 1 && pn == 8) fp *= 0.9;               // near-ties keep the weight stream disjoint across XCDs
            if (fp < best) { best = fp; a.pm = pm; a.pn = pn; a.pz = pz; a.bm = bm; a.bn = bn; a.bz = bz; }
        }
    a.mbm = ez_magic(a.bm); a.mbn = ez_magic(a.bn); a.msplit = ez_magic(a.splitk); a.mG = ez_magic(((a.N + BN - 1) / BN) * a.splitk);
    return 8 * a.bm * a.bn * a.bz;
}

// ping-pong kernel (gemm_pp.h): 8 waves, two groups one barrier interval apart
template <int BM, int BN, int WM, int WN, int NS, int EPI, int SCHED, int VAR = 0>
int launch_pp(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    dim3 grid(pick_boxes(a, BM, BN), 1, 1);
    if (EPI == EPI_PARTIAL && a.xcd_panel && BM == 128) grid.x = 8 * ((a.N + BN - 1) / BN) * a.splitk * (((a.M + BM - 1) / BM + 7) / 8);   // M tile tm -> XCD tm % 8 (gemm_pp.h)
    else a.xcd_panel = 0;
    constexpr int SMEM = pp_smem_bytes<BM, BN, NS, EPI>();   // ring + (mu, r) of the tile's rows + G' / C' (EPI_RESID: bias / gate / gain) of its columns (LayerNorm algebra) [+ EPI_QKV: warm-up sink]
    static_assert(SMEM <= 160 * 1024, "LDS budget of a CU");
    static std::atomic<bool> attr_set[32];   // per (kernel, device); two host threads may race here on first use (harmless double set)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return 1;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_pp<BM, BN, WM, WN, NS, EPI, SCHED, VAR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 1;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (a.ts && (long)grid.x > a.ts_cap) a.ts = nullptr;   // the stamp buffer has no room for this grid
    hipLaunchKernelGGL((k_gemm_pp<BM, BN, WM, WN, NS, EPI, SCHED, VAR>), grid, dim3(512), SMEM, st, a);
    return 0;
}

// co-resident kernel (gemm_co.h): 4-wave workgroups, two per CU
template <int BM, int BN, int EPI, int VAR = 0>
int launch_co(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.xcd_panel = 0;
    dim3 grid(pick_boxes(a, BM, BN), 1, 1);
    constexpr int SMEM = co_smem_bytes<BM, BN>();
    static std::atomic<bool> attr_set[32];   // per (kernel, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return 1;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_co<BM, BN, EPI, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 1;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (a.ts && (long)grid.x > a.ts_cap) a.ts = nullptr;   // the stamp buffer has no room for this grid
    hipLaunchKernelGGL((k_gemm_co<BM, BN, EPI, VAR>), grid, dim3(256), SMEM, st, a);
    return 0;
}

// K-split-inside-the-workgroup kernel (gemm_ks.h): 8 waves, each with private LDS slots for its own K chunks, no barrier in the K loop
template <int FM, int FN, int EPI, bool GATE, bool RES, int CK, int X = KS_PLAIN>
int launch_ks(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.xcd_panel = 0; a.splitk = 1;
    constexpr int BM = 16 * FM, BN = 16 * FN;
    (void)pick_boxes(a, BM, BN);   // pm x pn XCD boxes with the smallest operand footprint; the tiles of an N group are then dealt in equal runs (ks_tile_of_block)
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    const int gw = a.bn < tilesN ? a.bn : tilesN;
    dim3 grid(8 * ((tilesM * gw + a.pm - 1) / a.pm), 1, 1);
    constexpr int SMEM = 8 * (BM + BN) * 128;
    static_assert(SMEM <= 160 * 1024, "LDS budget of a CU");
    static std::atomic<bool> attr_set[32];   // per (kernel, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return 1;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_ks<FM, FN, EPI, GATE, RES, CK, X>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 1;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (a.ts && (long)grid.x > a.ts_cap) a.ts = nullptr;   // the stamp buffer has no room for this grid
    hipLaunchKernelGGL((k_gemm_ks<FM, FN, EPI, GATE, RES, CK, X>), grid, dim3(512), SMEM, st, a);
    return 0;
}
// tile ids of the K-split kernel: 70 = 48 x 96 (21 x 12 = 252 workgroups at M = 1000, N = 1152), 72 = 32 x 96, 73 = 48 x 64 (the final Linear, N = 128), one 64-wide K chunk
// per wave in flight.  Retired in round 5 (measured, never the default: profiles/r04a_gemm_microbench.txt, r05a_gemm_bench_pf.txt): 71 = 64 x 64, 75 = 32 x 128,
// 76 / 77 = 48 x 96 / 48 x 64 with two 32-wide chunks per wave
template <int EPI, bool GATE, bool RES>
int launch_ks_tile(const GemmArgs& a, hipStream_t st) {
    switch (a.tile) {
        case 70: return launch_ks<3, 6, EPI, GATE, RES, 64>(a, st);
        case 72: return launch_ks<2, 6, EPI, GATE, RES, 64>(a, st);
        case 73: return launch_ks<3, 4, EPI, GATE, RES, 64>(a, st);
        default: return 1;
    }
}

// lockstep kernel k_gemm with an NS-deep ring
template <int BM, int BN, int WM, int WN, int NS, int EPI>
int launch_t(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    const int tilesM = (a.M + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    const int S = a.splitk;
    dim3 grid(pick_boxes(a, BM, BN), 1, 1);
    if (EPI == EPI_PARTIAL && a.xcd_panel) grid.x = 8 * tilesN * S * ((tilesM + 7) / 8);   // M tile tm -> XCD tm % 8, see k_gemm
    else a.xcd_panel = 0;
    constexpr int SMEM = NS * (BM + BN) * 128;
    static_assert(SMEM <= 160 * 1024, "LDS budget of a CU");
    // > 64 KB of dynamic LDS needs the opt-in attribute once per (kernel, DEVICE): function attributes are per device
    static std::atomic<bool> attr_set[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return 1;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, BN, WM, WN, NS, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 1;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, NS, EPI>), grid, dim3(64 * WM * WN), SMEM, st, a);
    return 0;
}

// tile / pipeline configurations (GemmArgs.tile); ids are stable (retired: round 1's experimental ids 11, 14-24, 26-32; in round 3 the
// lockstep ids 0-5, 7, 8, 10, 12, 50, which no caller selected any more)
//   id  tile     waves  ring  LDS     kernel     note
//   6   128x64   4x1    2      48 KB  k_gemm     GEGLU-capable 128x64 (VAE default)
//   9   128x128  4x2    3      96 KB  k_gemm     8 waves: split-K residual GEMMs at M <= 2048
//   13  128x288  4x3    3     156 KB  k_gemm     12 waves: GEGLU GEMM fall-back (gemm_pp bit 0 off)
//   25  128x64   4x2    4      96 KB  k_gemm     8 waves: small fp32 GEMMs
//   (40-42: round 2's large-tile kernel k_gemm2, 256x256 / 192x256 / 256x128, deleted in round 5: since round 4 no default path launched it)
//   60  128x288  4x2 (32x144)         ring 3  156 KB  k_gemm_pp SCHED 1   GEGLU GEMM at M <= 2048
//   61  128x144  4x1 per group        ring 4  144 KB  k_gemm_pp SCHED 2 (k-split); fused QKV GEMM (two heads of 72 per tile)
//   62  128x128  4x2 (32x64)          ring 3   96 KB  k_gemm_pp SCHED 1
//   66  128x144  4 waves (32x144)     ring 2   74 KB  k_gemm_co: TWO workgroups per CU; GEGLU GEMM (round 6)
//   (63-65: ping-pong experiments 64x128 / 128x144 ring 3 / 128x128 2x2, deleted in round 5)
//   70, 72, 73  k_gemm_ks (K split over the waves of a workgroup, no ring): see launch_ks_tile
template <int EPI>
int launch_e(const GemmArgs& a, hipStream_t st) {
#ifdef EZ_ABLATE   // timing ablations of the ping-pong K loop (VAR bits 8 / 16 / 32, gemm_pp.h; a.debug >> 8 selects): experiment builds only
#define EZ_PP_ABL(BM_, BN_, WM_, WN_, NS_, SC_)                                                    \
            case 8: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 8>(a, st);                \
            case 16: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 16>(a, st);              \
            case 32: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 32>(a, st);              \
            case 24: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 24>(a, st);              \
            case 40: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 40>(a, st);              \
            case 48: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 48>(a, st);              \
            case 56: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 56>(a, st);
#else
#define EZ_PP_ABL(BM_, BN_, WM_, WN_, NS_, SC_)
#endif
#define EZ_PP(BM_, BN_, WM_, WN_, NS_, SC_)                                                        \
        switch (a.debug >> 8) {                                                                    \
            case 0: return launch_pp<BM_, BN_, WM_, WN_, NS_, EPI, SC_, 0>(a, st);                \
            EZ_PP_ABL(BM_, BN_, WM_, WN_, NS_, SC_)                                                \
            default: return 1;                                                                     \
        }
    switch (a.tile) {
        case 6: return launch_t<128, 64, 4, 1, 2, EPI>(a, st);
        case 9: return launch_t<128, 128, 4, 2, 3, EPI>(a, st);
        case 13: return launch_t<128, 288, 4, 3, 3, EPI>(a, st);
        case 25: return launch_t<128, 64, 4, 2, 4, EPI>(a, st);
        case 60: EZ_PP(128, 288, 4, 2, 3, 1)
        case 61: EZ_PP(128, 144, 4, 1, 4, 2)
        case 62: EZ_PP(128, 128, 4, 2, 3, 1)
        case 66:
            if constexpr (EPI == EPI_GEGLU) {   // (debug >> 8: phase-offset experiments of the microbenchmark, gemm_co.h)
                if ((a.debug >> 8) == 1) return launch_co<128, 144, EPI, 1>(a, st);
                if ((a.debug >> 8) == 2) return launch_co<128, 144, EPI, 2>(a, st);
            }
            return launch_co<128, 144, EPI>(a, st);
        default: break;
    }
#undef EZ_PP
#undef EZ_PP_ABL
    return 1;   // unknown tile id: refuse (the caller reports EZDIT_E_UNSUPPORTED) instead of silently running another configuration
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t st) {
    if (a.K <= 0 || a.K % BK) return 1;
    if (a.epi == EPI_QKV && a.tile >= 60) {   // ping-pong kernel, k-split schedule: 128 x (2 whole heads), ring 4
        if (a.zstat_in && !(a.zG && a.zC && a.zparts > 0 && a.zparts <= Z_MAXP && a.zs_stride > 0 && a.zw > 0)) return 1;
        if (a.tile == 66) {   // co-resident kernel (gemm_co.h): the register epilogue only
            if (!a.hn.perm) return 1;
            if (a.hn.dh == 72) return a.zstat_in ? launch_co<128, 144, EPI_QKV, 64>(a, st) : launch_co<128, 144, EPI_QKV, 0>(a, st);
            if (a.hn.dh == 64) return a.zstat_in ? launch_co<128, 128, EPI_QKV, 64>(a, st) : launch_co<128, 128, EPI_QKV, 0>(a, st);
            return 1;
        }
        if (a.hn.dh == 72) return a.zstat_in ? launch_pp<128, 144, 4, 1, 4, EPI_QKV, 2, 64>(a, st) : launch_pp<128, 144, 4, 1, 4, EPI_QKV, 2>(a, st);
        if (a.hn.dh == 64) return a.zstat_in ? launch_pp<128, 128, 4, 1, 4, EPI_QKV, 2, 64>(a, st) : launch_pp<128, 128, 4, 1, 4, EPI_QKV, 2>(a, st);
        return 1;
    }
    if (a.epi == EPI_QKV) return 1;   // (the lockstep kernel's 64 x four-head form was deleted in round 6: nothing but gemm_pp = 0 reached it)
    if (a.epi == EPI_RESID && a.tile >= 70) {   // K-split-inside-the-workgroup kernel: residual (optional) + gate (optional) + statistics + next operand
        if (!a.zu || !a.zg || !a.zstat_out || a.zs_stride <= 0 || !a.bias || a.splitk != 1 || (a.gate && !a.resid)) return 1;   // (null out: the fp32 stream is not stored)
        if (a.zd) {   // DUAL form: 48 x 96 tiles only
            if (!a.gate || !a.zg2 || a.tile != 70 || a.rows_per_b <= 0) return 1;
            return launch_ks<3, 6, EPI_RESID, true, true, 64, KS_DUAL>(a, st);
        }
        if (a.zu2) {   // COPY2 form: the in-blocks' MLP-out (gate + residual, 48 x 96 tiles)
            if (!a.gate || !a.zg2 || a.tile != 70 || a.ld_zu2 <= 0) return 1;
            return launch_ks<3, 6, EPI_RESID, true, true, 64, KS_COPY2>(a, st);
        }
        if (a.zstat_in2) {   // ZIN form: skip_linear (no gate, no residual, 48 x 96 tiles)
            if (a.gate || a.resid || a.tile != 70 || !a.zstat_in || !a.zG || a.zparts <= 0 || a.zparts > 16 || a.zD <= 0) return 1;
            return launch_ks<3, 6, EPI_RESID, false, false, 64, KS_ZIN>(a, st);
        }
        if (a.gate) return launch_ks_tile<EPI_RESID, true, true>(a, st);
        if (a.resid) return launch_ks_tile<EPI_RESID, false, true>(a, st);
        return launch_ks_tile<EPI_RESID, false, false>(a, st);
    }
    if (a.epi == EPI_F32 && a.tile >= 70 && a.tile < 80) {
        if (a.resid || a.conv_cpb) return 1;
        return launch_ks_tile<EPI_F32, false, false>(a, st);
    }
    if (a.epi == EPI_RESID && a.tile == 61) {   // batched prompts: un-split residual projection on the ping-pong kernel, 128 x 144 tiles (k-split schedule, ring 4)
        if (!a.zu || !a.zg || !a.zstat_out || a.zs_stride <= 0 || !a.bias || a.splitk != 1 || (a.gate && !a.resid) || a.row_slot) return 1;   // (null out: the fp32 stream is not stored)
        if (a.zd) {   // DUAL form (GemmArgs.zd)
            if (!a.gate || !a.zg2 || a.rows_per_b <= 0) return 1;
            return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 64 + 128 + 256>(a, st);
        }
        if (a.zu2) {   // COPY2 form: the in-blocks' MLP-out
            if (!a.gate || !a.zg2 || a.ld_zu2 <= 0) return 1;
            return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 64 + 128 + 512>(a, st);
        }
        if (a.zstat_in2) {   // ZIN form: skip_linear
            if (a.gate || a.resid || !a.zstat_in || !a.zG || a.zparts <= 0 || a.zparts > 8 || a.zD <= 0) return 1;
            return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 1024>(a, st);
        }
        if (a.gate) return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 64 + 128>(a, st);
        if (a.resid) return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 128>(a, st);
        return launch_pp<128, 144, 4, 1, 4, EPI_RESID, 2, 0>(a, st);
    }
    if (a.epi == EPI_RESID) return 1;
    if (a.epi == EPI_GEGLU && a.zstat_in) {   // GEGLU GEMM that finishes the LayerNorm of its operand (LayerNorm algebra): ping-pong kernel only
        if ((a.tile != 60 && a.tile != 66) || !(a.zG && a.zC && a.zparts > 0 && a.zparts <= Z_MAXP && a.zs_stride > 0 && a.zw > 0)) return 1;
        if (a.tile == 66) return launch_co<128, 144, EPI_GEGLU, 64>(a, st);
        return launch_pp<128, 288, 4, 2, 3, EPI_GEGLU, 1, 64>(a, st);
    }
    if (a.epi == EPI_GEGLU) return launch_e<EPI_GEGLU>(a, st);
    if (a.epi == EPI_PARTIAL) return launch_e<EPI_PARTIAL>(a, st);
    if (a.epi == EPI_F32) return launch_e<EPI_F32>(a, st);
    return 1;
}
