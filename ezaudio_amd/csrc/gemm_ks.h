// K-split-inside-the-workgroup bf16 MFMA GEMM for the narrow residual projections (N = D) at small M:
//      C[M,N] = A[M,K] . W[N,K]^T      (nn.Linear layout, both K-contiguous)
//
// Same math as k_gemm / k_gemm_pp (reference: the to_out / proj_out / mlp.net.2 aten::linear calls of src/models/utils/attention.py:148,
// src/models/utils/modules.py:277 and the gated residual adds of src/models/blocks.py:139,151,156), a different decomposition.
//
// Why: at M = 1000 a D x D projection is 2.65 GFLOP -- one microsecond of MFMA time -- and what a launch costs is how long the LAST
// workgroup waits for its operands.  Split-K over the grid (k_gemm, 216 workgroups, bf16 slabs, a row kernel to reduce them) pays a second
// launch and 18 MB of slab traffic; 64 x 128 un-split tiles (k_gemm_pp, round 3) fill 144 of 256 CUs and walk 18 dependent barrier intervals.
// Here the tile is small enough (48 x 96) for 21 x 12 = 252 workgroups -- one round of the chip -- and K is split over the EIGHT WAVES of the
// workgroup: wave w owns the 64-wide K chunks w, w + 8, ... and a PRIVATE LDS slot for one chunk of [A rows | W rows].  Every wave runs
//      wait for its own LDS-DMA (vmcnt) -> read the chunk's fragments -> re-issue the slot for its next chunk -> 36 MFMAs
// with no workgroup barrier anywhere in the K loop: all 144 KB of LDS are in flight from the first cycle, every operand byte crosses the
// LDS exactly once (each wave multiplies the WHOLE tile for its K slice), and the waves de-phase on their own.  After the loop the eight
// partial tiles are summed through the (dead) slots in a fixed order (deterministic), 8 lanes per output row, and the epilogue runs on whole
// row segments: gated residual, LayerNorm partial statistics over the tile's columns, the next GEMM's operand bf16(h g) -- EPI_RESID, the
// producer side of the LayerNorm algebra (common.h, GemmArgs.z*) -- or plain bias + store (EPI_F32).
//
// LDS image of a chunk (as in k_gemm_pp): rows [A tile | W tile], 128 B each, filled by global_load_lds (16 B per lane, one 1-KB piece =
// 8 rows) with the bank swizzle on the SOURCE address (chunk c of row r lands in 16-byte slot c ^ ((r >> 1) & 7)).
#pragma once
#include "common.h"

namespace {

// EPI_RESID operands of one thread (8 threads per output row; SL 16-byte column slots each): requested at kernel start, used after the loop.
// They are loaded by INLINE ASM straight into AGPRs (the accumulators take 4 FM FN of the 128, these 16 SL): hipcc neither counts them in its
// vmcnt bookkeeping nor shuffles them between register files.  Before (round 4, plain C++ loads): hipcc loaded three of the twelve into VGPRs,
// copied them to AGPRs in front of the first MFMA and put `s_waitcnt vmcnt(0)` there -- the second chunk's LDS-DMA could not be re-issued before
// the operands had landed -- and the device step counter, a dependent VECTOR load in front of them, drained the first chunk's DMA before
// the operands were even requested: two full memory latencies on the critical path of every launch (profiles/r05_experiments.txt).
template <int SL>
struct KsOperands {
    f32x4 b[SL], r[SL], g[SL], z[SL];
    f32x2 st[4];   // KS_ZIN: four partial statistics (sum, sum of squares) of this thread's row
};
// forms of the EPI_RESID epilogue (template parameter X)
constexpr int KS_PLAIN = 0;
constexpr int KS_DUAL = 1;    // constant cross-attention-out vector for the rows outside the active range (GemmArgs.zd)
constexpr int KS_COPY2 = 2;   // a SECOND operand bf16(h_new * zg2) -> zu2: the in-blocks' MLP-out also writes its half of the out-block's [x | skip] operand
constexpr int KS_ZIN = 3;     // the launch also is a LayerNorm-algebra CONSUMER: acc := r (acc - mu G') + C' with (mu, r) from two sets of partial statistics (skip_linear)
__device__ __forceinline__ f32x4 ks_ld16_agpr(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x2 ks_ld8_agpr(const float2* p) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(v) : "v"(p) : "memory");
    return v;
}
// all loads of this wave have landed; ties the operand registers to the wait so that no read of them can be scheduled above it
template <int SL, int EPI, bool GATE, bool RES, int X>
__device__ __forceinline__ void ks_operands_wait(KsOperands<SL>& op) {
    if constexpr (X == KS_ZIN) asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.st[0]), "+a"(op.st[1]), "+a"(op.st[2]), "+a"(op.st[3]));
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        if constexpr (EPI == EPI_RESID) {
            if constexpr (X == KS_ZIN) asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.b[q]), "+a"(op.g[q]), "+a"(op.z[q]));
            else if constexpr (GATE && RES) asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.b[q]), "+a"(op.r[q]), "+a"(op.g[q]), "+a"(op.z[q]));
            else if constexpr (RES) asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.b[q]), "+a"(op.r[q]), "+a"(op.z[q]));
            else asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.b[q]), "+a"(op.z[q]));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+a"(op.b[q]));
        }
    }
}

// column of 16-byte (4-float) slot q of lane j of a row: slots come in pairs (8 contiguous columns: one 16-byte bf16 store) while they last,
// then one single slot per lane (paired with lane ^ 1 for the bf16 store)
template <int SL>
__device__ __forceinline__ int ks_slot_of(int q, int j) {
    constexpr int NP = SL / 2;                       // pairs per lane
    return q < 2 * NP ? 16 * (q >> 1) + 2 * j + (q & 1) : 16 * NP + j;
}

// workgroup -> tile map of the K-split kernel: the N tiles are cut into a.pn groups of a.bn tiles, the (M tile, N tile) pairs of a group are
// linearised M-major and dealt in a.pm equal contiguous runs to the a.pm XCDs of the group (XCD = blockIdx % 8): every XCD sees a.bn weight
// panels and 1 / a.pm of the rows like the box map (tile_of_block), but no XCD gets more than ceil(tiles of a group / a.pm) workgroups --
// 21 x 12 tiles on 2 x 4 boxes are 33 / 30 tiles per XCD, i.e. a SECOND ROUND on the 32 CUs of four XCDs; the runs are 32 / 31
__device__ __forceinline__ bool ks_tile_of_block(const GemmArgs& a, int tilesM, int tilesN, int& tm, int& tn) {
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int lpm = __builtin_ctz(a.pm);           // a.pm is 1, 2, 4 or 8 (pick_boxes)
    const int xm = xcd & (a.pm - 1), gn = xcd >> lpm;
    const int n0 = gn * a.bn;
    int gw = tilesN - n0;                          // N tiles of this group
    gw = gw < a.bn ? gw : a.bn;
    if (gw <= 0) return false;
    const int T = tilesM * gw, run = (T + a.pm - 1) >> lpm;
    const int t = xm * run + l;
    if (l >= run || t >= T) return false;
    tm = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)gw));   // t / gw for the small integers of a tile grid, without the 40-instruction integer division
    tn = n0 + (t - tm * gw);
    return true;
}

// CK = K columns per chunk: 64 (one slot per wave) or 32 (two slots per wave: one is always in flight)
// DUAL (EPI_RESID with gate and residual; the attention-out projection when cross-attention is skipped for single-key batch elements, GemmArgs.zd):
// rows OUTSIDE [a.act_row0, a.act_row1) get  h_new = resid + gate (acc + bias) + zd[batch element][col]  and  A' = bf16(h_new * zg2): for them this
// launch also is the cross-attention-out projection (whose output is the constant vector zd) and the producer of the GEGLU GEMM's operand
// KS_COPY2 / KS_ZIN (round 6): the out-blocks' LN_2D([x | skip]) -> skip_linear (blocks.py:124-128) by the LayerNorm algebra.  The statistics of the concatenation are the sums of
// the halves' statistics: the in-block that produced `skip` stored them (and bf16(skip * g[D:]) into the right half of the out-block's operand, COPY2), the MLP-out projection in front
// of the out-block stores bf16(x * g[:D]) + its statistics, and skip_linear (ZIN) finishes the LayerNorm in its epilogue: no split-K slabs, no row kernel
template <int FM, int FN, int EPI, bool GATE, bool RES, int CK, int X = KS_PLAIN>
__global__ __launch_bounds__(512) void k_gemm_ks(GemmArgs a) {
    constexpr bool DUAL = X == KS_DUAL;
    static_assert(EPI == EPI_F32 || EPI == EPI_RESID, "epilogues");
    static_assert(!DUAL || (EPI == EPI_RESID && GATE && RES), "DUAL: the gated residual projection only");
    static_assert(X == KS_PLAIN || EPI == EPI_RESID, "forms of the residual epilogue");
    static_assert(X != KS_ZIN || (!GATE && !RES), "ZIN: the gate's register slot carries G'");
    static_assert(FN % 2 == 0 && FM >= 1 && FM <= 4, "tile geometry: 8 lanes per output row, FN / 2 column slots each");
    static_assert(CK == 64 || CK == 32, "chunk width");
    constexpr int BM = 16 * FM, BN = 16 * FN;
    constexpr int RPP = 1024 / (2 * CK);           // rows per 1-KB LDS-DMA piece: 8 (128-byte rows) or 16 (64-byte rows)
    constexpr int LPR = 64 / RPP;                  // lanes per row of a piece
    constexpr int NPC = (BM + BN) / RPP;           // pieces per chunk
    constexpr int PA = BM / RPP;                   // pieces [0, PA) come from A
    static_assert(BM % RPP == 0 && BN % RPP == 0, "whole pieces");
    constexpr int NSL = 64 / CK;                   // slots per wave
    constexpr int CHUNK = (BM + BN) * 2 * CK;      // bytes of a chunk
    constexpr int SLOT = NSL * CHUNK;              // bytes of a wave's LDS region = NSL chunks = (after the loop) its partial tile [BM][BN] fp32
    constexpr int KSTEPS = CK / 32;                // MFMA k-steps per chunk
    static_assert(BM * BN * 4 <= SLOT, "partial tile must fit the slot");
    static_assert(8 * SLOT <= 160 * 1024, "LDS budget of a CU");
    constexpr int SL = FN / 2;                     // 16-byte column slots per thread in the epilogue
    constexpr int S4 = BN / 4;                     // 16-byte slots per partial-tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // the kernel arguments the prologue needs in ONE batch (common.h "Kernel-argument batch"), then the device step counter (selects the modulation slot of
    // gate / gain) as a plain scalar load: used right in front of the operand requests, behind the first chunk's LDS-DMA.  (Round 5 issued it by inline asm
    // under `if (a.cur_step)`: ADVICE r05 -- a phi with 0 may become a copy of a register that has not been written yet.)
    int M_ = a.M;
    asm("" : "+s"(M_) : "s"(a.A), "s"(a.W), "s"(a.lda), "s"(a.ldw), "s"(a.wrows), "s"(a.N), "s"(a.K), "s"(a.pm), "s"(a.bn), "s"(a.cur_step), "s"(a.ts), "s"(a.ts_cap),
        "s"(a.row_slot), "s"(a.bias), "s"(a.resid), "s"(a.ldr), "s"(a.gate), "s"(a.zg));
    int slot0 = 0;
    if constexpr (EPI == EPI_RESID) slot0 = a.cur_step ? *a.cur_step : 0;
    const int tilesM = (M_ + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    int tm, tn;
    if (!ks_tile_of_block(a, tilesM, tilesN, tm, tn)) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = a.K / CK;
    unsigned long long* ts = (a.ts && wave == 0) ? a.ts + 8 * (long)blockIdx.x : nullptr;
    unsigned long long t_start = 0, t_sum = 0;   // test hook: [5] packs two epilogue marks relative to [0]: low 32 bits = partial sums complete, high 32 = stores issued
    if (ts && lane == 0) { t_start = __builtin_readcyclecounter(); ts[0] = t_start; ts[6] = ez_stamp_start(); }

    // epilogue thread layout: (row er, lane-in-row ej) of the first 8 BM threads
    const int er = tid >> 3, ej = tid & 7;
    const bool epi_thread = tid < 8 * BM;
    int erow = row0 + er;
    const bool row_ok = epi_thread && erow < a.M;
    erow = erow < a.M ? erow : a.M - 1;
    const int ncl = a.N - 4;                       // N is a multiple of 4 for every caller

    // ---- loop-invariant addressing: byte offset of this lane's 16 bytes of piece p (K offset excluded)
    // The pieces of this wave's FIRST chunk go out as soon as their offsets are known, not behind all NPC offset computations: the time to the
    // first byte is what the whole (latency-bound) kernel is shifted by
    const char* gA = reinterpret_cast<const char*>(a.A);
    const char* gW = reinterpret_cast<const char*>(a.W);
    char* slot = smem + wave * SLOT;               // wave-uniform
    const int nmine = wave < nk ? (nk - wave + 7) / 8 : 0;   // chunks of this wave: wave, wave + 8, ...; chunk number i lives in slot i % NSL
    uint32_t poff[NPC];
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
        const int row = RPP * p + lane / LPR, c = lane % LPR;
        // bank swizzle on the source: 128-byte rows: slot c ^ ((row >> 1) & 7); 64-byte rows (four rows span the 64 banks): c ^ ((row >> 2) & 3)
        const int r2 = p < PA ? row : row - BM;
        const int gc = CK == 64 ? (c ^ ((r2 >> 1) & 7)) : (c ^ ((r2 >> 2) & 3));
        if (p < PA) {
            int grow = row0 + row;
            grow = grow < a.M ? grow : a.M - 1;
            poff[p] = (uint32_t)(grow * a.lda + (gc << 3)) * 2u;
        } else {
            int gr = col0 + r2;
            gr = gr < a.wrows ? gr : a.wrows - 1;
            poff[p] = (uint32_t)(gr * a.ldw + (gc << 3)) * 2u;
        }
        if (nmine > 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((p < PA ? gA : gW) + (long)wave * (CK * 2) + poff[p]),
                                             (__attribute__((address_space(3))) void*)(slot + p * 1024), 16, 0, 0);
    }
    auto issue = [&](int c, char* dst) {           // chunk c (K columns [c CK, (c + 1) CK)) -> LDS at dst
        const long koff = (long)c * (CK * 2);
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            const char* src = (p < PA ? gA : gW) + koff + poff[p];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
        }
    };
    // fragment read offsets (k_gemm_pp): row (lane & 15) of a 16-row fragment, k-step ks (32 of K) -> the swizzled 16-byte slot of that row
    const int r16 = lane & 15, kq = lane >> 4;
    uint32_t foff[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
        foff[ks] = CK == 64 ? r16 * 128 + (((4 * ks + kq) ^ (r16 >> 1)) << 4) : r16 * 64 + ((kq ^ ((r16 >> 2) & 3)) << 4);
    constexpr int FRAG = 16 * 2 * CK;              // bytes of a 16-row fragment

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 1; i < NSL; ++i)                  // (chunk 0 went out with the offsets above)
        if (i < nmine) issue(wave + 8 * i, slot + i * CHUNK);
    // ---- epilogue operands (bias, residual rows, gate, LayerNorm gain): requested right BEHIND the prologue chunks (inline-asm loads into AGPRs,
    // KsOperands) and carried through the K loop.  EVERY thread issues them (rows / columns clamped; the threads beyond the 8 BM epilogue threads
    // never use theirs) so that each wave has exactly NOPL loads behind its prologue chunks: the counted waits below rely on it, the
    // sched_barriers pin the order, and tests/test_host.py counts the loads in the generated code
    constexpr int NOPL = SL * (1 + (EPI == EPI_RESID ? 1 + (RES ? 1 : 0) + (GATE || X == KS_ZIN ? 1 : 0) : 0)) + (X == KS_ZIN ? 4 : 0);
    KsOperands<SL> op;
    bool alt = false;          // DUAL: this thread's row belongs to a batch element whose cross-attention is the constant zd
    int brow = 0;              // ... its batch element
    __builtin_amdgcn_sched_barrier(0);
    {
        const float* bsrc = a.bias ? a.bias : reinterpret_cast<const float*>(a.W);   // null bias: any valid address, the value is dropped below
        const float* gsrc = nullptr;
        const float* zsrc = nullptr;
        auto request = [&](int slot_m) {
            if constexpr (EPI == EPI_RESID) {
                if constexpr (GATE) gsrc = a.gate + (long)slot_m * a.gate_slot_stride;
                zsrc = a.zg + (long)slot_m * a.zg_slot_stride;
                if constexpr (DUAL) {
                    alt = erow < a.act_row0 || erow >= a.act_row1;
                    brow = (int)(((float)erow + 0.5f) * __builtin_amdgcn_rcpf((float)a.rows_per_b));   // erow / rows_per_b (rows < 2^22)
                    if (alt) zsrc = a.zg2 + (long)slot_m * a.zg2_slot_stride;
                }
            }
#pragma unroll
            for (int q = 0; q < SL; ++q) {
                int col = col0 + 4 * ks_slot_of<SL>(q, ej);
                col = col < ncl ? col : ncl;
                op.b[q] = ks_ld16_agpr(bsrc + col);
                if constexpr (EPI == EPI_RESID) {
                    if constexpr (RES) op.r[q] = ks_ld16_agpr(a.resid + (long)erow * a.ldr + col);
                    if constexpr (GATE) op.g[q] = ks_ld16_agpr(gsrc + col);
                    if constexpr (X == KS_ZIN) op.g[q] = ks_ld16_agpr(a.zG + col);   // (static table: no slot)
                    op.z[q] = ks_ld16_agpr(zsrc + col);
                }
            }
            if constexpr (X == KS_ZIN) {   // the 8 lanes of a row share its 2 x zparts partial statistics: lane j takes parts j and j + 8 of both sets (weight 0 beyond zparts: ks_zin_mu_r)
                const int p0 = ej < a.zparts ? ej : a.zparts - 1;
                const int p1 = ej + 8 < a.zparts ? ej + 8 : a.zparts - 1;
                op.st[0] = ks_ld8_agpr(a.zstat_in + (long)p0 * a.zs_stride + erow);
                op.st[1] = ks_ld8_agpr(a.zstat_in + (long)p1 * a.zs_stride + erow);
                op.st[2] = ks_ld8_agpr(a.zstat_in2 + (long)p0 * a.zs_stride + erow);
                op.st[3] = ks_ld8_agpr(a.zstat_in2 + (long)p1 * a.zs_stride + erow);
            }
        };
        // per-row timesteps (ezdit_forward with one t per batch element, never the sampler): the row's slot offset is a dependent vector load.
        // Its own branch with its own copy of the requests: at a join hipcc's (path-insensitive) vmcnt bookkeeping would drain the LDS-DMA on BOTH paths
        if (EPI == EPI_RESID && a.row_slot) {
            request(slot0 + a.row_slot[erow / a.rows_per_b]);
            asm volatile("; per-row slot" ::: "memory");
        } else {
            request(slot0);
            asm volatile("; shared slot" ::: "memory");
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
    for (int i = 0; i < nmine; ++i) {
        // own LDS-DMA of chunk i landed (nothing else orders a ds_read behind it): wait until only the loads issued AFTER it are outstanding
        // -- the operand loads (behind the NSL prologue chunks) and, with two slots, the next chunk.  No other wave touches this slot: no barrier
        {
            const bool nxt = NSL == 2 && i + 1 < nmine;
            if (i < NSL) {
                if (nxt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC + NOPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOPL) : "memory");
            } else {
                if (nxt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        char* cur = slot + (NSL == 2 ? (i & 1) * CHUNK : 0);
        bf16x8 af[FM][KSTEPS], bfr[FN][KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int ii = 0; ii < FM; ++ii) af[ii][ks] = *reinterpret_cast<const bf16x8*>(cur + foff[ks] + ii * FRAG);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j][ks] = *reinterpret_cast<const bf16x8*>(cur + BM * 2 * CK + foff[ks] + j * FRAG);
        }
        // the reads must have returned before the refill may overwrite the slot
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (i + NSL < nmine) issue(wave + 8 * (i + NSL), cur);
        __builtin_amdgcn_sched_barrier(0);
        // transposed product (W fragment as the A operand): a lane owns output row lane & 15 and columns 4 (lane >> 4) + {0..3} of a fragment
        // The LAST MFMA of the chunk carries 20 wait states: an inline-asm MFMA is invisible to hipcc's hazard recogniser, and whatever the register
        // allocator puts on the loop's exit edge (round 5: the DUAL instantiation got `v_accvgpr_read a48 ..; v_accvgpr_mov a48, a52 ..` there, IN FRONT of
        // the wait states that used to follow the loop, and read accumulators the matrix pipe had not written yet: wrong, run-to-run different results)
        // now sits behind them by construction.  ~20 cycles per chunk, in front of a memory wait.
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int ii = 0; ii < FM; ++ii)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if (ks == KSTEPS - 1 && ii == FM - 1 && j == FN - 1)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc[ii][j]) : "v"(bfr[j][ks]), "v"(af[ii][ks]));
                    else
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ii][j]) : "v"(bfr[j][ks]), "v"(af[ii][ks]));
                }
    }
    if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();
    // (the wait states between the last MFMA and the first read of an accumulator are inside the loop: see the last MFMA of a chunk)

    ks_operands_wait<SL, EPI, GATE, RES, X>(op);   // (landed long ago whenever the wave had a second chunk: its vmcnt(0) covered them; here, in front of the DUAL requests below)
    // ---- park this wave's partial tile in its OWN slot (dead: its last chunk was read above and nothing is in flight): [BM][S4] 16-byte
    // slots, slot s of row r at s ^ (r & 7) -- the 8 lanes of a store group hold 8 different rows of one column slot
    {
        float4* mine = reinterpret_cast<float4*>(slot);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = 16 * i + r16, s = 4 * j + kq;
                mine[r * S4 + (s ^ (r & 7))] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
    }
    // DUAL: the constant cross-attention-out vector of this row's batch element, requested HERE by inline asm -- behind the park's LDS writes (hipcc drains
    // vmcnt in front of the first LDS write next to an LDS-DMA it cannot prove finished) and in FRONT of the barrier: the vector is cold (each block's
    // own, first touched here) and lands under the barrier and the partial sums; as plain C++ loads behind the barrier their miss sat in the epilogue
    // (in-situ stamps r05e: DUAL epilogue 9.3K cycles against 6.3 ... 7.8K of the plain form)
    f32x4 dv[SL];
    if constexpr (DUAL || X == KS_COPY2) {   // (COPY2: the gain of the second operand, a static vector)
        const float* dsrc = DUAL ? a.zd + (long)brow * a.zd_stride : a.zg2;
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            int col = col0 + 4 * ks_slot_of<SL>(q, ej);
            col = col < ncl ? col : ncl;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dv[q]) : "v"(dsrc + col) : "memory");
        }
    }
    __syncthreads();
    if (ts && lane == 0) ts[4] = __builtin_readcyclecounter();
    if constexpr (X == KS_COPY2) {
        // COPY2 waits for its vector HERE, right behind the barrier it landed under -- not behind the partial sums like DUAL: with 12 more live registers through the sums hipcc
        // parked one of the (to its knowledge defined) destination registers in an AGPR in front of the wait, i.e. copied data that had not arrived: results differed from run to
        // run (tools/diag_determinism.py; tests/test_host.py reads the generated code for exactly this)
#pragma unroll
        for (int q = 0; q < SL; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dv[q]));
    }
    if (!epi_thread) return;

    // ---- sum the eight partials in wave order (fixed: bit-reproducible) and finish the row segment
    float4 v[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int s = ks_slot_of<SL>(q, ej);
        const float4* p0 = reinterpret_cast<const float4*>(smem) + er * S4 + (s ^ (er & 7));
        float4 t = p0[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 u = p0[w * (SLOT / 16)];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        v[q] = t;
    }
    if (ts && lane == 0) {   // the sums are complete when their last value is: tie the stamp to it
        asm volatile("" :: "v"(v[SL - 1].w));
        t_sum = __builtin_readcyclecounter();
    }
    float* out = reinterpret_cast<float*>(a.out);
    if constexpr (DUAL) {   // the asm loads of dv have landed; ties the registers to the wait
#pragma unroll
        for (int q = 0; q < SL; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dv[q]));
    }
    if constexpr (EPI == EPI_F32) {
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const int col = col0 + 4 * ks_slot_of<SL>(q, ej);
            const f32x4 bb = a.bias ? op.b[q] : f32x4{0.f, 0.f, 0.f, 0.f};   // (a select: the dummy load of a null bias may hold NaN bit patterns)
            const float4 o = make_float4(v[q].x + bb[0], v[q].y + bb[1], v[q].z + bb[2], v[q].w + bb[3]);
            if (row_ok && col < a.N) {
                float* dst = out + (long)erow * a.ldo + col;
                if (a.wt) st16_wt(dst, o); else *reinterpret_cast<float4*>(dst) = o;
            }
        }
    } else {
        float zmu = 0.f, zr = 1.f;
        if constexpr (X == KS_ZIN) {   // (mu, r) of the row over the zD columns of BOTH statistics sets: fixed order, bit-reproducible
            const bool one = ej < a.zparts, two = ej + 8 < a.zparts;   // (fewer than 8 parts: the lanes beyond them contribute nothing)
            float zs = ((one ? op.st[0][0] : 0.f) + (two ? op.st[1][0] : 0.f)) + ((one ? op.st[2][0] : 0.f) + (two ? op.st[3][0] : 0.f));
            float zq = ((one ? op.st[0][1] : 0.f) + (two ? op.st[1][1] : 0.f)) + ((one ? op.st[2][1] : 0.f) + (two ? op.st[3][1] : 0.f));
            zs = oct_sum(zs);
            zq = oct_sum(zq);
            const float inv_d = __builtin_amdgcn_rcpf((float)a.zD);
            zmu = zs * inv_d;
            zr = rsqrtf(fmaxf(fmaf(zq, inv_d, -zmu * zmu), 0.f) + a.zeps);
        }
        float s1 = 0.f;
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const int col = col0 + 4 * ks_slot_of<SL>(q, ej);
            const bool ok = col < a.N;
            if constexpr (X == KS_ZIN) {   // finish the LayerNorm of the operand: r (acc - mu G') (+ C' = the `bias` operand, below)
                v[q].x = zr * fmaf(-zmu, op.g[q][0], v[q].x); v[q].y = zr * fmaf(-zmu, op.g[q][1], v[q].y);
                v[q].z = zr * fmaf(-zmu, op.g[q][2], v[q].z); v[q].w = zr * fmaf(-zmu, op.g[q][3], v[q].w);
            }
            // h_new = resid + gate * (acc + bias): the same two roundings per element as the row kernel (rowbody.h)
            float4 x = make_float4(v[q].x + op.b[q][0], v[q].y + op.b[q][1], v[q].z + op.b[q][2], v[q].w + op.b[q][3]);
            if constexpr (GATE) { x.x *= op.g[q][0]; x.y *= op.g[q][1]; x.z *= op.g[q][2]; x.w *= op.g[q][3]; }
            if constexpr (RES) { x.x += op.r[q][0]; x.y += op.r[q][1]; x.z += op.r[q][2]; x.w += op.r[q][3]; }
            if constexpr (DUAL) {   // + the cross-attention block's constant output for this batch element (exactly 0 for the rows that run cross-attention)
                x.x += alt ? dv[q][0] : 0.f; x.y += alt ? dv[q][1] : 0.f; x.z += alt ? dv[q][2] : 0.f; x.w += alt ? dv[q][3] : 0.f;
            }
            v[q] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
            s1 += (v[q].x + v[q].y) + (v[q].z + v[q].w);
            if (row_ok && ok && out) {   // (null out: nothing reads the fp32 stream behind this launch -- the MLP-out in front of an out-block)
                float* dst = out + (long)erow * a.ldo + col;
                if (a.wt) st16_wt(dst, v[q]); else *reinterpret_cast<float4*>(dst) = v[q];
            }
        }
        // statistics of this tile's valid columns: (sum, sum of squares) -- invalid columns hold zeros; the 8 lanes of a row are neighbours
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < SL; ++q) s2 = fmaf(v[q].x, v[q].x, fmaf(v[q].y, v[q].y, fmaf(v[q].z, v[q].z, fmaf(v[q].w, v[q].w, s2))));
        s1 = oct_sum(s1);   // DPP: three VALU moves instead of three dependent LDS-pipe round trips (ds_bpermute) per sum
        s2 = oct_sum(s2);
        if (ej == 0 && row_ok) a.zstat_out[(long)tn * a.zs_stride + erow] = make_float2(s1, s2);   // part-major: the consumer's loads are contiguous over rows
        // A' = bf16(h_new * zg) as whole 16-byte chunks (8 columns)
        uint2 pk[SL];
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            pk[q].x = pack_bf2(v[q].x * op.z[q][0], v[q].y * op.z[q][1]);
            pk[q].y = pack_bf2(v[q].z * op.z[q][2], v[q].w * op.z[q][3]);
        }
        bf16_t* zrow = a.zu + (long)erow * a.ld_zu;
        auto store8 = [&](int col, uint2 lo, uint2 hi, bf16_t* rowp) {   // columns [col, col + 8) of this row
            if (!row_ok || col >= a.N) return;
            bf16_t* dst = rowp + col;
            if (col + 8 <= a.N) {
                const float4 f = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y));
                if (a.wt) st16_wt(dst, f); else *reinterpret_cast<float4*>(dst) = f;
            } else {   // ragged tail: N is a multiple of 4
                *reinterpret_cast<uint2*>(dst) = lo;
            }
        };
        auto store_row = [&](bf16_t* rowp) {
#pragma unroll
            for (int q = 0; q + 1 < 2 * (SL / 2); q += 2) store8(col0 + 4 * ks_slot_of<SL>(q, ej), pk[q], pk[q + 1], rowp);
            if constexpr (SL & 1) {
                // the single slot: lanes (2 i, 2 i + 1) hold the two halves of one 8-column chunk; the even lane stores it
                const uint2 mine = pk[SL - 1];
                uint2 other;
                other.x = quad_xor1_u32(mine.x);
                other.y = quad_xor1_u32(mine.y);
                if ((ej & 1) == 0) store8(col0 + 4 * ks_slot_of<SL>(SL - 1, ej), mine, other, rowp);
            }
        };
        store_row(zrow);
        if constexpr (X == KS_COPY2) {   // A'' = bf16(h_new * zg2) -> zu2 (the out-block's [x | skip] operand, right half)
#pragma unroll
            for (int q = 0; q < SL; ++q) {
                pk[q].x = pack_bf2(v[q].x * dv[q][0], v[q].y * dv[q][1]);
                pk[q].y = pack_bf2(v[q].z * dv[q][2], v[q].w * dv[q][3]);
            }
            store_row(a.zu2 + (long)erow * a.ld_zu2);
        }
    }
    if (ts && lane == 0) {
        const unsigned long long t_iss = __builtin_readcyclecounter();   // every store of this wave is issued; below: landed (write-through: acknowledged by the memory side)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime();
        ts[5] = ((t_iss - t_start) << 32) | ((t_sum - t_start) & 0xffffffffull);
    }
}

}  // namespace
