// Co-resident bf16 MFMA GEMM for the wide projections (GEGLU-in, fused QKV):  C[M,N] = A[M,K] . W[N,K]^T  (nn.Linear layout, both K-contiguous)
//
// Same math and the same epilogues as k_gemm_pp (gemm_pp.h; reference: the aten::linear calls of src/models/utils/modules.py:263-277,341-374 and
// src/models/utils/attention.py:127-129), a different way of overlapping a workgroup's phases.
//
// Why: the ping-pong kernel is ONE 8-wave workgroup per CU (156 KB ring, 202 registers per wave).  Inside its K loop the two
// wave groups hide each other's LOAD phase, but nothing hides a workgroup's prologue (first tiles cold: ~8K cycles) and epilogue (GELU / LayerNorm
// math, park, copy-out: ~10K cycles) -- 45 % of a round at M = 4000, where the GEGLU GEMM is four rounds of workgroups.  Here a workgroup is HALF
// of that: 4 waves, a 128 x 144 tile, the SAME 32 x 144 wave tile (same LDS-read : MFMA ratio), a 2-deep ring (2 x 36 KB), the same ~200 registers
// per wave -- so that TWO workgroups are resident per CU (2 x 4 waves x 202 registers, 2 x 74 KB of LDS).  The two are independent kernels as far
// as the hardware is concerned: they de-phase on their own, one's LDS reads / barriers / DMA issue / prologue / epilogue run under the other's MFMAs, and when
// one exits the next workgroup of the grid starts beside the survivor -- the chip never has a round boundary.
//
// K loop of a workgroup (all four waves in lockstep; tile t lives in slot t & 1):
//      read ALL fragments of tile t into registers -> lgkmcnt(0) -> barrier (slot t & 1 is dead) -> LDS-DMA of tile t + 2 into it -> 36 MFMAs
//      -> counted vmcnt (own pieces of tile t + 1 landed) -> barrier
// The DMA of a tile has one MFMA phase plus one whole tile period to land.  LDS image, source-side bank swizzle and fragment addressing are k_gemm_pp's.
#pragma once
#include "gemm_pp.h"

namespace {

// dynamic LDS: ring (2 stages) | (mu, r) per row | G' / C' of the tile's columns | EPI_QKV: LayerNorm affine of the lanes' channels
template <int BM, int BN>
constexpr int co_smem_bytes() {
    return 2 * ((BM + BN + 31) / 32) * 4096 + BM * 8 + 2 * BN * 4 + 1024;   // (+ EPI_QKV: the LayerNorm-affine table, gemm_pp.h qkv_aff_*)
}

// VAR & 64: LayerNorm algebra in the epilogue (consumer side, GemmArgs.z*), as in k_gemm_pp
template <int BM, int BN, int EPI, int VAR>
__global__ __launch_bounds__(256, 2) void k_gemm_co(GemmArgs a) {
    constexpr int NT = 256;
    constexpr int WM = 4, TM = BM / WM, TN = BN, FM = TM / 16, FN = TN / 16;
    static_assert(TM * WM == BM && TM % 16 == 0 && TN % 16 == 0, "tile geometry: four waves, each TM rows x the whole tile width");
    constexpr int NP = (BM + BN + 31) / 32;    // 4-KB pieces (32 rows) per stage; every thread issues one 16-byte LDS-DMA per piece
    constexpr int PA = BM / 32;                // pieces [0, PA) come from A, [PA, NP) from W
    constexpr int STAGE = NP * 4096;
    static_assert(BM % 32 == 0, "whole A pieces");
    static_assert(2 * co_smem_bytes<BM, BN>() <= 160 * 1024, "TWO workgroups per CU");
    static_assert(NP < 32, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // (kernel arguments of the prologue in one batch, then the step counter as a plain scalar load: see k_gemm_pp)
    int M_ = a.M;
    asm("" : "+s"(M_) : "s"(a.A), "s"(a.W), "s"(a.lda), "s"(a.ldw), "s"(a.wrows), "s"(a.N), "s"(a.K), "s"(a.splitk), "s"(a.pm), "s"(a.pn), "s"(a.bm), "s"(a.bn), "s"(a.bz),
        "s"(a.cur_step), "s"(a.ts), "s"(a.row_slot), "s"(a.mbm), "s"(a.mbn), "s"(a.msplit));
    const int slot0 = a.cur_step ? *a.cur_step : 0;
    // G' / C' table bases PINNED in SGPRs: the per-thread choice between them (z_late_load) must be a select on two scalars -- left to hipcc it became a vector load of the chosen
    // pointer from the argument segment, and its `s_waitcnt vmcnt(0)` drained the first K tile's LDS-DMA in front of the G' / C' request (ISA of the first blind-load build of k_gemm_co)
    const float *zG_ = a.zG, *zC_ = a.zC;
    asm("" : "+s"(M_), "+s"(zG_), "+s"(zC_));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;

    const int tilesM = (M_ + BM - 1) / BM;
    const int tilesN = (a.N + BN - 1) / BN;
    int tm, tn, z;
    if (!tile_of_block(a, tilesM, tilesN, tm, tn, z)) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = a.K / BK;
    int kb, ke;
    ksplit_range(a, nk, z, kb, ke);
    const int nt = ke - kb;

    constexpr bool ZM = (VAR & 64) != 0 && (EPI == EPI_GEGLU || EPI == EPI_QKV);
    float* zgc = reinterpret_cast<float*>(smem + 2 * STAGE + BM * 8);     // [2][BN]: G' | C' of this tile's columns (shared modulation slot only)
    // LayerNorm algebra, consumer side (as in k_gemm_pp): the G' / C' slices of the tile's columns go straight into `zgc` by LDS-DMA right behind the first K tile and are not
    // waited for in front of the loop; the partial statistics of the rows a lane finishes are requested by that lane in front of the loop's last two tiles and merged behind
    // the loop (gemm_pp.h z_lane_load / z_lane_finish): nothing of the algebra sits between the prologue and the loop's first barrier
    constexpr int NZT = ZM ? FM * Z_PT : 0;   // blind statistics loads per lane (into AGPRs), issued by z_late_load
    ZLaneRegs<FM> zlr;
    const bool z_shared_slot = a.row_slot == nullptr;
    constexpr int ZNV = 2 * (BN / 4);   // threads that fetch one float4 of G' | C' by LDS-DMA straight into zgc[tid] (gemm_pp.h wait_younger_x)
    const bool zx = ZM && z_shared_slot && wave * 64 < ZNV;   // this wave issued that DMA: the counted wait in front of the loop leaves it (and the NZT statistics loads) in flight
    auto z_late_load = [&]() {
        if constexpr (ZM) {
            static_assert(2 * (BN / 4) <= NT, "one float4 of G' or C' per thread");
            if (z_shared_slot && tid < ZNV) {
                const int which = tid >= BN / 4, t4 = tid - which * (BN / 4);
                int cp = col0 + 4 * t4;
                cp = cp < a.N - 4 ? cp : a.N - 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((which ? zC_ : zG_) + (long)slot0 * a.zt_slot_stride + cp),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(zgc) + wave * 1024), 16, 0, 0);
            }
            z_lane_load<FM>(a.zstat_in, a.zs_stride, a.zparts, row0 + wm * TM + (lane & 15), a.M - 1, lane >> 4, zlr);   // blind loads into AGPRs, every lane
        }
    };
    unsigned long long* ts = (a.ts && wave == 0) ? a.ts + 8 * (long)blockIdx.x : nullptr;
    if (ts && lane == 0) { ts[0] = __builtin_readcyclecounter(); ts[6] = ez_stamp_start(); }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- loop-invariant addressing: byte offset of this thread's 16 bytes of piece p (K offset excluded); chunk c of row r lands in slot c ^ ((r >> 1) & 7)
    uint32_t poff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = p * NT + tid;
        const int row = q >> 3, c = q & 7;
        if (p < PA) {
            int grow = row0 + row;
            grow = grow < a.M ? grow : a.M - 1;
            poff[p] = (uint32_t)(grow * a.lda + ((c ^ ((row >> 1) & 7)) << 3)) * 2u;
        } else {
            const int r2 = row - BM;
            int gr = col0 + r2;
            gr = gr < a.wrows ? gr : a.wrows - 1;   // (the ragged last piece re-fetches a clamped row into the stage's padding: every wave issues NP loads per tile)
            poff[p] = (uint32_t)(gr * a.ldw + ((c ^ ((r2 >> 1) & 7)) << 3)) * 2u;
        }
    }
    const char* gA = reinterpret_cast<const char*>(a.A) + (long)kb * (BK * 2);
    const char* gW = reinterpret_cast<const char*>(a.W) + (long)kb * (BK * 2);
    auto issue = [&](int t) {
        char* dst = smem + (t & 1) * STAGE + wave * 1024;
        const long koff = (long)t * (BK * 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const char* src = (p < PA ? gA : gW) + koff + poff[p];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + p * 4096), 16, 0, 0);
        }
    };
    // fragment read offsets (k_gemm_pp): row (lane & 15) of a 16-row fragment, k-step ks (32 of K) -> 16-byte slot (4 ks + (lane >> 4)) ^ ((row >> 1) & 7)
    const int r16 = lane & 15, kq = lane >> 4;
    uint32_t foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = r16 * 128 + (((4 * ks + kq) ^ (r16 >> 1)) << 4);
    const int a_base = wm * TM * 128, b_base = BM * 128;
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // EXPERIMENT (VAR & 1 / & 2, microbenchmark only): the two workgroups of a CU start together and run in phase -- both in their LOAD phase, then both in their
    // MFMA phase.  1: the workgroup whose LDS allocation is not the first of the CU starts half a K-tile period late; 2: ... half a workgroup lifetime late, in the
    // first round of the grid only
    if constexpr ((VAR & 3) != 0) {
        const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6) ;   // HW_REG_LDS_ALLOC, LDS_BASE[7:0]
        if (lds_base != 0) {
            if constexpr ((VAR & 1) != 0) __builtin_amdgcn_s_sleep(12);
            if constexpr ((VAR & 2) != 0) { if (blockIdx.x < 512) { __builtin_amdgcn_s_sleep(100); __builtin_amdgcn_s_sleep(100); __builtin_amdgcn_s_sleep(100); } }
        }
    }
    // ---- prologue: tiles 0 and 1 in flight, the z requests between them; tile 0 (and the z loads) must have landed before the loop
    issue(0);
    z_late_load();
    if (nt > 1) issue(1);
    if (nt > 1) { if (zx) wait_vmcnt<NP + NZT + 1>(); else wait_vmcnt<NP + NZT>(); } else wait_vmcnt<0>();   // (the statistics loads and, zx, the G' | C' DMA -- issued between tile 0 and tile 1 -- stay in flight)
    barrier();
    if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
    // one K tile.  MODE 2: steady state (tile t + 2 exists and is issued here); 1: the last but one tile; 0: the last tile.  Compile-time, so that the
    // steady-state loop is straight-line code: with the `t + 2 < nt` tests inside ONE loop hipcc shuffled all 72 accumulators through VGPRs and spare
    // AGPRs on every trip (phi copies of the inline-asm MFMA operands across the branches: 140 v_accvgpr moves per K tile in the first build)
    auto step = [&](int t, auto MODE_) {
        constexpr int MODE = decltype(MODE_)::value;
        const char* cT = smem + (t & 1) * STAGE;
        bf16x8 af[FM][2], bfr[FN][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i][ks] = *reinterpret_cast<const bf16x8*>(cT + foff[ks] + a_base + i * 2048);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j][ks] = *reinterpret_cast<const bf16x8*>(cT + foff[ks] + b_base + j * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (MODE == 2) {
            barrier();                            // every wave has read slot t & 1
            issue(t + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bfr[j][ks]), "v"(af[i][ks]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE >= 1) {                // own pieces of tile t + 1 landed (tile t + 2 may stay in flight), then everyone's
            if constexpr (MODE == 2) wait_vmcnt<NP>(); else wait_vmcnt<0>();
            barrier();
        }
    };
    {
        int t = 0;
        for (; t + 2 < nt; ++t) step(t, std::integral_constant<int, 2>{});
        if (t + 1 < nt) { step(t, std::integral_constant<int, 1>{}); ++t; }
        step(t, std::integral_constant<int, 0>{});
    }
    if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();

    // ---- epilogue: the last MFMA's result is not interlocked against the VALU reads below (inline asm): 20 wait states tied to the accumulators
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
    float2 zmr[FM];   // LayerNorm algebra: (mu, r) of the rows this lane finishes
    if constexpr (ZM) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        z_lane_finish<FM>(zlr, a.zparts, lane >> 4, a.zD, a.zeps, zmr);
    }
    if constexpr (EPI == EPI_QKV) {
        // fused q | k | v projection, epilogue in registers (gemm_pp.h pp_store_qkv_reg): no k-split exchange here -- a wave holds its 32 rows x two whole heads
        static_assert(BM * (BN + 8) * 2 <= 2 * STAGE, "bf16 staging tile must fit the ring");
        QkvOperands<BN / 2, FM> qop;
        qkv_request<BN / 2, FM>(a, col0, row0 + wm * TM + (lane & 15), lane, tid, qop);
        pp_store_qkv_reg<BM, BN, BN / 2, FM, FN, TM, TN, NT, ZM>(a, acc, smem, row0, col0, wm, lane, tid, zmr, zgc, reinterpret_cast<float*>(smem + 2 * STAGE + BM * 8 + 2 * BN * 4), slot0, qop, ts);
        if (ts && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
    if constexpr (EPI == EPI_GEGLU || EPI == EPI_PARTIAL) {
        static_assert(BM * ((EPI == EPI_GEGLU ? BN / 2 : BN) + 8) * 2 <= 2 * STAGE, "output tile must fit the ring");
        if (EPI == EPI_GEGLU || a.part_bf16) {
            pp_store_lds<BM, BN, FM, FN, TM, TN, NT, EPI, ZM>(a, acc, smem, row0, col0, wm, 0, lane, tid, z, zmr, zgc, slot0, ts);
            if (ts && lane == 0) { ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
            return;
        }
    }
    if constexpr (EPI == EPI_F32 || EPI == EPI_PARTIAL) pp_store_direct<FM, FN, TM, TN, EPI>(a, acc, row0, col0, wm, 0, lane, z);
    if (ts && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
}

}  // namespace
