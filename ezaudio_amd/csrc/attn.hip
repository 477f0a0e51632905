// Flash-style attention for the DiT block: softmax(q k^T / sqrt(dh) [+ key mask]) v
// (F.scaled_dot_product_attention call of the reference, src/models/utils/attention.py:106-110; the boolean
// key mask of cross-attention is built at attention.py:30-37,131-135.)
//
// gfx950 design (v_mfma_f32_32x32x16_bf16 everywhere, fp32 softmax state):
//   * workgroup = 64 query rows of one (batch, head); 4 waves = 2 query sub-blocks (32 rows) x 2 key halves.
//     K and V^T are staged through LDS in 64-key tiles; wave (qs, kh) works on the 32 keys [32*kh, 32*kh+32) of
//     every tile and the two key halves are merged once at the end (log-sum-exp merge through LDS).  With
//     L = 500 this gives B*H*8 = 256 workgroups for one CFG pair: one per CU, all four SIMDs busy.
//   * scores are computed TRANSPOSED: S^T[key, q] = K . Q^T, so in the 32x32 C layout a lane owns ONE query
//     column (q = lane & 31) and 16 of the 32 keys of the tile.  The softmax row reductions are then
//     in-lane plus a single exchange with lane ^ 32 -- no LDS, no 5-step butterflies.
//   * the output is accumulated transposed as well: O^T[d, q] = V^T . P^T, so the online-softmax rescale
//     factor (per q) is lane-local for the accumulators, and P^T (B operand: lane = q, 8 keys per lane) is
//     exactly the S^T registers converted to bf16 -- P never leaves registers.  The MFMA contraction order
//     over keys is permuted accordingly (k-slot (hi, j) <-> key 16*step + 8*(j>>2) + 4*hi + (j&3)).  V is stored ROW-MAJOR
//     ([keys][DV], as the projection produces it) and the matching A fragment -- 4 + 4 consecutive KEYS of one channel -- is two
//     ds_read_b64_tr_b16 (gfx950's transposing LDS read: the 16 lanes of a group fetch a [4 keys][16 channels] block, 8 bytes each, and
//     every lane receives one channel's column of it).  Round 6; before, the producer stored V^T and the fragment was two plain 8-byte reads
//     (20-31 % of the LDS-active cycles were bank conflicts, and the QKV epilogue needed a transposing store path).
//   * staging is register-prefetched (issue the global loads of tile t+1, compute tile t out of LDS, then
//     write the registers to the other LDS buffer): one barrier per tile, HBM/L2 latency hidden behind the
//     MFMAs + softmax of a whole tile.  LDS row strides: K rows are padded by 16 B (conflict-free ds_read_b128); a V row is 192 B
//     (= DV bf16 at head_dim 72; head_dim 64 pads 128 -> 192): the 4 key rows x 64 B a half-wave's transposing read touches then fall
//     into the four quarters of the 64 banks.
//   * head_dim 72 (EzAudio-XL) is zero padded to 80 for the QK^T contraction (5 k-steps of 16) and to 96
//     output rows (3 tiles of 32) for P.V; head_dim 64 needs no padding.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// reductions over the lane pair (lane, lane ^ 32) by one VALU v_permlane32_swap (hipcc lowers __shfl_xor(x, 32) to ds_bpermute_b32: an
// LDS-pipe round trip of ~100+ cycles in the middle of the softmax dependency chain, twice per tile).  After swapping x with itself,
// r[0] keeps x in lanes 0..31 and holds x[lane - 32] in lanes 32..63; r[1] holds x[lane + 32] in lanes 0..31 and keeps x in lanes 32..63:
// in every lane {r[0], r[1]} = {own value, partner's value}.
__device__ __forceinline__ float pair_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int DH>
struct HeadGeom;
template <>
struct HeadGeom<64> { static constexpr int DQK = 64, DV = 64; };
template <>
struct HeadGeom<72> { static constexpr int DQK = 80, DV = 96; };

// NKH = key sub-blocks (of 32 keys) per staged tile = waves per query sub-block.  2: 64-key tiles, 4 waves.  4: 128-key tiles,
// 8 waves -- twice the waves per CU (one prompt = one workgroup per CU, so a SIMD otherwise hosts a single wave and its
// softmax VALU work never overlaps MFMAs) and half the tile-loop trips; needs Lkp % 128 == 0.
// ZQ: the fused projection finishes a LayerNorm in its epilogue (LayerNorm algebra, AttnArgs.z*): a template switch so that the plain kernel
// carries neither the registers nor the code
// QT = query rows per workgroup: 64, or 32 (the fused-projection form only, round 6): the projection is what a cross-attention launch spends its time on, and
// it is bound by what ONE CU can keep in flight -- 32-row tiles give every CU a workgroup at one prompt (256 instead of 128), 22 % fewer operand bytes
// per workgroup ((32 + 80) instead of (64 + 80) rows per K tile) and a ring of FOUR double stages (six K tiles in flight instead of four).  All eight
// waves split K in the projection (one 16-wide k-step of a double stage each); the attention proper runs on waves 0 - 3 (one query sub-block x 4 key sub-blocks)
template <int DH, int NKH, bool ZQ = false, int QT = 64>
__global__ __launch_bounds__(128 * NKH) void k_attn(AttnArgs a) {
    static_assert(QT == 64 || (QT == 32 && NKH == 4), "32-query tiles: the 8-wave fused-projection form only");
    constexpr int NQS = QT / 32;   // query sub-blocks (of 32 rows) per workgroup
    constexpr int NT = 128 * NKH;
    constexpr int TK = 32 * NKH;   // keys per staged tile
    constexpr int DQK = HeadGeom<DH>::DQK;
    constexpr int DV = HeadGeom<DH>::DV;
    constexpr int NKS = DQK / 16;  // k-steps of the QK^T contraction
    constexpr int NDT = DV / 32;   // 32-row tiles of O^T
    constexpr int KSTR = DQK * 2 + 16;   // LDS row stride of a K row (bytes)
    constexpr int VSTR = 192;            // LDS row stride of a V row (DV channels; see header)
    static_assert(DV * 2 <= VSTR, "V row");
    constexpr int KBYTES = TK * KSTR, VBYTES = TK * VSTR;
    constexpr int BUF = KBYTES + VBYTES;
    constexpr int KCH = TK * DQK * 2 / 16;   // 16-byte chunks of a K tile (contiguous in global memory)
    constexpr int VCPR = DV / 8;             // 16-byte chunks per V row
    constexpr int VCH = TK * VCPR;           // 16-byte chunks of a V tile (contiguous in global memory: rows are DV wide there)
    constexpr int KPT = (KCH + NT - 1) / NT, VPT = (VCH + NT - 1) / NT;
    static_assert(KPT <= 3 && VPT <= 3, "staging registers");
    // fused projection (NKH == 4): 3-deep ring of [64 rows of x | DN rows of W] K tiles, then 4 partial [64][DS] fp32 tiles
    constexpr int DN = DV >= 96 ? 96 : 64;    // W rows of a K tile's LDS slot (32-row MFMA fragments covering dh)
    constexpr int DW = DQK;                   // ... of which only the first DQK are fetched: rows [DQK, DN) produce projection columns that are dropped
                                              // below, so their LDS rows may hold anything (an MFMA output column depends on its own W row only) --
                                              // 10 instead of 12 DMA pieces per K tile at head_dim 72
    constexpr int DS = DQK;                   // columns kept of each partial
    constexpr int NRS = QT == 64 ? 3 : 4;                                   // ring slots of the projection
    constexpr int PSTAGE = (QT + DN) * 128, PRING = 2 * NRS * PSTAGE;       // ring slots of TWO K tiles each (a.xk2; QT = 64 without it: 3 of one)
    constexpr int PRED = 4 * 64 * DS * 4, PQS = QT * DQK * 2;               // partial tiles: [4][64][DS] or [8][32][DS] fp32
    constexpr int SMEM = NKH == 4 ? (2 * BUF > PRED + PQS ? (2 * BUF > PRING ? 2 * BUF : PRING) : (PRED + PQS > PRING ? PRED + PQS : PRING)) : 2 * BUF;
    constexpr int ZX = ZQ ? QT * 8 + 2 * DQK * 4 : 0;   // fused projection with the LayerNorm algebra: (mu, r) of the 64 query rows + G' | C' of the head
    __shared__ __attribute__((aligned(16))) char smem[SMEM + ZX];

    // kernel arguments in ONE batch (common.h "Kernel-argument batch": left alone, hipcc requests them in four dependent groups in front of the first load)
    int Lq_ = a.Lq;
    asm("" : "+s"(Lq_) : "s"(a.q), "s"(a.k), "s"(a.v), "s"(a.kmask), "s"(a.out), "s"(a.ldo), "s"(a.B), "s"(a.H), "s"(a.Lk), "s"(a.Lqp), "s"(a.Lkp), "s"(a.xu), "s"(a.ldu),
        "s"(a.xw), "s"(a.ldw), "s"(a.xw_rows), "s"(a.xK), "s"(a.xcd_map), "s"(a.nq), "s"(a.ppx), "s"(a.mnq), "s"(a.mH), "s"(a.q_raw), "s"(a.ts), "s"(a.xk2), "s"(a.zstat_in),
        "s"(a.zs_stride), "s"(a.zparts));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qs = QT == 64 ? (wave & 1) : 0, kh = QT == 64 ? (wave >> 1) : (wave & 3);   // kh in [0, NKH)
    const bool active = QT == 64 || __builtin_amdgcn_readfirstlane(wave) < 4;                   // 32-query tiles: waves 4 - 7 only project, stage and merge
    const int r32 = lane & 31, hi = lane >> 5;
    // workgroup -> (query tile, head, batch).  XCD-aware form: the dispatcher deals workgroup i to XCD i % 8, so XCD x gets the
    // ppx consecutive (batch, head) pairs [x * ppx, (x + 1) * ppx) with ALL their query tiles: a pair's K / V^T (and, in the fused
    // projection, its head's weight rows and its batch row's activations) is pulled into exactly one L2.
    int qt, h, b;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int sq = ez_div(sl, a.mnq);   // (division-free: common.h ez_div)
        const int pair = xcd * a.ppx + sq;
        qt = sl - sq * a.nq;
        if (pair >= a.B * a.H) return;   // whole workgroup, before any barrier
        b = ez_div(pair, a.mH); h = pair - b * a.H;
    } else {
        qt = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    }
    const int q0 = qt * QT + qs * 32;
    const long bh = (long)b * a.H + h;
    unsigned long long* ts = (a.ts && tid < 64) ? a.ts + 8 * (long)blockIdx.x : nullptr;
    if (ts && lane == 0) { ts[0] = __builtin_readcyclecounter(); ts[6] = ez_stamp_start(); }   // [6], [7]: 100 MHz device-wide clock

    const bf16_t* Q = a.q + (bh * a.Lqp + q0 + r32) * DQK + 8 * hi;
    const char* Kg = reinterpret_cast<const char*>(a.k + bh * a.Lkp * DQK);
    const bf16_t* Vg = a.v + bh * DV * (long)a.Lkp;
    const uint8_t* km = a.kmask ? a.kmask + (long)b * a.Lk : nullptr;

    bf16x8 qf[NKS];
    if (NKH == 4 && a.xu) {
        if constexpr (NKH == 4) {
            // ---- phase 1: Q_raw[QT][dh] = X[QT rows][K] . W_h[dh][K]^T.  QT = 64: wave = (mh, kq): rows 32 mh.., k-step kq of every K tile;
            // QT = 32: wave = k-step (wave & 3) of K tile (wave >> 2) of every double stage
            constexpr int FN = DN / 32;
            constexpr int NPART = QT == 64 ? 4 : 8;   // partial tiles summed in phase 1b
            const int rowb = b * a.Lq;
            // LayerNorm algebra: the partial statistics of the QT operand rows of this tile (the first 4 QT threads: FOUR threads per row, thread pg takes the
            // parts pg, pg + 4, pg + 8 of the part-major table: a wave's load covers 16 rows x 4 parts = four contiguous 128-byte runs) and G' | C' of this head's columns (one float4 per thread of wave 1) are requested right behind
            // the first two K tiles, land under the projection's K loop, and are merged / parked in LDS behind the loop's last barrier; used in phase 1b
            float2* zrow_l = reinterpret_cast<float2*>(smem + SMEM);            // [QT] (mu, r)
            float* zgc_l = reinterpret_cast<float*>(smem + SMEM + QT * 8);      // [2][DQK] G' | C'
            ZStatRegs zst;
            float4 zgc_reg = make_float4(0.f, 0.f, 0.f, 0.f);
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            const char* gA = reinterpret_cast<const char*>(a.xu);
            const char* gW = reinterpret_cast<const char*>(a.xw);
            const int nt = a.xK / 64;
            auto z_request = [&]() {
                if constexpr (ZQ) {
                    if (tid < 4 * QT) {   // four threads per row
                        int qr = qt * QT + (tid >> 2);
                        qr = qr < a.Lq ? qr : a.Lq - 1;
                        z_row_stats_load(a.zstat_in + (rowb + qr), a.zs_stride, a.zparts, tid & 3, zst);
                    }
                    if (tid < 2 * (DH / 4)) {
                        const int which = tid >= DH / 4, t4 = tid - which * (DH / 4);
                        zgc_reg = *reinterpret_cast<const float4*>((which ? a.zC : a.zG) + h * DH + 4 * t4);
                    }
                }
            };
            f32x16 acc[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            if constexpr (QT == 32) {
                // a K tile's LDS image: rows [32 of x | DW of W_h], 128 B each, as ONE run of (32 + DW) x 8 16-byte chunks dealt to the threads in two passes:
                // pass 0 = chunks [0, 512) (waves 0 - 3: the x rows, waves 4 - 7: W rows 0 .. 31), pass 1 = chunks [512, (32 + DW) 8) (W rows 32 ..; waves 0 .. NW1 - 1)
                constexpr int NCH = (32 + DW) * 8, NW1 = (NCH - NT) / 64;   // waves with a second piece per K tile
                static_assert(NCH > NT && NCH <= 2 * NT && (NCH - NT) % 64 == 0, "two passes, whole waves");
                uint32_t off0, off1;
                {
                    const int row = tid >> 3, c = tid & 7;   // chunk tid: row [0, 64)
                    if (row < 32) {
                        int grow = rowb + qt * 32 + row;
                        grow = grow < rowb + a.Lq - 1 ? grow : rowb + a.Lq - 1;
                        off0 = (uint32_t)(grow * a.ldu + ((c ^ ((row >> 1) & 7)) << 3)) * 2u;
                    } else {
                        const int r2 = row - 32;
                        int gr = h * DH + r2;
                        gr = gr < a.xw_rows - 1 ? gr : a.xw_rows - 1;
                        off0 = (uint32_t)(gr * a.ldw + ((c ^ ((r2 >> 1) & 7)) << 3)) * 2u;
                    }
                    const int r2 = (tid >> 3) + 32;           // chunk 512 + tid: W row 32 + (tid >> 3)
                    int gr = h * DH + r2;
                    gr = gr < a.xw_rows - 1 ? gr : a.xw_rows - 1;
                    off1 = (uint32_t)(gr * a.ldw + ((c ^ ((r2 >> 1) & 7)) << 3)) * 2u;
                }
                const char* g0 = wave_u < 4 ? gA : gW;      // wave-uniform base of pass 0
                const bool two = wave_u < NW1;
                auto stage2 = [&](int sidx) {                // double stage sidx = K tiles 2 sidx, 2 sidx + 1 -> ring slot sidx % NRS
                    char* dst = smem + (sidx % NRS) * 2 * PSTAGE + wave_u * 1024;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const long ko = (long)(2 * sidx + sub) * 128;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + ko + off0),
                                                         (__attribute__((address_space(3))) void*)(dst + sub * PSTAGE), 16, 0, 0);
                        if (two)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gW + ko + off1),
                                                             (__attribute__((address_space(3))) void*)(dst + sub * PSTAGE + NT * 16), 16, 0, 0);
                    }
                };
                const int ns = nt >> 1;                      // (launch_attention: the 32-query form needs an even number of K tiles)
                stage2(0);
                z_request();   // behind the first double stage: the first counted wait below covers them
                if (ns > 1) stage2(1);
                if (ns > 2) stage2(2);
                const int sub = wave_u >> 2, kq = wave_u & 3;
                const uint32_t fo = r32 * 128 + (((2 * kq + hi) ^ ((r32 >> 1) & 7)) << 4);
                for (int sidx = 0; sidx < ns; ++sidx) {
                    // stage sidx landed: at most the (up to two) younger stages of this wave stay in flight, 2 or 4 pieces each
                    const int younger = ns - 1 - sidx < NRS - 2 ? ns - 1 - sidx : NRS - 2;
                    if (two) {
                        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        if (younger >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    if (sidx + NRS - 1 < ns) stage2(sidx + NRS - 1);   // into the slot read in iteration sidx - 1: every wave is past those reads (its MFMAs consumed them before this barrier)
                    const char* cT = smem + (sidx % NRS) * 2 * PSTAGE + sub * PSTAGE;
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(cT + fo);
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(cT + 32 * 128 + fo + j * 4096);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[j]));
                }
            } else {
            const int mh = wave & 1, kq = wave >> 1;
            uint32_t aoff[1], boff[(DW * 8 + NT - 1) / NT];
            stage_offsets<64, NT>(aoff, a.ldu, rowb + qt * 64, rowb + a.Lq - 1, tid);
            stage_offsets<DW, NT>(boff, a.ldw, h * DH, a.xw_rows - 1, tid);
            auto stage = [&](int t) {
                char* dst = smem + (t % 3) * PSTAGE + wave_u * 1024;
                stage_tile<64, NT>(gA + (long)t * 128, aoff, dst, tid);
                stage_tile<DW, NT>(gW + (long)t * 128, boff, dst + 64 * 128, tid);
            };
            // loads per tile of this wave: 1 (x) + 1 or 2 (W: the second pass covers chunks [NT, DW * 8))
            const bool two = (DW * 8 > NT) && (wave_u * 64 + NT < DW * 8);
            stage(0);
            if (nt > 1) stage(1);
            // (behind the first two tiles: the requests overlap the first tile's wait; they only make the first counted wait below marginally stricter)
            z_request();
            const uint32_t fo = r32 * 128 + (((2 * kq + hi) ^ ((r32 >> 1) & 7)) << 4);
            if (a.xk2 && (nt & 1) == 0) {
                // double-width stages: one barrier and one counted wait per TWO K tiles (128 of K).  The per-tile work of a wave is 3 MFMAs, so
                // the loop is its fixed cost per iteration (wait + barrier + refill issue), which halves; the ring holds 3 x 40 KB.
                const int ns = nt >> 1;
                auto stage2 = [&](int sidx) {
                    char* dst = smem + (sidx % 3) * 2 * PSTAGE + wave_u * 1024;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        stage_tile<64, NT>(gA + (long)(2 * sidx + sub) * 128, aoff, dst + sub * PSTAGE, tid);
                        stage_tile<DW, NT>(gW + (long)(2 * sidx + sub) * 128, boff, dst + sub * PSTAGE + 64 * 128, tid);
                    }
                };
                if (ns > 1) stage2(1);   // the prologue above (tiles 0 and 1 into slots 0 and PSTAGE) IS stage 0 of this layout
                for (int sidx = 0; sidx < ns; ++sidx) {
                    if (sidx + 1 < ns) {
                        if (two) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    if (sidx + 2 < ns) stage2(sidx + 2);
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const char* cT = smem + (sidx % 3) * 2 * PSTAGE + sub * PSTAGE;
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(cT + fo + mh * 4096);
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(cT + 64 * 128 + fo + j * 4096);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[j], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[j]));
                }
            } else
            for (int t = 0; t < nt; ++t) {
                if (t + 1 < nt) {   // tile t + 1 may stay in flight
                    if (two) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (t + 2 < nt) stage(t + 2);
                const char* cT = smem + (t % 3) * PSTAGE;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(cT + fo + mh * 4096);
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(cT + 64 * 128 + fo + j * 4096);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[j]));
            }
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc[j]));
            if constexpr (ZQ) {   // behind the two barriers below: read in phase 1b
                if (tid < 4 * QT) {
                    const float2 mr = z_row_stats_finish(zst, a.zparts, tid & 3, a.zD, a.zeps);
                    if ((tid & 3) == 0) zrow_l[tid >> 2] = mr;
                }
                if (tid < 2 * (DH / 4)) {
                    const int which = tid >= DH / 4, t4 = tid - which * (DH / 4);
                    *reinterpret_cast<float4*>(zgc_l + which * DQK + 4 * t4) = zgc_reg;
                }
            }
            __syncthreads();   // the ring is dead: it becomes the reduction area
            // lane owns row prow and columns 32 j + 8 g + 4 hi + {0..3} of partial pw
            float* red = reinterpret_cast<float*>(smem);   // [NPART][QT][DS]
            {
                const int pw = QT == 64 ? (wave >> 1) : wave, prow = QT == 64 ? 32 * (wave & 1) + r32 : r32;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = 32 * j + 8 * g + 4 * hi;
                        if (col < DS)
                            *reinterpret_cast<float4*>(red + ((pw * QT + prow) * DS + col)) =
                                make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
                    }
            }
            __syncthreads();
            // ---- phase 1b: sum the NPART partials (fixed order), per-head LayerNorm (attention.py:141), bf16 -> Qs[QT][DQK]; 8 threads per row
            if (QT == 64 || tid < 8 * QT) {
                constexpr int CP = DH / 8;   // columns per thread
                bf16_t* qs_l = reinterpret_cast<bf16_t*>(smem + PRED);
                const int row = tid >> 3, part = tid & 7;
                float v[CP];
                float s1 = 0.f;
                const float2 zmr = ZQ ? zrow_l[row] : make_float2(0.f, 1.f);
                const float zmu = zmr.x, zr = zmr.y;
#pragma unroll
                for (int e = 0; e < CP; ++e) {
                    const int col = part * CP + e;
                    if constexpr (QT == 64) {
                        v[e] = red[(0 * 64 + row) * DS + col] + red[(1 * 64 + row) * DS + col] + red[(2 * 64 + row) * DS + col] +
                               red[(3 * 64 + row) * DS + col];
                    } else {
                        float t = red[row * DS + col];
#pragma unroll
                        for (int w = 1; w < NPART; ++w) t += red[(w * QT + row) * DS + col];
                        v[e] = t;
                    }
                    if constexpr (ZQ) v[e] = fmaf(zr, v[e], fmaf(-zr * zmu, zgc_l[col], zgc_l[DQK + col]));
                    s1 += v[e];
                }
                s1 = oct_sum(s1);   // the 8 threads of a row are an aligned octet (DPP, common.h)
                const float mean = s1 * (1.f / DH);
                float s2 = 0.f;
#pragma unroll
                for (int e = 0; e < CP; ++e) { const float d = v[e] - mean; s2 += d * d; }
                s2 = oct_sum(s2);
                const float rstd = rsqrtf(s2 * (1.f / DH) + 1e-5f);
#pragma unroll
                for (int e = 0; e < CP; ++e) {
                    const int col = part * CP + e;
                    qs_l[row * DQK + col] = f2bf((v[e] - mean) * rstd * a.qn_w[col] + a.qn_b[col]);
                }
                if (DQK > DH && part == 0)
                    for (int col = DH; col < DQK; ++col) qs_l[row * DQK + col] = 0;
            }
            __syncthreads();
            {
                const bf16_t* qs_l = reinterpret_cast<const bf16_t*>(smem + PRED);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qs_l + (32 * qs + r32) * DQK + 16 * ks + 8 * hi);
            }
            __syncthreads();   // Qs and the reduction area are dead: the K / V staging buffers may overwrite them
        }
    } else if (a.q_raw) {
        // lane (r32, hi) holds d = 16 ks + 8 hi + e of query row q0 + r32: the whole head sits in the lane pair (lane, lane ^ 32)
        int qr = q0 + r32;
        if (qr >= a.Lq) qr = a.Lq - 1;   // padding rows: any valid row, the result is never stored
        const float* src = a.q_raw + ((long)b * a.Lq + qr) * a.ld_qraw + h * DH + 8 * hi;
        float v[NKS][8];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bool ok = 16 * ks + 8 * hi < DH;   // DH is a multiple of 8: chunks are all-valid or all-padding
            const float4 lo = ok ? *reinterpret_cast<const float4*>(src + 16 * ks) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 hh = ok ? *reinterpret_cast<const float4*>(src + 16 * ks + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[ks][0] = lo.x; v[ks][1] = lo.y; v[ks][2] = lo.z; v[ks][3] = lo.w;
            v[ks][4] = hh.x; v[ks][5] = hh.y; v[ks][6] = hh.z; v[ks][7] = hh.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[ks][e];
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.f / DH);
        float q2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            if (16 * ks + 8 * hi < DH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q2 += d * d; }
            }
        q2 += __shfl_xor(q2, 32, 64);
        const float rstd = rsqrtf(q2 * (1.f / DH) + 1e-5f);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = 16 * ks + 8 * hi;
            const bool ok = d0 < DH;
            const int dw = ok ? d0 : 0;
            union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y0 = ok ? (v[ks][2 * e] - mean) * rstd * a.qn_w[dw + 2 * e] + a.qn_b[dw + 2 * e] : 0.f;
                const float y1 = ok ? (v[ks][2 * e + 1] - mean) * rstd * a.qn_w[dw + 2 * e + 1] + a.qn_b[dw + 2 * e + 1] : 0.f;
                pk.u[e] = pack_bf2(y0, y1);
            }
            qf[ks] = pk.v;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + 16 * ks);
    }

    // staging registers: NAMED scalars, not an array (hipcc demotes a register array that is live across the tile loop
    // to scratch); every thread issues all its loads unconditionally (chunk index clamped), only LDS writes are predicated
    uint4 k0 = {}, k1 = {}, k2 = {}, v0 = {}, v1 = {}, v2 = {};
    // per-thread byte offsets of the staged chunks, computed ONCE; the tile base pointers below are wave-uniform, so the loads take the
    // SGPR-base + 32-bit VGPR-offset form and the loop spends no VALU work on 64-bit addresses
    uint32_t koff[3], voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int ck = tid + NT * i, cv = tid + NT * i;
        ck = ck < KCH ? ck : KCH - 1;
        cv = cv < VCH ? cv : VCH - 1;
        koff[i] = (uint32_t)ck * 16u;
        voff[i] = (uint32_t)cv * 16u;
    }
    // buffer loads: descriptor (SGPRs, wave-uniform) + the per-thread 32-bit offset above + the tile offset as the scalar soffset -> no
    // per-tile address arithmetic at all
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kg), 0, a.Lkp * DQK * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Vg), 0, DV * a.Lkp * 2, 0x00020000);
    auto kchunk = [&](int i, int key0) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(krs, (int)koff[i], key0 * DQK * 2, 0));
    };
    auto vchunk = [&](int i, int key0) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(vrs, (int)voff[i], key0 * DV * 2, 0));
    };
    auto kstore = [&](int i, char* kb, const uint4& val) {
        const int c = tid + NT * i;
        if (c < KCH) *reinterpret_cast<uint4*>(kb + (c / (DQK / 8)) * KSTR + (c % (DQK / 8)) * 16) = val;
    };
    auto vstore = [&](int i, char* vb, const uint4& val) {
        const int c = tid + NT * i;
        if (c < VCH) *reinterpret_cast<uint4*>(vb + (c / VCPR) * VSTR + (c % VCPR) * 16) = val;
    };
#define LOAD_TILE(t)                                   \
    do {                                               \
        const int key0_ = (t) * TK;                    \
        k0 = kchunk(0, key0_);                         \
        k1 = kchunk(1, key0_);                         \
        if constexpr (KPT > 2) k2 = kchunk(2, key0_);  \
        v0 = vchunk(0, key0_);                         \
        v1 = vchunk(1, key0_);                         \
        if constexpr (VPT > 2) v2 = vchunk(2, key0_);  \
    } while (0)
#define WRITE_TILE(buf)                                \
    do {                                               \
        char* kb_ = smem + (buf) * BUF;                \
        char* vb_ = kb_ + KBYTES;                      \
        kstore(0, kb_, k0);                            \
        kstore(1, kb_, k1);                            \
        if constexpr (KPT > 2) kstore(2, kb_, k2);     \
        vstore(0, vb_, v0);                            \
        vstore(1, vb_, v1);                            \
        if constexpr (VPT > 2) vstore(2, vb_, v2);     \
    } while (0)

    f32x16 o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -1e30f, lsum = 0.f;
    constexpr float RESCALE_THR = 4.0f;  // log2 units: P <= 16
    const float c = 1.4426950408889634f * rsqrtf((float)DH);  // log2(e) / sqrt(dh)

    // two waves per SIMD: the later-dispatched half loses VALU arbitration to the older half (priority, then age); one static priority
    // step for it evens the pair out (MI355X_MICROARCH.md "Two waves per SIMD").  The guard must be provably wave-uniform: s_setprio
    // ignores EXEC.
    if (NKH == 4 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    const int ntiles = (a.Lk + TK - 1) / TK;
    LOAD_TILE(0);
    WRITE_TILE(0);
    __syncthreads();
    if (ts && lane == 0) ts[1] = __builtin_readcyclecounter();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) LOAD_TILE(t + 1);
        const char* kb = smem + (t & 1) * BUF;
        const char* vb = kb + KBYTES;
        const char* vtr = vb + (32 * kh + 4 * hi + ((lane & 15) >> 2)) * VSTR + ((lane & 16) + 4 * (lane & 3)) * 2;   // this lane's corner of the transposing reads
        const int key0 = t * TK + kh * 32;
        if (active && key0 < a.Lk) {  // wave-uniform: the whole 32-key sub-tile may lie beyond Lk (32-query tiles: waves 4 - 7 only stage)
            // validity of this wave's 32 keys as one bit mask (bit j <-> key0 + j), built BEFORE the MFMAs
            const int kidx = key0 + r32;
            bool kv = kidx < a.Lk;
            if (km) kv = kv && (km[kidx < a.Lk ? kidx : 0] != 0);
            const uint32_t ball = (uint32_t)__ballot(kv);               // lanes 0..31 fill bits 0..31
            const uint32_t tmask = ball >> (4 * hi);
            // wave-uniform: no key of this sub-tile is masked or beyond Lk (every self-attention tile but the last).  The loop is bound
            // by VALU issue (~300 instructions per wave and tile against 11 MFMAs, tools/microbench/attn_bench.cpp), so the common case
            // drops the per-element mask selects and folds scale and reference max into one FMA feeding exp2.
            const bool full = ball == 0xffffffffu;
            // S^T in VGPRs (the softmax reads every element once: through AGPRs that is 16 v_accvgpr_read per tile), written by inline-asm
            // MFMAs so that the register class is ours to choose.  The wait states hipcc would insert are inside the strings: 2 in front
            // (an operand may have just been written by a VALU or DS instruction the compiler scheduled right before the statement) and
            // 20 behind the last one (its final pass writes s[12..15]; any reader but a chained MFMA must stay that far behind).
            f32x16 s;
            {
                const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kb + (kh * 32 + r32) * KSTR + hi * 16);
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf0), "v"(qf[0]));
            }
#pragma unroll
            for (int ks = 1; ks < NKS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + (kh * 32 + r32) * KSTR + (2 * ks + hi) * 16);
                if (ks + 1 < NKS) asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "v"(qf[ks]));
                else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s) : "v"(kf), "v"(qf[ks]));
            }
            // lane holds S^T[key0 + (r&3) + 8*(r>>2) + 4*hi][q0 + r32]
            float tmax = -1e30f;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax *= c;   // c > 0: max(c s) == c max(s), rounding included
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool v = (tmask >> ((r & 3) + 8 * (r >> 2))) & 1u;
                    s[r] = v ? s[r] * c : -1e30f;
                    tmax = fmaxf(tmax, s[r]);
                }
            }
            tmax = pair_max(tmax);
            // deferred rescale: keep the old reference max while the tile max exceeds it by < 2^-? ... THR (log2 units);
            // P is then bounded by 2^THR instead of 1, which fp32 sums / bf16 P absorb; the accumulators are touched
            // only when some row really needs a new reference (wave-uniform branch).
            if (!__all(tmax <= m + RESCALE_THR)) {
                const float mnew = fmaxf(m, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                m = mnew;
                lsum *= alpha;
#pragma unroll
                for (int tt = 0; tt < NDT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
            }
            float psum = 0.f;
            float p[16];
            if (full) {
                const f32x2 c2 = {c, c}, nm2 = {-m, -m};
                f32x2 ps2 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 x = {s[r], s[r + 1]};
                    x = __builtin_elementwise_fma(x, c2, nm2);
                    p[r] = __builtin_amdgcn_exp2f(x[0]);
                    p[r + 1] = __builtin_amdgcn_exp2f(x[1]);
                    const f32x2 pp = {p[r], p[r + 1]};
                    ps2 += pp;
                }
                psum = ps2[0] + ps2[1];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool v = (tmask >> ((r & 3) + 8 * (r >> 2))) & 1u;
                    p[r] = v ? __builtin_amdgcn_exp2f(s[r] - m) : 0.f;
                    psum += p[r];
                }
            }
            psum = pair_sum(psum);
            lsum += psum;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf2(p[8 * step + 2 * e], p[8 * step + 2 * e + 1]);
#pragma unroll
                for (int tt = 0; tt < NDT; ++tt) {
                    // A fragment: V[keys 32kh + 16step + 4hi + {0..3} and ... + 8 + {0..3}][channel 32tt + r32], gathered by two transposing reads: the 16 lanes
                    // of a group (same hi, same half of the 32 channels) address the [4 keys][16 channels] block row by row -- lane i of the group: key i >> 2,
                    // channels 4 (i & 3) .. + 3 -- and lane i receives channel i of all four keys (measured layout: tools/microbench/tr_probe.hip)
                    const char* vp = vtr + (16 * step) * VSTR + 64 * tt;
                    union { bf16x8 v; s16x4 h2[2]; } vf;
                    vf.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
                    vf.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * VSTR));
                    // register classes by constraint: O^T (and S^T above) in VGPRs -- the rescale and the softmax are VALU code, which cannot
                    // address AGPRs; with the builtin hipcc parked O^T in AGPRs and copied all 16 NDT values out and back every tile.  The staging
                    // registers and part of the Q fragments end up in the AGPR half (hipcc's choice under the 128-VGPR cap)
                    asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(o[tt]) : "v"(vf.v), "v"(pf.v));
                }
            }
        }
        if (t + 1 < ntiles) WRITE_TILE((t + 1) & 1);
        __syncthreads();
    }

    if (ts && lane == 0) ts[2] = __builtin_readcyclecounter();
    // ---- merge the key sub-blocks (log-sum-exp) and store, spread over ALL waves ----
    // Every wave parks its partial (m, l, O^T) in LDS; after one barrier the (query sub-block, 8-column group) items of the tile are dealt
    // round-robin to the waves, so each wave sums NKH partials for 2-3 items and issues 2-3 eight-byte stores per lane.  (Before: the
    // waves kh >= 1 parked, the two kh = 0 waves merged NKH - 1 partners one after the other -- 3 x 48 LDS reads + FMAs per lane -- and
    // issued 12 stores per lane while six waves had already left; merge + store were 4.7K of the 19.4K cycles of a launch.)
#pragma unroll
    for (int tt = 0; tt < NDT; ++tt) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(o[tt]));
    constexpr int NG = DH / 8;                                        // column groups: G <-> d = 8 G + 4 hi + {0..3} = (tt = G / 4, g = G % 4)
    constexpr int NW = 2 * NKH;                                        // waves of the workgroup: all of them merge and store
    float4* xo = reinterpret_cast<float4*>(smem);                      // [NKH][NQS][NG][64 lanes]
    float* xml = reinterpret_cast<float*>(smem + NKH * NQS * NG * 64 * 16);   // [NKH][NQS][2][32]
    static_assert(SMEM >= NKH * NQS * NG * 64 * 16 + NKH * NQS * 2 * 32 * 4, "exchange area must fit the staging buffers");
    if (active) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            xo[((kh * NQS + qs) * NG + G) * 64 + lane] = make_float4(o[G / 4][4 * (G % 4)], o[G / 4][4 * (G % 4) + 1], o[G / 4][4 * (G % 4) + 2], o[G / 4][4 * (G % 4) + 3]);
        if (hi == 0) {
            xml[((kh * NQS + qs) * 2 + 0) * 32 + r32] = m;
            xml[((kh * NQS + qs) * 2 + 1) * 32 + r32] = lsum;
        }
    }
    __syncthreads();
    if (ts && lane == 0) ts[4] = __builtin_readcyclecounter();
    // weights of the NKH partials for query row q0 + r32 (the items of a wave all belong to its own query sub-block: NW is even)
    float wgt[NKH];
    {
        float mk[NKH], mm = -1e30f;
#pragma unroll
        for (int k = 0; k < NKH; ++k) { mk[k] = xml[((k * NQS + qs) * 2 + 0) * 32 + r32]; mm = fmaxf(mm, mk[k]); }
        float l = 0.f;
#pragma unroll
        for (int k = 0; k < NKH; ++k) { wgt[k] = __builtin_amdgcn_exp2f(mk[k] - mm); l = fmaf(xml[((k * NQS + qs) * 2 + 1) * 32 + r32], wgt[k], l); }
        const float inv = 1.f / l;
#pragma unroll
        for (int k = 0; k < NKH; ++k) wgt[k] *= inv;
    }
    const int qrow = q0 + r32;
    bf16_t* orow = a.out + ((long)b * a.Lq + (qrow < a.Lq ? qrow : a.Lq - 1)) * a.ldo + h * DH + 4 * hi;
#pragma unroll
    for (int i = 0; i < (NQS * NG + NW - 1) / NW; ++i) {
        const int G = (wave + i * NW) / NQS;   // item = wave + i NW = NQS G + qs
        if (G < NG) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < NKH; ++k) {
                const float4 v = xo[((k * NQS + qs) * NG + G) * 64 + lane];
                acc.x = fmaf(v.x, wgt[k], acc.x); acc.y = fmaf(v.y, wgt[k], acc.y); acc.z = fmaf(v.z, wgt[k], acc.z); acc.w = fmaf(v.w, wgt[k], acc.w);
            }
            if (qrow < a.Lq) {
                uint2 v;
                v.x = pack_bf2(acc.x, acc.y);
                v.y = pack_bf2(acc.z, acc.w);
                if (a.wt) st8_wt(orow + 8 * G, v); else *reinterpret_cast<uint2*>(orow + 8 * G) = v;
            }
        }
    }
    if (ts && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[3] = __builtin_readcyclecounter(); ts[7] = __builtin_amdgcn_s_memrealtime(); }
}

#undef LOAD_TILE
#undef WRITE_TILE

}  // namespace

int launch_attention(const AttnArgs& a0, hipStream_t st) {
    AttnArgs a = a0;
    if (a.b0 > 0) {   // batch sub-range [b0, b0 + B): advance every per-batch-element pointer; the kernel sees batch elements 0 .. B - 1
        const int DQK = a.dh == 64 ? 64 : 80, DV = a.dh == 64 ? 64 : 96;
        const long b0 = a.b0;
        if (a.q) a.q += b0 * a.H * a.Lqp * DQK;
        if (a.k) a.k += b0 * a.H * a.Lkp * DQK;
        if (a.v) a.v += b0 * a.H * DV * (long)a.Lkp;
        if (a.kmask) a.kmask += b0 * a.Lk;
        if (a.out) a.out += b0 * a.Lq * a.ldo;
        if (a.q_raw) a.q_raw += b0 * a.Lq * a.ld_qraw;
        if (a.xu) a.xu += b0 * a.Lq * a.ldu;
        if (a.zstat_in) a.zstat_in += b0 * a.Lq;
        a.b0 = 0;
    }
    const long nwg64 = (long)((a.Lq + 63) / 64) * a.H * a.B;
    int nkh = a.nkh;
    if (nkh != 2 && nkh != 4) nkh = nwg64 <= 512 ? 4 : 2;   // 0 = choose by grid size
    if (a.Lkp % 128) nkh = 2;
    if (a.xu && nkh != 4) return 1;   // the fused projection exists in the 8-wave form only (needs Lkp % 128 == 0)
    if (a.dh != 64 && a.dh != 72) return 1;
    const bool zq = a.xu && a.zstat_in;
    if (zq && !(a.zG && a.zC && a.zparts > 0 && a.zparts <= Z_MAXP && a.zs_stride > 0 && a.zw > 0)) return 1;
    // 32-query tiles (fused projection with the LayerNorm algebra, an even number of K tiles): when the 64-row grid leaves half the CUs without a workgroup
    // (one prompt with the single-key shortcut: 128), or when asked for (qtile = 32)
    const bool q32 = zq && nkh == 4 && (a.xK / 64) % 2 == 0 && a.xcd_map && (a.qtile == 32 || (a.qtile == 0 && nwg64 <= 128));
    const int QT = q32 ? 32 : 64;
    a.nq = (a.Lq + QT - 1) / QT;
    a.ppx = (a.B * a.H + 7) / 8;
    a.mnq = ez_magic(a.nq); a.mH = ez_magic(a.H);
    dim3 grid(a.nq, a.H, a.B);
    if (a.xcd_map) grid = dim3(8 * a.ppx * a.nq, 1, 1);
    if (a.ts && (long)grid.x * grid.y * grid.z > a.ts_cap) a.ts = nullptr;   // the stamp buffer has no room for this grid
    if (q32) {
        if (a.dh == 64) hipLaunchKernelGGL((k_attn<64, 4, true, 32>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attn<72, 4, true, 32>), grid, dim3(512), 0, st, a);
        return 0;
    }
    if (a.dh == 64) {
        if (nkh == 4 && zq) hipLaunchKernelGGL((k_attn<64, 4, true>), grid, dim3(512), 0, st, a);
        else if (nkh == 4) hipLaunchKernelGGL((k_attn<64, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attn<64, 2>), grid, dim3(256), 0, st, a);
    } else {
        if (nkh == 4 && zq) hipLaunchKernelGGL((k_attn<72, 4, true>), grid, dim3(512), 0, st, a);
        else if (nkh == 4) hipLaunchKernelGGL((k_attn<72, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attn<72, 2>), grid, dim3(256), 0, st, a);
    }
    return 0;
}
