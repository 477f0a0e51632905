// Shared device helpers and host launcher declarations for libezaudio_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;  // storage type

#define EZ_WAVE 64

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// native conversions: one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving) instead of a branchy bit trick
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// write-through stores (sc1): the line does not stay dirty in this XCD's L2, so the end-of-kernel write-back that the next
// kernel's launch waits for has nothing left to flush (every kernel here is consumed by all XCDs of the next one)
// (inline asm: hipcc pads nothing behind the statement, so the wait state a store of more than 8 bytes needs before a following instruction
// may overwrite its data registers is inside the string)
__device__ __forceinline__ void st16_wt(void* p, float4 v) {
    f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void st8_wt(void* p, uint2 v) {
    u32x2_t x = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void st4_wt(void* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// test-hook stamp [6]: the 100 MHz device-wide clock at kernel start, tagged in its top 16 bits with the CU the workgroup runs on
// (XCC_ID[3:0] << 8 | HW_ID[15:8] = SE / SH / CU): tools/microbench/gemm_bench.cpp rebuilds every CU's timeline from it (who shared a CU with whom, idle gaps between workgroups)
__device__ __forceinline__ unsigned long long ez_stamp_start() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned long long id = ((unsigned long long)(xcc & 15u) << 8) | ((hw >> 8) & 255u);
    return (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffull) | (id << 48);
}

// Division of a small non-negative integer by a launch constant WITHOUT the ~35-instruction integer division (v_rcp_iflag + readfirstlane + two correction
// steps, on the scalar unit of a prologue that has nothing else to do: the workgroup -> tile maps of the GEMM and attention kernels ran 4 - 8 of them in a
// row, ~1K cycles in front of the first LDS-DMA of every launch).  The host passes m = ceil(2^32 / d) (0 for d = 1); x / d = mulhi(x, m) exactly while x d < 2^32.
inline unsigned ez_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int ez_div(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

// Kernel-argument batch of a GEMM prologue (k_gemm_pp, k_gemm_co, k_gemm_ks):  int lda = a.lda;  asm("" : "+s"(lda) : "s"(a.A), "s"(a.W), ...);  makes hipcc
// request every listed argument BEFORE the first use of `lda` (a non-volatile asm that "modifies" it and reads the others).  Left alone, hipcc sinks each scalar load towards its first use and the prologue
// walks a chain of serialised kernarg round trips in front of its first LDS-DMA.  NOT `asm volatile`: hipcc treats that as a possible store, and the device
// step counter's load behind it (GemmArgs.cur_step: global memory) turned from a scalar load into a VECTOR load with `s_waitcnt vmcnt(0)` right behind it
// (ISA of the first round-6 build); an inline-asm s_load for the counter is no way out either -- the allocator copied its destination before the tied wait
// (`s_mov_b32 s20, s53`: tests/test_host.py::test_step_counter_scalar_load_is_not_touched_before_its_wait was written for it and caught it).

int ez_fail(int code, const char* fmt, ...);   // api.hip: record the message behind ezdit_last_error(), return code

// ------------------------------------------------------------------------------------------
// host-side launchers (one per kernel family); all asynchronous on `st`
// ------------------------------------------------------------------------------------------
enum GemmEpi {
    EPI_F32 = 0,      // out fp32 [M][ldo] = acc (+ bias[col])
    EPI_PARTIAL = 1,  // out fp32 slab z: [z][Mp][ldo] = acc          (split-K partials)
    EPI_GEGLU = 2,    // out bf16 [M][ldo]: (val + b) * gelu_erf(gate + b), W rows interleaved 8 value / 8 gate
    // un-split residual GEMM (k_gemm_ks): out (fp32) = resid + gate * (acc + bias) as EPI_F32, AND the bf16 operand of the next
    // GEMM, A' = h_new * zg (the LayerNorm gain only), AND per-(row, column tile) partial statistics of h_new: the LayerNorm itself
    // is finished by the CONSUMER (GemmArgs.z*, "LN algebra" in DESIGN.md).  Replaces split-K slabs + the row kernel.
    EPI_RESID = 4,
    EPI_QKV = 3       // fused q|k|v projection (tile 64 x 4 whole heads: 64x288 for head_dim 72, 64x256 for 64): per-head LayerNorm + RoPE of q / k and
                      // V -> V^T straight into the attention layouts through LDS (GemmArgs.hn); nothing is written to `out`
};

// ---- LDS-DMA staging of a bf16 operand tile with K = 64 (one 128-byte LDS row per tile row), shared by gemm.hip and attn.hip ----
constexpr int BK = 64;
constexpr int Z_MAXP = 12;   // LayerNorm algebra: column tiles (partial statistics) a row may have: D <= 12 x the producer's tile width (1152 = 12 x 96)

// ---- LayerNorm algebra (GemmArgs.z*), consumer side: (mu, r) of one row from the partial statistics its producer left per column tile:
// (S_k, Q_k) = (sum, sum of squares) of the row over the tile's columns, so that merging is two plain sums:
//      mu = sum_k S_k / D,   var = sum_k Q_k / D - mu^2
// (the textbook one-pass form: its relative error on var is eps_fp32 (1 + mu^2 / var) -- harmless while a row's mean is not orders of
// magnitude above its spread, which holds for the residual stream; the (sum, M2-about-the-tile-mean) form of round 3 needed a 12-term Chan
// merge with a data-dependent correction per part: 130 instructions on the one wave every other wave of the workgroup waits for).
// The statistics are stored PART-MAJOR ([part][row], GemmArgs.zs_stride rows apart): FOUR neighbouring threads per row load its parts (thread
// pg of a row takes the parts pg, pg + 4, pg + 8: a wave's load instruction covers 16 rows x 4 parts = four contiguous 128-byte runs) at
// kernel start -- 6 registers per thread ride through the K loop -- and merge them after it with two xor-shuffles: nothing at kernel start
// waits on them (round 3 let four threads per row gather 8-byte pieces of a row-major table and wait for them before the first LDS-DMA).
constexpr int Z_PT = Z_MAXP / 4;   // parts per thread
static_assert(Z_PT * 4 == Z_MAXP, "four threads per row");
struct ZStatRegs { float2 v[Z_PT]; };
// xor-shuffles inside a quad by DPP (hipcc lowers __shfl_xor to ds_bpermute: an LDS-pipe round trip)
__device__ __forceinline__ float quad_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); }
// sum over the 8 neighbouring lanes of an aligned octet, in every lane, in the order of  s += xor 1; s += xor 2; s += xor 4  (bit-identical to the
// __shfl_xor form): after the two quad steps all four lanes of a quad hold the quad's sum, and row_half_mirror (lane i <- lane 7 - i) hands every
// lane the OTHER quad's
__device__ __forceinline__ float oct_sum(float s) {
    s += quad_xor1(s);
    s += quad_xor2(s);
    return s + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t quad_xor1_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true); }
// The loads are UNCONDITIONAL (part index clamped) and their results are not touched here: a select on a loaded value made hipcc wait for
// the loads at kernel start, in front of the first LDS-DMA (k_attn: +1.7 us per launch); the unused parts are masked in z_row_stats_finish
__device__ __forceinline__ void z_row_stats_load(const float2* __restrict__ st /* + row */, long stride, int parts, int pg, ZStatRegs& z) {
#pragma unroll
    for (int k = 0; k < Z_PT; ++k) {
        const int p = pg + 4 * k;
        z.v[k] = st[(p < parts ? p : parts - 1) * stride];
    }
}
// all four threads of the row (lanes pg = 0 .. 3, neighbours) return (mu, r)
__device__ __forceinline__ float2 z_row_stats_finish(ZStatRegs& z, int parts, int pg, int D, float eps) {
    // nothing below may be hoisted above this point (hipcc moved the first addition up to the loads and waited for them at kernel start)
#pragma unroll
    for (int k = 0; k < Z_PT; ++k) asm volatile("" : "+v"(z.v[k].x), "+v"(z.v[k].y));
    const float inv_d = __builtin_amdgcn_rcpf((float)D);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < Z_PT; ++k) {   // fixed order: bit-reproducible
        const bool on = pg + 4 * k < parts;
        s += on ? z.v[k].x : 0.f;
        q += on ? z.v[k].y : 0.f;
    }
    s += quad_xor1(s); s += quad_xor2(s);   // the four threads of a row are one quad
    q += quad_xor1(q); q += quad_xor2(q);
    const float mu = s * inv_d;
    const float var = fmaxf(fmaf(q, inv_d, -mu * mu), 0.f);
    return make_float2(mu, rsqrtf(var + eps));
}

// Per-thread byte offsets of the 16-byte chunks this thread stages for one operand tile (K offset excluded): computed
// ONCE per workgroup.  Chunk q of the tile = (row q>>3, 16-byte slot q&7); the slot is filled from global chunk
// slot ^ ((row>>1)&7) (source-side swizzle, see header).  In the K loop a load is then  uniform base (SGPR) + this 32-bit
// offset (VGPR): no per-load VALU address arithmetic.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_offsets(uint32_t (&off)[(ROWS * 8 + NT - 1) / NT], int ld, int row0, int max_row, int tid) {
#pragma unroll
    for (int i = 0; i < (ROWS * 8 + NT - 1) / NT; ++i) {
        const int q = i * NT + tid;
        const int row = q >> 3;
        const int c = q & 7;
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row;
        const int gc = c ^ ((row >> 1) & 7);
        off[i] = (uint32_t)(grow * ld + gc * 8) * 2u;
    }
}

template <int ROWS, int NT>
__device__ __forceinline__ void stage_tile(const char* __restrict__ gbase /* uniform, K offset applied */,
                                           const uint32_t (&off)[(ROWS * 8 + NT - 1) / NT], char* lds_wave /* uniform */, int tid) {
#pragma unroll
    for (int i = 0; i < (ROWS * 8 + NT - 1) / NT; ++i) {
        if ((ROWS * 8) % NT != 0 && i * NT + tid >= ROWS * 8) break;  // ragged last pass (768-thread configs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + off[i]),
                                         (__attribute__((address_space(3))) void*)(lds_wave + i * NT * 16), 16, 0, 0);
    }
}


struct HeadNormArgs {
    const float* x; int ldx;  // fp32 [M][ldx]; q cols [0,D), k cols [D,2D), v cols [2D,3D) (as selected)
    int q_col, k_col, v_col;  // starting column of each part, -1 = absent
    const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b;  // [dh]
    const float* rope_cos; const float* rope_sin;  // [max_len][dh/2] or null (no RoPE)
    bf16_t* q; bf16_t* k; bf16_t* v;   // [B][H][Lp][DQK], [B][H][Lp][DQK], [B][H][Lp][DV] (row-major: keys x channels)
    int B, H, L, Lp, dh;
    int perm;   // EPI_QKV: the q / k weight rows are in the RoPE-pair order of EZDIT_T_QKROPE (gemm_pp.h qkrope_col): epilogue in registers, v tiles included
};

struct RowArgs {
    // h_new = (mode SET) sum_s part_s + bias | (RES) h_in + gate * (sum_s part_s + bias) | (COPY) h_in
    const float* h_in; float* h_out;  // h_out nullable (not stored)
    const float* part; int nsplit; long part_stride; int ld_part;  // strides in elements
    int part_bf16;                       // slabs hold bf16 instead of fp32
    const float* bias;
    const float* gate; long gate_slot_stride;  // gate nullable -> 1; per-slot vector when stride != 0
    int mode;  // 0 COPY, 1 RES, 2 SET
    // LN output: u = LN(x) * g + c (per-slot or static vectors), bf16, zero padded to ld_u
    const float* ln_g; const float* ln_c; long ln_slot_stride;
    float cn_scale;                      // multiplies cn (conditioning_scale, controlnet.py:313)
    const float* skip; const float* cn;  // concat mode: x = [h_new | skip (+ cn)], LN over 2D with ln_g/ln_c of length 2D
    bf16_t* u; int ld_u;
    int M, D, L;           // rows, width, rows per batch element
    const int* cur_step; const int* row_slot;
    int wt;                // output stores are write-through (sc1)
    int variant;           // 0: one 256-thread workgroup per row; 1: one wave per row (no LDS, no barriers)
    int affine;            // wave-per-row form: rows [128 p, 128 p + 128) are processed on XCD p % 8 (see k_row_w)
};

struct GemmArgs {
    const bf16_t* A; int lda;   // [M][lda] bf16, row-major, K contiguous, zero padded to K_pad
    const bf16_t* W; int ldw;   // [wrows][ldw] bf16 (nn.Linear layout: out x in), rows >= N zero padded
    int wrows;                  // allocated rows of W (tile loads clamp to wrows - 1)
    const float* bias;          // nullable; EPI_GEGLU: bias in the same interleaved order
    void* out; int ldo;
    long slab_stride;           // EPI_PARTIAL: elements between split-K slabs
    int M, N, K;                // K multiple of 64 (padded); N = valid output columns
    int splitk;                 // >= 1 (EPI_PARTIAL only)
    int epi;
    int tile;                   // tile / pipeline configuration, see gemm.hip
    int debug;                  // bits 8..: timing-ablation variant of the ping-pong K loop (EZ_ABLATE builds, gemm_pp.h); 0 in the product
    // convolution-as-GEMM addressing (VAE decoder): K tile t reads A at byte offset (t / conv_cpb) * conv_tap_bytes +
    // (t % conv_cpb) * 128, i.e. tap t/conv_cpb is the SAME activation rows shifted by a fixed number of rows.
    // conv_cpb = 0: plain GEMM (offset t * 128).
    int conv_cpb; long conv_tap_bytes;
    const float* resid; int ldr;  // EPI_F32: out = resid[row][col] + gate[col] * (acc + bias) (residual connection), nullable
    // gate (nullable = 1): per-column vector of modulation slot (*cur_step + row_slot[row / rows_per_b]), as in RowArgs
    const float* gate; long gate_slot_stride; const int* cur_step; const int* row_slot; int rows_per_b;
    // workgroup -> XCD placement: the (M tiles x N tiles x K splits) grid is cut into pm x pn x pz = 8 boxes, one per XCD
    // (hardware deals workgroups to XCDs round-robin), so each XCD's private 4 MB L2 sees only its box's slice of A and W.
    // Filled by launch_gemm (xcd_map: 0 = legacy 1 x 8 x 1, 1 = smallest per-XCD footprint).
    int xcd_map; int pm, pn, pz, bm, bn, bz;
    unsigned mbm, mbn, msplit, mG;   // ez_magic() of bm, bn, splitk and (N tiles x splitk): filled by the launchers (launch_magic, gemm.hip)
    int wt;                       // output stores are write-through (sc1)
    HeadNormArgs hn;              // EPI_QKV only (x / ldx / *_col unused)
    int part_bf16;                // EPI_PARTIAL: slabs are stored as bf16 (half the bytes written back and re-read by k_row)
    // split-K slabs with all workgroups of an M tile (N tiles x K splits) on ONE XCD (M tile tm -> XCD tm % 8): the slabs and the row kernel
    // that reduces them (row panel p on XCD p % 8, RowArgs.affine) stay inside that XCD's L2
    int xcd_panel;
    // ---- LayerNorm algebra (k_gemm_pp).  LN(x) g + c followed by a GEMM with W equals  r (x g) W^T - r mu (g W^T) + c W^T  row by row, with
    // (mu, r) the row's mean and 1 / sqrt(var + eps): the producer of x stores A' = bf16(x g) and partial statistics, the consumer runs the
    // plain GEMM on A' and applies  acc := r (acc - mu G'[col]) + C'[col]  in its epilogue, G' = g W^T and C' = c W^T (+ bias) precomputed
    // per modulation slot (ezdit_prepare_timesteps).  Exact up to where the bf16 rounding of the operand falls (on x g instead of LN(x) g + c).
    // producer (EPI_RESID):
    bf16_t* zu; int ld_zu;                      // A' [M][ld_zu]
    const float* zg; long zg_slot_stride;       // LayerNorm gain of the consumer (per slot when the stride is non-zero)
    float2* zstat_out;                          // [N tiles][zs_stride]: (sum, sum of squares) of h_new over the columns of each N tile, PART-MAJOR
    long zs_stride;                             // rows (elements) between the parts of zstat_out / zstat_in
    // producer, DUAL form (null zd = off; gate and resid required): cross-attention over ONE valid key is the constant  W_o v_key + b_o  for every query
    // (softmax over one key is 1; the uncond rows of classifier-free guidance: src/inference.py:44-50, attention.py:131-135), so for the rows of such
    // batch elements -- the rows OUTSIDE [act_row0, act_row1) -- the attention-out projection adds that vector (zd [B][zd_stride], per batch element)
    // on top of its own gated residual and emits the operand of the GEGLU GEMM (gain zg2) instead of the cross-attention q projection's
    const float* zd; long zd_stride; const float* zg2; long zg2_slot_stride; int act_row0, act_row1;
    // consumer (EPI_QKV, EPI_GEGLU; null zstat_in = plain GEMM):
    const float2* zstat_in; int zparts; int zD; int zw; // [zparts][zs_stride] partial statistics of the operand's rows: zparts = ceil(zD / zw) <= Z_MAXP parts of zw columns (the last one ragged; zw = the producer's tile width)
    const float* zG; const float* zC; long zt_slot_stride;   // G', C' [slots][N]
    float zeps;
    unsigned long long* ts;   // test hook (k_gemm_pp, k_gemm_ks): [workgroup][8] shader-clock stamps (kernel start, loop start, loop end, kernel end, 4 epilogue marks), nullable
    long ts_cap;              // workgroups `ts` has room for: the launcher drops `ts` for a larger grid
    int epi_lds;   // k_gemm bf16 epilogues (GEGLU output, bf16 slabs): park the tile in the dead ring and write whole rows, 16 bytes per lane
    // ---- round 6, the skip path's forms of the producers (k_gemm_ks, k_gemm_pp EPI_RESID).  At the END of the struct: the offsets of everything above -- and with them how hipcc groups
    // the kernel-argument loads of every GEMM prologue -- stay what they were (in the middle of the struct the same three fields cost the whole step 0.5 %, r06ah)
    // producer, COPY2 form (null zu2 = off): a second operand  bf16(h_new * zg2) -> zu2 [M][ld_zu2]  (zg2 static: no slot) -- the in-blocks' MLP-out
    // projection writes the `skip` half of the matching out-block's [x | skip] operand while it has the values in registers
    bf16_t* zu2; int ld_zu2;
    // consumer + producer, ZIN form (EPI_RESID without gate and residual; null zstat_in2 = off): the operand is A' = bf16([x | skip] * g) with partial statistics
    // in TWO sets of zparts parts (zstat_in: the x half, zstat_in2: the skip half; the row's statistics over zD = 2 D columns are their sums);
    // acc := r (acc - mu G'[col]) + C'[col] with G' = zG and C' = `bias` (static tables), then the producer's epilogue as usual
    const float2* zstat_in2;
};
int launch_gemm(const GemmArgs& a, hipStream_t st);   // 0 = launched, nonzero = configuration not supported (nothing launched)

struct AttnArgs {
    const bf16_t* q;    // [B][H][Lqp][DQK]
    const bf16_t* k;    // [B][H][Lkp][DQK]
    const bf16_t* v;    // [B][H][Lkp][DV] (row-major; the P.V operand is gathered with transposing LDS reads, attn.hip)
    const uint8_t* kmask;  // nullable [B][Lk], 1 = attend
    bf16_t* out; int ldo;  // [B*Lq][ldo], head h occupies cols [h*dh, (h+1)*dh)
    int B, H, Lq, Lk, Lqp, Lkp, dh;
    // optional fused query prologue (cross-attention): q_raw fp32 [B*Lq][ld_qraw] straight from the projection GEMM; the
    // per-head LayerNorm (attention.py:141, shared affine [dh]) is applied while the MFMA operand is built (q unused)
    const float* q_raw; int ld_qraw; const float* qn_w; const float* qn_b;
    int nkh;   // key sub-blocks (waves) per query sub-block: 2 (64-key tiles), 4 (128-key tiles, Lkp % 128 == 0), 0 = auto
    // optional fused query PROJECTION (cross-attention, 8-wave form only): q_raw = xu[rows][K] . xw[h*dh .. +dh][K]^T is computed
    // by the workgroup itself (bf16 MFMA, K split over 4 wave groups, reduced through LDS), then normalised as above.
    // xu bf16 [B*Lq][ldu], xw bf16 [xw_rows][ldw] (nn.Linear layout), xK multiple of 64.  q and q_raw unused.
    const bf16_t* xu; int ldu; const bf16_t* xw; int ldw; int xw_rows; int xK;
    // workgroup -> XCD placement: 1 = linear grid in which ALL query tiles of a (batch, head) pair -- and ceil(B*H/8) whole pairs --
    // run on ONE XCD (hardware deals workgroup i to XCD i % 8), so each pair's K / V^T is fetched into exactly one L2.
    // 0 = (query tile, head, batch) grid: the query tiles of a pair land on 8 different XCDs (8x the K/V traffic).
    int xcd_map; int nq, ppx;   // nq / ppx filled by launch_attention
    unsigned mnq, mH;           // ez_magic(nq), ez_magic(H): filled by launch_attention
    int wt;                     // output stores are write-through (sc1)
    int xk2;                    // fused projection: ring slots of TWO K tiles (one barrier + one counted wait per 128 of K)
    int qtile;                  // fused projection with the LayerNorm algebra: query rows per workgroup: 64, 32, 0 = 32 when the 64-row grid is <= 128 workgroups (attn.hip k_attn QT)
    // fused projection with the LayerNorm algebra (GemmArgs.z*): xu holds A' = bf16(x g); q_raw := r (acc - mu G'[col]) + C'[col] with (mu, r)
    // from the partial statistics of row (b * Lq + query row); G', C' [H * dh] of this block (the LayerNorm in front of to_q is static)
    const float2* zstat_in; long zs_stride; int zparts; int zD; int zw; const float* zG; const float* zC; float zeps;   // zstat_in [zparts][zs_stride], part-major
    unsigned long long* ts;     // test hook: [workgroup][8] shader-clock stamps (start, operands staged, tile loop end, merge end, end), nullable
    long ts_cap;                // workgroups `ts` has room for: the launcher drops `ts` for a larger grid
    // batch sub-range: the launch covers the batch elements [b0, b0 + B) of the buffers above (all pointers are given for batch element 0 and
    // advanced by the launcher).  Cross-attention of a CFG step runs over the conditional rows only: an uncond row's mask has ONE valid key, its
    // output is a constant that the attention-out projection adds (GemmArgs.zd)
    int b0;
};
int launch_attention(const AttnArgs& a, hipStream_t st);   // 0 = launched, nonzero = configuration not supported

void launch_row(const RowArgs& a, hipStream_t st);

void launch_headnorm(const HeadNormArgs& a, hipStream_t st);

struct AssembleArgs {
    const float* x; int x_rows; int in_ch;  // in_ch = C: build [x | gt' | m]; in_ch = 2C+1: x is already assembled
    const float* gt; const uint8_t* gt_mask; const float* mask_embed;
    bf16_t* out; int ldo;  // [B*L][ldo], zero padded
    int B, C, L;
};
void launch_assemble(const AssembleArgs& a, hipStream_t st);

struct FinalConvArgs {
    const float* y; int ldy;  // fp32 [B*L][ldy], token-major, C valid cols
    const float* w; const float* b;  // [C][C][3], [C]
    float* out;               // [B][C][L]
    int B, C, L;
};
void launch_final_conv(const FinalConvArgs& a, hipStream_t st);

// y[n][N] = act(x[n][K] . W[N][K]^T + b) in fp32; act: 0 none, 1 silu.  x_mode 1: x is the sinusoidal
// embedding of timesteps ts[n] (K = 256), computed on the fly.
void launch_linear_f32(const float* x, const int* ts, int x_mode, const float* W, const float* b, float* y,
                       int n, int N, int K, int act, long y_stride, hipStream_t st);

struct ModFinalizeArgs {
    const float* ada;       // [n][6D]  time_ada(tt)
    const float* lora;      // [n][nblk][6D]  lora_b(lora_a(tt)) (unscaled)
    float scaling;          // alpha / r
    const float* table;     // nblk pointers are not contiguous -> passed as base + stride
    long table_stride;      // elements between blocks' scale_shift_table
    const float* n1w; const float* n1b; const float* n3w; const float* n3b; long norm_stride;
    float* mod;             // [n][nblk][6][D]: g1,c1,a1,g3,c3,a3
    const float* ada_final; // [n][2D]
    const float* nfw; const float* nfb;
    float* mod_final;       // [n][2][D]: gF,cF
    int n, nblk, D;
    int has_final;          // 0 for the ControlNet (no FinalBlock)
};
void launch_mod_finalize(const ModFinalizeArgs& a, hipStream_t st);

void launch_rope_table(float* cosT, float* sinT, int max_len, int dh, hipStream_t st);

struct CfgDdimArgs {
    const float* pred;   // [B][C][L]; rows [0,P) cond, [P,2P) uncond (or B = P without CFG)
    float* latents;      // [P][C][L] in/out
    const float* noise;  // [n_steps][P][C][L] or null
    const float* coef;   // [n_steps][8] (sa, sb, c_x0, c_dir, sigma, 0,0,0)
    const int* cur_step; // null: standalone step, coefficients in hc[], noise is this step's slice
    float hc[5];         // (sa, sb, c_x0, c_dir, sigma) when cur_step is null
    float guidance_scale, guidance_rescale;  // guidance_scale <= 0: no CFG
    int P, n;            // n = C*L elements per sample
    // fused sampler: the LAST workgroup to finish (arrival counter `done`, zero between launches) advances the device step
    // counter, so the step needs no separate single-thread launch.  Both null: stand-alone operator.
    int* step_inc; unsigned* done;
};
void launch_cfg_ddim(const CfgDdimArgs& a, float* partial /* [P][64][4] scratch */, hipStream_t st);
void launch_set_int(int* p, int v, int add, hipStream_t st);  // *p = add ? *p + v : v
struct Conv1dArgs {  // out[b][co][lo] = act(bias[co] + sum_{ci,k} w[co][ci][k] * x[b][ci][lo*stride + k - pad]); fp32
    const float* x; const float* w; const float* b; float* out;
    int B, Cin, Cout, Lin, Lout, ksize, stride, pad, act;  // act 1 = SiLU
    int cin_valid;      // channels >= cin_valid of x are implicit zeros (the eval-time mask channel)
    int out_token_major;  // 1: out[b][lo][co] (token-major rows, ld = Cout)
};
void launch_conv1d(const Conv1dArgs& a, hipStream_t st);
void launch_cast_bf16(const float* x, int ldx, bf16_t* out, int ldo, int M, int N, int act, hipStream_t st);  // act 1 = silu
// y[yoff[i] + n] = bias[n] + sum_k x[xoff[i] + k] W[n][k] for i < n vectors in ONE launch (x fp32, rounded to bf16 first when x_bf16 -- what an activation
// that went through a bf16 buffer would be; W bf16 [N][ldw]; once per call)
constexpr int GEMV_MAXB = 32;
struct GemvBatch { const float* x; float* y; int n; long xoff[GEMV_MAXB]; long yoff[GEMV_MAXB]; };
void launch_gemv_bf16w(const GemvBatch& g, int x_bf16, const bf16_t* W, int ldw, const float* bias, int N, int K, hipStream_t st);
// LayerNorm algebra tables (rowops.hip): (g, c) [n_slots][D] (stride slot_stride) -> bf16 rows (g hi, g lo, c hi, c lo) per slot, zero padded to ldo;
// and back: zG[s][n] = tmp[4s][n] + tmp[4s+1][n], zC[s][n] = tmp[4s+2][n] + tmp[4s+3][n] (+ bias[n])
void launch_z_hilo(const float* g, const float* c, long slot_stride, bf16_t* out, int ldo, int n_slots, int D, hipStream_t st);
void launch_z_combine(const float* tmp, int ld_tmp, const float* bias, float* zG, float* zC, long slot_stride, int n_slots, int N, hipStream_t st);
