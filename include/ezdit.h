/*
 * ezdit.h -- C ABI of libezaudio_hip.so: the MI355X (gfx950) EzAudio denoising path.
 *
 * The reference (haidog-yaqub/EzAudio) is pure Python/PyTorch and has no FFI layer; its hot path
 * sits behind plain Python callables.  This header is therefore the boundary a maintainer would
 * bind (ctypes stub in INTEGRATION.md); each entry point names the reference interface it stands
 * in for (paths relative to the reference tree).
 *
 * Conventions (surface B3):
 *   - every pointer marked `dev` is a DEVICE pointer owned by the caller; the library never
 *     allocates or frees persistent device memory.  Weights live in one caller-owned blob laid out
 *     by ezdit_param_info(); all activations / tables live in one caller-owned workspace.
 *   - the PER-STEP entry points (ezdit_forward, ezdit_controlnet_forward, ezdit_sampler_run, ezdit_set_step,
 *     ezdit_cfg_ddim_step, the ezvae_* ops) are ASYNCHRONOUS on their stream: no device sync, no host read of
 *     device data (a sampler step is hipGraph-capturable).  The once-per-call SET-UP entry points synchronise the
 *     stream and must not be called inside a stream capture: ezdit_bind_workspace (diagnostic builds only),
 *     ezdit_prepare_context (reads the context mask back to find single-key batch elements when `xkey1` is on),
 *     ezdit_prepare_timesteps and ezdit_sampler_begin (host staging buffers of the timestep / coefficient tables).
 *   - return 0 = OK, negative = error; ezdit_last_error() gives a thread-local message.  No C++
 *     exception crosses the ABI.
 *   - a handle is bound to the device that was current at ezdit_create() and is NOT thread-safe:
 *     one handle per device/stream; multi-GPU = one process + one handle per GPU.
 *   - layouts: latents channel-major fp32 [rows, C, L] exactly as the reference passes them
 *     (src/inference.py:67,75); context fp32 [B, Lc, Cctx]; masks uint8 (0/1).
 */
#ifndef EZDIT_H
#define EZDIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EZDIT_ABI_VERSION 4   /* 4 (round 6): ezdit_test_attention takes V row-major [B][H][Lkp][DV] (was V^T [B][H][DV][Lkp]); the internal buffers "vt" / "vct" are "v" / "vc" */

typedef struct ezdit_handle ezdit_handle;
typedef void* ezdit_stream; /* hipStream_t */

/* Mirrors the `model:` section of ckpts/ezaudio-{l,xl}.yml (the keys UDiT.__init__ consumes,
 * src/models/udit.py:11-30).  Only the shipped combination is implemented; ezdit_create returns
 * EZDIT_E_UNSUPPORTED for anything else, mirroring the NotImplementedError sites udit.py:83,113,127. */
typedef struct {
    int32_t embed_dim;     /* D */
    int32_t num_heads;     /* H, head_dim = D / H, must be 64 or 72 */
    int32_t depth;         /* depth//2 in-blocks + mid + depth//2 out-blocks */
    int32_t in_chans;      /* 257 = 2*out_chans + 1 (MaskDiT concat, conditioners.py:161-176) */
    int32_t out_chans;     /* C = 128 latent channels */
    int32_t context_dim;   /* T5 width */
    int32_t ada_sola_rank; /* r */
    float   ada_sola_alpha;
    float   mlp_ratio;     /* 4.0 */
    int32_t max_len;       /* longest latent sequence the RoPE table covers */
    /* ControlNet variant (src/models/controlnet.py:87-315): the first depth/2 blocks of the backbone + condition embed +
     * one zero-initialised Linear per skip; no mid/out blocks, no final block.  0 = plain UDiT. */
    int32_t controlnet;
    int32_t cond_in;       /* channels of the control signal (1 for energy) */
    int32_t cond_c0;       /* cond_blocks[0] = 64 (+1 mask channel when cond_mask) */
    int32_t cond_c1;       /* cond_blocks[1] = 128 */
    int32_t cond_mask;     /* DiTControlNetEmbed cond_mask: append the (all-zero at inference) mask channel */
} ezdit_config;

enum {
    EZDIT_OK = 0,
    EZDIT_E_INVALID = -1,      /* bad argument / shape (reference: AssertionError class) */
    EZDIT_E_UNSUPPORTED = -2,  /* config value outside the implemented set (NotImplementedError) */
    EZDIT_E_STATE = -3,        /* call order: weights/workspace/context/timesteps not prepared */
    EZDIT_E_HIP = -4           /* a HIP runtime call failed */
};

/* ---- parameter blob layout (replaces MaskDiT.load_state_dict, api/ezaudio.py:83-85) ---------- */
enum { EZDIT_P_F32 = 0, EZDIT_P_BF16 = 1 };
enum {
    EZDIT_T_NONE = 0,
    EZDIT_T_GEGLU8 = 1, /* rows re-ordered so each 16-row group = 8 value rows then their 8 gate rows */
    /* fused to_q | to_k | to_v weight [3D][D] of an even head count (round 6): within the q rows [0, D) and the k rows [D, 2D), the rows of every PAIR of
     * heads (2 dh rows) are re-ordered so that stored row c = 16 j + 4 g + 2 e + s of the pair holds channel f + (dh / 2) s of head hh, with the RoPE pair
     * index f = 8 (j mod dh/16) + 2 g + e for the dh/16 full 16-row groups of a head (hh = 0 for the first dh/16 groups, 1 for the last), and -- head_dim 72:
     * 4.5 groups per head -- the middle group split by g: g < 2 -> head 0, g >= 2 -> head 1, f = 32 + 2 (g & 1) + e (csrc/gemm_pp.h qkrope_col).  A
     * channel and its rotate-half partner then sit side by side, two pairs per lane of the 16x16 MFMA output: per-head LayerNorm + RoPE run on the
     * accumulators.  The v rows [2D, 3D) keep their order.  q and k are stored in this channel order; q . k^T is invariant under it. */
    EZDIT_T_QKROPE = 2
};
typedef struct {
    char    name[64];      /* our slot name, e.g. "blk3.wqkv" */
    char    src[3][96];    /* state-dict keys concatenated along dim 0 (nsrc of them) */
    int32_t nsrc;
    int32_t dtype;         /* EZDIT_P_* */
    int32_t transform;     /* EZDIT_T_* */
    int64_t rows, cols;    /* logical shape after concatenation (trailing dims flattened) */
    int64_t rows_pad, ld;  /* stored shape: zero padded to [rows_pad][ld] */
    int64_t offset;        /* byte offset in the blob, 256-byte aligned */
} ezdit_param_info_t;

int         ezdit_abi_version(void);
const char* ezdit_last_error(void);

int    ezdit_create(const ezdit_config* cfg, ezdit_handle** out);
int    ezdit_destroy(ezdit_handle* h);

int    ezdit_param_count(const ezdit_handle* h);
int    ezdit_param_info(const ezdit_handle* h, int index, ezdit_param_info_t* out);
size_t ezdit_param_bytes(const ezdit_handle* h);
int    ezdit_bind_weights(ezdit_handle* h, const void* dev_blob, size_t bytes);

/* ---- workspace ------------------------------------------------------------------------------- */
/* B = denoiser batch rows (2 x prompts with CFG), L = latent frames, Lc = context tokens,
 * n_slots = number of distinct timesteps whose modulation tables are resident (sampler: n_steps). */
size_t ezdit_workspace_bytes(const ezdit_handle* h, int B, int L, int Lc, int n_slots);
int    ezdit_bind_workspace(ezdit_handle* h, void* dev_ws, size_t bytes, int B, int L, int Lc, int n_slots,
                            ezdit_stream stream);

/* ---- step-invariant work, hoisted (reference recomputes it every step) ----------------------- */
/* context_embed + per-block norm_context + cross to_k/to_v + head LayerNorm(k):
 * src/models/udit.py:94-97,295-296; blocks.py:84-85,150; utils/attention.py:128-142.
 * ctx fp32 [B,Lc,Cctx]; mask uint8 [B,Lc], 1 = attend (T5 attention_mask, src/inference.py:42,47). */
int ezdit_prepare_context(ezdit_handle* h, const float* dev_ctx, const uint8_t* dev_mask, ezdit_stream stream);

/* timestep embedding, TimestepEmbedder, time_act, time_ada(_final), per-block AdaLN-SOLA:
 * utils/modules.py:19-61; udit.py:305-316; blocks.py:39-45,132-133.  `timesteps` is a HOST array.
 * per_row = 0: every batch row uses slot `cur_step` (see ezdit_set_step); sampler usage.
 * per_row = 1: n must equal B and row b uses slot b (MaskDiT.forward with a [B] timesteps tensor). */
int ezdit_prepare_timesteps(ezdit_handle* h, const int32_t* timesteps, int n, int per_row, ezdit_stream stream);
int ezdit_set_step(ezdit_handle* h, int step, ezdit_stream stream);

/* ---- the denoiser operator: MaskDiT.forward / UDiT.forward ----------------------------------- */
/* src/models/conditioners.py:156-183 (in_ch = C: x [x_rows,C,L], optional gt/gt_mask [x_rows,C,L];
 * batch row b reads latent row b % x_rows, which is how the CFG pair shares one latent,
 * src/inference.py:75) or src/models/udit.py:281-362 directly (in_ch = 2C+1: x is the assembled
 * [B,257,L] input, as src/inference_controlnet.py:97-99 calls unet.model).
 * cn_skips: optional n_cn device pointers to fp32 [B,L,D] ControlNet residuals in the reference's
 * list order (popped from the end, udit.py:345-348), each already multiplied by conditioning_scale.
 * out: fp32 [B,C,L]. */
int ezdit_forward(ezdit_handle* h, const float* dev_x, int in_ch, int x_rows,
                  const float* dev_gt, const uint8_t* dev_gt_mask,
                  const float* const* cn_skips, int n_cn,
                  float* dev_out, ezdit_stream stream);

/* ---- ControlNet (handle created with cfg.controlnet = 1) ------------------------------------------------ */
/* DiTControlNetEmbed (controlnet.py:65-84) on the control signal, hoisted out of the step loop (it depends on the
 * condition only): cond fp32 [B, cond_in, Lcond] with Lcond = 2 L (the embed has one stride-2 conv). */
int ezdit_prepare_condition(ezdit_handle* cn, const float* dev_cond, int Lcond, ezdit_stream stream);
/* DiTControlNet.forward (controlnet.py:252-315): x as in ezdit_forward (in_ch = C needs `dev_mask_embed`, the [C] fp32
 * mask_embed of the MaskDiT that assembles the input, conditioners.py:161-176); the depth/2 residuals
 * zero_linear_i(skip_i) land in the ControlNet workspace.  They are NOT yet multiplied by conditioning_scale. */
int ezdit_controlnet_forward(ezdit_handle* cn, const float* dev_x, int in_ch, int x_rows, const float* dev_gt,
                             const uint8_t* dev_gt_mask, const float* dev_mask_embed, ezdit_stream stream);
/* pointers to the residuals of the last ezdit_controlnet_forward, in the reference's list order (out[i] <-> skip i) */
int ezdit_controlnet_residuals(ezdit_handle* cn, const float** out, int n);
/* the backbone's sampler then runs ControlNet + backbone per step (src/inference_controlnet.py:89-99) */
int ezdit_sampler_attach_controlnet(ezdit_handle* h, ezdit_handle* cn, float conditioning_scale);
/* scale applied to cn_skips inside ezdit_forward (default 1.0: residuals already scaled by the caller, as DiTControlNet.forward
 * returns them).  Independent of the conditioning_scale of a ControlNet attached to the fused sampler. */
int ezdit_set_cn_scale(ezdit_handle* h, float scale);

/* ---- sampler: CFG + rescale + DDIM, src/inference.py:70-100 + diffusers DDIMScheduler.step ---- */
typedef struct {
    float sa, sb;      /* sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t) */
    float c_x0, c_dir; /* sqrt(alpha_bar_prev), sqrt(1 - alpha_bar_prev - sigma^2) */
    float sigma;       /* eta * sqrt(variance) */
} ezdit_ddim_coef;

/* ONE stand-alone CFG + guidance-rescale + DDIM update (src/inference.py:88-100 `rescale_noise_cfg` + `scheduler.step`), for
 * callers that keep their own Python loop: dev_pred fp32 [2P][n] (rows [0,P) conditional, [P,2P) unconditional; [P][n] when
 * guidance_scale <= 0), dev_latents fp32 [P][n] updated in place, dev_noise fp32 [P][n] for THIS step or NULL, coef by value
 * (host), n = C*L elements per sample, dev_scratch >= P*256 floats (needed when guidance_rescale > 0).  No handle needed. */
int ezdit_cfg_ddim_step(const float* dev_pred, float* dev_latents, const float* dev_noise, const ezdit_ddim_coef* coef,
                        float guidance_scale, float guidance_rescale, int P, int n, float* dev_scratch, ezdit_stream stream);

/* dev_latents fp32 [P,C,L] is updated in place each step; dev_noise fp32 [n_steps,P,C,L] or NULL
 * (eta == 0); coefs is a HOST array of n_steps entries (copied into the workspace);
 * guidance_scale <= 0 disables CFG (B = P, src/inference.py:94-96), otherwise B = 2P with rows
 * [0,P) conditional and [P,2P) unconditional.  gt/gt_mask as in ezdit_forward (editing). */
int ezdit_sampler_begin(ezdit_handle* h, float* dev_latents, int P, const float* dev_noise,
                        const ezdit_ddim_coef* coefs, int n_steps,
                        float guidance_scale, float guidance_rescale,
                        const float* dev_gt, const uint8_t* dev_gt_mask, ezdit_stream stream);
/* run `n` consecutive steps from the current step counter; use_graph != 0 captures one step into a
 * hipGraph on first use and replays it (no host work between kernels).  Running past the prepared steps (n_steps of
 * ezdit_sampler_begin / n of ezdit_prepare_timesteps) is refused with EZDIT_E_STATE; ezdit_set_step rewinds. */
int ezdit_sampler_run(ezdit_handle* h, int n, int use_graph, ezdit_stream stream);

/* ---- Oobleck VAE decoder building blocks (src/modules/stable_vae/models/autoencoders.py:38-61,82-113,149-190) --------
 * Stateless ops on caller-owned device buffers; the layer sequence is host code (ezaudio_amd/vae.py), run once per call.
 * Activations are token-major [L][C] with zero halo rows so that convolutions are GEMMs over shifted rows. */
/* out fp32 [M][ldo] = A[M][K] . W[N][K]^T (+ bias[N]) (+ resid[M][ldr]); K tile t (64 wide) of A is read at byte offset
 * (t / conv_cpb) * conv_tap_bytes + (t % conv_cpb) * 128 (conv_cpb = 0: plain GEMM).  A, W bf16. */
int ezvae_gemm(const void* dev_a, int lda, const void* dev_w, int ldw, int wrows, const float* dev_bias, const float* dev_resid,
               int ldr, float* dev_out, int ldo, int M, int N, int K, int conv_cpb, long conv_tap_bytes, int tile,
               ezdit_stream stream);
/* SnakeBeta (models/blocks.py:317-358) fused with the fp32 -> bf16 cast: out = x + inv_beta * sin(alpha x)^2; alpha NULL = cast only */
int ezvae_snake_bf16(const float* dev_x, int ldx, const float* dev_alpha, const float* dev_inv_beta, void* dev_out, int ldo,
                     long L, int C, ezdit_stream stream);
/* final WNConv1d(C -> 1, k 7, pad 3, no bias): xb bf16 haloed (row 0 = position -3), w fp32 [7][C] -> out fp32 [L] */
int ezvae_conv_out1(const void* dev_xb, int ldx, const float* dev_w, float* dev_out, long L, int C, ezdit_stream stream);

/* encoder input WNConv1d(1 -> C, k 7, pad 3) (autoencoders.py:130-132): wav fp32 [T], w fp32 [7][C], bias [C] -> out fp32 [T][C] */
int ezvae_conv_in1(const float* dev_wav, const float* dev_w, const float* dev_bias, float* dev_out, long T, int C, ezdit_stream stream);
/* VAEBottleneck.encode (models/bottleneck.py:67-71,77-87): enc fp32 [L][2*latent] token-major (mean | scale), noise fp32
 * [latent][L] (caller's randn; NULL = return the mean) -> z fp32 [latent][L] = noise * (softplus(scale) + 1e-4) + mean */
int ezvae_sample(const float* dev_enc, const float* dev_noise, float* dev_z, int L, int latent_dim, ezdit_stream stream);

/* ---- unit-test hooks: one kernel family each, same code the forward uses ---------------------- */
int ezdit_test_gemm(ezdit_handle* h, int variant, const void* dev_a_bf16, int lda, const void* dev_w_bf16, int ldw,
                    const float* dev_bias, void* dev_out, int ldo, int M, int N, int K, int splitk,
                    ezdit_stream stream);
/* unit-test hook of the un-split residual projection (LayerNorm algebra, producer side; csrc/gemm_ks.h): h_out = h_in + gate * (A . W^T + bias)
 * (fp32 [M][N]; gate NULL = 1, h_in NULL = 0), zu = bf16(h_out * zg) ([M][ld_zu]) and zstat ([ceil(N / cw)][M] float pairs, part-major: sum and
 * sum of squares of each cw-column tile, cw = the kernel's tile width: 96 for tile 70, see csrc/gemm.hip). */
int ezdit_test_resid(int tile, const void* dev_a_bf16, int lda, const void* dev_w_bf16, int ldw, const float* dev_bias, const float* dev_h_in,
                     const float* dev_gate, const float* dev_zg, float* dev_h_out, void* dev_zu_bf16, int ld_zu, void* dev_zstat,
                     int M, int N, int K, ezdit_stream stream);
/* ... and of its two skip-path forms (csrc/common.h GemmArgs COPY2 / ZIN; DESIGN.md "The skip path"): dev_zu2 != NULL -- additionally zu2 = bf16(h_out * zg2) ([M][ld_zu2]; gate and
 * h_in required); dev_zstat_in2 != NULL -- the launch first finishes a LayerNorm over zD columns whose partial statistics come in two part-major sets of zparts parts
 * (dev_zstat_in, dev_zstat_in2: [zparts][M] float pairs): acc := r (acc - mu zG[col]) + bias[col], no gate, no h_in.  dev_h_out NULL = the fp32 stream is not stored. */
int ezdit_test_resid_skip(int tile, const void* dev_a_bf16, int lda, const void* dev_w_bf16, int ldw, const float* dev_bias, const float* dev_h_in,
                          const float* dev_gate, const float* dev_zg, float* dev_h_out, void* dev_zu_bf16, int ld_zu, void* dev_zstat,
                          int M, int N, int K, const float* dev_zg2, void* dev_zu2_bf16, int ld_zu2, const void* dev_zstat_in, const void* dev_zstat_in2,
                          int zparts, int zD, const float* dev_zG, ezdit_stream stream);
/* test hook: launches of k_gemm_pp / k_gemm_ks / k_attn record, per workgroup, eight 64-bit shader-clock stamps (kernel start, K-loop
 * start, K-loop end, kernel end, then epilogue internals) into dev_buf ([capacity_workgroups][8] uint64; NULL switches it off).  A launch
 * whose grid exceeds capacity_workgroups writes no stamps.  Un-register (NULL) before freeing the buffer. */
int ezdit_debug_gemm_timestamps(void* dev_buf, long capacity_workgroups);
/* q, k bf16 [B][H][L*p][DQK], v bf16 [B][H][Lkp][DV] (DQK / DV = 64 / 64 or 80 / 96 for head_dim 64 / 72; padding zero), out bf16 [B*Lq][ldD] */
int ezdit_test_attention(ezdit_handle* h, const void* dev_q, const void* dev_k, const void* dev_v,
                         const uint8_t* dev_kmask, void* dev_out, int B, int Lq, int Lk, int Lqp, int Lkp,
                         ezdit_stream stream);
/* copy an internal fp32/bf16 buffer (by name, e.g. "h", "u", "q", "k", "v", "mod") for debugging. */
int ezdit_debug_buffer(ezdit_handle* h, const char* name, void** dev_ptr, size_t* bytes);
/* number of kernel launches issued by the last ezdit_forward (host counter). */
int ezdit_last_launch_count(const ezdit_handle* h);
/* n > 0: ezdit_forward returns after n kernel launches so a test can inspect intermediates; 0 = off. */
int ezdit_debug_stop_after(ezdit_handle* h, int n_launches);
/* A/B knobs (a captured graph is dropped and re-captured; set them BEFORE ezdit_prepare_timesteps when they select the LayerNorm-algebra path).
 * Defaults are the measured best on MI355X; none changes results beyond fp rounding.  Unknown names return EZDIT_E_INVALID.  Every name below is set
 * by a test (tests/test_host.py: all of them on a handle; tests/test_gpu.py, tests/test_controlnet.py: each non-default value against the reference
 * goldens or bitwise against the default).  Round 5 removed 30 names that only the experiments which settled them ever set (tile / split-K ids per
 * shape, zbig*, ztile, zmlp, zskip, pp_max_m, prefetch, gemm_debug, fuse_resid, fuse_qkv, fuse_qnorm, qkv_waves9, xcd_map, slab_bf16, geglu_tile ...).
 *   zfuse 0/1, default 1 (LayerNorm algebra: attention-out / cross-attention-out / skip_linear / in-block MLP-out projections UN-SPLIT with the
 *     residual, per-column-tile LayerNorm statistics and the next GEMM's operand in their epilogue -- k_gemm_ks (csrc/gemm_ks.h) up to 2048 rows,
 *     the ping-pong kernel's 128 x 144 tile above -- the consumer GEMM finishing the LayerNorm in its epilogue: no split-K slabs, no row
 *     kernel on those edges; needs gemm_pp = 3 and a LayerNorm-algebra q projection (fuse_q2 at small grids, q2_pp above); 0 = split-K slabs + row kernel)
 *   xkey1 0/1, default 1 (single-key cross-attention shortcut, needs zfuse: a batch element whose context mask has ONE valid key -- every unconditional
 *     row of classifier-free guidance -- gets the constant W_o v_key + b_o from the attention-out projection instead of a cross-attention launch;
 *     cross-attention and its out-projection then run over the other batch elements only.  Exact (softmax over one key is 1); 0 = every row through k_attn)
 *   skip_z 0/1, default 1 (needs zfuse; M <= 2048 token rows, no ControlNet residuals: the out-blocks' LayerNorm over [x | skip] in front of skip_linear by the same algebra --
 *     the in-block that produces a skip keeps its statistics and writes its half of the out-block's operand, the MLP-out projection in front of the out-block runs un-split,
 *     skip_linear finishes the LayerNorm in its epilogue: one launch less per out-block; 0 = split-K slabs + the row kernel on that edge)
 *   geglu_co / qkv_co 0/1/2 (GEGLU GEMM / fused QKV GEMM on the co-resident kernel k_gemm_co (csrc/gemm_co.h): 4-wave workgroups, 128 x 144 tiles, TWO per CU, so that one
 *     workgroup's prologue / epilogue runs under the other's K loop; 1 = above 2048 token rows (batched prompts), 2 = always, 0 = the ping-pong kernel's 128 x 288 tile.
 *     Defaults: geglu_co 0, qkv_co 1 -- four prompts per GPU -1.5 % per step, one prompt untouched)
 *   gemm_pp (ping-pong kernel k_gemm_pp: bit 0 GEGLU GEMM; 0 = the round-1 lockstep kernel for it and no LayerNorm algebra.  Bit 1 -- the fused QKV GEMM -- is
 *     retired: since round 6 that GEMM always runs on the ping-pong kernel, its weights are packed for it, EZDIT_T_QKROPE)
 *   tile_partial (tile id of the split-K residual GEMMs at M <= 2048 rows: 9 = lockstep 128 x 128, 62 = the same tile on the ping-pong kernel; csrc/gemm.hip table)
 *   wt 0/1/2 (write-through (sc1) output stores; 2 = default: on while B L <= 2048)
 *   fuse_q2 0/1/2 (cross-attention computes its own q projection; 2 = also for large grids), q2_pp 0/1 (cross-attention q projection at grids too large
 *     for fuse_q2: ping-pong GEMM with the per-head LayerNorm in its epilogue; 0 = fp32 GEMM + normalisation inside k_attn)
 *   attn_nkh 0/2/4 (attention key sub-blocks per tile, 0 = by grid size), attn_xk2 0/1 (cross-attention q projection: two K tiles per ring slot and barrier)
 *   attn_qtile 0/32/64 (fused cross-attention: query rows per workgroup; 0 = 32 when the 64-row grid has <= 128 workgroups -- one prompt with the single-key shortcut)
 *   attn_xcd 0/1 (attention: all query tiles of a (batch, head) pair on one XCD), gemm_panel (bit mask over 1 D x D projections, 2 skip_linear, 4 MLP-out; M <= 1024: the split-K GEMM puts all workgroups of
 *     an M tile on XCD tm % 8) and row_affine 0/1 (the row kernel processes row panel p on XCD p % 8).  Placement only: bitwise identical results.
 *   row_variant 0/1 (row kernel: one workgroup / one wave per row), epi_lds 0/1 (bf16 GEMM epilogues staged through LDS and written as 16-byte row chunks)
 *   cn_overlap 0/1 (fused sampler: ControlNet branch on a side stream next to the backbone's in-blocks)
 *   stamp_launch i / trace_launches 0/1 (diagnostics, eager launches only: launch i of a forward writes its in-kernel cycle stamps to the buffer
 *     registered with ezdit_debug_gemm_timestamps; every launch of a forward is named on stderr -- tools/diag_stamps.py)
 * A library built with -DEZ_DIAG (EZAUDIO_DIAG=1 python -m ezaudio_amd.build) also knows zfake 0/1: the LayerNorm-algebra consumers run on a finished
 * LayerNorm with neutral tables (what the consumer side costs by itself; results change by the factor rsqrt(1 + 1e-5)). */
int ezdit_set_option(ezdit_handle* h, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* EZDIT_H */
