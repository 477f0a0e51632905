// C ABI of libezaudio_hip.so (include/ezdit.h): parameter layout, workspace carving, and the host-side
// sequencing of the denoising step.  The sequencing restates UDiT.forward (src/models/udit.py:281-362),
// DiTBlock._forward (src/models/blocks.py:120-160) and the sampler loop body (src/inference.py:70-100)
// as a fixed chain of asynchronous kernel launches on one stream (hipGraph-capturable: no allocation, no
// sync, no host read-back anywhere below ezdit_forward / ezdit_sampler_run).
#include "../../include/ezdit.h"
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static thread_local std::string g_err;

// sets the thread-local message behind ezdit_last_error() and returns `code` (shared with vae.hip through common.h)
int ez_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define fail ez_fail

namespace {

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(EZDIT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline long rup(long x, long a) { return (x + a - 1) / a * a; }

struct Buf {  // one carved workspace region
    size_t off = 0, bytes = 0;
};

}  // namespace

struct WRef {   // one bf16 weight matrix of the blob, resolved once at ezdit_bind_weights
    const bf16_t* W = nullptr; int ld = 0; int rows = 0;
};
struct BlkW {   // per-block parameters used on the per-step path (no string / map work inside ezdit_forward)
    const float *bskip, *aqnw, *aqnb, *aknw, *aknb, *bo, *n2w, *n2b, *cqnw, *cqnb, *bo2, *b1, *b2, *snw, *snb, *zb;
    WRef wskip, wqkv, wo, wq2, wo2, w1, w2, zw;
};
struct WsPtrs {  // workspace regions used on the per-step path, resolved once at ezdit_bind_workspace
    int* ints; float *rope_cos, *rope_sin, *coef, *cfgpart;
    bf16_t* ape; float *h, *skips; bf16_t *u, *ucat; float* qkv; bf16_t *q, *k, *v, *ao, *act; float *part, *y, *pred;
    uint8_t* kmask; bf16_t *kc, *vc; float *mod, *modf;
    float *cembed, *cnres; bf16_t* skipbf;   // ControlNet only
    float2* zstat; float *zt_qkv, *zt_geglu, *zt_q2;   // LayerNorm algebra: partial row statistics, G' / C' tables
    float2* zstat_skip; float* zt_skip; bf16_t* ucat_z;              // ... of the out-blocks' LN_2D([x | skip]) -> skip_linear: the skips' statistics (kept from the in-block to its out-block), static tables
    float* zd;   // [nblk][B][D] constant cross-attention-out vectors of the single-key batch elements (opt_xkey1)
};

static unsigned long long* g_gemm_ts = nullptr;   // ezdit_debug_gemm_timestamps: device buffer for in-kernel cycle stamps (gemm_pp.h, gemm_ks.h, attn.hip)
static long g_gemm_ts_cap = 0;                    // ... and the workgroups it has room for ([workgroup][8] uint64): a launch with a larger grid gets no stamps

struct ezdit_handle {
    ezdit_config cfg;
    int D, H, dh, nblk, nhalf, I, C, Cin, Cctx, r6;
    int DQK, DV;       // padded head dims used by the attention layouts
    int ldD, ld2D, ldI, ldPE, ldCtx;  // bf16 leading dimensions (multiples of 64)
    float scaling;

    std::vector<ezdit_param_info_t> params;
    std::map<std::string, int> pidx;
    size_t param_bytes = 0;
    const char* wblob = nullptr;

    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int B = 0, L = 0, Lc = 0, n_slots = 0, M = 0, Mp = 0, Lp = 0, Lcp = 0, Mc = 0;
    std::map<std::string, Buf> bufs;
    bool ctx_ready = false, ts_ready = false, cond_ready = false;
    bool z_tables_ready = false;   // the LayerNorm-algebra tables of the prepared timesteps exist (opt_zfuse was on at ezdit_prepare_timesteps)
    bool skip_tables_ready = false;   // ... and the static tables of the skip path (opt_skip_z was on as well): a forward never reads tables that were not built
    int n_ts = 0, per_row = 0;

    // sampler
    float* latents = nullptr;
    const float* noise = nullptr;
    const float* s_gt = nullptr;
    const uint8_t* s_gt_mask = nullptr;
    int P = 0, n_steps = 0;
    float gscale = 0.f, grescale = 0.f;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;

    int launches = 0;
    bool is_cn = false;          // ControlNet variant (cfg.controlnet)
    int c0 = 0, c0m = 0, c1 = 0; // condition-embed channel counts
    ezdit_handle* cn = nullptr;  // ControlNet attached to this backbone's sampler
    float cn_scale = 1.0f;       // conditioning_scale of the ATTACHED ControlNet (fused sampler only)
    float fwd_cn_scale = 1.0f;   // scale ezdit_forward applies to caller-provided residuals (ezdit_set_cn_scale; default 1)
    std::vector<ezdit_handle*> cn_users;  // backbones whose sampler has this ControlNet attached (cleared on destroy)
    const float* ext_mask_embed = nullptr;
    // ---- fixed tile / split-K choices (measured on MI355X in rounds 1-4; the per-shape override options they used to be
    // were deleted in round 5: nothing but the experiments that settled them ever set a non-default value) ----
    // M <= 2048 rows: 128 x 128 8-wave tiles with a 3-deep ring and split-K 3 (216 workgroups, 3 bf16 slabs) for the residual GEMMs that still run
    // split-K, 128 x 64 8-wave ring 4 for the small fp32 ones; patch embed and final Linear on the K-split kernel (k_gemm_ks: 252 / 42 workgroups)
    static constexpr int kSplitK = 3, kTileF32 = 25, kTilePE = 70, kTileFin = 73;
    // M > 2048 rows (batched prompts): the split-K residual GEMMs that remain on the ping-pong kernel's 128 x 288 tile with split-K 2 (32 x 4 x 2 =
    // 256 workgroups at M = 4000, no N padding); the un-split LayerNorm-algebra producer is the ping-pong kernel's 128 x 144 tile above kZBigM rows
    // (one round of 256 workgroups at M = 4000) and k_gemm_ks's 48 x 96 tile (id 70) up to there
    static constexpr int kTilePartialBig = 60, kSplitBig = 2, kZTile = 70, kZBigM = 2048;
    int opt_tile_partial = 9;   // split-K residual GEMMs at M <= 2048: 9 = lockstep 128 x 128 (k_gemm), 62 = the same tile on the ping-pong kernel
    int opt_gemm_pp = 3;   // ping-pong kernel (k_gemm_pp): bit 0 GEGLU GEMM (128x288; 0 = the round-1 lockstep kernel and no LayerNorm algebra); bit 1 retired in round 6 (the fused QKV GEMM always runs on it)
    // LayerNorm algebra (common.h, GemmArgs.z*): the attention-out, cross-attention-out, skip_linear and (in front of an in / mid block) MLP-out
    // projections run UN-SPLIT (k_gemm_ks, gemm_ks.h: 48 x 96 tiles, 252 workgroups at M = 1000; the ping-pong kernel's 128 x 144 tile above 2048 rows) with
    // the gated residual, per-column-tile LayerNorm statistics and the next GEMM's operand bf16(h g) in their epilogue; the consumer (fused QKV
    // GEMM, cross-attention q projection, GEGLU GEMM) finishes the LayerNorm in ITS epilogue as r (acc - mu G') + C': no split-K slabs and 86
    // of the 102 row-kernel launches of an XL step less (239 launches instead of 325), the same algebra as the reference (goldens pass at the
    // same gates).  Needs gemm_pp bits 0 and 1 and a LayerNorm-algebra q projection; otherwise the step falls back to split-K slabs + the row kernel.
    int opt_zfuse = 1;
    // ... and the out-blocks' LN_2D([x | skip]) -> skip_linear (blocks.py:124-128) the same way (round 6, k_gemm_ks forms COPY2 / ZIN, gemm_ks.h): the statistics of the concatenation
    // are the sums of the halves'; the in-block that produces `skip` keeps them and writes bf16(skip g[D:]) into the right half of ITS out-block's operand (one operand buffer per
    // out-block), the MLP-out projection in front of the out-block runs un-split and writes bf16(x g[:D]) + statistics, skip_linear finishes the LayerNorm in its epilogue:
    // 14 split-K GEMMs + 14 row-kernel launches of an XL step become 14 un-split launches (225 launches instead of 239).  The ping-pong producer (M > kZBigM rows) has the same forms.
    // ControlNet residuals change the skips (skip + scale * residual) and take the row-kernel path.
    int opt_skip_z = 1;
    bool skip_z_usable() const { return opt_skip_z && !is_cn && nhalf > 0 && (D + zwidth() - 1) / zwidth() <= (ztile() == kZTile ? 16 : 8); }   // (statistics parts a lane group of the consumer covers)
    // GEGLU GEMM on the co-resident kernel (k_gemm_co, gemm_co.h: 4-wave workgroups with a 128 x 144 tile, TWO per CU, so that a workgroup's prologue and
    // epilogue run under its neighbour's K loop): 0 = never, 1 = above kCoM rows (batched prompts: the ping-pong kernel runs 4 rounds of workgroups there), 2 = always
    int opt_geglu_co = 0;
    int opt_qkv_co = 1;   // the same choice for the fused QKV GEMM (no k-split exchange in the 4-wave form).  Default 1 since round 6: four prompts per GPU 10.25 -> 10.07 ... 10.11 ms per step (-1.5 %),
                          // 768 workgroups = three per CU, two of them resident; at one prompt (192 workgroups, one per CU: nothing to overlap with) the 4-wave form loses 1.3 %
    int qkv_tile() const { return (opt_qkv_co == 2 || (opt_qkv_co == 1 && M > kCoM)) ? 66 : 61; }
    static constexpr int kCoM = 2048;
    int geglu_tile() const { return !(opt_gemm_pp & 1) ? 13 : (opt_geglu_co == 2 || (opt_geglu_co == 1 && M > kCoM)) ? 66 : 60; }
    // Single-key cross-attention shortcut (needs the LayerNorm-algebra path).  Softmax over ONE valid key is exactly 1, so for a batch element whose
    // context mask has a single valid key -- every unconditional row of classifier-free guidance: the T5 encoding of "" is one EOS token
    // (src/inference.py:44-50, attention.py:131-135) -- cross-attention returns v_key for every query and the cross-attention block adds the
    // step-invariant vector  d = W_o v_key + b_o  (blocks.py:147-151).  ezdit_prepare_context finds those batch elements and computes d per block
    // (zd); the attention-out projection adds it for their rows and emits the GEGLU GEMM's operand for them (k_gemm_ks DUAL / the ping-pong
    // producer), and cross-attention + its out-projection run over the remaining batch elements only -- which must be one contiguous range
    // [act_b0, act_b1), as in the CFG layout [cond.. | uncond..]; otherwise the general path runs.  Exact; the FLOP count of the bench line is NOT reduced for it.
    int opt_xkey1 = 1;
    bool xkey1 = false;          // set by ezdit_prepare_context: the bound context qualifies
    int act_b0 = 0, act_b1 = 0;  // batch elements that still run cross-attention
#ifdef EZ_DIAG
    int opt_zfake = 0;   // DIAGNOSTIC build only: the consumers run their LayerNorm-algebra variant on a FINISHED LayerNorm with neutral statistics (mean 0, variance 1, G' = 0, C' = bias): what the consumer side costs by itself
#endif
    // fused QKV GEMM (per-head LayerNorm + RoPE + the attention layouts in the epilogue, all in registers): 2 = ping-pong kernel (tiles of two whole heads),
    // 0 = no (odd head counts: fp32 projection + k_headnorm).  Static since round 6: the q / k rows of `wqkv` are PACKED in the RoPE-pair order the
    // register epilogue needs (EZDIT_T_QKROPE) whenever two heads tile the projection, so no option can route such a model to the unfused path.
    bool qk_perm() const { return D % (2 * dh) == 0; }
    int qkv_mode() const { return qk_perm() ? 2 : 0; }
    // the cross-attention kernel computes its own q projection (8-wave form: small grids, or forced by fuse_q2 = 2)
    // (nb = batch elements the cross-attention launch covers: with the single-key shortcut only the multi-key ones)
    bool q2_fused(int nb) const { return opt_fuse_q2 && ((long)nb * H * ((L + 63) / 64) <= 512 || opt_fuse_q2 == 2) && Lcp % 128 == 0; }
    // the LayerNorm-algebra path can run for the bound shape under the current options (its tables are built by ezdit_prepare_timesteps only then)
    bool zfuse_usable() const {
        // the cross-attention q projection must be one of the two LayerNorm-algebra consumers: fused into k_attn (small grids) or the ping-pong GEMM with the q epilogue
        return opt_zfuse && (D + zwidth() - 1) / zwidth() <= Z_MAXP && (opt_gemm_pp & 1) && qkv_mode() == 2 && (q2_fused(B) || opt_q2_pp);
    }
    int ztile() const { return (M > kZBigM && !per_row) ? 61 : kZTile; }   // producer of the LayerNorm algebra for the bound shape (the ping-pong producer shares one modulation slot per launch)
    int zwidth() const { return ztile() == 61 ? 144 : 96; }   // statistics part = the producer's tile width
    int opt_wt = 2;   // write-through (sc1) output stores: 0 off, 1 on, 2 = on while B L <= 2048.  The end-of-kernel write-back then has nothing left to flush: -3.5 % step time
                      // for one prompt (4.19 -> 4.04 ms), +1.4 % for four (the step is throughput-bound there and the stores compete with the loads)
    int wt() const { return opt_wt == 2 ? (B * L <= 2048) : opt_wt; }
    int opt_fuse_q2 = 1;                                                                  // cross-attn q projection inside k_attn (one prompt); 2 = at every grid size
    int opt_attn_nkh = 0;                                                                 // attention key sub-blocks per tile (0 = auto)
    int opt_q2_pp = 1;                                                                    // cross-attn q projection at large grids: ping-pong GEMM with the per-head LayerNorm in its epilogue (0: fp32 GEMM + normalisation inside k_attn)
    int opt_attn_xcd = 1;                                                                 // attention: all query tiles of a (batch, head) on one XCD
    // XCD affinity of the residual path at M <= 1024 (placement only, results bit-identical): gemm_panel = shapes (1 D x D, 2 skip, 4 MLP-out)
    // whose split-K GEMM puts ALL workgroups of an M tile on XCD tm % 8; row_affine = the row kernel processes row panel p on XCD p % 8, so
    // a panel's slabs, residual stream and LayerNorm output stay in one XCD's L2 (the L2 keeps its lines across kernel boundaries: a
    // same-XCD consumer of 4 MB is 2 us faster than a cross-XCD one, tools/microbench/xcd_bench.hip).  In situ on MI355X: XL 4.497 ->
    // 4.442 ms/step (+1.2 %, three boxes +0.9 ... +1.5 %), L 3.611 -> 3.518 (+2.7 %); the MLP-out shape gains nothing more and re-reads W
    // 8x (94 vs 66 MB fetched per launch), so it keeps the box placement.
    int opt_gemm_panel = 3, opt_row_affine = 1;
    // bf16 GEMM epilogues (GEGLU output, split-K slabs, q / k heads of the fused QKV GEMM) staged through LDS and written as whole 16-byte
    // chunks: in the MFMA C layout a lane owns one row, so a direct store instruction scatters 4-8 bytes into 32-64 different lines.
    // Bit-identical; XL 4.384 -> 4.281 ms/step (+2.4 %), L +1.8 % for the GEGLU / slab part alone.
    int opt_epi_lds = 1;
    // cross-attention q projection: two K tiles per ring slot, barrier and counted wait (a wave's work per K tile is 3 MFMAs: the loop is its
    // fixed cost per iteration).  Bit-identical; XL 4.319 -> 4.285 ms/step (+0.8 %), L +1.1 %.
    int opt_attn_xk2 = 1;
    // fused cross-attention with the LayerNorm algebra: query rows per workgroup (attn.hip k_attn QT): 0 = 32 when the 64-row grid is <= 128 workgroups (one prompt with the
    // single-key shortcut: 256 workgroups of 32 rows instead of 128 of 64), 64 / 32 = always
    int opt_attn_qtile = 0;
    int opt_cn_overlap = 1;                                                               // fused sampler: ControlNet branch on a side stream, concurrent with the backbone's in-blocks
    hipStream_t cn_stream = nullptr; hipEvent_t cn_fork = nullptr, cn_join = nullptr;
    int opt_row_variant = 1;                                                              // row kernel: 0 = one workgroup per row, 1 = one wave per row
    int debug_stop = 0;  // > 0: ezdit_forward returns after this many launches (unit-test hook)
    int opt_stamp_launch = -1;   // in-situ cycle stamps: the launch with this index of a forward writes them (ezdit_debug_gemm_timestamps buffer; eager launches only)
    int opt_trace_launches = 0;  // print 'index name' of every launch of a forward to stderr
    int steps_done = 0;  // host mirror of the device step counter (ezdit_sampler_run bounds check)
    int device = -1;     // device that was current at ezdit_create
    std::vector<BlkW> blk;
    WRef w_pe, w_fin;
    const float *b_pe = nullptr, *b_fin = nullptr, *w_mask_embed = nullptr, *w_fin_cw = nullptr, *w_fin_cb = nullptr;
    WsPtrs p;
    WRef wref(const std::string& name) const {
        const ezdit_param_info_t& pi = params[pidx.at(name)];
        WRef r; r.W = reinterpret_cast<const bf16_t*>(wblob + pi.offset); r.ld = (int)pi.ld; r.rows = (int)pi.rows_pad;
        return r;
    }

    template <typename T>
    const T* w(const std::string& name) const {
        auto it = pidx.find(name);
        if (it == pidx.end()) {
            fprintf(stderr, "ezdit: unknown parameter slot %s\n", name.c_str());
            abort();
        }
        return reinterpret_cast<const T*>(wblob + params[it->second].offset);
    }
    int pld(const std::string& name) const { return (int)params[pidx.at(name)].ld; }
    template <typename T>
    T* buf(const std::string& name) const {
        return reinterpret_cast<T*>(ws + bufs.at(name).off);
    }
};

namespace {

// ------------------------------------------------------------------------------------------------------
// parameter layout
// ------------------------------------------------------------------------------------------------------
void add_param(ezdit_handle* h, const std::string& name, std::vector<std::string> srcs, int dtype, long rows, long cols,
               int transform = EZDIT_T_NONE) {
    ezdit_param_info_t p;
    memset(&p, 0, sizeof p);
    snprintf(p.name, sizeof p.name, "%s", name.c_str());
    p.nsrc = (int)srcs.size();
    for (int i = 0; i < p.nsrc; ++i) snprintf(p.src[i], sizeof p.src[i], "%s", srcs[i].c_str());
    p.dtype = dtype;
    p.transform = transform;
    p.rows = rows;
    p.cols = cols;
    if (dtype == EZDIT_P_BF16) {
        p.rows_pad = rup(rows, 128);
        p.ld = rup(cols, 64);
    } else {
        p.rows_pad = rows;
        p.ld = cols;
    }
    p.offset = (int64_t)h->param_bytes;
    const size_t bytes = (size_t)p.rows_pad * p.ld * (dtype == EZDIT_P_BF16 ? 2 : 4);
    h->param_bytes += rup((long)bytes, 256);
    h->pidx[name] = (int)h->params.size();
    h->params.push_back(p);
}

std::string blk_prefix(const ezdit_handle* h, int b) {
    char s[64];
    if (h->is_cn) snprintf(s, sizeof s, "in_blocks.%d", b);  // DiTControlNet is not wrapped: no "model." prefix
    else if (b < h->nhalf) snprintf(s, sizeof s, "model.in_blocks.%d", b);
    else if (b == h->nhalf) snprintf(s, sizeof s, "model.mid_block");
    else snprintf(s, sizeof s, "model.out_blocks.%d", b - h->nhalf - 1);
    return s;
}
std::string bn(int b, const char* k) {
    char s[64];
    snprintf(s, sizeof s, "blk%d.%s", b, k);
    return s;
}

void build_params(ezdit_handle* h) {
    const long D = h->D, C = h->C, I = h->I, dh = h->dh, r6 = h->r6;
    const int F = EZDIT_P_F32, Bf = EZDIT_P_BF16;
    const std::string m = h->is_cn ? "" : "model.";
    if (!h->is_cn) add_param(h, "mask_embed", {"mask_embed"}, F, 1, C);
    add_param(h, "pe.w", {m + "patch_embed.proj.weight"}, Bf, D, h->Cin);
    add_param(h, "pe.b", {m + "patch_embed.proj.bias"}, F, 1, D);
    add_param(h, "te.w1", {m + "time_embed.mlp.0.weight"}, F, D, 256);
    add_param(h, "te.b1", {m + "time_embed.mlp.0.bias"}, F, 1, D);
    add_param(h, "te.w2", {m + "time_embed.mlp.2.weight"}, F, D, D);
    add_param(h, "te.b2", {m + "time_embed.mlp.2.bias"}, F, 1, D);
    add_param(h, "ada.w", {m + "time_ada.weight"}, F, 6 * D, D);
    add_param(h, "ada.b", {m + "time_ada.bias"}, F, 1, 6 * D);
    add_param(h, "ce.w1", {m + "context_embed.0.weight"}, Bf, D, h->Cctx);
    add_param(h, "ce.b1", {m + "context_embed.0.bias"}, F, 1, D);
    add_param(h, "ce.w2", {m + "context_embed.2.weight"}, Bf, D, D);
    add_param(h, "ce.b2", {m + "context_embed.2.bias"}, F, 1, D);
    if (!h->is_cn) {
        add_param(h, "adaf.w", {"model.time_ada_final.weight"}, F, 2 * D, D);
        add_param(h, "adaf.b", {"model.time_ada_final.bias"}, F, 1, 2 * D);
        add_param(h, "fin.nw", {"model.final_block.norm.weight"}, F, 1, D);
        add_param(h, "fin.nb", {"model.final_block.norm.bias"}, F, 1, D);
        add_param(h, "fin.w", {"model.final_block.linear.weight"}, Bf, C, D);
        add_param(h, "fin.b", {"model.final_block.linear.bias"}, F, 1, C);
        add_param(h, "fin.cw", {"model.final_block.final_layer.weight"}, F, C, C * 3);
        add_param(h, "fin.cb", {"model.final_block.final_layer.bias"}, F, 1, C);
    } else {
        // DiTControlNetEmbed (controlnet.py:10-39) and the zero-initialised output Linears (:228-234)
        const long c0 = h->c0, c0m = h->c0m, c1 = h->c1, ci = h->cfg.cond_in;
        add_param(h, "cn.cin.w", {"controlnet_pre.conv_in.weight"}, F, c0, ci);
        add_param(h, "cn.cin.b", {"controlnet_pre.conv_in.bias"}, F, 1, c0);
        if (h->cfg.cond_mask) add_param(h, "cn.mask_embed", {"controlnet_pre.mask_embed"}, F, 1, c0);
        add_param(h, "cn.c0.w", {"controlnet_pre.blocks.0.0.weight"}, F, c0m, c0m * 3);
        add_param(h, "cn.c0.b", {"controlnet_pre.blocks.0.0.bias"}, F, 1, c0m);
        add_param(h, "cn.c1.w", {"controlnet_pre.blocks.0.2.weight"}, F, c1, c0m * 3);
        add_param(h, "cn.c1.b", {"controlnet_pre.blocks.0.2.bias"}, F, 1, c1);
        add_param(h, "cn.cout.w", {"controlnet_pre.conv_out.weight"}, F, D, c1);
        add_param(h, "cn.cout.b", {"controlnet_pre.conv_out.bias"}, F, 1, D);
        for (int b = 0; b < h->nblk; ++b) {
            char k[64];
            snprintf(k, sizeof k, "controlnet_zero_blocks.%d", b);
            add_param(h, bn(b, "zw"), {std::string(k) + ".weight"}, Bf, D, D);
            add_param(h, bn(b, "zb"), {std::string(k) + ".bias"}, F, 1, D);
        }
    }
    // per-block parameters, kind-major so that one kind is a constant-stride array over blocks
    struct V { const char* name; const char* key; long n; };
    const V vecs[] = {
        {"n1w", "norm1.weight", D}, {"n1b", "norm1.bias", D}, {"n2w", "norm2.weight", D}, {"n2b", "norm2.bias", D},
        {"n3w", "norm3.weight", D}, {"n3b", "norm3.bias", D}, {"ncw", "norm_context.weight", D},
        {"ncb", "norm_context.bias", D},
        {"a.qnw", "attn.norm_q.weight", dh}, {"a.qnb", "attn.norm_q.bias", dh}, {"a.knw", "attn.norm_k.weight", dh},
        {"a.knb", "attn.norm_k.bias", dh}, {"c.qnw", "cross_attn.norm_q.weight", dh},
        {"c.qnb", "cross_attn.norm_q.bias", dh}, {"c.knw", "cross_attn.norm_k.weight", dh},
        {"c.knb", "cross_attn.norm_k.bias", dh},
        {"bo", "attn.proj.bias", D}, {"bo2", "cross_attn.proj.bias", D}, {"b2", "mlp.net.2.bias", D},
        {"table", "adaln.scale_shift_table", 6 * D},
    };
    for (const V& v : vecs)
        for (int b = 0; b < h->nblk; ++b) add_param(h, bn(b, v.name), {blk_prefix(h, b) + "." + v.key}, F, 1, v.n);
    // per-STEP matrices of one block are contiguous
    for (int b = 0; b < h->nblk; ++b) {
        const std::string p = blk_prefix(h, b);
        add_param(h, bn(b, "b1"), {p + ".mlp.net.0.proj.bias"}, F, 1, 2 * I, EZDIT_T_GEGLU8);
        if (b > h->nhalf) {
            add_param(h, bn(b, "snw"), {p + ".skip_norm.weight"}, F, 1, 2 * D);
            add_param(h, bn(b, "snb"), {p + ".skip_norm.bias"}, F, 1, 2 * D);
            add_param(h, bn(b, "wskip"), {p + ".skip_linear.weight"}, Bf, D, 2 * D);
            add_param(h, bn(b, "bskip"), {p + ".skip_linear.bias"}, F, 1, D);
        }
        add_param(h, bn(b, "wqkv"), {p + ".attn.to_q.weight", p + ".attn.to_k.weight", p + ".attn.to_v.weight"}, Bf, 3 * D, D, h->qk_perm() ? EZDIT_T_QKROPE : EZDIT_T_NONE);
        add_param(h, bn(b, "wo"), {p + ".attn.proj.weight"}, Bf, D, D);
        add_param(h, bn(b, "wq2"), {p + ".cross_attn.to_q.weight"}, Bf, D, D);
        add_param(h, bn(b, "wo2"), {p + ".cross_attn.proj.weight"}, Bf, D, D);
        add_param(h, bn(b, "w1"), {p + ".mlp.net.0.proj.weight"}, Bf, 2 * I, D, EZDIT_T_GEGLU8);
        add_param(h, bn(b, "w2"), {p + ".mlp.net.2.weight"}, Bf, D, I);
    }
    // used once per CALL only (context K/V projection, AdaLN-SOLA low-rank factors)
    for (int b = 0; b < h->nblk; ++b) {
        const std::string p = blk_prefix(h, b);
        add_param(h, bn(b, "wkv2"), {p + ".cross_attn.to_k.weight", p + ".cross_attn.to_v.weight"}, Bf, 2 * D, D);
        add_param(h, bn(b, "lora_a"), {p + ".adaln.lora_a.weight"}, F, r6, D);
        add_param(h, bn(b, "lora_b"), {p + ".adaln.lora_b.weight"}, F, 6 * D, r6);
    }
}

// ------------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------------
size_t carve(const ezdit_handle* h, int B, int L, int Lc, int n_slots, std::map<std::string, Buf>* out) {
    size_t off = 0;
    auto add = [&](const char* name, size_t bytes) {
        Buf b;
        b.off = off;
        b.bytes = bytes;
        if (out) (*out)[name] = b;
        off += (size_t)rup((long)bytes, 256);
    };
    const long D = h->D, C = h->C, H = h->H;
    const long M = (long)B * L, Mp = rup(M, 128), Lp = rup(L, 128), Lcp = rup(Lc, 128);  // attention stages 64- or 128-key tiles
    const long Mc = (long)B * Lc, Mcp = rup(Mc, 128);
    const int nblk = h->nblk;
    add("ints", 256 * sizeof(int));                       // [0] cur_step, [8] CFG arrival counter, [16..] row_slot (<= 240 rows)
    add("rope_cos", (size_t)h->cfg.max_len * (h->dh / 2) * 4);
    add("rope_sin", (size_t)h->cfg.max_len * (h->dh / 2) * 4);
    add("coef", (size_t)(n_slots > 0 ? n_slots : 1) * 8 * 4);
    add("cfgpart", (size_t)B * 64 * 4 * 4);
    add("sink", 256);
    if (h->is_cn) {
        const long Lc2 = 2L * L;
        add("cn_e0", (size_t)B * h->c0 * Lc2 * 4);
        add("cn_e1", (size_t)B * h->c0m * Lc2 * 4);
        add("cn_e2", (size_t)B * h->c1 * L * 4);
        add("cembed", (size_t)rup((long)B * L, 128) * h->D * 4);
        add("cnres", (size_t)h->nhalf * rup((long)B * L, 128) * h->D * 4);
        add("skipbf", (size_t)rup((long)B * L, 128) * h->ldD * 2);
    }
    add("ape", Mp * h->ldPE * 2);
    add("h", Mp * D * 4);
    add("skips", (size_t)h->nhalf * Mp * D * 4);
    add("u", Mp * h->ldD * 2);
    add("ucat", Mp * h->ld2D * 2);                        // LN_2D([x | skip]) of the out-blocks: its own buffer, because the skip GEMM that reads it
                                                          // writes `u` from inside the same launch (fused row operator)
    add("qkv", Mp * 3 * D * 4);
    add("q", (size_t)B * H * Lp * h->DQK * 2);
    add("k", (size_t)B * H * Lp * h->DQK * 2);
    add("v", (size_t)B * H * Lp * h->DV * 2);
    add("ao", Mp * h->ldD * 2);
    add("act", Mp * h->ldI * 2);
    add("part", (size_t)8 * Mp * D * 4);
    add("y", Mp * C * 4);
    add("pred", (size_t)B * C * L * 4);
    // context (step invariant)
    add("kmask", rup(Mc, 256));
    add("ctx_bf", Mcp * h->ldCtx * 2);
    add("c1", Mcp * D * 4);
    add("c1b", Mcp * h->ldD * 2);
    add("c2", Mcp * D * 4);
    add("cu", Mcp * h->ldD * 2);
    add("ckv", Mcp * 2 * D * 4);
    add("kc", (size_t)nblk * B * H * Lcp * h->DQK * 2);
    add("vc", (size_t)nblk * B * H * Lcp * h->DV * 2);
    // time path
    const long ns = n_slots > 0 ? n_slots : 1;
    add("ts", ns * 4);
    add("t1", ns * D * 4);
    add("tt", ns * D * 4);
    add("ada", ns * 6 * D * 4);
    add("adaf", ns * 2 * D * 4);
    add("la", ns * h->r6 * 4);
    add("lora", ns * nblk * 6 * D * 4);
    add("mod", ns * nblk * 6 * D * 4);
    add("modf", ns * 2 * D * 4);
    // LayerNorm algebra: partial row statistics (chunks of 64 columns, up to the 2D-wide concat), G' / C' tables per modulation slot
    {
        const long I2 = 2L * h->I, N3 = 3L * D, nmax = I2 > N3 ? I2 : N3;
        add("zstat", (size_t)Mp * (((2 * D + 63) / 64) > 2 * Z_MAXP ? ((2 * D + 63) / 64) : 2 * Z_MAXP) * 8);   // two part ranges of Z_MAXP: the second one holds the statistics of the x half in front of an out-block (skip_z)
        add("zt_qkv", (size_t)ns * nblk * 2 * N3 * 4);
        add("zt_geglu", (size_t)ns * nblk * 2 * I2 * 4);
        add("zt_q2", (size_t)nblk * 2 * D * 4);
        add("zd", (size_t)nblk * B * D * 4);
        add("zA", (size_t)rup(4 * ns, 128) * h->ldD * 2);
        add("ztmp", (size_t)rup(4 * ns, 128) * nmax * 4);
#ifdef EZ_DIAG
        add("zneutral", (size_t)Z_MAXP * Mp * 8);
        add("zzeros", (size_t)nmax * 4);
#endif
        // skip path by the LayerNorm algebra (opt_skip_z), BEHIND everything else: the buffers above keep the offsets -- the relative placement in the HBM channels -- the
        // round's measurements were made with.  One [x | skip] operand per out-block (the in-block fills the right half long before the out-block runs), the skips' statistics, static tables
        add("ucat_z", (size_t)((!h->is_cn && h->nhalf > 0) ? h->nhalf : 1) * Mp * h->ld2D * 2);
        add("zstat_skip", (size_t)(h->nhalf > 0 ? h->nhalf : 1) * Z_MAXP * Mp * 8);
        add("zt_skip", (size_t)(h->nhalf > 0 ? h->nhalf : 1) * 2 * D * 4);
    }
    return off;
}

// ------------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------------
struct Ctx {
    ezdit_handle* h;
    hipStream_t st;
    const HeadNormArgs* hn = nullptr;   // one-shot: EPI_QKV epilogue arguments
    bool panel = false;                 // one-shot: panel placement of a split-K GEMM (GemmArgs.xcd_panel)
    const float* zG = nullptr; const float* zC = nullptr; long zt_stride = 0;   // one-shot: LayerNorm algebra in the consumer's epilogue
    const float2* zstat = nullptr;      // one-shot: ... its partial statistics when they are not in the shared buffer (null = WsPtrs.zstat)
    int zrow0 = 0, zb0 = 0;   // one-shot: the launch covers a row sub-range that starts at row zrow0 = batch element zb0 (statistics table / per-row slot offsets)
    const int* cur = nullptr; const int* row_slot = nullptr;   // modulation slot of this forward (a ControlNet attached to the fused sampler reads the BACKBONE's step counter)
    // first launch failure of this call (hipGetLastError after EVERY launch: a rejected launch -- LDS limit, bad grid,
    // unsupported fused configuration -- must surface as an error code, never as stale numbers)
    hipError_t err = hipSuccess;
    int rc = EZDIT_OK;
    const char* where = nullptr;
    // cycle-stamp buffer for the launch about to be issued (nullptr unless the 'stamp_launch' option selects it)
    unsigned long long* stamps() const { return (g_gemm_ts && h->launches == h->opt_stamp_launch) ? g_gemm_ts : nullptr; }
    void launched(const char* what, int launch_rc = 0) {
        if (h->opt_trace_launches) fprintf(stderr, "launch %d %s\n", h->launches, what);
        h->launches++;
        if (launch_rc != 0 && rc == EZDIT_OK) { rc = launch_rc; where = what; }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess && err == hipSuccess) { err = e; where = what; }
    }
    bool bad() const { return err != hipSuccess || rc != EZDIT_OK; }
    int result() const {
        if (rc != EZDIT_OK) return fail(rc, "%s: configuration not supported by the HIP kernels", where ? where : "?");
        if (err != hipSuccess) return fail(EZDIT_E_HIP, "launch of %s failed: %s", where ? where : "?", hipGetErrorString(err));
        return EZDIT_OK;
    }
};

void gemm(Ctx& c, const bf16_t* A, int lda, const WRef& w, const float* bias, void* out, int ldo, int M, int N,
          int epi, int tile, int splitk = 1, long slab = 0) {
    ezdit_handle* h = c.h;
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = A;
    g.lda = lda;
    g.W = w.W;
    g.ldw = w.ld;
    g.wrows = w.rows;
    g.bias = bias;
    g.out = out;
    g.ldo = ldo;
    g.slab_stride = slab;
    g.M = M;
    g.N = N;
    g.K = w.ld;
    g.splitk = splitk;
    g.epi = epi;
    g.tile = tile;
    g.xcd_map = 1;
    g.part_bf16 = (epi == EPI_PARTIAL) ? 1 : 0;   // split-K slabs in bf16
    g.wt = h->wt();
    g.epi_lds = h->opt_epi_lds;
    g.rows_per_b = 1;
    if (c.hn) { g.hn = *c.hn; c.hn = nullptr; }
    if (c.panel) { g.xcd_panel = 1; c.panel = false; }
#ifdef EZ_DIAG
    if (!c.zG && h->opt_zfake && (epi == EPI_QKV || epi == EPI_GEGLU) && (tile == 60 || tile == 61 || tile == 66) && (h->D + h->zwidth() - 1) / h->zwidth() <= Z_MAXP) {
        g.zw = h->zwidth(); g.zstat_in = h->buf<float2>("zneutral"); g.zs_stride = h->Mp; g.zparts = (h->D + g.zw - 1) / g.zw; g.zD = h->D;
        g.zG = h->buf<float>("zzeros"); g.zC = bias ? bias : g.zG; g.zt_slot_stride = 0; g.zeps = 1e-5f; g.rows_per_b = h->L;
    }
#endif
    if (c.zG) {
        g.zw = h->zwidth(); g.zstat_in = (c.zstat ? c.zstat : h->p.zstat) + c.zrow0; g.zs_stride = h->Mp; g.zparts = (h->D + g.zw - 1) / g.zw; g.zD = h->D; g.zG = c.zG; g.zC = c.zC; g.zt_slot_stride = c.zt_stride; g.zeps = 1e-5f;
        g.cur_step = c.cur ? c.cur : h->p.ints; g.row_slot = c.cur ? c.row_slot : (h->per_row ? h->p.ints + 16 : nullptr); g.rows_per_b = h->L;
        if (g.row_slot) g.row_slot += c.zb0;
        c.zG = nullptr; c.zstat = nullptr;
    }
    c.zrow0 = 0; c.zb0 = 0;
    g.ts = c.stamps(); g.ts_cap = g_gemm_ts_cap;
    c.launched(epi == EPI_GEGLU ? "k_gemm (GEGLU)" : epi == EPI_QKV ? "k_gemm (QKV)" : epi == EPI_PARTIAL ? "k_gemm (split-K slabs)" : "k_gemm", launch_gemm(g, c.st));
}

// Tile / split-K choices: measured in situ on MI355X (DESIGN.md section 4 has the numbers); fixed since round 5 except tile_partial.
int tile_for(const ezdit_handle* h, int M, bool partial) {
    if (M <= 2048) return partial ? h->opt_tile_partial : ezdit_handle::kTileF32;
    return partial ? ezdit_handle::kTilePartialBig : ezdit_handle::kTileF32;
}

int pick_splitk(const ezdit_handle* h, int M, int N, int K) {
    const int tiles = ((M + 127) / 128) * ((N + 63) / 64);
    const int nk = K / 64;
    if (M > 2048) return ezdit_handle::kSplitBig < nk ? ezdit_handle::kSplitBig : nk;
    if (tiles >= 256) return 1;
    return ezdit_handle::kSplitK < nk ? ezdit_handle::kSplitK : nk;  // fewer slabs = less row-kernel traffic
}

// residual GEMM: part = A . W^T as split-K slabs (reduced by the row kernel that follows)
int gemm_partial(Ctx& c, const bf16_t* A, int lda, const WRef& w, int M, int N) {
    ezdit_handle* h = c.h;
    const int K = w.ld;
    const int s = pick_splitk(h, M, N, K);
    const int shape_bit = K >= 4 * N ? 4 : K >= 2 * N ? 2 : 1;
    c.panel = (h->opt_gemm_panel & shape_bit) && M <= 1024;
    gemm(c, A, lda, w, nullptr, h->p.part, h->D, M, N, EPI_PARTIAL, tile_for(h, M, true), s, (long)h->Mp * h->D);
    return s;
}

void resolve_weights(ezdit_handle* h) {
    h->blk.assign(h->nblk, BlkW{});
    for (int b = 0; b < h->nblk; ++b) {
        BlkW& k = h->blk[b];
        k.aqnw = h->w<float>(bn(b, "a.qnw")); k.aqnb = h->w<float>(bn(b, "a.qnb"));
        k.aknw = h->w<float>(bn(b, "a.knw")); k.aknb = h->w<float>(bn(b, "a.knb"));
        k.cqnw = h->w<float>(bn(b, "c.qnw")); k.cqnb = h->w<float>(bn(b, "c.qnb"));
        k.bo = h->w<float>(bn(b, "bo")); k.bo2 = h->w<float>(bn(b, "bo2")); k.b2 = h->w<float>(bn(b, "b2"));
        k.n2w = h->w<float>(bn(b, "n2w")); k.n2b = h->w<float>(bn(b, "n2b"));
        k.b1 = h->w<float>(bn(b, "b1"));
        k.wqkv = h->wref(bn(b, "wqkv")); k.wo = h->wref(bn(b, "wo")); k.wq2 = h->wref(bn(b, "wq2"));
        k.wo2 = h->wref(bn(b, "wo2")); k.w1 = h->wref(bn(b, "w1")); k.w2 = h->wref(bn(b, "w2"));
        if (!h->is_cn && b > h->nhalf) {
            k.snw = h->w<float>(bn(b, "snw")); k.snb = h->w<float>(bn(b, "snb"));
            k.wskip = h->wref(bn(b, "wskip")); k.bskip = h->w<float>(bn(b, "bskip"));
        }
        if (h->is_cn) { k.zw = h->wref(bn(b, "zw")); k.zb = h->w<float>(bn(b, "zb")); }
    }
    h->w_pe = h->wref("pe.w"); h->b_pe = h->w<float>("pe.b");
    if (!h->is_cn) {
        h->w_fin = h->wref("fin.w"); h->b_fin = h->w<float>("fin.b");
        h->w_fin_cw = h->w<float>("fin.cw"); h->w_fin_cb = h->w<float>("fin.cb");
        h->w_mask_embed = h->w<float>("mask_embed");
    }
}

void resolve_workspace(ezdit_handle* h) {
    WsPtrs& p = h->p;
    memset(&p, 0, sizeof p);
    p.ints = h->buf<int>("ints"); p.rope_cos = h->buf<float>("rope_cos"); p.rope_sin = h->buf<float>("rope_sin");
    p.coef = h->buf<float>("coef"); p.cfgpart = h->buf<float>("cfgpart");
    p.ape = h->buf<bf16_t>("ape"); p.h = h->buf<float>("h"); p.skips = h->buf<float>("skips"); p.u = h->buf<bf16_t>("u"); p.ucat = h->buf<bf16_t>("ucat");
    p.qkv = h->buf<float>("qkv"); p.q = h->buf<bf16_t>("q"); p.k = h->buf<bf16_t>("k"); p.v = h->buf<bf16_t>("v");
    p.ao = h->buf<bf16_t>("ao"); p.act = h->buf<bf16_t>("act"); p.part = h->buf<float>("part"); p.y = h->buf<float>("y");
    p.pred = h->buf<float>("pred"); p.kmask = h->buf<uint8_t>("kmask"); p.kc = h->buf<bf16_t>("kc"); p.vc = h->buf<bf16_t>("vc");
    p.mod = h->buf<float>("mod"); p.modf = h->buf<float>("modf");
    p.zd = h->buf<float>("zd");
    p.zstat = h->buf<float2>("zstat"); p.zt_qkv = h->buf<float>("zt_qkv"); p.zt_geglu = h->buf<float>("zt_geglu"); p.zt_q2 = h->buf<float>("zt_q2");
    p.zstat_skip = h->buf<float2>("zstat_skip"); p.zt_skip = h->buf<float>("zt_skip"); p.ucat_z = h->buf<bf16_t>("ucat_z");
    if (h->is_cn) { p.cembed = h->buf<float>("cembed"); p.cnres = h->buf<float>("cnres"); p.skipbf = h->buf<bf16_t>("skipbf"); }
}

}  // namespace

// ======================================================================================================
extern "C" {

int ezdit_abi_version(void) { return EZDIT_ABI_VERSION; }
const char* ezdit_last_error(void) { return g_err.c_str(); }

int ezdit_create(const ezdit_config* cfg, ezdit_handle** out) {
    if (!cfg || !out) return fail(EZDIT_E_INVALID, "null argument");
    if (cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->embed_dim % cfg->num_heads)
        return fail(EZDIT_E_INVALID, "embed_dim %d not divisible by num_heads %d", cfg->embed_dim, cfg->num_heads);
    const int dh = cfg->embed_dim / cfg->num_heads;
    if (dh != 64 && dh != 72) return fail(EZDIT_E_UNSUPPORTED, "head_dim %d: only 64 and 72 are implemented", dh);
    if (cfg->embed_dim > 1280 || cfg->embed_dim % 4) return fail(EZDIT_E_UNSUPPORTED, "embed_dim %d > 1280", cfg->embed_dim);
    if (cfg->in_chans != 2 * cfg->out_chans + 1)
        return fail(EZDIT_E_UNSUPPORTED, "in_chans %d != 2*out_chans+1 (MaskDiT concat)", cfg->in_chans);
    if (cfg->depth < 2 || cfg->depth % 2) return fail(EZDIT_E_UNSUPPORTED, "depth %d must be even", cfg->depth);
    if (cfg->mlp_ratio != 4.0f) return fail(EZDIT_E_UNSUPPORTED, "mlp_ratio %g", (double)cfg->mlp_ratio);
    ezdit_handle* h = new ezdit_handle();
    h->cfg = *cfg;
    if (h->cfg.max_len <= 0) h->cfg.max_len = 2048;
    h->D = cfg->embed_dim;
    h->H = cfg->num_heads;
    h->dh = dh;
    h->nhalf = cfg->depth / 2;
    h->is_cn = cfg->controlnet != 0;
    h->nblk = h->is_cn ? cfg->depth / 2 : cfg->depth + 1;
    if (h->is_cn) {
        if (cfg->cond_in <= 0 || cfg->cond_c0 <= 0 || cfg->cond_c1 <= 0) {
            delete h;
            return fail(EZDIT_E_INVALID, "controlnet needs cond_in / cond_blocks");
        }
        h->c0 = cfg->cond_c0;
        h->c0m = cfg->cond_c0 + (cfg->cond_mask ? 1 : 0);
        h->c1 = cfg->cond_c1;
    }
    h->I = (int)(cfg->embed_dim * cfg->mlp_ratio);
    h->C = cfg->out_chans;
    h->Cin = cfg->in_chans;
    h->Cctx = cfg->context_dim;
    h->r6 = 6 * cfg->ada_sola_rank;
    h->scaling = cfg->ada_sola_alpha / (float)cfg->ada_sola_rank;
    h->DQK = dh == 64 ? 64 : 80;
    h->DV = dh == 64 ? 64 : 96;
    h->ldD = (int)rup(h->D, 64);
    h->ld2D = (int)rup(2 * h->D, 64);
    h->ldI = (int)rup(h->I, 64);
    h->ldPE = (int)rup(h->Cin, 64);
    h->ldCtx = (int)rup(h->Cctx, 64);
    build_params(h);
    (void)hipGetDevice(&h->device);
    *out = h;
    return EZDIT_OK;
}

static void drop_graph(ezdit_handle* h) {
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
}

int ezdit_destroy(ezdit_handle* h) {
    if (!h) return EZDIT_OK;
    for (ezdit_handle* u : h->cn_users)   // a backbone must not keep a dangling pointer to this ControlNet
        if (u->cn == h) { u->cn = nullptr; u->cn_scale = 1.0f; drop_graph(u); }
    if (h->cn) {
        auto& v = h->cn->cn_users;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == h) { v.erase(v.begin() + i); break; }
    }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    if (h->cn_fork) (void)hipEventDestroy(h->cn_fork);
    if (h->cn_join) (void)hipEventDestroy(h->cn_join);
    if (h->cn_stream) (void)hipStreamDestroy(h->cn_stream);
    delete h;
    return EZDIT_OK;
}

int ezdit_param_count(const ezdit_handle* h) { return h ? (int)h->params.size() : 0; }
int ezdit_param_info(const ezdit_handle* h, int i, ezdit_param_info_t* out) {
    if (!h || !out || i < 0 || i >= (int)h->params.size()) return fail(EZDIT_E_INVALID, "bad parameter index %d", i);
    *out = h->params[i];
    return EZDIT_OK;
}
size_t ezdit_param_bytes(const ezdit_handle* h) { return h ? h->param_bytes : 0; }

int ezdit_bind_weights(ezdit_handle* h, const void* blob, size_t bytes) {
    if (!h || !blob) return fail(EZDIT_E_INVALID, "null argument");
    if (bytes < h->param_bytes) return fail(EZDIT_E_INVALID, "weight blob %zu < required %zu bytes", bytes, h->param_bytes);
    h->wblob = reinterpret_cast<const char*>(blob);
    h->ctx_ready = h->ts_ready = false;
    resolve_weights(h);
    return EZDIT_OK;
}

size_t ezdit_workspace_bytes(const ezdit_handle* h, int B, int L, int Lc, int n_slots) {
    if (!h || B <= 0 || L <= 0 || Lc <= 0) return 0;
    return carve(h, B, L, Lc, n_slots, nullptr);
}

int ezdit_bind_workspace(ezdit_handle* h, void* ws, size_t bytes, int B, int L, int Lc, int n_slots, ezdit_stream stream) {
    if (!h || !ws) return fail(EZDIT_E_INVALID, "null argument");
    if (B <= 0 || B > 240 || L <= 0 || Lc <= 0) return fail(EZDIT_E_INVALID, "bad shape B=%d L=%d Lc=%d", B, L, Lc);
    if (L > h->cfg.max_len) return fail(EZDIT_E_INVALID, "L=%d exceeds max_len=%d", L, h->cfg.max_len);
    std::map<std::string, Buf> bufs;
    const size_t need = carve(h, B, L, Lc, n_slots, &bufs);
    if (bytes < need) return fail(EZDIT_E_INVALID, "workspace %zu < required %zu bytes", bytes, need);
    hipStream_t st = (hipStream_t)stream;
    h->ws = reinterpret_cast<char*>(ws);
    h->ws_bytes = bytes;
    h->bufs = bufs;
    h->B = B; h->L = L; h->Lc = Lc; h->n_slots = n_slots > 0 ? n_slots : 1;
    h->M = B * L; h->Mp = (int)rup(h->M, 128); h->Lp = (int)rup(L, 128); h->Lcp = (int)rup(Lc, 128);
    h->Mc = B * Lc;
    resolve_workspace(h);
    h->steps_done = 0;
    h->ctx_ready = h->ts_ready = h->cond_ready = false;
    drop_graph(h);
    for (ezdit_handle* u : h->cn_users) drop_graph(u);   // a graph captured with this ControlNet attached points at its old buffers
    // zero everything once: all padding rows / columns / keys stay zero for the lifetime of the binding
    HIPCHK(hipMemsetAsync(ws, 0, need, st));
    launch_rope_table(h->buf<float>("rope_cos"), h->buf<float>("rope_sin"), h->cfg.max_len, h->dh, st);
#ifdef EZ_DIAG
    {   // zfake diagnostic: statistics of a zero-mean, unit-variance row in parts of zwidth() columns
        const int zw = h->zwidth(), parts = (h->D + zw - 1) / zw;
        if (parts <= Z_MAXP) {
            std::vector<float2> hst((size_t)Z_MAXP * h->Mp, make_float2(0.f, 0.f));
            for (int k = 0; k < parts; ++k)
                for (int r = 0; r < h->Mp; ++r) hst[(size_t)k * h->Mp + r] = make_float2(0.f, (float)(k == parts - 1 ? h->D - zw * (parts - 1) : zw));
            HIPCHK(hipMemcpyAsync(h->buf<float2>("zneutral"), hst.data(), hst.size() * sizeof(float2), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
#endif
    return EZDIT_OK;
}

// ------------------------------------------------------------------------------------------------------
int ezdit_prepare_context(ezdit_handle* h, const float* ctx, const uint8_t* mask, ezdit_stream stream) {
    if (!h || !ctx) return fail(EZDIT_E_INVALID, "null argument");
    if (!h->wblob || !h->ws) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    Ctx c{h, (hipStream_t)stream};
    const int Mc = h->Mc, D = h->D;
    uint8_t* km = h->buf<uint8_t>("kmask");
    if (mask) HIPCHK(hipMemcpyAsync(km, mask, Mc, hipMemcpyDeviceToDevice, c.st));
    else HIPCHK(hipMemsetAsync(km, 1, Mc, c.st));
    // single-key batch elements (opt_xkey1): their cross-attention output is a constant.  The mask is read back once per call (B Lc bytes)
    std::vector<int> key1(h->B, -1);   // the valid key of a single-key batch element, -1 otherwise
    const bool old_x1 = h->xkey1; const int old_b0 = h->act_b0, old_b1 = h->act_b1;
    h->xkey1 = false; h->act_b0 = 0; h->act_b1 = h->B;
    // (the read-back below synchronises the stream: ezdit_prepare_context is not a per-step entry point and must not be called inside a stream capture;
    // it is skipped when its result could not be used -- the option must be set BEFORE this call)
    if (mask && h->opt_xkey1 && h->zfuse_usable()) {
        std::vector<uint8_t> hm((size_t)Mc);
        HIPCHK(hipMemcpyAsync(hm.data(), mask, Mc, hipMemcpyDeviceToHost, c.st));
        HIPCHK(hipStreamSynchronize(c.st));
        int first = -1, last = -1, n1 = 0;
        for (int b = 0; b < h->B; ++b) {
            int cnt = 0, at = -1;
            for (int j = 0; j < h->Lc; ++j) if (hm[(size_t)b * h->Lc + j]) { ++cnt; at = j; }
            if (cnt == 1) { key1[b] = at; ++n1; }
            else { if (first < 0) first = b; last = b; }
        }
        bool contiguous = true;
        for (int b = first; first >= 0 && b <= last; ++b) if (key1[b] >= 0) contiguous = false;
        if (n1 > 0 && contiguous) {
            h->xkey1 = true;
            h->act_b0 = first < 0 ? 0 : first;
            h->act_b1 = first < 0 ? 0 : last + 1;
        }
    }
    // the captured step bakes (xkey1, act_b0, act_b1) into its launch structure (DUAL attention-out, cross-attention over a batch sub-range):
    // a context with another single-key pattern must not replay the old graph -- nor may a backbone that captured this ControlNet's kernels
    if (h->xkey1 != old_x1 || h->act_b0 != old_b0 || h->act_b1 != old_b1) {
        drop_graph(h);
        for (ezdit_handle* u : h->cn_users) drop_graph(u);
    }
    // context_embed: Linear -> SiLU -> Linear  (udit.py:94-97)
    launch_cast_bf16(ctx, h->Cctx, h->buf<bf16_t>("ctx_bf"), h->ldCtx, Mc, h->Cctx, 0, c.st);
    gemm(c, h->buf<bf16_t>("ctx_bf"), h->ldCtx, h->wref("ce.w1"), h->w<float>("ce.b1"), h->buf<float>("c1"), D, Mc, D, EPI_F32, tile_for(h, Mc, false));
    launch_cast_bf16(h->buf<float>("c1"), D, h->buf<bf16_t>("c1b"), h->ldD, Mc, D, 1, c.st);
    gemm(c, h->buf<bf16_t>("c1b"), h->ldD, h->wref("ce.w2"), h->w<float>("ce.b2"), h->buf<float>("c2"), D, Mc, D, EPI_F32, tile_for(h, Mc, false));
    for (int b = 0; b < h->nblk; ++b) {
        // norm_context (blocks.py:150) -> to_k / to_v -> head LayerNorm on k (attention.py:128-142)
        RowArgs r;
        memset(&r, 0, sizeof r);
        r.h_in = h->buf<float>("c2");
        r.mode = 0;
        r.ln_g = h->w<float>(bn(b, "ncw"));
        r.ln_c = h->w<float>(bn(b, "ncb"));
        r.u = h->buf<bf16_t>("cu");
        r.ld_u = h->ldD;
        r.M = Mc; r.D = D; r.L = h->Lc;
        launch_row(r, c.st);
        gemm(c, h->buf<bf16_t>("cu"), h->ldD, h->wref(bn(b, "wkv2")), nullptr, h->buf<float>("ckv"), 2 * D, Mc, 2 * D, EPI_F32, tile_for(h, Mc, false));
        HeadNormArgs hn;
        memset(&hn, 0, sizeof hn);
        hn.x = h->buf<float>("ckv"); hn.ldx = 2 * D;
        hn.q_col = -1; hn.k_col = 0; hn.v_col = D;
        hn.kn_w = h->w<float>(bn(b, "c.knw")); hn.kn_b = h->w<float>(bn(b, "c.knb"));
        hn.k = h->buf<bf16_t>("kc") + (size_t)b * h->B * h->H * h->Lcp * h->DQK;
        hn.v = h->buf<bf16_t>("vc") + (size_t)b * h->B * h->H * h->Lcp * h->DV;
        hn.B = h->B; hn.H = h->H; hn.L = h->Lc; hn.Lp = h->Lcp; hn.dh = h->dh;
        launch_headnorm(hn, c.st);
        c.launched("k_headnorm");
        if (h->xkey1) {
            // d[b] = W_o v_key + b_o for the single-key batch elements: v_key is row (b Lc + key) of the fp32 value projection above, rounded to bf16
            // as the attention kernel's V^T operand and output would be (P = 1 exactly), W_o the bf16 out-projection the step uses
            const WRef wo2 = h->wref(bn(b, "wo2"));
            GemvBatch gb;
            gb.x = h->buf<float>("ckv"); gb.y = h->p.zd + (size_t)b * h->B * D; gb.n = 0;
            for (int e = 0; e <= h->B; ++e) {   // one launch per GEMV_MAXB single-key batch elements (one launch per block for every CFG batch up to 32 prompts)
                if (e < h->B && key1[e] >= 0) {
                    gb.xoff[gb.n] = ((long)e * h->Lc + key1[e]) * 2 * D + D;
                    gb.yoff[gb.n] = (long)e * D;
                    gb.n++;
                }
                if (gb.n == GEMV_MAXB || (e == h->B && gb.n > 0)) {
                    launch_gemv_bf16w(gb, 1, wo2.W, wo2.ld, h->w<float>(bn(b, "bo2")), D, D, c.st);
                    c.launched("k_gemv_bf16w");
                    gb.n = 0;
                }
            }
        }
    }
    if (c.bad()) return c.result();
    h->ctx_ready = true;
    return EZDIT_OK;
}

int ezdit_prepare_timesteps(ezdit_handle* h, const int32_t* ts, int n, int per_row, ezdit_stream stream) {
    if (!h || !ts) return fail(EZDIT_E_INVALID, "null argument");
    if (!h->wblob || !h->ws) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    if (n <= 0 || n > h->n_slots) return fail(EZDIT_E_INVALID, "n=%d timesteps but workspace holds %d slots", n, h->n_slots);
    if (per_row && n != h->B) return fail(EZDIT_E_INVALID, "per_row needs n == B (%d != %d)", n, h->B);
    hipStream_t st = (hipStream_t)stream;
    const int D = h->D, nblk = h->nblk;
    h->per_row = per_row;   // (the producer choice of the LayerNorm algebra depends on it: ztile())
    int* ints = h->buf<int>("ints");
    HIPCHK(hipMemcpyAsync(h->buf<int>("ts"), ts, (size_t)n * 4, hipMemcpyHostToDevice, st));
    std::vector<int> rs(240, 0);
    if (per_row) for (int b = 0; b < h->B; ++b) rs[b] = b;
    HIPCHK(hipMemcpyAsync(ints + 16, rs.data(), 240 * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // rs is a stack/heap temporary; prepare is not on the per-step path
    launch_set_int(ints, 0, 0, st);
    // TimestepEmbedder + time_act (modules.py:50-60, udit.py:313)
    launch_linear_f32(nullptr, h->buf<int>("ts"), 1, h->w<float>("te.w1"), h->w<float>("te.b1"), h->buf<float>("t1"), n, D, 256, 1, D, st);
    launch_linear_f32(h->buf<float>("t1"), nullptr, 0, h->w<float>("te.w2"), h->w<float>("te.b2"), h->buf<float>("tt"), n, D, D, 1, D, st);
    launch_linear_f32(h->buf<float>("tt"), nullptr, 0, h->w<float>("ada.w"), h->w<float>("ada.b"), h->buf<float>("ada"), n, 6 * D, D, 0, 6 * D, st);
    if (!h->is_cn)
        launch_linear_f32(h->buf<float>("tt"), nullptr, 0, h->w<float>("adaf.w"), h->w<float>("adaf.b"), h->buf<float>("adaf"), n, 2 * D, D, 0, 2 * D, st);
    for (int b = 0; b < nblk; ++b) {
        launch_linear_f32(h->buf<float>("tt"), nullptr, 0, h->w<float>(bn(b, "lora_a")), nullptr, h->buf<float>("la"), n, h->r6, D, 0, h->r6, st);
        launch_linear_f32(h->buf<float>("la"), nullptr, 0, h->w<float>(bn(b, "lora_b")), nullptr,
                          h->buf<float>("lora") + (size_t)b * 6 * D, n, 6 * D, h->r6, 0, (long)nblk * 6 * D, st);
    }
    ModFinalizeArgs m;
    memset(&m, 0, sizeof m);
    m.ada = h->buf<float>("ada");
    m.lora = h->buf<float>("lora");
    m.scaling = h->scaling;
    m.table = h->w<float>(bn(0, "table"));
    m.table_stride = nblk > 1 ? (h->w<float>(bn(1, "table")) - h->w<float>(bn(0, "table"))) : 0;
    m.n1w = h->w<float>(bn(0, "n1w")); m.n1b = h->w<float>(bn(0, "n1b"));
    m.n3w = h->w<float>(bn(0, "n3w")); m.n3b = h->w<float>(bn(0, "n3b"));
    m.norm_stride = nblk > 1 ? (h->w<float>(bn(1, "n1w")) - h->w<float>(bn(0, "n1w"))) : 0;
    m.mod = h->buf<float>("mod");
    m.has_final = h->is_cn ? 0 : 1;
    if (!h->is_cn) {
        m.ada_final = h->buf<float>("adaf");
        m.nfw = h->w<float>("fin.nw"); m.nfb = h->w<float>("fin.nb");
        m.mod_final = h->buf<float>("modf");
    }
    m.n = n; m.nblk = nblk; m.D = D;
    launch_mod_finalize(m, st);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(EZDIT_E_HIP, "time-path launch failed: %s", hipGetErrorString(e));
    }
    if (h->zfuse_usable()) {
        // LayerNorm algebra tables (GemmArgs.z*): G' = g W^T, C' = c W^T (+ bias) of the LayerNorm in front of the fused QKV GEMM (norm1,
        // modulated: per slot), of the GEGLU GEMM (norm3, modulated; C' carries mlp.net.0.proj.bias) and of cross-attention's to_q (norm2,
        // static).  Small bf16 GEMMs on exact hi + lo splits of the fp32 vectors; once per call.
        Ctx c{h, st};
        const long mod_slot = (long)nblk * 6 * D;
        bf16_t* zA = h->buf<bf16_t>("zA");
        float* ztmp = h->buf<float>("ztmp");
        const int I2 = 2 * h->I, N3 = 3 * D;
        for (int b = 0; b < nblk; ++b) {
            const BlkW& w = h->blk[b];
            const float* modb = h->p.mod + (long)b * 6 * D;
            launch_z_hilo(modb + 0 * D, modb + 1 * D, mod_slot, zA, h->ldD, n, D, st);
            gemm(c, zA, h->ldD, w.wqkv, nullptr, ztmp, N3, 4 * n, N3, EPI_F32, 25);
            launch_z_combine(ztmp, N3, nullptr, h->p.zt_qkv + (long)b * 2 * N3, h->p.zt_qkv + (long)b * 2 * N3 + N3, (long)nblk * 2 * N3, n, N3, st);
            launch_z_hilo(modb + 3 * D, modb + 4 * D, mod_slot, zA, h->ldD, n, D, st);
            gemm(c, zA, h->ldD, w.w1, nullptr, ztmp, I2, 4 * n, I2, EPI_F32, 25);
            launch_z_combine(ztmp, I2, w.b1, h->p.zt_geglu + (long)b * 2 * I2, h->p.zt_geglu + (long)b * 2 * I2 + I2, (long)nblk * 2 * I2, n, I2, st);
            launch_z_hilo(w.n2w, w.n2b, 0, zA, h->ldD, 1, D, st);
            gemm(c, zA, h->ldD, w.wq2, nullptr, ztmp, D, 4, D, EPI_F32, 25);
            launch_z_combine(ztmp, D, nullptr, h->p.zt_q2 + (long)b * 2 * D, h->p.zt_q2 + (long)b * 2 * D + D, 0, 1, D, st);
            if (b > h->nhalf && h->skip_z_usable()) {   // skip_norm (static, over the 2 D columns of [x | skip]) in front of skip_linear; C' carries skip_linear.bias
                const int j = b - h->nhalf - 1;
                launch_z_hilo(w.snw, w.snb, 0, zA, h->ld2D, 1, 2 * D, st);
                gemm(c, zA, h->ld2D, w.wskip, nullptr, ztmp, D, 4, D, EPI_F32, 25);
                launch_z_combine(ztmp, D, w.bskip, h->p.zt_skip + (long)j * 2 * D, h->p.zt_skip + (long)j * 2 * D + D, 0, 1, D, st);
            }
        }
        const hipError_t e = hipGetLastError();
        if (c.bad() || e != hipSuccess) return fail(EZDIT_E_HIP, "LayerNorm-algebra table launch failed: %s", hipGetErrorString(e));
    }
    h->steps_done = 0;
    h->ts_ready = true;
    h->z_tables_ready = h->zfuse_usable();
    h->skip_tables_ready = h->z_tables_ready && h->skip_z_usable();
    h->n_ts = n;
    h->per_row = per_row;
    return EZDIT_OK;
}

int ezdit_set_step(ezdit_handle* h, int step, ezdit_stream stream) {
    if (!h || !h->ws) return fail(EZDIT_E_STATE, "bind workspace first");
    if (step < 0 || step >= h->n_slots) return fail(EZDIT_E_INVALID, "step %d out of range", step);
    launch_set_int(h->p.ints, step, 0, (hipStream_t)stream);
    h->steps_done = step;
    return EZDIT_OK;
}

// ------------------------------------------------------------------------------------------------------
// cn_scale multiplies the ControlNet residuals `cn` (conditioning_scale, controlnet.py:313): the fused sampler passes the
// attached ControlNet's scale, ezdit_forward passes 1 (the caller's residuals are already scaled, as DiTControlNet.forward returns them)
// cn_ready (nullable): event the residuals `cn` become valid at; waited for right before their first consumer, so a ControlNet
// forward on another stream overlaps the backbone's in-blocks and mid block (the two chains are independent until then,
// src/inference_controlnet.py:89-99 + udit.py:345-348).
static int forward_impl(ezdit_handle* h, const float* x, int in_ch, int x_rows, const float* gt, const uint8_t* gt_mask,
                        const float* const* cn, int n_cn, float cn_scale, float* out, hipStream_t st, hipEvent_t cn_ready = nullptr,
                        const int* cur_override = nullptr) {
    const bool cn_mode = h->is_cn;  // ControlNet: in-blocks only, then one zero-Linear per skip (controlnet.py:303-315)
    Ctx c{h, st};
    const WsPtrs& p = h->p;
    const int D = h->D, M = h->M, nblk = h->nblk, nhalf = h->nhalf, Mp = h->Mp;
    // device step counter that selects the modulation slot.  A ControlNet attached to the fused sampler reads the BACKBONE's counter
    // (cur_override): only that one is advanced by the CFG / DDIM kernel, and the reference evaluates the ControlNet at the current t
    // every step (src/inference_controlnet.py:92-96)
    const int* cur = cur_override ? cur_override : p.ints;
    const int* row_slot = h->per_row ? p.ints + 16 : nullptr;
    c.cur = cur; c.row_slot = row_slot;
    float* hA = p.h;
    float* skips = p.skips;
    bf16_t* u = p.u;
    float* part = p.part;
    const float* mod = p.mod;
    const long mod_slot = (long)nblk * 6 * D;
    h->launches = 0;
    (void)hipGetLastError();   // a stale error of an earlier, unrelated call must not be blamed on this one
#define STOPCHK() do { if (c.bad()) return c.result(); if (h->debug_stop > 0 && h->launches >= h->debug_stop) return EZDIT_OK; } while (0)

    // A4 + A5: input assembly and patch embed (Conv1d k=1 == per-token Linear)
    AssembleArgs as;
    as.x = x; as.x_rows = x_rows; as.in_ch = in_ch;
    as.gt = gt; as.gt_mask = gt_mask; as.mask_embed = cn_mode ? h->ext_mask_embed : h->w_mask_embed;
    as.out = p.ape; as.ldo = h->ldPE;
    as.B = h->B; as.C = h->C; as.L = h->L;
    STOPCHK();
    launch_assemble(as, st);
    c.launched("k_assemble");
    const float* part_src = part;
    bool u_is_z = false;   // `u` holds A' = bf16(h g) + partial statistics (LayerNorm algebra) instead of a finished LayerNorm
    bool* u_is_z_ptr = &u_is_z;
    auto make_row = [&](int mode, const float* h_in, float* h_out, int nsplit, const float* bias, const float* gate,
                        long gate_stride, const float* lg, const float* lc, long ln_stride, const float* skip, const float* cnp,
                        int ld_u) {
        RowArgs r;
        memset(&r, 0, sizeof r);
        r.h_in = h_in; r.h_out = h_out;
        r.part = part_src; r.nsplit = nsplit; r.part_stride = (long)Mp * D; r.ld_part = D;
        r.part_bf16 = (part_src == part) ? 1 : 0;   // the split-K slabs are bf16
        r.cn_scale = cn_scale;
        r.bias = bias; r.gate = gate; r.gate_slot_stride = gate_stride; r.mode = mode;
        r.ln_g = lg; r.ln_c = lc; r.ln_slot_stride = ln_stride;
        r.skip = skip; r.cn = cnp;
        r.u = lg ? (skip ? p.ucat : u) : nullptr; r.ld_u = ld_u;
        r.M = M; r.D = D; r.L = h->L;
        r.cur_step = cur; r.row_slot = row_slot; r.wt = h->wt();
        r.variant = h->opt_row_variant;
        r.affine = h->opt_row_affine && M <= 1024;
        return r;
    };
    auto row = [&](int mode, const float* h_in, float* h_out, int nsplit, const float* bias, const float* gate,
                   long gate_stride, const float* lg, const float* lc, long ln_stride, const float* skip, const float* cnp,
                   int ld_u) {
        const RowArgs r = make_row(mode, h_in, h_out, nsplit, bias, gate, gate_stride, lg, lc, ln_stride, skip, cnp, ld_u);
        launch_row(r, st);
        c.launched("k_row");
        if (lg && !skip) u_is_z_ptr[0] = false;   // `u` now holds a finished LayerNorm
    };
    // residual GEMM + its row operator: out = rowop(A . W^T as split-K slabs): two launches (the one-launch form with an in-launch
    // hand-off measured slower in round 2 and was removed)
    auto resid = [&](const bf16_t* A, int lda, const WRef& w, int mode, const float* h_in, float* h_out, const float* bias, const float* gate,
                     long gate_stride, const float* lg, const float* lc, long ln_stride, const float* skip, const float* cnp, int ld_u) {
        const int s2 = gemm_partial(c, A, lda, w, M, D);
        if (c.bad() || (h->debug_stop > 0 && h->launches >= h->debug_stop)) return;
        row(mode, h_in, h_out, s2, bias, gate, gate_stride, lg, lc, ln_stride, skip, cnp, ld_u);
    };
    auto modv = [&](int blk, int which) { return mod + ((long)blk * 6 + which) * D; };
    // LayerNorm algebra (opt_zfuse): un-split residual projection whose epilogue produces h_new, its partial LayerNorm statistics and
    // A' = bf16(h_new * zg) for the next GEMM; the consumer finishes the LayerNorm.  u_is_z tells the next consumer what `u` holds.
    const int qkv_mode = h->qkv_mode();   // 2: fused QKV GEMM (ping-pong kernel), 0: fp32 projection + k_headnorm
    const bool zf = h->z_tables_ready && h->zfuse_usable();
    // eb0 / enb: the launch covers the batch elements [eb0, eb0 + enb) only (enb < 0: all); dual_blk >= 0: DUAL form for block dual_blk (GemmArgs.zd)
    struct ZExtra {   // the skip path's forms of the un-split residual projection (GemmArgs: COPY2 / ZIN); whole-batch launches only
        bf16_t* zu = nullptr; int ld_zu = 0;          // operand buffer instead of `u`
        float2* zstat_out = nullptr;                  // statistics instead of the shared buffer
        bf16_t* zu2 = nullptr; int ld_zu2 = 0; const float* zg2 = nullptr;   // COPY2
        const float2 *zin = nullptr, *zin2 = nullptr; const float* zG = nullptr; int zparts = 0, zD = 0;   // ZIN
    };
    const float2* zstat_next = nullptr;   // where the last producer put its statistics when not in the shared buffer (handed to the next consumer: Ctx.zstat)
    auto resid_z = [&](const bf16_t* A, int lda, const WRef& w, const float* h_in, float* h_out, const float* bias, const float* gate, long gate_stride,
                       const float* zg, long zg_stride, const char* what, int eb0 = 0, int enb = -1, int dual_blk = -1, const ZExtra* zx = nullptr) {
        const long r0 = (long)eb0 * h->L;
        const int Ms = enb < 0 ? M : enb * h->L;
        GemmArgs g;
        memset(&g, 0, sizeof g);
        g.A = A + r0 * lda; g.lda = lda; g.W = w.W; g.ldw = w.ld; g.wrows = w.rows; g.bias = bias;
        g.out = h_out ? h_out + r0 * D : nullptr; g.ldo = D; g.M = Ms; g.N = D; g.K = w.ld; g.splitk = 1; g.epi = EPI_RESID; g.tile = h->ztile();
        // few rows (the cross-attention-out projection over one prompt's conditional rows, M = 500): 48-row tiles leave 11 x 12 = 132 workgroups for 256 CUs;
        // 32 x 96 tiles (id 72) give 16 x 12 = 192 with 11 % fewer operand bytes each: 3.968 -> 3.951 ms per step, bit-identical (profiles/r05_experiments.txt)
        if (g.tile == ezdit_handle::kZTile && dual_blk < 0 && !(zx && (zx->zu2 || zx->zin2)) && Ms <= 672 && g.K <= 2 * D) g.tile = 72;
        g.xcd_map = 1; g.wt = h->wt();
        g.resid = h_in ? h_in + r0 * D : nullptr; g.ldr = D; g.gate = gate; g.gate_slot_stride = gate_stride;
        g.cur_step = cur; g.row_slot = row_slot ? row_slot + eb0 : nullptr; g.rows_per_b = h->L;
        g.zu = u + r0 * h->ldD; g.ld_zu = h->ldD; g.zg = zg; g.zg_slot_stride = zg_stride; g.zstat_out = p.zstat + r0; g.zs_stride = h->Mp;
        if (dual_blk >= 0) {
            g.zd = p.zd + (size_t)dual_blk * h->B * D; g.zd_stride = D;
            g.zg2 = modv(dual_blk, 3); g.zg2_slot_stride = mod_slot;
            g.act_row0 = h->act_b0 * h->L; g.act_row1 = h->act_b1 * h->L;
        }
        zstat_next = nullptr;
        if (zx) {
            if (zx->zu) { g.zu = zx->zu; g.ld_zu = zx->ld_zu; }
            if (zx->zstat_out) { g.zstat_out = zx->zstat_out; zstat_next = zx->zstat_out; }
            g.zu2 = zx->zu2; g.ld_zu2 = zx->ld_zu2;
            if (zx->zu2) { g.zg2 = zx->zg2; g.zg2_slot_stride = 0; }
            if (zx->zin2) { g.zstat_in = zx->zin; g.zstat_in2 = zx->zin2; g.zG = zx->zG; g.zparts = zx->zparts; g.zD = zx->zD; g.zeps = 1e-5f; }
        }
        g.ts = c.stamps(); g.ts_cap = g_gemm_ts_cap;
        c.launched(what, launch_gemm(g, st));
        u_is_z = true;
    };
    // LN_2D([x | skip]) -> skip_linear by the LayerNorm algebra (opt_skip_z): not with ControlNet residuals (they change the skips)
    const bool skipz = zf && h->skip_tables_ready && h->skip_z_usable() && !(cn && n_cn > 0);
    const int zsp = (D + h->zwidth() - 1) / h->zwidth();                      // statistics parts of a D-wide producer
    auto ucat_of = [&](int j) { return p.ucat_z + (size_t)j * Mp * h->ld2D; };   // operand [x | skip] of out-block j
    auto zskip_of = [&](int i) { return p.zstat_skip + (size_t)i * Z_MAXP * Mp; };   // statistics of skip i
    // single-key shortcut (opt_xkey1): cross-attention + its out-projection cover the batch elements [xb0, xb0 + xnb) only
    const bool x1 = zf && h->opt_xkey1 && h->xkey1;
    const int xb0 = x1 ? h->act_b0 : 0, xnb = x1 ? h->act_b1 - h->act_b0 : h->B;

    // LN1 of block 0 on the patch embedding (ControlNet: x = patch_embed(x) + controlnet_pre(condition) first, :263-266)
    STOPCHK();
    if (zf && !cn_mode) {
        // LayerNorm algebra: the patch embed itself is the producer for block 0's norm1 (no gate, no residual: h = acc + bias, statistics, bf16(h g)) -- no row-kernel launch
        resid_z(p.ape, h->ldPE, h->w_pe, nullptr, hA, h->b_pe, nullptr, 0, modv(0, 0), mod_slot, "k_gemm (un-split residual: patch embed)");
    } else {
    gemm(c, p.ape, h->ldPE, h->w_pe, h->b_pe, hA, D, M, D, EPI_F32, M <= 2048 ? ezdit_handle::kTilePE : tile_for(h, M, false));
    STOPCHK();
    if (cn_mode) {
        part_src = p.cembed;
        row(1, hA, hA, 1, nullptr, nullptr, 0, modv(0, 0), modv(0, 1), mod_slot, nullptr, nullptr, h->ldD);
        part_src = part;
    } else {
        row(0, hA, nullptr, 0, nullptr, nullptr, 0, modv(0, 0), modv(0, 1), mod_slot, nullptr, nullptr, h->ldD);
    }
    }
    const float* hcur = hA;

    for (int b = 0; b < nblk; ++b) {
        const BlkW& w = h->blk[b];
        const bool is_in = b < nhalf, is_out = b > nhalf;
        if (is_out) {
            // u holds LN_2D([x | skip]) -> skip_linear (blocks.py:124-128)
            STOPCHK();
            if (skipz) {
                const int j = b - nhalf - 1;
                ZExtra zx;
                zx.zin = p.zstat + (size_t)Z_MAXP * Mp; zx.zin2 = zskip_of(nhalf - 1 - j); zx.zG = p.zt_skip + (long)j * 2 * D; zx.zparts = zsp; zx.zD = 2 * D;
                resid_z(ucat_of(j), h->ld2D, w.wskip, nullptr, hA, zx.zG + D, nullptr, 0, modv(b, 0), mod_slot, "k_gemm (un-split residual: skip_linear ZIN)", 0, -1, -1, &zx);
            } else if (zf) resid_z(p.ucat, h->ld2D, w.wskip, nullptr, hA, w.bskip, nullptr, 0, modv(b, 0), mod_slot, "k_gemm (un-split residual: skip_linear)");
            else resid(p.ucat, h->ld2D, w.wskip, 2, nullptr, hA, w.bskip, nullptr, 0, modv(b, 0), modv(b, 1), mod_slot, nullptr, nullptr, h->ldD);
            hcur = hA;
        }
        // ---- self attention (blocks.py:136-141) ----
        STOPCHK();
        HeadNormArgs hn;
        memset(&hn, 0, sizeof hn);
        hn.x = p.qkv; hn.ldx = 3 * D;
        hn.q_col = 0; hn.k_col = D; hn.v_col = 2 * D;
        hn.qn_w = w.aqnw; hn.qn_b = w.aqnb;
        hn.kn_w = w.aknw; hn.kn_b = w.aknb;
        hn.rope_cos = p.rope_cos; hn.rope_sin = p.rope_sin;
        hn.q = p.q; hn.k = p.k; hn.v = p.v;
        hn.B = h->B; hn.H = h->H; hn.L = h->L; hn.Lp = h->Lp; hn.dh = h->dh;
        if (qkv_mode) {
            // head-norm + RoPE + the attention layouts inside the projection GEMM (128 x two-head tiles): no fp32 q|k|v round trip, one launch less
            hn.perm = 1;   // (qkv_mode == 2 <=> the weights were packed with EZDIT_T_QKROPE)
            c.hn = &hn;
            if (u_is_z) { c.zG = p.zt_qkv + (long)b * 2 * 3 * D; c.zC = c.zG + 3 * D; c.zt_stride = (long)nblk * 2 * 3 * D; c.zstat = zstat_next; }
            gemm(c, u, h->ldD, w.wqkv, nullptr, nullptr, 0, M, 3 * D, EPI_QKV, h->qkv_tile());
        } else {
            gemm(c, u, h->ldD, w.wqkv, nullptr, p.qkv, 3 * D, M, 3 * D, EPI_F32, tile_for(h, M, false));
            STOPCHK();
            launch_headnorm(hn, st);
            c.launched("k_headnorm");
        }
        AttnArgs at;
        memset(&at, 0, sizeof at);
        at.q = hn.q; at.k = hn.k; at.v = hn.v; at.kmask = nullptr;
        at.nkh = h->opt_attn_nkh; at.xcd_map = h->opt_attn_xcd; at.wt = h->wt();
        at.out = p.ao; at.ldo = h->ldD;
        at.B = h->B; at.H = h->H; at.Lq = h->L; at.Lk = h->L; at.Lqp = h->Lp; at.Lkp = h->Lp; at.dh = h->dh;
        STOPCHK();
        at.ts = c.stamps(); at.ts_cap = g_gemm_ts_cap;
        c.launched("k_attn (self)", launch_attention(at, st));
        STOPCHK();
        // x += (1 - gate_msa) * (proj + bias); then norm2 (plain affine LN) for cross-attention q
        if (zf) {
            resid_z(at.out, h->ldD, w.wo, hcur, hA, w.bo, modv(b, 2), mod_slot, w.n2w, 0, x1 ? "k_gemm (un-split residual: attention-out DUAL)" : "k_gemm (un-split residual: attention-out)", 0, -1, x1 ? b : -1);
        } else {
            resid(at.out, h->ldD, w.wo, 1, hcur, hA, w.bo, modv(b, 2), mod_slot, w.n2w, w.n2b, 0, nullptr, nullptr, h->ldD);
        }
        hcur = hA;
        // ---- cross attention (blocks.py:147-151) ----
        STOPCHK();
        // one prompt: the cross-attention kernel also computes its own q = LN_head(u . Wq^T) (8-wave form, Lcp % 128 == 0)
        // (x1: over the batch elements [xb0, xb0 + xnb) only -- the others are single-key rows served by the attention-out projection above)
        const bool fuse_q2 = h->q2_fused(xnb);   // four prompts per GPU with the shortcut: 4 x 16 x 8 = 512 workgroups -> fused (10.38 -> 10.22 ms per step of 4)
        const long xr0 = (long)xb0 * h->L;          // first row of the sub-range
        const int xM = xnb * h->L;
        if (xnb > 0) {
            memset(&hn, 0, sizeof hn);
            hn.x = p.qkv; hn.ldx = D;
            hn.q_col = 0; hn.k_col = -1; hn.v_col = -1;
            hn.qn_w = w.cqnw; hn.qn_b = w.cqnb;
            hn.q = p.q + (size_t)xb0 * h->H * h->Lp * h->DQK;
            hn.B = xnb; hn.H = h->H; hn.L = h->L; hn.Lp = h->Lp; hn.dh = h->dh;
            at.xu = nullptr; at.nkh = h->opt_attn_nkh;
            at.B = xnb; at.b0 = xb0;
            if (fuse_q2) {
                at.xu = u; at.ldu = h->ldD; at.xw = w.wq2.W; at.ldw = w.wq2.ld;
                at.xw_rows = w.wq2.rows; at.xK = at.ldw;
                at.qn_w = hn.qn_w; at.qn_b = hn.qn_b; at.nkh = 4; at.xk2 = h->opt_attn_xk2; at.qtile = h->opt_attn_qtile;
#ifdef EZ_DIAG
                if (!u_is_z && h->opt_zfake && (D + h->zwidth() - 1) / h->zwidth() <= Z_MAXP) {
                    at.zw = h->zwidth(); at.zstat_in = h->buf<float2>("zneutral"); at.zs_stride = h->Mp; at.zparts = (D + at.zw - 1) / at.zw; at.zD = D; at.zeps = 1e-5f;
                    at.zG = h->buf<float>("zzeros"); at.zC = at.zG;
                }
#endif
                if (u_is_z) {
                    at.zw = h->zwidth(); at.zstat_in = p.zstat; at.zs_stride = h->Mp; at.zparts = (D + at.zw - 1) / at.zw; at.zD = D; at.zeps = 1e-5f;
                    at.zG = p.zt_q2 + (long)b * 2 * D; at.zC = at.zG + D;
                }
            } else if (qkv_mode == 2 && h->opt_q2_pp) {
                // batched prompts: the q projection on the ping-pong kernel with the fused-QKV epilogue restricted to its q part (N = D, no RoPE):
                // per-head LayerNorm and the bf16 attention layout straight out of the GEMM -- no fp32 q round trip, no 128 x 64 tile at M = 4000
                // (k_gemm<128,64> + normalisation inside k_attn: 30 us; this: one round of 256 workgroups).  LayerNorm-algebra capable (zt_q2 is static)
                c.hn = &hn;
                if (u_is_z) { c.zG = p.zt_q2 + (long)b * 2 * D; c.zC = c.zG + D; c.zt_stride = 0; c.zrow0 = (int)xr0; c.zb0 = xb0; }
                gemm(c, u + xr0 * h->ldD, h->ldD, w.wq2, nullptr, nullptr, 0, xM, D, EPI_QKV, 61);
                at.q = p.q;   // (launch_attention advances it by b0)
            } else {
                gemm(c, u + xr0 * h->ldD, h->ldD, w.wq2, nullptr, p.qkv + xr0 * D, D, xM, D, EPI_F32, tile_for(h, xM, false));
                at.q_raw = p.qkv; at.ld_qraw = D; at.qn_w = hn.qn_w; at.qn_b = hn.qn_b;   // the cross-attention kernel normalises q itself
            }
            at.q = p.q;
            at.k = p.kc + (size_t)b * h->B * h->H * h->Lcp * h->DQK;
            at.v = p.vc + (size_t)b * h->B * h->H * h->Lcp * h->DV;
            at.kmask = p.kmask;
            at.Lk = h->Lc; at.Lkp = h->Lcp;
            STOPCHK();
            at.ts = c.stamps(); at.ts_cap = g_gemm_ts_cap;
            c.launched("k_attn (cross)", launch_attention(at, st));
            STOPCHK();
            if (zf) {
                at.zstat_in = nullptr;
                resid_z(at.out, h->ldD, w.wo2, hA, hA, w.bo2, nullptr, 0, modv(b, 3), mod_slot, "k_gemm (un-split residual: cross-out)", xb0, x1 ? xnb : -1);
            } else {
                resid(at.out, h->ldD, w.wo2, 1, hA, hA, w.bo2, nullptr, 0, modv(b, 3), modv(b, 4), mod_slot, nullptr, nullptr, h->ldD);
            }
            at.B = h->B; at.b0 = 0;
        }
        // ---- GEGLU MLP (blocks.py:154-156) ----
        STOPCHK();
        if (u_is_z) { c.zG = p.zt_geglu + (long)b * 2 * 2 * h->I; c.zC = c.zG + 2 * h->I; c.zt_stride = (long)nblk * 2 * 2 * h->I; }
        gemm(c, u, h->ldD, w.w1, w.b1, p.act, h->ldI, M, 2 * h->I, EPI_GEGLU, h->geglu_tile());   // 128 x 288 ping-pong kernel / 128 x 144 co-resident kernel / round-1 lockstep kernel
        STOPCHK();
        // x += (1 - gate_mlp) * (mlp + bias); the LN that follows belongs to the NEXT consumer
        const float* b2 = w.b2;
        if (cn_mode && b == nblk - 1) {
            resid(p.act, h->ldI, w.w2, 1, hA, skips + (size_t)b * Mp * D, b2, modv(b, 5), mod_slot, nullptr, nullptr, 0, nullptr, nullptr, h->ldD);
        } else if (b == nblk - 1) {
            const float* mf = p.modf;
            resid(p.act, h->ldI, w.w2, 1, hA, nullptr, b2, modv(b, 5), mod_slot, mf, mf + D, 2L * D, nullptr, nullptr, h->ldD);
        } else if (b + 1 > nhalf) {
            const int j = b + 1 - nhalf - 1;  // out block index of the consumer
            const float* skip = skips + (size_t)(nhalf - 1 - j) * Mp * D;
            const float* cnp = (cn && n_cn > 0) ? cn[n_cn - 1 - j] : nullptr;
            if (cnp && cn_ready) { (void)hipStreamWaitEvent(st, cn_ready, 0); cn_ready = nullptr; }   // join the ControlNet stream
            if (skipz) {   // un-split: x_new only lives on as bf16(x_new g[:D]) in the left half of the out-block's operand, statistics in the second part range of the shared buffer
                ZExtra zx;
                zx.zu = ucat_of(j); zx.ld_zu = h->ld2D; zx.zstat_out = p.zstat + (size_t)Z_MAXP * Mp;
                resid_z(p.act, h->ldI, w.w2, hA, nullptr, b2, modv(b, 5), mod_slot, h->blk[b + 1].snw, 0, "k_gemm (un-split residual: MLP-out -> [x | skip])", 0, -1, -1, &zx);
            } else
            resid(p.act, h->ldI, w.w2, 1, hA, nullptr, b2, modv(b, 5), mod_slot, h->blk[b + 1].snw, h->blk[b + 1].snb, 0, skip, cnp, h->ld2D);
        } else {
            float* dst = is_in ? skips + (size_t)b * Mp * D : hA;
            if (zf && skipz && is_in) {   // COPY2: + the skip half of the matching out-block's operand, statistics kept until that out-block
                const int j = nhalf - 1 - b;
                ZExtra zx;
                zx.zstat_out = zskip_of(b); zx.zu2 = ucat_of(j) + D; zx.ld_zu2 = h->ld2D; zx.zg2 = h->blk[nhalf + 1 + j].snw + D;
                resid_z(p.act, h->ldI, w.w2, hA, dst, b2, modv(b, 5), mod_slot, modv(b + 1, 0), mod_slot, "k_gemm (un-split residual: MLP-out COPY2)", 0, -1, -1, &zx);
            } else if (zf) resid_z(p.act, h->ldI, w.w2, hA, dst, b2, modv(b, 5), mod_slot, modv(b + 1, 0), mod_slot, "k_gemm (un-split residual: MLP-out)");
            else resid(p.act, h->ldI, w.w2, 1, hA, dst, b2, modv(b, 5), mod_slot, modv(b + 1, 0), modv(b + 1, 1), mod_slot, nullptr, nullptr, h->ldD);
            hcur = dst;
        }
    }
    if (cn_mode) {
        // controlnet_skips[i] = zero_Linear_i(skip_i) (* conditioning_scale, applied by the consumer)  controlnet.py:311-313
        for (int i = 0; i < nblk; ++i) {
            launch_cast_bf16(skips + (size_t)i * Mp * D, D, p.skipbf, h->ldD, M, D, 0, st);
            c.launched("k_cast_bf16");
            gemm(c, p.skipbf, h->ldD, h->blk[i].zw, h->blk[i].zb, p.cnres + (size_t)i * Mp * D, D, M, D, EPI_F32, tile_for(h, M, false));
        }
        return c.result();
    }
    // A18 FinalBlock: u = LN(x)*(1+scale)+shift -> Linear(D->C) -> transpose -> Conv1d(C,C,3,pad 1)
    STOPCHK();
    gemm(c, u, h->ldD, h->w_fin, h->b_fin, p.y, h->C, M, h->C, EPI_F32, M <= 2048 ? ezdit_handle::kTileFin : tile_for(h, M, false));
    FinalConvArgs fc;
    fc.y = p.y; fc.ldy = h->C;
    fc.w = h->w_fin_cw; fc.b = h->w_fin_cb;
    fc.out = out; fc.B = h->B; fc.C = h->C; fc.L = h->L;
    STOPCHK();
    launch_final_conv(fc, st);
    c.launched("k_final_conv");
    return c.result();
}

int ezdit_forward(ezdit_handle* h, const float* x, int in_ch, int x_rows, const float* gt, const uint8_t* gt_mask,
                  const float* const* cn_skips, int n_cn, float* out, ezdit_stream stream) {
    if (!h || !x || !out) return fail(EZDIT_E_INVALID, "null argument");
    if (h->is_cn) return fail(EZDIT_E_INVALID, "this handle is a ControlNet: use ezdit_controlnet_forward");
    if (!h->wblob || !h->ws) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    if (!h->ctx_ready) return fail(EZDIT_E_STATE, "ezdit_prepare_context has not run for this workspace");
    if (!h->ts_ready) return fail(EZDIT_E_STATE, "ezdit_prepare_timesteps has not run for this workspace");
    if (in_ch != h->C && in_ch != h->Cin) return fail(EZDIT_E_INVALID, "in_ch %d: expected %d or %d", in_ch, h->C, h->Cin);
    if (in_ch == h->C && (x_rows <= 0 || h->B % x_rows)) return fail(EZDIT_E_INVALID, "x_rows %d does not divide B %d", x_rows, h->B);
    if ((gt == nullptr) != (gt_mask == nullptr)) return fail(EZDIT_E_INVALID, "gt and gt_mask must be given together");
    if (n_cn != 0 && n_cn != h->nhalf) return fail(EZDIT_E_INVALID, "n_cn %d: expected 0 or %d", n_cn, h->nhalf);
    // the caller's residuals are already multiplied by conditioning_scale (DiTControlNet.forward returns them scaled,
    // controlnet.py:313); a scale left on the handle by the fused sampler must not be applied a second time
    return forward_impl(h, x, in_ch, x_rows, gt, gt_mask, cn_skips, n_cn, h->fwd_cn_scale, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------
int ezdit_prepare_condition(ezdit_handle* h, const float* cond, int Lcond, ezdit_stream stream) {
    if (!h || !cond) return fail(EZDIT_E_INVALID, "null argument");
    if (!h->is_cn) return fail(EZDIT_E_INVALID, "not a ControlNet handle");
    if (!h->wblob || !h->ws) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    if (Lcond != 2 * h->L) return fail(EZDIT_E_INVALID, "condition length %d != 2 * L (%d): the embed has one stride-2 conv", Lcond, 2 * h->L);
    hipStream_t st = (hipStream_t)stream;
    Conv1dArgs a;
    memset(&a, 0, sizeof a);
    // conv_in: Conv1d(cond_in, c0, 1)                                               controlnet.py:15,66
    a.x = cond; a.w = h->w<float>("cn.cin.w"); a.b = h->w<float>("cn.cin.b"); a.out = h->buf<float>("cn_e0");
    a.B = h->B; a.Cin = h->cfg.cond_in; a.cin_valid = a.Cin; a.Cout = h->c0; a.Lin = Lcond; a.Lout = Lcond; a.ksize = 1; a.stride = 1; a.pad = 0;
    launch_conv1d(a, st);
    // eval: no position is masked, the appended mask channel is all zeros (:68-74) -> channel c0 is an implicit zero input
    a.x = h->buf<float>("cn_e0"); a.w = h->w<float>("cn.c0.w"); a.b = h->w<float>("cn.c0.b"); a.out = h->buf<float>("cn_e1");
    a.Cin = h->c0m; a.cin_valid = h->c0; a.Cout = h->c0m; a.ksize = 3; a.pad = 1; a.act = 1;
    launch_conv1d(a, st);
    a.x = h->buf<float>("cn_e1"); a.w = h->w<float>("cn.c1.w"); a.b = h->w<float>("cn.c1.b"); a.out = h->buf<float>("cn_e2");
    a.Cin = h->c0m; a.cin_valid = h->c0m; a.Cout = h->c1; a.Lout = h->L; a.stride = 2;
    launch_conv1d(a, st);
    // conv_out: Conv1d(c1, D, 1) then transpose to [B, L, D]                         :37,79-82
    a.x = h->buf<float>("cn_e2"); a.w = h->w<float>("cn.cout.w"); a.b = h->w<float>("cn.cout.b"); a.out = h->buf<float>("cembed");
    a.Cin = h->c1; a.cin_valid = h->c1; a.Cout = h->D; a.Lin = h->L; a.Lout = h->L; a.ksize = 1; a.stride = 1; a.pad = 0; a.act = 0;
    a.out_token_major = 1;
    launch_conv1d(a, st);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(EZDIT_E_HIP, "condition-embed launch failed: %s", hipGetErrorString(e));
    }
    h->cond_ready = true;
    return EZDIT_OK;
}

int ezdit_controlnet_forward(ezdit_handle* h, const float* x, int in_ch, int x_rows, const float* gt, const uint8_t* gt_mask,
                             const float* mask_embed, ezdit_stream stream) {
    if (!h || !x) return fail(EZDIT_E_INVALID, "null argument");
    if (!h->is_cn) return fail(EZDIT_E_INVALID, "not a ControlNet handle");
    if (!h->wblob || !h->ws) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    if (!h->ctx_ready || !h->ts_ready || !h->cond_ready) return fail(EZDIT_E_STATE, "prepare context, timesteps and condition first");
    if (in_ch != h->C && in_ch != h->Cin) return fail(EZDIT_E_INVALID, "in_ch %d: expected %d or %d", in_ch, h->C, h->Cin);
    if (in_ch == h->C && !mask_embed) return fail(EZDIT_E_INVALID, "in_ch = C needs the backbone's mask_embed");
    if (in_ch == h->C && (x_rows <= 0 || h->B % x_rows)) return fail(EZDIT_E_INVALID, "x_rows %d does not divide B %d", x_rows, h->B);
    h->ext_mask_embed = mask_embed;
    return forward_impl(h, x, in_ch, x_rows, gt, gt_mask, nullptr, 0, 1.0f, nullptr, (hipStream_t)stream);
}

int ezdit_controlnet_residuals(ezdit_handle* h, const float** out, int n) {
    if (!h || !out || !h->is_cn || !h->ws) return fail(EZDIT_E_INVALID, "need a bound ControlNet handle");
    if (n != h->nhalf) return fail(EZDIT_E_INVALID, "n %d != depth/2 %d", n, h->nhalf);
    for (int i = 0; i < n; ++i) out[i] = h->buf<float>("cnres") + (size_t)i * h->Mp * h->D;
    return EZDIT_OK;
}

int ezdit_sampler_attach_controlnet(ezdit_handle* h, ezdit_handle* cn, float conditioning_scale) {
    if (!h || h->is_cn) return fail(EZDIT_E_INVALID, "first argument must be a backbone handle");
    if (cn && !cn->is_cn) return fail(EZDIT_E_INVALID, "second argument must be a ControlNet handle");
    if (h->cn && h->cn != cn) {
        auto& v = h->cn->cn_users;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == h) { v.erase(v.begin() + i); break; }
    }
    if (cn && h->cn != cn) cn->cn_users.push_back(h);
    h->cn = cn;
    h->cn_scale = cn ? conditioning_scale : 1.0f;
    if (cn && !h->cn_stream) {   // side stream + fork / join events of the overlapped ControlNet branch: created here, never inside a stream capture
        HIPCHK(hipStreamCreateWithFlags(&h->cn_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->cn_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->cn_join, hipEventDisableTiming));
    }
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    return EZDIT_OK;
}

int ezdit_set_cn_scale(ezdit_handle* h, float scale) {
    if (!h) return fail(EZDIT_E_INVALID, "null handle");
    h->fwd_cn_scale = scale;
    return EZDIT_OK;
}

// ------------------------------------------------------------------------------------------------------
int ezdit_sampler_begin(ezdit_handle* h, float* latents, int P, const float* noise, const ezdit_ddim_coef* coefs, int n_steps,
                        float guidance_scale, float guidance_rescale, const float* gt, const uint8_t* gt_mask,
                        ezdit_stream stream) {
    if (!h || !latents || !coefs) return fail(EZDIT_E_INVALID, "null argument");
    if (!h->ws || !h->wblob) return fail(EZDIT_E_STATE, "bind weights and workspace first");
    if (n_steps <= 0 || n_steps > h->n_slots) return fail(EZDIT_E_INVALID, "n_steps %d > workspace slots %d", n_steps, h->n_slots);
    const int needB = guidance_scale > 0.f ? 2 * P : P;
    if (P <= 0 || needB != h->B) return fail(EZDIT_E_INVALID, "P=%d with guidance %g needs B=%d, workspace has B=%d", P, (double)guidance_scale, needB, h->B);
    if ((gt == nullptr) != (gt_mask == nullptr)) return fail(EZDIT_E_INVALID, "gt and gt_mask must be given together");
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> cf((size_t)n_steps * 8, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        cf[i * 8 + 0] = coefs[i].sa; cf[i * 8 + 1] = coefs[i].sb; cf[i * 8 + 2] = coefs[i].c_x0;
        cf[i * 8 + 3] = coefs[i].c_dir; cf[i * 8 + 4] = coefs[i].sigma;
    }
    HIPCHK(hipMemcpyAsync(h->buf<float>("coef"), cf.data(), cf.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    launch_set_int(h->p.ints, 0, 0, st);
    h->steps_done = 0;
    h->latents = latents; h->noise = noise; h->P = P; h->n_steps = n_steps;
    h->gscale = guidance_scale; h->grescale = guidance_rescale;
    h->s_gt = gt; h->s_gt_mask = gt_mask;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    return EZDIT_OK;
}

static int sampler_step(ezdit_handle* h, hipStream_t st) {
    float* pred = h->p.pred;
    const float* cnp[64];
    int n_cn = 0;
    hipEvent_t cn_ready = nullptr;
    if (h->cn) {  // src/inference_controlnet.py:89-99: ControlNet on the same assembled input, then the backbone with its skips
        ezdit_handle* cn = h->cn;
        if (cn->B != h->B || cn->L != h->L || cn->nhalf != h->nhalf || cn->D != h->D || !cn->ctx_ready || !cn->ts_ready || !cn->cond_ready)
            return fail(EZDIT_E_STATE, "attached ControlNet is not prepared for this shape (bind/context/timesteps/condition)");
        // the ControlNet indexes ITS modulation tables with the backbone's device step counter: its prepared schedule must cover the backbone's
        if (cn->n_ts < h->n_steps || cn->n_slots < h->n_steps || cn->per_row != h->per_row)
            return fail(EZDIT_E_STATE, "attached ControlNet was prepared for %d timesteps (per_row %d), the sampler runs %d (per_row %d)", cn->n_ts, cn->per_row, h->n_steps, h->per_row);
        cn->ext_mask_embed = h->w_mask_embed;
        hipStream_t cst = st;
        if (h->opt_cn_overlap && h->debug_stop == 0) {   // fork: the ControlNet chain runs next to the backbone's first half
            if (!h->cn_stream) return fail(EZDIT_E_STATE, "side stream missing: attach the ControlNet through ezdit_sampler_attach_controlnet");
            HIPCHK(hipEventRecord(h->cn_fork, st));
            HIPCHK(hipStreamWaitEvent(h->cn_stream, h->cn_fork, 0));
            cst = h->cn_stream;
        }
        int rc0 = forward_impl(cn, h->latents, h->C, h->P, h->s_gt, h->s_gt_mask, nullptr, 0, 1.0f, nullptr, cst, nullptr, h->p.ints);
        if (cst != st) {
            HIPCHK(hipEventRecord(h->cn_join, cst));
            cn_ready = h->cn_join;
            if (rc0) (void)hipStreamWaitEvent(st, cn_ready, 0);   // never leave the side stream unjoined (capture)
        }
        if (rc0) return rc0;
        n_cn = cn->nhalf;
        for (int i = 0; i < n_cn; ++i) cnp[i] = cn->p.cnres + (size_t)i * cn->Mp * cn->D;
    }
    int rc = forward_impl(h, h->latents, h->C, h->P, h->s_gt, h->s_gt_mask, n_cn ? cnp : nullptr, n_cn, h->cn_scale, pred, st, cn_ready);
    if (rc && cn_ready) (void)hipStreamWaitEvent(st, cn_ready, 0);
    if (rc) return rc;
    CfgDdimArgs a;
    a.pred = pred; a.latents = h->latents; a.noise = h->noise;
    a.coef = h->p.coef; a.cur_step = h->p.ints;
    for (float& v : a.hc) v = 0.f;
    a.guidance_scale = h->gscale; a.guidance_rescale = h->grescale;
    a.P = h->P; a.n = h->C * h->L;
    a.step_inc = h->p.ints; a.done = reinterpret_cast<unsigned*>(h->p.ints + 8);
    launch_cfg_ddim(a, h->p.cfgpart, st);   // its last kernel also advances the device step counter
    h->launches += (h->gscale > 0.f && h->grescale > 0.f) ? 2 : 1;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EZDIT_E_HIP, "launch of k_cfg_* failed: %s", hipGetErrorString(e));
    return EZDIT_OK;
}

// A2 + A3 + S of SURVEY.md section 8a as ONE stand-alone operator (src/inference.py:88-100): what a caller that keeps the
// reference's Python loop binds in place of `rescale_noise_cfg` + `scheduler.step`.
int ezdit_cfg_ddim_step(const float* pred, float* latents, const float* noise, const ezdit_ddim_coef* coef, float guidance_scale,
                        float guidance_rescale, int P, int n, float* scratch, ezdit_stream stream) {
    if (!pred || !latents || !coef || P < 1 || n < 2) return fail(EZDIT_E_INVALID, "bad argument");
    if (guidance_scale > 0.f && guidance_rescale > 0.f && !scratch) return fail(EZDIT_E_INVALID, "guidance_rescale needs %d scratch floats", P * 256);
    CfgDdimArgs a;
    a.pred = pred; a.latents = latents; a.noise = noise; a.coef = nullptr; a.cur_step = nullptr;
    a.hc[0] = coef->sa; a.hc[1] = coef->sb; a.hc[2] = coef->c_x0; a.hc[3] = coef->c_dir; a.hc[4] = coef->sigma;
    a.guidance_scale = guidance_scale; a.guidance_rescale = guidance_rescale;
    a.P = P; a.n = n;
    a.step_inc = nullptr; a.done = nullptr;
    launch_cfg_ddim(a, scratch, (hipStream_t)stream);
    return EZDIT_OK;
}

int ezdit_sampler_run(ezdit_handle* h, int n, int use_graph, ezdit_stream stream) {
    if (!h || !h->latents) return fail(EZDIT_E_STATE, "ezdit_sampler_begin first");
    if (!h->ctx_ready || !h->ts_ready) return fail(EZDIT_E_STATE, "prepare context and timesteps first");
    if (h->per_row) return fail(EZDIT_E_STATE, "sampler needs ezdit_prepare_timesteps(per_row = 0)");
    // the device-side step counter selects the modulation slot, the DDIM coefficients and the noise slice: running past the
    // prepared steps would index all three out of bounds
    if (n < 0 || h->steps_done + n > h->n_steps || h->steps_done + n > h->n_ts)
        return fail(EZDIT_E_STATE, "run of %d steps from step %d exceeds the %d prepared steps (ezdit_set_step rewinds)", n,
                    h->steps_done, h->n_steps < h->n_ts ? h->n_steps : h->n_ts);
    hipStream_t st = (hipStream_t)stream;
    if (!use_graph) {
        for (int i = 0; i < n; ++i) {
            int rc = sampler_step(h, st);
            if (rc) return rc;
            h->steps_done++;
        }
        return EZDIT_OK;
    }
    if (!h->graph_exec) {
        if (st == nullptr) return fail(EZDIT_E_INVALID, "graph capture needs a non-default stream");
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = sampler_step(h, st);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return fail(EZDIT_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        h->graph = g;
        HIPCHK(hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0));
    }
    for (int i = 0; i < n; ++i) { HIPCHK(hipGraphLaunch(h->graph_exec, st)); h->steps_done++; }
    return EZDIT_OK;
}

// ------------------------------------------------------------------------------------------------------
int ezdit_debug_gemm_timestamps(void* dev_buf, long capacity_workgroups) {
    g_gemm_ts = static_cast<unsigned long long*>(dev_buf);
    g_gemm_ts_cap = dev_buf ? capacity_workgroups : 0;
    return EZDIT_OK;
}

int ezdit_test_gemm(ezdit_handle* h, int variant, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                    int ldo, int M, int N, int K, int splitk, ezdit_stream stream) {
    if (K % 64) return fail(EZDIT_E_INVALID, "K=%d must be a multiple of 64", K);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.wrows = (int)rup(N, 128); g.bias = bias; g.out = out; g.ldo = ldo;
    g.M = M; g.N = N; g.K = K; g.splitk = splitk < 1 ? 1 : splitk;
    g.slab_stride = (long)rup(M, 128) * ldo;
    g.conv_cpb = 0; g.conv_tap_bytes = 0; g.resid = nullptr; g.ldr = 0; g.xcd_map = 1; g.part_bf16 = 0; g.wt = h ? h->wt() : 0; memset(&g.hn, 0, sizeof g.hn);
    g.gate = nullptr; g.gate_slot_stride = 0; g.cur_step = nullptr; g.row_slot = nullptr; g.rows_per_b = 1;
    g.debug = variant / 1000; variant %= 1000;   // 1000 * bits + v: bit 3 LDS-staged bf16 epilogue, bit 4 bf16 slabs, bit 6 LayerNorm-algebra consumer on neutral tables, bits 8.. k_gemm_pp ablation variant (EZ_ABLATE builds)
    if (g.debug & 8) g.epi_lds = 1;
    if (g.debug & 16) g.part_bf16 = 1;   // 16000 + v: bf16 split-K slabs
    g.ts = g_gemm_ts; g.ts_cap = g_gemm_ts_cap;
    g.epi = variant % 4; g.tile = variant / 4;   // variant = tile_config * 4 + epilogue
    if (g.debug & 64) {
        // 64000 + v: run the consumer side of the LayerNorm algebra on neutral tables (statistics of a zero-mean, unit-variance row, G' = 0,
        // C' = bias): same results as the plain epilogue up to the factor rsqrt(1 + 1e-5), same code path and memory traffic as the real thing
        static float2* zs = nullptr; static float* zg0 = nullptr; static float* zc = nullptr; static size_t cap_rows = 0, cap_n = 0;
        const size_t rows = (size_t)rup(M, 128), nn = (size_t)rup(N, 128);
        if (rows > cap_rows) { if (zs) (void)hipFree(zs); HIPCHK(hipMalloc(&zs, rows * 12 * sizeof(float2))); cap_rows = rows;
            std::vector<float2> hst(rows * 12, make_float2(0.f, 96.f)); HIPCHK(hipMemcpy(zs, hst.data(), hst.size() * sizeof(float2), hipMemcpyHostToDevice)); }
        if (nn > cap_n) { if (zg0) (void)hipFree(zg0); if (zc) (void)hipFree(zc); HIPCHK(hipMalloc(&zg0, nn * 4)); HIPCHK(hipMalloc(&zc, nn * 4)); cap_n = nn; HIPCHK(hipMemset(zg0, 0, nn * 4)); HIPCHK(hipMemset(zc, 0, nn * 4)); }
        // C' = the bias itself: no copy in the timed path
        g.zstat_in = zs; g.zs_stride = (long)cap_rows; g.zparts = 12; g.zD = 1152; g.zw = 96; g.zG = zg0; g.zC = bias ? bias : zc;
        g.zt_slot_stride = 0; g.zeps = 1e-5f;
        g.debug &= ~64;
    }
    if (g.epi > EPI_GEGLU || g.tile > 127) return fail(EZDIT_E_INVALID, "bad gemm variant %d", variant);
    if (g.epi != EPI_PARTIAL) g.splitk = 1;
    (void)hipGetLastError();
    if (launch_gemm(g, (hipStream_t)stream)) return fail(EZDIT_E_UNSUPPORTED, "gemm variant %d not supported", variant);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EZDIT_E_HIP, "launch of k_gemm failed: %s", hipGetErrorString(e));
    return EZDIT_OK;
}

// EPI_RESID (un-split residual projection + partial LayerNorm statistics + next operand), stand-alone:
// h_out = h_in + gate * (A . W^T + bias); zu = bf16(h_out * zg); zstat[N / tile-width chunks][row] = (sum, sum of squares) over the chunk's columns (part-major)
int ezdit_test_resid(int tile, const void* A, int lda, const void* W, int ldw, const float* bias, const float* h_in, const float* gate, const float* zg,
                     float* h_out, void* zu, int ld_zu, void* zstat, int M, int N, int K, ezdit_stream stream) {
    if (K % 64) return fail(EZDIT_E_INVALID, "K=%d must be a multiple of 64", K);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.wrows = (int)rup(N, 128); g.bias = bias;
    g.out = h_out; g.ldo = N; g.M = M; g.N = N; g.K = K; g.splitk = 1; g.epi = EPI_RESID; g.tile = tile; g.xcd_map = 1;
    g.resid = h_in; g.ldr = N; g.gate = gate; g.rows_per_b = 1;
    g.zu = (bf16_t*)zu; g.ld_zu = ld_zu; g.zg = zg; g.zstat_out = (float2*)zstat; g.zs_stride = M;   // zstat [N tiles][M]
    g.ts = g_gemm_ts; g.ts_cap = g_gemm_ts_cap;
    (void)hipGetLastError();
    if (launch_gemm(g, (hipStream_t)stream)) return fail(EZDIT_E_UNSUPPORTED, "residual GEMM configuration not supported");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EZDIT_E_HIP, "launch of the residual GEMM failed: %s", hipGetErrorString(e));
    return EZDIT_OK;
}

// the skip path's forms of the same launch (GemmArgs COPY2 / ZIN), stand-alone:
//   zu2 != NULL (COPY2; gate and h_in required): additionally zu2 = bf16(h_out * zg2)
//   zstat_in2 != NULL (ZIN; no gate, no h_in): the operand A is bf16(x * g) of rows whose (sum, sum of squares) over zD columns come in two part-major sets of zparts parts
//   (zstat_in, zstat_in2: [zparts][M] float pairs); acc := r (acc - mu zG[col]) + bias[col] (bias = C'), then the epilogue as above.  h_out NULL = the fp32 stream is not stored
int ezdit_test_resid_skip(int tile, const void* A, int lda, const void* W, int ldw, const float* bias, const float* h_in, const float* gate, const float* zg,
                          float* h_out, void* zu, int ld_zu, void* zstat, int M, int N, int K,
                          const float* zg2, void* zu2, int ld_zu2, const void* zstat_in, const void* zstat_in2, int zparts, int zD, const float* zG, ezdit_stream stream) {
    if (K % 64) return fail(EZDIT_E_INVALID, "K=%d must be a multiple of 64", K);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.wrows = (int)rup(N, 128); g.bias = bias;
    g.out = h_out; g.ldo = N; g.M = M; g.N = N; g.K = K; g.splitk = 1; g.epi = EPI_RESID; g.tile = tile; g.xcd_map = 1;
    g.resid = h_in; g.ldr = N; g.gate = gate; g.rows_per_b = 1;
    g.zu = (bf16_t*)zu; g.ld_zu = ld_zu; g.zg = zg; g.zstat_out = (float2*)zstat; g.zs_stride = M;
    g.zu2 = (bf16_t*)zu2; g.ld_zu2 = ld_zu2; g.zg2 = zg2;
    g.zstat_in = (const float2*)zstat_in; g.zstat_in2 = (const float2*)zstat_in2; g.zparts = zparts; g.zD = zD; g.zG = zG; g.zeps = 1e-5f;
    (void)hipGetLastError();
    if (launch_gemm(g, (hipStream_t)stream)) return fail(EZDIT_E_UNSUPPORTED, "residual GEMM configuration not supported");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EZDIT_E_HIP, "launch of the residual GEMM failed: %s", hipGetErrorString(e));
    return EZDIT_OK;
}

int ezdit_test_attention(ezdit_handle* h, const void* q, const void* k, const void* v, const uint8_t* kmask, void* out, int B,
                         int Lq, int Lk, int Lqp, int Lkp, ezdit_stream stream) {
    if (!h) return fail(EZDIT_E_INVALID, "null handle");
    AttnArgs a;
    memset(&a, 0, sizeof a);
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kmask = kmask;
    a.nkh = h->opt_attn_nkh; a.xcd_map = h->opt_attn_xcd;
    a.out = (bf16_t*)out; a.ldo = h->ldD;
    a.B = B; a.H = h->H; a.Lq = Lq; a.Lk = Lk; a.Lqp = Lqp; a.Lkp = Lkp; a.dh = h->dh;
    a.ts = g_gemm_ts; a.ts_cap = g_gemm_ts_cap;
    (void)hipGetLastError();
    if (launch_attention(a, (hipStream_t)stream)) return fail(EZDIT_E_UNSUPPORTED, "attention configuration not supported");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(EZDIT_E_HIP, "launch of k_attn failed: %s", hipGetErrorString(e));
    return EZDIT_OK;
}

int ezdit_debug_buffer(ezdit_handle* h, const char* name, void** ptr, size_t* bytes) {
    if (!h || !h->ws || !name) return fail(EZDIT_E_STATE, "bind workspace first");
    auto it = h->bufs.find(name);
    if (it == h->bufs.end()) return fail(EZDIT_E_INVALID, "no workspace buffer named %s", name);
    if (ptr) *ptr = h->ws + it->second.off;
    if (bytes) *bytes = it->second.bytes;
    return EZDIT_OK;
}

int ezdit_last_launch_count(const ezdit_handle* h) { return h ? h->launches : 0; }
int ezdit_debug_stop_after(ezdit_handle* h, int n) {
    if (!h) return fail(EZDIT_E_INVALID, "null handle");
    h->debug_stop = n;
    return EZDIT_OK;
}
int ezdit_set_option(ezdit_handle* h, const char* name, int value) {
    if (!h || !name) return fail(EZDIT_E_INVALID, "null argument");
    if (!strcmp(name, "zfuse")) h->opt_zfuse = value;
    else if (!strcmp(name, "xkey1")) h->opt_xkey1 = value;
    else if (!strcmp(name, "skip_z")) h->opt_skip_z = value;
    else if (!strcmp(name, "geglu_co")) h->opt_geglu_co = value;
    else if (!strcmp(name, "qkv_co")) h->opt_qkv_co = value;
    else if (!strcmp(name, "wt")) h->opt_wt = value;
    else if (!strcmp(name, "gemm_pp")) h->opt_gemm_pp = value;
    else if (!strcmp(name, "tile_partial")) h->opt_tile_partial = value;
    else if (!strcmp(name, "attn_xcd")) h->opt_attn_xcd = value;
    else if (!strcmp(name, "row_variant")) h->opt_row_variant = value;
    else if (!strcmp(name, "gemm_panel")) h->opt_gemm_panel = value;
    else if (!strcmp(name, "row_affine")) h->opt_row_affine = value;
    else if (!strcmp(name, "epi_lds")) h->opt_epi_lds = value;
    else if (!strcmp(name, "attn_xk2")) h->opt_attn_xk2 = value;
    else if (!strcmp(name, "attn_nkh")) h->opt_attn_nkh = value;
    else if (!strcmp(name, "attn_qtile")) h->opt_attn_qtile = value;
    else if (!strcmp(name, "cn_overlap")) h->opt_cn_overlap = value;
    else if (!strcmp(name, "fuse_q2")) h->opt_fuse_q2 = value;
    else if (!strcmp(name, "q2_pp")) h->opt_q2_pp = value;
    else if (!strcmp(name, "stamp_launch")) h->opt_stamp_launch = value;
    else if (!strcmp(name, "trace_launches")) h->opt_trace_launches = value;
#ifdef EZ_DIAG
    else if (!strcmp(name, "zfake")) h->opt_zfake = value;
#endif
    else return fail(EZDIT_E_INVALID, "unknown option %s", name);
    drop_graph(h);
    for (ezdit_handle* u : h->cn_users) drop_graph(u);   // a backbone's captured step embeds the attached ControlNet's kernels and arguments
    return EZDIT_OK;
}

}  // extern "C"
