"""numpy restatement of the reference denoiser forward (MaskDiT -> UDiT).  TEST INFRASTRUCTURE.

Every function cites the reference lines it follows (paths relative to /root/reference).
Arithmetic is done in `dtype` (float32 mirrors the reference; float64 is used by tests to
measure how far fp32 itself sits from the exact answer).  Pinned against the reference's own
modules by oracle/mint_golden.py -> tests/golden/dit_*.npz.
"""
import math

import numpy as np
from scipy.special import erf

LN_EPS = 1e-5  # nn.LayerNorm default, selected at src/models/udit.py:122-123, attention.py:63-65


def layer_norm(x, w, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim: biased variance, affine."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def silu(x):
    return x / (1 + np.exp(-x))


def gelu_erf(x):
    """F.gelu default (exact erf), src/models/utils/modules.py:268-272."""
    return x * (x.dtype.type(0.5) * (1 + erf(x * x.dtype.type(1.0 / math.sqrt(2.0)))))


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def film_modulate(x, shift, scale):
    """src/models/utils/modules.py:15-16."""
    return x * (1 + scale) + shift


def timestep_embedding(t, dim, dtype, max_period=10000):
    """src/models/utils/modules.py:19-37 (freqs are built in fp32 there as well)."""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)
    args = np.asarray(t, dtype=np.float32)[:, None] * freqs[None]
    return np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(dtype)


def rope_tables(L, inv_freq):
    """src/models/utils/rotary.py:56-68: angles p*inv_freq duplicated over both halves, fp32."""
    t = np.arange(L, dtype=np.float32)
    freqs = np.einsum('i,j->ij', t, inv_freq.astype(np.float32))
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb), np.sin(emb)


def apply_rope(x, cos, sin):
    """rotary.py:6-18: x*cos + rotate_half(x)*sin with rotate_half = [-x2 | x1] (half split)."""
    half = x.shape[-1] // 2
    rot = np.concatenate([-x[..., half:], x[..., :half]], axis=-1)
    return x * cos + rot * sin


def split_heads(x, H):
    """einops 'B L (H D) -> B H L D', attention.py:137-139."""
    B, L, C = x.shape
    return x.reshape(B, L, H, C // H).transpose(0, 2, 1, 3)


def merge_heads(x):
    B, H, L, D = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, H * D)


def sdpa(q, k, v, key_mask=None):
    """F.scaled_dot_product_attention with a boolean keep-mask over keys (attention.py:106-110);
    scale = head_dim**-0.5 (attention.py:47)."""
    dh = q.shape[-1]
    s = (q @ k.transpose(0, 1, 3, 2)) * q.dtype.type(dh ** -0.5)
    if key_mask is not None:
        s = np.where(key_mask[:, None, None, :], s, -np.inf)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    return p @ v


class DiTOracle:
    """MaskDiT.forward restated (src/models/conditioners.py:156-183 -> src/models/udit.py:281-362)."""

    def __init__(self, cfg, sd, dtype=np.float32, prefix='model.'):
        self.cfg = cfg
        self.dtype = dtype
        self.prefix = prefix  # '' for DiTControlNet (not wrapped by MaskDiT)
        self.sd = {k: np.asarray(v).astype(dtype) for k, v in sd.items()}
        self.D = cfg['embed_dim']
        self.H = cfg['num_heads']
        self.dh = self.D // self.H
        self.n_half = cfg['depth'] // 2
        self.scaling = cfg['ada_sola_alpha'] / cfg['ada_sola_rank']  # blocks.py:24
        for key, want in (('time_fusion', 'ada_sola_bias'), ('context_fusion', 'cross'), ('rope_mode', 'shared'),
                          ('qk_norm', 'layernorm'), ('norm_layer', 'layernorm'), ('act_layer', 'geglu'),
                          ('pe_method', 'none'), ('context_pe_method', 'none'), ('input_type', '1d')):
            if cfg.get(key) != want:
                raise NotImplementedError(f'oracle restates only {key}={want!r} (got {cfg.get(key)!r})')
        self.taps = None

    def p(self, name):
        return self.sd[name]

    # --- A4: MaskDiT input assembly, conditioners.py:151-154,161-176 -------------------------------
    def assemble_input(self, x, gt=None, mae_mask_infer=None):
        x = np.asarray(x, dtype=self.dtype)
        B, C, L = x.shape
        me = self.p('mask_embed').reshape(1, C, 1)
        if gt is None:
            gt2 = np.broadcast_to(me, x.shape)
            mae_mask = np.ones_like(x)
        else:
            if mae_mask_infer is None:
                raise NotImplementedError('training-time random span masking is out of scope (inference path only)')
            mask = np.broadcast_to(np.asarray(mae_mask_infer, dtype=bool), x.shape)
            gt2 = np.where(mask, np.broadcast_to(me, x.shape), np.asarray(gt, dtype=self.dtype))
            mae_mask = mask.astype(self.dtype)
        x257 = np.concatenate([x, gt2, mae_mask[:, 0:1, :]], axis=1)
        return x257, mae_mask

    # --- A11/A13: Attention.forward, attention.py:122-149 -------------------------------------------
    def attention(self, pfx, x, context=None, key_mask=None, rope=None, tap=None):
        H = self.H
        ctx = x if context is None else context
        q = split_heads(linear(x, self.p(f'{pfx}.to_q.weight')), H)
        k = split_heads(linear(ctx, self.p(f'{pfx}.to_k.weight')), H)
        v = split_heads(linear(ctx, self.p(f'{pfx}.to_v.weight')), H)
        q = layer_norm(q, self.p(f'{pfx}.norm_q.weight'), self.p(f'{pfx}.norm_q.bias'))
        k = layer_norm(k, self.p(f'{pfx}.norm_k.weight'), self.p(f'{pfx}.norm_k.bias'))
        if rope is not None:  # rope_mode 'shared' -> self-attention only (attention.py:77-81)
            cos, sin = rope
            # rotary.py:78-84 computes in fp32 and casts back
            q = apply_rope(q.astype(np.float32), cos, sin).astype(self.dtype) if self.dtype == np.float32 else apply_rope(q, cos, sin)
            k = apply_rope(k.astype(np.float32), cos, sin).astype(self.dtype) if self.dtype == np.float32 else apply_rope(k, cos, sin)
        o = merge_heads(sdpa(q, k, v, key_mask))
        if tap is not None:
            kind = 'x' if context is not None else 's'
            tap(kind + 'q', q); tap(kind + 'k', k); tap(kind + 'v', v); tap(kind + 'o', o)
        return linear(o, self.p(f'{pfx}.proj.weight'), self.p(f'{pfx}.proj.bias'))

    # --- A8: AdaLN.forward (ada_sola_bias), blocks.py:39-45 -----------------------------------------
    def adaln(self, pfx, time_token, time_ada):
        B = time_ada.shape[0]
        lora = linear(linear(time_token, self.p(f'{pfx}.adaln.lora_a.weight')),
                      self.p(f'{pfx}.adaln.lora_b.weight')) * self.dtype(self.scaling)
        ta = (time_ada + lora).reshape(B, 6, -1)
        return self.p(f'{pfx}.adaln.scale_shift_table')[None] + ta

    # --- A14: FeedForward / GEGLU, modules.py:263-277,341-374 --------------------------------------
    def mlp(self, pfx, x, tap=None):
        h = linear(x, self.p(f'{pfx}.mlp.net.0.proj.weight'), self.p(f'{pfx}.mlp.net.0.proj.bias'))
        inner = h.shape[-1] // 2
        val, gate = h[..., :inner], h[..., inner:]  # chunk(2): first half value, second half gate
        if tap is not None:
            tap('act', val * gelu_erf(gate))
        return linear(val * gelu_erf(gate), self.p(f'{pfx}.mlp.net.2.weight'), self.p(f'{pfx}.mlp.net.2.bias'))

    # --- A16: DiTBlock._forward, blocks.py:120-160 --------------------------------------------------
    def block(self, pfx, x, time_token, time_ada, skip, context, ctx_mask, rope):
        tp = self.taps if (self.taps is not None and self.taps.get('_fine')) else None

        def tap(name, v):
            if tp is not None:
                tp[f'{pfx}:{name}'] = np.array(v, copy=True)
        if skip is not None:  # blocks.py:124-128
            cat = np.concatenate([x, skip], axis=-1)
            cat = layer_norm(cat, self.p(f'{pfx}.skip_norm.weight'), self.p(f'{pfx}.skip_norm.bias'))
            tap('ucat', cat)
            x = linear(cat, self.p(f'{pfx}.skip_linear.weight'), self.p(f'{pfx}.skip_linear.bias'))
            tap('h_skip', x)
        ta = self.adaln(pfx, time_token, time_ada)
        tap('ada6', ta)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = [ta[:, i:i + 1, :] for i in range(6)]
        xn = film_modulate(layer_norm(x, self.p(f'{pfx}.norm1.weight'), self.p(f'{pfx}.norm1.bias')), shift_msa, scale_msa)
        tap('u1', xn)
        x = x + (1 - gate_msa) * self.attention(f'{pfx}.attn', xn, rope=rope, tap=tap)  # blocks.py:139: (1 - gate)
        tap('h_attn', x)
        cn = layer_norm(context, self.p(f'{pfx}.norm_context.weight'), self.p(f'{pfx}.norm_context.bias'))
        u2 = layer_norm(x, self.p(f'{pfx}.norm2.weight'), self.p(f'{pfx}.norm2.bias'))
        tap('u2', u2)
        x = x + self.attention(f'{pfx}.cross_attn', u2, context=cn, key_mask=ctx_mask, tap=tap)  # blocks.py:147-151: no gate
        tap('h_cross', x)
        xn = film_modulate(layer_norm(x, self.p(f'{pfx}.norm3.weight'), self.p(f'{pfx}.norm3.bias')), shift_mlp, scale_mlp)
        tap('u3', xn)
        x = x + (1 - gate_mlp) * self.mlp(pfx, xn, tap=tap)
        tap('h_out', x)
        return x

    # --- A6: context path, udit.py:94-97,295-296 ----------------------------------------------------
    def context_embed(self, ctx):
        c = linear(np.asarray(ctx, dtype=self.dtype), self.p(self.prefix + 'context_embed.0.weight'), self.p(self.prefix + 'context_embed.0.bias'))
        return linear(silu(c), self.p(self.prefix + 'context_embed.2.weight'), self.p(self.prefix + 'context_embed.2.bias'))

    # --- A7: time path, modules.py:50-60; udit.py:305-316 -------------------------------------------
    def time_path(self, t, B):
        t = np.asarray(t)
        if t.ndim == 0:  # udit.py:286-287
            t = np.broadcast_to(t, (B,))
        e = timestep_embedding(t, 256, self.dtype)
        m = self.prefix
        tt = linear(silu(linear(e, self.p(m + 'time_embed.mlp.0.weight'), self.p(m + 'time_embed.mlp.0.bias'))),
                    self.p(m + 'time_embed.mlp.2.weight'), self.p(m + 'time_embed.mlp.2.bias'))
        tt = silu(tt)  # time_act, udit.py:313
        ada_final = None
        if (m + 'time_ada_final.weight') in self.sd:  # the ControlNet has no FinalBlock (controlnet.py:160-168)
            ada_final = linear(tt, self.p(m + 'time_ada_final.weight'), self.p(m + 'time_ada_final.bias'))
        ada = linear(tt, self.p(m + 'time_ada.weight'), self.p(m + 'time_ada.bias'))
        return tt, ada, ada_final

    # --- A17: UDiT.forward, udit.py:281-362 ---------------------------------------------------------
    def udit_forward(self, x257, t, ctx, ctx_mask, controlnet_skips=None):
        x257 = np.asarray(x257, dtype=self.dtype)
        B, Cin, L = x257.shape
        # A5 PatchEmbed Conv1d(k=1) -> token major, modules.py:102-111
        w = self.p('model.patch_embed.proj.weight')[:, :, 0]
        x = x257.transpose(0, 2, 1) @ w.T + self.p('model.patch_embed.proj.bias')
        c = self.context_embed(ctx)
        tt, ada, ada_final = self.time_path(t, B)
        inv_freq = self.sd.get('model.mid_block.attn.rotary.inv_freq')
        if inv_freq is None:  # a buffer, not a parameter: rebuild it as rotary.py:42 does
            inv_freq = (1.0 / (10000.0 ** (np.arange(0, self.dh, 2, dtype=np.float32) / np.float32(self.dh)))).astype(np.float32)
        rope = rope_tables(L, inv_freq)
        if self.dtype != np.float32:
            rope = (rope[0].astype(self.dtype), rope[1].astype(self.dtype))
        ctx_mask = None if ctx_mask is None else np.asarray(ctx_mask, dtype=bool)
        taps = self.taps
        if taps is not None:
            taps['patch'] = x.copy(); taps['ctx'] = c.copy(); taps['tt'] = tt.copy()
        skips = []
        cn = list(controlnet_skips) if controlnet_skips else None
        for i in range(self.n_half):
            x = self.block(f'model.in_blocks.{i}', x, tt, ada, None, c, ctx_mask, rope)
            skips.append(x)
            if taps is not None:
                taps[f'in{i}'] = x.copy()
        x = self.block('model.mid_block', x, tt, ada, None, c, ctx_mask, rope)
        if taps is not None:
            taps['mid'] = x.copy()
        for i in range(self.n_half):
            skip = skips.pop()
            if cn:
                skip = skip + np.asarray(cn.pop(), dtype=self.dtype)  # udit.py:345-348
            x = self.block(f'model.out_blocks.{i}', x, tt, ada, skip, c, ctx_mask, rope)
            if taps is not None:
                taps[f'out{i}'] = x.copy()
        # A18 FinalBlock, blocks.py:199-211: (shift, scale) = chunk(2) -> shift first
        D = self.D
        shift, scale = ada_final[:, None, :D], ada_final[:, None, D:]
        y = film_modulate(layer_norm(x, self.p('model.final_block.norm.weight'), self.p('model.final_block.norm.bias')), shift, scale)
        y = linear(y, self.p('model.final_block.linear.weight'), self.p('model.final_block.linear.bias'))
        y = y.transpose(0, 2, 1)  # unpatchify 'B h (p1 C) -> B C (h p1)', p1 = 1
        wc = self.p('model.final_block.final_layer.weight')  # [C, C, 3], Conv1d padding=1
        yp = np.pad(y, ((0, 0), (0, 0), (1, 1)))
        out = sum(np.einsum('oc,bcl->bol', wc[:, :, k], yp[:, :, k:k + L]) for k in range(3))
        return out + self.p('model.final_block.final_layer.bias')[None, :, None]

    def forward(self, x, t, ctx, ctx_mask=None, gt=None, mae_mask_infer=None, controlnet_skips=None):
        x257, mae_mask = self.assemble_input(x, gt, mae_mask_infer)
        return self.udit_forward(x257, t, ctx, ctx_mask, controlnet_skips), mae_mask


def flops_per_step(cfg, B, L, Lc, hoisted=True):
    """Algorithmic GEMM FLOPs of one denoiser evaluation (SURVEY.md section 8d formula):
    per sample MACs = nblk*(18 L D^2 + 2 L^2 D + 2 L Lc D) + nskip*2 L D^2 + L D (257+128) + 3*128^2 L,
    plus the step-invariant context work when it is not hoisted out of the loop."""
    D = cfg['embed_dim']
    nblk = cfg['depth'] + 1
    nskip = cfg['depth'] // 2
    C = cfg['out_chans']
    macs = nblk * (18 * L * D * D + 2 * L * L * D + 2 * L * Lc * D) + nskip * 2 * L * D * D \
        + L * D * (cfg['in_chans'] + C) + 3 * C * C * L
    if not hoisted:
        macs += nblk * 2 * Lc * D * D + Lc * (cfg['context_dim'] * D + D * D)
    return 2.0 * macs * B
