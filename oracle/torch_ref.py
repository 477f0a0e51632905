"""PyTorch fp32 eager restatement of the reference denoiser forward -- the CPU BASELINE leg.  TEST INFRASTRUCTURE.

Why it exists: BASELINE.md section 3 asks for the reference's own CPU path (fp32, PyTorch eager, all host cores) to be timed
next to every MI355X number.  /root/reference does not exist on the GPU box, so `bench.py`'s `cpu_baseline` cannot import
the reference there.  This file restates `MaskDiT.forward` -> `UDiT.forward` with THE SAME aten operators the reference
modules call (F.linear, F.layer_norm, F.scaled_dot_product_attention, F.gelu, F.silu, F.conv1d), so its CPU time is the
reference's CPU time to within noise: `tools/ref_cpu_baseline.py` times both side by side in the build container
(profiles/ref_cpu_baseline.json).  The numpy oracle (oracle/dit.py) stays the numerics checker; this port is pinned against
the same reference-minted goldens (tests/test_oracle.py).

Reference lines: src/models/conditioners.py:156-183, src/models/udit.py:281-362, src/models/blocks.py:39-45,120-160,199-211,
src/models/utils/attention.py:106-149, src/models/utils/rotary.py:6-18,56-84, src/models/utils/modules.py:15-61,263-277.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _rope(x, cos, sin):   # rotary.py:6-18 (half split)
    half = x.shape[-1] // 2
    rot = torch.cat([-x[..., half:], x[..., :half]], dim=-1)
    return x * cos + rot * sin


class DiTTorchRef:
    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.sd = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in sd.items()}
        self.D, self.H = cfg['embed_dim'], cfg['num_heads']
        self.dh = self.D // self.H
        self.n_half = cfg['depth'] // 2
        self.scaling = cfg['ada_sola_alpha'] / cfg['ada_sola_rank']

    def p(self, k):
        return self.sd[k]

    def ln(self, x, pfx):
        return F.layer_norm(x, (x.shape[-1],), self.p(pfx + '.weight'), self.p(pfx + '.bias'), 1e-5)

    def attention(self, pfx, x, context=None, key_mask=None, rope=None):   # attention.py:122-149
        H = self.H
        ctx = x if context is None else context
        B, L, _ = x.shape

        def heads(t):
            return t.view(t.shape[0], t.shape[1], H, -1).transpose(1, 2)
        q = heads(F.linear(x, self.p(pfx + '.to_q.weight')))
        k = heads(F.linear(ctx, self.p(pfx + '.to_k.weight')))
        v = heads(F.linear(ctx, self.p(pfx + '.to_v.weight')))
        q = self.ln(q, pfx + '.norm_q')
        k = self.ln(k, pfx + '.norm_k')
        if rope is not None:
            q, k = _rope(q, *rope), _rope(k, *rope)
        mask = None if key_mask is None else key_mask[:, None, None, :].expand(B, H, L, key_mask.shape[-1])
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        o = o.transpose(1, 2).reshape(B, L, H * self.dh)
        return F.linear(o, self.p(pfx + '.proj.weight'), self.p(pfx + '.proj.bias'))

    def block(self, pfx, x, tt, ada, skip, c, ctx_mask, rope):              # blocks.py:120-160
        if skip is not None:
            cat = self.ln(torch.cat([x, skip], dim=-1), pfx + '.skip_norm')
            x = F.linear(cat, self.p(pfx + '.skip_linear.weight'), self.p(pfx + '.skip_linear.bias'))
        lora = F.linear(F.linear(tt, self.p(pfx + '.adaln.lora_a.weight')), self.p(pfx + '.adaln.lora_b.weight')) * self.scaling
        ta = self.p(pfx + '.adaln.scale_shift_table')[None] + (ada + lora).view(ada.shape[0], 6, -1)
        sh1, sc1, g1, sh2, sc2, g2 = [ta[:, i:i + 1, :] for i in range(6)]
        xn = self.ln(x, pfx + '.norm1') * (1 + sc1) + sh1
        x = x + (1 - g1) * self.attention(pfx + '.attn', xn, rope=rope)
        cn = self.ln(c, pfx + '.norm_context')
        x = x + self.attention(pfx + '.cross_attn', self.ln(x, pfx + '.norm2'), context=cn, key_mask=ctx_mask)
        xn = self.ln(x, pfx + '.norm3') * (1 + sc2) + sh2
        h = F.linear(xn, self.p(pfx + '.mlp.net.0.proj.weight'), self.p(pfx + '.mlp.net.0.proj.bias'))
        val, gate = h.chunk(2, dim=-1)
        return x + (1 - g2) * F.linear(val * F.gelu(gate), self.p(pfx + '.mlp.net.2.weight'), self.p(pfx + '.mlp.net.2.bias'))

    @torch.no_grad()
    def forward(self, x, t, ctx, ctx_mask=None, gt=None, mae_mask_infer=None):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32) if not torch.is_tensor(x) else x.float()
        ctx = torch.as_tensor(np.asarray(ctx), dtype=torch.float32) if not torch.is_tensor(ctx) else ctx.float()
        if ctx_mask is not None and not torch.is_tensor(ctx_mask):
            ctx_mask = torch.as_tensor(np.asarray(ctx_mask), dtype=torch.bool)
        B, C, L = x.shape
        me = self.p('mask_embed').view(1, C, 1).expand_as(x)                # conditioners.py:161-176
        if gt is None:
            g, mae = me, torch.ones_like(x)
        else:
            m = torch.as_tensor(np.asarray(mae_mask_infer), dtype=torch.bool).expand_as(x)
            g, mae = torch.where(m, me, torch.as_tensor(np.asarray(gt), dtype=torch.float32)), m.float()
        x257 = torch.cat([x, g, mae[:, 0:1, :]], dim=1)
        h = F.conv1d(x257, self.p('model.patch_embed.proj.weight'), self.p('model.patch_embed.proj.bias')).transpose(1, 2)
        c = F.linear(F.silu(F.linear(ctx, self.p('model.context_embed.0.weight'), self.p('model.context_embed.0.bias'))),
                     self.p('model.context_embed.2.weight'), self.p('model.context_embed.2.bias'))
        tv = torch.as_tensor(np.asarray(t), dtype=torch.float32).reshape(-1)
        if tv.numel() == 1:
            tv = tv.expand(B)
        half = 128                                                           # modules.py:19-37, dim 256
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = tv[:, None] * freqs[None]
        e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        tt = F.linear(F.silu(F.linear(e, self.p('model.time_embed.mlp.0.weight'), self.p('model.time_embed.mlp.0.bias'))),
                      self.p('model.time_embed.mlp.2.weight'), self.p('model.time_embed.mlp.2.bias'))
        tt = F.silu(tt)
        ada_final = F.linear(tt, self.p('model.time_ada_final.weight'), self.p('model.time_ada_final.bias'))
        ada = F.linear(tt, self.p('model.time_ada.weight'), self.p('model.time_ada.bias'))
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, self.dh, 2, dtype=torch.float32) / self.dh))
        fr = torch.einsum('i,j->ij', torch.arange(L, dtype=torch.float32), inv_freq)
        emb = torch.cat([fr, fr], dim=-1)
        rope = (emb.cos(), emb.sin())
        skips = []
        for i in range(self.n_half):
            h = self.block(f'model.in_blocks.{i}', h, tt, ada, None, c, ctx_mask, rope)
            skips.append(h)
        h = self.block('model.mid_block', h, tt, ada, None, c, ctx_mask, rope)
        for i in range(self.n_half):
            h = self.block(f'model.out_blocks.{i}', h, tt, ada, skips.pop(), c, ctx_mask, rope)
        D = self.D
        shift, scale = ada_final[:, None, :D], ada_final[:, None, D:]      # blocks.py:203-204: shift first
        y = self.ln(h, 'model.final_block.norm') * (1 + scale) + shift
        y = F.linear(y, self.p('model.final_block.linear.weight'), self.p('model.final_block.linear.bias')).transpose(1, 2)
        out = F.conv1d(y, self.p('model.final_block.final_layer.weight'), self.p('model.final_block.final_layer.bias'), padding=1)
        return out, mae
