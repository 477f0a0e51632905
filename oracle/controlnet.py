"""numpy restatement of the reference ControlNet branch.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/models/controlnet.py: ``DiTControlNetEmbed.forward`` (:65-84, eval path: no position is
masked, the appended mask channel is all zeros) and ``DiTControlNet.forward`` (:252-315), plus the energy control curve
``EnergyExtractor.forward`` (src/models/conditions/energy.py:19-56).  Pinned by tests/golden/cn_*.npz, minted from the
reference's own ``DiTControlNet`` by oracle/mint_golden.py.
"""
import numpy as np

from .dit import DiTOracle, linear, rope_tables, silu
from .weights import make_tensor, param_shapes

CN_DEFAULT = dict(cond_in=1, cond_blocks=[64, 128], cond_mask=True, cond_mask_prob=0.25, cond_mask_ratio=[0.25, 0.50],
                  cond_mask_span=10)   # ckpts/controlnet/energy_l.yml:38-44


def controlnet_param_shapes(cfg, cn=CN_DEFAULT):
    """State-dict of DiTControlNet: the backbone's patch/time/context embeds + the first depth/2 blocks (no 'model.'
    prefix, no mask_embed / time_ada_final / final_block / mid / out blocks) + controlnet_pre + controlnet_zero_blocks."""
    D = cfg['embed_dim']
    c0, c1 = cn['cond_blocks']
    c0m = c0 + (1 if cn['cond_mask'] else 0)
    sh = {}
    for k, v in param_shapes(cfg).items():
        if not k.startswith('model.'):
            continue
        k2 = k[len('model.'):]
        if k2.startswith(('mid_block', 'out_blocks', 'final_block', 'time_ada_final')):
            continue
        sh[k2] = v
    sh['controlnet_pre.conv_in.weight'] = ((c0, cn['cond_in'], 1), 'xavier')
    sh['controlnet_pre.conv_in.bias'] = ((c0,), 'small')
    if cn['cond_mask']:
        sh['controlnet_pre.mask_embed'] = ((c0,), 'small')
    sh['controlnet_pre.blocks.0.0.weight'] = ((c0m, c0m, 3), 'xavier')
    sh['controlnet_pre.blocks.0.0.bias'] = ((c0m,), 'small')
    sh['controlnet_pre.blocks.0.2.weight'] = ((c1, c0m, 3), 'xavier')
    sh['controlnet_pre.blocks.0.2.bias'] = ((c1,), 'small')
    sh['controlnet_pre.conv_out.weight'] = ((D, c1, 1), 'small')    # zero-initialised in the reference (:38-39)
    sh['controlnet_pre.conv_out.bias'] = ((D,), 'small')
    for i in range(cfg['depth'] // 2):
        sh[f'controlnet_zero_blocks.{i}.weight'] = ((D, D), 'xavier')   # zero-initialised in the reference (:231-232)
        sh[f'controlnet_zero_blocks.{i}.bias'] = ((D,), 'small')
    return sh


def make_controlnet_state_dict(cfg, cn=CN_DEFAULT, seed=0):
    return {k: make_tensor('cn.' + k, s, kind, seed) for k, (s, kind) in controlnet_param_shapes(cfg, cn).items()}


def conv1d(x, w, b, stride=1, pad=0):
    """x [B,Ci,L], w [Co,Ci,K] -> [B,Co,Lout] (nn.Conv1d)."""
    B, Ci, L = x.shape
    Co, _, K = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    Lout = (L + 2 * pad - K) // stride + 1
    out = np.zeros((B, Co, Lout), dtype=x.dtype)
    for k in range(K):
        out += np.einsum('oc,bcl->bol', w[:, :, k], xp[:, :, k:k + stride * Lout:stride])
    return out + b[None, :, None]


def energy_curve(audio, hop_size=240, window_size=1920, min_db=-60.0, norm=True):
    """EnergyExtractor.forward with padding='reflect' (energy.py:19-56) -> [B, T, 1]."""
    audio = np.asarray(audio, dtype=np.float32)
    n_frames = audio.shape[-1] // hop_size
    pad = (window_size - hop_size) // 2
    sq = np.pad(audio, ((0, 0), (pad, pad)), mode='reflect') ** 2
    idx = np.arange(n_frames)[:, None] * hop_size + np.arange(window_size)[None, :]
    energy = sq[:, idx].mean(axis=-1, dtype=np.float32)
    gain_db = 10 * np.log10(np.maximum(energy, np.float32(10 ** (min_db / 10))))
    if norm:
        mx = gain_db.max(axis=-1, keepdims=True)
        gain_db = (gain_db - min_db) / (mx - min_db + 1e-8)
    return gain_db[..., None].astype(np.float32)


class ControlNetOracle(DiTOracle):
    def __init__(self, cfg, sd, cn=CN_DEFAULT, dtype=np.float32):
        super().__init__(cfg, sd, dtype, prefix='')
        self.cn = cn

    def embed(self, condition):
        """controlnet.py:65-84 at inference."""
        e = conv1d(np.asarray(condition, dtype=self.dtype), self.p('controlnet_pre.conv_in.weight'), self.p('controlnet_pre.conv_in.bias'))
        if self.cn['cond_mask']:
            e = np.concatenate([e, np.zeros_like(e[:, :1])], axis=1)   # nothing masked; mask channel = 0
        e = silu(conv1d(e, self.p('controlnet_pre.blocks.0.0.weight'), self.p('controlnet_pre.blocks.0.0.bias'), 1, 1))
        e = silu(conv1d(e, self.p('controlnet_pre.blocks.0.2.weight'), self.p('controlnet_pre.blocks.0.2.bias'), 2, 1))
        e = conv1d(e, self.p('controlnet_pre.conv_out.weight'), self.p('controlnet_pre.conv_out.bias'))
        return e.transpose(0, 2, 1)

    def forward(self, x257, t, ctx, ctx_mask, condition, conditioning_scale=1.0):
        """controlnet.py:252-315 -> list of depth/2 residuals [B, L, D]."""
        x257 = np.asarray(x257, dtype=self.dtype)
        B, _, L = x257.shape
        w = self.p('patch_embed.proj.weight')[:, :, 0]
        x = x257.transpose(0, 2, 1) @ w.T + self.p('patch_embed.proj.bias')
        x = x + self.embed(condition)
        c = self.context_embed(ctx)
        tt, ada, _ = self.time_path(t, B)
        inv_freq = (1.0 / (10000.0 ** (np.arange(0, self.dh, 2, dtype=np.float32) / np.float32(self.dh)))).astype(np.float32)
        rope = rope_tables(L, inv_freq)
        ctx_mask = None if ctx_mask is None else np.asarray(ctx_mask, dtype=bool)
        out = []
        for i in range(self.n_half):
            x = self.block(f'in_blocks.{i}', x, tt, ada, None, c, ctx_mask, rope)
            out.append(linear(x, self.p(f'controlnet_zero_blocks.{i}.weight'), self.p(f'controlnet_zero_blocks.{i}.bias'))
                       * self.dtype(conditioning_scale))
        return out
