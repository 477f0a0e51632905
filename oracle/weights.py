"""Deterministic synthetic checkpoints for the oracle, the golden fixtures and the GPU tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference ships no weights
(they are downloaded at run time, /root/reference/api/ezaudio.py:20-28,44-65)
and this environment has no network, so every parity check runs on seeded
synthetic weights.  The generator below is a counter-based splitmix64 hash in
pure numpy uint64 arithmetic: bit-identical on any machine / numpy / torch
version, so the weights behind a committed golden vector can be regenerated on
the GPU box without shipping gigabytes.

Key names and shapes follow the reference's ``MaskDiT.state_dict()``
(/root/reference/src/models/conditioners.py:124-133, src/models/udit.py:11-180,
src/models/blocks.py:9-105, src/models/utils/attention.py:40-88,
src/models/utils/modules.py:40-61,92-100,263-277,341-374); oracle/mint_golden.py
asserts the key set and shapes against the real module.

"Zero-init trap" (SURVEY.md section 7 item 6): the reference zero-initialises
cross_attn.proj, time_ada, time_ada_final, lora_b, scale_shift_table,
mask_embed and every bias (udit.py:199-243).  A fresh module would make
cross-attention / AdaLN / biases no-ops, so *every* tensor here gets non-zero
values.
"""
import zlib

import numpy as np

# ---------------------------------------------------------------------------------------------
# model-section presets (same key set as the reference's ckpts/ezaudio-*.yml `model:` section)
# ---------------------------------------------------------------------------------------------
_COMMON = dict(
    mae=True, mae_prob=0.25, mask_ratio=[0.25, 1.0], mask_span=10,
    img_size=500, patch_size=1, in_chans=257, out_chans=128, input_type='1d',
    mlp_ratio=4.0, qkv_bias=False, qk_scale=None, qk_norm='layernorm',
    norm_layer='layernorm', act_layer='geglu', context_norm=True, use_checkpoint=True,
    time_fusion='ada_sola_bias', cls_dim=None, context_fusion='cross',
    context_max_length=None, context_pe_method='none', pe_method='none',
    rope_mode='shared', use_conv=True, skip=True, skip_norm=True,
)


def model_config(name):
    """Return the `model:` dict for a named size.

    xl / l    : ckpts/ezaudio-xl.yml:3-36, ckpts/ezaudio-l.yml:3-36 of the reference.
    s / s64   : BASELINE config #1 "EzAudio-S" as defined in SURVEY.md section 8d
                (not in the reference): xl recipe shrunk to 5 blocks, head_dim 72 / 64.
    xs / xs64 : 3-block toys for fast CPU unit tests (head_dim 72 / 64).
    """
    sizes = {
        'xl':   dict(embed_dim=1152, depth=28, num_heads=16, ada_sola_rank=36, ada_sola_alpha=36, context_dim=2048),
        'l':    dict(embed_dim=1024, depth=24, num_heads=16, ada_sola_rank=32, ada_sola_alpha=32, context_dim=1024),
        's':    dict(embed_dim=576, depth=4, num_heads=8, ada_sola_rank=18, ada_sola_alpha=18, context_dim=768),
        's64':  dict(embed_dim=512, depth=4, num_heads=8, ada_sola_rank=16, ada_sola_alpha=16, context_dim=768),
        'xs':   dict(embed_dim=144, depth=2, num_heads=2, ada_sola_rank=4, ada_sola_alpha=4, context_dim=96),
        'xs64': dict(embed_dim=128, depth=2, num_heads=2, ada_sola_rank=4, ada_sola_alpha=4, context_dim=96),
    }
    cfg = dict(_COMMON)
    cfg.update(sizes[name])
    return cfg


def block_prefixes(cfg):
    n = cfg['depth'] // 2
    return ([f'model.in_blocks.{i}' for i in range(n)] + ['model.mid_block'] +
            [f'model.out_blocks.{i}' for i in range(n)])


def param_shapes(cfg):
    """name -> (shape, kind).  kind picks the fill scale, see `_scale`."""
    D = cfg['embed_dim']
    H = cfg['num_heads']
    dh = D // H
    inner = int(D * cfg['mlp_ratio'])
    r6 = 6 * cfg['ada_sola_rank']
    C = cfg['out_chans']
    cin = cfg['in_chans']
    cctx = cfg['context_dim']
    sh = {}
    sh['mask_embed'] = ((C,), 'small')
    sh['model.patch_embed.proj.weight'] = ((D, cin, 1), 'xavier')
    sh['model.patch_embed.proj.bias'] = ((D,), 'small')
    sh['model.time_embed.mlp.0.weight'] = ((D, 256), 'xavier')
    sh['model.time_embed.mlp.0.bias'] = ((D,), 'small')
    sh['model.time_embed.mlp.2.weight'] = ((D, D), 'xavier')
    sh['model.time_embed.mlp.2.bias'] = ((D,), 'small')
    sh['model.time_ada_final.weight'] = ((2 * D, D), 'small')
    sh['model.time_ada_final.bias'] = ((2 * D,), 'small')
    sh['model.time_ada.weight'] = ((6 * D, D), 'small')
    sh['model.time_ada.bias'] = ((6 * D,), 'small')
    sh['model.context_embed.0.weight'] = ((D, cctx), 'xavier')
    sh['model.context_embed.0.bias'] = ((D,), 'small')
    sh['model.context_embed.2.weight'] = ((D, D), 'xavier')
    sh['model.context_embed.2.bias'] = ((D,), 'small')
    nskip = cfg['depth'] // 2
    for bi, p in enumerate(block_prefixes(cfg)):
        for nm in ('norm1', 'norm2', 'norm3', 'norm_context'):
            sh[f'{p}.{nm}.weight'] = ((D,), 'ln_w')
            sh[f'{p}.{nm}.bias'] = ((D,), 'ln_b')
        for att in ('attn', 'cross_attn'):
            for w in ('to_q', 'to_k', 'to_v'):
                sh[f'{p}.{att}.{w}.weight'] = ((D, D), 'xavier')
            for nm in ('norm_q', 'norm_k'):
                sh[f'{p}.{att}.{nm}.weight'] = ((dh,), 'ln_w')
                sh[f'{p}.{att}.{nm}.bias'] = ((dh,), 'ln_b')
            sh[f'{p}.{att}.proj.weight'] = ((D, D), 'xavier' if att == 'attn' else 'small')
            sh[f'{p}.{att}.proj.bias'] = ((D,), 'small')
            if att == 'attn':
                sh[f'{p}.{att}.rotary.inv_freq'] = ((dh // 2,), 'inv_freq')
        sh[f'{p}.mlp.net.0.proj.weight'] = ((2 * inner, D), 'xavier')
        sh[f'{p}.mlp.net.0.proj.bias'] = ((2 * inner,), 'small')
        sh[f'{p}.mlp.net.2.weight'] = ((D, inner), 'xavier')
        sh[f'{p}.mlp.net.2.bias'] = ((D,), 'small')
        sh[f'{p}.adaln.scale_shift_table'] = ((6, D), 'table')
        sh[f'{p}.adaln.lora_a.weight'] = ((r6, D), 'xavier')
        sh[f'{p}.adaln.lora_b.weight'] = ((6 * D, r6), 'small')
        if bi > nskip:  # out blocks carry the U-ViT long skip
            sh[f'{p}.skip_norm.weight'] = ((2 * D,), 'ln_w')
            sh[f'{p}.skip_norm.bias'] = ((2 * D,), 'ln_b')
            sh[f'{p}.skip_linear.weight'] = ((D, 2 * D), 'xavier')
            sh[f'{p}.skip_linear.bias'] = ((D,), 'small')
    sh['model.final_block.norm.weight'] = ((D,), 'ln_w')
    sh['model.final_block.norm.bias'] = ((D,), 'ln_b')
    sh['model.final_block.linear.weight'] = ((C, D), 'xavier')
    sh['model.final_block.linear.bias'] = ((C,), 'small')
    sh['model.final_block.final_layer.weight'] = ((C, C, 3), 'xavier')
    sh['model.final_block.final_layer.bias'] = ((C,), 'small')
    return sh


# ---------------------------------------------------------------------------------------------
# counter-based deterministic fill
# ---------------------------------------------------------------------------------------------
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x):
    with np.errstate(over='ignore'):
        x = (x + _GOLD)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def uniform_pm1(key, n, seed=0):
    """n float32 values in [-1, 1), a pure function of (key, seed, index)."""
    base = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) << np.uint64(32)
    with np.errstate(over='ignore'):
        base = base + np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D)
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        idx = np.arange(s, e, dtype=np.uint64)
        with np.errstate(over='ignore'):
            z = _splitmix64(idx + base)
        u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        out[s:e] = u * np.float32(2.0) - np.float32(1.0)
    return out


def _scale(kind, shape):
    if kind == 'xavier':
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:]))
        return float(np.sqrt(6.0 / (fan_in + fan_out)))  # nn.init.xavier_uniform_ bound
    if kind == 'small':
        return 0.02 * np.sqrt(3.0)  # uniform with std 0.02
    if kind == 'table':
        return 0.1
    raise KeyError(kind)


def make_tensor(name, shape, kind, seed=0):
    n = int(np.prod(shape))
    if kind == 'inv_freq':  # buffer, rotary.py:42
        dh = shape[0] * 2
        return (1.0 / (10000.0 ** (np.arange(0, dh, 2, dtype=np.float32) / np.float32(dh)))).astype(np.float32)
    u = uniform_pm1(name, n, seed)
    if kind == 'ln_w':
        return (np.float32(1.0) + np.float32(0.1) * u).reshape(shape)
    if kind == 'ln_b':
        return (np.float32(0.05) * u).reshape(shape)
    return (np.float32(_scale(kind, shape)) * u).reshape(shape)


def make_state_dict(cfg, seed=0):
    """Full synthetic checkpoint: reference key name -> float32 ndarray."""
    return {k: make_tensor(k, s, kind, seed) for k, (s, kind) in param_shapes(cfg).items()}


# ---------------------------------------------------------------------------------------------
# deterministic inputs
# ---------------------------------------------------------------------------------------------
def make_inputs(cfg, B=2, L=500, Lc=100, n_valid=(12, 1), seed=11, with_gt=False):
    """x ~ U-ish noise scaled to unit variance, random T5-like context, key masks.

    cond rows: first n_valid[i] tokens valid; the uncond row of the reference is the
    empty prompt which tokenises to EOS + padding -> one valid token.
    """
    C = cfg['out_chans']
    s3 = np.float32(np.sqrt(3.0))
    x = (uniform_pm1('in.x', B * C * L, seed) * s3).reshape(B, C, L)
    ctx = (uniform_pm1('in.ctx', B * Lc * cfg['context_dim'], seed) * s3).reshape(B, Lc, cfg['context_dim'])
    mask = np.zeros((B, Lc), dtype=bool)
    for b in range(B):
        mask[b, :n_valid[b % len(n_valid)]] = True
    out = dict(x=x, ctx=ctx, ctx_mask=mask)
    if with_gt:
        gt = (uniform_pm1('in.gt', B * C * L, seed) * s3).reshape(B, C, L)
        m = np.zeros((B, C, L), dtype=bool)
        m[:, :, L // 5: L // 5 + max(1, L // 3)] = True  # True = region to regenerate
        out.update(gt=gt, gt_mask=m)
    return out
