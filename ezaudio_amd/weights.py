"""state-dict -> packed device blob, following the layout libezaudio_hip.so declares.

The C library is the single source of truth for the layout (``ezdit_param_info``): slot name, the
checkpoint keys concatenated into it, dtype, padding, row transform and byte offset.  Accepts exactly
what the reference loads: ``torch.load(ckpt, map_location='cpu')['model']`` with the key names of
``MaskDiT.state_dict()`` (/root/reference/api/ezaudio.py:83-85; key list in SURVEY.md section 8a row W).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def param_table(handle):
    lib = _lib.load()
    out = []
    info = _lib.EzditParamInfo()
    for i in range(lib.ezdit_param_count(handle)):
        _lib.check(lib.ezdit_param_info(handle, i, C.byref(info)))
        out.append(dict(name=info.name.decode(),
                        src=[bytes(info.src[j]).split(b'\0')[0].decode() for j in range(info.nsrc)],
                        dtype=info.dtype, transform=info.transform, rows=info.rows, cols=info.cols,
                        rows_pad=info.rows_pad, ld=info.ld, offset=info.offset))
    return out


def _geglu8(t):
    """[2I, cols] with value rows first, gate rows second (GEGLU chunk order, modules.py:274-275) ->
    groups of 16 rows = 8 value rows followed by their 8 gate rows (what the GEMM epilogue pairs in-lane)."""
    two_i, cols = t.shape
    inner = two_i // 2
    if inner % 8:
        raise NotImplementedError(f'GEGLU inner dim {inner} not a multiple of 8')
    val = t[:inner].reshape(inner // 8, 8, cols)
    gate = t[inner:].reshape(inner // 8, 8, cols)
    return torch.cat([val, gate], dim=1).reshape(two_i, cols)


def qkrope_col(dh, c):
    """(head within the pair, channel) that tile column c of a two-head q / k tile holds (EZDIT_T_QKROPE; csrc/gemm_pp.h qkrope_col)."""
    j, g, e, s, fh = c >> 4, (c >> 2) & 3, (c >> 1) & 1, c & 1, dh // 16
    if dh % 16 == 0:
        hh, f = j // fh, 8 * (j % fh) + 2 * g + e
    elif j < fh:
        hh, f = 0, 8 * j + 2 * g + e
    elif j == fh:
        hh, f = g >> 1, 8 * fh + 2 * (g & 1) + e
    else:
        hh, f = 1, 8 * (j - fh - 1) + 2 * g + e
    return hh, f + (dh // 2) * s


def _qkrope(t, dh):
    """[3D, D] fused to_q | to_k | to_v weight: the q and k rows of every pair of heads re-ordered into RoPE-pair order; v rows untouched."""
    D = t.shape[0] // 3
    if D % (2 * dh):
        raise NotImplementedError(f'EZDIT_T_QKROPE needs an even head count (D={D}, head_dim={dh})')
    src = []
    for c in range(2 * dh):
        hh, ch = qkrope_col(dh, c)
        src.append(hh * dh + ch)
    assert sorted(src) == list(range(2 * dh))
    idx = torch.arange(3 * D)
    pair = torch.tensor(src)
    for part in range(2):
        for p0 in range(0, D, 2 * dh):
            idx[part * D + p0:part * D + p0 + 2 * dh] = part * D + p0 + pair
    return t[idx]


def pack_state_dict(handle, state_dict, strict=True):
    """Returns a CPU uint8 tensor holding the packed blob."""
    lib = _lib.load()
    total = lib.ezdit_param_bytes(handle)
    blob = torch.zeros(total, dtype=torch.uint8)
    used = set()
    table = param_table(handle)
    head_dim = next(int(p['cols']) for p in table if p['name'].endswith('.a.qnw'))   # attn.norm_q.weight is [head_dim]
    for p in table:
        parts = []
        for key in p['src']:
            if key not in state_dict:
                raise KeyError(f'missing key in state_dict: {key}')
            v = state_dict[key]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach().cpu()
            parts.append(v.to(torch.float32))
            used.add(key)
        vec = p['rows'] == 1
        t = torch.cat([x.reshape(1, -1) if vec else x.reshape(x.shape[0], -1) for x in parts], dim=1 if vec else 0)
        if p['transform'] == _lib.T_GEGLU8:
            t = _geglu8(t.reshape(-1, 1)).reshape(1, -1) if vec else _geglu8(t)
        elif p['transform'] == _lib.T_QKROPE:
            t = _qkrope(t, head_dim)
        if tuple(t.shape) != (p['rows'], p['cols']):
            raise ValueError(f"{p['name']}: checkpoint shape {tuple(t.shape)} != expected {(p['rows'], p['cols'])}")
        dt = torch.bfloat16 if p['dtype'] == _lib.P_BF16 else torch.float32
        padded = torch.zeros(p['rows_pad'], p['ld'], dtype=dt)
        padded[:p['rows'], :p['cols']] = t.to(dt)
        raw = padded.view(torch.uint8).reshape(-1)
        blob[p['offset']:p['offset'] + raw.numel()] = raw
    if strict:
        extra = [k for k in state_dict if k not in used and not k.endswith('rotary.inv_freq')]
        if extra:
            raise KeyError(f'unexpected keys in state_dict: {extra[:5]}{"..." if len(extra) > 5 else ""}')
    return blob


def random_state_dict(model_cfg, seed=0):
    """Random-init weights of the architecture, with the reference's state-dict key names and shapes
    (benchmarks and smoke tests: there is no network for real checkpoints).  Every tensor is non-zero --
    the reference zero-initialises cross-attention / AdaLN / biases (src/models/udit.py:199-243), which
    would turn those paths into no-ops.  Linear weights use the xavier scale of the reference init."""
    D, C = model_cfg['embed_dim'], model_cfg['out_chans']
    H = model_cfg['num_heads']
    dh = D // H
    inner = int(D * model_cfg['mlp_ratio'])
    r6 = 6 * model_cfg['ada_sola_rank']
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def mat(name, *shape, small=False):
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        std = 0.02 if small else (2.0 / (fan_in + fan_out)) ** 0.5
        sd[name] = torch.randn(*shape, generator=g) * std

    def vec(name, n, ln=False):
        sd[name] = (1.0 + 0.1 * torch.randn(n, generator=g)) if ln else 0.02 * torch.randn(n, generator=g)
    vec('mask_embed', C)
    mat('model.patch_embed.proj.weight', D, model_cfg['in_chans'], 1); vec('model.patch_embed.proj.bias', D)
    mat('model.time_embed.mlp.0.weight', D, 256); vec('model.time_embed.mlp.0.bias', D)
    mat('model.time_embed.mlp.2.weight', D, D); vec('model.time_embed.mlp.2.bias', D)
    mat('model.time_ada.weight', 6 * D, D, small=True); vec('model.time_ada.bias', 6 * D)
    mat('model.time_ada_final.weight', 2 * D, D, small=True); vec('model.time_ada_final.bias', 2 * D)
    mat('model.context_embed.0.weight', D, model_cfg['context_dim']); vec('model.context_embed.0.bias', D)
    mat('model.context_embed.2.weight', D, D); vec('model.context_embed.2.bias', D)
    n = model_cfg['depth'] // 2
    prefixes = [f'model.in_blocks.{i}' for i in range(n)] + ['model.mid_block'] + [f'model.out_blocks.{i}' for i in range(n)]
    for bi, p in enumerate(prefixes):
        for nm in ('norm1', 'norm2', 'norm3', 'norm_context'):
            vec(f'{p}.{nm}.weight', D, ln=True); vec(f'{p}.{nm}.bias', D)
        for att in ('attn', 'cross_attn'):
            for w in ('to_q', 'to_k', 'to_v'):
                mat(f'{p}.{att}.{w}.weight', D, D)
            for nm in ('norm_q', 'norm_k'):
                vec(f'{p}.{att}.{nm}.weight', dh, ln=True); vec(f'{p}.{att}.{nm}.bias', dh)
            mat(f'{p}.{att}.proj.weight', D, D, small=(att == 'cross_attn')); vec(f'{p}.{att}.proj.bias', D)
        mat(f'{p}.mlp.net.0.proj.weight', 2 * inner, D); vec(f'{p}.mlp.net.0.proj.bias', 2 * inner)
        mat(f'{p}.mlp.net.2.weight', D, inner); vec(f'{p}.mlp.net.2.bias', D)
        sd[f'{p}.adaln.scale_shift_table'] = 0.05 * torch.randn(6, D, generator=g)
        mat(f'{p}.adaln.lora_a.weight', r6, D); mat(f'{p}.adaln.lora_b.weight', 6 * D, r6, small=True)
        if bi > n:
            vec(f'{p}.skip_norm.weight', 2 * D, ln=True); vec(f'{p}.skip_norm.bias', 2 * D)
            mat(f'{p}.skip_linear.weight', D, 2 * D); vec(f'{p}.skip_linear.bias', D)
    vec('model.final_block.norm.weight', D, ln=True); vec('model.final_block.norm.bias', D)
    mat('model.final_block.linear.weight', C, D); vec('model.final_block.linear.bias', C)
    mat('model.final_block.final_layer.weight', C, C, 3); vec('model.final_block.final_layer.bias', C)
    return sd


def random_controlnet_state_dict(model_cfg, cn_cfg, seed=0):
    """Random-init DiTControlNet weights with the reference's key names (src/models/controlnet.py): the backbone's
    embeds + first depth/2 blocks without the 'model.' prefix, controlnet_pre.*, controlnet_zero_blocks.* -- all non-zero
    (the reference zero-initialises conv_out and the zero blocks, which would make the branch a no-op)."""
    full = random_state_dict(model_cfg, seed)
    D = model_cfg['embed_dim']
    sd = {}
    for k, v in full.items():
        if not k.startswith('model.'):
            continue
        k2 = k[len('model.'):]
        if k2.startswith(('mid_block', 'out_blocks', 'final_block', 'time_ada_final')):
            continue
        sd[k2] = v
    g = torch.Generator().manual_seed(seed + 1)
    c0, c1 = cn_cfg['cond_blocks']
    c0m = c0 + (1 if cn_cfg.get('cond_mask') else 0)

    def t(*shape, std):
        return torch.randn(*shape, generator=g) * std
    sd['controlnet_pre.conv_in.weight'] = t(c0, cn_cfg['cond_in'], 1, std=0.5)
    sd['controlnet_pre.conv_in.bias'] = t(c0, std=0.02)
    if cn_cfg.get('cond_mask'):
        sd['controlnet_pre.mask_embed'] = t(c0, std=0.02)
    sd['controlnet_pre.blocks.0.0.weight'] = t(c0m, c0m, 3, std=(1.0 / (3 * c0m)) ** 0.5)
    sd['controlnet_pre.blocks.0.0.bias'] = t(c0m, std=0.02)
    sd['controlnet_pre.blocks.0.2.weight'] = t(c1, c0m, 3, std=(1.0 / (3 * c0m)) ** 0.5)
    sd['controlnet_pre.blocks.0.2.bias'] = t(c1, std=0.02)
    sd['controlnet_pre.conv_out.weight'] = t(D, c1, 1, std=0.02)
    sd['controlnet_pre.conv_out.bias'] = t(D, std=0.02)
    for i in range(model_cfg['depth'] // 2):
        sd[f'controlnet_zero_blocks.{i}.weight'] = t(D, D, std=(1.0 / D) ** 0.5 * 0.2)
        sd[f'controlnet_zero_blocks.{i}.bias'] = t(D, std=0.02)
    return sd
