"""GPU diagnostic (not a pytest): stage-by-stage comparison of the HIP forward against the numpy oracle's
taps for a small config, using the ezdit_debug_stop_after hook.  Prints rel-L2 per stage so one gpurun call
localises a wrong kernel.   python tools/diag_forward.py [xs|xs64|s|s64] [L]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.dit import DiTOracle  # noqa: E402
from oracle.weights import make_inputs, make_state_dict, model_config  # noqa: E402
from ezaudio_amd import MaskDiT  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main(size='xs', L=96, Lc=20, t=499):
    cfg = model_config(size)
    sd = make_state_dict(cfg, 1)
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, n_valid=(7, 1), seed=11)
    o = DiTOracle(cfg, sd)
    o.taps = {'_fine': True}
    ref, _ = o.forward(inp['x'], t, inp['ctx'], inp['ctx_mask'])
    T = o.taps
    m = MaskDiT(device='cuda', **cfg)
    m.load_state_dict(sd)
    B, D, H = 2, cfg['embed_dim'], cfg['num_heads']
    dh = D // H
    M = B * L
    Lp = (L + 63) // 64 * 64
    DQK = 64 if dh == 64 else 80
    DV = 64 if dh == 64 else 96
    ldD = (D + 63) // 64 * 64
    inner = 4 * D
    x = torch.from_numpy(inp['x']).cuda()
    ctx = torch.from_numpy(inp['ctx']).cuda()
    msk = torch.from_numpy(inp['ctx_mask']).cuda()

    def run(stop):
        m.lib.ezdit_debug_stop_after(m._h, stop)
        out, _ = m(x, torch.tensor(t), ctx, context_mask=msk)
        torch.cuda.synchronize()
        return out

    def f32(name, shape):
        return m.debug_buffer(name, torch.float32, shape).cpu().numpy()

    def b16(name, shape):
        return m.debug_buffer(name, torch.bfloat16, shape).float().cpu().numpy()
    pfx = 'model.in_blocks.0'
    rows = []
    # time path / context (prepared before forward): mod table of block 0
    run(1)
    mod = f32('mod', (1, cfg['depth'] + 1, 6, D))[0, 0]
    ada6 = T[f'{pfx}:ada6'][0]
    n1w, n1b = sd[f'{pfx}.norm1.weight'], sd[f'{pfx}.norm1.bias']
    rows.append(('mod.g1', rel(mod[0], n1w * (1 + ada6[1]))))
    rows.append(('mod.c1', rel(mod[1], n1b * (1 + ada6[1]) + ada6[0])))
    rows.append(('mod.a1', rel(mod[2], 1 - ada6[2])))
    rows.append(('mod.a3', rel(mod[5], 1 - ada6[5])))
    ape = b16('ape', (M, (cfg['in_chans'] + 63) // 64 * 64))
    x257, _ = o.assemble_input(inp['x'])
    rows.append(('assemble', rel(ape[:, :cfg['in_chans']], x257.transpose(0, 2, 1).reshape(M, -1))))
    run(3)
    rows.append(('patch h', rel(f32('h', (M, D)), T['patch'].reshape(M, D))))
    u = b16('u', (M, 2 * ((2 * D + 63) // 64 * 64) // 2))
    rows.append(('u1 (LN1+mod)', rel(m.debug_buffer('u', torch.bfloat16).float().cpu().numpy()[:M * ldD].reshape(M, ldD)[:, :D],
                                     T[f'{pfx}:u1'].reshape(M, D))))
    run(4)
    qkv = f32('qkv', (M, 3 * D))
    u1 = T[f'{pfx}:u1'].reshape(M, D)
    wq = np.concatenate([sd[f'{pfx}.attn.to_q.weight'], sd[f'{pfx}.attn.to_k.weight'], sd[f'{pfx}.attn.to_v.weight']])
    rows.append(('qkv gemm', rel(qkv, u1 @ wq.T)))
    run(6)
    q = b16('q', (B, H, Lp, DQK))[:, :, :L, :dh]
    k = b16('k', (B, H, Lp, DQK))[:, :, :L, :dh]
    vt = b16('v', (B, H, Lp, DV))[:, :, :L, :dh]
    rows.append(('q (headLN+rope)', rel(q, T[f'{pfx}:sq'])))
    rows.append(('k (headLN+rope)', rel(k, T[f'{pfx}:sk'])))
    rows.append(('v', rel(vt, T[f'{pfx}:sv'])))
    run(7)
    ao = b16('ao', (M, ldD))[:, :D]
    rows.append(('self-attn out', rel(ao, T[f'{pfx}:so'].reshape(M, D))))
    run(9)
    rows.append(('h after self-attn', rel(f32('h', (M, D)), T[f'{pfx}:h_attn'].reshape(M, D))))
    rows.append(('u2 (LN2)', rel(m.debug_buffer('u', torch.bfloat16).float().cpu().numpy()[:M * ldD].reshape(M, ldD)[:, :D],
                                 T[f'{pfx}:u2'].reshape(M, D))))
    run(11)
    q2 = b16('q', (B, H, Lp, DQK))[:, :, :L, :dh]
    rows.append(('cross q', rel(q2, T[f'{pfx}:xq'])))
    Lcp = (Lc + 63) // 64 * 64
    kc = b16('kc', (cfg['depth'] + 1, B, H, Lcp, DQK))[0, :, :, :Lc, :dh]
    vct = b16('vc', (cfg['depth'] + 1, B, H, Lcp, DV))[0, :, :, :Lc, :dh]
    rows.append(('cross k (ctx path)', rel(kc, T[f'{pfx}:xk'])))
    rows.append(('cross v', rel(vct, T[f'{pfx}:xv'])))
    run(12)
    ao = b16('ao', (M, ldD))[:, :D]
    rows.append(('cross-attn out', rel(ao, T[f'{pfx}:xo'].reshape(M, D))))
    run(14)
    rows.append(('h after cross', rel(f32('h', (M, D)), T[f'{pfx}:h_cross'].reshape(M, D))))
    rows.append(('u3 (LN3+mod)', rel(m.debug_buffer('u', torch.bfloat16).float().cpu().numpy()[:M * ldD].reshape(M, ldD)[:, :D],
                                     T[f'{pfx}:u3'].reshape(M, D))))
    run(15)
    act = b16('act', (M, inner))
    rows.append(('geglu act', rel(act, T[f'{pfx}:act'].reshape(M, inner))))
    run(17)
    Mp = (M + 127) // 128 * 128
    skips = f32('skips', (cfg['depth'] // 2, Mp, D))
    rows.append(('h after block0 (skip slot 0)', rel(skips[0, :M], T['in0'].reshape(M, D))))
    out = run(0)
    rows.append(('final pred', rel(out.cpu().numpy(), ref)))
    rows.append(('launches', m.last_launch_count))
    for name, v in rows:
        print(f'{name:32s} {v:.3e}' if isinstance(v, float) else f'{name:32s} {v}')
    return rows


if __name__ == '__main__':
    size = sys.argv[1] if len(sys.argv) > 1 else 'xs'
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    main(size, L)
