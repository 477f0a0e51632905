"""GPU diagnostic (not a pytest): the LayerNorm-algebra path (zfuse 1) against the split-K + row-kernel path (zfuse 0), stage by stage
through the ezdit_debug_stop_after hook.   python tools/diag_zfuse.py [xs64|xs|s] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.weights import make_inputs, make_state_dict, model_config  # noqa: E402
from ezaudio_amd import MaskDiT  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main(size='xs64', L=96, Lc=20, t=499):
    cfg = model_config(size)
    sd = make_state_dict(cfg, 1)
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, n_valid=(7, 1), seed=11)
    m = MaskDiT(device='cuda', **cfg)
    m.load_state_dict(sd)
    D = cfg['embed_dim']
    M = 2 * L
    Mp = (M + 127) // 128 * 128
    x = torch.from_numpy(inp['x']).cuda()
    ctx = torch.from_numpy(inp['ctx']).cuda()
    msk = torch.from_numpy(inp['ctx_mask']).cuda()

    def run(z, stop):
        assert m.lib.ezdit_set_option(m._h, b'zfuse', z) == 0
        m.lib.ezdit_debug_stop_after(m._h, stop)
        out, _ = m(x, torch.tensor(t), ctx, context_mask=msk)
        torch.cuda.synchronize()
        h = m.debug_buffer('h', torch.float32, (Mp, D)).cpu().numpy()[:M].copy()
        sk = m.debug_buffer('skips', torch.float32, (Mp, D)).cpu().numpy()[:M].copy()
        act = m.debug_buffer('act', torch.bfloat16, (Mp, 4 * D)).float().cpu().numpy()[:M].copy()
        ao = m.debug_buffer('ao', torch.bfloat16, (Mp, (D + 63) // 64 * 64)).float().cpu().numpy()[:M, :D].copy()
        q = m.debug_buffer('q', torch.bfloat16, None).float().cpu().numpy().copy()
        return h, sk, act, ao, q, out.cpu().numpy()
    # (zfuse 0 launch count, zfuse 1 launch count, label)
    pairs = [(3, 3, 'LN1 of block 0'), (4, 4, 'QKV (plain)'), (5, 5, 'self-attention'), (7, 6, 'attn-out residual'), (8, 7, 'cross-attention (q algebra)'),
             (10, 8, 'cross-out residual'), (11, 9, 'GEGLU (algebra)'), (13, 10, 'MLP-out residual -> skips[0]'), (14, 11, 'QKV of mid block (algebra)'),
             (15, 12, 'self-attention mid')]
    for s0, s1, label in pairs:
        a = run(0, s0)
        b = run(1, s1)
        print(f'{label:36s} h {rel(b[0], a[0]):.3e}  skips0 {rel(b[1], a[1]):.3e}  act {rel(b[2], a[2]):.3e}  ao {rel(b[3], a[3]):.3e}  q {rel(b[4], a[4]):.3e}', flush=True)
    m.lib.ezdit_debug_stop_after(m._h, 0)
    a = run(0, 0)[5]
    b = run(1, 0)[5]
    print('final prediction zfuse 1 vs 0:', rel(b, a))


if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['xs64']))
