"""GPU diagnostic: which keys does the attention kernel weight wrongly?  Uniform scores (q = 0) and V = one-hot(key)
-> O[q, d] = (number of keys mapped to channel d) / Lk."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezaudio_amd import MaskDiT  # noqa: E402
from oracle.weights import model_config  # noqa: E402


def run(size, Lq, Lk, qscale=0.0):
    cfg = model_config(size)
    m = MaskDiT(device='cuda', **cfg)
    H, D = cfg['num_heads'], cfg['embed_dim']
    dh = D // H
    DQK, DV = (64, 64) if dh == 64 else (80, 96)
    B = 1
    Lqp, Lkp = (Lq + 63) // 64 * 64, (Lk + 63) // 64 * 64
    g = torch.Generator().manual_seed(0)
    q = qscale * torch.randn(B, H, Lq, dh, generator=g)
    k = torch.randn(B, H, Lk, dh, generator=g)
    v = torch.zeros(B, H, Lk, dh)
    for key in range(Lk):
        v[:, :, key, key % dh] = 1.0 + key // dh   # channel d collects keys d, d+dh, ... with weights 1, 2, ...
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    qp = torch.zeros(B, H, Lqp, DQK, dtype=torch.bfloat16); qp[:, :, :Lq, :dh] = q
    kp = torch.zeros(B, H, Lkp, DQK, dtype=torch.bfloat16); kp[:, :, :Lk, :dh] = k
    vt = torch.zeros(B, H, DV, Lkp, dtype=torch.bfloat16); vt[:, :, :dh, :Lk] = v.transpose(2, 3)
    ldD = (D + 63) // 64 * 64
    out = torch.zeros(B * Lq, ldD, dtype=torch.bfloat16, device='cuda')
    qd, kd, vd = qp.cuda(), kp.cuda(), vt.cuda()
    rc = m.lib.ezdit_test_attention(m._h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), None, out.data_ptr(), B, Lq, Lk, Lqp, Lkp, None)
    torch.cuda.synchronize()
    s = (q.double() @ k.double().transpose(2, 3)) * dh ** -0.5
    ref = (torch.softmax(s, -1) @ v.double()).transpose(1, 2).reshape(B * Lq, D)
    got = out.float().cpu()[:, :D].double()
    err = (got - ref).abs()
    print(f'{size} Lq={Lq} Lk={Lk} qscale={qscale}: rc={rc} max err {err.max():.4f}')
    if qscale == 0.0:
        # per-key weight = O[q, key % dh] * Lk / (1 + key // dh) -> should be 1 for every key
        w = got[:, :dh] * Lk   # head 0
        bad_rows = (err.max(dim=1).values > 0.01).nonzero().flatten().tolist()
        print('  rows with error:', bad_rows[:20], '...' if len(bad_rows) > 20 else '')
        r = bad_rows[0] if bad_rows else 0
        print('  row', r, 'channel sums x Lk (want sum_j (1+j) over keys d + j*dh):')
        print('   got ', np.round(w[r].numpy()[:24], 2))
        print('   want', np.round((ref[r, :dh] * Lk).numpy()[:24], 2))


if __name__ == '__main__':
    run('xs', 96, 96)
    run('xs', 96, 64)
    run('xs', 64, 32)
    run('xs', 96, 96, 1.0)
    run('xs64', 96, 96)
    run('xs', 500, 500)


def run_keys(size, Lk, col=0):
    """Per-key softmax weights: q = e_col, K[key, col] = small integer pattern, V = one-hot(key) (Lk <= dh)."""
    cfg = model_config(size)
    m = MaskDiT(device='cuda', **cfg)
    H, D = cfg['num_heads'], cfg['embed_dim']
    dh = D // H
    DQK, DV = (64, 64) if dh == 64 else (80, 96)
    B, Lq = 1, 64
    Lqp, Lkp = 64, (Lk + 63) // 64 * 64
    q = torch.zeros(B, H, Lq, dh); q[..., col] = 4.0
    k = torch.zeros(B, H, Lk, dh)
    for key in range(Lk):
        k[:, :, key, col] = float((key * 7) % 5) - 2.0
    v = torch.zeros(B, H, Lk, dh)
    for key in range(Lk):
        v[:, :, key, key] = 1.0
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    qp = torch.zeros(B, H, Lqp, DQK, dtype=torch.bfloat16); qp[:, :, :Lq, :dh] = q
    kp = torch.zeros(B, H, Lkp, DQK, dtype=torch.bfloat16); kp[:, :, :Lk, :dh] = k
    vt = torch.zeros(B, H, DV, Lkp, dtype=torch.bfloat16); vt[:, :, :dh, :Lk] = v.transpose(2, 3)
    ldD = (D + 63) // 64 * 64
    out = torch.zeros(B * Lq, ldD, dtype=torch.bfloat16, device='cuda')
    qd, kd, vd = qp.cuda(), kp.cuda(), vt.cuda()
    m.lib.ezdit_test_attention(m._h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), None, out.data_ptr(), B, Lq, Lk, Lqp, Lkp, None)
    torch.cuda.synchronize()
    s = (q.double() @ k.double().transpose(2, 3)) * dh ** -0.5
    ref = torch.softmax(s, -1)[0, 0, 0]            # weights of query 0, head 0
    got = out.float().cpu()[0, :Lk].double()
    bad = ((got - ref).abs() > 0.002).nonzero().flatten().tolist()
    print(f'{size} Lk={Lk} col={col}: keys with wrong weight: {bad}')
    if bad:
        print('   got ', np.round(got[bad[:12]].numpy(), 4), '\n   want', np.round(ref[bad[:12]].numpy(), 4))


if __name__ == '__main__':
    for col in (0, 7, 8, 40, 63, 71):
        run_keys('xs', 64, col)
    for col in (0, 8, 63):
        run_keys('xs64', 64, col)
    run_keys('xs', 20, 3)
