"""GPU diagnostic (not a pytest): in-kernel cycle stamps of chosen launches INSIDE the denoiser forward (eager launches, XL, one prompt),
i.e. with the instruction and data caches in the state the step leaves them in -- next to the same kernels' warm microbenchmark numbers
(tools/microbench/{gemm,attn}_bench.cpp).      python tools/diag_stamps.py [xl|l]

The 'trace_launches' option names the launches of a forward; 'stamp_launch' = i makes launch i write its stamps into the buffer registered
with ezdit_debug_gemm_timestamps (k_gemm_pp: start / prologue done / K loop done / stores done; k_attn: start / first tile staged / tile loop
done / stores done / partials parked).
"""
import ctypes as C
import os
import re
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import model_section  # noqa: E402
from ezaudio_amd import MaskDiT  # noqa: E402
from ezaudio_amd.weights import random_state_dict  # noqa: E402


def main(size='xl'):
    params = model_section(size)
    cfg = params['model']
    L = 10 * params['autoencoder']['latent_sr']
    Lc = params['text_encoder']['max_length']
    dev = torch.device('cuda', 0)
    m = MaskDiT(device=dev, **cfg)
    m.load_state_dict(random_state_dict(cfg, seed=1234))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, cfg['out_chans'], L, generator=g).to(dev)
    ctx = torch.randn(2, Lc, cfg['context_dim'], generator=g).to(dev)
    msk = torch.zeros(2, Lc, dtype=torch.bool)
    msk[0, :23] = True
    msk[1, :1] = True
    msk = msk.to(dev)
    t = torch.tensor(499)

    def fwd():
        m(x, t, ctx, context_mask=msk)

    for kv in os.environ.get('EZ_OPTS', '').split():   # e.g. EZ_OPTS='attn_xk2=0 gemm_pp=0'
        k, v = kv.split('=')
        assert m.lib.ezdit_set_option(m._h, k.encode(), int(v)) == 0, kv
    fwd()
    torch.cuda.synchronize()
    # names of the launches, through the library's stderr trace
    with tempfile.TemporaryFile(mode='w+') as tf:
        sys.stderr.flush()
        saved = os.dup(2)
        os.dup2(tf.fileno(), 2)
        assert m.lib.ezdit_set_option(m._h, b'trace_launches', 1) == 0
        fwd()
        torch.cuda.synchronize()
        assert m.lib.ezdit_set_option(m._h, b'trace_launches', 0) == 0
        os.dup2(saved, 2)
        os.close(saved)
        tf.seek(0)
        names = {}
        for ln in tf.read().splitlines():
            mm = re.match(r'launch (\d+) (.*)', ln)
            if mm:
                names[int(mm.group(1))] = mm.group(2)
    n_launch = len(names)
    print(f'{n_launch} launches per forward', flush=True)
    NWG = 4096
    ts = torch.zeros(NWG * 8, dtype=torch.int64, device=dev)
    # one launch of each kind from the middle of the network (the mid block's neighbourhood), plus the first block's for contrast
    want = os.environ.get('STAMP_KERNELS', 'k_gemm (QKV);k_attn (self);k_attn (cross);k_gemm (GEGLU)').split(';')
    mid = n_launch // 2
    picks = []
    for w in want:
        idx = [i for i, nm in names.items() if nm == w]
        if idx:
            picks.append(min(idx, key=lambda i: abs(i - mid)))
            picks.append(idx[0])
    for i in sorted(set(picks)):
        acc = None
        reps = 6
        for _ in range(reps):
            ts.zero_()
            m.lib.ezdit_debug_gemm_timestamps(C.c_void_p(ts.data_ptr()), NWG)
            assert m.lib.ezdit_set_option(m._h, b'stamp_launch', i) == 0
            fwd()
            torch.cuda.synchronize()
            m.lib.ezdit_debug_gemm_timestamps(None, 0)
            raw = ts.cpu().numpy().reshape(NWG, 8)
            okr = (raw[:, 0] > 0) & (raw[:, 3] > 0)
            if not okr.any():
                break
            # cycle counters are per workgroup-local clock domain; [6] / [7] hold the 100 MHz device-wide clock at the start and the end
            rt0, rt1 = raw[okr, 6] & ((1 << 48) - 1), raw[okr, 7]   # ([6]'s top 16 bits name the CU: common.h ez_stamp_start)
            span = float(rt1.max() - rt0.min()) * 10.0      # ns: first workgroup start -> last workgroup end
            skew = float(rt0.max() - rt0.min()) * 10.0      # ns: first -> last workgroup start
            mine = float((rt1 - rt0).mean()) * 10.0         # ns: mean lifetime of a workgroup
            a = raw[okr].astype(np.float64) - float(raw[okr, 0].min())
            if names[i].startswith('k_attn'):
                row = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 4] - a[:, 2], a[:, 3] - a[:, 4]]
                lab = ['first tile staged', 'tile loop', 'partials parked', 'merge + stores']
                if (a[:, 5] > 0).all():   # fused projection: K loop done, queries normalised
                    row += [a[:, 5] - a[:, 0], a[:, 6] - a[:, 5]]
                    lab += ['(projection K loop', 'reduce + LayerNorm)']
            else:
                row = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]]
                lab = ['prologue', 'K loop', 'epilogue + stores']
                if 'un-split' in names[i] and (raw[okr, 4] > 0).all() and (raw[okr, 5] > 0).all():   # k_gemm_ks: [4] = partials parked + barrier, [5] packs (sums complete, stores issued) relative to [0]
                    base0 = raw[okr, 0].astype(np.float64) - float(raw[okr, 0].min())
                    t_sum = base0 + (raw[okr, 5] & 0xffffffff).astype(np.float64)
                    t_iss = base0 + (raw[okr, 5] >> 32).astype(np.float64)
                    row += [a[:, 4] - a[:, 2], t_sum - a[:, 4], t_iss - t_sum, a[:, 3] - t_iss]
                    lab += ['(epilogue: operand wait + park + barrier', 'partial sums', 'arithmetic + store issue', 'stores landed)']
                elif (raw[okr, 4] > 0).all() and (raw[okr, 5] > 0).all():   # epilogue marks ([4], [5]): GEGLU: ring dead / parked; fused QKV: exchange done / arithmetic done
                    row += [a[:, 4] - a[:, 2], a[:, 5] - a[:, 4], a[:, 3] - a[:, 5]]
                    lab += ['(epilogue: -> mark 4', 'mark 4 -> 5', 'mark 5 -> end)']
            if not names[i].startswith('k_attn') and _ == 0:
                e = np.sort(row[2])
                print(f'    epilogue over workgroups: min {e[0]:.0f} p25 {e[len(e) // 4]:.0f} median {e[len(e) // 2]:.0f} p75 {e[3 * len(e) // 4]:.0f} max {e[-1]:.0f}; '
                      f'prologue min {row[0].min():.0f} max {row[0].max():.0f}; loop min {row[1].min():.0f} max {row[1].max():.0f}')
            v = np.array([r.mean() for r in row] + [mine, span, skew, len(a)])
            nlab = len(lab)
            acc = v if acc is None else acc + v
        assert m.lib.ezdit_set_option(m._h, b'stamp_launch', -1) == 0
        if acc is None:
            print(f'launch {i:3d} {names[i]:24s}: no stamps (kernel without stamp support)')
            continue
        acc /= reps
        parts = ' | '.join(f'{l} {x:.0f}' for l, x in zip(lab, acc[:nlab]))
        print(f'launch {i:3d} {names[i]:24s}: {parts} | 100 MHz clock: workgroup lifetime {acc[-4] / 1e3:.2f} us, first start -> last end {acc[-3] / 1e3:.2f} us, start skew {acc[-2] / 1e3:.2f} us  ({acc[-1]:.0f} WGs)', flush=True)
    print('note: stamps are s_memtime ticks (the shader clock counter): compare with the warm numbers of gemm_bench / attn_bench, same unit')


if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['xl']))
