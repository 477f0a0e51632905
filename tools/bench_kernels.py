"""GPU micro-benchmarks of the kernel families through the C ABI test hooks (not a pytest).
    python tools/bench_kernels.py gemm|attn|all      (large-tile kernel: tools/bench_gemm2.py)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezaudio_amd import _lib, build  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def bench_gemm(lib):
    dev = 'cuda'
    names = {0: '128x128 r4', 1: '128x64 r3', 2: '128x128 r2', 3: '128x64 r4', 4: '128x128 r3', 5: '128x64 r2',
             6: '128x64 4x1 r2', 7: '128x128 8w r2', 8: '256x128 8w r2', 9: '128x128 8w r3', 12: '128x288 12w r2', 13: '128x288 12w r3'}
    geglu_ok = tuple(range(14))
    shapes = [('qkv', 1000, 3456, 1152), ('geglu-in', 1000, 9216, 1152), ('proj', 1000, 1152, 1152),
              ('skip', 1000, 1152, 2304), ('mlp-out', 1000, 1152, 4608),
              ('qkv B8', 4000, 3456, 1152), ('geglu-in B8', 4000, 9216, 1152), ('proj B8', 4000, 1152, 1152), ('mlp-out B8', 4000, 1152, 4608)]
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn((N + 287) // 288 * 288 + 288, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        Mp = (M + 127) // 128 * 128
        out = torch.empty(8 * Mp * max(N, 1152), device=dev)
        fl = 2.0 * M * N * K
        res = []
        for tile in (2, 5, 6, 7, 12, 13):
            for epi, splits in ((0, [1]), (1, [1, 2, 3, 4, 6])) if N <= 1152 else ((0, [1]), (2, [1])):
                if epi == 2 and tile not in geglu_ok:
                    continue
                for sk in splits:
                    v = tile * 4 + epi
                    ldo = N // 2 if epi == 2 else N
                    us = timeit(lambda: lib.ezdit_test_gemm(None, v, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(),
                                                            ldo, M, N, K, sk, None))
                    res.append((us, f'{names[tile]} epi{epi} split{sk}'))
        res.sort()
        print(f'{name:12s} M={M} N={N} K={K}: ' + ' | '.join(f'{n}: {us:.1f}us {fl/us/1e6:.0f}TF' for us, n in res[:6]))
        print(f'{"":12s} worst: ' + ' | '.join(f'{n}: {us:.1f}us' for us, n in res[-3:]))


def bench_attn(lib):
    from oracle.weights import model_config
    dev = 'cuda'
    for size, B in (('xl', 2), ('l', 2), ('xl', 8)):
        cfg = model_config(size)
        c = _lib.EzditConfig(cfg['embed_dim'], cfg['num_heads'], cfg['depth'], cfg['in_chans'], cfg['out_chans'],
                             cfg['context_dim'], cfg['ada_sola_rank'], float(cfg['ada_sola_alpha']), 4.0, 2048)
        h = C.c_void_p()
        lib.ezdit_create(C.byref(c), C.byref(h))
        H, D = cfg['num_heads'], cfg['embed_dim']
        dh = D // H
        DQK, DV = (64, 64) if dh == 64 else (80, 96)
        for Lq, Lk in ((500, 500), (500, 100)):
            Lqp, Lkp = (Lq + 63) // 64 * 64, (Lk + 63) // 64 * 64
            q = torch.randn(B, H, Lqp, DQK, device=dev).to(torch.bfloat16)
            k = torch.randn(B, H, Lkp, DQK, device=dev).to(torch.bfloat16)
            vt = torch.randn(B, H, DV, Lkp, device=dev).to(torch.bfloat16)
            out = torch.empty(B * Lq, D, dtype=torch.bfloat16, device=dev)
            us = timeit(lambda: lib.ezdit_test_attention(h, q.data_ptr(), k.data_ptr(), vt.data_ptr(), None, out.data_ptr(), B, Lq, Lk, Lqp, Lkp, None))
            fl = 4.0 * B * H * Lq * Lk * dh
            print(f'attn {size} B={B} Lq={Lq} Lk={Lk} dh={dh}: {us:.1f} us  {fl/us/1e6:.0f} TF (algorithmic)')
        lib.ezdit_destroy(h)


if __name__ == '__main__':
    build.build(verbose=False)
    lib = _lib.load()
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('gemm', 'all'):
        bench_gemm(lib)
    if what in ('attn', 'all'):
        bench_attn(lib)
