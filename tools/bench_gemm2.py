"""Micro-benchmark + correctness check of the large-tile GEMM (k_gemm2, tile ids 40-42) against the round-1 tiles, through the
C ABI test hook (diagnostic, GPU only):   python tools/bench_gemm2.py [M ...]
Weights rotate through a pool larger than the L2s so every launch streams them like the sampler does."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
dev = 'cuda'
NAMES = {9: '128x128 8w r3', 13: '128x288 12w r3', 10: '256x128 8w r3', 11: '256x256 8w r2', 5: '128x64 r2', 40: 'g2 256x256', 41: 'g2 192x256',
         42: 'g2 256x128'}


def timeit(fn, iters=40, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def run(M, N, K, epi, tiles, splits=(1,)):
    nW = 6
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    Np = (N + 287) // 288 * 288 + 288
    Ws = [(torch.randn(Np, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16) for _ in range(nW)]
    bias = torch.zeros(N, device=dev)
    Mp = (M + 255) // 256 * 256
    fl = 2.0 * M * N * K
    ref = A.float() @ Ws[0][:N].float().T
    if epi == 2:   # rows interleaved 8 value / 8 gate
        r = ref.view(M, N // 16, 2, 8)
        ref = (r[:, :, 0] * torch.nn.functional.gelu(r[:, :, 1])).reshape(M, N // 2)
    rows = []
    for tile in tiles:
        for sk in splits:
            if epi != 1 and sk != 1:
                continue
            ldo = N // 2 if epi == 2 else N
            out = torch.zeros(max(sk, 1) * Mp * ldo + 64, device=dev, dtype=torch.bfloat16 if epi == 2 else torch.float32)
            v = tile * 4 + epi

            def f(i, v=v, out=out, ldo=ldo, sk=sk):
                rc = lib.ezdit_test_gemm(None, v, A.data_ptr(), K, Ws[i % nW].data_ptr(), K, bias.data_ptr(), out.data_ptr(), ldo, M, N, K, sk, None)
                assert rc == 0, lib.ezdit_last_error()
            f(0)
            torch.cuda.synchronize()
            if epi == 1:
                got = out[:sk * ((M + 127) // 128 * 128) * ldo].view(sk, -1, ldo).sum(0)[:M]
            else:
                got = out[:M * ldo].view(M, ldo).float()
            err = float((got - ref).norm() / ref.norm())
            us = timeit(f)
            rows.append((us, f'{NAMES.get(tile, tile)}{" sk" + str(sk) if epi == 1 else ""}: {us:.1f}us {fl / us / 1e6:.0f}TF err {err:.1e}'))
    rows.sort()
    print(f'M={M} N={N} K={K} epi{epi}: ' + ' | '.join(r for _, r in rows), flush=True)


if __name__ == '__main__':
    Ms = [int(x) for x in sys.argv[1:]] or [4000, 1000]
    for M in Ms:
        run(M, 9216, 1152, 2, (13, 40, 41, 42))
        run(M, 3456, 1152, 0, (9, 10, 40, 41, 42))
        run(M, 1152, 1152, 1, (9, 5, 40, 42), (1, 2, 3))
        run(M, 1152, 2304, 1, (9, 40, 42), (1, 2, 3))
        run(M, 1152, 4608, 1, (9, 40, 42), (1, 3, 4, 6))
