"""GEMM micro-benchmark under in-situ conditions (diagnostic): every launch reads a weight matrix that is cold in every
cache (pool > Infinity Cache), an activation that a different kernel has just written, and follows a different kernel.
    python tools/bench_cold.py [shape ...]      shapes: proj skip mlpout qkv geglu q2"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
WARM = 'warm' in sys.argv
dev = 'cuda'
M = 1000
SHAPES = {'proj': (1152, 1152, 1), 'skip': (1152, 2304, 1), 'mlpout': (1152, 4608, 1), 'qkv': (3456, 1152, 0), 'q2': (1152, 1152, 0),
          'geglu': (9216, 1152, 2)}
NAMES = {0: '128x128 2x2 r4', 1: '128x64 2x2 r3', 2: '128x128 2x2 r2', 3: '128x64 2x2 r4', 4: '128x128 2x2 r3', 5: '128x64 2x2 r2',
         6: '128x64 4x1 r2', 7: '128x128 4x2 r2', 8: '256x128 r2', 9: '128x128 4x2 r3', 12: '128x288 r2', 13: '128x288 r3',
         14: '128x64 4x1 r3', 15: '128x64 4x1 r4', 16: '64x64 r4', 17: '64x128 r3', 18: '128x64 4x1 r6', 19: '64x64 r8',
         20: '128x64 4x1 r5', 21: '64x64 r5', 22: '128x128 16w r3', 23: '128x128 16w r2', 24: '128x64 8w r3', 25: '128x64 8w r4',
         26: '128x128 16w r4', 27: '256x128 16w r2', 28: '128x256 16w r2', 29: '128x288 6w r3', 30: '128x288 6w r2', 31: '128x192 8w r3', 32: '128x192 8w r2'}


def run(name, configs, iters=192, pad=0):
    N, K, epi = SHAPES[name]
    ld = K + pad
    nW = 1 if WARM else max(8, int(420e6 / (N * K * 2)) + 1)
    rows = (N + 287) // 288 * 288 + 288
    Wp = [(torch.randn(rows, ld, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(nW)]
    src = torch.randn(M, K, device=dev)
    Ab = [torch.zeros(M + 24, ld, device=dev, dtype=torch.bfloat16) for _ in range(4)]
    bias = torch.zeros(N, device=dev)
    Mp = 1024
    outs = [torch.empty(6 * Mp * max(N if epi != 2 else N // 2, 1152), device=dev) for _ in range(2)]
    st = None

    def producer(i):      # stands for the row kernel that writes the GEMM input right before it
        lib.ezvae_snake_bf16(src.data_ptr(), K, None, None, Ab[i % 4].data_ptr(), ld, M, K, st)

    def loop(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(16):
            fn(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    base = loop(producer)
    fl = 2.0 * M * N * K
    res = []
    for tile, sk in configs:
        v = tile * 4 + epi
        ldo = N // 2 if epi == 2 else N

        def both(i):
            producer(i)
            rc = lib.ezdit_test_gemm(None, v, Ab[i % 4].data_ptr(), ld, Wp[i % nW].data_ptr(), ld, bias.data_ptr(), outs[i % 2].data_ptr(),
                                     ldo, M, N, K, sk, st)
            assert rc == 0
        us = loop(both) - base
        res.append((us, f'{NAMES.get(tile, tile)} S{sk}'))
    res.sort()
    print(f'{name:7s} {"warm" if WARM else "cold"} pad={pad} N={N} K={K} (producer {base:.1f} us): ' + ' | '.join(f'{n}: {us:.1f}us {fl / us / 1e6:.0f}TF' for us, n in res), flush=True)


which = [a for a in sys.argv[1:] if not a.startswith('pad=') and a != 'warm'] or ['proj', 'mlpout', 'qkv', 'q2', 'skip', 'geglu']
pads = [int(a[4:]) for a in sys.argv[1:] if a.startswith('pad=')] or [0]
# (tile id, split-K) candidates per shape; ids as in csrc/gemm.hip (round 1's experimental ids 11, 14-24, 26-32 are retired)
SHORT = {'proj': [(9, 3), (9, 2), (5, 2), (4, 3), (7, 3)], 'skip': [(9, 3), (9, 2), (4, 3), (7, 3)], 'mlpout': [(9, 3), (9, 4), (5, 4), (4, 3)],
         'qkv': [(9, 1), (10, 1), (25, 1), (8, 1)], 'q2': [(25, 1), (9, 1), (3, 1)], 'geglu': [(12, 1), (13, 1), (9, 1)]}
for w in which:
    for pad in pads:
        run(w, SHORT[w], pad=pad)
