"""Fixed cost of one GEMM launch inside a replayed graph: time vs K at fixed tile (diagnostic)."""
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
dev = 'cuda'
M, N = 1000, 1152


def bench(K, tile, epi, sk, n=200, nW=24):
    Ws = [(torch.randn(N + 288, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(nW)]
    As = [torch.randn(M + 24, K, device=dev).to(torch.bfloat16) for _ in range(2)]
    bias = torch.zeros(N, device=dev)
    outs = [torch.empty(4 * 1024 * N, device=dev) for _ in range(2)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(n):
                rc = lib.ezdit_test_gemm(None, tile * 4 + epi, As[i % 2].data_ptr(), K, Ws[i % nW].data_ptr(), K, bias.data_ptr(),
                                         outs[i % 2].data_ptr(), N, M, N, K, sk, st)
                assert rc == 0
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best / n * 1e6


for tile, epi, sk, label in ((25, 0, 1, '128x64 8w r4 F32'), (14, 0, 1, '128x64 4w r3 F32'), (9, 1, 3, '128x128 8w r3 PARTIAL S3'), (9, 1, 1, '128x128 8w r3 PARTIAL S1'),
                             (5, 1, 2, '128x64 4w r2 PARTIAL S2')):
    row = []
    for K in (64, 128, 256, 576, 1152, 2304, 4608):
        if sk > K // 64:
            row.append('   -  ')
            continue
        row.append(f'{bench(K, tile, epi, sk):6.2f}')
    print(f'{label:28s} K=64..4608: ' + ' '.join(row), flush=True)
