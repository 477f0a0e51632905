"""Does the K-tile time depend on how scattered a tile's rows are in memory?  Same GEMM, A either row-major [M][K] or
K-tile-major [K/64][M][64] (a 128-row K tile = 16 KB contiguous) via the conv tap addressing of ezvae_gemm (diagnostic)."""
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
dev = 'cuda'
M, N = 1000, 1152


def bench(K, tile, tiled, n=200, nW=24):
    Ws = [(torch.randn(N + 288, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(nW)]
    As = [torch.randn(K // 64 * 1024 * 64 + 4096, device=dev).to(torch.bfloat16) for _ in range(2)]
    bias = torch.zeros(N, device=dev)
    outs = [torch.empty(1024 * N, device=dev) for _ in range(2)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(n):
                if tiled:   # lda = 64 elements, K tile t at byte offset t * 1024 rows * 128 B
                    rc = lib.ezvae_gemm(As[i % 2].data_ptr(), 64, Ws[i % nW].data_ptr(), K, N + 288, bias.data_ptr(), None, 0,
                                        outs[i % 2].data_ptr(), N, M, N, K, 1, 1024 * 128, tile, st)
                else:
                    rc = lib.ezvae_gemm(As[i % 2].data_ptr(), K, Ws[i % nW].data_ptr(), K, N + 288, bias.data_ptr(), None, 0,
                                        outs[i % 2].data_ptr(), N, M, N, K, 0, 0, tile, st)
                assert rc == 0
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best / n * 1e6


for tile, label in ((25, '128x64 8w r4'), (9, '128x128 8w r3'), (14, '128x64 4w r3')):
    for K in (1152, 4608):
        a, b = bench(K, tile, False), bench(K, tile, True)
        print(f'{label:16s} K={K:5d}: A row-major {a:6.2f} us   A K-tile-major {b:6.2f} us', flush=True)
