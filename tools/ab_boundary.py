"""How much does one dependent kernel boundary cost inside a replayed hipGraph on this box? (diagnostic)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
x = torch.zeros(1 << 20, device='cuda')
xb = torch.zeros(1 << 20, device='cuda', dtype=torch.bfloat16)


def capture(n, L, C):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(n):
                lib.ezvae_snake_bf16(x.data_ptr(), C, None, None, xb.data_ptr(), C, L, C, st)
    return g


for L, C, label in ((1, 4, '1 thread'), (256, 256, '64 WGs'), (1024, 256, '256 WGs'), (1000, 1152, '1125 WGs, 2.3 MB out')):
    n = 400
    g = capture(n, L, C)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    print(f'{label:24s}: {best / n * 1e6:.2f} us per kernel node ({n} dependent nodes, one replay)')
