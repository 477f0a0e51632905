"""Run ONE kernel configuration a few times (for rocprofv3 --pmc runs).  python tests/probe_one.py gemm <variant> M N K [splitk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezaudio_amd import _lib  # noqa: E402

lib = _lib.load()
v, M, N, K = (int(x) for x in sys.argv[2:6])
sk = int(sys.argv[6]) if len(sys.argv) > 6 else 1
A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
W = (torch.randn((N + 255) // 256 * 256, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
bias = torch.zeros(N, device='cuda')
out = torch.empty(8 * ((M + 255) // 256 * 256) * max(N, 1152), device='cuda')
epi = v % 4
for _ in range(10):
    lib.ezdit_test_gemm(None, v, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N // 2 if epi == 2 else N, M, N, K, sk, None)
torch.cuda.synchronize()
