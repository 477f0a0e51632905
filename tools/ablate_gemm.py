"""Kernel-level timing of the GEMM K loop on MI355X (diagnostic; eager launches through ezdit_test_gemm).

    EZAUDIO_ABLATE=1 python -m ezaudio_amd.build --force     # adds the timing-only ABL variants of k_gemm (results are garbage by construction)
    python tools/ablate_gemm.py            # per-component cost of a K tile: variants without MFMAs / fragment reads / refill / barrier
    python tools/ablate_gemm.py rot        # lockstep vs rotating-phase (ROT) variants of the step's tiles (shipped build)

Per-K-tile costs are the slopes between the K = 1152 and K = 4608 lines of the same variant (short kernels are host-bound in eager mode).
Measured on MI355X, 128x288 tile (GEGLU GEMM): MFMAs 0.51 us, LDS-DMA refill 0.34, fragment reads 0.10, barrier 0.12, loop 0.09 = 1.23 us:
the parts add up, nothing overlaps."""

import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from ezaudio_amd import _lib
lib = _lib.load()
dev = 'cuda'
st = torch.cuda.Stream()

def time_variant(variant, M, N, K, splitk, epi_geglu, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N + 288, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    if epi_geglu:
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev); ldo = N // 2
    else:
        out = torch.empty(max(splitk, 1) * ((M + 127) // 128 * 128) * N, dtype=torch.float32, device=dev); ldo = N
    with torch.cuda.stream(st):
        for _ in range(3):
            rc = lib.ezdit_test_gemm(None, variant, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr() if epi_geglu else None, out.data_ptr(), ldo, M, N, K, splitk, C.c_void_p(st.cuda_stream))
            assert rc == 0, (variant, _lib.last_error() if hasattr(_lib, 'last_error') else rc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            lib.ezdit_test_gemm(None, variant, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr() if epi_geglu else None, out.data_ptr(), ldo, M, N, K, splitk, C.c_void_p(st.cuda_stream))
        e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def ablations():
    names = {0: 'full', 1: 'no MFMA', 2: 'no frag reads', 3: 'no MFMA, no reads', 4: 'no refill', 5: 'no MFMA, no refill', 6: 'no reads, no refill',
             7: 'no MFMA/reads/refill (barriers only)', 8: 'no barrier', 12: 'no refill, no barrier', 15: 'nothing (loop skeleton)', 9: 'no MFMA, no barrier', 11: 'no MFMA, no reads, no barrier'}
    for label, tile, epi, M, N, K, sk in (('GEGLU 128x288 12 waves, K=1152 (18 K tiles)', 13, 2, 1000, 9216, 1152, 1),
                                          ('GEGLU shape, K=4608 (72 K tiles)', 13, 2, 1000, 9216, 4608, 1),
                                          ('residual 128x128 8 waves split-K 3, K=4608 (24 K tiles per slice)', 9, 1, 1000, 1152, 4608, 3),
                                          ('residual 128x128 8 waves split-K 3, K=1152 (6 K tiles per slice)', 9, 1, 1000, 1152, 1152, 3)):
        print(label, flush=True)
        base = None
        for abl in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 15):
            us = time_variant(1000 * (abl << 4) + tile * 4 + epi, M, N, K, sk, epi == 2)
            base = base or us
            nt = K // 64 // sk
            print(f'   {names[abl]:42s} {us:8.2f} us   ({us / nt:6.3f} us per K tile if it were all loop)', flush=True)


def rotating():

    for label, tile, epi, M, N, K, sk in (('GEGLU 128x288 K=1152', 13, 2, 1000, 9216, 1152, 1), ('GEGLU shape K=4608', 13, 2, 1000, 9216, 4608, 1),
                                          ('residual 128x128 split 3 K=4608', 9, 1, 1000, 1152, 4608, 3), ('residual 128x128 split 3 K=1152', 9, 1, 1000, 1152, 1152, 3),
                                          ('residual 128x128 split 3 K=2304', 9, 1, 1000, 1152, 2304, 3)):
        a = time_variant(tile * 4 + epi, M, N, K, sk, epi == 2)
        b = time_variant(4000 + tile * 4 + epi, M, N, K, sk, epi == 2)
        print(f'{label:40s} lockstep {a:7.2f} us   rotating {b:7.2f} us   ({a / b:.3f}x)', flush=True)
    for label, tile, epi, M, N, K, sk in (('residual 128x128 ring 4, split 3 K=4608', 50, 1, 1000, 1152, 4608, 3), ('residual 128x128 ring 4, split 3 K=1152', 50, 1, 1000, 1152, 1152, 3),
                                          ('residual ring 4, split 2 K=4608', 50, 1, 1000, 1152, 4608, 2), ('residual ring 3, split 2 K=4608', 9, 1, 1000, 1152, 4608, 2)):
        a = time_variant(tile * 4 + epi, M, N, K, sk, epi == 2)
        b = time_variant(4000 + tile * 4 + epi, M, N, K, sk, epi == 2)
        print(f'{label:40s} lockstep {a:7.2f} us   rotating {b:7.2f} us   ({a / b:.3f}x)', flush=True)


if __name__ == '__main__':
    rotating() if 'rot' in sys.argv[1:] else ablations()
