"""VAE decoder / encoder timing on the GPU (diagnostic, not a test): python tools/bench_vae.py [L]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import vae as V            # noqa: E402  (synthetic weights only)
from ezaudio_amd.vae import OobleckDecoder, OobleckEncoder  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
cfg = dict(V.VAE_DEFAULT)
dec = OobleckDecoder(device='cuda').load_state_dict(V.make_vae_state_dict(cfg, 6))
enc = OobleckEncoder(device='cuda').load_state_dict(V.make_vae_state_dict(cfg, 6, encoder=True))
z = torch.randn(1, 128, L, device='cuda')
wav = torch.randn(1, 1, L * 480, device='cuda') * 0.3


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for tile in (2, 5, 6, 7, 8, 9, 0, 4):
    dec.tile = enc.tile = tile
    try:
        td = timeit(lambda: dec(z))
        te = timeit(lambda: enc(wav))
    except Exception as e:  # noqa: BLE001
        print(tile, 'failed', e)
        continue
    print(f'tile {tile}: decode {td:.3f} ms ({dec.flops(L) / td / 1e9:.1f} TFLOP/s)  encode {te:.3f} ms', flush=True)
