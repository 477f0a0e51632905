"""Experiment: one B=2 (CFG pair) sampler vs two concurrent B=1 samplers on two streams (diagnostic)."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import MaskDiT, DDIMScheduler                       # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.sampler import LatentSampler                         # noqa: E402
from ezaudio_amd.weights import random_state_dict                     # noqa: E402

params = load_yaml_with_includes(configs['s3_xl']['config'])
cfg = params['model']
sd = random_state_dict(cfg, seed=0)
L, Lc, n = 500, 100, 50
g = torch.Generator().manual_seed(1)


def make(guidance):
    unet = MaskDiT(device='cuda', **cfg)
    unet.load_state_dict(sd)
    text = torch.randn(1, Lc, cfg['context_dim'], generator=g)
    mask = torch.zeros(1, Lc, dtype=torch.bool); mask[:, :12] = True
    um = torch.zeros(1, Lc, dtype=torch.bool); um[:, :1] = True
    init = torch.randn(1, 128, L, generator=g)
    noise = torch.randn(n, 1, 128, L, generator=g)
    smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
    smp.prepare(text, mask, torch.randn(1, Lc, cfg['context_dim'], generator=g), um, init, noise, guidance, 0.75 if guidance else 0.0, n, 1.0)
    return smp


def timeit(samplers, reps=3):
    best = 1e9
    for _ in range(reps + 1):
        for s in samplers:
            with torch.cuda.stream(s.stream):
                s.unet.lib.ezdit_set_step(s.unet._h, 0, C.c_void_p(s.stream.cuda_stream))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for s in samplers:
            s.run(n, use_graph=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


pair = make(5.0)
print(f'one sampler, CFG pair (B=2, M=1000): {timeit([pair]):.3f} ms/step', flush=True)
a, b = make(None), make(None)
print(f'one sampler, single row (B=1, M=500): {timeit([a]):.3f} ms/step', flush=True)
print(f'two concurrent single-row samplers on two streams: {timeit([a, b]):.3f} ms per pair of steps', flush=True)
