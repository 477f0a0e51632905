"""In-situ A/B of the GEMM tuning knobs on the real sampler (diagnostic): python tools/ab_sweep.py [size] [prompts] name=v1,v2 ..."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import MaskDiT, DDIMScheduler                       # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.sampler import LatentSampler                         # noqa: E402
from ezaudio_amd.weights import random_state_dict                     # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else 'xl'
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
combos = [a for a in sys.argv[3:] if '+' in a or a.count('=') > 1]      # "a=1+b=2": one measurement with several knobs set
sweeps = [a.split('=') for a in sys.argv[3:] if a not in combos]
params = load_yaml_with_includes(configs['s3_' + size]['config'])
cfg = params['model']
unet = MaskDiT(device='cuda', **cfg)
unet.load_state_dict(random_state_dict(cfg, seed=0))
L, Lc, n = 500, 100, 50
g = torch.Generator().manual_seed(1)
text = torch.randn(P, Lc, cfg['context_dim'], generator=g)
mask = torch.zeros(P, Lc, dtype=torch.bool); mask[:, :12] = True
um = torch.zeros(P, Lc, dtype=torch.bool); um[:, :1] = True
init = torch.randn(P, 128, L, generator=g)
noise = torch.randn(n, P, 128, L, generator=g)
smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
smp.prepare(text, mask, torch.randn(P, Lc, cfg['context_dim'], generator=g), um, init, noise, 5.0, 0.75, n, 1.0)
init_dev = init.cuda()


def measure(reps=3):
    best = 1e9
    for _ in range(reps + 1):
        with torch.cuda.stream(smp.stream):
            smp.latents.copy_(init_dev, non_blocking=True)
            unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))
        torch.cuda.synchronize()
        t = time.perf_counter()
        smp.run(n, use_graph=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


def setopt(name, v):
    rc = unet.lib.ezdit_set_option(unet._h, name.encode(), int(v))
    assert rc == 0, (name, v)


print(f'baseline: {measure():.3f} ms/step', flush=True)
for name, vals in sweeps:
    vals = vals.split(',')
    for v in vals:
        setopt(name, v)
        try:
            ms = measure()
            ok = bool(torch.isfinite(smp.latents).all())
        except Exception as e:  # noqa: BLE001
            ms, ok = float('nan'), repr(e)
        print(f'{name}={v}: {ms:.3f} ms/step finite={ok} |latents|={float(smp.latents.abs().mean()):.6f}', flush=True)
    setopt(name, vals[0])       # first value listed is the one to restore (list the default first)

for combo in combos:
    pairs = [kv.split('=') for kv in combo.split('+')]
    for k, v in pairs:
        setopt(k, v)
    try:
        ms = measure()
        ok = bool(torch.isfinite(smp.latents).all())
    except Exception as e:  # noqa: BLE001
        ms, ok = float('nan'), repr(e)
    print(f'{combo}: {ms:.3f} ms/step finite={ok}', flush=True)
