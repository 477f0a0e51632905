"""Per-step duration of the first N graph replays after prepare (diagnostic: why `bench.py --steps 20 --warmup 5` reads slower than --steps 200 --warmup 50).
    python tools/ramp_probe.py [n_steps]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from bench import model_section  # noqa: E402
from ezaudio_amd import MaskDiT, DDIMScheduler  # noqa: E402
from ezaudio_amd.sampler import LatentSampler  # noqa: E402
from ezaudio_amd.weights import random_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
params = model_section('xl')
cfg = params['model']
unet = MaskDiT(device='cuda', **cfg)
unet.load_state_dict(random_state_dict(cfg, seed=0))
L, Lc, n = 500, 100, 50
g = torch.Generator().manual_seed(1)
text = torch.randn(1, Lc, cfg['context_dim'], generator=g)
mask = torch.zeros(1, Lc, dtype=torch.bool); mask[:, :12] = True
um = torch.zeros(1, Lc, dtype=torch.bool); um[:, :1] = True
init = torch.randn(1, 128, L, generator=g)
noise = torch.randn(n, 1, 128, L, generator=g)
smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
smp.prepare(text, mask, torch.randn(1, Lc, cfg['context_dim'], generator=g), um, init, noise, 5.0, 0.75, n, 1.0)
torch.cuda.synchronize()
if '--sleep' in sys.argv:
    time.sleep(2.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record(smp.stream)
for i in range(N):
    if i % n == 0:
        with torch.cuda.stream(smp.stream):
            unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))
    smp.run(1, use_graph=True)
    ev[i + 1].record(smp.stream)
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print('per-step ms:', ' '.join(f'{x:.2f}' for x in ms))
for a, b in ((0, 5), (5, 25), (25, 50), (50, N)):
    if b <= N:
        print(f'steps [{a},{b}): mean {sum(ms[a:b]) / (b - a):.3f} ms')


def span(label, fn, k):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(smp.stream):
        unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))
    t0 = time.perf_counter()
    a.record(smp.stream)
    fn()
    b.record(smp.stream)
    torch.cuda.synchronize()
    print(f'{label}: event {a.elapsed_time(b) / k:.3f} ms/step, wall {(time.perf_counter() - t0) * 1e3 / k:.3f}')


init_dev = init.cuda()


def with_copy():
    with torch.cuda.stream(smp.stream):
        smp.latents.copy_(init_dev, non_blocking=True)
    smp.run(20, use_graph=True)


def with_kernel_copy():
    with torch.cuda.stream(smp.stream):
        torch.add(init_dev, 0.0, out=smp.latents)     # an elementwise kernel instead of hipMemcpyAsync
    smp.run(20, use_graph=True)


for rep in range(2):
    span('latents.copy_ + one call of 20', with_copy, 20)
    span('kernel copy + one call of 20', with_kernel_copy, 20)
    span('one call of 20', lambda: smp.run(20, use_graph=True), 20)
    span('20 calls of 1', lambda: [smp.run(1, use_graph=True) for _ in range(20)], 20)
    span('one call of 50', lambda: smp.run(50, use_graph=True), 50)
    span('eager 20', lambda: smp.run(20, use_graph=False), 20)
