"""In-situ A/B of knob COMBINATIONS on the real sampler, with a result check (diagnostic):

    python tools/exp_sweep.py [size] [prompts] [--cn] cfg cfg ...      cfg = "name=v+name=v" (one measurement with those knobs set)

Every configuration runs the 50-step hipGraph loop from the same initial latents; reported per configuration: best ms/step of 3
runs, max |latents - baseline latents| (0.0 = bit-identical), the device status and whether a fused residual GEMM took its
agent-scope path (sync word 1001).  All knobs named anywhere on the command line are reset to their first-seen default (0 unless
listed in DEFAULTS) between configurations."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import MaskDiT, DDIMScheduler                       # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.sampler import LatentSampler                         # noqa: E402
from ezaudio_amd.weights import random_state_dict                     # noqa: E402

DEFAULTS = dict(gemm_pp=3, fuse_q2=1, tile_partial=9, geglu_tile=-1, split18=3, split36=3, split72=3, row_variant=1, xcd_map=1,
                attn_xcd=1, tile_f32=25, qkv_waves9=1, geglu_big=40, tile_partial_big=40, gemm_panel=3, row_affine=1, epi_lds=1, qkv_affine=1, attn_xk2=1)

argv = [a for a in sys.argv[1:] if a != '--cn']
size = argv[0] if len(argv) > 0 else 'xl'
P = int(argv[1]) if len(argv) > 1 else 1
cfgs = argv[2:]
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
hip = C.CDLL('libamdhip64.so')


def sync_word(i):
    ptr, nbytes = C.c_void_p(), C.c_size_t()
    assert unet.lib.ezdit_debug_buffer(unet._h, b'sync', C.byref(ptr), C.byref(nbytes)) == 0
    v = C.c_uint(0)
    hip.hipMemcpy(C.byref(v), C.c_void_p(ptr.value + 4 * i), C.c_size_t(4), C.c_int(2))
    return v.value


def clear_sync_word(i):
    ptr, nbytes = C.c_void_p(), C.c_size_t()
    assert unet.lib.ezdit_debug_buffer(unet._h, b'sync', C.byref(ptr), C.byref(nbytes)) == 0
    hip.hipMemset(C.c_void_p(ptr.value + 4 * i), C.c_int(0), C.c_size_t(4))


def one_run(steps):
    with torch.cuda.stream(smp.stream):
        smp.latents.copy_(init_dev, non_blocking=True)
        unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))
    torch.cuda.synchronize()
    t = time.perf_counter()
    smp.run(steps, use_graph=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


def measure(reps=3):
    # ONE step first: a configuration whose in-launch waits time out must not be repeated 200 times
    one_run(1)
    st = unet.lib.ezdit_device_status(unet._h, C.c_void_p(smp.stream.cuda_stream))
    if st != 0:
        return float('nan'), None, 'device status %d after one step' % st
    best = 1e9
    for _ in range(reps):
        best = min(best, one_run(n))
    st = unet.lib.ezdit_device_status(unet._h, C.c_void_p(smp.stream.cuda_stream))
    return best, smp.latents.clone(), ('ok' if st == 0 else 'device status %d' % st)


def setopt(name, v):
    rc = unet.lib.ezdit_set_option(unet._h, name.encode(), int(v))
    assert rc == 0, (name, v)


names = sorted({kv.split('=')[0] for c in cfgs for kv in c.split('+')})
ms0, ref, st0 = measure()
print(f'[{size} P={P}] baseline: {ms0:.3f} ms/step {st0} launches/step={unet.lib.ezdit_last_launch_count(unet._h)}', flush=True)
for c in cfgs:
    for k in names:
        setopt(k, DEFAULTS.get(k, 0))
    for kv in c.split('+'):
        k, v = kv.split('=')
        setopt(k, v)
    clear_sync_word(1001)
    try:
        ms, lat, st = measure()
        diff = float((lat - ref).abs().max()) if lat is not None else float('nan')
        fin = bool(torch.isfinite(lat).all()) if lat is not None else False
    except Exception as e:  # noqa: BLE001
        ms, diff, fin, st = float('nan'), float('nan'), False, repr(e)
    print(f'[{size} P={P}] {c}: {ms:.3f} ms/step ({ms0 / ms if ms == ms else 0:.3f}x) maxdiff={diff:.3e} finite={fin} {st} '
          f'agent_path={sync_word(1001)} launches/step={unet.lib.ezdit_last_launch_count(unet._h)}', flush=True)
for k in names:
    setopt(k, DEFAULTS.get(k, 0))
ms1, lat, st1 = measure()
print(f'[{size} P={P}] baseline again: {ms1:.3f} ms/step maxdiff={float((lat - ref).abs().max()):.3e} {st1}', flush=True)
