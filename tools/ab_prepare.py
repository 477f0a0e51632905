"""In-situ A/B of option sets that must be in force when the sampler is PREPARED (the LayerNorm-algebra tables are built by
ezdit_prepare_timesteps): every combo re-runs prepare, then times the 50-step graph loop.

    python tools/ab_prepare.py [size] [prompts] [--reps N] [--cn] combo combo ...      combo = "name=v+name=v" or "base" (defaults)

Prints ms/step (best of reps) per combo, twice around (A B C A B C) so that drift of the box shows up, and the latent checksum.
"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import MaskDiT, DDIMScheduler                       # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.sampler import LatentSampler                         # noqa: E402
from ezaudio_amd.weights import random_state_dict                     # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 3
if '--reps' in sys.argv:
    args.remove(sys.argv[sys.argv.index('--reps') + 1])
rounds = 1 if '--once' in sys.argv else 2
size = args[0] if args else 'xl'
P = int(args[1]) if len(args) > 1 else 1
combos = args[2:] or ['base']
params = load_yaml_with_includes(configs['s3_' + size]['config'])
cfg = params['model']
unet = MaskDiT(device='cuda', **cfg)
unet.load_state_dict(random_state_dict(cfg, seed=0))
L, Lc, n = 500, 100, 50
g = torch.Generator().manual_seed(1)
text = torch.randn(P, Lc, cfg['context_dim'], generator=g)
mask = torch.zeros(P, Lc, dtype=torch.bool); mask[:, :12] = True
um = torch.zeros(P, Lc, dtype=torch.bool); um[:, :1] = True
utext = torch.randn(P, Lc, cfg['context_dim'], generator=g)
init = torch.randn(P, 128, L, generator=g)
noise = torch.randn(n, P, 128, L, generator=g)
init_dev = init.cuda()
touched = {}


def setopt(name, v):
    rc = unet.lib.ezdit_set_option(unet._h, name.encode(), int(v))
    assert rc == 0, (name, v)


def measure(combo):
    for k, v in touched.items():      # back to the defaults recorded at first touch
        setopt(k, v)
    if combo != 'base':
        for kv in combo.split('+'):
            k, v = kv.split('=')
            touched.setdefault(k, DEFAULTS[k])   # an option missing here would be 'restored' to a wrong value
            setopt(k, v)
    smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
    smp.prepare(text, mask, utext, um, init, noise, 5.0, 0.75, n, 1.0)
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
    lat = smp.latents
    return best, bool(torch.isfinite(lat).all()), float(lat.abs().mean()), unet.last_launch_count


# defaults of the options this script may touch (csrc/api.hip)
DEFAULTS = dict(zfuse=1, gemm_pp=3, tile_partial=9, wt=2, fuse_q2=1, attn_xk2=1, gemm_panel=3, row_affine=1, attn_nkh=0, q2_pp=1, attn_xcd=1, row_variant=1,
                epi_lds=1, cn_overlap=1, zfake=0, xkey1=1, geglu_co=0, qkv_co=1, attn_qtile=0, skip_z=1)
for r in range(rounds):
    for combo in combos:
        try:
            ms, ok, chk, nl = measure(combo)
            print(f'[{r}] {combo:40s} {ms:7.3f} ms/step  finite={ok} |latents|={chk:.6f} launches/step~{nl}', flush=True)
        except Exception as e:  # noqa: BLE001
            print(f'[{r}] {combo:40s} FAILED {e!r}', flush=True)
