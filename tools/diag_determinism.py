"""Is a forward bit-reproducible, and if not: which launch is the first whose outputs differ between two runs?

    python tools/diag_determinism.py [size] [batch] [--runs N] [name=v ...]

Runs the forward N times on the same inputs and counts the distinct results; on a mismatch it bisects over ezdit_debug_stop_after(n): the forward is cut after n
launches, the workspace buffers named below are hashed, and the first n at which two repetitions disagree is printed with the launch's name (option trace_launches).
"""
import ctypes as C
import hashlib
import sys

import torch

sys.path.insert(0, '.')
from ezaudio_amd import MaskDiT                                       # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.weights import random_state_dict                     # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--') and '=' not in a]
opts = [a for a in sys.argv[1:] if '=' in a]
runs = int(sys.argv[sys.argv.index('--runs') + 1]) if '--runs' in sys.argv else 12
if '--runs' in sys.argv:
    args.remove(sys.argv[sys.argv.index('--runs') + 1])
size = args[0] if args else 'xl'
B = int(args[1]) if len(args) > 1 else 2
cfg = load_yaml_with_includes(configs['s3_' + size]['config'])['model']
m = MaskDiT(device='cuda', **cfg)
m.load_state_dict(random_state_dict(cfg, seed=0))
lib, h = m.lib, m._h
for kv in opts:
    k, v = kv.split('=')
    assert lib.ezdit_set_option(h, k.encode(), int(v)) == 0, kv
L, Lc = 500, 100
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 128, L, generator=g).cuda()
ctx = torch.randn(B, Lc, cfg['context_dim'], generator=g).cuda()
mask = torch.zeros(B, Lc, dtype=torch.bool)
mask[:B // 2 or 1, :12] = True
mask[B // 2:, :1] = True      # CFG layout: cond rows, then single-key uncond rows
mask = mask.cuda()
t = torch.tensor(499)


def fwd():
    out = m(x, t, ctx, context_mask=mask)[0]
    torch.cuda.synchronize()
    return out


def digest(tensor):
    return hashlib.sha1(tensor.detach().cpu().numpy().tobytes()).hexdigest()[:12]


seen = {}
for i in range(runs):
    seen.setdefault(digest(fwd()), []).append(i)
print(f'{size} B={B} {" ".join(opts) or "defaults"}: {len(seen)} distinct result(s) in {runs} runs: {seen}; launches {m.last_launch_count}', flush=True)
if len(seen) == 1:
    sys.exit(0)

HIP = C.CDLL('libamdhip64.so')
NAMES = ["h", "u", "ucat", "ucat_z", 'skips', 'zstat', 'zstat_skip', 'ao', 'act', 'q', 'k', 'v']


def state_after(n):
    lib.ezdit_debug_stop_after(h, n)
    m(x, t, ctx, context_mask=mask)
    torch.cuda.synchronize()
    d = {}
    for name in NAMES:
        p, nb = C.c_void_p(), C.c_size_t()
        if lib.ezdit_debug_buffer(h, name.encode(), C.byref(p), C.byref(nb)) != 0:
            continue
        tns = torch.empty(nb.value, dtype=torch.uint8, device='cuda')
        assert HIP.hipMemcpy(C.c_void_p(tns.data_ptr()), p, nb, 3) == 0   # device to device
        torch.cuda.synchronize()
        d[name] = digest(tns)
    return d


def differs(n, reps=6):
    base = state_after(n)
    for _ in range(reps):
        cur = state_after(n)
        bad = [k for k in base if base[k] != cur[k]]
        if bad:
            return bad
    return []


total = m.last_launch_count
lo, hi = 0, total      # invariant: differs(lo) empty (assumed), differs(hi) non-empty
while hi - lo > 1:
    mid = (lo + hi) // 2
    bad = differs(mid)
    print(f'  after {mid} launches: {"DIFFERENT " + str(bad) if bad else "same"}', flush=True)
    if bad:
        hi = mid
    else:
        lo = mid
print(f'first launch whose outputs differ between runs: index {hi - 1} (0-based; run with trace_launches=1 for its name)')
lib.ezdit_debug_stop_after(h, 0)
assert lib.ezdit_set_option(h, b'trace_launches', 1) == 0
lib.ezdit_debug_stop_after(h, hi)
m(x, t, ctx, context_mask=mask)
torch.cuda.synchronize()
