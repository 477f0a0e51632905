"""Time the REFERENCE's own CPU path next to the torch port that bench.py can carry to the GPU box.

    python tools/ref_cpu_baseline.py [--size xl] [--runs 5] [--out profiles/ref_cpu_baseline.json]

Runs only in the build container (imports /root/reference, read-only).  For the headline workload (EzAudio-XL, CFG pair
B = 2, L = 500, Lc = 100, fp32, PyTorch eager, torch.set_num_threads(os.cpu_count())) it reports, as median over `runs`
forwards after one warm-up:
  * reference   -- src/models/conditioners.py:MaskDiT, unmodified, loaded with the synthetic checkpoint
  * port        -- oracle/torch_ref.py (same aten operators; what bench.py's cpu_baseline times on the GPU box)
  * numpy       -- oracle/dit.py (the numerics checker; round 1's baseline)
and the max difference between reference and port outputs.  The JSON is committed so the bench line can cite it.
"""
import argparse
import json
import os
import platform
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def med_time(fn, runs):
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', default='xl')
    ap.add_argument('--runs', type=int, default=5)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'ref_cpu_baseline.json'))
    a = ap.parse_args()
    from oracle.dit import DiTOracle
    from oracle.mint_golden import build_reference
    from oracle.torch_ref import DiTTorchRef
    from oracle.weights import make_inputs, model_config
    torch.set_num_threads(os.cpu_count())
    cfg = model_config(a.size)
    ref, sd = build_reference(cfg, 1234)
    inp = make_inputs(cfg, B=2, L=500, Lc=100, seed=11)
    x, ctx, msk = torch.from_numpy(inp['x'].copy()), torch.from_numpy(inp['ctx']), torch.from_numpy(inp['ctx_mask'])
    t = torch.tensor(499)
    out = {}
    with torch.no_grad():
        out['ref'] = ref(x, t, ctx, context_mask=msk, cls_token=None)[0]
        r_med, r_min = med_time(lambda: ref(x, t, ctx, context_mask=msk, cls_token=None), a.runs)
    port = DiTTorchRef(cfg, sd)
    out['port'] = port.forward(x, 499, ctx, msk)[0]
    p_med, p_min = med_time(lambda: port.forward(x, 499, ctx, msk), a.runs)
    o = DiTOracle(cfg, sd, np.float32)
    n_med, n_min = med_time(lambda: o.forward(inp['x'], 499, inp['ctx'], inp['ctx_mask']), max(2, a.runs // 2))
    diff = float((out['ref'] - out['port']).abs().max())
    commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    res = {
        'workload': f'EzAudio-{a.size.upper()} denoiser forward, CFG pair (B=2 rows), L=500, Lc=100, fp32, PyTorch {torch.__version__} CPU eager',
        'host': {'cpu_model': cpu_model(), 'cores': os.cpu_count(), 'torch_threads': torch.get_num_threads()},
        'runs': a.runs,
        'reference': {'impl': '/root/reference src/models/conditioners.py:MaskDiT (unmodified)', 'median_s': r_med, 'min_s': r_min,
                      'steps_per_s': 1.0 / r_med},
        'port': {'impl': 'oracle/torch_ref.py (same aten ops)', 'median_s': p_med, 'min_s': p_min, 'steps_per_s': 1.0 / p_med,
                 'max_abs_diff_vs_reference': diff, 'time_ratio_vs_reference': p_med / r_med},
        'numpy_oracle': {'impl': 'oracle/dit.py', 'median_s': n_med, 'min_s': n_min, 'steps_per_s': 1.0 / n_med},
        'commit': commit,
    }
    with open(a.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
