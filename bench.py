"""Headline benchmark: denoising steps/sec, EzAudio-XL, 10 s latent (500 frames), CFG on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size xl] [--prompts P]

One "step" = one pass of the hot path: CFG denoiser evaluation (cond + uncond rows) + CFG combine +
guidance rescale + DDIM update for every prompt of the batch.  Weights are random-init of
the named architecture, inputs are synthetic T5 embeddings and seeded noise (no network here).
N > 1: one rank per GPU over RCCL (torch.distributed backend "nccl").  Either launch it under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the environment) or just
call `python bench.py --gpus N`: without a launcher environment the script re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port.  Prompts are sharded (weak scaling: `--prompts` per GPU), there is
no collective inside the step loop, and finished latents are all-gathered once at the end of the timed region.
Single-prompt N > 1 is therefore N independent replicas + one gather (no CFG-split mode exists: DESIGN.md section 6.2).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # gfx950 dense bf16 MFMA peak (MI355X_MICROARCH.md)
GEGLU_VARIANT = 72000 + 60 * 4 + 2  # ezdit_test_gemm: tile config 60 (ping-pong kernel k_gemm_pp<128,288,4,2,3,EPI_GEGLU,1,64>: what the step launches for mlp.net.0.proj),
                                    # GEGLU epilogue that also finishes the LayerNorm of its operand (LayerNorm algebra, the default path; + 72000: that variant, on neutral statistics)


def load_yaml(path):
    from ezaudio_amd.config import load_yaml_with_includes
    return load_yaml_with_includes(path)


def model_section(size):
    from ezaudio_amd.config import configs, load_yaml_with_includes
    if size in ('xl', 'l'):
        return load_yaml_with_includes(configs['s3_' + size]['config'])
    raise SystemExit(f'unknown --size {size}')


def flops_per_step(cfg, B, L, Lc):
    """Algorithmic GEMM FLOPs of one denoiser evaluation with the step-invariant context work hoisted
    (SURVEY.md section 8d): 2 FLOP per MAC, no padding credit."""
    D, nblk, nskip, C = cfg['embed_dim'], cfg['depth'] + 1, cfg['depth'] // 2, cfg['out_chans']
    macs = nblk * (18 * L * D * D + 2 * L * L * D + 2 * L * Lc * D) + nskip * 2 * L * D * D \
        + L * D * (cfg['in_chans'] + C) + 3 * C * C * L
    return 2.0 * macs * B


def dominant_kernel_probe(unet, cfg, M, stream, iters=20):
    """Time the dominant kernel of the step (the GEGLU-in GEMM: 41 % of the step's FLOPs) with HIP events on
    the stream it is launched on, same kernel + shape the step uses (through the ABI's unit-test hook)."""
    D = cfg['embed_dim']
    inner = 4 * D
    dev = unet.device
    A = torch.randn(M, D, device=dev).to(torch.bfloat16)
    W = (torch.randn(2 * inner + 288, D, device=dev) / D ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(2 * inner, device=dev)
    out = torch.empty(M, inner, dtype=torch.bfloat16, device=dev)
    lib = unet.lib
    with torch.cuda.stream(stream):
        for _ in range(3):
            lib.ezdit_test_gemm(None, GEGLU_VARIANT, A.data_ptr(), D, W.data_ptr(), D, bias.data_ptr(), out.data_ptr(), inner, M,
                                2 * inner, D, 1, C.c_void_p(stream.cuda_stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            lib.ezdit_test_gemm(None, GEGLU_VARIANT, A.data_ptr(), D, W.data_ptr(), D, bias.data_ptr(), out.data_ptr(), inner, M,
                                2 * inner, D, 1, C.c_void_p(stream.cuda_stream))
        e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * M * D * 2 * inner
    return dict(name='k_gemm_pp<128,288,4,2,3,EPI_GEGLU,1,64> (mlp.net.0.proj + LayerNorm-finishing GEGLU epilogue)', launches_per_step=cfg['depth'] + 1,
                flops_per_launch=fl, avg_us=us, tflops=fl / us / 1e6, frac=fl / us / 1e6 / PEAK_BF16_TFLOPS,
                note='back-to-back launches, includes launch gaps')


def shard4_measure(unet, params, cfg, dev, n_ddim, use_graph, steps=100, warmup=50):
    """BASELINE.json config #4 (32 prompts over 8 GPUs) seen from ONE GPU: 4 prompts = 8 denoiser rows per launch, same sampler, same
    kernels, same clock discipline as the headline loop.  Reported as an extra key of the default line; never the headline `value`."""
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    P, L, Lc = 4, 10 * params['autoencoder']['latent_sr'], params['text_encoder']['max_length']
    g = torch.Generator().manual_seed(4711)
    text = torch.randn(P, Lc, cfg['context_dim'], generator=g)
    uncond = torch.randn(1, Lc, cfg['context_dim'], generator=g).repeat(P, 1, 1)
    text_mask = torch.zeros(P, Lc, dtype=torch.bool)
    for i in range(P):
        text_mask[i, :4 + (9 * i) % 37] = True
    uncond_mask = torch.zeros(P, Lc, dtype=torch.bool)
    uncond_mask[:, :1] = True
    init = torch.randn(P, cfg['out_chans'], L, generator=g)
    noise = torch.randn(n_ddim, P, cfg['out_chans'], L, generator=g)
    smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
    smp.prepare(text, text_mask, uncond, uncond_mask, init, noise, 5.0, 0.75, n_ddim, 1.0)
    init_dev = init.to(dev)

    def run_steps(k):
        done = 0
        while done < k:
            n = min(n_ddim, k - done)
            with torch.cuda.stream(smp.stream):
                smp.latents.copy_(init_dev, non_blocking=True)
                unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))
            smp.run(n, use_graph=use_graph)
            done += n

    run_steps(warmup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(smp.stream)
    run_steps(steps)
    e1.record(smp.stream)
    lat = smp.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(lat).all(), 'non-finite latents'
    fl = flops_per_step(cfg, 2 * P, L, Lc)
    ach = fl * (steps / (e0.elapsed_time(e1) * 1e-3)) / 1e12
    return {'workload': 'same sampler, 4 prompts/GPU = 8 denoiser rows (the per-GPU shard of BASELINE config #4: 32 prompts over 8 GPUs)',
            'prompts_per_gpu': P, 'steps': steps, 'warmup': warmup, 'ms_per_step': dt * 1e3 / steps, 'sample_steps_per_s': steps / dt * P,
            'flops_per_step': fl, 'achieved_tflops': ach, 'roofline_frac': ach / PEAK_BF16_TFLOPS,
            # executed: the single-key shortcut skips 2 L D^2 + 2 L Lc D MACs per block and unconditional row (see the headline line's flops_note)
            'roofline_frac_executed': ach * (1.0 - 2.0 * P * (cfg['depth'] + 1) * (2 * L * cfg['embed_dim'] ** 2 + 2 * L * Lc * cfg['embed_dim']) / fl) / PEAK_BF16_TFLOPS,
            'kernel_launches_per_step': unet.last_launch_count}


def relaunch_command(n_gpus, argv, port=None):
    """The torch.distributed.run command line `python bench.py --gpus N` turns itself into when no launcher started it."""
    if port is None:
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + list(argv)


def host_cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def usable_cpus():
    """Host cores this process may actually use: the scheduler affinity mask and the cgroup CPU quota both bound os.cpu_count()
    (a 256-core box with a 32-CPU quota must not get a 256-thread OpenMP pool: the spinning threads starve each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline_worker(size, threads, budget_s):
    """Runs in a child process (bench.py --cpu-baseline-worker): fp32 PyTorch eager forward of the same workload, `threads` threads."""
    from ezaudio_amd.weights import random_state_dict
    from oracle.torch_ref import DiTTorchRef
    torch.set_num_threads(threads)
    params = model_section(size)
    cfg = params['model']
    L = 10 * params['autoencoder']['latent_sr']
    Lc = params['text_encoder']['max_length']
    o = DiTTorchRef(cfg, random_state_dict(cfg, seed=1234))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, cfg['out_chans'], L, generator=g)
    ctx = torch.randn(2, Lc, cfg['context_dim'], generator=g)
    msk = torch.zeros(2, Lc, dtype=torch.bool)
    msk[0, :13] = True
    msk[1, :1] = True
    o.forward(x, 499, ctx, msk)                       # warm-up (thread pool, allocator)
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 25):
        t0 = time.perf_counter()
        o.forward(x, 499, ctx, msk)
        times.append(time.perf_counter() - t0)
    times.sort()
    print('CPU_BASELINE ' + json.dumps({'median_s': times[len(times) // 2], 'min_s': times[0], 'n': len(times), 'threads': torch.get_num_threads(),
                                        'L': L, 'Lc': Lc}))


def cpu_baseline(size, budget_s=20.0, timeout_s=240):
    """The reference's CPU path -- fp32, PyTorch eager, every usable host core -- timed on THIS box on a bounded
    sample of the same workload, in a child process with a hard timeout (a mis-sized thread pool must never hang the bench line).
    /root/reference does not travel to the GPU box, so the timed implementation is oracle/torch_ref.py, a restatement on the same
    aten operators; profiles/ref_cpu_baseline.json (tools/ref_cpu_baseline.py, build container) holds the unmodified reference
    timed next to it (time ratio 0.95) and is attached for provenance."""
    n = usable_cpus()
    tried = []
    for threads in dict.fromkeys([n, min(n, 64), min(n, 16)]):
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-baseline-worker', '--size', size, '--threads', str(threads),
               '--budget', str(budget_s)]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('CPU_BASELINE ')]
            if r.returncode == 0 and line:
                m = json.loads(line[-1][len('CPU_BASELINE '):])
                res = dict(value=1.0 / m['median_s'], unit='steps/s', cores=m['threads'], host_cores=os.cpu_count(), kind='port',
                           cpu_model=host_cpu_model(),
                           sample=f"{m['n']} CFG denoiser evaluations (B=2 rows, L={m['L']}, Lc={m['Lc']}), median; fp32 PyTorch eager on the aten "
                                  f"operators the reference modules call (oracle/torch_ref.py), {m['threads']} threads; CFG/DDIM update excluded (negligible)",
                           seconds_per_step=m['median_s'], min_seconds_per_step=m['min_s'])
                if tried:
                    res['note'] = 'thread counts that did not finish in %d s: %s' % (timeout_s, tried)
                ref_json = os.path.join(ROOT, 'profiles', 'ref_cpu_baseline.json')
                if os.path.exists(ref_json):
                    with open(ref_json) as f:
                        rj = json.load(f)
                    res['reference_on_build_box'] = {'steps_per_s': rj['reference']['steps_per_s'], 'cores': rj['host']['cores'],
                                                     'cpu_model': rj['host']['cpu_model'], 'kind': 'reference',
                                                     'port_time_ratio': rj['port']['time_ratio_vs_reference'],
                                                     'source': 'profiles/ref_cpu_baseline.json (tools/ref_cpu_baseline.py)'}
                return res
            tried.append((threads, 'rc=%d %s' % (r.returncode, r.stderr[-200:])))
        except subprocess.TimeoutExpired:
            tried.append((threads, 'timeout'))
    return {'error': 'cpu baseline did not finish', 'tried': tried, 'kind': 'port'}


def measured_traffic():
    """L2<->fabric bytes per step from profiles/*_pmc_step.json (scripts/pmc_step.sh), only if it was measured on THIS tree's kernels."""
    from ezaudio_amd.build import source_hash
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith('_pmc_step.json'):
            with open(os.path.join(pdir, f)) as fh:
                d = json.load(fh)
            if d.get('src_hash') == source_hash():
                best = (f, d)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed denoising steps (the 50-step sampler loop is repeated)')
    ap.add_argument('--warmup', type=int, default=50, help='untimed steps before (hipGraph capture, clocks)')
    ap.add_argument('--size', default='xl')
    ap.add_argument('--prompts', type=int, default=1, help='prompts per GPU (each is a cond+uncond pair)')
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-probe', action='store_true', help='skip the dominant-kernel probe (profiling passes)')
    ap.add_argument('--opt', action='append', default=[], help='name=value tuning knob passed to ezdit_set_option')
    ap.add_argument('--controlnet', action='store_true', help='BASELINE config #5: add an energy ControlNet of the same width')
    ap.add_argument('--dist-backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL over xGMI, the real thing) or 'gloo' "
                                                           '(test mode: latents are gathered through host memory)')
    ap.add_argument('--shared-device', action='store_true',
                    help='test mode for 1-GPU boxes: every rank uses cuda:0 (needs --dist-backend gloo; RCCL refuses duplicate devices)')
    ap.add_argument('--no-shard4', action='store_true', help="skip the extra 4-prompts/GPU measurement (BASELINE config #4's per-GPU shard) of the default line")
    ap.add_argument('--dump-latents', default='', help='rank 0 saves the gathered final latents [prompts x world, C, L] here (torch.save): placement tests')
    ap.add_argument('--as-rank', type=int, default=-1, help='single process: seed the synthetic inputs as rank R of a multi-rank job would (placement tests)')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--threads', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--budget', type=float, default=20.0, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.size, a.threads or usable_cpus(), a.budget)
        return

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        if 'RANK' not in os.environ and a.gpus > 1:
            # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
            env = dict(os.environ)
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            raise SystemExit(subprocess.call(relaunch_command(a.gpus, sys.argv[1:]), env=env))
        a.gpus = world
    if a.shared_device:
        if a.dist_backend != 'gloo':
            raise SystemExit('--shared-device needs --dist-backend gloo')
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from ezaudio_amd.dist import init_from_env
        init_from_env(a.dist_backend, dev if a.dist_backend == 'nccl' else None)

    from ezaudio_amd import MaskDiT
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    from ezaudio_amd.weights import random_state_dict

    params = model_section(a.size)
    cfg = params['model']
    L = 10 * params['autoencoder']['latent_sr']          # 10 s latent = 500 frames (api/ezaudio.py:105)
    Lc = params['text_encoder']['max_length']
    P = a.prompts
    sd = random_state_dict(cfg, seed=1234)
    unet = MaskDiT(device=dev, **cfg)
    unet.load_state_dict(sd)
    for kv in a.opt:
        k, v = kv.split('=')
        assert unet.lib.ezdit_set_option(unet._h, k.encode(), int(v)) == 0, kv

    # synthetic inputs, generated on CPU with seeded generators (identical on every box)
    irank = a.as_rank if (a.as_rank >= 0 and world == 1) else rank
    g = torch.Generator().manual_seed(11 + irank)
    text = torch.randn(P, Lc, cfg['context_dim'], generator=g)
    uncond = torch.randn(1, Lc, cfg['context_dim'], generator=g).repeat(P, 1, 1)
    text_mask = torch.zeros(P, Lc, dtype=torch.bool)
    for i in range(P):
        text_mask[i, :4 + (9 * (irank * P + i)) % 37] = True      # 4..40 valid tokens
    uncond_mask = torch.zeros(P, Lc, dtype=torch.bool)
    uncond_mask[:, :1] = True                                      # "" -> EOS only
    n_ddim = a.ddim_steps
    init = torch.randn(P, cfg['out_chans'], L, generator=g)
    noise = torch.randn(n_ddim, P, cfg['out_chans'], L, generator=g)

    smp = LatentSampler(unet, DDIMScheduler(**params['diff']))
    cn_kw, gs, gr = {}, 5.0, 0.75
    if a.controlnet:   # api/controlnet.py:113-118 defaults: guidance 3.5, rescale 0, conditioning_scale 1
        from ezaudio_amd import DiTControlNet
        from ezaudio_amd.config import controlnet_configs
        from ezaudio_amd.weights import random_controlnet_state_dict
        cn_cfg = load_yaml(controlnet_configs['energy']['config'])['controlnet']
        ccfg = dict(cfg); ccfg.update(cn_cfg)
        cn = DiTControlNet(device=dev, **ccfg)
        cn.load_state_dict(random_controlnet_state_dict(cfg, cn_cfg, seed=99))
        cond = torch.rand(P, 1, 2 * L, generator=g)      # a [0,1] control curve at 100 Hz (EnergyExtractor output range)
        cn_kw = dict(controlnet=cn, condition=cond, conditioning_scale=1.0)
        gs, gr = 3.5, 0.0
    # API defaults of generate_audio except the step count: guidance 5, rescale 0.75, eta 1
    smp.prepare(text, text_mask, uncond, uncond_mask, init, noise, gs, gr, n_ddim, 1.0, **cn_kw)
    init_dev = init.to(dev)
    use_graph = not a.no_graph

    def reset():
        with torch.cuda.stream(smp.stream):
            smp.latents.copy_(init_dev, non_blocking=True)
            unet.lib.ezdit_set_step(unet._h, 0, C.c_void_p(smp.stream.cuda_stream))

    def run_steps(k):
        done = 0
        while done < k:
            n = min(n_ddim, k - done)
            reset()
            smp.run(n, use_graph=use_graph)
            done += n

    run_steps(max(a.warmup, 1))          # includes hipGraph capture
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(smp.stream)
    run_steps(a.steps)
    e1.record(smp.stream)
    lat = smp.finish()
    if dist:  # the only collective of the job: gather the finished latents (256 KB per sample)
        from ezaudio_amd.dist import gather_samples
        all_lat = gather_samples(lat if a.dist_backend == 'nccl' else lat.cpu(), P * world)
        assert all_lat.shape[0] == P * world
    else:
        all_lat = lat
    if a.dump_latents and rank == 0:
        torch.save(all_lat.detach().float().cpu(), a.dump_latents)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if dist:
        tt = torch.tensor([dt], device=dev if a.dist_backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(lat).all(), 'non-finite latents'

    if rank == 0:
        B = 2 * P
        fl = flops_per_step(cfg, B, L, Lc)
        if a.controlnet:  # + depth/2 ControlNet blocks, its patch embed and depth/2 zero-Linears (2.29 TFLOP for XL)
            D_, nh = cfg['embed_dim'], cfg['depth'] // 2
            fl += 2.0 * B * (nh * (18 * L * D_ * D_ + 2 * L * L * D_ + 2 * L * Lc * D_) + L * D_ * cfg['in_chans'] + nh * L * D_ * D_)
        # FLOPs actually EXECUTED: the single-key shortcut (xkey1, default on) skips the cross-attention q projection, attention and out-projection of the P unconditional rows
        # (their context mask has one valid key): 2 L D^2 + 2 L Lc D MACs per block and row (report both, so that MFU comparisons across rounds stay meaningful)
        D_ = cfg['embed_dim']
        xkey1_on = not any(kv.replace(' ', '') in ('xkey1=0', 'zfuse=0') for kv in a.opt)
        nblk_x = (cfg['depth'] + 1) + (cfg['depth'] // 2 if a.controlnet else 0)
        fl_exec = fl - (2.0 * P * nblk_x * (2 * L * D_ * D_ + 2 * L * Lc * D_) if xkey1_on else 0.0)
        steps_per_s = a.steps / dt                       # loop iterations per second (per GPU)
        value = steps_per_s * P * world                  # sample-steps/s over the whole job
        ach = fl * (a.steps / (ev_ms * 1e-3)) / 1e12     # TFLOP/s from HIP events around the timed loop
        res = {
            'metric': 'denoising steps/sec (EzAudio-%s%s, 10 s latent, CFG on)' % (a.size.upper(), ' + ControlNet(energy)' if a.controlnet else ''),
            'value': value, 'unit': 'sample-steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt * 1e3 / a.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'EzAudio-%s (ezaudio-%s.yml) %d-step DDIM sampler, 10 s latent (L=%d, Lc=%d), CFG on '
                                   '(guidance %g, rescale %g, eta 1)%s, %d prompt(s)/GPU = %d denoiser rows/GPU, random-init '
                                   'weights, bf16 MFMA / fp32 accumulate + fp32 residual stream'
                                   % (a.size.upper(), a.size, n_ddim, L, Lc, gs, gr,
                                      ' + energy ControlNet (conditioning_scale 1)' if a.controlnet else '', P, B),
                       'prompts_per_gpu': P, 'rows_per_gpu': B, 'latent_frames': L, 'hipgraph': use_graph,
                       'kernel_launches_per_step': unet.last_launch_count},
            'loop_steps_per_s_per_gpu': steps_per_s,
            'distributed': {'world_size': (dist.get_world_size() if dist else 1), 'backend': (dist.get_backend() if dist else None),
                            'mode': 'prompt-sharded replicas, one all-gather of the finished latents; no collective in the step loop'},
            'roofline': {'bound': 'mfma', 'kernel': 'whole denoising step (all kernels of the DiT forward + CFG/DDIM)',
                         'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                         'flops_per_step': fl, 'event_ms_per_step': ev_ms / a.steps, 'traffic': None,
                         'executed_flops_per_step': fl_exec, 'achieved_executed': ach * fl_exec / fl, 'frac_executed': ach * fl_exec / fl / PEAK_BF16_TFLOPS,
                         'flops_note': 'algorithmic FLOPs of the FULL step: NOT reduced for the exact single-key shortcut (the unconditional rows have one valid context key, '
                                       'so their cross-attention + out-projection is the constant W_o v + b_o added by the attention-out projection; the cross-attention q projection, '
                                       'attention and out-projection of those rows -- 2 L D^2 + 2 L Lc D of the 18 L D^2 + ... MACs per block and unconditional row, 5.4 % of flops_per_step -- are not executed)'},
        }
        if a.size == 'xl' and P == 1 and not a.controlnet and L == 500:
            mt = measured_traffic()
            if mt is not None:   # separate rocprofv3 --pmc passes of this same command, on THIS tree's kernels (else: null)
                res['roofline']['traffic'] = mt[1]['traffic_bytes_per_step']
                res['roofline']['traffic_unit'] = 'L2<->fabric bytes per step (FETCH_SIZE x2 + WRITE_SIZE; largely served by the Infinity Cache, not all of it reaches HBM)'
                res['roofline']['traffic_source'] = f'profiles/{mt[0]} (src_hash {mt[1]["src_hash"]}; FETCH_SIZE x2 + WRITE_SIZE)'
                res['roofline']['traffic_over_weights'] = mt[1]['traffic_bytes_per_step'] / float(unet._blob.numel())
                res['roofline']['mfma_busy_frac_pmc'] = mt[1].get('mfma_busy_frac')
        if not a.no_probe:
            try:
                res['roofline']['dominant_kernel'] = dominant_kernel_probe(unet, cfg, B * L, smp.stream)
            except Exception as e:  # the probe must never cost the headline number
                res['roofline']['dominant_kernel'] = {'error': repr(e)}
        if world == 1 and P == 1 and a.size == 'xl' and not a.controlnet and not a.no_shard4 and a.as_rank < 0:
            try:   # after the headline loop: it re-binds the workspace for 8 rows
                res['config4_shard'] = shard4_measure(unet, params, cfg, dev, n_ddim, use_graph)
            except Exception as e:
                res['config4_shard'] = {'error': repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(a.size)
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
