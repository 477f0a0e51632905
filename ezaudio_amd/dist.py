"""Multi-GPU batched generation: one process per GPU, prompts sharded, ONE all-gather at the end.

The reference has no inference-time parallelism (single process, batch 1: src/inference.py:67).  Samples never interact --
the only cross-row operation of the whole path is the CFG pair (src/inference.py:88-90) and its per-sample std (:17-18) --
so the natural partition is by prompt with each cond/uncond pair kept on one GPU: weights are
replicated, there is NO collective inside the denoising loop, and finished latents (P x 128 x L fp32, 256 KB per sample)
are gathered once with `all_gather` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).
Per-sample RNG streams (seed + global sample index, sampler.draw_noises) make results independent of the placement.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = torch.device(device)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_items, rank, world):
    """Contiguous balanced split: the first (n_items % world) ranks get one extra item."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_samples(local, n_total, group=None):
    """local: [n_local, ...] on this rank (contiguous shard per shard_range) -> [n_total, ...] on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = -(-n_total // world)  # all_gather needs equal shapes: pad every shard to the largest
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        parts.append(out[r][:e - s])
    return torch.cat(parts, dim=0)


def sample_sharded(sample_fn, n_prompts, group=None):
    """Run `sample_fn(start, end) -> tensor [end-start, ...]` on this rank's shard of the prompts and gather.

    sample_fn receives GLOBAL prompt indices so that it can derive per-sample seeds (seed + index)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    s, e = shard_range(n_prompts, rank, world)
    local = sample_fn(s, e)
    if local.shape[0] != e - s:
        raise AssertionError(f'sample_fn returned {local.shape[0]} samples for shard [{s},{e})')
    return gather_samples(local, n_prompts, group)
