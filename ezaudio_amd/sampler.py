"""Sampler driver: the reference's ``inference()`` (/root/reference/src/inference.py:26-107) with the
whole per-step loop body -- CFG batch, denoiser, CFG combine, guidance rescale, DDIM update -- resident on
the GPU: one hipGraph replayed ``ddim_steps`` times, no host synchronisation inside the loop
(the reference syncs every step through the CPU-side scheduler).

Beyond the reference: ``P`` independent prompts per call (the reference hard-codes one noise row,
src/inference.py:67); each prompt's cond/uncond pair stays on one GPU.
"""
import ctypes as C

import torch

from . import _lib
from .denoiser import _ptr


def scale_shift_re(x, scale, shift):
    """src/utils/utils.py:24-25."""
    return (x / scale) - shift


class LatentSampler:
    """prepare() once per call, then run(n) advances n DDIM steps on the device."""

    def __init__(self, unet, scheduler):
        self.unet = unet
        self.scheduler = scheduler
        self.stream = torch.cuda.Stream(device=unet.device)
        self.latents = None

    def prepare(self, text, text_mask, uncond_text, uncond_mask, init_noise, step_noises, guidance_scale,
                guidance_rescale, ddim_steps, eta, gt=None, gt_mask=None, controlnet=None, condition=None,
                conditioning_scale=1.0):
        u = self.unet
        dev = u.device
        P, Cc, L = init_noise.shape
        self.scheduler.set_timesteps(ddim_steps)
        ts = [int(t) for t in self.scheduler.timesteps]
        coefs = self.scheduler.ddim_coefficients(eta)
        use_cfg = bool(guidance_scale)
        if use_cfg:
            ctx = torch.cat([text, uncond_text], dim=0)          # src/inference.py:76-77
            msk = torch.cat([text_mask, uncond_mask], dim=0)
        else:
            ctx, msk = text, text_mask
        B = ctx.shape[0]
        self.latents = init_noise.to(dev, torch.float32).contiguous().clone()
        self.noise = None if (eta <= 0 or step_noises is None) else step_noises.to(dev, torch.float32).contiguous()
        if eta > 0 and self.noise is None:
            raise ValueError('eta > 0 needs step_noises [n_steps, P, C, L]')
        if (gt is None) != (gt_mask is None):
            raise ValueError('gt and gt_mask must be given together')
        # the kernels index gt / gt_mask per latent row ([P, C, L]): a shared [1, C, L] reference (one clip edited under P prompts,
        # which is also what the sharded driver forwards un-sliced) is broadcast here
        for name, t in (('gt', gt), ('gt_mask', gt_mask)):
            if t is not None and t.shape[0] not in (1, P):
                raise ValueError(f'{name} has {t.shape[0]} rows; expected 1 or P={P}')
        self.gt = None if gt is None else gt.to(dev, torch.float32).expand(P, Cc, L).contiguous()
        self.gt_mask = None if gt_mask is None else gt_mask.to(dev).expand(P, Cc, L).to(torch.uint8).contiguous()
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            u.bind(B, L, ctx.shape[1], ddim_steps)
            u.prepare_context(ctx, msk)
            u.prepare_timesteps(ts, per_row=False)
            if controlnet is not None:  # src/inference_controlnet.py:78-99: condition duplicated for the CFG pair
                cond = torch.cat([condition, condition], dim=0) if use_cfg else condition
                controlnet.bind(B, L, ctx.shape[1], ddim_steps)
                controlnet.prepare_context(ctx, msk)
                controlnet.prepare_timesteps(ts, per_row=False)
                controlnet.prepare_condition(cond)
            _lib.check(u.lib.ezdit_sampler_attach_controlnet(u._h, controlnet._h if controlnet is not None else None,
                                                             float(conditioning_scale)))
            self.controlnet = controlnet
            arr = (_lib.EzditDdimCoef * ddim_steps)(*[_lib.EzditDdimCoef(*c) for c in coefs])
            _lib.check(u.lib.ezdit_sampler_begin(u._h, _ptr(self.latents), P, _ptr(self.noise), arr, ddim_steps,
                                                 float(guidance_scale or 0.0), float(guidance_rescale or 0.0),
                                                 _ptr(self.gt), _ptr(self.gt_mask),
                                                 C.c_void_p(self.stream.cuda_stream)))
        self.n_steps = ddim_steps

    def run(self, n=None, use_graph=True):
        n = self.n_steps if n is None else n
        with torch.cuda.stream(self.stream):
            _lib.check(self.unet.lib.ezdit_sampler_run(self.unet._h, n, 1 if use_graph else 0,
                                                       C.c_void_p(self.stream.cuda_stream)))
        return self.latents

    def finish(self, block=True):
        """Hand back the latents.  BLOCKS THE HOST by default (`stream.synchronize()` on the sampler stream), so the caller's stream needs no
        ordering afterwards and the returned tensor is final.

        The host wait is deliberate: queueing a cross-stream wait (`current_stream.wait_stream(self.stream)`) while the step graphs are
        still executing makes the graphs themselves run 5-6 % slower on MI355X / ROCm 7.2 (4.01 vs 4.24 ms per step over a 20-step
        loop, HIP events on the sampler stream, alternating in one process: profiles/r04_experiments.txt) -- a second hardware queue
        parked on a barrier packet next to the running one.

        block=False keeps the round-3 contract for callers that pipeline host work: no host synchronisation, the CURRENT stream is made to wait
        for the sampler stream (stream-ordered; costs the running graphs the 5-6 % above)."""
        if block:
            self.stream.synchronize()
        else:
            torch.cuda.current_stream(self.latents.device).wait_stream(self.stream)
        return self.latents


def draw_noises(codec_dim, audio_frames, ddim_steps, eta, random_seed, device, n_prompts=1, first_index=0):
    """Init noise + per-step DDIM noise in the order the reference draws them from ONE generator
    (src/inference.py:58-67 then one randn per scheduler.step, diffusers `randn_tensor`).  For several
    prompts each sample gets its own generator seeded seed + index, so results do not depend on how
    prompts are sharded over GPUs."""
    inits, steps = [], []
    for i in range(n_prompts):
        g = torch.Generator(device=device)
        if random_seed is not None:
            g.manual_seed(random_seed + first_index + i)
        else:
            g.seed()
        inits.append(torch.randn((1, codec_dim, audio_frames), generator=g, device=device))
        if eta > 0:
            steps.append(torch.stack([torch.randn((1, codec_dim, audio_frames), generator=g, device=device)
                                      for _ in range(ddim_steps)], dim=0))
    init = torch.cat(inits, dim=0)
    step = torch.cat(steps, dim=1) if steps else None
    return init, step


@torch.no_grad()
def inference_controlnet(autoencoder, unet, controlnet, gt, gt_mask, condition, tokenizer, text_encoder, params,
                         noise_scheduler, text_raw, neg_text=None, audio_frames=500, guidance_scale=3,
                         guidance_rescale=0.0, ddim_steps=50, eta=1, random_seed=2024, conditioning_scale=1.0,
                         device='cuda', use_graph=True):
    """Same signature and semantics as the reference's ControlNet ``inference`` (src/inference_controlnet.py:27-129)."""
    return inference(autoencoder, unet, gt, gt_mask, tokenizer, text_encoder, params, noise_scheduler, text_raw, neg_text,
                     audio_frames, guidance_scale, guidance_rescale, ddim_steps, eta, random_seed, device, use_graph,
                     controlnet=controlnet, condition=condition, conditioning_scale=conditioning_scale)


@torch.no_grad()
def inference(autoencoder, unet, gt, gt_mask, tokenizer, text_encoder, params, noise_scheduler, text_raw,
              neg_text=None, audio_frames=500, guidance_scale=3, guidance_rescale=0.0, ddim_steps=50, eta=1,
              random_seed=2024, device='cuda', use_graph=True, controlnet=None, condition=None, conditioning_scale=1.0,
              first_index=None):
    """Same signature and semantics as the reference's ``inference`` (src/inference.py:26-107).

    Extension: with ``torch.distributed`` initialised and several prompts, every rank samples AND
    VAE-decodes its own contiguous shard of the prompts and the waveforms are all-gathered once (RCCL)."""
    if neg_text is None:
        neg_text = [""]
    if isinstance(text_raw, str):
        text_raw = [text_raw]
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and len(text_raw) > 1 and first_index is None:
        from .dist import sample_sharded
        n_all = len(text_raw)
        neg_all = list(neg_text) * n_all if len(neg_text) == 1 else list(neg_text)

        def local(s, e):
            if e == s:   # more ranks than prompts: contribute an empty shard of the common shape
                ratio = params['autoencoder']['sr'] // params['autoencoder']['latent_sr']
                return torch.zeros(0, 1, audio_frames * ratio, device=device)
            sl = lambda t: t if t is None or t.shape[0] == 1 else t[s:e]   # noqa: E731  per-prompt tensors are sliced
            return inference(autoencoder, unet, sl(gt), sl(gt_mask), tokenizer, text_encoder, params, noise_scheduler,
                             list(text_raw[s:e]), neg_all[s:e], audio_frames, guidance_scale, guidance_rescale, ddim_steps, eta,
                             random_seed, device, use_graph, controlnet, sl(condition), conditioning_scale, first_index=s)
        return sample_sharded(local, n_all)
    first_index = first_index or 0
    n_prompts = len(text_raw)
    if tokenizer is not None:
        max_len = params['text_encoder']['max_length']
        tb = tokenizer(text_raw, max_length=max_len, padding="max_length", truncation=True, return_tensors="pt")
        text, text_mask = tb.input_ids.to(device), tb.attention_mask.to(device).bool()
        text = text_encoder(input_ids=text, attention_mask=text_mask).last_hidden_state
        ub = tokenizer(neg_text * n_prompts if len(neg_text) == 1 else neg_text, max_length=max_len,
                       padding="max_length", truncation=True, return_tensors="pt")
        uncond_text, uncond_mask = ub.input_ids.to(device), ub.attention_mask.to(device).bool()
        uncond_text = text_encoder(input_ids=uncond_text, attention_mask=uncond_mask).last_hidden_state
    else:
        raise NotImplementedError('tokenizer=None (unconditional model) is not a shipped configuration')
    codec_dim = params['model']['out_chans']
    unet.eval()
    init, step_noises = draw_noises(codec_dim, audio_frames, ddim_steps, eta, random_seed, device, n_prompts, first_index)
    smp = LatentSampler(unet, noise_scheduler)
    smp.prepare(text.float(), text_mask, uncond_text.float(), uncond_mask, init, step_noises, guidance_scale,
                guidance_rescale, ddim_steps, eta, gt=gt, gt_mask=gt_mask, controlnet=controlnet, condition=condition,
                conditioning_scale=conditioning_scale)
    smp.run(use_graph=use_graph)
    latents = smp.finish()
    pred = scale_shift_re(latents, params['autoencoder']['scale'], params['autoencoder']['shift'])
    if gt is not None:   # src/inference.py:103-104, with a shared [1, C, L] reference broadcast over the prompts
        keep = ~gt_mask.to(pred.device).expand_as(pred)
        pred = torch.where(keep, gt.to(pred.device, pred.dtype).expand_as(pred), pred)
    return autoencoder(embedding=pred)
