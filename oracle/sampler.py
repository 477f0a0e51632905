"""numpy restatement of the reference sampling loop.  TEST INFRASTRUCTURE.

Follows /root/reference/src/inference.py:26-107 (`inference`) and :12-23 (`rescale_noise_cfg`)
with the text encoder, tokenizer and VAE factored out: the caller passes the T5 embeddings and
masks (BASELINE uses random T5 embeddings) and receives the final latent (what the reference hands
to ``autoencoder(embedding=pred)``, src/inference.py:106).  Noise is injected, not drawn, so that
both sides of a parity test see identical tensors.
"""
import numpy as np

from .ddim import DDIMOracle


def cfg_combine(pred_text, pred_uncond, guidance_scale):
    """src/inference.py:88-90."""
    return pred_uncond + np.float32(guidance_scale) * (pred_text - pred_uncond)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale):
    """src/inference.py:12-23.  torch.std default = unbiased (N-1), over all non-batch dims."""
    axes = tuple(range(1, noise_cfg.ndim))
    std_text = noise_pred_text.std(axis=axes, keepdims=True, ddof=1, dtype=np.float32)
    std_cfg = noise_cfg.std(axis=axes, keepdims=True, ddof=1, dtype=np.float32)
    rescaled = noise_cfg * (std_text / std_cfg)
    phi = np.float32(guidance_rescale)
    return phi * rescaled + (np.float32(1) - phi) * noise_cfg


def sample(denoise, text, text_mask, uncond_text, uncond_mask, init_noise, step_noises,
           guidance_scale=5.0, guidance_rescale=0.75, ddim_steps=50, eta=1.0,
           gt=None, gt_mask=None, diff_params=None, trace=None):
    """One prompt, batch 1, exactly the reference loop.

    denoise(x[B,C,L], t:int, ctx[B,Lc,Cc], ctx_mask[B,Lc], gt, gt_mask) -> pred[B,C,L]
    init_noise [1,C,L]; step_noises: sequence of [1,C,L], one per step in loop order
    (the reference draws one per step whenever eta > 0, including the last).
    """
    sched = DDIMOracle(**(diff_params or {}))
    sched.set_timesteps(ddim_steps)
    latents = np.asarray(init_noise, dtype=np.float32)
    for i, t in enumerate(sched.timesteps):
        if guidance_scale:
            lat2 = np.concatenate([latents, latents], axis=0)            # inference.py:75
            ctx2 = np.concatenate([text, uncond_text], axis=0)           # :76
            msk2 = np.concatenate([text_mask, uncond_mask], axis=0)      # :77
            gt2 = None if gt is None else np.concatenate([gt, gt], axis=0)
            gm2 = None if gt is None else np.concatenate([gt_mask, gt_mask], axis=0)
            out = denoise(lat2, int(t), ctx2, msk2, gt2, gm2)
            o_text, o_unc = out[:1], out[1:]                             # :88
            pred = cfg_combine(o_text, o_unc, guidance_scale)
            if guidance_rescale > 0.0:
                pred = rescale_noise_cfg(pred, o_text, guidance_rescale)
        else:
            pred = denoise(latents, int(t), text, text_mask, gt, gt_mask)
        latents = sched.step(pred, t, latents, eta, None if eta <= 0 else step_noises[i])
        if trace is not None:
            trace.append(latents.copy())
    pred = latents / np.float32(1.0) - np.float32(0.0)  # scale_shift_re with scale 1, shift 0 (:102-103)
    if gt is not None:
        pred = np.where(np.asarray(gt_mask, dtype=bool), pred, np.asarray(gt, dtype=np.float32))  # :104-105
    return pred
