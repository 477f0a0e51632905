"""DDIM scheduler with the call surface the reference uses from `diffusers.DDIMScheduler`
(api/ezaudio.py:11,92-97; src/inference.py:64,70-71,98-100): constructor kwargs = the `diff:` section
of the yml, `set_timesteps`, `timesteps`, `scale_model_input`, `step(...).prev_sample`, `add_noise`,
`config.num_train_timesteps`.

diffusers is not vendored by the reference and not installable here, so this is a from-scratch host-side
implementation of the published algorithm (scaled_linear betas, zero-terminal-SNR rescale, trailing
spacing, v-prediction step).  Scalar tables are built with torch float32 CPU ops exactly as diffusers
builds them, and `ddim_coefficients` exports the per-step scalars the fused HIP kernel consumes.
"""
import types

import numpy as np
import torch


def _linspace_f32(start, end, steps):
    """torch.linspace(..., dtype=float32) by its scalar definition (step = (end-start)/(steps-1); first half counted
    up from start, second half down from end).  torch's own CPU kernel evaluates this with SIMD-width-dependent
    rounding, i.e. the last bit of beta depends on the host's vector ISA; the zero-terminal-SNR schedule then lands
    either side of 0 in `1 - alpha_prev - sigma^2` at t = 999 (see `ddim_coefficients`).  A fixed scalar evaluation
    makes the schedule identical on every machine."""
    start, end = np.float32(start), np.float32(end)
    step = np.float32((end - start) / np.float32(steps - 1))
    idx = np.arange(steps)
    up = (start + step * idx.astype(np.float32)).astype(np.float32)
    down = (end - step * (steps - 1 - idx).astype(np.float32)).astype(np.float32)
    return torch.from_numpy(np.where(idx < steps // 2, up, down).astype(np.float32))


def _rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    abar_sqrt = torch.cumprod(alphas, dim=0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = abar_sqrt - aT
    abar_sqrt = abar_sqrt * (a0 / (a0 - aT))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 prediction_type='epsilon', rescale_betas_zero_snr=False, timestep_spacing='leading',
                 clip_sample=True, set_alpha_to_one=True, **unused):
        if beta_schedule != 'scaled_linear':
            raise NotImplementedError(f'beta_schedule={beta_schedule!r}')
        if prediction_type != 'v_prediction':
            raise NotImplementedError(f'prediction_type={prediction_type!r}')
        if timestep_spacing != 'trailing':
            raise NotImplementedError(f'timestep_spacing={timestep_spacing!r}')
        if clip_sample:
            raise NotImplementedError('clip_sample=True')
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                            beta_end=beta_end, beta_schedule=beta_schedule,
                                            prediction_type=prediction_type,
                                            rescale_betas_zero_snr=rescale_betas_zero_snr,
                                            timestep_spacing=timestep_spacing, clip_sample=clip_sample)
        betas = _linspace_f32(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
        if rescale_betas_zero_snr:
            betas = _rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps / num_inference_steps
        ts = np.round(np.arange(self.config.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _scalars(self, timestep):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def _coef(self, t, eta):
        a_t, a_prev = self._scalars(t)
        b_t = 1 - a_t
        variance = ((1 - a_prev) / b_t) * (1 - a_t / a_prev)
        sigma = eta * variance ** 0.5
        # With zero terminal SNR and eta = 1 the radicand is 0 in exact arithmetic at t = 999 (alpha_bar_t = 0 =>
        # sigma^2 = 1 - alpha_bar_prev) and +-6e-8 in fp32: the reference (diffusers) takes sqrt of whichever sign
        # the host's rounding produced (NaN for e.g. 25 steps).  Clamp at 0: identical whenever the reference is
        # finite, finite where it is not.
        c_dir = torch.clamp(1 - a_prev - sigma ** 2, min=0.0) ** 0.5
        return tuple(float(v) for v in (a_t ** 0.5, b_t ** 0.5, a_prev ** 0.5, c_dir, sigma))

    def ddim_coefficients(self, eta):
        """[(sa, sb, c_x0, c_dir, sigma)] per step, float32, for ezdit_sampler_begin / ezdit_cfg_ddim_step."""
        return [self._coef(t, eta) for t in self.timesteps]

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None, **unused):
        if sample.is_cuda:   # the reference's own loop driving this scheduler on the GPU: one HIP launch, no torch math
            return self._step_hip(model_output, timestep, sample, eta, generator, variance_noise)
        a_t, a_prev = self._scalars(timestep)
        b_t = 1 - a_t
        x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        variance = ((1 - a_prev) / b_t) * (1 - a_t / a_prev)
        sigma = eta * variance ** 0.5
        prev = a_prev ** 0.5 * x0 + torch.clamp(1 - a_prev - sigma ** 2, min=0.0) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + sigma * variance_noise
        return types.SimpleNamespace(prev_sample=prev)

    def _step_hip(self, model_output, timestep, sample, eta, generator, variance_noise):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        coef = _lib.EzditDdimCoef(*self._coef(timestep, eta))
        pred = model_output.contiguous().float()
        prev = sample.contiguous().float().clone()
        noise = None
        if eta > 0:
            noise = variance_noise if variance_noise is not None else torch.randn(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            noise = noise.contiguous().float()
        P = prev.shape[0]
        st = torch.cuda.current_stream(prev.device).cuda_stream
        _lib.check(lib.ezdit_cfg_ddim_step(pred.data_ptr(), prev.data_ptr(), noise.data_ptr() if noise is not None else None,
                                           C.byref(coef), 0.0, 0.0, P, prev[0].numel(), None, C.c_void_p(st)))
        return types.SimpleNamespace(prev_sample=prev.to(sample.dtype))

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = ac[timesteps] ** 0.5
        sb = (1 - ac[timesteps]) ** 0.5
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise
