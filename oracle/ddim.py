"""numpy restatement of the DDIM scheduler the reference drives.  TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference instantiates ``diffusers.DDIMScheduler(**params['diff'])``
(/root/reference/api/ezaudio.py:11,92; parameters ckpts/ezaudio-xl.yml:52-60; calls at
src/inference.py:64,70-71,98-100).  diffusers is a third-party dependency that is not vendored,
not version-pinned (requirements.txt:2) and not installable here, and the reference has no tests
for it.  What follows restates the published algorithm of
``diffusers/schedulers/scheduling_ddim.py`` (``__init__`` 'scaled_linear' branch,
``rescale_zero_terminal_snr``, ``set_timesteps`` 'trailing' branch, ``_get_variance``, ``step``
'v_prediction' branch with clip_sample=False, thresholding=False, use_clipped_model_output=False,
set_alpha_to_one=True).  It is pinned only by the anchors recorded in SURVEY.md section 8a row S
(tests/test_oracle_ddim.py) and by the invariants alpha_bar[999] == 0 and last-step sigma == 0.

All scalar arithmetic is float32, mirroring the 0-dim float32 torch tensors diffusers computes
with (``x ** 0.5`` on a torch tensor lowers to sqrt, ``x ** 2`` to x*x).
"""
import numpy as np

F32 = np.float32


def torch_like_linspace(start, end, steps):
    """torch.linspace(float32) CPU semantics: step = (end-start)/(steps-1); the first half counts
    up from start, the second half counts down from end (aten RangeFactories)."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    idx = np.arange(steps)
    half = steps // 2
    up = (start + step * idx.astype(np.float32)).astype(np.float32)
    down = (end - step * (steps - 1 - idx).astype(np.float32)).astype(np.float32)
    return np.where(idx < half, up, down).astype(np.float32)


def make_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                        rescale_betas_zero_snr=True):
    """betas 'scaled_linear' -> optional zero-terminal-SNR rescale -> cumprod(1 - beta), float32."""
    # diffusers: torch.linspace(beta_start**0.5, beta_end**0.5, T, dtype=float32) ** 2 (python-float sqrt)
    betas = torch_like_linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps) ** 2
    betas = betas.astype(np.float32)
    if rescale_betas_zero_snr:
        alphas = F32(1.0) - betas
        abar_sqrt = np.sqrt(np.cumprod(alphas, dtype=np.float32))
        a0, aT = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
        abar_sqrt = abar_sqrt - aT
        abar_sqrt = abar_sqrt * F32(a0 / F32(a0 - aT))
        abar = abar_sqrt * abar_sqrt
        alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]]).astype(np.float32)
        betas = F32(1.0) - alphas
    alphas = (F32(1.0) - betas).astype(np.float32)
    return np.cumprod(alphas, dtype=np.float32)


def trailing_timesteps(num_inference_steps, num_train_timesteps=1000):
    """set_timesteps, timestep_spacing='trailing': round(arange(T, 0, -T/n)) - 1."""
    ratio = num_train_timesteps / num_inference_steps
    return np.round(np.arange(num_train_timesteps, 0, -ratio)).astype(np.int64) - 1


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule='scaled_linear', prediction_type='v_prediction',
                 rescale_betas_zero_snr=True, timestep_spacing='trailing', clip_sample=False):
        if beta_schedule != 'scaled_linear' or prediction_type != 'v_prediction' or \
                timestep_spacing != 'trailing' or clip_sample:
            raise NotImplementedError('oracle restates only the configuration in ckpts/ezaudio-*.yml `diff:`')
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = make_alphas_cumprod(num_train_timesteps, beta_start, beta_end, rescale_betas_zero_snr)
        self.final_alpha_cumprod = F32(1.0)  # set_alpha_to_one=True (diffusers default)
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        self.timesteps = trailing_timesteps(n, self.num_train_timesteps)

    def coefficients(self, t, eta):
        """Scalars of one v-prediction DDIM step.
        x_prev = c_x0 * x0 + c_dir * eps + sigma * z with
        x0 = sa*x - sb*v, eps = sa*v + sb*x."""
        t = int(t)
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.final_alpha_cumprod
        b_t = F32(1.0) - a_t
        b_prev = F32(1.0) - a_prev
        variance = F32(b_prev / b_t) * F32(F32(1.0) - F32(a_t / a_prev))
        sigma = F32(eta) * np.sqrt(variance)
        with np.errstate(invalid='ignore'):
            c_dir = np.sqrt(F32(F32(F32(1.0) - a_prev) - F32(sigma * sigma)))  # NaN for n=25 at t=999, as in diffusers
        return dict(sa=np.sqrt(a_t), sb=np.sqrt(b_t), c_x0=np.sqrt(a_prev), c_dir=c_dir, sigma=sigma)

    def step(self, model_output, t, sample, eta, noise=None):
        c = self.coefficients(t, eta)
        v = np.asarray(model_output, dtype=np.float32)
        x = np.asarray(sample, dtype=np.float32)
        x0 = c['sa'] * x - c['sb'] * v
        eps = c['sa'] * v + c['sb'] * x
        prev = c['c_x0'] * x0 + c['c_dir'] * eps
        if eta > 0:
            prev = prev + c['sigma'] * np.asarray(noise, dtype=np.float32)
        return prev.astype(np.float32)
