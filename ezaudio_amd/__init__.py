"""ezaudio_amd: the EzAudio diffusion-transformer denoising path, MI355X (gfx950) native.

Python host code over a C-ABI HIP library (include/ezdit.h); see DESIGN.md.
"""
from .config import configs, load_yaml_with_includes  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch / the built library
    if name in ('EzAudio',):
        from .api import EzAudio
        return EzAudio
    if name in ('MaskDiT',):
        from .denoiser import MaskDiT
        return MaskDiT
    if name in ('DDIMScheduler',):
        from .scheduler import DDIMScheduler
        return DDIMScheduler
    if name in ('DiTControlNet',):
        from .controlnet import DiTControlNet
        return DiTControlNet
    if name in ('EzAudio_ControlNet',):
        from .api import EzAudio_ControlNet
        return EzAudio_ControlNet
    if name in ('inference', 'inference_controlnet', 'LatentSampler'):
        from . import sampler
        return getattr(sampler, name)
    raise AttributeError(name)
