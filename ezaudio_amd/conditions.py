"""Control-signal extractors.  ``EnergyExtractor`` / ``Conditioner`` mirror the reference's
src/models/conditions/energy.py:8-56 and condition_wrapper.py:9-43 (energy only: the shipped ControlNet config,
ckpts/controlnet/energy_l.yml:46-52, uses nothing else).  Pre-processing on torch ops: 1000 frames per call."""
import numpy as np
import torch
import torch.nn.functional as F


class EnergyExtractor:
    def __init__(self, hop_size=512, window_size=1024, padding='reflect', min_db=-60, norm=True, quantize_levels=None):
        self.hop_size, self.window_size, self.padding = hop_size, window_size, padding
        self.min_db, self.norm, self.quantize_levels = min_db, norm, quantize_levels

    def __call__(self, audio):
        n_frames = int(audio.size(-1) // self.hop_size)
        pad = (self.window_size - self.hop_size) // 2
        sq = F.pad(audio, (pad, pad), mode=self.padding) ** 2                     # energy.py:24-28
        energy = sq.unfold(-1, self.window_size, self.hop_size)[:, :n_frames].mean(dim=-1)   # framed mean square (:31-33)
        gain = torch.maximum(energy, torch.tensor(np.power(10, self.min_db / 10), device=audio.device, dtype=energy.dtype))
        gain_db = 10 * torch.log10(gain)
        if self.norm:                                                              # :42-50
            max_db = torch.max(gain_db, dim=-1, keepdim=True)[0]
            gain_db = (gain_db - self.min_db) / (max_db - self.min_db + 1e-8)
        if self.quantize_levels is not None:
            gain_db = torch.round(gain_db * (self.quantize_levels - 1)) / (self.quantize_levels - 1)
        return gain_db.unsqueeze(-1)                                               # [B, T, 1]


class Conditioner:
    def __init__(self, condition_type, **kwargs):
        if condition_type != 'energy':
            raise NotImplementedError(f'condition_type={condition_type!r}: only "energy" ships with a ControlNet checkpoint')
        self.conditioner = EnergyExtractor(**kwargs)

    def to(self, device):
        return self

    def __call__(self, waveform, latent_shape):
        cond = self.conditioner(waveform).permute(0, 2, 1).contiguous()            # B C T (condition_wrapper.py:26-29)
        if len(latent_shape) != 3:
            raise NotImplementedError('1-D latents only')
        return cond
