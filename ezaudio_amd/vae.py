"""Oobleck VAE decoder on the MI355X path.

Mirrors the reference's call surface:
  * ``OobleckDecoder(**config)``  -- /root/reference/src/modules/stable_vae/models/autoencoders.py:149-190
    (``use_snake=True``, ``final_tanh=False``, ``out_channels=1`` as in ckpts/vae/config.json; anything else raises),
    ``load_state_dict`` takes the reference checkpoint keys (``decoder.layers.N...weight_g / weight_v / bias / alpha / beta``),
    ``__call__(z[B, latent, L]) -> audio[B, 1, L * prod(strides)]``.
  * ``Autoencoder(ckpt_path, model_type='stable_vae', quantization_first=True)`` -- src/modules/autoencoder_wrapper.py:7-83:
    ``ae(embedding=z)`` decodes; ``ae(audio=wav)`` (the encoder) is not built yet and raises.

How it runs: every Conv1d / ConvTranspose1d is one launch of the same bf16 MFMA GEMM that serves the DiT (csrc/gemm.hip) over
token-major activations with zero halo rows (csrc/vae.hip explains the addressing); SnakeBeta is fused with the fp32 -> bf16
cast; residual adds ride in the GEMM epilogue.  weight_norm is folded into the weights once at load time.  The layer
sequence below is host code: it runs once per call (~70 launches), not per denoising step.  No CPU fallback.
"""
import json
import math
import os

import numpy as np
import torch

from . import _lib


def _fold_weight_norm(g, v):
    """torch.nn.utils.weight_norm (dim 0): w = g * v / ||v||, norm over all dims but the first (nn/layers.py:9-14)."""
    v64 = v.double()
    nrm = v64.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return (g.double() * v64 / nrm).float()


class OobleckDecoder:
    def __init__(self, out_channels=1, channels=128, latent_dim=128, c_mults=(1, 2, 4, 8), strides=(2, 4, 6, 10),
                 use_snake=True, antialias_activation=False, use_nearest_upsample=False, final_tanh=False, device='cuda'):
        if not use_snake or antialias_activation or use_nearest_upsample or final_tanh or out_channels != 1:
            raise NotImplementedError('only the EzAudio VAE recipe is built: snake activations, transposed-conv upsampling, '
                                      'mono output without tanh (ckpts/vae/config.json)')
        if channels % 64 or latent_dim % 64:
            raise NotImplementedError('channels and latent_dim must be multiples of 64 (GEMM K tile)')
        if any(s % 2 for s in strides):
            raise NotImplementedError('odd strides are not built (output length (L+1)s - 2 ceil(s/2) != L s)')
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.channels, self.latent_dim = channels, latent_dim
        self.c_mults = [1] + list(c_mults)
        self.strides = list(strides)
        self.ratio = int(np.prod(self.strides))
        self.tile = 2
        self._w = None
        self._bufs = {}

    # ---- weights -------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        sd = {(k[len('decoder.'):] if k.startswith('decoder.') else k): torch.as_tensor(v).detach().float().cpu()
              for k, v in sd.items() if not k.startswith(('encoder.', 'bottleneck.'))}
        used = set()

        def conv(name, bias=True):
            w = _fold_weight_norm(sd[name + '.weight_g'], sd[name + '.weight_v'])      # [Co, Ci, K]
            used.update({name + '.weight_g', name + '.weight_v'})
            b = None
            if bias:
                b = sd[name + '.bias'].to(self.device)
                used.add(name + '.bias')
            co, ci, k = w.shape
            wm = w.permute(0, 2, 1).reshape(co, k * ci).contiguous()                   # [Co][tap][Ci]
            return dict(w=wm.to(self.device, torch.bfloat16), b=b, co=co, ci=ci, k=k)

        def conv_t(name, s):
            w = _fold_weight_norm(sd[name + '.weight_g'], sd[name + '.weight_v'])      # [Ci, Co, 2s]
            used.update({name + '.weight_g', name + '.weight_v', name + '.bias'})
            ci, co, k = w.shape
            assert k == 2 * s
            # out[q][r*Co + co] = x[q] . w[:, co, r] + x[q-1] . w[:, co, r + s]
            wm = w.reshape(ci, co, 2, s).permute(3, 1, 2, 0).reshape(s * co, 2 * ci).contiguous()
            b = sd[name + '.bias'].repeat(s).to(self.device)
            return dict(w=wm.to(self.device, torch.bfloat16), b=b, co=co, ci=ci, s=s)

        def snake(name):
            used.update({name + '.alpha', name + '.beta'})
            a = torch.exp(sd[name + '.alpha'])
            ib = 1.0 / (torch.exp(sd[name + '.beta']) + 1e-9)                           # blocks.py:317-318,351-356
            return dict(a=a.to(self.device), ib=ib.to(self.device))

        n = len(self.strides)
        w = {'conv_in': conv('layers.0'), 'blocks': []}
        for bi in range(n):
            s = self.strides[n - 1 - bi]
            p = f'layers.{1 + bi}.layers'
            blk = dict(snake=snake(p + '.0'), up=conv_t(p + '.1', s), units=[])
            for u in range(3):
                q = f'{p}.{2 + u}.layers'
                blk['units'].append(dict(s0=snake(q + '.0'), c7=conv(q + '.1'), s1=snake(q + '.2'), c1=conv(q + '.3')))
            w['blocks'].append(blk)
        w['snake_out'] = snake(f'layers.{1 + n}')
        wo = _fold_weight_norm(sd[f'layers.{2 + n}.weight_g'], sd[f'layers.{2 + n}.weight_v'])   # [1, C, 7]
        used.update({f'layers.{2 + n}.weight_g', f'layers.{2 + n}.weight_v'})
        w['conv_out'] = wo[0].t().contiguous().to(self.device)                                     # [7][C] fp32
        if strict and set(sd) - used:
            raise KeyError(f'unexpected keys in VAE decoder state dict: {sorted(set(sd) - used)[:5]}')
        self._w = w
        return self

    # ---- buffers -------------------------------------------------------------------------------------------------------
    def _buf(self, tag, rows, cols, dtype):
        key = (tag, rows, cols, dtype)
        b = self._bufs.get(key)
        if b is None:
            b = torch.zeros(rows, cols, dtype=dtype, device=self.device)    # halo rows stay zero: only interiors are written
            self._bufs[key] = b
        return b

    # ---- launches ------------------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f'ezvae call failed ({rc}): {self.lib.ezdit_last_error().decode()}')

    def _snake(self, x_ptr, ldx, sn, out, halo, L, C, st):
        """out[halo + l][:] = bf16(snake(x[l][:])); rows outside [halo, halo + L) are never written (zero)."""
        self._check(self.lib.ezvae_snake_bf16(x_ptr, ldx, sn['a'].data_ptr() if sn else None, sn['ib'].data_ptr() if sn else None,
                                              out.data_ptr() + halo * out.shape[1] * 2, out.shape[1], L, C, st))

    def _gemm(self, a_ptr, lda, cw, out_ptr, ldo, M, N, K, cpb, tap_bytes, st, resid_ptr=None, ldr=0):
        self._check(self.lib.ezvae_gemm(a_ptr, lda, cw['w'].data_ptr(), K, N, cw['b'].data_ptr() if cw['b'] is not None else None,
                                        resid_ptr, ldr, out_ptr, ldo, M, N, K, cpb, tap_bytes, self.tile, st))

    def _decode_one(self, zt, out, st):
        """zt fp32 [L][latent] (token-major), out fp32 [L * ratio]."""
        w = self._w
        L, lat = zt.shape
        f32, bf16 = torch.float32, torch.bfloat16
        # conv_in: k7 pad 3
        xb = self._buf('h3', L + 6, lat, bf16)
        self._snake(zt.data_ptr(), lat, None, xb, 3, L, lat, st)
        c = w['conv_in']
        x = self._buf('x0', L, c['co'], f32)
        self._gemm(xb.data_ptr(), lat, c, x.data_ptr(), c['co'], L, c['co'], 7 * lat, lat // 64, lat * 2, st)
        x_ptr, C = x.data_ptr(), c['co']
        for blk in w['blocks']:
            up = blk['up']
            s, ci, co = up['s'], up['ci'], up['co']
            assert ci == C
            # snake -> [1 | L | 1] halo, transposed conv as one GEMM over (x[q], x[q-1]), q = 0..L
            xb = self._buf('h1', L + 2, ci, bf16)
            self._snake(x_ptr, C, blk['snake'], xb, 1, L, ci, st)
            y = self._buf('up', (L + 1) * s, co, f32)
            self._gemm(xb.data_ptr() + ci * 2, ci, up, y.data_ptr(), s * co, L + 1, s * co, 2 * ci, ci // 64, -ci * 2, st)
            p = math.ceil(s / 2)
            L, C = L * s, co
            x_ptr = y.data_ptr() + p * co * 4                      # rows p .. p + L of the [(L_in+1) s][Co] view
            for ui, (unit, d) in enumerate(zip(blk['units'], (1, 3, 9))):
                hb = self._buf(f'h{3 * d}', L + 6 * d, C, bf16)
                self._snake(x_ptr, C, unit['s0'], hb, 3 * d, L, C, st)
                t = self._buf('t', L, C, f32)
                self._gemm(hb.data_ptr(), C, unit['c7'], t.data_ptr(), C, L, C, 7 * C, C // 64, d * C * 2, st)
                tb = self._buf('h0', L, C, bf16)
                self._snake(t.data_ptr(), C, unit['s1'], tb, 0, L, C, st)
                xn = self._buf(f'r{ui & 1}', L, C, f32)
                self._gemm(tb.data_ptr(), C, unit['c1'], xn.data_ptr(), C, L, C, C, 0, 0, st, resid_ptr=x_ptr, ldr=C)
                x_ptr = xn.data_ptr()
        xb = self._buf('h3', L + 6, C, bf16)
        self._snake(x_ptr, C, w['snake_out'], xb, 3, L, C, st)
        self._check(self.lib.ezvae_conv_out1(xb.data_ptr(), C, w['conv_out'].data_ptr(), out.data_ptr(), L, C, st))

    @torch.no_grad()
    def __call__(self, z):
        if self._w is None:
            raise RuntimeError('load_state_dict() first')
        z = torch.as_tensor(z).to(self.device, torch.float32)
        if z.dim() != 3 or z.shape[1] != self.latent_dim:
            raise ValueError(f'expected latents [B, {self.latent_dim}, L], got {tuple(z.shape)}')
        B, _, L = z.shape
        zt = z.transpose(1, 2).contiguous()
        out = torch.empty(B, 1, L * self.ratio, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        for b in range(B):
            self._decode_one(zt[b], out[b, 0], st)
        return out

    forward = __call__

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def flops(self, L):
        ch = self.channels
        cm = self.c_mults
        f = 2 * L * cm[-1] * ch * self.latent_dim * 7
        for i in range(len(cm) - 1, 0, -1):
            ci, co, s = cm[i] * ch, cm[i - 1] * ch, self.strides[i - 1]
            f += 2 * L * ci * co * 2 * s
            L *= s
            f += 3 * 2 * L * co * co * 8
        return f + 2 * L * ch * 7


class _AE:
    """``Autoencoder.ae`` of the reference (an AudioAutoencoder): only the pieces the inference path touches."""

    def __init__(self, decoder):
        self.decoder = decoder
        self.encoder = None


class Autoencoder:
    """src/modules/autoencoder_wrapper.py:7-83 for model_type='stable_vae' (the only type EzAudio's configs use)."""

    def __init__(self, ckpt_path=None, model_type='stable_vae', quantization_first=True, device='cuda', config=None, state_dict=None):
        if model_type != 'stable_vae':
            raise NotImplementedError(f'Model type not implemented: {model_type}')
        if not quantization_first:
            raise NotImplementedError('quantization_first=False decodes through the VAE bottleneck sampler; EzAudio uses True '
                                      '(api/ezaudio.py:76-78)')
        if config is None:
            with open(os.path.join(os.path.dirname(ckpt_path), 'config.json')) as f:     # stable_vae/__init__.py:14-19
                config = json.load(f)
        dec = config['model']['decoder']
        if dec['type'] != 'oobleck':
            raise NotImplementedError(f"decoder type {dec['type']}")
        self.config = config
        decoder = OobleckDecoder(device=device, **dec['config'])
        if state_dict is None:
            sd = torch.load(ckpt_path, map_location='cpu')['state_dict']                 # stable_vae/__init__.py:25-28
            state_dict = {k[len('autoencoder.'):]: v for k, v in sd.items() if k.startswith('autoencoder.')}
        decoder.load_state_dict({k: v for k, v in state_dict.items() if k.startswith('decoder.')})
        self.ae = _AE(decoder)
        self.model_type = model_type
        self.quantization_first = quantization_first

    def __call__(self, audio=None, embedding=None):
        if audio is not None:
            raise NotImplementedError('the Oobleck encoder (editing_audio input) is not built yet; SURVEY.md section 8f rank 3')
        if embedding is not None:
            return self.ae.decoder(embedding)
        raise ValueError('Either audio or embedding must be provided.')

    forward = __call__

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
