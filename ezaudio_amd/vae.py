"""Oobleck VAE (decoder, encoder, bottleneck sampler) on the MI355X path.

Mirrors the reference's call surface:
  * ``OobleckDecoder(**config)``  -- /root/reference/src/modules/stable_vae/models/autoencoders.py:149-190
    (``use_snake=True``, ``final_tanh=False``, ``out_channels=1`` as in ckpts/vae/config.json; anything else raises),
    ``load_state_dict`` takes the reference checkpoint keys (``decoder.layers.N...weight_g / weight_v / bias / alpha / beta``),
    ``__call__(z[B, latent, L]) -> audio[B, 1, L * prod(strides)]``.
  * ``Autoencoder(ckpt_path, model_type='stable_vae', quantization_first=True)`` -- src/modules/autoencoder_wrapper.py:7-83:
    ``ae(embedding=z)`` decodes; ``ae(audio=wav)`` = ``OobleckEncoder`` (autoencoders.py:115-147) + ``VAEBottleneck.encode``
    (models/bottleneck.py:67-87) -> latents for ``editing_audio``.

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


def pack_conv_weight(w):
    """Conv1d weight [Co, Ci, K] -> GEMM operand [Co][K * Ci] (tap-major): with token-major activations and `halo` zero rows in
    front, out[l][co] = sum_{tap, ci} x[l + tap * dilation][ci] * W[co][tap * Ci + ci] -- K tile t of the GEMM reads the
    activation rows shifted by (t // (Ci / 64)) * dilation (GemmArgs.conv_*).  A strided conv (k = 2s, stride s) uses the same
    matrix on the buffer viewed as [T / s + 1][s * Ci]."""
    co, ci, k = w.shape
    return w.permute(0, 2, 1).reshape(co, k * ci).contiguous()


def pack_conv_transpose_weight(w, s):
    """ConvTranspose1d weight [Ci, Co, 2s] (stride s, padding ceil(s / 2)) -> GEMM operand [s * Co][2 * Ci]:
    out[q][r * Co + co] = x[q] . w[:, co, r] + x[q - 1] . w[:, co, r + s]; read as [(L + 1) s][Co] and shifted by the padding
    this IS the up-sampled sequence."""
    ci, co, k = w.shape
    assert k == 2 * s
    return w.reshape(ci, co, 2, s).permute(3, 1, 2, 0).reshape(s * co, 2 * ci).contiguous()


class _OobleckNet:
    """Buffers, launches and the ResidualUnit shared by decoder and encoder."""

    def _init_common(self, channels, latent_dim, c_mults, strides, device):
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
        self.tile = 6      # 128x64 tiles: best of tools/bench_vae.py at 10 s (M up to 120000, N = 128..5120)
        self._w = None
        self._bufs = {}

    # ---- weight packing (host, once) -----------------------------------------------------------------------------------
    def _pack_conv(self, sd, used, name, bias=True):
        w = _fold_weight_norm(sd[name + '.weight_g'], sd[name + '.weight_v'])      # [Co, Ci, K]
        used.update({name + '.weight_g', name + '.weight_v'})
        b = None
        if bias:
            b = sd[name + '.bias'].to(self.device)
            used.add(name + '.bias')
        co, ci, k = w.shape
        return dict(w=pack_conv_weight(w).to(self.device, torch.bfloat16), b=b, co=co, ci=ci, k=k)

    def _pack_snake(self, sd, used, name):
        used.update({name + '.alpha', name + '.beta'})
        a = torch.exp(sd[name + '.alpha'])
        ib = 1.0 / (torch.exp(sd[name + '.beta']) + 1e-9)                           # blocks.py:317-318,351-356
        return dict(a=a.to(self.device), ib=ib.to(self.device))

    def _pack_unit(self, sd, used, q):
        return dict(s0=self._pack_snake(sd, used, q + '.0'), c7=self._pack_conv(sd, used, q + '.1'),
                    s1=self._pack_snake(sd, used, q + '.2'), c1=self._pack_conv(sd, used, q + '.3'))

    # ---- buffers -------------------------------------------------------------------------------------------------------
    def _buf(self, tag, rows, cols, dtype):
        """Activation buffer [rows][cols] for role `tag`.  ONE allocation per (tag, width), grown to the longest sequence seen and
        handed out as a view: a long-running process that decodes / encodes many different lengths (`generate_audio(length=...)`,
        `editing_audio` on arbitrary clips) keeps a bounded working set instead of one full buffer set per length (the widths are
        fixed by the architecture, one per level).  Haloed buffers rely on their halo rows being zero and only interiors are
        ever written, so a view with a NEW row count over an old allocation is re-zeroed once, on the current stream (the one
        the kernels run on): stale interior rows would otherwise sit where the new halo is."""
        key = (tag, cols, dtype)
        need = rows * cols
        ent = self._bufs.get(key)
        if ent is None or ent[0].numel() < need:
            ent = [torch.zeros(need, dtype=dtype, device=self.device), rows]
            self._bufs[key] = ent
        elif ent[1] != rows:
            ent[0][:max(need, ent[1] * cols)].zero_()
            ent[1] = rows
        return ent[0][:need].view(rows, cols)

    def release_buffers(self):
        """Drop the cached activation buffers (they are re-created on the next call)."""
        self._bufs = {}

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

    def _residual_units(self, units, x_ptr, L, C, st):
        """three ResidualUnits (autoencoders.py:38-61), dilations 1, 3, 9; returns the pointer of the fp32 [L][C] result"""
        f32, bf16 = torch.float32, torch.bfloat16
        for ui, (unit, d) in enumerate(zip(units, (1, 3, 9))):
            hb = self._buf(f'h{3 * d}', L + 6 * d, C, bf16)
            self._snake(x_ptr, C, unit['s0'], hb, 3 * d, L, C, st)
            t = self._buf('t', L, C, f32)
            self._gemm(hb.data_ptr(), C, unit['c7'], t.data_ptr(), C, L, C, 7 * C, C // 64, d * C * 2, st)
            tb = self._buf('h0', L, C, bf16)
            self._snake(t.data_ptr(), C, unit['s1'], tb, 0, L, C, st)
            xn = self._buf(f'r{ui & 1}', L, C, f32)
            self._gemm(tb.data_ptr(), C, unit['c1'], xn.data_ptr(), C, L, C, C, 0, 0, st, resid_ptr=x_ptr, ldr=C)
            x_ptr = xn.data_ptr()
        return x_ptr

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


class OobleckDecoder(_OobleckNet):
    """autoencoders.py:149-190"""

    def __init__(self, out_channels=1, channels=128, latent_dim=128, c_mults=(1, 2, 4, 8), strides=(2, 4, 6, 10),
                 use_snake=True, antialias_activation=False, use_nearest_upsample=False, final_tanh=False, device='cuda'):
        if not use_snake or antialias_activation or use_nearest_upsample or final_tanh or out_channels != 1:
            raise NotImplementedError('only the EzAudio VAE recipe is built: snake activations, transposed-conv upsampling, '
                                      'mono output without tanh (ckpts/vae/config.json)')
        self._init_common(channels, latent_dim, c_mults, strides, device)

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len('decoder.'):] if k.startswith('decoder.') else k): torch.as_tensor(v).detach().float().cpu()
              for k, v in sd.items() if not k.startswith(('encoder.', 'bottleneck.'))}
        used = set()

        def conv_t(name, s):
            w = _fold_weight_norm(sd[name + '.weight_g'], sd[name + '.weight_v'])      # [Ci, Co, 2s]
            used.update({name + '.weight_g', name + '.weight_v', name + '.bias'})
            ci, co, k = w.shape
            b = sd[name + '.bias'].repeat(s).to(self.device)
            return dict(w=pack_conv_transpose_weight(w, s).to(self.device, torch.bfloat16), b=b, co=co, ci=ci, s=s)

        n = len(self.strides)
        w = {'conv_in': self._pack_conv(sd, used, 'layers.0'), 'blocks': []}
        for bi in range(n):
            s = self.strides[n - 1 - bi]
            p = f'layers.{1 + bi}.layers'
            w['blocks'].append(dict(snake=self._pack_snake(sd, used, p + '.0'), up=conv_t(p + '.1', s),
                                    units=[self._pack_unit(sd, used, f'{p}.{2 + u}.layers') for u in range(3)]))
        w['snake_out'] = self._pack_snake(sd, used, f'layers.{1 + n}')
        wo = _fold_weight_norm(sd[f'layers.{2 + n}.weight_g'], sd[f'layers.{2 + n}.weight_v'])   # [1, C, 7]
        used.update({f'layers.{2 + n}.weight_g', f'layers.{2 + n}.weight_v'})
        w['conv_out'] = wo[0].t().contiguous().to(self.device)                                     # [7][C] fp32
        if strict and set(sd) - used:
            raise KeyError(f'unexpected keys in VAE decoder state dict: {sorted(set(sd) - used)[:5]}')
        self._w = w
        return self

    def _decode_one(self, zt, out, st):
        """zt fp32 [L][latent] (token-major), out fp32 [L * ratio]."""
        w = self._w
        L, lat = zt.shape
        f32, bf16 = torch.float32, torch.bfloat16
        # conv_in: k7 pad 3
        xb = self._buf('h3in', L + 6, lat, bf16)
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
            x_ptr = self._residual_units(blk['units'], x_ptr, L, C, st)
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


class OobleckEncoder(_OobleckNet):
    """autoencoders.py:115-147.  ``latent_dim`` is the encoder's output width (2 x the VAE latent: mean | scale).

    ``__call__(wav[B, 1, T]) -> [B, latent_dim, T // prod(strides)]`` (reference layout).  A strided WNConv1d (k = 2s, stride s,
    padding s/2) is a plain GEMM: with s/2 zero halo rows in front, output row q reads the 2 s Ci contiguous values that start
    at token q s -- the activation buffer viewed as [T/s + 1][s Ci] with K = 2 s Ci spanning two view rows.
    """

    def __init__(self, in_channels=1, channels=128, latent_dim=256, c_mults=(1, 2, 4, 8), strides=(2, 4, 6, 10),
                 use_snake=True, antialias_activation=False, device='cuda'):
        if not use_snake or antialias_activation or in_channels != 1:
            raise NotImplementedError('only the EzAudio VAE recipe is built: mono input, snake activations')
        self._init_common(channels, latent_dim, c_mults, strides, device)

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len('encoder.'):] if k.startswith('encoder.') else k): torch.as_tensor(v).detach().float().cpu()
              for k, v in sd.items() if not k.startswith(('decoder.', 'bottleneck.'))}
        used = set()
        n = len(self.strides)
        wi = _fold_weight_norm(sd['layers.0.weight_g'], sd['layers.0.weight_v'])                   # [C, 1, 7]
        used.update({'layers.0.weight_g', 'layers.0.weight_v', 'layers.0.bias'})
        w = {'conv_in': dict(w=wi[:, 0].t().contiguous().to(self.device), b=sd['layers.0.bias'].to(self.device)), 'blocks': []}
        for bi in range(n):
            p = f'layers.{1 + bi}.layers'
            w['blocks'].append(dict(units=[self._pack_unit(sd, used, f'{p}.{u}.layers') for u in range(3)],
                                    snake=self._pack_snake(sd, used, p + '.3'), down=self._pack_conv(sd, used, p + '.4'),
                                    s=self.strides[bi]))
        w['snake_out'] = self._pack_snake(sd, used, f'layers.{1 + n}')
        w['conv_out'] = self._pack_conv(sd, used, f'layers.{2 + n}')
        if strict and set(sd) - used:
            raise KeyError(f'unexpected keys in VAE encoder state dict: {sorted(set(sd) - used)[:5]}')
        self._w = w
        return self

    def _encode_one(self, wav, st):
        """wav fp32 [T] -> fp32 [L][latent_dim] token-major (a cached buffer)."""
        w = self._w
        T = wav.shape[0]
        f32, bf16 = torch.float32, torch.bfloat16
        C = self.channels
        x = self._buf('x0', T, C, f32)
        self._check(self.lib.ezvae_conv_in1(wav.data_ptr(), w['conv_in']['w'].data_ptr(), w['conv_in']['b'].data_ptr(),
                                            x.data_ptr(), T, C, st))
        x_ptr, L = x.data_ptr(), T
        for blk in w['blocks']:
            s, dn = blk['s'], blk['down']
            assert dn['ci'] == C and dn['k'] == 2 * s
            x_ptr = self._residual_units(blk['units'], x_ptr, L, C, st)
            p = s // 2
            xb = self._buf(f'h{p}', L + 2 * p, C, bf16)
            self._snake(x_ptr, C, blk['snake'], xb, p, L, C, st)
            Lo = L // s                                              # floor((L + 2p - 2s) / s) + 1 for even s
            if Lo < 1:
                raise ValueError('audio too short for the encoder strides')
            y = self._buf('dn', Lo, dn['co'], f32)
            self._gemm(xb.data_ptr(), s * C, dn, y.data_ptr(), dn['co'], Lo, dn['co'], 2 * s * C, 0, 0, st)
            x_ptr, L, C = y.data_ptr(), Lo, dn['co']
        xb = self._buf('h1', L + 2, C, bf16)
        self._snake(x_ptr, C, w['snake_out'], xb, 1, L, C, st)
        co = w['conv_out']
        out = self._buf('out', L, co['co'], f32)
        self._gemm(xb.data_ptr(), C, co, out.data_ptr(), co['co'], L, co['co'], 3 * C, C // 64, C * 2, st)
        return out

    @torch.no_grad()
    def __call__(self, audio):
        if self._w is None:
            raise RuntimeError('load_state_dict() first')
        audio = torch.as_tensor(audio).to(self.device, torch.float32).contiguous()
        if audio.dim() != 3 or audio.shape[1] != 1:
            raise ValueError(f'expected audio [B, 1, T], got {tuple(audio.shape)}')
        st = torch.cuda.current_stream(self.device).cuda_stream
        outs = [self._encode_one(audio[b, 0], st).t().clone() for b in range(audio.shape[0])]
        return torch.stack(outs)

    forward = __call__


class VAEBottleneck:
    """models/bottleneck.py:73-90: encode = split (mean | scale), sample; decode = identity."""

    def __init__(self, device='cuda'):
        self.lib = _lib.load()
        self.device = torch.device(device)

    def encode(self, x, return_info=False, noise=None, **kwargs):
        if return_info:
            raise NotImplementedError('the KL term is a training quantity')
        x = torch.as_tensor(x).to(self.device, torch.float32)
        B, C2, L = x.shape
        lat = C2 // 2
        if noise is None:
            noise = torch.randn(B, lat, L, device=self.device, dtype=torch.float32)        # torch.randn_like(mean), bottleneck.py:69
        xt = x.transpose(1, 2).contiguous()
        z = torch.empty(B, lat, L, device=self.device, dtype=torch.float32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        for b in range(B):
            rc = self.lib.ezvae_sample(xt[b].data_ptr(), noise[b].data_ptr(), z[b].data_ptr(), L, lat, st)
            if rc != 0:
                raise RuntimeError(f'ezvae_sample failed ({rc})')
        return z

    def decode(self, x):
        return x


class _AE:
    """``Autoencoder.ae`` of the reference (an AudioAutoencoder): only the pieces the inference path touches."""

    def __init__(self, encoder, decoder, bottleneck):
        self.encoder = encoder
        self.decoder = decoder
        self.bottleneck = bottleneck


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
        encoder = None
        enc = config['model'].get('encoder')
        if enc is not None and any(k.startswith('encoder.') for k in state_dict):
            if enc['type'] != 'oobleck' or config['model'].get('bottleneck', {}).get('type') != 'vae':
                raise NotImplementedError(f"encoder type {enc['type']} / bottleneck {config['model'].get('bottleneck')}")
            encoder = OobleckEncoder(device=device, **enc['config'])
            encoder.load_state_dict({k: v for k, v in state_dict.items() if k.startswith('encoder.')})
        self.ae = _AE(encoder, decoder, VAEBottleneck(device))
        self.model_type = model_type
        self.quantization_first = quantization_first

    def __call__(self, audio=None, embedding=None):
        if audio is not None:                                                            # autoencoder_wrapper.py:69-73
            if self.ae.encoder is None:
                raise RuntimeError('this Autoencoder was loaded without encoder weights')
            return self.ae.bottleneck.encode(self.ae.encoder(audio))
        if embedding is not None:
            return self.ae.decoder(embedding)
        raise ValueError('Either audio or embedding must be provided.')

    forward = __call__

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
