"""CPU restatement of the Oobleck VAE decoder / encoder (stable-audio-tools recipe) used by EzAudio.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  Pinned by tests/golden/vae_*.npz, minted by oracle/mint_golden.py
from the reference's own ``OobleckDecoder`` / ``OobleckEncoder`` modules.

Reference (all under /root/reference/src/modules/stable_vae/):
  models/autoencoders.py:38-61    ResidualUnit  (snake, WNConv1d k7 dilated, snake, WNConv1d k1, + x)
  models/autoencoders.py:63-80    EncoderBlock  (3 ResidualUnits, snake, WNConv1d k=2s stride s pad ceil(s/2))
  models/autoencoders.py:82-113   DecoderBlock  (snake, WNConvTranspose1d k=2s stride s pad ceil(s/2), 3 ResidualUnits)
  models/autoencoders.py:115-147  OobleckEncoder
  models/autoencoders.py:149-190  OobleckDecoder
  models/blocks.py:317-358        SnakeBeta (log-scale alpha/beta), snake_beta()
  models/nn/layers.py:9-14        WNConv1d / WNConvTranspose1d = torch.nn.utils.weight_norm (dim 0)
  models/bottleneck.py:56-87      vae_sample / VAEBottleneck.encode
  ckpts/vae/config.json           channels 128, c_mults [1,2,4,8], strides [2,4,6,10], latent 128, snake, no tanh
  src/modules/autoencoder_wrapper.py:68-83   process_stable_vae (quantization_first=True: decoder(z) directly)

Layout here is the reference's: [B, C, T] float32 arrays.
"""
import math

import numpy as np

from .weights import uniform_pm1

VAE_DEFAULT = dict(channels=128, c_mults=[1, 2, 4, 8], strides=[2, 4, 6, 10], latent_dim=128, out_channels=1)
VAE_TINY = dict(channels=64, c_mults=[1, 2], strides=[2, 4], latent_dim=64, out_channels=1)


# ---------------------------------------------------------------------------------------------
# synthetic checkpoints (key names of OobleckDecoder.state_dict() under "decoder.")
# ---------------------------------------------------------------------------------------------
def decoder_param_shapes(cfg):
    """name -> (shape, kind) in module order (autoencoders.py:163-187)."""
    ch, lat = cfg['channels'], cfg['latent_dim']
    cm = [1] + list(cfg['c_mults'])
    st = cfg['strides']
    sh = {}

    def wn_conv(prefix, co, ci, k, bias=True, transposed=False):
        shape = (ci, co, k) if transposed else (co, ci, k)
        sh[prefix + '.weight_g'] = ((shape[0], 1, 1), 'g')
        sh[prefix + '.weight_v'] = (shape, 'vt' if transposed else 'v')
        if bias:
            sh[prefix + '.bias'] = ((co,), 'bias')

    def snake(prefix, c):
        sh[prefix + '.alpha'] = ((c,), 'log')
        sh[prefix + '.beta'] = ((c,), 'log')

    wn_conv('decoder.layers.0', cm[-1] * ch, lat, 7)
    li = 1
    for i in range(len(cm) - 1, 0, -1):
        ci, co, s = cm[i] * ch, cm[i - 1] * ch, st[i - 1]
        p = f'decoder.layers.{li}.layers'
        snake(p + '.0', ci)
        wn_conv(p + '.1', co, ci, 2 * s, transposed=True)
        for u in range(3):
            q = f'{p}.{2 + u}.layers'
            snake(q + '.0', co)
            wn_conv(q + '.1', co, co, 7)
            snake(q + '.2', co)
            wn_conv(q + '.3', co, co, 1)
        li += 1
    snake(f'decoder.layers.{li}', cm[0] * ch)
    wn_conv(f'decoder.layers.{li + 1}', cfg['out_channels'], cm[0] * ch, 7, bias=False)
    return sh


def encoder_param_shapes(cfg, in_channels=1):
    """OobleckEncoder (autoencoders.py:130-144); encoder latent is 2 x latent_dim (mean | scale) for the VAE bottleneck."""
    ch, lat = cfg['channels'], 2 * cfg['latent_dim']
    cm = [1] + list(cfg['c_mults'])
    st = cfg['strides']
    sh = {}

    def wn_conv(prefix, co, ci, k):
        sh[prefix + '.weight_g'] = ((co, 1, 1), 'g')
        sh[prefix + '.weight_v'] = ((co, ci, k), 'v')
        sh[prefix + '.bias'] = ((co,), 'bias')

    def snake(prefix, c):
        sh[prefix + '.alpha'] = ((c,), 'log')
        sh[prefix + '.beta'] = ((c,), 'log')

    wn_conv('encoder.layers.0', cm[0] * ch, in_channels, 7)
    li = 1
    for i in range(len(cm) - 1):
        ci, co, s = cm[i] * ch, cm[i + 1] * ch, st[i]
        p = f'encoder.layers.{li}.layers'
        for u in range(3):
            q = f'{p}.{u}.layers'
            snake(q + '.0', ci)
            wn_conv(q + '.1', ci, ci, 7)
            snake(q + '.2', ci)
            wn_conv(q + '.3', ci, ci, 1)
        snake(p + '.3', ci)
        wn_conv(p + '.4', co, ci, 2 * s)
        li += 1
    snake(f'encoder.layers.{li}', cm[-1] * ch)
    wn_conv(f'encoder.layers.{li + 1}', lat, cm[-1] * ch, 3)
    return sh


def make_vae_state_dict(cfg, seed=0, encoder=False):
    """Deterministic, everywhere-non-trivial weights: name -> float32 ndarray."""
    shapes = encoder_param_shapes(cfg) if encoder else decoder_param_shapes(cfg)
    sd = {}
    for name, (shape, kind) in shapes.items():
        n = int(np.prod(shape))
        u = uniform_pm1(name, n, seed).reshape(shape)
        if kind == 'v':
            fan_in = shape[1] * shape[2]
            sd[name] = (np.float32(math.sqrt(3.0 / fan_in)) * u).astype(np.float32)
        elif kind == 'vt':   # ConvTranspose1d [Ci, Co, 2s]: every output sample sees 2 taps x Ci inputs
            sd[name] = (np.float32(math.sqrt(3.0 / (2 * shape[0]))) * u).astype(np.float32)
        elif kind == 'g':
            v = sd.get(name[:-1] + 'v')
            if v is None:      # weight_g precedes weight_v in module order
                vshape, vkind = shapes[name[:-1] + 'v']
                vu = uniform_pm1(name[:-1] + 'v', int(np.prod(vshape)), seed).reshape(vshape)
                sc = math.sqrt(3.0 / (vshape[1] * vshape[2])) if vkind == 'v' else math.sqrt(3.0 / (2 * vshape[0]))
                v = np.float32(sc) * vu
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
            sd[name] = (nrm * np.float32(0.7) * (np.float32(1.0) + np.float32(0.2) * u)).astype(np.float32)
        elif kind == 'bias':
            sd[name] = (np.float32(0.05) * u).astype(np.float32)
        elif kind == 'log':
            sd[name] = (np.float32(0.3) * u).astype(np.float32)
        else:
            raise KeyError(kind)
    return sd


# ---------------------------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------------------------
def weight_norm(g, v):
    """torch.nn.utils.weight_norm, dim=0: w = g * v / ||v|| with the norm over every dim but 0."""
    nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    return (g.astype(np.float64) * v / nrm).astype(np.float32)


def snake_beta(x, alpha, beta):
    """blocks.py:317-318 with alpha_logscale (blocks.py:351-356)."""
    a = np.exp(alpha)[None, :, None]
    b = np.exp(beta)[None, :, None]
    return x + (np.float32(1.0) / (b + np.float32(1e-9))) * np.sin(x * a) ** 2


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1):
    """nn.Conv1d: x [B,Ci,L], w [Co,Ci,K] -> [B,Co,Lout]."""
    B, Ci, L = x.shape
    Co, _, K = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
    Lout = (L + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    out = np.zeros((B, Co, Lout), dtype=np.float32)
    for k in range(K):
        seg = xp[:, :, k * dilation:k * dilation + stride * (Lout - 1) + 1:stride]
        wk = np.ascontiguousarray(w[:, :, k])
        for bi in range(B):
            out[bi] += wk @ np.ascontiguousarray(seg[bi])
    if b is not None:
        out += b[None, :, None]
    return out


def conv_transpose1d(x, w, b=None, stride=1, padding=0):
    """nn.ConvTranspose1d: x [B,Ci,L], w [Ci,Co,K] -> [B,Co,(L-1)*stride - 2*padding + K]."""
    B, Ci, L = x.shape
    _, Co, K = w.shape
    full = np.zeros((B, Co, (L - 1) * stride + K), dtype=np.float32)
    for k in range(K):
        wk = np.ascontiguousarray(w[:, :, k].T)
        for bi in range(B):
            full[bi, :, k:k + (L - 1) * stride + 1:stride] += wk @ x[bi]
    out = full[:, :, padding:full.shape[2] - padding]
    if b is not None:
        out = out + b[None, :, None]
    return out


class _Net:
    def __init__(self, sd, prefix):
        self.sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
        self.prefix = prefix

    def w(self, name):
        return weight_norm(self.sd[name + '.weight_g'], self.sd[name + '.weight_v'])

    def b(self, name):
        return self.sd.get(name + '.bias')

    def snake(self, name, x):
        return snake_beta(x, self.sd[name + '.alpha'], self.sd[name + '.beta'])

    def residual_unit(self, p, x, dilation):
        """autoencoders.py:38-61"""
        h = self.snake(p + '.0', x)
        h = conv1d(h, self.w(p + '.1'), self.b(p + '.1'), padding=(dilation * 6) // 2, dilation=dilation)
        h = self.snake(p + '.2', h)
        h = conv1d(h, self.w(p + '.3'), self.b(p + '.3'))
        return h + x


class DecoderOracle(_Net):
    """OobleckDecoder.forward (autoencoders.py:149-190), use_snake=True, final_tanh=False."""

    def __init__(self, cfg, sd):
        super().__init__(sd, 'decoder')
        self.cfg = cfg

    def __call__(self, z, taps=None):
        cfg = self.cfg
        st = cfg['strides']
        n = len(st)
        x = conv1d(np.asarray(z, dtype=np.float32), self.w('decoder.layers.0'), self.b('decoder.layers.0'), padding=3)
        if taps is not None:
            taps['conv_in'] = x
        for bi in range(n):
            s = st[n - 1 - bi]
            p = f'decoder.layers.{1 + bi}.layers'
            x = self.snake(p + '.0', x)
            x = conv_transpose1d(x, self.w(p + '.1'), self.b(p + '.1'), stride=s, padding=math.ceil(s / 2))
            if taps is not None:
                taps[f'up{bi}'] = x
            for u, d in enumerate((1, 3, 9)):
                x = self.residual_unit(f'{p}.{2 + u}.layers', x, d)
            if taps is not None:
                taps[f'block{bi}'] = x
        x = self.snake(f'decoder.layers.{1 + n}', x)
        return conv1d(x, self.w(f'decoder.layers.{2 + n}'), None, padding=3)

    def flops(self, L):
        """2 x MACs of every convolution for a latent of L frames."""
        cfg = self.cfg
        ch = cfg['channels']
        cm = [1] + list(cfg['c_mults'])
        st = cfg['strides']
        f = 2 * L * cm[-1] * ch * cfg['latent_dim'] * 7
        for i in range(len(cm) - 1, 0, -1):
            ci, co, s = cm[i] * ch, cm[i - 1] * ch, st[i - 1]
            f += 2 * L * ci * co * 2 * s
            L *= s
            f += 3 * 2 * L * co * co * 8
        f += 2 * L * ch * 7 * cfg['out_channels']
        return f


class EncoderOracle(_Net):
    """OobleckEncoder.forward (autoencoders.py:115-147), use_snake=True -> [B, 2*latent, L/prod(strides)]."""

    def __init__(self, cfg, sd):
        super().__init__(sd, 'encoder')
        self.cfg = cfg

    def __call__(self, audio, taps=None):
        st = self.cfg['strides']
        x = conv1d(np.asarray(audio, dtype=np.float32), self.w('encoder.layers.0'), self.b('encoder.layers.0'), padding=3)
        if taps is not None:
            taps['conv_in'] = x
        for bi, s in enumerate(st):
            p = f'encoder.layers.{1 + bi}.layers'
            for u, d in enumerate((1, 3, 9)):
                x = self.residual_unit(f'{p}.{u}.layers', x, d)
            x = self.snake(p + '.3', x)
            x = conv1d(x, self.w(p + '.4'), self.b(p + '.4'), stride=s, padding=math.ceil(s / 2))
            if taps is not None:
                taps[f'block{bi}'] = x
        n = len(st)
        x = self.snake(f'encoder.layers.{1 + n}', x)
        return conv1d(x, self.w(f'encoder.layers.{2 + n}'), self.b(f'encoder.layers.{2 + n}'), padding=1)


def vae_sample(mean, scale, noise):
    """bottleneck.py:56-60 with the randn made explicit: softplus(scale)+1e-4 is the std."""
    stdev = np.logaddexp(np.float32(0.0), scale).astype(np.float32) + np.float32(1e-4)
    return noise * stdev + mean
