"""Oobleck VAE: oracle vs goldens minted from the reference's OobleckDecoder / OobleckEncoder
on CPU; the HIP decoder vs the goldens and vs the oracle at the real 10 s size on the GPU."""
import os

import numpy as np
import pytest

from oracle import vae as V
from oracle.weights import uniform_pm1
from tests.util import GOLDEN, record, rel_l2


def dec_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = dict(getattr(V, str(g['cfg_name'])))
    L = int(g['L'])
    sd = V.make_vae_state_dict(cfg, int(g['seed_w']))
    z = (1.2 * uniform_pm1(f'vae_z_{name}', 2 * cfg['latent_dim'] * L, int(g['seed_in']))).reshape(2, cfg['latent_dim'], L)
    return cfg, sd, z.astype(np.float32), g


def enc_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = dict(getattr(V, str(g['cfg_name'])))
    T = int(g['T'])
    sd = V.make_vae_state_dict(cfg, int(g['seed_w']), encoder=True)
    wav = (0.5 * uniform_pm1(f'vae_wav_{name}', 2 * T, int(g['seed_in']))).reshape(2, 1, T)
    return cfg, sd, wav.astype(np.float32), g


@pytest.mark.parametrize('name', ['vae_dec_tiny', 'vae_dec'])
def test_decoder_oracle_matches_reference_golden(name):
    cfg, sd, z, g = dec_case(name)
    audio = V.DecoderOracle(cfg, sd)(z)
    assert audio.shape == g['audio'].shape == (2, 1, z.shape[2] * int(np.prod(cfg['strides'])))
    assert rel_l2(audio, g['audio']) < 2e-5


@pytest.mark.parametrize('name', ['vae_enc_tiny', 'vae_enc'])
def test_encoder_oracle_matches_reference_golden(name):
    cfg, sd, wav, g = enc_case(name)
    lat = V.EncoderOracle(cfg, sd)(wav)
    assert lat.shape == g['latent'].shape == (2, 2 * cfg['latent_dim'], wav.shape[2] // int(np.prod(cfg['strides'])))
    assert rel_l2(lat, g['latent']) < 2e-5


def test_oracle_ops_against_torch_functional():
    import torch
    import torch.nn.functional as F
    x = uniform_pm1('x', 2 * 6 * 50, 0).reshape(2, 6, 50)
    w = uniform_pm1('w', 5 * 6 * 7, 1).reshape(5, 6, 7)
    b = uniform_pm1('b', 5, 2)
    for stride, pad, dil in ((1, 3, 1), (1, 9, 3), (2, 1, 1), (1, 27, 9)):
        ref = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=pad, dilation=dil)
        np.testing.assert_allclose(V.conv1d(x, w, b, stride, pad, dil), ref.numpy(), rtol=1e-5, atol=1e-6)
    for s in (2, 4, 6, 10, 3):
        wt = uniform_pm1(f'wt{s}', 6 * 5 * 2 * s, 3).reshape(6, 5, 2 * s)
        p = -(-s // 2)
        ref = F.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride=s, padding=p)
        np.testing.assert_allclose(V.conv_transpose1d(x, wt, b, s, p), ref.numpy(), rtol=1e-5, atol=1e-6)
    g = 1.0 + 0.3 * uniform_pm1('g', 5, 4).reshape(5, 1, 1)
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 5, 7))
    conv.weight_g.data = torch.from_numpy(g)
    conv.weight_v.data = torch.from_numpy(w)
    conv(torch.zeros(1, 6, 8))
    np.testing.assert_allclose(V.weight_norm(g, w), conv.weight.detach().numpy(), rtol=1e-5, atol=1e-7)
    from ezaudio_amd.vae import _fold_weight_norm
    np.testing.assert_allclose(_fold_weight_norm(torch.from_numpy(g), torch.from_numpy(w)).numpy(), V.weight_norm(g, w), rtol=1e-6)
    al, be = 0.3 * uniform_pm1('al', 6, 5), 0.3 * uniform_pm1('be', 6, 6)
    t = torch.from_numpy(x)
    ref = t + (1.0 / (torch.exp(torch.from_numpy(be))[None, :, None] + 1e-9)) * torch.sin(t * torch.exp(torch.from_numpy(al))[None, :, None]) ** 2
    np.testing.assert_allclose(V.snake_beta(x, al, be), ref.numpy(), rtol=1e-5, atol=1e-6)


def _tap_gemm(xh, W, M, K, cpb, tap_rows):
    """numpy model of ezvae_gemm's addressing on a token-major [rows][C] buffer: K tile t (64 wide) of output row m reads
    xh[m + (t // cpb) * tap_rows][(t % cpb) * 64 : +64]."""
    C = xh.shape[1]
    out = np.zeros((M, W.shape[0]), dtype=np.float64)
    for t in range(K // 64):
        tap, sub = divmod(t, cpb)
        rows = np.arange(M) + tap * tap_rows
        out += xh[rows][:, sub * 64:(sub + 1) * 64].astype(np.float64) @ W[:, t * 64:(t + 1) * 64].astype(np.float64).T
    return out


def test_conv_as_gemm_algebra_of_the_hip_path():
    """The weight re-layouts and halo / tap addressing that ezaudio_amd/vae.py feeds to the GEMM, checked in numpy against
    nn.Conv1d / ConvTranspose1d semantics (oracle ops): dilated k7, transposed k = 2s, strided k = 2s."""
    import torch
    from ezaudio_amd.vae import pack_conv_transpose_weight, pack_conv_weight
    Ci, Co, L = 64, 128, 37
    x = uniform_pm1('agx', Ci * L, 1).reshape(1, Ci, L)
    xt = x[0].T                                                        # token-major [L][Ci]
    for d in (1, 3, 9):
        w = uniform_pm1(f'agw{d}', Co * Ci * 7, 2).reshape(Co, Ci, 7)
        W = pack_conv_weight(torch.from_numpy(w)).numpy()
        xh = np.zeros((L + 6 * d, Ci), np.float32); xh[3 * d:3 * d + L] = xt
        got = _tap_gemm(xh, W, L, 7 * Ci, Ci // 64, d)
        np.testing.assert_allclose(got.T[None], V.conv1d(x, w, None, padding=3 * d, dilation=d), rtol=1e-4, atol=1e-5)
    for s in (2, 4, 6, 10):
        wt = uniform_pm1(f'agt{s}', Ci * Co * 2 * s, 3).reshape(Ci, Co, 2 * s)
        W = pack_conv_transpose_weight(torch.from_numpy(wt), s).numpy()
        xh = np.zeros((L + 2, Ci), np.float32); xh[1:1 + L] = xt        # [x[-1] = 0 | x | x[L] = 0]
        # A pointer = row of x[0]; tap 1 = the previous row (tap_rows = -1); M = L + 1 output rows
        out = np.zeros((L + 1, s * Co))
        for q in range(L + 1):
            out[q] = W[:, :Ci].astype(np.float64) @ xh[1 + q] + W[:, Ci:].astype(np.float64) @ xh[q]
        p = -(-s // 2)
        up = out.reshape((L + 1) * s, Co)[p:p + L * s]
        np.testing.assert_allclose(up.T[None], V.conv_transpose1d(x, wt, None, stride=s, padding=p), rtol=1e-4, atol=1e-5)
        # strided conv (encoder): halo p in front, buffer viewed as [T / s + 1][s * Ci], plain GEMM with K = 2 s Ci
        T = 4 * s + 3                                                  # not a multiple of the stride
        xs = uniform_pm1(f'ags{s}', Ci * T, 4).reshape(1, Ci, T)
        wd = uniform_pm1(f'agd{s}', Co * Ci * 2 * s, 5).reshape(Co, Ci, 2 * s)
        Wd = pack_conv_weight(torch.from_numpy(wd)).numpy()
        buf = np.zeros((T + 2 * p, Ci), np.float32); buf[p:p + T] = xs[0].T
        flat = np.concatenate([buf.reshape(-1), np.zeros(s * Ci, np.float32)])
        Lo = T // s
        got = np.stack([Wd.astype(np.float64) @ flat[q * s * Ci:q * s * Ci + 2 * s * Ci] for q in range(Lo)])
        np.testing.assert_allclose(got.T[None], V.conv1d(xs, wd, None, stride=s, padding=p), rtol=1e-4, atol=1e-5)


def test_decoder_rejects_unbuilt_recipes():
    from ezaudio_amd.vae import OobleckDecoder
    for kw in (dict(use_snake=False), dict(final_tanh=True), dict(out_channels=2), dict(use_nearest_upsample=True),
               dict(channels=96), dict(strides=(2, 3, 4, 5))):
        with pytest.raises(NotImplementedError):
            OobleckDecoder(**kw)


# ----------------------------------------------------------------------------------------------------------------------
# GPU: the HIP decoder through the C ABI
# ----------------------------------------------------------------------------------------------------------------------
def _hip_decoder(cfg, sd):
    from ezaudio_amd.vae import OobleckDecoder
    dec = OobleckDecoder(out_channels=1, channels=cfg['channels'], latent_dim=cfg['latent_dim'], c_mults=cfg['c_mults'],
                         strides=cfg['strides'], use_snake=True, final_tanh=False, device='cuda')
    return dec.load_state_dict(sd)


# Tolerance: GEMM operands (activations after SnakeBeta, folded weights) are bf16 with fp32 accumulation, residual stream /
# snake / final conv fp32: rel-L2 2e-2 and max-abs 6 % of the output's max through the 35 convolutions of the decoder.
VAE_REL, VAE_MAX = 2e-2, 0.06


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['vae_dec_tiny', 'vae_dec'])
def test_hip_decoder_matches_reference_golden(name):
    import torch
    cfg, sd, z, g = dec_case(name)
    dec = _hip_decoder(cfg, sd)
    audio = dec(torch.from_numpy(z).cuda()).cpu().numpy()
    assert audio.shape == g['audio'].shape
    assert np.isfinite(audio).all()
    r = rel_l2(audio, g['audio'])
    m = np.abs(audio - g['audio']).max() / np.abs(g['audio']).max()
    record(f'{name}: rel_l2 {r:.3e} max/max {m:.3e}')
    assert r < VAE_REL and m < VAE_MAX
    # a second call reuses the cached halo buffers: must be bitwise identical, and batch rows independent
    again = dec(torch.from_numpy(z[1:]).cuda()).cpu().numpy()
    assert np.array_equal(again[0], audio[1])


@pytest.mark.gpu
def test_hip_decoder_buffer_cache_is_bounded_across_lengths():
    """A long-running process decodes many different lengths: the activation cache must not grow per length, and a buffer that
    is re-used with a new geometry must still have clean halo rows (the result of a repeated length is bitwise unchanged)."""
    import torch
    cfg = dict(V.VAE_TINY)
    sd = V.make_vae_state_dict(cfg, 5)
    dec = _hip_decoder(cfg, sd)
    lat = cfg['latent_dim']
    zs = {L: torch.from_numpy((1.2 * uniform_pm1(f'vae_z_len{L}', lat * L, 3)).reshape(1, lat, L).astype(np.float32)).cuda()
          for L in (40, 17, 64, 23)}
    first = {L: dec(z).clone() for L, z in zs.items()}
    torch.cuda.synchronize()
    n_bufs = len(dec._bufs)
    held = sum(e[0].numel() * e[0].element_size() for e in dec._bufs.values())
    for L in (23, 64, 17, 40, 17):
        assert torch.equal(dec(zs[L]), first[L]), L
    torch.cuda.synchronize()
    assert len(dec._bufs) == n_bufs
    assert sum(e[0].numel() * e[0].element_size() for e in dec._bufs.values()) == held
    ref = V.DecoderOracle(cfg, sd)(zs[17].cpu().numpy())
    assert rel_l2(first[17].cpu().numpy(), ref) < VAE_REL


@pytest.mark.gpu
def test_hip_decoder_full_length_vs_oracle():
    """10 s of audio (250 latent frames -> 120000 samples), the size generate_audio() decodes."""
    import torch
    cfg = dict(V.VAE_DEFAULT)
    sd = V.make_vae_state_dict(cfg, 6)
    L = 250
    z = (1.2 * uniform_pm1('vae_z_full', cfg['latent_dim'] * L, 7)).reshape(1, cfg['latent_dim'], L).astype(np.float32)
    ref = V.DecoderOracle(cfg, sd)(z)
    dec = _hip_decoder(cfg, sd)
    audio = dec(torch.from_numpy(z).cuda()).cpu().numpy()
    assert audio.shape == (1, 1, 120000)
    r = rel_l2(audio, ref)
    m = np.abs(audio - ref).max() / np.abs(ref).max()
    record(f'full: rel_l2 {r:.3e} max/max {m:.3e}')
    assert r < VAE_REL and m < VAE_MAX
    # shorter latent afterwards (editing / variable length): halo geometry changes, buffers are per-shape
    z2 = z[:, :, :77]
    a2 = dec(torch.from_numpy(z2).cuda()).cpu().numpy()
    assert rel_l2(a2, V.DecoderOracle(cfg, sd)(z2)) < VAE_REL


def _hip_encoder(cfg, sd):
    from ezaudio_amd.vae import OobleckEncoder
    enc = OobleckEncoder(in_channels=1, channels=cfg['channels'], latent_dim=2 * cfg['latent_dim'], c_mults=cfg['c_mults'],
                         strides=cfg['strides'], use_snake=True, device='cuda')
    return enc.load_state_dict(sd)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['vae_enc_tiny', 'vae_enc'])
def test_hip_encoder_matches_reference_golden(name):
    import torch
    cfg, sd, wav, g = enc_case(name)
    enc = _hip_encoder(cfg, sd)
    lat = enc(torch.from_numpy(wav).cuda()).cpu().numpy()
    assert lat.shape == g['latent'].shape
    r = rel_l2(lat, g['latent'])
    m = np.abs(lat - g['latent']).max() / np.abs(g['latent']).max()
    record(f'{name}: rel_l2 {r:.3e} max/max {m:.3e}')
    assert r < VAE_REL and m < VAE_MAX


@pytest.mark.gpu
def test_hip_encoder_ragged_length_and_bottleneck():
    """A waveform whose length is not a multiple of any stride (editing_audio crops arbitrarily): every strided conv floors."""
    import torch
    from ezaudio_amd.vae import VAEBottleneck
    cfg = dict(V.VAE_DEFAULT)
    sd = V.make_vae_state_dict(cfg, 6, encoder=True)
    T = 480 * 9 + 317
    wav = (0.5 * uniform_pm1('vae_wav_ragged', T, 3)).reshape(1, 1, T).astype(np.float32)
    ref = V.EncoderOracle(cfg, sd)(wav)
    lat = _hip_encoder(cfg, sd)(torch.from_numpy(wav).cuda())
    assert tuple(lat.shape) == ref.shape
    assert rel_l2(lat.cpu().numpy(), ref) < VAE_REL
    noise = uniform_pm1('vae_noise', ref.shape[1] // 2 * ref.shape[2], 4).reshape(1, ref.shape[1] // 2, ref.shape[2])
    z = VAEBottleneck('cuda').encode(torch.from_numpy(ref).cuda(), noise=torch.from_numpy(noise).cuda()).cpu().numpy()
    np.testing.assert_allclose(z, V.vae_sample(ref[:, :128], ref[:, 128:], noise), rtol=1e-5, atol=1e-5)
    z0 = VAEBottleneck('cuda').encode(torch.from_numpy(ref).cuda())
    assert z0.shape == (1, 128, ref.shape[2]) and torch.isfinite(z0).all()


MINI_VAE = dict(channels=64, c_mults=[1, 2], strides=[2, 4], latent_dim=128, out_channels=1)   # DiT-compatible latent width


def _mini_autoencoder(device='cuda'):
    import torch
    from ezaudio_amd.vae import Autoencoder
    cfg = MINI_VAE
    sd = {k: torch.from_numpy(v) for k, v in V.make_vae_state_dict(cfg, 5).items()}
    sd.update({k: torch.from_numpy(v) for k, v in V.make_vae_state_dict(cfg, 5, encoder=True).items()})
    common = dict(channels=cfg['channels'], c_mults=cfg['c_mults'], strides=cfg['strides'], use_snake=True)
    config = {'model': {'encoder': {'type': 'oobleck', 'config': dict(in_channels=1, latent_dim=2 * cfg['latent_dim'], **common)},
                        'decoder': {'type': 'oobleck', 'config': dict(out_channels=1, latent_dim=cfg['latent_dim'], final_tanh=False, **common)},
                        'bottleneck': {'type': 'vae'}}}
    return Autoencoder(model_type='stable_vae', quantization_first=True, config=config, state_dict=sd, device=device), config, sd


@pytest.mark.gpu
def test_autoencoder_wrapper_surface(tmp_path):
    """src/modules/autoencoder_wrapper.py:68-83: embedding -> audio, audio -> sampled latent; ckpt + config.json loading."""
    import json
    import torch
    from ezaudio_amd.vae import Autoencoder
    ae, config, sd = _mini_autoencoder()
    z = torch.from_numpy(uniform_pm1('z', 128 * 16, 1).reshape(1, 128, 16)).cuda()
    wav = ae(embedding=z)
    assert wav.shape == (1, 1, 16 * 8)
    lat = ae(audio=wav)
    assert lat.shape == (1, 128, 16)
    with pytest.raises(ValueError):
        ae()
    with pytest.raises(NotImplementedError):
        Autoencoder(model_type='dac', config=config, state_dict=sd)
    # the reference's on-disk format: <dir>/config.json + torch checkpoint {'state_dict': {'autoencoder.<key>': tensor}}
    with open(tmp_path / 'config.json', 'w') as f:
        json.dump(config, f)
    torch.save({'state_dict': {'autoencoder.' + k: v for k, v in sd.items()}}, tmp_path / 'vae.pt')
    ae2 = Autoencoder(ckpt_path=str(tmp_path / 'vae.pt'), model_type='stable_vae', quantization_first=True)
    assert torch.equal(ae2(embedding=z), wav)


class _Tok:
    """Stand-in for T5Tokenizer (no checkpoints offline): deterministic ids, per-prompt valid length."""

    def __call__(self, texts, max_length, padding, truncation, return_tensors):
        import torch
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        mask = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, t in enumerate(texts):
            n = max(1, min(max_length, len(t.split()) + 1))
            ids[i, :n] = torch.tensor([(j % 97) + 1 for j in range(len(t), len(t) + n)])
            mask[i, :n] = 1
        return type('Batch', (), dict(input_ids=ids, attention_mask=mask))()


class _Enc:
    def __init__(self, dim):
        self.dim = dim

    def __call__(self, input_ids, attention_mask):
        import torch
        g = torch.Generator().manual_seed(7)
        table = torch.randn(128, self.dim, generator=g).to(input_ids.device)
        return type('Out', (), dict(last_hidden_state=table[input_ids % 128]))()


@pytest.mark.gpu
def test_public_api_text_to_audio_and_editing_end_to_end(tmp_path, monkeypatch):
    """api/ezaudio.py:101-207 through the native sampler AND the native VAE: prompt -> waveform, waveform -> edited waveform."""
    import sys
    import types
    import yaml
    import torch
    import ezaudio_amd
    from ezaudio_amd import api as A
    from ezaudio_amd.config import load_yaml_with_includes
    from oracle.weights import make_state_dict, model_config
    base = os.path.join(os.path.dirname(ezaudio_amd.__file__), 'configs', 'ezaudio-xl.yml')
    params = load_yaml_with_includes(base)
    cfg = model_config('xs')
    params['model'] = dict(cfg)
    params['text_encoder']['dim'] = cfg['context_dim']
    yml = tmp_path / 'mini.yml'
    with open(yml, 'w') as f:
        yaml.safe_dump(params, f)
    monkeypatch.setitem(A.configs, 'mini', {'path': str(tmp_path / 'none.pt'), 'url': '', 'config': str(yml)})
    ae, _, _ = _mini_autoencoder()
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 1).items()}
    ez = A.EzAudio('mini', autoencoder=ae, tokenizer=_Tok(), text_encoder=_Enc(cfg['context_dim']), state_dict=sd)
    sr, audio = ez.generate_audio('a dog barking', length=2, ddim_steps=20, random_seed=3)
    assert sr == 24000 and audio.shape == (100 * 8,) and np.isfinite(audio).all() and audio.std() > 0
    sr, again = ez.generate_audio('a dog barking', length=2, ddim_steps=20, random_seed=3)
    assert np.array_equal(audio, again)
    # editing: librosa is absent offline; the API only uses librosa.load(file, sr=) -> (waveform, sr)
    src = (0.3 * uniform_pm1('edit_src', 24000, 9)).astype(np.float32)
    monkeypatch.setitem(sys.modules, 'librosa', types.SimpleNamespace(load=lambda f, sr: (src.copy(), sr)))
    sr, edited = ez.editing_audio('rain', boundary=0.1, gt_file='unused.wav', mask_start=0.3, mask_length=0.2, ddim_steps=20,
                                  random_seed=3)
    assert edited.shape == src.shape and np.isfinite(edited).all()
    keep = np.ones(len(src), bool)
    keep[round(0.2 * 24000):round(0.6 * 24000)] = False
    assert np.array_equal(edited[keep], (src / (np.abs(src).max() + 1e-9))[keep])      # outside the re-synthesised chunk: untouched


@pytest.mark.gpu
def test_public_api_controlnet_text_plus_energy_to_audio(tmp_path, monkeypatch):
    """api/controlnet.py:113-160 end to end: reference recording -> energy curve -> ControlNet + backbone sampler -> VAE -> waveform."""
    import sys
    import types
    import yaml
    import torch
    import ezaudio_amd
    from ezaudio_amd import api as A
    from ezaudio_amd.config import load_yaml_with_includes
    from oracle.controlnet import CN_DEFAULT, make_controlnet_state_dict
    from oracle.weights import make_state_dict, model_config
    base = os.path.join(os.path.dirname(ezaudio_amd.__file__), 'configs', 'controlnet', 'energy_l.yml')
    params = load_yaml_with_includes(base)
    cfg = model_config('xs')
    params['model'] = dict(cfg)
    yml = tmp_path / 'mini_cn.yml'
    with open(yml, 'w') as f:
        yaml.safe_dump(params, f)
    monkeypatch.setitem(A.controlnet_configs, 'mini_energy', {'path': str(tmp_path / 'none.pt'), 'url': '', 'config': str(yml)})
    ae, _, _ = _mini_autoencoder()
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(cfg, 1).items()}
    csd = {k: torch.from_numpy(v) for k, v in make_controlnet_state_dict(cfg, CN_DEFAULT, 1).items()}
    ez = A.EzAudio_ControlNet('mini_energy', autoencoder=ae, tokenizer=_Tok(), text_encoder=_Enc(cfg['context_dim']), state_dict=sd,
                              controlnet_state_dict=csd)
    ref = (0.3 * uniform_pm1('cn_ref', 24000 * 3, 9)).astype(np.float32)          # a 3 s reference recording
    ref[24000:36000] *= 0.05                                                       # with a quiet second in the middle
    monkeypatch.setitem(sys.modules, 'librosa', types.SimpleNamespace(load=lambda f, sr: (ref.copy(), sr)))
    sr, audio = ez.generate_audio('a dog barking', 'unused.wav', ddim_steps=20, random_seed=3)
    # the mini VAE up-samples 8x instead of 480x, so the 500 latent frames give 4000 samples; the API trims to len(ref) at most
    assert sr == 24000 and audio.ndim == 1 and audio.shape[0] == 500 * 8 and np.isfinite(audio).all() and audio.std() > 0
    sr, again = ez.generate_audio('a dog barking', 'unused.wav', ddim_steps=20, random_seed=3)
    assert np.array_equal(audio, again)
    # conditioning_scale 0 must remove the ControlNet's influence entirely: same result as the plain backbone sampler
    sr, off = ez.generate_audio('a dog barking', 'unused.wav', ddim_steps=20, random_seed=3, conditioning_scale=0)
    assert not np.array_equal(off, audio)
    plain = A.EzAudio.__new__(A.EzAudio)
    plain.__dict__.update(autoencoder=ez.autoencoder, unet=ez.unet, tokenizer=ez.tokenizer, text_encoder=ez.text_encoder,
                          noise_scheduler=ez.noise_scheduler, params=ez.params, device=ez.device)
    # (ControlNet residuals change the skips, so the backbone takes the row-kernel path for LN([x | skip]) whatever their scale; the plain sampler's default is the
    # LayerNorm-algebra form of that edge, option skip_z -- the same math with other bf16 rounding points.  Same path: equal to fp32 rounding; default path: inside the loop gate)
    sr, default_audio = plain.generate_audio('a dog barking', length=10, guidance_scale=3.5, guidance_rescale=0, ddim_steps=20, random_seed=3)
    assert ez.unet.lib.ezdit_set_option(ez.unet._h, b'skip_z', 0) == 0
    try:
        sr, base_audio = plain.generate_audio('a dog barking', length=10, guidance_scale=3.5, guidance_rescale=0, ddim_steps=20, random_seed=3)
    finally:
        assert ez.unet.lib.ezdit_set_option(ez.unet._h, b'skip_z', 1) == 0
    assert rel_l2(off, base_audio) < 1e-5
    assert rel_l2(off, default_audio) < 5e-2
