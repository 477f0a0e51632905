"""Mint golden vectors from the REFERENCE ITSELF (runs only in the build container).

    python -m oracle.mint_golden [--only NAME] [--out tests/golden]

Imports the reference's own modules from /root/reference (read-only; nothing is copied), loads
the deterministic synthetic checkpoint of oracle/weights.py into ``MaskDiT`` via
``load_state_dict`` (after asserting that key set and shapes agree with the real module), runs it
in fp32 on CPU on the deterministic inputs of ``oracle.weights.make_inputs`` and stores
inputs-by-seed + outputs under tests/golden/.  /root/reference does not exist on the GPU box, so
tests only ever read the committed .npz files; this script is the provenance record.

The reference has no tests and ships no golden vectors, so these fixtures
are what pins oracle/dit.py and oracle/sampler.py.  The DDIM scheduler (third-party diffusers,
absent) is NOT pinned by them: the sampler fixture drives the reference's unmodified
``inference()`` with a scheduler object backed by oracle/ddim.py.
"""
import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = '/root/reference'


def _import_reference():
    import torch  # noqa: F401
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for stub in ('librosa', 'soundfile'):  # unused top-level imports of src/inference.py:5,7
        sys.modules.setdefault(stub, types.ModuleType(stub))
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.conditioners import MaskDiT
        from src.inference import inference
    return MaskDiT, inference


def build_reference(cfg, seed):
    import torch
    from .weights import make_state_dict
    MaskDiT, _ = _import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        m = MaskDiT(**cfg).eval()
    sd = make_state_dict(cfg, seed)
    ref_sd = m.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), (sorted(set(ref_sd) ^ set(sd))[:10])
    for k, v in ref_sd.items():
        assert tuple(v.shape) == sd[k].shape, (k, tuple(v.shape), sd[k].shape)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def mint_forward(name, size, L, Lc, timesteps, seed_w, seed_in, n_valid=(12, 1), with_gt=False,
                 cn_skips=False, B=2, out_dir='tests/golden'):
    import torch
    from .weights import model_config, make_inputs, uniform_pm1
    cfg = model_config(size)
    m, _ = build_reference(cfg, seed_w)
    inp = make_inputs(cfg, B=B, L=L, Lc=Lc, n_valid=n_valid, seed=seed_in, with_gt=with_gt)
    outs = {}
    torch.set_num_threads(os.cpu_count())
    with torch.no_grad():
        for t in timesteps:
            kw = {}
            if with_gt:
                kw = dict(gt=torch.from_numpy(inp['gt'].copy()), mae_mask_infer=torch.from_numpy(inp['gt_mask'].copy()))
            if cn_skips:
                # UDiT.forward(controlnet_skips=...) is reached through unet.model (inference_controlnet.py:89-99)
                x257, _ = m(torch.from_numpy(inp['x'].copy()), torch.tensor(t), None, forward_model=False, **kw)
                D = cfg['embed_dim']
                skips = [torch.from_numpy((0.1 * uniform_pm1(f'in.cn{i}', 2 * L * D, seed_in)).reshape(2, L, D))
                         for i in range(cfg['depth'] // 2)]
                pred = m.model(x257, torch.tensor(t), torch.from_numpy(inp['ctx']),
                               context_mask=torch.from_numpy(inp['ctx_mask']), cls_token=None,
                               controlnet_skips=list(skips))
            else:
                pred, _ = m(torch.from_numpy(inp['x'].copy()), torch.tensor(t), torch.from_numpy(inp['ctx']),
                            context_mask=torch.from_numpy(inp['ctx_mask']), cls_token=None, **kw)
            outs[f'pred_t{t}'] = pred.numpy().astype(np.float32)
    meta = dict(size=size, L=L, Lc=Lc, seed_w=seed_w, seed_in=seed_in, n_valid=list(n_valid),
                with_gt=with_gt, cn_skips=cn_skips, timesteps=list(timesteps), B=B)
    path = os.path.join(out_dir, f'dit_{name}.npz')
    np.savez(path, meta=np.array(repr(meta)), **outs)
    print('wrote', path, {k: (v.shape, float(v.std())) for k, v in outs.items()})


class _FakeTok:
    """Duck-typed tokenizer: text is a key into a table of pre-made (ids, mask)."""
    def __init__(self, table):
        self.table = table

    def __call__(self, text, max_length=None, padding=None, truncation=None, return_tensors=None):
        import torch
        key = text[0] if isinstance(text, (list, tuple)) else text
        ids, mask = self.table[key]
        return types.SimpleNamespace(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))


class _FakeT5:
    def __init__(self, emb):
        self.emb = emb

    def __call__(self, input_ids=None, attention_mask=None):
        import torch
        return types.SimpleNamespace(last_hidden_state=torch.from_numpy(self.emb[int(input_ids[0, 0])]))


class _OracleScheduler:
    """Scheduler with diffusers' call surface (set_timesteps / scale_model_input / step), backed by
    oracle/ddim.py; `step` pops pre-drawn noise so the loop is deterministic."""
    def __init__(self, diff, noises):
        from .ddim import DDIMOracle
        self.o = DDIMOracle(**diff)
        self.noises = list(noises)
        self.i = 0

    def set_timesteps(self, n):
        import torch
        self.o.set_timesteps(n)
        self.timesteps = torch.from_numpy(self.o.timesteps)

    def scale_model_input(self, x, t):
        return x

    def step(self, model_output, timestep, sample, eta, generator):
        import torch
        z = self.noises[self.i]
        self.i += 1
        out = self.o.step(model_output.numpy(), int(timestep), sample.numpy(), eta, z)
        return types.SimpleNamespace(prev_sample=torch.from_numpy(out))


def mint_controlnet(name, size, L, Lc, t, seed_w, seed_in, scale=1.0, row_stride=1, out_dir='tests/golden'):
    """Reference DiTControlNet residuals + the backbone's prediction with them (src/inference_controlnet.py:89-99).
    `t` may be a list of timesteps (keys then carry a `_t<t>` suffix); `row_stride` > 1 keeps every row_stride-th token row
    of each residual (the full set is 14 x 4.6 MB per timestep at XL width)."""
    import torch
    from .controlnet import CN_DEFAULT, make_controlnet_state_dict
    from .weights import model_config, make_inputs, uniform_pm1
    _import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.controlnet import DiTControlNet
    cfg = model_config(size)
    m, _ = build_reference(cfg, seed_w)
    ccfg = dict(cfg)
    ccfg.update({k: (list(v) if isinstance(v, list) else v) for k, v in CN_DEFAULT.items()})
    with contextlib.redirect_stdout(io.StringIO()):
        cn = DiTControlNet(**ccfg).eval()
    sd = make_controlnet_state_dict(cfg, CN_DEFAULT, seed_w)
    ref_sd = cn.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()) | {k for k in ref_sd if k.endswith('inv_freq')}, sorted(set(ref_sd) ^ set(sd))[:10]
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == v.shape, (k, tuple(ref_sd[k].shape), v.shape)
    cn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, n_valid=(7, 1), seed=seed_in)
    cond = (0.5 + 0.5 * uniform_pm1('in.cond', 2 * 2 * L, seed_in)).reshape(2, 1, 2 * L)
    torch.set_num_threads(os.cpu_count())
    ts = list(t) if isinstance(t, (list, tuple)) else [t]
    multi = isinstance(t, (list, tuple))
    arrs = {}
    for tt in ts:
        sfx = f'_t{tt}' if multi else ''
        with torch.no_grad():
            x257, _ = m(torch.from_numpy(inp['x'].copy()), torch.tensor(tt), None, forward_model=False)
            skips = cn(x257, torch.tensor(tt), torch.from_numpy(inp['ctx']), context_mask=torch.from_numpy(inp['ctx_mask']),
                       cls_token=None, condition=torch.from_numpy(cond), conditioning_scale=scale)
            res = [s_.numpy().copy() for s_ in skips]
            pred = m.model(x257, torch.tensor(tt), torch.from_numpy(inp['ctx']), context_mask=torch.from_numpy(inp['ctx_mask']),
                           cls_token=None, controlnet_skips=list(skips))
        arrs['pred' + sfx] = pred.numpy().astype(np.float32)
        for i, r in enumerate(res):
            arrs[f'res{i}{sfx}'] = np.ascontiguousarray(r[:, ::row_stride]).astype(np.float32)
        print('minted', name, tt, pred.shape, float(pred.std()), [round(float(r.std()), 4) for r in res], flush=True)
    meta = dict(size=size, L=L, Lc=Lc, t=t, seed_w=seed_w, seed_in=seed_in, scale=scale, row_stride=row_stride)
    path = os.path.join(out_dir, f'{name}.npz')
    np.savez(path, meta=np.array(repr(meta)), **arrs)
    print('wrote', path)


DIFF = dict(num_train_timesteps=1000, beta_schedule='scaled_linear', beta_start=0.00085, beta_end=0.012,
            prediction_type='v_prediction', rescale_betas_zero_snr=True, timestep_spacing='trailing',
            clip_sample=False)


def mint_sampler(name, size, L, Lc, steps, seed_w, seed_in, guidance_scale, guidance_rescale, eta,
                 with_gt=False, out_dir='tests/golden'):
    """Run the reference's unmodified inference() (src/inference.py:26-107) with duck-typed fakes."""
    import torch
    from .weights import model_config, make_inputs, uniform_pm1
    _, inference = _import_reference()
    cfg = model_config(size)
    m, _ = build_reference(cfg, seed_w)
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, seed=seed_in, with_gt=with_gt)
    C = cfg['out_chans']
    s3 = np.float32(np.sqrt(3.0))
    init = (uniform_pm1('smp.init', C * L, seed_in) * s3).reshape(1, C, L)
    noises = [(uniform_pm1(f'smp.z{i}', C * L, seed_in) * s3).reshape(1, C, L) for i in range(steps)]
    tok = _FakeTok({'prompt': (np.array([[0]]), inp['ctx_mask'][0:1].astype(np.int64)),
                    '': (np.array([[1]]), inp['ctx_mask'][1:2].astype(np.int64))})
    t5 = _FakeT5({0: inp['ctx'][0:1], 1: inp['ctx'][1:2]})
    sched = _OracleScheduler(DIFF, noises)
    params = dict(text_encoder=dict(max_length=Lc), model=cfg, autoencoder=dict(scale=1.0, shift=0.0))
    gt = torch.from_numpy(inp['gt'][0:1].copy()) if with_gt else None
    gm = torch.from_numpy(inp['gt_mask'][0:1].copy()) if with_gt else None
    # inference() draws the init noise itself from torch.Generator; inject ours by patching randn once
    real_randn = torch.randn
    torch.randn = lambda *a, **k: torch.from_numpy(init.copy())
    try:
        torch.set_num_threads(os.cpu_count())
        out = inference(lambda embedding: embedding, m, gt, gm, tok, t5, params, sched,
                        ['prompt'], None, L, guidance_scale, guidance_rescale, steps, eta, 2024, 'cpu')
    finally:
        torch.randn = real_randn
    meta = dict(size=size, L=L, Lc=Lc, steps=steps, seed_w=seed_w, seed_in=seed_in, with_gt=with_gt,
                guidance_scale=guidance_scale, guidance_rescale=guidance_rescale, eta=eta)
    path = os.path.join(out_dir, f'sampler_{name}.npz')
    np.savez(path, meta=np.array(repr(meta)), latent=out.numpy().astype(np.float32))
    print('wrote', path, out.shape, float(out.std()))


def mint_cn_sampler(name, size, L, Lc, steps, seed_w, seed_in, guidance_scale, guidance_rescale, eta, scale, out_dir='tests/golden'):
    """Run the reference's UNMODIFIED ControlNet sampler, src/inference_controlnet.py:27-129 `inference`, on the reference's own MaskDiT +
    DiTControlNet (synthetic checkpoints), with the duck-typed tokenizer / T5 / scheduler fakes of mint_sampler."""
    import importlib
    import torch
    from .controlnet import CN_DEFAULT, make_controlnet_state_dict
    from .weights import model_config, make_inputs, uniform_pm1
    _import_reference()
    for stub in ('pandas',):
        try:
            importlib.import_module(stub)
        except Exception:
            sys.modules.setdefault(stub, types.ModuleType(stub))
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.controlnet import DiTControlNet
        from src.inference_controlnet import inference as cn_inference
    cfg = model_config(size)
    m, _ = build_reference(cfg, seed_w)
    ccfg = dict(cfg)
    ccfg.update({k: (list(v) if isinstance(v, list) else v) for k, v in CN_DEFAULT.items()})
    with contextlib.redirect_stdout(io.StringIO()):
        cn = DiTControlNet(**ccfg).eval()
    sd = make_controlnet_state_dict(cfg, CN_DEFAULT, seed_w)
    cn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, seed=seed_in)
    C = cfg['out_chans']
    s3 = np.float32(np.sqrt(3.0))
    init = (uniform_pm1('smp.init', C * L, seed_in) * s3).reshape(1, C, L)
    noises = [(uniform_pm1(f'smp.z{i}', C * L, seed_in) * s3).reshape(1, C, L) for i in range(steps)]
    cond = (0.5 + 0.5 * uniform_pm1('smp.cond', 2 * L, seed_in)).reshape(1, 1, 2 * L).astype(np.float32)
    tok = _FakeTok({'prompt': (np.array([[0]]), inp['ctx_mask'][0:1].astype(np.int64)),
                    '': (np.array([[1]]), inp['ctx_mask'][1:2].astype(np.int64))})
    t5 = _FakeT5({0: inp['ctx'][0:1], 1: inp['ctx'][1:2]})
    sched = _OracleScheduler(DIFF, noises)
    params = dict(text_encoder=dict(max_length=Lc), model=cfg, autoencoder=dict(scale=1.0, shift=0.0))
    real_randn = torch.randn
    torch.randn = lambda *a, **k: torch.from_numpy(init.copy())
    try:
        torch.set_num_threads(os.cpu_count())
        out = cn_inference(lambda embedding: embedding, m, cn, None, None, torch.from_numpy(cond), tok, t5, params, sched,
                           ['prompt'], None, L, guidance_scale, guidance_rescale, steps, eta, 2024, scale, 'cpu')
    finally:
        torch.randn = real_randn
    meta = dict(size=size, L=L, Lc=Lc, steps=steps, seed_w=seed_w, seed_in=seed_in, guidance_scale=guidance_scale,
                guidance_rescale=guidance_rescale, eta=eta, scale=scale)
    path = os.path.join(out_dir, f'sampler_{name}.npz')
    np.savez(path, meta=np.array(repr(meta)), latent=out.numpy().astype(np.float32))
    print('wrote', path, out.shape, float(out.std()))


def mint_energy(name, out_dir='tests/golden'):
    """The reference's EnergyExtractor, loaded BY FILE PATH from src/models/conditions/energy.py:7-56 (the package __init__ pulls in unrelated
    modules), with the conditioner section of ckpts/controlnet/energy_l.yml:46-52, on deterministic waveforms."""
    import importlib.util
    import torch
    from .weights import uniform_pm1
    spec = importlib.util.spec_from_file_location('ref_energy', os.path.join(REF, 'src/models/conditions/energy.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ex = mod.EnergyExtractor(hop_size=240, window_size=1920, padding='reflect', min_db=-60, norm=True)
    arrs = {}
    for i, (n, amp) in enumerate(((24000, 0.1), (240000, 0.5), (7201, 0.02))):
        t = np.arange(n, dtype=np.float32)
        wav = (amp * uniform_pm1(f'energy.wav{i}', n, 7) * (0.55 + 0.45 * np.sin(2 * np.pi * t / 5000.0))).astype(np.float32).reshape(1, n)
        wav[:, n // 3:n // 3 + 2000] = 0.0   # a silent stretch: exercises the -60 dB floor
        with torch.no_grad():
            e = ex(torch.from_numpy(wav.copy())).numpy()
        arrs[f'n{i}'] = np.array([n, amp], dtype=np.float64)
        arrs[f'energy{i}'] = e.astype(np.float32)
        print('minted energy', i, e.shape, float(e.mean()))
    np.savez(os.path.join(out_dir, f'{name}.npz'), **arrs)


def _import_reference_vae():
    """OobleckDecoder / OobleckEncoder from the reference; third-party modules its file imports at the top but the
    Oobleck path never touches (torchaudio, alias_free_torch, vector_quantize_pytorch; autoencoders.py:7-8, bottleneck.py:6)
    and the in-tree audiotools (needs flatten_dict; pulled in only by nn/loss.py:6-7) are absent / unimportable here and stubbed."""
    import torch  # noqa: F401
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, attrs in (('torchaudio', ()), ('torchaudio.transforms', ()), ('alias_free_torch', ('Activation1d',)),
                        ('vector_quantize_pytorch', ('ResidualVQ', 'FSQ')), ('audiotools', ('AudioSignal', 'STFTParams'))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {}))
            sys.modules[name] = m
    sys.modules['torchaudio'].transforms = sys.modules['torchaudio.transforms']
    import warnings
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from src.modules.stable_vae.models.autoencoders import OobleckDecoder, OobleckEncoder
    return OobleckDecoder, OobleckEncoder


def _load_vae_module(mod, sd, prefix):
    import torch
    ref_sd = mod.state_dict()
    mine = {k[len(prefix):]: v for k, v in sd.items()}
    assert set(ref_sd.keys()) == set(mine.keys()), sorted(set(ref_sd) ^ set(mine))[:10]
    for k, v in ref_sd.items():
        assert tuple(v.shape) == mine[k].shape, (k, tuple(v.shape), mine[k].shape)
    mod.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in mine.items()})
    return mod.eval()


def mint_vae_decoder(name, cfg_name, L, seed_w, seed_in, out_dir='tests/golden'):
    """Reference OobleckDecoder (autoencoders.py:149-190) in fp32 on CPU on a deterministic latent [2, latent_dim, L]."""
    import torch
    import warnings
    from . import vae as V
    from .weights import uniform_pm1
    cfg = dict(getattr(V, cfg_name))
    OobleckDecoder, _ = _import_reference_vae()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dec = OobleckDecoder(out_channels=cfg['out_channels'], channels=cfg['channels'], latent_dim=cfg['latent_dim'],
                             c_mults=cfg['c_mults'], strides=cfg['strides'], use_snake=True, final_tanh=False)
    sd = V.make_vae_state_dict(cfg, seed_w)
    _load_vae_module(dec, sd, 'decoder.')
    z = (1.2 * uniform_pm1(f'vae_z_{name}', 2 * cfg['latent_dim'] * L, seed_in)).reshape(2, cfg['latent_dim'], L).astype(np.float32)
    torch.set_num_threads(os.cpu_count())
    with torch.no_grad():
        audio = dec(torch.from_numpy(z.copy())).numpy()
    np.savez_compressed(os.path.join(out_dir, f'{name}.npz'), cfg_name=cfg_name, L=L, seed_w=seed_w, seed_in=seed_in,
                        audio=audio.astype(np.float32))
    print(f'[mint] {name}: audio {audio.shape} std {audio.std():.4f} max {np.abs(audio).max():.4f}', flush=True)


def mint_vae_encoder(name, cfg_name, T, seed_w, seed_in, out_dir='tests/golden'):
    """Reference OobleckEncoder (autoencoders.py:115-147) on a deterministic waveform [2, 1, T] -> [2, 2*latent, T/ratio]."""
    import torch
    import warnings
    from . import vae as V
    from .weights import uniform_pm1
    cfg = dict(getattr(V, cfg_name))
    _, OobleckEncoder = _import_reference_vae()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        enc = OobleckEncoder(in_channels=1, channels=cfg['channels'], latent_dim=2 * cfg['latent_dim'],
                             c_mults=cfg['c_mults'], strides=cfg['strides'], use_snake=True)
    sd = V.make_vae_state_dict(cfg, seed_w, encoder=True)
    _load_vae_module(enc, sd, 'encoder.')
    wav = (0.5 * uniform_pm1(f'vae_wav_{name}', 2 * T, seed_in)).reshape(2, 1, T).astype(np.float32)
    torch.set_num_threads(os.cpu_count())
    with torch.no_grad():
        lat = enc(torch.from_numpy(wav.copy())).numpy()
    np.savez_compressed(os.path.join(out_dir, f'{name}.npz'), cfg_name=cfg_name, T=T, seed_w=seed_w, seed_in=seed_in,
                        latent=lat.astype(np.float32))
    print(f'[mint] {name}: latent {lat.shape} std {lat.std():.4f}', flush=True)


JOBS = {
    # name: (fn, kwargs)
    'xs':        (mint_forward, dict(size='xs', L=96, Lc=20, timesteps=[999, 499, 19], seed_w=1, seed_in=11, n_valid=(7, 1))),
    'xs64':      (mint_forward, dict(size='xs64', L=96, Lc=20, timesteps=[979, 19], seed_w=1, seed_in=11, n_valid=(7, 1))),
    'xs_edit':   (mint_forward, dict(size='xs', L=77, Lc=20, timesteps=[499], seed_w=1, seed_in=12, n_valid=(5, 1), with_gt=True)),
    'xs_cn':     (mint_forward, dict(size='xs', L=96, Lc=20, timesteps=[499], seed_w=1, seed_in=13, n_valid=(5, 1), cn_skips=True)),
    's':         (mint_forward, dict(size='s', L=500, Lc=100, timesteps=[999, 979, 499, 19], seed_w=1234, seed_in=11)),
    's64':       (mint_forward, dict(size='s64', L=500, Lc=100, timesteps=[999, 19], seed_w=1234, seed_in=11)),
    's_edit':    (mint_forward, dict(size='s', L=300, Lc=100, timesteps=[499], seed_w=1234, seed_in=12, with_gt=True)),
    'l':         (mint_forward, dict(size='l', L=500, Lc=100, timesteps=[499], seed_w=1234, seed_in=11)),
    'xl':        (mint_forward, dict(size='xl', L=500, Lc=100, timesteps=[499], seed_w=1234, seed_in=11)),
    'cn_xs':     (mint_controlnet, dict(size='xs', L=96, Lc=20, t=499, seed_w=1, seed_in=31, scale=1.0)),
    'cn_s':      (mint_controlnet, dict(size='s', L=100, Lc=20, t=979, seed_w=1234, seed_in=32, scale=0.7)),
    'vae_dec_tiny': (mint_vae_decoder, dict(cfg_name='VAE_TINY', L=40, seed_w=5, seed_in=41)),
    'vae_dec':      (mint_vae_decoder, dict(cfg_name='VAE_DEFAULT', L=25, seed_w=6, seed_in=42)),
    'vae_enc_tiny': (mint_vae_encoder, dict(cfg_name='VAE_TINY', T=8 * 48, seed_w=5, seed_in=43)),
    'vae_enc':      (mint_vae_encoder, dict(cfg_name='VAE_DEFAULT', T=480 * 20, seed_w=6, seed_in=44)),
    'smp_xs':    (mint_sampler, dict(size='xs', L=96, Lc=20, steps=50, seed_w=1, seed_in=21, guidance_scale=5.0, guidance_rescale=0.75, eta=1.0)),
    'smp_xs_e0': (mint_sampler, dict(size='xs', L=96, Lc=20, steps=20, seed_w=1, seed_in=22, guidance_scale=3.5, guidance_rescale=0.0, eta=0.0, with_gt=True)),
    'smp_s':     (mint_sampler, dict(size='s', L=500, Lc=100, steps=50, seed_w=1234, seed_in=21, guidance_scale=5.0, guidance_rescale=0.75, eta=1.0)),
    # BASELINE.json configs #2 / #3: the shipped L and XL models through the reference's unmodified 50-step inference() loop
    'smp_l':     (mint_sampler, dict(size='l', L=500, Lc=100, steps=50, seed_w=1234, seed_in=21, guidance_scale=5.0, guidance_rescale=0.75, eta=1.0)),
    'smp_xl':    (mint_sampler, dict(size='xl', L=500, Lc=100, steps=50, seed_w=1234, seed_in=21, guidance_scale=5.0, guidance_rescale=0.75, eta=1.0)),
    # config #4's per-GPU shape: 4 prompts = 8 denoiser rows (M = 4000 token rows) at XL width
    'xl_b8':     (mint_forward, dict(size='xl', L=500, Lc=100, timesteps=[499], seed_w=1234, seed_in=14, n_valid=(12, 1, 30, 1, 5, 1, 100, 1), B=8)),
    # config #5: XL width + the energy_l.yml controlnet section, 10 s latent; residual rows sampled every 25 tokens
    'cn_xl':     (mint_controlnet, dict(size='xl', L=500, Lc=100, t=[979, 499], seed_w=1234, seed_in=33, scale=1.0, row_stride=25)),
    # the ONE ControlNet configuration the reference ships: EzAudio-L + ckpts/controlnet/energy_l.yml, 10 s latent
    'cn_l':      (mint_controlnet, dict(size='l', L=500, Lc=100, t=[979, 499], seed_w=1234, seed_in=34, scale=1.0, row_stride=25)),
    # ... and its sampler: the reference's unmodified src/inference_controlnet.py::inference, 50 steps, guidance 3.5, no rescale, eta 1
    'smp_cn_l':  (mint_cn_sampler, dict(size='l', L=500, Lc=100, steps=50, seed_w=1234, seed_in=23, guidance_scale=3.5, guidance_rescale=0.0, eta=1.0, scale=1.0)),
    # BASELINE config #5 as benchmarked: XL width + the energy ControlNet through the reference's unmodified ControlNet sampler
    'smp_cn_xl': (mint_cn_sampler, dict(size='xl', L=500, Lc=100, steps=50, seed_w=1234, seed_in=25, guidance_scale=3.5, guidance_rescale=0.0, eta=1.0, scale=1.0)),
    # editing above toy size (conditioners.py:151-176, inference.py:79-86,103-104): forward with gt + mask and the full loop, L = 300
    'l_edit':    (mint_forward, dict(size='l', L=300, Lc=100, timesteps=[499], seed_w=1234, seed_in=12, with_gt=True)),
    'smp_l_edit': (mint_sampler, dict(size='l', L=300, Lc=100, steps=50, seed_w=1234, seed_in=24, guidance_scale=5.0, guidance_rescale=0.75, eta=1.0, with_gt=True)),
    'energy':    (mint_energy, dict()),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    ap.add_argument('--out', default='tests/golden')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    for name, (fn, kw) in JOBS.items():
        if a.only and name not in a.only:
            continue
        fn(name, out_dir=a.out, **kw)


if __name__ == '__main__':
    main()
