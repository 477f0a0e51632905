"""ControlNet branch: oracle vs the reference goldens on CPU, HIP path vs both on the GPU."""
import ast
import os

import numpy as np
import pytest

from oracle.controlnet import CN_DEFAULT, ControlNetOracle, conv1d, energy_curve, make_controlnet_state_dict
from oracle.dit import DiTOracle
from oracle.weights import make_inputs, make_state_dict, model_config, uniform_pm1  # noqa: F401
from tests.util import GOLDEN, record, rel_l2


def cn_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = ast.literal_eval(str(g['meta']))
    cfg = model_config(meta['size'])
    sd = make_state_dict(cfg, meta['seed_w'])
    csd = make_controlnet_state_dict(cfg, CN_DEFAULT, meta['seed_w'])
    inp = make_inputs(cfg, B=2, L=meta['L'], Lc=meta['Lc'], n_valid=(7, 1), seed=meta['seed_in'])
    cond = (0.5 + 0.5 * uniform_pm1('in.cond', 2 * 2 * meta['L'], meta['seed_in'])).reshape(2, 1, 2 * meta['L'])
    return cfg, sd, csd, inp, cond, g, meta


def _cn_keys(meta):
    """(timestep, key suffix) pairs and the row stride of the stored residuals (cn_xl keeps every 25th token row)."""
    ts = meta['t']
    pairs = [(t, f'_t{t}') for t in ts] if isinstance(ts, (list, tuple)) else [(ts, '')]
    return pairs, meta.get('row_stride', 1)


@pytest.mark.parametrize('name,only_t', [('cn_xs', None), ('cn_s', None), ('cn_xl', 499), ('cn_l', 499)])
def test_controlnet_oracle_matches_reference_golden(name, only_t):
    """cn_xl = BASELINE config #5's shape (XL width, energy_l.yml controlnet section, 10 s latent); cn_l = the one ControlNet configuration
    the reference ships (EzAudio-L + ckpts/controlnet/energy_l.yml); one of their two timesteps on CPU."""
    cfg, sd, csd, inp, cond, g, meta = cn_case(name)
    o = DiTOracle(cfg, sd)
    co = ControlNetOracle(cfg, csd)
    x257, _ = o.assemble_input(inp['x'])
    pairs, rs = _cn_keys(meta)
    for t, sfx in pairs:
        if only_t is not None and t != only_t:
            continue
        res = co.forward(x257, t, inp['ctx'], inp['ctx_mask'], cond, meta['scale'])
        assert len(res) == cfg['depth'] // 2
        for i, r in enumerate(res):
            assert rel_l2(r[:, ::rs], g[f'res{i}{sfx}']) < 1e-5
        pred = o.udit_forward(x257, t, inp['ctx'], inp['ctx_mask'], controlnet_skips=res)
        assert rel_l2(pred, g['pred' + sfx]) < 1e-5


def test_conv1d_and_energy_curve_against_torch():
    import torch
    import torch.nn.functional as F
    x = uniform_pm1('x', 2 * 5 * 40, 0).reshape(2, 5, 40)
    w = uniform_pm1('w', 7 * 5 * 3, 1).reshape(7, 5, 3)
    b = uniform_pm1('b', 7, 2)
    for stride in (1, 2):
        ref = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=1).numpy()
        np.testing.assert_allclose(conv1d(x, w, b, stride, 1), ref, rtol=1e-5, atol=1e-6)
    # EnergyExtractor restated with torch ops exactly as energy.py:19-56 writes them
    wav = 0.1 * uniform_pm1('wav', 24000, 5).reshape(1, 24000)
    a = torch.from_numpy(wav)
    pad = (1920 - 240) // 2
    sq = F.pad(a, (pad, pad), mode='reflect') ** 2
    en = F.unfold(sq[:, None, None, :], (1, 1920), stride=240)[:, :, :100].mean(dim=1)
    gdb = 10 * torch.log10(torch.maximum(en, torch.tensor(np.power(10, -60 / 10))))
    gdb = (gdb + 60) / (gdb.max(dim=-1, keepdim=True)[0] + 60 + 1e-8)
    np.testing.assert_allclose(energy_curve(wav)[..., 0], gdb.numpy(), rtol=1e-4, atol=1e-5)
    from ezaudio_amd.conditions import Conditioner
    c = Conditioner('energy', hop_size=240, window_size=1920, padding='reflect', min_db=-60, norm=True)(a, (1, 128, 50))
    assert c.shape == (1, 1, 100)
    np.testing.assert_allclose(c[:, 0].numpy(), gdb.numpy(), rtol=1e-4, atol=1e-5)


def test_energy_extractor_matches_the_reference_file_golden():
    """tests/golden/energy.npz was minted by loading /root/reference/src/models/conditions/energy.py BY FILE PATH (oracle/mint_golden.py::
    mint_energy; conditioner section of ckpts/controlnet/energy_l.yml): the product's Conditioner and the oracle's energy_curve against the
    reference's own outputs, incl. a silent stretch (the -60 dB floor) and a length that is not a multiple of the hop."""
    import torch
    from ezaudio_amd.conditions import Conditioner
    g = np.load(os.path.join(GOLDEN, 'energy.npz'))
    cond = Conditioner('energy', hop_size=240, window_size=1920, padding='reflect', min_db=-60, norm=True)
    for i in range(3):
        n, amp = int(g[f'n{i}'][0]), float(g[f'n{i}'][1])
        t = np.arange(n, dtype=np.float32)
        wav = (amp * uniform_pm1(f'energy.wav{i}', n, 7) * (0.55 + 0.45 * np.sin(2 * np.pi * t / 5000.0))).astype(np.float32).reshape(1, n)
        wav[:, n // 3:n // 3 + 2000] = 0.0
        ref = g[f'energy{i}']                                    # [1, frames, 1]
        frames = ref.shape[1]
        np.testing.assert_allclose(energy_curve(wav)[:, :frames], ref, rtol=2e-4, atol=2e-5)
        c = cond(torch.from_numpy(wav), (1, 128, frames // 2))     # [1, 1, 2 * latent frames]
        m = min(c.shape[-1], frames)
        np.testing.assert_allclose(c[0, 0, :m].numpy(), ref[0, :m, 0], rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('gname', ['smp_cn_l', 'smp_cn_xl'])
def test_controlnet_sampler_matches_the_reference_controlnet_loop_golden(lib, gname):
    """sampler_smp_cn_l: the reference's UNMODIFIED src/inference_controlnet.py::inference (50 steps, guidance 3.5, no rescale, eta 1) on the
    configuration it ships -- EzAudio-L + the energy ControlNet of ckpts/controlnet/energy_l.yml -- against the fused HIP sampler with the
    ControlNet attached (one captured graph per step: ControlNet on a side stream + backbone + CFG / DDIM).
    sampler_smp_cn_xl: the same loop at XL width = BASELINE config #5 as benchmarked (the reference ships no XL ControlNet: synthetic weights)."""
    import ast
    import torch
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    from tests.util import DIFF
    g = np.load(os.path.join(GOLDEN, f'sampler_{gname}.npz'))
    meta = ast.literal_eval(str(g['meta']))
    cfg = model_config(meta['size'])
    sd = make_state_dict(cfg, meta['seed_w'])
    csd = make_controlnet_state_dict(cfg, CN_DEFAULT, meta['seed_w'])
    m, cn = _models(cfg, sd, csd)
    L, Lc, steps = meta['L'], meta['Lc'], meta['steps']
    inp = make_inputs(cfg, B=2, L=L, Lc=Lc, seed=meta['seed_in'])
    C = cfg['out_chans']
    s3 = np.float32(np.sqrt(3.0))
    init = (uniform_pm1('smp.init', C * L, meta['seed_in']) * s3).reshape(1, C, L)
    noises = [(uniform_pm1(f'smp.z{i}', C * L, meta['seed_in']) * s3).reshape(1, C, L) for i in range(steps)]
    cond = (0.5 + 0.5 * uniform_pm1('smp.cond', 2 * L, meta['seed_in'])).reshape(1, 1, 2 * L).astype(np.float32)
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    smp.prepare(_t(inp['ctx'][0:1]), _t(inp['ctx_mask'][0:1]), _t(inp['ctx'][1:2]), _t(inp['ctx_mask'][1:2]), _t(init),
                torch.stack([_t(z) for z in noises], 0), meta['guidance_scale'], meta['guidance_rescale'], steps, meta['eta'],
                controlnet=cn, condition=_t(cond), conditioning_scale=meta['scale'])
    smp.run(steps)
    lat = smp.finish()
    torch.cuda.synchronize()
    r = rel_l2(lat.cpu().numpy(), g['latent'])
    record(f'{gname}: final-latent rel-L2 {r:.3e}')
    assert torch.isfinite(lat).all() and r < 2e-2


# ---------------------------------------------------------------------------------------------------
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _models(cfg, sd, csd):
    from ezaudio_amd import DiTControlNet, MaskDiT
    m = MaskDiT(device='cuda:0', **cfg)
    m.load_state_dict(sd)
    ccfg = dict(cfg)
    ccfg.update(CN_DEFAULT)
    cn = DiTControlNet(device='cuda:0', **ccfg)
    cn.load_state_dict(csd)
    return m, cn


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['cn_xs', 'cn_s', 'cn_xl', 'cn_l'])
def test_controlnet_hip_matches_reference_golden(lib, name):
    """cn_xl: BASELINE config #5 (XL width + the energy ControlNet, L = 500) at t in {979, 499} against the reference's own
    DiTControlNet residuals and the backbone prediction that consumes them."""
    import torch
    cfg, sd, csd, inp, cond, g, meta = cn_case(name)
    m, cn = _models(cfg, sd, csd)
    pairs, rs = _cn_keys(meta)
    for tt, sfx in pairs:
        t = torch.tensor(tt)
        x257, _ = m(_t(inp['x']), t, None, forward_model=False)               # src/inference_controlnet.py:89-91
        skips = cn(x257, t, _t(inp['ctx']), context_mask=_t(inp['ctx_mask']), cls_token=None, condition=_t(cond),
                   conditioning_scale=meta['scale'])
        assert len(skips) == cfg['depth'] // 2
        for i, s in enumerate(skips):
            r = rel_l2(s.cpu().numpy()[:, ::rs], g[f'res{i}{sfx}'])
            record(f'{name} t={tt} residual {i}: rel-L2 {r:.3e}')
            assert r < 2e-2
        pred = m.model(x257, t, _t(inp['ctx']), context_mask=_t(inp['ctx_mask']), cls_token=None, controlnet_skips=skips)
        ref = g['pred' + sfx]
        r = rel_l2(pred.cpu().numpy(), ref)
        record(f'{name} t={tt} backbone prediction with ControlNet skips: rel-L2 {r:.3e}')
        assert r < 2e-2 and float(np.abs(pred.cpu().numpy() - ref).max()) < 0.15 * max(1.0, float(ref.std()) / 1.48)


@pytest.mark.gpu
@pytest.mark.parametrize('zfuse', [0, 1])
def test_controlnet_fused_sampler_equals_stepwise_calls(lib, zfuse):
    """The device loop with an attached ControlNet (one hipGraph per step: ControlNet + backbone + CFG/DDIM) against the
    same step assembled from the public call surfaces, driven by the oracle's restatement of the reference loop.
    zfuse: both settings of the LayerNorm-algebra path -- its consumers must index their G' / C' tables with the BACKBONE's step
    counter inside the attached ControlNet too (round-3 ADVICE: they read the ControlNet's own, never-advanced counter)."""
    import torch
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    from oracle.sampler import sample as oracle_sample
    from tests.util import DIFF
    cfg, sd, csd, inp, cond, g, meta = cn_case('cn_xs')
    m, cn = _models(cfg, sd, csd)
    for hdl in (m._h, cn._h):
        assert lib.ezdit_set_option(hdl, b'zfuse', zfuse) == 0
    C, L, steps, scale = cfg['out_chans'], meta['L'], 6, 0.8
    s3 = np.float32(np.sqrt(3.0))
    init = (uniform_pm1('c.init', C * L, 1) * s3).reshape(1, C, L)
    noises = [(uniform_pm1(f'c.z{i}', C * L, 1) * s3).reshape(1, C, L) for i in range(50)]
    cond1 = cond[0:1]

    def denoise(x, t, ctx, msk, gt, gm, mm=None, cc=None):
        mm, cc = mm or m, cc or cn
        tt = torch.tensor(t)
        x257, _ = mm(_t(x), tt, None, forward_model=False)
        sk = cc(x257, tt, _t(ctx), context_mask=_t(msk), cls_token=None, condition=_t(np.concatenate([cond1, cond1], 0)),
                conditioning_scale=scale)
        return mm.model(x257, tt, _t(ctx), context_mask=_t(msk), cls_token=None, controlnet_skips=sk).cpu().numpy()
    tr = []
    oracle_sample(denoise, inp['ctx'][0:1], inp['ctx_mask'][0:1], inp['ctx'][1:2], inp['ctx_mask'][1:2], init, noises,
                  guidance_scale=3.5, guidance_rescale=0.0, ddim_steps=50, eta=1.0, diff_params=DIFF, trace=tr)
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    smp.prepare(_t(inp['ctx'][0:1]), _t(inp['ctx_mask'][0:1]), _t(inp['ctx'][1:2]), _t(inp['ctx_mask'][1:2]), _t(init),
                torch.stack([_t(z) for z in noises], 0), 3.5, 0.0, 50, 1.0, controlnet=cn, condition=_t(cond1),
                conditioning_scale=scale)
    smp.run(steps)
    lat = smp.finish()
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all()
    assert rel_l2(lat.cpu().numpy(), tr[steps - 1]) < 5e-3
    # ... and the rest of the 50 steps: the ControlNet must see the CURRENT timestep's modulation at every step (the reference passes t
    # into it each step, src/inference_controlnet.py:92-96); a ControlNet pinned to the first timestep's slot drifts far beyond this gate
    smp.run(50 - steps)
    lat = smp.finish().clone()
    torch.cuda.synchronize()
    assert rel_l2(lat.cpu().numpy(), tr[49]) < 2e-2
    # option cn_overlap = 0: the ControlNet chain on the sampler's own stream instead of the side stream -- same kernels, same order per chain: bitwise
    assert lib.ezdit_set_option(m._h, b'cn_overlap', 0) == 0
    try:
        smp.prepare(_t(inp['ctx'][0:1]), _t(inp['ctx_mask'][0:1]), _t(inp['ctx'][1:2]), _t(inp['ctx_mask'][1:2]), _t(init),
                    torch.stack([_t(z) for z in noises], 0), 3.5, 0.0, 50, 1.0, controlnet=cn, condition=_t(cond1),
                    conditioning_scale=scale)
        smp.run(50)
        serial = smp.finish().clone()
        torch.cuda.synchronize()
    finally:
        assert lib.ezdit_set_option(m._h, b'cn_overlap', 1) == 0
    assert torch.equal(serial, lat)
    # the fused run left conditioning_scale 0.8 attached to the backbone: the drop-in call surface (residuals already scaled by
    # DiTControlNet.forward, controlnet.py:313) must not apply it a second time
    x_probe = np.concatenate([init, init], 0)
    again = denoise(x_probe, 499, inp['ctx'], inp['ctx_mask'], None, None)
    m2, cn2 = _models(cfg, sd, csd)   # a pair that never saw a fused run
    for hdl in (m2._h, cn2._h):
        assert lib.ezdit_set_option(hdl, b'zfuse', zfuse) == 0
    fresh = denoise(x_probe, 499, inp['ctx'], inp['ctx_mask'], None, None, m2, cn2)
    np.testing.assert_array_equal(again, fresh)
    # and the ControlNet really matters: detaching it changes the trajectory
    smp2 = LatentSampler(m, DDIMScheduler(**DIFF))
    smp2.prepare(_t(inp['ctx'][0:1]), _t(inp['ctx_mask'][0:1]), _t(inp['ctx'][1:2]), _t(inp['ctx_mask'][1:2]), _t(init),
                 torch.stack([_t(z) for z in noises], 0), 3.5, 0.0, 50, 1.0)
    smp2.run(steps)
    lat2 = smp2.finish()
    torch.cuda.synchronize()
    assert rel_l2(lat2.cpu().numpy(), tr[steps - 1]) > 1e-2
