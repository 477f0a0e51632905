"""CPU tests: the numpy oracle against the golden vectors minted from the reference itself
(oracle/mint_golden.py), and against the DDIM anchors recorded in SURVEY.md section 8a row S."""
import numpy as np
import pytest

from oracle.ddim import DDIMOracle, make_alphas_cumprod, trailing_timesteps
from oracle.dit import DiTOracle, flops_per_step
from oracle.sampler import rescale_noise_cfg, sample
from oracle.weights import model_config, param_shapes, uniform_pm1
from tests.util import DIFF, golden_case, rel_l2, sampler_case


@pytest.mark.parametrize('name', ['xs', 'xs64', 'xs_edit', 'xs_cn', 's64', 's_edit'])
def test_dit_oracle_matches_reference_golden(name):
    cfg, sd, inp, kw, g, meta = golden_case(name)
    o = DiTOracle(cfg, sd, np.float32)
    for t in meta['timesteps']:
        pred, _ = o.forward(inp['x'], t, inp['ctx'], inp['ctx_mask'], **kw)
        ref = g[f'pred_t{t}']
        # both sides are fp32; the reference's own fp32-vs-fp64 gap is 7.5e-7 rel-L2
        assert rel_l2(pred, ref) < 1e-5, (name, t)
        assert np.abs(pred - ref).max() < 1e-4


def test_dit_oracle_config1_s_all_timesteps():
    """BASELINE config #1 (EzAudio-S single denoise step, t in {999, 979, 499, 19}) -- the plumbing gate."""
    cfg, sd, inp, kw, g, meta = golden_case('s')
    o = DiTOracle(cfg, sd, np.float32)
    for t in (999, 979, 499, 19):
        pred, mae_mask = o.forward(inp['x'], t, inp['ctx'], inp['ctx_mask'])
        assert rel_l2(pred, g[f'pred_t{t}']) < 1e-5
        assert mae_mask.shape == inp['x'].shape and (mae_mask == 1).all()


def test_oracle_fp64_close_to_fp32():
    cfg, sd, inp, kw, g, meta = golden_case('xs')
    p32, _ = DiTOracle(cfg, sd, np.float32).forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    p64, _ = DiTOracle(cfg, sd, np.float64).forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    assert rel_l2(p32, p64) < 1e-5


def test_timestep_is_broadcast_or_per_row():
    cfg, sd, inp, kw, g, meta = golden_case('xs')
    o = DiTOracle(cfg, sd)
    a, _ = o.forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    b, _ = o.forward(inp['x'], np.array([499, 499]), inp['ctx'], inp['ctx_mask'])
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('name', ['smp_xs', 'smp_xs_e0'])
def test_sampler_oracle_matches_reference_loop(name):
    """The reference's unmodified inference() (driven with fakes) vs the numpy loop restatement."""
    cfg, sd, inp, init, noises, g, meta = sampler_case(name)
    o = DiTOracle(cfg, sd)

    def denoise(x, t, ctx, msk, gt, gm):
        return o.forward(x, t, ctx, msk, gt=gt, mae_mask_infer=gm)[0]
    gt = inp['gt'][0:1] if meta['with_gt'] else None
    gm = inp['gt_mask'][0:1] if meta['with_gt'] else None
    lat = sample(denoise, inp['ctx'][0:1], inp['ctx_mask'][0:1], inp['ctx'][1:2], inp['ctx_mask'][1:2], init, noises,
                 guidance_scale=meta['guidance_scale'], guidance_rescale=meta['guidance_rescale'],
                 ddim_steps=meta['steps'], eta=meta['eta'], gt=gt, gt_mask=gm, diff_params=DIFF)
    assert rel_l2(lat, g['latent']) < 2e-4  # 20-50 chained fp32 forwards


# ---- DDIM restatement: PARITY UNPINNED by the reference (diffusers absent); anchors from SURVEY.md --------
def test_ddim_anchors():
    a = make_alphas_cumprod()
    assert a.dtype == np.float32 and a.shape == (1000,)
    np.testing.assert_allclose(a[0], 0.99915, rtol=2e-6)
    np.testing.assert_allclose(a[19], 0.981010, rtol=2e-6)
    np.testing.assert_allclose(a[499], 0.242359, rtol=2e-6)
    np.testing.assert_allclose(a[979], 8.5788e-5, rtol=2e-5)
    assert a[999] == 0.0  # zero terminal SNR, exactly
    assert (np.diff(a) < 0).all()


def test_ddim_timesteps_trailing():
    np.testing.assert_array_equal(trailing_timesteps(50), np.arange(999, 0, -20))
    np.testing.assert_array_equal(trailing_timesteps(100), np.arange(999, 0, -10))
    assert trailing_timesteps(50)[-1] == 19 and trailing_timesteps(100)[-1] == 9


def test_ddim_step_invariants():
    o = DDIMOracle(**DIFF)
    o.set_timesteps(50)
    c_last = o.coefficients(19, 1.0)
    assert c_last['sigma'] == 0.0 and c_last['c_x0'] == 1.0 and c_last['c_dir'] == 0.0  # alpha_prev = 1
    c_first = o.coefficients(999, 1.0)
    assert c_first['sa'] == 0.0 and c_first['sb'] == 1.0      # alpha_bar[999] == 0
    assert 0 < c_first['c_dir'] < 1e-3                          # (1 - a_prev - sigma^2) ~ 6e-8, SURVEY section 7 item 5
    o.set_timesteps(25)
    assert np.isnan(o.coefficients(999, 1.0)['c_dir'])          # the reference NaNs for 25 steps too
    # eta = 0: deterministic DDIM, x_prev reproduces x0 direction
    o.set_timesteps(50)
    x = uniform_pm1('x', 64, 0).reshape(1, 8, 8)
    v = uniform_pm1('v', 64, 0).reshape(1, 8, 8)
    out = o.step(v, 499, x, 0.0)
    c = o.coefficients(499, 0.0)
    x0 = c['sa'] * x - c['sb'] * v
    eps = c['sa'] * v + c['sb'] * x
    np.testing.assert_allclose(out, c['c_x0'] * x0 + np.sqrt(1 - c['c_x0'] ** 2) * eps, rtol=1e-5, atol=1e-6)


def test_rescale_noise_cfg_unbiased_std():
    c = uniform_pm1('c', 2 * 3 * 50, 1).reshape(2, 3, 50)
    g = 3.0 * uniform_pm1('g', 2 * 3 * 50, 2).reshape(2, 3, 50)
    out = rescale_noise_cfg(g, c, 1.0)
    np.testing.assert_allclose(out.reshape(2, -1).std(axis=1, ddof=1), c.reshape(2, -1).std(axis=1, ddof=1), rtol=1e-5)
    np.testing.assert_allclose(rescale_noise_cfg(g, c, 0.0), g, rtol=1e-6)


def test_flops_formula_matches_survey():
    xl, l = model_config('xl'), model_config('l')
    assert abs(flops_per_step(xl, 2, 500, 100) / 1e12 - 1.541) < 2e-3
    assert abs(flops_per_step(xl, 2, 500, 100, hoisted=False) / 1e12 - 1.5735) < 2e-3
    assert abs(flops_per_step(l, 2, 500, 100) / 1e12 - 1.057) < 2e-3
    n = sum(int(np.prod(s)) for k, (s, kind) in param_shapes(xl).items() if kind != 'inv_freq')
    assert abs(n / 1e6 - 874.76) < 0.01  # 874.76 M parameters


def test_single_key_cross_attention_is_the_constant_the_shortcut_adds():
    """The identity the product's `xkey1` shortcut rests on (csrc/api.hip), checked on the oracle that is pinned to the reference:
    when a batch element's context mask has ONE valid key, the cross-attention module (attention.py:122-149 -- to_q / to_k / to_v, per-head LayerNorm, masked
    softmax, proj) returns proj(v_key) for EVERY query row, whatever the queries are and wherever the key sits; with more valid keys it does not."""
    cfg = model_config('xs')
    from oracle.weights import make_inputs, make_state_dict
    sd = make_state_dict(cfg, 3)
    o = DiTOracle(cfg, sd, np.float64)
    D, L, Lc = cfg['embed_dim'], 40, 20
    inp = make_inputs(cfg, B=3, L=L, Lc=Lc, n_valid=(1, 1, 6), seed=5)
    mask = inp['ctx_mask'].copy()
    mask[1] = np.roll(mask[1], 7)                      # the single key of element 1 is key 7
    ctx = o.context_embed(inp['ctx'].astype(np.float64))
    x = uniform_pm1('q.x', 3 * L * D, 9).reshape(3, L, D).astype(np.float64) * 1.7
    pfx = 'model.in_blocks.0.cross_attn'
    out = o.attention(pfx, x, context=ctx, key_mask=mask)
    wv, wo, bo = o.p(f'{pfx}.to_v.weight'), o.p(f'{pfx}.proj.weight'), o.p(f'{pfx}.proj.bias')
    for b, key in ((0, 0), (1, 7)):
        d = (ctx[b, key] @ wv.T) @ wo.T + bo           # W_o v_key + b_o
        np.testing.assert_allclose(out[b], np.broadcast_to(d, (L, D)), rtol=0, atol=1e-12)
    assert np.abs(out[2] - out[2][0:1]).max() > 1e-3   # six valid keys: the rows differ


def test_skip_connection_layernorm_is_the_algebra_the_skip_path_runs():
    """The identity the product's `skip_z` path rests on (csrc/gemm_ks.h forms COPY2 / ZIN, DESIGN.md "The skip path"), checked on the oracle that is pinned to the reference
    (blocks.py:124-128: x = skip_linear(skip_norm(cat[x, skip]))): with the per-tile partial (sum, sum of squares) of the two HALVES -- the skip's produced many blocks earlier --
    the row statistics of the concatenation are plain sums, and  r (([x | skip] * g) W^T - mu G') + C'  with G' = g W^T, C' = c W^T + b  is the reference's result."""
    cfg = model_config('xs')
    from oracle.dit import layer_norm, linear, LN_EPS
    from oracle.weights import make_state_dict
    sd = make_state_dict(cfg, 4)
    o = DiTOracle(cfg, sd, np.float64)
    D, M, cw = cfg['embed_dim'], 37, 96
    pfx = 'model.out_blocks.0'
    g, c = o.p(f'{pfx}.skip_norm.weight'), o.p(f'{pfx}.skip_norm.bias')
    W, b = o.p(f'{pfx}.skip_linear.weight'), o.p(f'{pfx}.skip_linear.bias')
    x = uniform_pm1('skipz.x', M * D, 11).reshape(M, D).astype(np.float64) * 1.3 + 0.4
    skip = uniform_pm1('skipz.s', M * D, 12).reshape(M, D).astype(np.float64) * 0.6 - 0.2
    want = linear(layer_norm(np.concatenate([x, skip], -1), g, c), W, b)

    def parts(h):       # what a producer stores: (sum, sum of squares) over each cw-column tile of its D columns
        return [(h[:, i:i + cw].sum(1), (h[:, i:i + cw] ** 2).sum(1)) for i in range(0, D, cw)]
    s1 = sum(p[0] for p in parts(x)) + sum(p[0] for p in parts(skip))
    s2 = sum(p[1] for p in parts(x)) + sum(p[1] for p in parts(skip))
    mu = s1 / (2 * D)
    r = 1.0 / np.sqrt(s2 / (2 * D) - mu * mu + LN_EPS)
    A = np.concatenate([x * g[:D], skip * g[D:]], -1)            # the two halves of the operand, each written by its own producer
    Gp, Cp = g @ W.T, c @ W.T + b
    got = r[:, None] * (A @ W.T - mu[:, None] * Gp) + Cp
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
