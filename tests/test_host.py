"""CPU tests of the host side: C-ABI surface, parameter layout / packer, config surface, scheduler.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle.ddim import DDIMOracle
from oracle.weights import make_state_dict, model_config
from tests.util import DIFF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'ezdit.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ez(?:dit|vae)_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_symbol_the_header_declares(lib):
    from ezaudio_amd import _lib
    declared = _header_functions()
    assert len(declared) >= 20
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (ez(?:dit|vae)_[a-z_0-9]+)', out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(_lib.PROTOTYPES), sorted(set(declared) ^ set(_lib.PROTOTYPES))
    assert lib.ezdit_abi_version() == 4


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle."""
    for dp, _, files in os.walk(os.path.join(ROOT, 'ezaudio_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+\.*oracle', txt, flags=re.M), f
                # the reference tree is cited in comments / docstrings only: nothing opens, imports or path-joins it at run time
                assert not re.search(r'''(open|sys\.path\S*|import_module|spec_from_file_location)\([^)]*/root/reference''', txt), f


def _handle(lib, size):
    from ezaudio_amd import _lib
    cfg = model_config(size)
    c = _lib.EzditConfig(cfg['embed_dim'], cfg['num_heads'], cfg['depth'], cfg['in_chans'], cfg['out_chans'],
                         cfg['context_dim'], cfg['ada_sola_rank'], float(cfg['ada_sola_alpha']), float(cfg['mlp_ratio']), 2048)
    h = C.c_void_p()
    assert lib.ezdit_create(C.byref(c), C.byref(h)) == 0
    return cfg, h


@pytest.mark.parametrize('size', ['xs', 'xs64', 's'])
def test_param_layout_covers_state_dict_exactly_once(lib, size):
    from ezaudio_amd.weights import param_table
    cfg, h = _handle(lib, size)
    sd = make_state_dict(cfg, 3)
    table = param_table(h)
    used = [k for p in table for k in p['src']]
    assert len(used) == len(set(used))
    assert set(used) == {k for k in sd if not k.endswith('rotary.inv_freq')}
    # offsets: 256-byte aligned, non-overlapping, inside the blob
    spans = sorted((p['offset'], p['offset'] + p['rows_pad'] * p['ld'] * (2 if p['dtype'] == 1 else 4)) for p in table)
    assert all(o % 256 == 0 for o, _ in spans)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    assert spans[-1][1] <= lib.ezdit_param_bytes(h)
    lib.ezdit_destroy(h)


@pytest.mark.parametrize('size', ['xs', 'xs64'])
def test_pack_roundtrip_and_geglu_interleave(lib, size):
    from ezaudio_amd.weights import pack_state_dict, param_table
    cfg, h = _handle(lib, size)
    sd = make_state_dict(cfg, 5)
    blob = pack_state_dict(h, sd)
    D, inner = cfg['embed_dim'], 4 * cfg['embed_dim']
    for p in param_table(h):
        n = p['rows_pad'] * p['ld']
        dt = torch.bfloat16 if p['dtype'] == 1 else torch.float32
        raw = blob[p['offset']:p['offset'] + n * (2 if p['dtype'] == 1 else 4)].view(dt).reshape(p['rows_pad'], p['ld']).float().numpy()
        assert (raw[p['rows']:] == 0).all() and (raw[:, p['cols']:] == 0).all()   # zero padding
        got = raw[:p['rows'], :p['cols']]
        want = np.concatenate([np.asarray(sd[k]).reshape(1, -1) if p['rows'] == 1 else np.asarray(sd[k]).reshape(sd[k].shape[0], -1)
                               for k in p['src']], axis=1 if p['rows'] == 1 else 0)
        if p['transform'] == 1:  # GEGLU8: row 16g+i (i<8) = value row 8g+i ; row 16g+8+i = gate row inner+8g+i
            w2 = want.reshape(-1, 1) if p['rows'] == 1 else want
            perm = np.concatenate([np.concatenate([np.arange(8 * g, 8 * g + 8), inner + np.arange(8 * g, 8 * g + 8)])
                                   for g in range(inner // 8)])
            w2 = w2[perm]
            want = w2.reshape(1, -1) if p['rows'] == 1 else w2
        if p['transform'] == 2:  # QKROPE (include/ezdit.h): restated here from the header's words, not from the packer
            dh = D // cfg['num_heads']
            perm = np.arange(3 * D)
            for part in range(2):
                for p0 in range(0, D, 2 * dh):
                    for c in range(2 * dh):
                        j, g, e, sft = c // 16, (c // 4) % 4, (c // 2) % 2, c % 2
                        full = dh // 16
                        if dh % 16 == 0:
                            hh, f = j // full, 8 * (j % full) + 2 * g + e
                        elif j < full:
                            hh, f = 0, 8 * j + 2 * g + e
                        elif j == full:
                            hh, f = g // 2, 32 + 2 * (g % 2) + e
                        else:
                            hh, f = 1, 8 * (j - full - 1) + 2 * g + e
                        perm[part * D + p0 + c] = part * D + p0 + hh * dh + f + (dh // 2) * sft
            assert sorted(perm) == list(range(3 * D))
            want = want[perm]
            # what the permutation is for: every aligned row pair (2 k, 2 k + 1) of q and k is a rotate-half pair (channel, channel + dh / 2) of ONE head
            qk = perm[:2 * D].reshape(-1, 2)
            assert ((qk[:, 1] - qk[:, 0]) == dh // 2).all() and ((qk[:, 0] % D) // dh == (qk[:, 1] % D) // dh).all()
        tol = 0 if p['dtype'] == 0 else 2 ** -8
        np.testing.assert_allclose(got, want, rtol=tol, atol=1e-30, err_msg=p['name'])
    with pytest.raises(KeyError):
        bad = dict(sd); bad.pop('mask_embed'); pack_state_dict(h, bad)
    with pytest.raises(KeyError):
        bad = dict(sd); bad['model.extra'] = sd['mask_embed']; pack_state_dict(h, bad)
    lib.ezdit_destroy(h)


def test_unsupported_configs_raise_like_the_reference(lib):
    from ezaudio_amd import MaskDiT
    cfg = model_config('xs')
    for key, val in (('time_fusion', 'token'), ('context_fusion', 'concat'), ('norm_layer', 'rmsnorm'),
                     ('rope_mode', 'dual'), ('act_layer', 'gelu')):
        bad = dict(cfg); bad[key] = val
        with pytest.raises(NotImplementedError):
            MaskDiT(device='cpu', **bad)
    bad = dict(cfg); bad['num_heads'] = 3  # head_dim 48
    with pytest.raises(NotImplementedError):
        MaskDiT(device='cpu', **bad)
    m = MaskDiT(device='cpu', **cfg)
    # call order errors surface as exceptions, not crashes
    from ezaudio_amd._lib import EzditError
    with pytest.raises(EzditError):
        m.bind(2, 96, 20, 1)


def test_workspace_size_scales(lib):
    cfg, h = _handle(lib, 'xl')
    a = lib.ezdit_workspace_bytes(h, 2, 500, 100, 50)
    b = lib.ezdit_workspace_bytes(h, 8, 500, 100, 50)
    assert 100e6 < a < 1e9 and a < b < 4 * a + 200e6
    assert lib.ezdit_param_bytes(h) < 2.2e9  # bf16 matrices + fp32 time path, XL
    lib.ezdit_destroy(h)


def test_yaml_surface_matches_reference_key_set():
    from ezaudio_amd.config import configs, load_yaml_with_includes, validate_model_config
    for name, embed, depth, cdim in (('s3_xl', 1152, 28, 2048), ('s3_l', 1024, 24, 1024)):
        p = load_yaml_with_includes(configs[name]['config'])
        assert set(p) >= {'model', 'autoencoder', 'text_encoder', 'diff'}
        m = validate_model_config(p['model'])
        assert (m['embed_dim'], m['depth'], m['context_dim']) == (embed, depth, cdim)
        assert {k: m[k] for k in model_config('xl') if k not in ('embed_dim', 'depth', 'num_heads', 'ada_sola_rank',
                                                                  'ada_sola_alpha', 'context_dim')} == \
               {k: v for k, v in model_config('xl').items() if k not in ('embed_dim', 'depth', 'num_heads', 'ada_sola_rank',
                                                                          'ada_sola_alpha', 'context_dim')}
        assert p['autoencoder']['latent_sr'] == 50 and p['text_encoder']['max_length'] == 100
    assert model_config('xl')['embed_dim'] == 1152 and model_config('l')['ada_sola_rank'] == 32


def test_yaml_include_tag(tmp_path):
    from ezaudio_amd.config import load_yaml_with_includes
    (tmp_path / 'inner.yml').write_text('a: 1\nb: [2, 3]\n')
    (tmp_path / 'outer.yml').write_text('x: !include inner.yml\ny: 5\n')
    assert load_yaml_with_includes(str(tmp_path / 'outer.yml')) == {'x': {'a': 1, 'b': [2, 3]}, 'y': 5}


def test_product_scheduler_agrees_with_oracle_restatement():
    from ezaudio_amd.scheduler import DDIMScheduler
    s = DDIMScheduler(**DIFF)
    o = DDIMOracle(**DIFF)
    # torch.linspace and the numpy restatement differ by <= 1 ulp in beta -> ~4e-7 in alpha_bar
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), o.alphas_cumprod, rtol=0, atol=1e-6)
    assert float(s.alphas_cumprod[999]) == 0.0
    for n, eta in ((50, 1.0), (100, 1.0), (50, 0.0)):
        s.set_timesteps(n); o.set_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), o.timesteps)
        for (sa, sb, cx0, cdir, sig), t in zip(s.ddim_coefficients(eta), o.timesteps):
            c = o.coefficients(t, eta)
            np.testing.assert_allclose([sa, sb, cx0, sig], [c['sa'], c['sb'], c['c_x0'], c['sigma']], rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(cdir, c['c_dir'], rtol=0, atol=3e-4)  # sqrt of a ~6e-8 cancellation at t=999
    # step() against the oracle on random tensors
    s.set_timesteps(50); o.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, v, z = (torch.randn(1, 8, 16, generator=g) for _ in range(3))
    for t in (979, 499, 19):
        a = s.step(v, torch.tensor(t), x, eta=1.0, variance_noise=z).prev_sample.numpy()
        b = o.step(v.numpy(), t, x.numpy(), 1.0, z.numpy())
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)
    with pytest.raises(NotImplementedError):
        DDIMScheduler(**dict(DIFF, prediction_type='epsilon'))


OPTION_NAMES = ['zfuse', 'xkey1', 'skip_z', 'geglu_co', 'qkv_co', 'wt', 'gemm_pp', 'tile_partial', 'attn_xcd', 'row_variant', 'gemm_panel', 'row_affine', 'epi_lds', 'attn_xk2',
                'attn_nkh', 'attn_qtile', 'cn_overlap', 'fuse_q2', 'q2_pp', 'stamp_launch', 'trace_launches']


def test_tuning_knobs_named_in_the_header_exist(lib):
    """The option list is closed: every name the library accepts is documented in the header and accepted here, names retired in round 5
    and unknown names are refused with EZDIT_E_INVALID, and the parser in csrc/api.hip knows exactly this list (<= 25 names)."""
    import re
    _, h = _handle(lib, 'xs')
    src = open(os.path.join(ROOT, 'include', 'ezdit.h')).read()
    for n in OPTION_NAMES:
        assert re.search(r'\b%s\b' % n, src), n
        assert lib.ezdit_set_option(h, n.encode(), 0) == 0, n
    for n in ('no_such_knob', 'qkv_affine', 'prefetch', 'geglu_tile', 'ztile', 'zfake', 'gemm_debug', 'fuse_resid', 'tile_p18', 'pp_max_m'):
        assert lib.ezdit_set_option(h, n.encode(), 1) == -1, n
        assert n.encode() in lib.ezdit_last_error()
    api = open(os.path.join(ROOT, 'ezaudio_amd', 'csrc', 'api.hip')).read()
    body = api[api.index('int ezdit_set_option('):]
    parsed = re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', body)
    assert sorted(set(parsed) - {'zfake'}) == sorted(OPTION_NAMES), parsed   # zfake: EZ_DIAG builds only
    assert len(OPTION_NAMES) <= 25
    lib.ezdit_destroy(h)


def test_bench_relaunches_itself_for_multi_gpu_and_integration_stub_matches_header():
    """`python bench.py --gpus N` (the driver's form) must start N ranks by itself; the ctypes struct INTEGRATION.md shows a
    maintainer must have the header's field list (a short struct makes ezdit_create read stack garbage)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.relaunch_command(4, ['--gpus', '4', '--steps', '7'], port=12345)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '127.0.0.1' in cmd and '12345' in cmd
    assert cmd[-4:] == ['--gpus', '4', '--steps', '7'] and cmd[-5].endswith('bench.py')
    hdr = open(os.path.join(ROOT, 'include', 'ezdit.h')).read()
    body = re.search(r'typedef struct \{(.*?)\} ezdit_config;', hdr, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = re.findall(r'(?:int32_t|float)\s+([a-z_0-9]+);', body)
    from ezaudio_amd import _lib
    assert fields == [f[0] for f in _lib.EzditConfig._fields_]
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = re.search(r'class EzditConfig.*?_fields_ = \[(.*?)\]\n', doc, flags=re.S)
    assert stub, 'INTEGRATION.md must show the ctypes struct'
    assert re.findall(r"\('([a-z_0-9]+)'", stub.group(1)) == fields


def test_torch_cpu_baseline_port_matches_reference_golden():
    """oracle/torch_ref.py (what bench.py times as the CPU baseline on the GPU box) against the reference-minted goldens."""
    from oracle.torch_ref import DiTTorchRef
    from tests.util import golden_case, rel_l2
    for name in ('xs', 'xs64', 'xs_edit'):
        cfg, sd, inp, kw, g, meta = golden_case(name)
        m = DiTTorchRef(cfg, sd)
        for t in meta['timesteps']:
            kk = dict(gt=kw['gt'], mae_mask_infer=kw['mae_mask_infer']) if 'gt' in kw else {}
            pred, _ = m.forward(inp['x'], t, inp['ctx'], inp['ctx_mask'], **kk)
            assert rel_l2(pred.numpy(), g[f'pred_t{t}']) < 1e-5, (name, t)


def test_source_hash_tracks_kernel_sources():
    from ezaudio_amd.build import source_hash
    h = source_hash()
    assert len(h) == 16 and h == source_hash()


def test_ping_pong_schedules_are_hazard_free_model():
    """Executable restatement of the two schedules of k_gemm_pp (csrc/gemm_pp.h): two wave groups one barrier interval apart.  The model
    replays the per-group programs against one barrier counter (an event's time = barriers the group has executed before it) and
    checks, for every ring depth and K-tile count: both groups execute the same number of barriers; a tile is read only after every wave
    that issued pieces of it has waited for them BEFORE an earlier barrier (RAW); the counted wait allows exactly the pieces of the
    group's younger tiles to stay in flight; a ring slot is refilled only behind a barrier that follows its last read (WAR)."""
    for NS in (3, 4, 5, 6):
        PD = NS - 1
        for nt in range(1, 14):
            # ---------------- SCHED 1: both groups issue a share of every tile and read every tile
            ev, nb = [], []
            for G in (0, 1):
                k = 0
                for t in range(min(PD, nt)):
                    ev.append((k, G, 'issue', t, None))
                ev.append((k, G, 'wait', 0, min(nt, PD) - 1))
                k += 1
                if G == 1:
                    k += 1
                for t in range(nt):
                    rf = t + PD < nt
                    ev.append((k, G, 'read', t, None))
                    if rf:
                        ev.append((k, G, 'issue', t + PD, None))
                    wait = (t + 1, PD - 1) if rf else ((t + 1, nt - 2 - t) if t + 1 < nt else None)
                    if G == 1 and wait:
                        ev.append((k, G, 'wait') + wait)
                    k += 1
                    if G == 0:
                        if wait:
                            ev.append((k, G, 'wait') + wait)
                        k += 1
                    elif rf or t + 1 < nt:
                        k += 1
                nb.append(k)
            assert nb[0] == nb[1] == 2 * nt + 1, (NS, nt, nb)
            for t in range(nt):
                first_read = min(tm for tm, g, kind, tt, _ in ev if kind == 'read' and tt == t)
                for G in (0, 1):
                    w = [(tm, y) for tm, g, kind, tt, y in ev if kind == 'wait' and tt == t and g == G]
                    assert len(w) == 1, (NS, nt, t, G)
                    tm_w, younger = w[0]
                    assert tm_w < first_read, ('RAW', NS, nt, t, G)
                    issued = [tt for tm, g, kind, tt, _ in ev if kind == 'issue' and g == G and tm <= tm_w]
                    assert t in issued and younger == len([u for u in issued if u > t]), ('count', NS, nt, t, G)
                if t >= NS:
                    last_read = max(tm for tm, g, kind, tt, _ in ev if kind == 'read' and tt == t - NS)
                    first_issue = min(tm for tm, g, kind, tt, _ in ev if kind == 'issue' and tt == t)
                    assert last_read < first_issue, ('WAR', NS, nt, t)
            # ---------------- SCHED 2: tile u is issued by group (u + PD) & 1 and read by group u & 1
            def own_younger(u):
                last = min(u - 1 + PD, nt - 1)
                return (last - u) // 2 if last >= u + 2 else 0
            ev, nb = [], []
            for G in (0, 1):
                k = 0
                for u in range(PD):
                    if ((u + PD) & 1) == G and u < nt:
                        ev.append((k, G, 'issue', u, None))
                if (PD & 1) == G:
                    last = min(PD - 1, nt - 1)
                    ev.append((k, G, 'wait', 0, last // 2 if last >= 2 else 0))
                k += 1

                def end_wait(i, rf, k):
                    if rf:
                        ev.append((k, G, 'wait', i + 1, (PD - 1) // 2))
                    elif i + 1 < nt:
                        ev.append((k, G, 'wait', i + 1, own_younger(i + 1)))
                if G == 1:
                    if PD % 2 == 0:
                        end_wait(0, False, k)
                    k += 1
                t = G
                while t < nt:
                    rf = t + 1 + PD < nt
                    ev.append((k, G, 'read', t, None))
                    if rf or t + PD < nt:
                        ev.append((k, G, 'issue', t + PD, None))
                    if PD % 2 == 1:
                        end_wait(t, rf, k)
                    k += 1
                    if rf or t + 1 < nt:
                        if PD % 2 == 0:
                            end_wait(t + 1, rf, k)
                        k += 1
                    t += 2
                nb.append(k)
            assert nb[0] == nb[1] == nt + 1, (NS, nt, nb)
            for t in range(nt):
                reads = [(tm, g) for tm, g, kind, tt, _ in ev if kind == 'read' and tt == t]
                assert len(reads) == 1 and reads[0][1] == (t & 1), (NS, nt, t)
                issues = [(tm, g) for tm, g, kind, tt, _ in ev if kind == 'issue' and tt == t]
                assert len(issues) == 1, (NS, nt, t)
                waits = [(tm, g, y) for tm, g, kind, tt, y in ev if kind == 'wait' and tt == t]
                assert len(waits) == 1 and waits[0][1] == issues[0][1], ('owner waits', NS, nt, t)
                tm_w, G, younger = waits[0]
                assert issues[0][0] <= tm_w < reads[0][0], ('RAW', NS, nt, t)
                issued = [tt for tm, g, kind, tt, _ in ev if kind == 'issue' and g == G and tm <= tm_w]
                assert younger == len([u for u in issued if u > t]), ('count', NS, nt, t, younger, issued)
                if t >= NS:
                    last_read = [tm for tm, g, kind, tt, _ in ev if kind == 'read' and tt == t - NS][0]
                    assert last_read < issues[0][0], ('WAR', NS, nt, t)


def test_pmc_step_summary_folds_the_lds_and_l2_passes(tmp_path):
    """tools/pmc_step_summary.py on a synthetic set of rocprofv3 counter CSVs: per-kernel bytes with the gfx950 FETCH_SIZE correction, MFMA busy
    against the 2.4 GHz peak clock, LDS conflict / wait fractions and the L2 hit rate -- the arithmetic DESIGN.md's roofline table rests on."""
    import json
    import subprocess
    import sys

    def write(tag, rows):
        d = tmp_path / tag
        d.mkdir()
        with open(d / f'{tag}_counter_collection.csv', 'w') as f:
            f.write('Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n')
            for did, k, c, v in rows:
                f.write(f'{did},"{k}",{c},{v},1000,11000\n')   # every dispatch lasts 10 us
    kg, kr = 'void (anonymous namespace)::k_gemm_pp<128, 288>(GemmArgs)', '(anonymous namespace)::k_row_w<false>(RowArgs)'
    write('fetch', [(1, kg, 'FETCH_SIZE', 1000), (2, kg, 'FETCH_SIZE', 3000), (3, kr, 'FETCH_SIZE', 500)])           # KB
    write('write', [(1, kg, 'WRITE_SIZE', 100), (2, kg, 'WRITE_SIZE', 300), (3, kr, 'WRITE_SIZE', 50)])
    write('sq', [(1, kg, 'SQ_VALU_MFMA_BUSY_CYCLES', 1024 * 12000), (2, kg, 'SQ_VALU_MFMA_BUSY_CYCLES', 1024 * 12000),
                 (1, kg, 'SQ_WAVE_CYCLES', 1000), (2, kg, 'SQ_WAVE_CYCLES', 1000), (3, kr, 'SQ_WAVE_CYCLES', 10)])
    write('lds', [(1, kg, 'SQ_LDS_BANK_CONFLICT', 30), (1, kg, 'SQ_LDS_IDX_ACTIVE', 1000), (1, kg, 'SQ_WAIT_INST_LDS', 50), (1, kg, 'SQ_WAVE_CYCLES', 1000)])
    write('tcc', [(1, kg, 'TCC_HIT_sum', 750), (1, kg, 'TCC_MISS_sum', 250), (1, kg, 'TCC_EA0_RDREQ_sum', 200)])
    write('dram', [(1, kg, 'TCC_EA0_RDREQ_DRAM_sum', 200)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_step_summary.py'), str(tmp_path), '2', '--tag', 't'],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    k = {row['kernel']: row for row in d['kernels']}
    g = k['k_gemm_pp<128, 288>']
    assert g['dispatches'] == 2 and g['fetch_bytes'] == 2 * 4000 * 1024 and g['write_bytes'] == 400 * 1024
    assert abs(g['mfma_busy_frac'] - 0.5) < 1e-9          # 12 000 busy cycles of the 24 000 a SIMD has in 10 us at 2.4 GHz
    assert abs(g['lds_bank_conflict_frac'] - 0.03) < 1e-12 and abs(g['wait_inst_lds_frac_of_wave_cycles'] - 0.05) < 1e-12
    assert abs(g['tcc_hit_rate'] - 0.75) < 1e-12 and g['ea_rdreq_dram_frac'] == 1.0
    assert d['fetch_bytes_per_step'] == (2 * 4000 + 2 * 500) * 1024 / 2 and d['write_bytes_per_step'] == 450 * 1024 / 2
    assert 'lds_bank_conflict_frac' not in k['k_row_w<false>']          # a kernel the optional passes did not see keeps the old columns only


def test_bench_line_extras_are_wired():
    """The keys and flags round 3 added to bench.py: the config-#4 shard measurement, the placement-test dump flags and the traffic label."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for needle in ("'config4_shard'", '--no-shard4', '--dump-latents', '--as-rank', 'L2<->fabric bytes per step', 'k_gemm_pp<128,288,4,2,3,EPI_GEGLU,1,64>'):
        assert needle in src, needle
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.GEGLU_VARIANT == 72000 + 60 * 4 + 2 and callable(bench.shard4_measure)   # tile 60, GEGLU epilogue, LayerNorm-algebra variant (the default path)
    # the XL step's algorithmic work the roofline fraction is computed from: 1.541 TFLOP for B = 2, L = 500, Lc = 100
    cfg = bench.model_section('xl')['model']
    assert abs(bench.flops_per_step(cfg, 2, 500, 100) / 1e12 - 1.541) < 1e-3


_ISA_CACHE = {}


def _gfx950_isa(src):
    """{mangled kernel name: its gfx950 assembly} of one translation unit of csrc/, compiled with the flags of the shipped build (cached per test session)."""
    if src not in _ISA_CACHE:
        import re
        import subprocess
        import tempfile
        from ezaudio_amd import build
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, src + '.s')
            subprocess.check_call([build._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S',
                                   os.path.join(ROOT, 'ezaudio_amd', 'csrc', src), '-o', out], stderr=subprocess.DEVNULL)
            text = open(out).read()
        _ISA_CACHE[src] = {f.split(':', 1)[0]: f for f in re.split(r'\n(?=_Z\w+:)', text) if f.startswith('_Z')}
    return _ISA_CACHE[src]


def test_kernels_of_the_default_step_neither_spill_nor_shuffle_through_the_lds():
    """Structural facts the round-4 epilogue work rests on, checked on the code hipcc generates today: (1) no kernel of the default step (one or four
    prompts) uses scratch memory -- the fused-QKV epilogue's table prefetch spilled when all of it was requested up front, and a spilling kernel
    pays a scratch set-up per launch; (2) the fused-QKV and K-split kernels contain no ds_bpermute (their quad / octet shuffles are DPP: a
    __shfl_xor that creeps back in is 100+ cycles of LDS pipe inside a dependency chain); (3) the fused-QKV kernels wait for their epilogue
    tables with COUNTED vmcnt (>= 10 younger loads left in flight somewhere), not with a drain at the first use."""
    import re
    funcs = _gfx950_isa('gemm.hip')

    def tmpl(name):   # '..k_gemm_ppILi128ELi144E..' -> ('k_gemm_pp', [128, 144, ...]); Lb1E / Lb0E are booleans
        m = re.search(r'(k_gemm_pp|k_gemm_ks|k_gemm)I((?:L[ib]\d+E)+)', name)
        return (m.group(1), [int(x) for x in re.findall(r'L[ib](\d+)E', m.group(2))]) if m else (None, [])

    default_step = {   # launch_gemm's instantiations on the default path (profiles/r04b_kernel_trace*.txt)
        ('k_gemm_pp', (128, 288, 4, 2, 3)), ('k_gemm_pp', (128, 144, 4, 1, 4)), ('k_gemm_ks', (3, 6)), ('k_gemm_ks', (3, 4)),
        ('k_gemm', (128, 128, 4, 2, 3)), ('k_gemm', (128, 64, 4, 2, 4)),
    }
    seen = set()
    for name, f in funcs.items():
        kind, args = tmpl(name)
        key = next((k for k in default_step if k[0] == kind and tuple(args[:len(k[1])]) == k[1]), None)
        if key is None:
            continue
        seen.add(key)
        assert 'scratch_' not in f, f'{name} spills to scratch'
        if kind == 'k_gemm_ks' or (kind == 'k_gemm_pp' and args[5] == 3):   # EPI_QKV = 3
            assert 'ds_bpermute' not in f, f'{name} shuffles through the LDS pipe'
        if kind == 'k_gemm_pp' and args[5] == 3:
            waits = [int(x) for x in re.findall(r's_waitcnt vmcnt\((\d+)\)', f)]
            assert max(waits) >= 20, (name, sorted(set(waits)))   # weights behind 2 x 10 younger table loads
    assert seen == default_step, default_step - seen


def test_k_split_kernel_counts_exactly_its_operand_loads_behind_the_prologue_dma():
    """k_gemm_ks decides that its own LDS-DMA has landed with COUNTED waits (`s_waitcnt vmcnt(NOPL)`): that is only right if exactly NOPL vector
    loads sit between the prologue's global_load_lds and the K loop, in that order (round-4 ADVICE).  Since round 5 those loads are inline asm into
    AGPRs, fenced by sched_barriers; this test reads the gfx950 code hipcc generates today and checks, per instantiation: (1) the shared-slot path
    issues exactly NOPL `global_load_dwordx4 a[..]` and nothing else that counts, (2) nothing waits on vmcnt between the first LDS-DMA and the loop's first
    wait except inside the per-row-timestep branch (whose slot is a dependent vector load), (3) the loop's first wait is vmcnt(NOPL), (4) the operand registers
    are disjoint from the MFMA accumulators and are not read before a vmcnt(0) behind the loop."""
    import re
    funcs = _gfx950_isa('gemm.hip')
    checked = 0
    for name, f in funcs.items():
        m = re.search(r'k_gemm_ksI((?:L[ib]\d+E)+)', name)
        if not m:
            continue
        fm, fn, epi, gate, res, ck, form = [int(x) for x in re.findall(r'L[ib](\d+)E', m.group(1))]   # form 3 = KS_ZIN: G' rides in the gate's slot, + 4 partial statistics (dwordx2)
        nopl4 = (fn // 2) * (1 + ((1 + res + (gate or form == 3)) if epi == 4 else 0))
        nopl2 = 4 if form == 3 else 0
        nopl = nopl4 + nopl2
        lines = [l.strip() for l in f.splitlines()]
        first_dma = next(i for i, l in enumerate(lines) if l.startswith('global_load_lds'))
        shared = next(i for i, l in enumerate(lines) if l == '; shared slot')
        blk = shared
        while not re.match(r'\.LBB\d+_\d+:', lines[blk]):
            blk -= 1
        body = lines[blk:shared]
        assert sum(l.startswith('global_load_dwordx4 a[') for l in body) == nopl4, (name, nopl4)
        assert sum(l.startswith('global_load_dwordx2 a[') for l in body) == nopl2, (name, nopl2)
        assert not any(l.startswith(('global_load_lds', 'buffer_load', 's_waitcnt vmcnt')) or (l.startswith('global_load') and not l.startswith(('global_load_dwordx4 a[', 'global_load_dwordx2 a['))) for l in body), name
        per_row = [i for i, l in enumerate(lines) if l == '; per-row slot']
        for i in range(first_dma, shared):
            if lines[i].startswith('s_waitcnt vmcnt'):
                assert per_row and i < per_row[0] and any(l.startswith('global_load_dword v') for l in lines[first_dma:i]), (name, i, lines[i])
        nxt = next(i for i in range(shared, len(lines)) if lines[i].startswith('s_waitcnt vmcnt'))
        assert lines[nxt] == 's_waitcnt vmcnt(%d)' % nopl, (name, lines[nxt])
        assert not any(l.startswith('global_load') for l in lines[shared:nxt]), name
        acc = [(int(a), int(b)) for a, b in re.findall(r'v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\]', f)]
        ops = [(int(a), int(b)) for a, b in re.findall(r'global_load_dwordx[24] a\[(\d+):(\d+)\]', f)]
        acc_regs = {r for a, b in acc for r in range(a, b + 1)}
        op_regs = {r for a, b in ops for r in range(a, b + 1)}
        assert not (acc_regs & op_regs), name
        last_mfma = max(i for i, l in enumerate(lines) if l.startswith('v_mfma'))
        first_read = next(i for i, l in enumerate(lines) if (mm := re.match(r'v_accvgpr_read_b32 v\d+, a(\d+)', l)) and int(mm.group(1)) in op_regs)
        assert first_read > last_mfma and any(l == 's_waitcnt vmcnt(0)' for l in lines[last_mfma:first_read]), name
        assert 'scratch_' not in f, name
        checked += 1
    assert checked >= 8, checked


def test_step_counter_scalar_load_is_not_touched_before_its_wait():
    """The device step counter must be ONE SCALAR load (`s_load_dword`, base not the kernarg pointer s[0:1]) whose destination nothing names before an
    `s_waitcnt` with lgkmcnt(0) -- along every path (control-flow aware).  History: round 5 requested it by inline asm under `if (a.cur_step)` (the
    compiler believes an asm output to be defined at once, so a copy the allocator inserts moves a value that has not arrived); the first round-6 build made
    the request unconditional and this test caught `s_mov_b32 s20, s53` between request and wait; since then it is a plain C++ load behind a NON-volatile
    kernel-argument batch (common.h) -- behind an `asm volatile` hipcc turns it into a vector load with `s_waitcnt vmcnt(0)` in front of the first LDS-DMA."""
    import re
    funcs = _gfx950_isa('gemm.hip')
    checked = 0
    for name, f in funcs.items():
        if not re.search(r'k_gemm_(pp|ks|co)I', name):
            continue
        lines = [l.strip().split(';')[0].strip() for l in f.splitlines()]
        lines = [l for l in lines if l and not l.startswith(('.p2align', '.long', '.byte', '.section', '.set', '.size', '.type', '.globl', '.amdhsa', '.end_amdhsa', '.text'))]
        req = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r's_load_dword (s\d+), s\[(?!0:1\])\d+:\d+\], 0x0$', l))]
        if not req:   # instantiations that do not use the modulation slot: the load is dead code
            continue
        assert len(req) == 1, (name, req)
        start, reg = req[0]
        pat = re.compile(r'\b%s\b' % reg)
        pending_in = {}
        for _ in range(8):
            pending, new_in = False, {}
            for i, ins in enumerate(lines):
                m = re.match(r'(\.LBB\d+_\d+):', ins)
                if m:
                    pending = pending or pending_in.get(m.group(1), False)
                    continue
                if i == start:
                    pending = True
                    continue
                op = ins.split()[0]
                if op == 's_waitcnt' and 'lgkmcnt(0)' in ins:
                    pending = False
                    continue
                if pending:
                    assert not pat.search(ins.split(None, 1)[1] if ' ' in ins else ''), f'{name}: {ins!r} touches {reg} before its wait'
                if op == 's_branch' or op.startswith('s_cbranch'):
                    tgt = ins.split()[-1]
                    new_in[tgt] = new_in.get(tgt, False) or pending
                    if op == 's_branch':
                        pending = False
                if op == 's_endpgm':
                    pending = False
            if new_in == pending_in:
                break
            pending_in = new_in
        checked += 1
    assert checked >= 12, checked


def test_dual_form_constant_vector_registers_are_not_touched_before_their_wait():
    """the DUAL form of k_gemm_ks requests the constant cross-attention-out vector by inline asm (`global_load_dwordx4 v[..]` between the park and the
    barrier) -- the compiler believes the outputs defined at once.  Straight-line check on the generated gfx950 code: between each of those loads (the vector
    loads into VGPRs behind the last MFMA) and the first `s_waitcnt vmcnt(0)` that follows them, no instruction names a destination register."""
    import re
    funcs = _gfx950_isa('gemm.hip')
    checked = 0
    for name, f in funcs.items():
        m = re.search(r'k_gemm_ksI((?:L[ib]\d+E)+)', name)
        if not m or [int(x) for x in re.findall(r'L[ib](\d+)E', m.group(1))][6] not in (1, 2):   # KS_DUAL, KS_COPY2 (the second operand's gain takes the same route)
            continue
        lines = [l.strip().split(';')[0].strip() for l in f.splitlines()]
        lines = [l for l in lines if l and not l.startswith('.')]
        last_mfma = max(i for i, l in enumerate(lines) if l.startswith('v_mfma'))
        loads = [(i, int(mm.group(1)), int(mm.group(2))) for i, l in enumerate(lines) if i > last_mfma and (mm := re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', l))]
        assert len(loads) == 3, (name, loads)   # SL = FN / 2 = 3 column slots per thread
        wait = next(i for i in range(loads[-1][0], len(lines)) if lines[i].startswith('s_waitcnt') and 'vmcnt(0)' in lines[i])
        dest = {i: set(range(a, b + 1)) for i, a, b in loads}
        regs = set()   # destinations of the loads issued so far
        for i in range(loads[0][0], wait):
            if i in dest:
                regs |= dest[i]
                continue
            ins = lines[i]
            ops = ins.split(None, 1)[1] if ' ' in ins else ''
            named = {int(x) for x in re.findall(r'\bv(\d+)\b', ops)}
            for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', ops):
                named |= set(range(int(a), int(b) + 1))
            assert not (named & regs), (name, ins)
        checked += 1
    assert checked >= 2, checked


def test_blind_statistics_loads_of_the_layernorm_algebra_consumers_are_not_touched_before_their_wait():
    """The LayerNorm-algebra consumers (k_gemm_pp, k_gemm_co) fetch the row statistics a lane needs by inline-asm loads into AGPRs (gemm_pp.h ld8_blind) behind their first K
    tile and merge them behind the loop; G' / C' go straight into the LDS by LDS-DMA.  hipcc neither counts nor waits for an asm load -- and believes its output defined at once
    (ADVICE r05).  On the generated gfx950 code of every consumer instantiation: (1) in front of the first MFMA hipcc itself neither waits on vmcnt nor loads a vector from global
    memory nor stores to the LDS (everything of that kind sits inside an inline-asm block): the loop starts behind the first K tile and nothing else; (2) from each inline-asm
    `global_load_dwordx2 a[..]`, in program-text order, no instruction READS a destination register before an `s_waitcnt` with vmcnt(0) has been passed (a copy the
    allocator inserts next to the accumulators would move data that has not arrived; an MFMA that accumulates into one of them would be worse)."""
    import re
    funcs = _gfx950_isa('gemm.hip')
    checked = 0
    for name, f in funcs.items():
        m = re.search(r'k_gemm_(pp|co)I((?:L[ib]\d+E)+)', name)
        if not m:
            continue
        targs = [int(x) for x in re.findall(r'L[ib](\d+)E', m.group(2))]
        epi, var = (targs[5], targs[7]) if m.group(1) == 'pp' else (targs[2], targs[3])
        if not (var & 64) or epi not in (2, 3):   # EPI_GEGLU = 2, EPI_QKV = 3 with the LayerNorm algebra
            continue
        raw = f.splitlines()
        first_mfma_raw = next(i for i, l in enumerate(raw) if l.strip().startswith('v_mfma'))
        inasm = False
        for l in raw[:first_mfma_raw]:
            t = l.strip()
            if t.startswith(';;#ASMSTART'):
                inasm = True
            elif t.startswith(';;#ASMEND'):
                inasm = False
            elif not inasm:
                assert not (t.startswith('s_waitcnt') and 'vmcnt' in t), (name, t)
                assert not re.match(r'global_load_dword\w* v', t), (name, t)
                assert not t.startswith('ds_write'), (name, t)
        lines, blind, inasm = [], set(), False   # blind = indices (into `lines`) of the vector loads that sit inside an inline-asm block (hipcc tracks the others itself)
        for l in raw:
            t = l.strip()
            if t.startswith(';;#ASMSTART'):
                inasm = True
                continue
            if t.startswith(';;#ASMEND'):
                inasm = False
                continue
            t = t.split(';')[0].strip()
            if t and (not t.startswith('.') or re.match(r'\.LBB\d+_\d+:', t)):
                if inasm and t.startswith('global_load_dwordx2 a['):
                    blind.add(len(lines))
                lines.append(t)
        loads = {i: set(range(int(mm.group(1)), int(mm.group(2)) + 1)) for i in blind if (mm := re.match(r'global_load_dwordx2 a\[(\d+):(\d+)\]', lines[i]))}
        assert len(loads) >= 3, (name, len(loads))

        def regs_of(text):
            regs = {int(x) for x in re.findall(r'\ba(\d+)\b', text)}   # (the blind loads land in AGPRs)
            for x, y in re.findall(r'\ba\[(\d+):(\d+)\]', text):
                regs |= set(range(int(x), int(y) + 1))
            return regs

        # from each blind load, in program-text order: an instruction that READS a destination register must have an `s_waitcnt .. vmcnt(0)` between the load and itself; an
        # instruction that only WRITES one ends that register's watch (a new value: the code of the other wave group, which the layout puts behind this one's)
        for i, dest in loads.items():
            pending, waited = set(dest), False
            for ins in lines[i + 1:]:
                if not pending or ins.startswith('s_endpgm'):
                    break
                if re.match(r'\.LBB\d+_\d+:', ins) or ' ' not in ins:
                    continue
                op, ops = ins.split(None, 1)
                if op == 's_waitcnt':
                    waited = waited or 'vmcnt(0)' in ops
                    continue
                is_store = op.startswith(('global_store', 'ds_write', 'buffer_store', 'scratch_store'))
                first, _, rest = ops.partition(',')
                read = regs_of(ops if is_store else rest) & pending
                if op.startswith(('global_load', 'buffer_load', 'ds_read')):   # (address operands)
                    read = regs_of(rest) & pending
                assert not read or waited, f'{name}: {ins!r} reads a blind load\'s destination before any vmcnt(0)'
                if not is_store:
                    pending -= regs_of(first)
        checked += 1
    assert checked >= 4, checked


def test_inline_asm_mfma_results_are_read_behind_their_wait_states():
    """k_gemm_pp, k_gemm_ks and k_attn issue their MFMAs from inline asm, so hipcc's hazard recogniser neither sees them nor pads behind them
    (round-3 ADVICE): the wait states between the LAST MFMA of an accumulation chain and the first non-MFMA read of its accumulator are
    hand-placed `s_nop`s.  This test disassembles the gfx950 code hipcc generates for those sources and checks, per kernel, that every
    instruction which reads an accumulator register (AGPR operand of a v_accvgpr_read / ds_write / global_store / VALU instruction, or the
    VGPR accumulators of k_attn) sits at least 18 issue slots (s_nop N counts N + 1) behind the textually preceding MFMA -- a refactor or a
    compiler upgrade that moves a reader up fails here instead of corrupting tiles silently."""
    import re
    INF = 10 ** 9
    for src, kernels in (('gemm.hip', ('k_gemm_pp', 'k_gemm_ks')),):
        funcs = _gfx950_isa(src)
        checked = 0
        for name, f in funcs.items():
            if not any(k in name for k in kernels) or 'v_mfma' not in f:
                continue
            lines = [l.strip() for l in f.splitlines() if (l.startswith('\t') or re.match(r'\.LBB\d+_\d+:', l)) and not l.strip().startswith((';', '.p2align', '.long', '.byte'))]
            # the accumulators = the AGPRs some MFMA of this kernel writes; hipcc also parks spilled VGPRs in the AGPRs above them (v_accvgpr_write / _read around the loops:
            # no matrix-pipe hazard on those)
            acc_regs = set()
            for ins in lines:
                mm = re.match(r'v_mfma\S* a\[(\d+):(\d+)\]', ins)
                if mm:
                    acc_regs |= set(range(int(mm.group(1)), int(mm.group(2)) + 1))

            def names_acc(ops):
                named = {int(x) for x in re.findall(r'\ba(\d+)\b', ops)}
                for x, y in re.findall(r'\ba\[(\d+):(\d+)\]', ops):
                    named |= set(range(int(x), int(y) + 1))
                return bool(named & acc_regs)
            # dist = issue slots since the last MFMA on ANY path into this point: a label takes the minimum over its fall-through and every branch that
            # targets it (round 5: the register allocator put accumulator copies on a loop-exit EDGE, which a purely textual scan -- whose predecessor
            # was an unrelated block -- did not see); iterated to a fixed point because loop back-edges come textually after their header
            label_in = {}
            for _ in range(6):
                dist, new_in, reads = INF, {}, []
                for ins in lines:
                    m = re.match(r'(\.LBB\d+_\d+):', ins)
                    if m:
                        dist = min(dist, label_in.get(m.group(1), INF))
                        continue
                    op = ins.split()[0]
                    if op.startswith('v_mfma'):
                        m2 = re.match(r's_nop (\d+)', '')
                        dist = 0
                        continue
                    if op in ('s_branch',) or op.startswith('s_cbranch'):
                        tgt = ins.split()[-1]
                        new_in[tgt] = min(new_in.get(tgt, INF), dist + 1)
                        dist = INF if op == 's_branch' else dist + 1
                        continue
                    opnds = ins.split(None, 1)[1] if ' ' in ins else ''
                    if op in ('v_accvgpr_read_b32', 'v_accvgpr_mov_b32'):
                        reads_acc = names_acc(opnds.split(',', 1)[1] if ',' in opnds else opnds)   # (the source operand)
                    else:
                        reads_acc = op.startswith(('ds_write', 'global_store', 'v_')) and not op.startswith('v_accvgpr_write') and names_acc(opnds)
                    if reads_acc and dist < INF:
                        reads.append((ins, dist))
                        dist = INF          # chain consumed; the next MFMA re-arms the check
                        continue
                    m2 = re.match(r's_nop (\d+)', ins)
                    if dist < INF:
                        dist += int(m2.group(1)) + 1 if m2 else 1
                if new_in == label_in:
                    break
                label_in = new_in
            for ins, d in reads:
                assert d >= 18, (name, ins, d)
            checked += len(reads)
        assert checked >= 10, checked
