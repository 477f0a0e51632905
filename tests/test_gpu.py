"""GPU parity tests (run with -m gpu on an MI355X).  Everything calls through the C ABI of
libezaudio_hip.so; the numpy oracle and the golden vectors minted from the reference are the checkers.

Tolerances (bf16 storage / fp32 accumulate path):
  single forward vs the reference's fp32 output:  rel-L2 <= 2e-2 and max-abs <= 0.15
  (the reference's own bf16-vs-fp64 gap on XL is 9.3e-3 / 6.2e-2 on outputs with sigma ~ 1.5).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.dit import DiTOracle
from oracle.sampler import sample as oracle_sample
from oracle.weights import make_inputs, make_state_dict, model_config, uniform_pm1
from tests.util import DIFF, golden_case, record, rel_l2, sampler_case

pytestmark = pytest.mark.gpu

REL_TOL, ABS_TOL = 2e-2, 0.15


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return 'cuda:0'


_models = {}


def get_model(size, seed):
    from ezaudio_amd import MaskDiT
    key = (size, seed)
    if key not in _models:
        if len(_models) >= 3:
            _models.pop(next(iter(_models)))
        cfg = model_config(size)
        m = MaskDiT(device='cuda:0', **cfg)
        m.load_state_dict(make_state_dict(cfg, seed))
        _models[key] = m
    return _models[key]


def t_(a, dev='cuda:0'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ---------------------------------------------------------------------------------------------------
# kernel level
# ---------------------------------------------------------------------------------------------------
# variant = tile_config * 4 + epilogue; epilogue 0 = fp32 (+bias), 1 = split-K partial slabs, 2 = GEGLU
# tile_config: the table in csrc/gemm.hip (6, 9, 13, 25 lockstep k_gemm; 60-62 ping-pong k_gemm_pp; 66 co-resident k_gemm_co; 70, 72, 73 K-split k_gemm_ks)
@pytest.mark.parametrize('M,N,K,variant,splitk', [
    (1000, 1152, 1152, 6 * 4 + 0, 1),   # 128x64, waves 4x1
    (1000, 1152, 1152, 9 * 4 + 1, 3),   # 128x128, 8 waves, ring 3
    (1000, 1152, 1152, 13 * 4 + 0, 1),  # 128x288, ring 3
    (1000, 1152, 1152, 13 * 4 + 1, 6),  # 128x288, ring 3, 3 K tiles per slice
    # ping-pong kernel (k_gemm_pp): two wave groups one barrier interval apart.  SCHED 1 (60: 128x288, 62: 128x128) and the k-split
    # SCHED 2 (61: 128x144 ring 4); every K-tile count from 1 up (prologue, steady and drain paths, odd /
    # even tile counts of the two groups), ragged M / N, uneven K splits; + 16000: bf16 slabs through LDS
    (1000, 1152, 1152, 60 * 4 + 0, 1),
    (130, 288, 64, 60 * 4 + 0, 1),             # single K tile
    (200, 432, 128, 60 * 4 + 0, 1),            # two K tiles
    (77, 288, 192, 60 * 4 + 0, 1),             # three
    (1000, 300, 320, 60 * 4 + 1, 1),           # five, ragged N
    (1000, 1152, 1152, 60 * 4 + 1, 6),         # 3 K tiles per slice
    (1000, 1152, 1152, 62 * 4 + 1, 3),
    (1000, 1152, 4608, 62 * 4 + 1, 3),
    (1000, 1152, 1152, 62 * 4 + 1, 9),         # 2 K tiles per slice
    (300, 256, 192, 62 * 4 + 1, 2),            # uneven split: 1 + 2 tiles
    (130, 128, 64, 62 * 4 + 0, 1),
    (1000, 1152, 2304, 16000 + 62 * 4 + 1, 3), # bf16 slabs
    (300, 200, 192, 16000 + 62 * 4 + 1, 2),
    (4000, 1152, 4608, 16000 + 8000 + 60 * 4 + 1, 2),   # the batched-prompt MLP-out configuration: 128x288, split-K 2, bf16 slabs through LDS
    (4000, 1152, 1152, 16000 + 8000 + 60 * 4 + 1, 2),
    (1000, 3456, 1152, 61 * 4 + 0, 1),
    (130, 144, 64, 61 * 4 + 0, 1),
    (200, 288, 128, 61 * 4 + 0, 1),
    (77, 144, 192, 61 * 4 + 0, 1),
    (500, 300, 256, 61 * 4 + 1, 1),            # four K tiles
    (500, 300, 320, 61 * 4 + 1, 1),            # five
    (1000, 1152, 1152, 61 * 4 + 1, 3),         # six per slice
    # co-resident kernel (k_gemm_co, 66: 128x144, 4 waves, ring 2, two workgroups per CU): every K-tile count from 1 up (prologue-only, the two tail
    # forms, the steady loop), ragged M / N, K splits
    (1000, 3456, 1152, 66 * 4 + 0, 1),
    (4000, 9216, 1152, 66 * 4 + 0, 1),
    (130, 144, 64, 66 * 4 + 0, 1),             # one K tile
    (200, 288, 128, 66 * 4 + 0, 1),            # two
    (77, 144, 192, 66 * 4 + 0, 1),             # three
    (500, 300, 256, 66 * 4 + 1, 1),            # four, ragged N
    (500, 300, 320, 66 * 4 + 1, 1),            # five
    (1000, 1152, 1152, 66 * 4 + 1, 3),         # six per slice
    (300, 200, 192, 16000 + 8000 + 66 * 4 + 1, 2),   # bf16 slabs through LDS, uneven split
    # K-split-inside-the-workgroup kernel (k_gemm_ks; 70: 48x96, 72: 32x96, 73: 48x64): K chunk counts below, at and
    # above the eight waves (idle waves, uneven shares), ragged M / N
    (1000, 1152, 1152, 70 * 4 + 0, 1),
    (1000, 1152, 4608, 70 * 4 + 0, 1),
    (1000, 1152, 2304, 70 * 4 + 0, 1),
    (1000, 1024, 1024, 70 * 4 + 0, 1),         # ragged last N tile (1024 = 10 * 96 + 64)
    (77, 100, 64, 70 * 4 + 0, 1),              # one chunk: seven waves idle
    (130, 96, 192, 70 * 4 + 0, 1),
    (50, 200, 576, 70 * 4 + 0, 1),             # nine chunks
    (4000, 1152, 1152, 70 * 4 + 0, 1),
    (1000, 1152, 1152, 72 * 4 + 0, 1),
    (90, 100, 128, 72 * 4 + 0, 1),
    (1000, 1152, 1152, 73 * 4 + 0, 1),
    (90, 100, 640, 73 * 4 + 0, 1),
])
def test_gemm_against_fp32_matmul(lib, dev, M, N, K, variant, splitk):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    Np = (N + 127) // 128 * 128
    W = torch.zeros(Np, K, dtype=torch.bfloat16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    ref = A.float().double() @ W[:N].float().double().T
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    Mp = (M + 127) // 128 * 128
    epi = variant % 1000 % 4
    if epi == 0:
        out = torch.full((M, N), float('nan'), device=dev)
        rc = lib.ezdit_test_gemm(None, variant, Ad.data_ptr(), K, Wd.data_ptr(), K, bd.data_ptr(), out.data_ptr(), N, M, N, K, 1, None)
        assert rc == 0
        got = out.cpu().double()
        ref = ref + bias.double()
    else:
        slab_bf16 = (variant // 1000) & 16
        out = torch.zeros((splitk, Mp, N), device=dev, dtype=torch.bfloat16 if slab_bf16 else torch.float32)
        rc = lib.ezdit_test_gemm(None, variant, Ad.data_ptr(), K, Wd.data_ptr(), K, None, out.data_ptr(), N, M, N, K, splitk, None)
        assert rc == 0
        got = out.cpu().double().sum(0)[:M]
        if slab_bf16:   # one bf16 rounding per slab
            torch.cuda.synchronize()
            assert rel_l2(got.numpy(), ref.numpy()) < 4e-3
            return
    torch.cuda.synchronize()
    err = (got - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err   # fp32 accumulation of exact bf16 products
    assert rel_l2(got.numpy(), ref.numpy()) < 1e-5


ZW = {61: 144, 70: 96, 72: 96, 73: 64}   # statistics part width = the producer's tile width


@pytest.mark.parametrize('tile,mode', [(61, 'gr'), (61, 'r'), (61, ''), (70, 'gr'), (70, 'r'), (70, ''), (72, 'gr'), (72, 'r'), (72, ''), (73, 'r')])
@pytest.mark.parametrize('M,N,K', [(1000, 1152, 1152), (1000, 1152, 4608), (192, 144, 192), (192, 128, 128), (77, 576, 64), (500, 1024, 320), (1000, 1152, 2304)])
def test_residual_gemm_with_layernorm_statistics(lib, dev, M, N, K, tile, mode):
    _resid_case(lib, dev, M, N, K, tile, mode)


@pytest.mark.parametrize('tile', [61, 70])
def test_residual_gemm_statistics_survive_a_row_mean_far_above_the_spread(lib, dev, tile):
    """The statistics are the one-pass (sum, sum of squares) form: the relative error of the merged variance is eps_fp32 (1 + mu^2 / var) (common.h).
    Rows whose mean is 50x their spread -- far beyond what the residual stream shows -- must still give a variance good to 1e-2 (measured ~1e-3)
    and a mean good to 1e-5 relative (round-4 ADVICE)."""
    _resid_case(lib, dev, 192, 1152, 128, tile, 'gr', hmean=50.0, var_rtol=1e-2, mean_atol=1e-3)


def _resid_case(lib, dev, M, N, K, tile, mode, hmean=0.7, var_rtol=2e-5, mean_atol=2e-5):
    """Producer side of the LayerNorm algebra (EPI_RESID of the K-split-inside-the-workgroup kernel k_gemm_ks, tiles 70+, and of the ping-pong
    kernel's 128 x 144 tile, 61, the producer for batched prompts):
    h_new = h + gate * (A W^T + b) in fp32 (mode 'gr'; 'r': no gate; '': no residual either -- skip_linear), its per-column-tile
    (sum, sum of squares) statistics (part-major: [N tiles][M]) -- merged here and compared with the row's true mean / variance --
    and A' = bf16(h_new * g)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    Np = (N + 127) // 128 * 128
    W = torch.zeros(Np, K, dtype=torch.bfloat16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias, gate, zg = torch.randn(N, generator=g), torch.rand(N, generator=g), 1 + 0.3 * torch.randn(N, generator=g)
    h_in = torch.randn(M, N, generator=g) + hmean        # a row mean that is not small against the spread
    ref = A.float().double() @ W[:N].float().double().T + bias.double()
    if 'g' in mode:
        ref = gate.double() * ref
    if 'r' in mode:
        ref = h_in.double() + ref
    ld = (N + 63) // 64 * 64
    cw = ZW[tile]
    parts = (N + cw - 1) // cw
    Ad, Wd, bd, gd, zd, hd = A.to(dev), W.to(dev), bias.to(dev), gate.to(dev), zg.to(dev), h_in.to(dev)
    h_out = torch.full((M, N), float('nan'), device=dev)
    zu = torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
    zs = torch.zeros(parts, M, 2, device=dev)
    rc = lib.ezdit_test_resid(tile, Ad.data_ptr(), K, Wd.data_ptr(), K, bd.data_ptr(), hd.data_ptr() if 'r' in mode else None,
                              gd.data_ptr() if 'g' in mode else None, zd.data_ptr(), h_out.data_ptr(), zu.data_ptr(), ld, zs.data_ptr(), M, N, K, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = h_out.cpu().double()
    assert rel_l2(got.numpy(), ref.numpy()) < 1e-5
    st = zs.cpu().double().permute(1, 0, 2)
    n = torch.tensor([min(cw, N - cw * p) for p in range(parts)], dtype=torch.float64)
    mean = st[:, :, 0].sum(1) / N                                  # part-wise (sum, sum of squares): the merge is two plain sums
    var = st[:, :, 1].sum(1) / N - mean ** 2
    np.testing.assert_allclose(mean.numpy(), ref.mean(1).numpy(), rtol=0, atol=mean_atol)
    np.testing.assert_allclose(var.numpy(), ref.var(1, unbiased=False).numpy(), rtol=var_rtol)
    del n
    want = (got * zg.double()).float()
    assert rel_l2(zu.float().cpu().numpy()[:, :N], want.numpy()) < 3e-3     # one bf16 rounding
    assert (zu.float().cpu()[:, N:] == 0).all()


@pytest.mark.parametrize('tile', [70, 61])   # k_gemm_ks (one prompt) and the ping-pong producer (batched prompts)
@pytest.mark.parametrize('M,D', [(1000, 1152), (300, 576), (77, 160)])   # 12 | 8, 6 | 4 and 2 statistics parts per half (fewer parts than lanes that fetch them)
def test_skip_path_forms_of_the_residual_gemm(lib, dev, tile, M, D):
    """The two forms the skip path adds to the un-split residual projection (GemmArgs COPY2 / ZIN), stand-alone against fp64.
    COPY2 (the in-blocks' MLP-out, K = 4 D): everything _resid_case checks + the second operand bf16(h_new * g2).
    ZIN (skip_linear, K = 2 D): the operand is bf16([x | skip] * g) with the halves' partial statistics in two part-major sets; the launch must produce
    LN_2D([x | skip]) g-folded through W -- r (acc - mu G') + C' -- and then its own statistics and next operand like any producer.  Reference: the LayerNorm of the
    UNROUNDED rows through fp64 (the operand's bf16 rounding is the only difference: 3e-3)."""
    cw = ZW[tile]
    parts = (D + cw - 1) // cw
    g = torch.Generator().manual_seed(tile + M + D)
    ld = (D + 63) // 64 * 64
    Np = (D + 127) // 128 * 128
    # ---- COPY2 ----
    K = 4 * D
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = torch.zeros(Np, K, dtype=torch.bfloat16)
    W[:D] = (torch.randn(D, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias, gate = torch.randn(D, generator=g), torch.rand(D, generator=g)
    zg, zg2 = 1 + 0.3 * torch.randn(D, generator=g), 1 + 0.3 * torch.randn(D, generator=g)
    h_in = torch.randn(M, D, generator=g) + 0.7
    ref = h_in.double() + gate.double() * (A.float().double() @ W[:D].float().double().T + bias.double())
    dv = [t.to(dev) for t in (A, W, bias, h_in, gate, zg, zg2)]
    h_out = torch.full((M, D), float('nan'), device=dev)
    zu = torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
    zu2 = torch.zeros(M, 2 * ld, dtype=torch.bfloat16, device=dev)     # the right half of an [M][2 ld] operand, as in the step
    zs = torch.zeros(parts, M, 2, device=dev)
    rc = lib.ezdit_test_resid_skip(tile, dv[0].data_ptr(), K, dv[1].data_ptr(), K, dv[2].data_ptr(), dv[3].data_ptr(), dv[4].data_ptr(), dv[5].data_ptr(),
                                   h_out.data_ptr(), zu.data_ptr(), ld, zs.data_ptr(), M, D, K, dv[6].data_ptr(), zu2.data_ptr() + 2 * ld, 2 * ld, None, None, 0, 0, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = h_out.cpu().double()
    assert rel_l2(got.numpy(), ref.numpy()) < 1e-5
    assert rel_l2(zu.float().cpu().numpy()[:, :D], (got * zg.double()).float().numpy()) < 3e-3
    z2 = zu2.float().cpu()
    assert rel_l2(z2.numpy()[:, ld:ld + D], (got * zg2.double()).float().numpy()) < 3e-3
    assert (z2[:, :ld] == 0).all() and (z2[:, ld + D:] == 0).all()          # nothing outside the half it owns
    st = zs.cpu().double().permute(1, 0, 2)
    np.testing.assert_allclose((st[:, :, 0].sum(1) / D).numpy(), ref.mean(1).numpy(), rtol=0, atol=2e-5)
    # ---- ZIN ----
    K = 2 * D
    x = torch.randn(M, K, generator=g) * (1 + torch.rand(M, 1, generator=g)) + 0.5 * torch.randn(M, 1, generator=g)    # rows [x | skip] with their own mean and spread
    gam, bet = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    Wf = torch.randn(D, K, generator=g) / K ** 0.5
    W = torch.zeros(Np, K, dtype=torch.bfloat16)
    W[:D] = Wf.to(torch.bfloat16)
    Wd = W[:D].float().double()
    b = torch.randn(D, generator=g)
    A = (x * gam).to(torch.bfloat16)
    mu, var = x.double().mean(1, keepdim=True), x.double().var(1, unbiased=False, keepdim=True)
    ref = ((x.double() - mu) / torch.sqrt(var + 1e-5) * gam.double() + bet.double()) @ Wd.T + b.double()
    Gp, Cp = (gam.double() @ Wd.T).float(), (bet.double() @ Wd.T + b.double()).float()
    xs = x.double()

    def part_stats(half):      # [parts][M][2]: (sum, sum of squares) over each cw-column tile of a half
        out = torch.zeros(parts, M, 2, dtype=torch.float64)
        for p_ in range(parts):
            c = half[:, p_ * cw:min((p_ + 1) * cw, D)]
            out[p_, :, 0], out[p_, :, 1] = c.sum(1), (c * c).sum(1)
        return out.float()
    s1, s2 = part_stats(xs[:, :D]), part_stats(xs[:, D:])
    zg = 1 + 0.3 * torch.randn(D, generator=g)
    dv = [t.to(dev) for t in (A, W, Cp, zg, s1, s2, Gp)]
    h_out = torch.full((M, D), float('nan'), device=dev)
    zu = torch.zeros(M, ld, dtype=torch.bfloat16, device=dev)
    zs = torch.zeros(parts, M, 2, device=dev)
    rc = lib.ezdit_test_resid_skip(tile, dv[0].data_ptr(), K, dv[1].data_ptr(), K, dv[2].data_ptr(), None, None, dv[3].data_ptr(),
                                   h_out.data_ptr(), zu.data_ptr(), ld, zs.data_ptr(), M, D, K, None, None, 0, dv[4].data_ptr(), dv[5].data_ptr(), parts, K, dv[6].data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    got = h_out.cpu().double()
    r = rel_l2(got.numpy(), ref.numpy())
    record(f'ZIN tile {tile} M={M} D={D}: rel-L2 {r:.3e}')
    assert r < 5e-3                                                     # the bf16 rounding of the operand
    assert rel_l2(zu.float().cpu().numpy()[:, :D], (got * zg.double()).float().numpy()) < 3e-3
    st = zs.cpu().double().permute(1, 0, 2)
    np.testing.assert_allclose((st[:, :, 0].sum(1) / D).numpy(), got.mean(1).numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose((st[:, :, 1].sum(1) / D - (st[:, :, 0].sum(1) / D) ** 2).numpy(), got.var(1, unbiased=False).numpy(), rtol=1e-4)


@pytest.mark.parametrize('tile', [6, 13, 2013, 2060, 2061, 2062, 2066])   # + 2000: LDS-staged epilogue; 60+: ping-pong kernel; 66: co-resident kernel
def test_gemm_geglu_epilogue(lib, dev, tile):
    M, D, inner = 300, 128, 576
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, D, generator=g).to(torch.bfloat16)
    W = (torch.randn(2 * inner, D, generator=g) / D ** 0.5).to(torch.bfloat16)
    b = torch.randn(2 * inner, generator=g) * 0.1
    from ezaudio_amd.weights import _geglu8
    Wi, bi = _geglu8(W.float()).to(torch.bfloat16), _geglu8(b.reshape(-1, 1)).reshape(-1)
    h = A.float().double() @ W.float().double().T + b.double()
    val, gate = h[:, :inner], h[:, inner:]
    ref = val * torch.nn.functional.gelu(gate)
    out = torch.zeros(M, inner, dtype=torch.bfloat16, device=dev)
    Ad, Wd, bd = A.to(dev), Wi.to(dev), bi.to(dev)   # keep the device tensors alive across the launch
    variant = 4000 * (tile // 1000) + (tile % 1000) * 4 + 2
    rc = lib.ezdit_test_gemm(None, variant, Ad.data_ptr(), D, Wd.data_ptr(), D, bd.data_ptr(), out.data_ptr(),
                             inner, M, 2 * inner, D, 1, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < 4e-3  # one bf16 rounding of the output


@pytest.mark.parametrize('nkh', [2, 4])   # 64-key tiles / 4 waves and 128-key tiles / 8 waves
@pytest.mark.parametrize('size,Lq,Lk,masked', [('xs', 96, 96, False), ('xs64', 96, 96, False), ('xs', 500, 100, True), ('xs', 64, 20, True),
                                               ('xs64', 500, 100, True), ('xs', 500, 500, False), ('xs64', 77, 500, False),
                                               ('xs', 300, 300, False), ('xs', 500, 129, True), ('xs64', 130, 512, False), ('xs', 64, 385, True)])
def test_attention_against_softmax_reference(lib, dev, size, Lq, Lk, masked, nkh):
    m = get_model(size, 1)
    cfg = model_config(size)
    H, D = cfg['num_heads'], cfg['embed_dim']
    dh = D // H
    DQK, DV = (64, 64) if dh == 64 else (80, 96)
    B = 2
    pad = 32 * nkh
    Lqp, Lkp = (Lq + 63) // 64 * 64, (Lk + pad - 1) // pad * pad
    assert lib.ezdit_set_option(m._h, b'attn_nkh', nkh) == 0
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = torch.randn(B, H, Lq, dh, generator=g).to(torch.bfloat16)
    k = torch.randn(B, H, Lk, dh, generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, Lk, dh, generator=g).to(torch.bfloat16)
    k[0, 0, 3] *= 6.0  # a spiked key forces a large running-max jump in the online softmax
    mask = torch.ones(B, Lk, dtype=torch.bool)
    if masked:
        mask[0, 12:] = False
        mask[1, 1:] = False
    qp = torch.zeros(B, H, Lqp, DQK, dtype=torch.bfloat16); qp[:, :, :Lq, :dh] = q
    kp = torch.zeros(B, H, Lkp, DQK, dtype=torch.bfloat16); kp[:, :, :Lk, :dh] = k
    vp = torch.zeros(B, H, Lkp, DV, dtype=torch.bfloat16); vp[:, :, :Lk, :dh] = v   # row-major: keys x channels
    ldD = (D + 63) // 64 * 64
    out = torch.zeros(B * Lq, ldD, dtype=torch.bfloat16, device=dev)
    md = mask.to(torch.uint8).to(dev)
    qd, kd, vd = qp.to(dev), kp.to(dev), vp.to(dev)   # keep the device tensors alive across the launch
    s = (q.double() @ k.double().transpose(2, 3)) * dh ** -0.5
    s = s.masked_fill(~mask[:, None, None, :], float('-inf'))
    ref = (torch.softmax(s, -1) @ v.double()).transpose(1, 2).reshape(B * Lq, D)
    out.zero_()
    rc = lib.ezdit_test_attention(m._h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                  md.data_ptr() if masked else None, out.data_ptr(), B, Lq, Lk, Lqp, Lkp, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.float().cpu()[:, :D]
    assert torch.isfinite(got).all()
    assert rel_l2(got.numpy(), ref.numpy()) < 1.2e-2  # P and O rounded to bf16
    assert (got.double() - ref).abs().max().item() < 0.06
    assert lib.ezdit_set_option(m._h, b'attn_nkh', 0) == 0


# ---------------------------------------------------------------------------------------------------
# model level: MaskDiT.forward vs the reference's golden outputs and the oracle
# ---------------------------------------------------------------------------------------------------
def _forward(m, inp, t, kw):
    tk = {}
    if 'gt' in kw:
        tk = dict(gt=t_(kw['gt']), mae_mask_infer=t_(kw['mae_mask_infer']))
    if 'controlnet_skips' in kw:
        x257, _ = m(t_(inp['x']), torch.tensor(t), None, forward_model=False, **tk)
        return m.model(x257, torch.tensor(t), t_(inp['ctx']), context_mask=t_(inp['ctx_mask']), cls_token=None,
                       controlnet_skips=[t_(s) for s in kw['controlnet_skips']])
    pred, mae_mask = m(t_(inp['x']), torch.tensor(t), t_(inp['ctx']), context_mask=t_(inp['ctx_mask']), cls_token=None, **tk)
    assert mae_mask.shape == pred.shape
    return pred


@pytest.mark.parametrize('name', ['xs', 'xs64', 'xs_edit', 'xs_cn', 's', 's64', 's_edit', 'l', 'l_edit', 'xl', 'xl_b8'])
def test_forward_matches_reference_golden(lib, dev, name):
    """xl_b8: BASELINE config #4's per-GPU shape (4 prompts = 8 denoiser rows, M = 4000 token rows at XL width), which takes the
    large-M tile configurations."""
    cfg, sd, inp, kw, g, meta = golden_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    for t in meta['timesteps']:
        pred = _forward(m, inp, t, kw).cpu().numpy()
        ref = g[f'pred_t{t}']
        assert np.isfinite(pred).all()
        r, a = rel_l2(pred, ref), float(np.abs(pred - ref).max())
        record(f'{name} t={t}: rel-L2 {r:.3e} max-abs {a:.3e}')
        assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48), (name, t, r, a)


@pytest.mark.parametrize('name', ['xs', 'xs64', 'xs_edit', 's', 's64', 'l', 'xl', 'xl_b8'])
def test_layernorm_algebra_and_split_k_paths_match_reference_golden(lib, dev, name):
    """Option zfuse (default ON): un-split residual projections (k_gemm_ks) whose epilogue emits h, per-column-tile LayerNorm
    statistics and the next GEMM's operand h * g, the consumer GEMM finishing the LayerNorm as r (acc - mu G') + C' in its epilogue -- against
    zfuse = 0 (split-K slabs + the row kernel): both within the same gates of the reference's own outputs, and the launch counts that say
    which path ran (no slabs and no row kernel on the attention-out / cross-out / skip / in-block MLP-out edges)."""
    cfg, sd, inp, kw, g, meta = golden_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    t = meta['timesteps'][0]
    ref = g[f'pred_t{t}']
    pred = _forward(m, inp, t, kw).cpu().numpy()
    n_z = m.last_launch_count
    assert lib.ezdit_set_option(m._h, b'zfuse', 0) == 0
    try:
        base = _forward(m, inp, t, kw).cpu().numpy()
        n_base = m.last_launch_count
    finally:
        assert lib.ezdit_set_option(m._h, b'zfuse', 1) == 0
    nblk = cfg['depth'] + 1
    # attention-out and cross-out of every block, MLP-out in front of in / mid blocks, skip_linear of the out-blocks; + the MLP-out in front of every out-block
    # (option skip_z: LN_2D([x | skip]) -> skip_linear by the algebra as well); + block 0's norm1 (the patch embed is its producer)
    assert n_z == n_base - (2 * nblk + cfg['depth'] + cfg['depth'] // 2 + 1)
    for what, p in (('zfuse', pred), ('split-K', base)):
        r, a = rel_l2(p, ref), float(np.abs(p - ref).max())
        record(f'{name} t={t} {what}: rel-L2 {r:.3e} max-abs {a:.3e}; launches {n_base} -> {n_z}')
        assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48)


@pytest.mark.parametrize('name', ['xs', 'xs64', 'xs_edit', 'xs_cn', 's', 'l', 'xl', 'xl_b8'])   # xl_b8: 4000 token rows -- the ping-pong producer's forms of the same epilogues
def test_skip_connection_layernorm_by_the_algebra_matches_reference_golden(lib, dev, name):
    """Option skip_z (default ON; blocks.py:124-128): the out-blocks' LayerNorm over [x | skip] in front of skip_linear by the LayerNorm algebra -- the in-block that produces a skip
    keeps its partial statistics and writes bf16(skip g[D:]) into the right half of its out-block's operand (k_gemm_ks COPY2), the MLP-out projection in front of the out-block runs
    un-split, skip_linear finishes the LayerNorm in its epilogue (ZIN) -- against skip_z = 0 (split-K slabs + the row kernel on that edge): both inside the gates of the reference's
    own outputs, one launch less per out-block.  With ControlNet residuals (xs_cn: skip + residual changes the statistics) the row-kernel path must run: same launch count, same bits."""
    cfg, sd, inp, kw, g, meta = golden_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    t = meta['timesteps'][0]
    ref = g[f'pred_t{t}']
    outs, launches = {}, {}
    try:
        for v in (1, 0):
            assert lib.ezdit_set_option(m._h, b'skip_z', v) == 0
            outs[v] = _forward(m, inp, t, kw).cpu().numpy()
            launches[v] = m.last_launch_count
    finally:
        assert lib.ezdit_set_option(m._h, b'skip_z', 1) == 0
    if 'controlnet_skips' in kw:
        assert launches[1] == launches[0]
        np.testing.assert_array_equal(outs[1], outs[0])
    else:
        assert launches[1] == launches[0] - cfg['depth'] // 2, launches
        assert rel_l2(outs[1], outs[0]) < 1e-2
    for v in (1, 0):
        r, a = rel_l2(outs[v], ref), float(np.abs(outs[v] - ref).max())
        record(f'{name} t={t} skip_z={v}: rel-L2 {r:.3e} max-abs {a:.3e}; launches {launches[v]}')
        assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48), (name, v, r, a)


def test_skip_path_is_not_taken_on_tables_that_were_not_built(lib, dev):
    """The static G' / C' tables of the skip path are built by ezdit_prepare_timesteps when skip_z is on.  A C-ABI caller that prepares with skip_z = 0, switches the option on and
    calls ezdit_forward WITHOUT preparing again must get the row-kernel path (same bits, same launch count), not a LayerNorm finished on tables that do not exist."""
    cfg, sd, inp, kw, g, meta = golden_case('s')
    m = get_model('s', meta['seed_w'])
    t = meta['timesteps'][0]
    try:
        assert lib.ezdit_set_option(m._h, b'skip_z', 0) == 0
        want = _forward(m, inp, t, kw)                       # binds, prepares context and timesteps with skip_z off
        n_off = m.last_launch_count
        x = m._keep[0]
        assert lib.ezdit_set_option(m._h, b'skip_z', 1) == 0
        got = torch.empty_like(want)
        assert lib.ezdit_forward(m._h, x.data_ptr(), x.shape[1], x.shape[0], None, None, None, 0, got.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert m.last_launch_count == n_off
        assert torch.equal(got, want)
        again = _forward(m, inp, t, kw)                      # prepared again with the option on: the skip path runs
        assert m.last_launch_count == n_off - cfg['depth'] // 2
        assert rel_l2(again.cpu().numpy(), want.cpu().numpy()) < 1e-2
    finally:
        assert lib.ezdit_set_option(m._h, b'skip_z', 1) == 0


@pytest.mark.parametrize('size,n_valid,act,L', [('s', (9, 1), (0, 1), 100), ('s', (1, 9), (1, 2), 100), ('s64', (1, 1), (0, 0), 100), ('s', (9, 1, 5), None, 100), ('s', (9, 4, 1, 1), (0, 2), 100),
                                                ('xs', (1, 6, 20, 1), (1, 3), 100),
                                                # > 2048 token rows: the ping-pong kernel's 128 x 144 producer in its DUAL form (BASELINE config #4's layout: P cond rows, then P uncond rows)
                                                ('s', (12, 20, 5, 17, 1, 1, 1, 1), (0, 4), 300), ('s64', (1, 1, 1, 9, 4, 20, 7, 2), (3, 8), 290), ('s', (9, 1, 5, 1, 1, 1, 7, 1), None, 300)])
def test_single_key_cross_attention_shortcut_matches_the_general_path(lib, dev, size, n_valid, act, L):
    """Option xkey1 (default ON): a batch element whose context mask has ONE valid key gets its cross-attention block as the constant
    W_o v_key + b_o, added by the attention-out projection (k_gemm_ks DUAL), and cross-attention + cross-out run over the remaining contiguous batch
    range `act` only (None: the multi-key elements are not contiguous -> the general path runs, bitwise).  Judge: the numpy oracle (which does the
    masked softmax like the reference); the general path (xkey1 = 0, every row through k_attn) must agree with the shortcut far inside the gate."""
    from oracle.dit import DiTOracle
    cfg = model_config(size)
    sd = make_state_dict(cfg, 77)
    B, Lc = len(n_valid), 20
    inp = make_inputs(cfg, B=B, L=L, Lc=Lc, n_valid=n_valid, seed=23)
    for e, nv in enumerate(n_valid):   # the single valid key need not be key 0: move it to position 3 + e for every other single-key element
        if nv == 1 and e % 2 == 1:
            inp['ctx_mask'][e] = np.roll(inp['ctx_mask'][e], 3 + e)
    m = get_model(size, 77)
    ref, _ = DiTOracle(cfg, sd).forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    outs, launches = {}, {}
    try:
        for v in (1, 0):
            assert lib.ezdit_set_option(m._h, b'xkey1', v) == 0
            outs[v] = _forward(m, inp, 499, {}).cpu().numpy()
            launches[v] = m.last_launch_count
    finally:
        assert lib.ezdit_set_option(m._h, b'xkey1', 1) == 0
    nblk = cfg['depth'] + 1
    if act is None:
        np.testing.assert_array_equal(outs[1], outs[0])
        assert launches[1] == launches[0]
    else:
        assert launches[1] == launches[0] - (2 * nblk if act[0] == act[1] else 0)   # nothing left for cross-attention: both launches of every block go
        assert rel_l2(outs[1], outs[0]) < 4e-3
    for v in (1, 0):
        r, a = rel_l2(outs[v], ref), float(np.abs(outs[v] - ref).max())
        record(f'{size} L={L} n_valid={n_valid} xkey1={v}: rel-L2 {r:.3e} max-abs {a:.3e}; launches {launches[v]}')
        assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48)


@pytest.mark.parametrize('size,L', [('s', 77), ('s64', 131), ('s', 1), ('s', 1500)])
def test_forward_odd_lengths_against_oracle(lib, dev, size, L):
    """Latent lengths that are odd / not a multiple of any tile (editing crops arbitrarily): the fused QKV epilogue takes its
    scalar V^T path, batch boundaries fall inside row tiles, attention pads keys.  No golden exists, so the oracle is the judge."""
    from oracle.dit import DiTOracle
    cfg = model_config(size)
    sd = make_state_dict(cfg, 1234)
    inp = make_inputs(cfg, B=2, L=L, Lc=20, n_valid=(9, 1), seed=17)
    m = get_model(size, 1234)
    pred = _forward(m, inp, 499, {}).cpu().numpy()
    o = DiTOracle(cfg, sd)
    ref, _ = o.forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    assert pred.shape == ref.shape == (2, cfg['out_chans'], L)
    r, a = rel_l2(pred, ref), float(np.abs(pred - ref).max())
    record(f'{size} L={L}: rel-L2 {r:.3e} max-abs {a:.3e}')
    assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48)


@pytest.mark.parametrize('heads,dh', [(3, 72), (5, 64)])
def test_forward_with_an_odd_head_count_takes_the_unfused_qkv_path(lib, dev, heads, dh):
    """No tile of the fused QKV GEMMs holds whole heads when the head count is odd (ping-pong: two heads per tile, lockstep: four): the step then
    runs the fp32 q|k|v projection + k_headnorm and normalises the cross-attention q inside k_attn (qkv_mode 0 in csrc/api.hip) -- a path no shipped
    config takes and that lost its option in round 5, so it gets its own fixture here.  Judge: the numpy oracle."""
    from ezaudio_amd import MaskDiT
    from oracle.dit import DiTOracle
    cfg = dict(model_config('xs'), embed_dim=heads * dh, num_heads=heads)
    sd = make_state_dict(cfg, 5)
    inp = make_inputs(cfg, B=2, L=90, Lc=20, n_valid=(11, 1), seed=29)
    m = MaskDiT(device='cuda:0', **cfg)
    m.load_state_dict(sd)
    pred = _forward(m, inp, 499, {}).cpu().numpy()
    ref, _ = DiTOracle(cfg, sd).forward(inp['x'], 499, inp['ctx'], inp['ctx_mask'])
    r, a = rel_l2(pred, ref), float(np.abs(pred - ref).max())
    record(f'odd heads H={heads} dh={dh}: rel-L2 {r:.3e} max-abs {a:.3e}; launches {m.last_launch_count}')
    assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48)
    nblk = cfg['depth'] + 1
    assert m.last_launch_count > nblk * 11   # the unfused path really ran: projection + head-norm launches on top of the split-K + row-kernel edges


def test_forward_per_row_timesteps_and_determinism(lib, dev):
    cfg, sd, inp, kw, g, meta = golden_case('xs')
    m = get_model('xs', meta['seed_w'])
    x, ctx, msk = t_(inp['x']), t_(inp['ctx']), t_(inp['ctx_mask'])
    a = m(x, torch.tensor(499), ctx, context_mask=msk)[0]
    b = m(x, torch.tensor([499, 499]), ctx, context_mask=msk)[0]
    assert torch.equal(a, b)
    c = m(x, torch.tensor([499, 19]), ctx, context_mask=msk)[0]
    d = m(x, torch.tensor(19), ctx, context_mask=msk)[0]
    assert torch.equal(c[0], a[0]) and torch.equal(c[1], d[1])   # rows never interact
    assert torch.equal(m(x, torch.tensor(499), ctx, context_mask=msk)[0], a)  # bitwise repeatable


@pytest.mark.parametrize('name', ['xl', 'xl_b8'])
def test_forward_is_bit_reproducible_at_full_width(lib, dev, name):
    """Twenty forwards of the XL shape (one prompt: k_gemm_ks producers; four prompts: the ping-pong producer) on the same inputs give the same bits.  A kernel that reads a
    register an inline-asm load has not filled yet, or a K tile a counted wait did not cover, shows up as a different result once in ten runs, only at full width and
    under register pressure (round 6: the COPY2 form's gain vector; tools/diag_determinism.py finds the launch)."""
    cfg, sd, inp, kw, g, meta = golden_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    t = meta['timesteps'][0]
    first = _forward(m, inp, t, kw).cpu().numpy()
    for i in range(19):
        again = _forward(m, inp, t, kw).cpu().numpy()
        assert np.array_equal(first, again), (name, i)


def test_forward_input_validation(lib, dev):
    m = get_model('xs', 1)
    cfg = model_config('xs')
    inp = make_inputs(cfg, B=2, L=64, Lc=20, seed=3)
    with pytest.raises(AssertionError):
        m(t_(inp['x']), torch.tensor(5), t_(inp['ctx'][:1]), context_mask=t_(inp['ctx_mask'][:1]))
    with pytest.raises(NotImplementedError):
        m(t_(inp['x']), torch.tensor(5), t_(inp['ctx']), context_mask=t_(inp['ctx_mask']), gt=t_(inp['x']))


# ---------------------------------------------------------------------------------------------------
# sampler level
# ---------------------------------------------------------------------------------------------------
def _run_sampler(m, inp, init, noises, meta, use_graph=True, P=1):
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    steps = meta['steps']
    text, tm = t_(inp['ctx'][0:1]).repeat(P, 1, 1), t_(inp['ctx_mask'][0:1]).repeat(P, 1)
    un, um = t_(inp['ctx'][1:2]).repeat(P, 1, 1), t_(inp['ctx_mask'][1:2]).repeat(P, 1)
    gt = t_(inp['gt'][0:1]).repeat(P, 1, 1) if meta['with_gt'] else None
    gm = t_(inp['gt_mask'][0:1]).repeat(P, 1, 1) if meta['with_gt'] else None
    sn = torch.stack([t_(z) for z in noises], 0).repeat(1, P, 1, 1) if meta['eta'] > 0 else None
    smp.prepare(text, tm, un, um, t_(init).repeat(P, 1, 1), sn, meta['guidance_scale'], meta['guidance_rescale'], steps,
                meta['eta'], gt=gt, gt_mask=gm)
    smp.run(use_graph=use_graph)
    lat = smp.finish()
    torch.cuda.synchronize()
    lat = lat.clone()
    if meta['with_gt']:
        lat = torch.where(gm, lat, gt)  # src/inference.py:104-105
    return lat


@pytest.mark.parametrize('name,tol', [('smp_xs', 2e-2), ('smp_xs_e0', 2e-2), ('smp_s', 2e-2), ('smp_l', 2e-2), ('smp_l_edit', 2e-2), ('smp_xl', 2e-2)])
def test_sampler_matches_reference_loop_golden(lib, dev, name, tol):
    """Final latent of the reference's own unmodified inference() (fp32, 20-50 steps) vs the HIP sampler (bf16 denoiser).
    smp_l / smp_xl are BASELINE.json configs #2 / #3 (the shipped L and XL architectures, 50 steps, 10 s latent, guidance 5,
    rescale 0.75, eta 1); smp_l_edit is the editing loop (gt + mask through every step, inference.py:79-86,103-104) at L width, 6 s latent.  bf16 error compounds over the trajectory: measured 7.0e-3 (xs) / 8.3e-3 (s) in round 1, gate 2e-2
    (the per-step gate above is 2e-2 as well)."""
    cfg, sd, inp, init, noises, g, meta = sampler_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    lat = _run_sampler(m, inp, init, noises, meta).cpu().numpy()
    r = rel_l2(lat, g['latent'])
    record(f'{name}: final-latent rel-L2 {r:.3e}')
    assert np.isfinite(lat).all() and r < tol


def test_sampler_graph_equals_eager_and_batch_equals_single(lib, dev):
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    m = get_model('xs', meta['seed_w'])
    a = _run_sampler(m, inp, init, noises, meta, use_graph=True)
    b = _run_sampler(m, inp, init, noises, meta, use_graph=False)
    assert torch.equal(a, b)
    c = _run_sampler(m, inp, init, noises, meta, use_graph=True, P=3)
    for i in range(3):
        assert torch.equal(c[i:i + 1], a)   # samples never interact -> sharding-invariant, bitwise


def test_context_with_another_single_key_pattern_drops_the_captured_step(lib, dev):
    """the captured step bakes the single-key layout (DUAL attention-out rows, cross-attention batch sub-range) into its launch
    structure.  A C-ABI caller that re-prepares the context with ANOTHER pattern -- here: cond | uncond swapped, so the single-key element moves from batch
    row 1 to row 0 -- then rewinds with ezdit_set_step(0) and replays with use_graph = 1 must get the graph of the NEW pattern: bitwise the eager result."""
    import ctypes as C
    from ezaudio_amd import _lib
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    m = get_model('xs', meta['seed_w'])
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    steps = 4
    text, tm = t_(inp['ctx'][0:1]), t_(inp['ctx_mask'][0:1])
    un, um = t_(inp['ctx'][1:2]), t_(inp['ctx_mask'][1:2])
    assert int(tm.sum()) > 1 and int(um.sum()) == 1
    sn = torch.stack([t_(z) for z in noises[:steps]], 0)
    st = C.c_void_p(smp.stream.cuda_stream)

    def run(use_graph):
        with torch.cuda.stream(smp.stream):
            _lib.check(lib.ezdit_sampler_run(m._h, steps, 1 if use_graph else 0, st))
        smp.stream.synchronize()
        return smp.latents.clone()

    smp.prepare(text, tm, un, um, t_(init), sn, 5.0, 0.75, steps, 1.0)
    first = run(True)                                   # captures the step for the layout [multi-key | single-key]
    outs = {}
    for use_graph in (True, False):
        with torch.cuda.stream(smp.stream):
            m.prepare_context(torch.cat([un, text], 0), torch.cat([um, tm], 0))   # [single-key | multi-key]: no sampler_begin in between
            _lib.check(lib.ezdit_set_step(m._h, 0, st))
        smp.latents.copy_(t_(init))
        outs[use_graph] = run(use_graph)
    assert torch.isfinite(outs[True]).all()
    assert torch.equal(outs[True], outs[False])
    assert not torch.equal(outs[True], first)           # the swapped pair really is another computation


def test_fused_cfg_ddim_step_against_oracle_loop(lib, dev):
    """Drive OUR denoiser from the oracle's restatement of the reference loop (B2 surface) and compare with the
    fully fused device loop: isolates CFG + rescale + DDIM (fp32 on both sides)."""
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    m = get_model('xs', meta['seed_w'])

    def denoise(x, t, ctx, msk, gt, gm):
        return m(t_(x), torch.tensor(t), t_(ctx), context_mask=t_(msk))[0].cpu().numpy()
    steps = 12
    meta2 = dict(meta, steps=steps)
    ref = oracle_sample(denoise, inp['ctx'][0:1], inp['ctx_mask'][0:1], inp['ctx'][1:2], inp['ctx_mask'][1:2], init, noises,
                        guidance_scale=meta['guidance_scale'], guidance_rescale=meta['guidance_rescale'], ddim_steps=50,
                        eta=meta['eta'], diff_params=DIFF, trace=(tr := []))
    del ref
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    smp.prepare(t_(inp['ctx'][0:1]), t_(inp['ctx_mask'][0:1]), t_(inp['ctx'][1:2]), t_(inp['ctx_mask'][1:2]), t_(init),
                torch.stack([t_(z) for z in noises], 0), meta['guidance_scale'], meta['guidance_rescale'], 50, meta['eta'])
    smp.run(steps)
    lat = smp.finish()
    torch.cuda.synchronize()
    # both loops call the SAME bf16 denoiser; fp32 rounding differences in CFG/DDIM (<1e-6) flip bf16 roundings inside the
    # next forward, so trajectories separate at the 1e-3 level after a dozen steps (measured 1.0e-3)
    assert rel_l2(lat.cpu().numpy(), tr[steps - 1]) < 5e-3, meta2


def test_reference_style_loop_drives_the_operator(lib, dev):
    """Surface B2 of SURVEY.md section 8b: the reference's loop body (src/inference.py:70-100), restated here line by line
    with OUR denoiser called once per step as `unet(...)` and OUR DDIMScheduler.step as the scheduler -- what a maintainer
    gets by swapping only the two objects -- against the reference's own final latent (smp_xs)."""
    from ezaudio_amd.scheduler import DDIMScheduler
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    unet = get_model('xs', meta['seed_w'])
    sched = DDIMScheduler(**DIFF)
    text, text_mask = t_(inp['ctx'][0:1]), t_(inp['ctx_mask'][0:1])
    uncond_text, uncond_mask = t_(inp['ctx'][1:2]), t_(inp['ctx_mask'][1:2])
    guidance_scale, guidance_rescale, eta = meta['guidance_scale'], meta['guidance_rescale'], meta['eta']
    sched.set_timesteps(meta['steps'])
    noise = t_(init)
    latents = noise
    for i, t in enumerate(sched.timesteps):
        latents = sched.scale_model_input(latents, t)
        latent_model_input = torch.cat([latents] * 2)                     # inference.py:75
        context = torch.cat([text, uncond_text], dim=0)
        context_mask = torch.cat([text_mask, uncond_mask], dim=0)
        noise_pred, _ = unet(latent_model_input, t, context, context_mask=context_mask, cls_token=None, gt=None,
                             mae_mask_infer=None)                         # inference.py:82-86
        noise_pred_text, noise_pred_uncond = noise_pred.chunk(2)
        out = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
        std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)   # rescale_noise_cfg, :12-23
        std_cfg = out.std(dim=list(range(1, out.ndim)), keepdim=True)
        out = guidance_rescale * (out * (std_text / std_cfg)) + (1 - guidance_rescale) * out
        latents = sched.step(model_output=out, timestep=t, sample=latents, eta=eta, variance_noise=t_(noises[i])).prev_sample
    torch.cuda.synchronize()
    r = rel_l2(latents.cpu().numpy(), g['latent'])
    record(f'reference-style loop on the drop-in operator: final-latent rel-L2 {r:.3e}')
    assert r < 2e-2


def test_sampler_run_rejects_steps_beyond_the_prepared_schedule(lib, dev):
    from ezaudio_amd import _lib
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    m = get_model('xs', meta['seed_w'])
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    smp = LatentSampler(m, DDIMScheduler(**DIFF))
    smp.prepare(t_(inp['ctx'][0:1]), t_(inp['ctx_mask'][0:1]), t_(inp['ctx'][1:2]), t_(inp['ctx_mask'][1:2]), t_(init),
                torch.stack([t_(z) for z in noises], 0), 5.0, 0.75, 50, 1.0)
    smp.run(30)
    with pytest.raises(_lib.EzditError):
        smp.run(30)          # 60 > 50 prepared steps: coefficient / modulation / noise tables would be read out of bounds
    smp.run(20)              # exactly to the end is fine
    with pytest.raises(_lib.EzditError):
        smp.run(1)
    torch.cuda.synchronize()
    assert torch.isfinite(smp.finish()).all()


def test_editing_with_one_reference_clip_shared_by_several_prompts(lib, dev):
    """P prompts editing the same clip: gt / gt_mask given once ([1, C, L]) must behave exactly like P copies."""
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs_e0')
    m = get_model('xs', meta['seed_w'])
    from ezaudio_amd.sampler import LatentSampler
    from ezaudio_amd.scheduler import DDIMScheduler
    P, steps = 3, meta['steps']
    text, tm = t_(inp['ctx'][0:1]).repeat(P, 1, 1), t_(inp['ctx_mask'][0:1]).repeat(P, 1)
    un, um = t_(inp['ctx'][1:2]).repeat(P, 1, 1), t_(inp['ctx_mask'][1:2]).repeat(P, 1)
    gt1, gm1 = t_(inp['gt'][0:1]), t_(inp['gt_mask'][0:1])
    outs = []
    for gt, gm in ((gt1, gm1), (gt1.repeat(P, 1, 1), gm1.repeat(P, 1, 1))):
        smp = LatentSampler(m, DDIMScheduler(**DIFF))
        smp.prepare(text, tm, un, um, t_(init).repeat(P, 1, 1), None, meta['guidance_scale'], meta['guidance_rescale'], steps,
                    meta['eta'], gt=gt, gt_mask=gm)
        smp.run()
        outs.append(smp.finish().clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    lat = torch.where(gm1, outs[0][0:1], gt1)
    assert rel_l2(lat.cpu().numpy(), g['latent']) < 2e-2
    with pytest.raises(ValueError):
        smp.prepare(text, tm, un, um, t_(init).repeat(P, 1, 1), None, 3.5, 0.0, steps, 0.0, gt=gt1.repeat(2, 1, 1), gt_mask=gm1.repeat(2, 1, 1))


DEFAULT_OPTS = dict(attn_xcd=1, row_variant=1, gemm_panel=3, row_affine=1, epi_lds=1, attn_xk2=1, gemm_pp=3, tile_partial=9, zfuse=1, wt=2, fuse_q2=1, q2_pp=1, xkey1=1, geglu_co=0, qkv_co=1, attn_qtile=0, skip_z=1)


@pytest.mark.parametrize('opt,values', [('attn_qtile', (64, 32)), ('attn_xcd', (0, 1)), ('row_variant', (0, 1)), ('gemm_panel', (0, 7)), ('row_affine', (0, 1)),
                                        ('epi_lds', (0, 1)), ('attn_xk2', (0, 1)), ('gemm_pp', (0, 3)), ('geglu_co', (0, 2)), ('qkv_co', (0, 2)), ('tile_partial', (9, 62)), ('zfuse', (1, 0)), ('wt', (0, 1)),
                                        ('fuse_q2', (1, 0)), ('q2_pp', (1, 0)), ('xkey1', (1, 0)), ('skip_z', (1, 0))])
def test_placement_and_row_kernel_variants_agree(lib, dev, opt, values):
    """attn_xcd only moves workgroups between XCDs (bitwise identical); row_variant changes the summation tree of the
    LayerNorm statistics: fp32 rounding only, but a last-bit change of a statistic flips bf16 roundings of the GEMM operands
    downstream, so two variants sit as far apart as either sits from the fp32 reference at bf16 resolution (measured 4.6e-3
    after 5 blocks); the judge is the reference golden."""
    cfg, sd, inp, kw, g, meta = golden_case('s')
    m = get_model('s', meta['seed_w'])
    outs = []
    for v in values:
        assert lib.ezdit_set_option(m._h, opt.encode(), v) == 0
        outs.append(_forward(m, inp, 499, kw).cpu().numpy())
    assert lib.ezdit_set_option(m._h, opt.encode(), DEFAULT_OPTS[opt]) == 0   # shipped defaults
    if opt == 'q2_pp':   # only consulted when the q projection is not fused into k_attn
        assert lib.ezdit_set_option(m._h, b'fuse_q2', 0) == 0
        outs = []
        for v in values:
            assert lib.ezdit_set_option(m._h, opt.encode(), v) == 0
            outs.append(_forward(m, inp, 499, kw).cpu().numpy())
        assert lib.ezdit_set_option(m._h, opt.encode(), DEFAULT_OPTS[opt]) == 0
        assert lib.ezdit_set_option(m._h, b'fuse_q2', 1) == 0
    if opt not in ('row_variant', 'gemm_pp', 'tile_partial', 'zfuse', 'fuse_q2', 'q2_pp', 'xkey1', 'geglu_co', 'qkv_co', 'attn_qtile', 'skip_z'):   # placement / issue order / launch structure only: bitwise identical
        for o in outs[1:]:
            np.testing.assert_array_equal(outs[0], o)
    else:   # row_variant: same math, different rounding points; gemm_pp / tile_partial: another kernel (other MFMA shape, other fp32 summation order over K)
        assert rel_l2(outs[0], outs[1]) < 1e-2
        for o in outs:
            assert rel_l2(o, g['pred_t499']) < REL_TOL


@pytest.mark.parametrize('name', ['xs64', 's64', 'xl', 'xl_b8'])
def test_co_resident_gemm_kernels_match_reference_golden(lib, dev, name):
    """geglu_co = qkv_co = 2: the GEGLU GEMM and the fused QKV GEMM on the 4-wave co-resident kernel (csrc/gemm_co.h; the QKV epilogue in registers without the
    k-split exchange, head_dim 64 and 72) against the reference goldens, at the gate of the default path."""
    cfg, sd, inp, kw, g, meta = golden_case(name)
    m = get_model(meta['size'], meta['seed_w'])
    try:
        for o in (b'geglu_co', b'qkv_co'):
            assert lib.ezdit_set_option(m._h, o, 2) == 0
        for t in meta['timesteps']:
            ref = g[f'pred_t{t}']
            pred = _forward(m, inp, t, kw).cpu().numpy()
            r, a = rel_l2(pred, ref), float(np.abs(pred - ref).max())
            record(f'{name} t={t} co-resident GEGLU + QKV: rel-L2 {r:.3e} max-abs {a:.3e}')
            assert r < REL_TOL and a < ABS_TOL * max(1.0, float(ref.std()) / 1.48)
    finally:
        for o in (b'geglu_co', b'qkv_co'):
            assert lib.ezdit_set_option(m._h, o, DEFAULT_OPTS[o.decode()]) == 0


def test_unsupported_kernel_configuration_is_an_error_not_a_silent_skip(lib, dev):
    from ezaudio_amd import _lib
    A = torch.zeros(128, 64, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(128, 128, device=dev)
    rc = lib.ezdit_test_gemm(None, 99 * 4, A.data_ptr(), 64, A.data_ptr(), 64, None, out.data_ptr(), 128, 128, 128, 64, 1, None)
    assert rc != 0 and lib.ezdit_last_error()
    rc = lib.ezdit_test_gemm(None, 0, A.data_ptr(), 64, A.data_ptr(), 64, None, out.data_ptr(), 128, 128, 128, 100, 1, None)
    assert rc != 0   # K not a multiple of 64


# ---------------------------------------------------------------------------------------------------
# stand-alone CFG + rescale + DDIM operator (SURVEY.md section 8b minimum export set) and the scheduler.step() that uses it
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('gs,gr,eta', [(5.0, 0.75, 1.0), (3.5, 0.0, 0.0), (0.0, 0.0, 1.0)])
def test_cfg_ddim_step_operator_against_oracle(lib, dev, gs, gr, eta):
    import ctypes as C
    from ezaudio_amd import DDIMScheduler, _lib
    from oracle.sampler import cfg_combine, rescale_noise_cfg
    from oracle.ddim import DDIMOracle
    P, Cc, L = 3, 128, 500
    n = Cc * L
    g = torch.Generator().manual_seed(5)
    rows = 2 * P if gs > 0 else P
    pred = torch.randn(rows, Cc, L, generator=g) * 1.3
    lat = torch.randn(P, Cc, L, generator=g)
    noise = torch.randn(P, Cc, L, generator=g)
    sch = DDIMScheduler(**DIFF)
    sch.set_timesteps(50)
    t = int(sch.timesteps[7])
    coef = _lib.EzditDdimCoef(*sch._coef(t, eta))
    pd, ld, nd = pred.to(dev), lat.clone().to(dev), noise.to(dev)
    scratch = torch.zeros(P * 256, device=dev)
    rc = lib.ezdit_cfg_ddim_step(pd.data_ptr(), ld.data_ptr(), nd.data_ptr() if eta > 0 else None, C.byref(coef), gs, gr, P, n,
                                 scratch.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    o = DDIMOracle(**DIFF)
    o.set_timesteps(50)
    if gs > 0:
        v = cfg_combine(pred[:P].numpy(), pred[P:].numpy(), gs)
        if gr > 0:
            v = rescale_noise_cfg(v, pred[:P].numpy(), gr)
    else:
        v = pred.numpy()
    ref = o.step(v, t, lat.numpy(), eta=eta, noise=noise.numpy() if eta > 0 else None)
    assert rel_l2(ld.cpu().numpy(), ref) < 5e-6
    # DDIMScheduler.step on CUDA tensors goes through the same operator (the reference's loop calls it once per step)
    v_t = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(dev)
    out = sch.step(v_t, t, lat.to(dev), eta=eta, variance_noise=noise.to(dev) if eta > 0 else None).prev_sample
    assert rel_l2(out.cpu().numpy(), ref) < 5e-6
