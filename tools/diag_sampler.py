"""GPU diagnostic (not a pytest): step-by-step sampler trace vs the oracle loop for the xs fixture."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.dit import DiTOracle  # noqa: E402
from oracle.sampler import sample as oracle_sample  # noqa: E402
from tests.util import DIFF, sampler_case  # noqa: E402
from ezaudio_amd import MaskDiT  # noqa: E402
from ezaudio_amd.sampler import LatentSampler  # noqa: E402
from ezaudio_amd.scheduler import DDIMScheduler  # noqa: E402


def t_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main():
    cfg, sd, inp, init, noises, g, meta = sampler_case('smp_xs')
    o = DiTOracle(cfg, sd)
    m = MaskDiT(device='cuda', **cfg)
    m.load_state_dict(sd)
    for (gs, gr, eta) in ((5.0, 0.75, 1.0), (5.0, 0.0, 1.0), (5.0, 0.75, 0.0), (5.0, 0.0, 0.0)):
        tr = []
        oracle_sample(lambda x, t, c, k, a, b: o.forward(x, t, c, k)[0], inp['ctx'][0:1], inp['ctx_mask'][0:1], inp['ctx'][1:2],
                      inp['ctx_mask'][1:2], init, noises, guidance_scale=gs, guidance_rescale=gr, ddim_steps=50, eta=eta,
                      diff_params=DIFF, trace=tr)
        smp = LatentSampler(m, DDIMScheduler(**DIFF))
        sn = torch.stack([t_(z) for z in noises], 0) if eta > 0 else None
        smp.prepare(t_(inp['ctx'][0:1]), t_(inp['ctx_mask'][0:1]), t_(inp['ctx'][1:2]), t_(inp['ctx_mask'][1:2]), t_(init), sn,
                    gs, gr, 50, eta)
        print(f'--- guidance {gs} rescale {gr} eta {eta}')
        for i in range(50):
            smp.run(1, use_graph=False)
            lat = smp.finish()
            torch.cuda.synchronize()
            pred = m.debug_buffer('pred', torch.float32, (2, cfg['out_chans'], meta['L'])).cpu().numpy()
            l = lat.cpu().numpy()
            rel = float(np.linalg.norm(l - tr[i]) / np.linalg.norm(tr[i]))
            bad = not np.isfinite(l).all()
            if i < 6 or i % 10 == 9 or bad:
                print(f'step {i:2d}: lat std {l.std():.4f} (oracle {tr[i].std():.4f}) rel {rel:.3e} pred finite {np.isfinite(pred).all()} '
                      f'pred std {pred[0].std():.3f}/{pred[1].std():.3f} nan-count {int(np.isnan(l).sum())}')
            if bad:
                break


if __name__ == '__main__':
    main()
