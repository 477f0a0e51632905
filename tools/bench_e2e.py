"""Wall time of one EzAudio.generate_audio() call (XL, 10 s, random-init weights, stand-in T5) -- diagnostic, GPU only."""
import sys
import time

import torch

sys.path.insert(0, '.')
from ezaudio_amd import api as A                      # noqa: E402
from ezaudio_amd.config import configs, load_yaml_with_includes      # noqa: E402
from ezaudio_amd.vae import Autoencoder               # noqa: E402
from ezaudio_amd.weights import random_state_dict     # noqa: E402
from oracle import vae as V                           # noqa: E402  (synthetic VAE weights only)


class Tok:
    def __call__(self, texts, max_length, padding, truncation, return_tensors):
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        mask = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, t in enumerate(texts):
            n = max(1, min(max_length, len(t.split()) + 1))
            ids[i, :n] = torch.arange(1, n + 1)
            mask[i, :n] = 1
        return type('B', (), dict(input_ids=ids, attention_mask=mask))()


class Enc:
    def __init__(self, dim):
        self.table = torch.randn(128, dim, generator=torch.Generator().manual_seed(7)).cuda()

    def __call__(self, input_ids, attention_mask):
        return type('O', (), dict(last_hidden_state=self.table[input_ids % 128]))()


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
params = load_yaml_with_includes(configs['s3_xl']['config'])
cfg = params['model']
vcfg = dict(V.VAE_DEFAULT)
vsd = {k: torch.from_numpy(v) for k, v in V.make_vae_state_dict(vcfg, 6).items()}
common = dict(channels=128, c_mults=[1, 2, 4, 8], strides=[2, 4, 6, 10], use_snake=True)
vconf = {'model': {'decoder': {'type': 'oobleck', 'config': dict(out_channels=1, latent_dim=128, final_tanh=False, **common)}}}
ae = Autoencoder(config=vconf, state_dict=vsd)
ez = A.EzAudio('s3_xl', autoencoder=ae, tokenizer=Tok(), text_encoder=Enc(cfg['context_dim']), state_dict=random_state_dict(cfg, seed=0))
for i in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    sr, audio = ez.generate_audio('a dog barking in the rain', length=10, ddim_steps=steps, random_seed=1)
    torch.cuda.synchronize()
    print(f'generate_audio({steps} steps, 10 s): {1e3 * (time.perf_counter() - t):.1f} ms  -> {audio.shape[0] / sr:.1f} s of audio', flush=True)

# ---- stage breakdown of the same call (extra syncs, so the sum is slightly above the fused wall time) ----
from ezaudio_amd.sampler import LatentSampler, draw_noises   # noqa: E402


def tick(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f'  {label:34s} {1e3 * (t1 - t0):7.2f} ms', flush=True)
    return t1


for _ in range(2):
    print('breakdown:')
    t = time.perf_counter()
    tok = ez.tokenizer(['a dog barking in the rain'], max_length=params['text_encoder']['max_length'], padding='max_length', truncation=True,
                       return_tensors='pt')
    text = ez.text_encoder(input_ids=tok.input_ids.cuda(), attention_mask=tok.attention_mask.cuda()).last_hidden_state
    un = ez.tokenizer([''], max_length=params['text_encoder']['max_length'], padding='max_length', truncation=True, return_tensors='pt')
    utext = ez.text_encoder(input_ids=un.input_ids.cuda(), attention_mask=un.attention_mask.cuda()).last_hidden_state
    t = tick('stand-in tokenizer + encoder', t)
    init, noises = draw_noises(128, 500, steps, 1.0, 1, 'cuda', 1)
    t = tick('draw init + per-step noise', t)
    smp = LatentSampler(ez.unet, ez.noise_scheduler)
    smp.prepare(text.float(), tok.attention_mask.cuda().bool(), utext.float(), un.attention_mask.cuda().bool(), init, noises, 5.0, 0.75,
                steps, 1.0)
    t = tick('prepare (context K/V, AdaLN tables)', t)
    smp.run(use_graph=True)
    lat = smp.finish()
    t = tick(f'{steps} denoising steps (+ graph capture)', t)
    wav = ez.autoencoder(embedding=lat)
    t = tick('VAE decode', t)
    _ = wav.cpu().numpy()
    t = tick('copy to host', t)
