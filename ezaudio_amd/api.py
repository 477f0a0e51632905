"""Public API, signature-compatible with the reference's ``api/ezaudio.py`` (EzAudio) -- the drop-in
surface B1 of SURVEY.md section 8b.  Everything on the denoising path runs in libezaudio_hip.so; T5
(``transformers``) and the VAE are pre/post models outside this round's scope, so
they can be injected; when not injected, T5 is loaded the way the reference loads it.
"""
import random
import sys
import urllib.request
from pathlib import Path

import numpy as np
import torch

from .config import configs, controlnet_configs, load_yaml_with_includes
from .denoiser import MaskDiT
from .sampler import inference, inference_controlnet
from .scheduler import DDIMScheduler

MAX_SEED = np.iinfo(np.int32).max


class EzAudio:
    def __init__(self, model_name, ckpt_path=None, vae_path=None, device='cuda',
                 autoencoder=None, tokenizer=None, text_encoder=None, state_dict=None):
        self.device = device
        config_name = configs[model_name]['config']
        if ckpt_path is None and state_dict is None:
            ckpt_path = self.download_ckpt(configs[model_name])
        if vae_path is None and autoencoder is None:
            vae_path = self.download_ckpt(configs['vae'])
        (self.autoencoder, self.unet, self.tokenizer, self.text_encoder, self.noise_scheduler,
         self.params) = self.load_models(config_name, ckpt_path, vae_path, device, autoencoder, tokenizer,
                                         text_encoder, state_dict)

    def download_ckpt(self, model_dict):
        """api/ezaudio.py:44-65."""
        local_path = Path(model_dict['path'])
        url = model_dict['url']
        local_path.parent.mkdir(parents=True, exist_ok=True)
        if not local_path.exists() and url:
            print(f"Downloading from {url} to {local_path}...")

            def progress_bar(block_num, block_size, total_size):
                sys.stdout.write(f"\rProgress: {block_num * block_size / total_size * 100:.2f}%")
                sys.stdout.flush()
            try:
                urllib.request.urlretrieve(url, local_path, reporthook=progress_bar)
                print(f"Downloaded checkpoint to {local_path}")
            except Exception as e:
                # the reference prints and continues (api/ezaudio.py:61-62), which only moves the failure to a confusing
                # torch.load on a missing file: fail here, naming the URL
                raise RuntimeError(f'could not download {url} to {local_path}: {e}') from e
        else:
            print(f"Checkpoint already exists at {local_path}")
        return local_path

    def load_models(self, config_name, ckpt_path, vae_path, device, autoencoder=None, tokenizer=None,
                    text_encoder=None, state_dict=None):
        """api/ezaudio.py:68-99."""
        params = load_yaml_with_includes(config_name)
        if autoencoder is None:   # api/ezaudio.py:75-79; an injected callable with the same surface is also accepted
            from .vae import Autoencoder
            autoencoder = Autoencoder(ckpt_path=vae_path, model_type=params['autoencoder']['name'],
                                      quantization_first=params['autoencoder']['q_first'], device=device)
        if tokenizer is None or text_encoder is None:
            from transformers import T5EncoderModel, T5Tokenizer
            tokenizer = T5Tokenizer.from_pretrained(params['text_encoder']['model'])
            text_encoder = T5EncoderModel.from_pretrained(params['text_encoder']['model']).to(device)
            text_encoder.eval()
        unet = MaskDiT(device=device, **params['model'])
        if state_dict is None:
            state_dict = torch.load(ckpt_path, map_location='cpu')['model']
        unet.load_state_dict(state_dict)
        unet.eval()
        noise_scheduler = DDIMScheduler(**params['diff'])
        return autoencoder, unet, tokenizer, text_encoder, noise_scheduler, params

    def generate_audio(self, text, length=10, guidance_scale=5, guidance_rescale=0.75, ddim_steps=100, eta=1,
                       random_seed=None, randomize_seed=False):
        """api/ezaudio.py:101-130.  `text` may also be a list of prompts (batched extension): the result is then
        an array [N, T]."""
        neg_text = None
        length = length * self.params['autoencoder']['latent_sr']
        gt, gt_mask = None, None
        if text == '':
            guidance_scale = None
            print('empty input')
        if randomize_seed:
            random_seed = random.randint(0, MAX_SEED)
        pred = inference(self.autoencoder, self.unet, gt, gt_mask, self.tokenizer, self.text_encoder, self.params,
                         self.noise_scheduler, text, neg_text, length, guidance_scale, guidance_rescale, ddim_steps,
                         eta, random_seed, self.device)
        pred = pred.cpu().numpy()
        pred = pred.squeeze(0).squeeze(0) if isinstance(text, str) or len(text) == 1 else pred.squeeze(1)
        return self.params['autoencoder']['sr'], pred

    def editing_audio(self, text, boundary, gt_file, mask_start, mask_length, guidance_scale=3.5, guidance_rescale=0,
                      ddim_steps=100, eta=1, random_seed=None, randomize_seed=False):
        """api/ezaudio.py:132-207 (crop / pad / mask bookkeeping on the host, sampling on the GPU)."""
        import librosa
        neg_text = None
        if text == '':
            guidance_scale = None
            print('empty input')
        sr = self.params['autoencoder']['sr']
        latent_sr = self.params['autoencoder']['latent_sr']
        mask_end = mask_start + mask_length
        gt, _ = librosa.load(gt_file, sr=sr)
        gt = gt / (np.max(np.abs(gt)) + 1e-9)
        audio_length = len(gt) / sr
        mask_start = min(mask_start, audio_length)
        if mask_end > audio_length:  # out-padding mode
            gt = np.pad(gt, (0, round((mask_end - audio_length) * sr)), 'constant')
            audio_length = len(gt) / sr
        output_audio = gt.copy()
        gt = torch.tensor(gt).unsqueeze(0).unsqueeze(1).to(self.device)
        boundary = min((mask_end - mask_start) / 2, boundary)
        start_idx = max(mask_start - boundary, 0)
        end_idx = min(mask_end + boundary, audio_length)
        mask_start -= start_idx
        mask_end -= start_idx
        gt = gt[:, :, round(start_idx * sr):round(end_idx * sr)]
        gt_latent = self.autoencoder(audio=gt)
        B, D, L = gt_latent.shape
        gt_mask = torch.zeros(B, D, L).to(self.device)
        gt_mask[:, :, round(mask_start * latent_sr): round(mask_end * latent_sr)] = 1
        gt_mask = gt_mask.bool()
        if randomize_seed:
            random_seed = random.randint(0, MAX_SEED)
        pred = inference(self.autoencoder, self.unet, gt_latent, gt_mask, self.tokenizer, self.text_encoder,
                         self.params, self.noise_scheduler, text, neg_text, L, guidance_scale, guidance_rescale,
                         ddim_steps, eta, random_seed, self.device)
        pred = pred.cpu().numpy().squeeze(0).squeeze(0)
        chunk_length = end_idx - start_idx
        pred = pred[:round(chunk_length * sr)]
        output_audio[round(start_idx * sr):round(end_idx * sr)] = pred
        return sr, output_audio


class EzAudio_ControlNet(EzAudio):
    """api/controlnet.py:31-160: EzAudio-L + energy ControlNet (``model_name='energy'``)."""

    def __init__(self, model_name, ckpt_path=None, controlnet_path=None, vae_path=None, device='cuda', autoencoder=None,
                 tokenizer=None, text_encoder=None, state_dict=None, controlnet_state_dict=None):
        self.device = device
        config_name = controlnet_configs[model_name]['config']
        if ckpt_path is None and state_dict is None:
            ckpt_path = self.download_ckpt(controlnet_configs['model'])
        if controlnet_path is None and controlnet_state_dict is None:
            controlnet_path = self.download_ckpt(controlnet_configs[model_name])
        if vae_path is None and autoencoder is None:
            vae_path = self.download_ckpt(controlnet_configs['vae'])
        (self.autoencoder, self.unet, self.tokenizer, self.text_encoder, self.noise_scheduler,
         self.params) = self.load_models(config_name, ckpt_path, vae_path, device, autoencoder, tokenizer, text_encoder,
                                         state_dict)
        from .conditions import Conditioner
        from .controlnet import DiTControlNet
        cfg = self.params['model'].copy()
        cfg.update(self.params['controlnet'])                      # api/controlnet.py:92-95
        self.controlnet = DiTControlNet(device=device, **cfg)
        if controlnet_state_dict is None:
            controlnet_state_dict = torch.load(controlnet_path, map_location='cpu')['model']
        self.controlnet.load_state_dict(controlnet_state_dict)
        self.conditioner = Conditioner(**self.params['conditioner'])

    def generate_audio(self, text, audio_path, surpass_noise=0, guidance_scale=3.5, guidance_rescale=0, ddim_steps=50,
                       eta=1, conditioning_scale=1, random_seed=None, randomize_seed=False):
        """api/controlnet.py:113-160: the control curve is the frame energy of a reference recording."""
        import librosa
        sr = self.params['autoencoder']['sr']
        gt, _ = librosa.load(audio_path, sr=sr)
        gt = gt / (np.max(np.abs(gt)) + 1e-9)
        if surpass_noise > 0:
            gt[np.abs(gt) <= surpass_noise] = 0
        original_length = len(gt)
        num_samples = int(10 * sr)
        audio_frames = round(num_samples / sr * self.params['autoencoder']['latent_sr'])
        gt = np.pad(gt, (0, num_samples - len(gt)), 'constant') if len(gt) < num_samples else gt[:num_samples]
        gt_audio = torch.tensor(gt).unsqueeze(0).unsqueeze(1).to(self.device)
        latent_shape = (1, self.params['autoencoder']['dim'], audio_frames)   # the reference encodes only to get this shape
        condition = self.conditioner(gt_audio.squeeze(1), latent_shape)
        if randomize_seed:
            random_seed = random.randint(0, MAX_SEED)
        pred = inference_controlnet(self.autoencoder, self.unet, self.controlnet, None, None, condition, self.tokenizer,
                                    self.text_encoder, self.params, self.noise_scheduler, text, neg_text=None,
                                    audio_frames=audio_frames, guidance_scale=guidance_scale,
                                    guidance_rescale=guidance_rescale, ddim_steps=ddim_steps, eta=eta,
                                    random_seed=random_seed, conditioning_scale=conditioning_scale, device=self.device)
        pred = pred.cpu().numpy().squeeze(0).squeeze(0)[:original_length]
        return sr, pred
