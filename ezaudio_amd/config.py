"""Config surface: the reference's YAML files with the custom `!include` tag.

Mirrors /root/reference/src/utils/utils.py:7-17 (`load_yaml_with_includes`) and the model registry
dicts of api/ezaudio.py:20-28.  The `model:` key set of ckpts/ezaudio-{l,xl}.yml is the config
surface this package keeps; values outside the implemented combination raise NotImplementedError
(mirroring src/models/udit.py:83,113,127) instead of silently computing something else.
"""
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(_HERE, 'configs')

# model_name -> files, same names as the reference registry (api/ezaudio.py:20-28)
configs = {
    's3_xl': {'path': 'ckpts/s3/ezaudio_s3_xl.pt',
              'url': 'https://huggingface.co/OpenSound/EzAudio/resolve/main/ckpts/s3/ezaudio_s3_xl.pt',
              'config': os.path.join(CONFIG_DIR, 'ezaudio-xl.yml')},
    's3_l': {'path': 'ckpts/s3/ezaudio_s3_l.pt',
             'url': 'https://huggingface.co/OpenSound/EzAudio/resolve/main/ckpts/s3/ezaudio_s3_l.pt',
             'config': os.path.join(CONFIG_DIR, 'ezaudio-l.yml')},
    'vae': {'path': 'ckpts/vae/1m.pt',
            'url': 'https://huggingface.co/OpenSound/EzAudio/resolve/main/ckpts/vae/1m.pt'},
}


def load_yaml_with_includes(yaml_file):
    class _Loader(yaml.FullLoader):
        pass

    def _include(loader, node):
        path = os.path.join(os.path.dirname(yaml_file), loader.construct_scalar(node))
        with open(path, 'r') as f:
            return yaml.load(f, Loader=_Loader)

    _Loader.add_constructor('!include', _include)
    with open(yaml_file, 'r') as f:
        return yaml.load(f, Loader=_Loader)


# ControlNet registry: same keys, paths and URLs as api/controlnet.py:20-27
controlnet_configs = {
    'model': {'path': 'ckpts/s3/ezaudio_s3_l.pt',
              'url': 'https://huggingface.co/OpenSound/EzAudio/resolve/main/ckpts/s3/ezaudio_s3_l.pt'},
    'energy': {'path': 'ckpts/controlnet/s3_l_energy.pt',
               'url': 'https://huggingface.co/OpenSound/EzAudio/resolve/main/ckpts/controlnet/s3_l_energy.pt',
               'config': os.path.join(CONFIG_DIR, 'controlnet', 'energy_l.yml')},
    'vae': configs['vae'],
}


# the one combination of UDiT options the shipped checkpoints use (ckpts/ezaudio-xl.yml:3-36)
_REQUIRED = dict(input_type='1d', patch_size=1, qkv_bias=False, qk_scale=None, qk_norm='layernorm',
                 norm_layer='layernorm', act_layer='geglu', context_norm=True,
                 time_fusion='ada_sola_bias', cls_dim=None, context_fusion='cross',
                 context_pe_method='none', pe_method='none', rope_mode='shared', use_conv=True,
                 skip=True, skip_norm=True, mae=True)


def validate_model_config(cfg):
    """Raise NotImplementedError for any value the HIP path does not implement."""
    for k, want in _REQUIRED.items():
        if k in cfg and cfg[k] != want:
            raise NotImplementedError(f'model.{k}={cfg[k]!r} is not implemented (only {want!r})')
    if cfg.get('in_chans') != 2 * cfg.get('out_chans', 128) + 1:
        raise NotImplementedError('in_chans must be 2*out_chans+1 (MaskDiT concat of x, gt, mask)')
    return cfg
