"""Drop-in for the reference's ``DiTControlNet`` (/root/reference/src/models/controlnet.py:87-315): same constructor
kwargs (``params['model']`` updated with ``params['controlnet']``, api/controlnet.py:92-95), same call surface, same
state-dict keys; the forward runs in libezaudio_hip.so with the same kernels as the backbone.
"""
import ctypes as C

import torch

from . import _lib
from .config import validate_model_config
from .denoiser import _ptr, _stream
from .weights import pack_state_dict


class DiTControlNet:
    def __init__(self, cond_in=None, cond_blocks=None, cond_mask=False, cond_mask_prob=None, cond_mask_ratio=None,
                 cond_mask_span=None, device='cuda', max_len=2048, **kwargs):
        cfg = dict(kwargs)
        validate_model_config({k: v for k, v in cfg.items() if k != 'mae'})
        if cond_in is None or not cond_blocks or len(cond_blocks) != 2:
            raise NotImplementedError('ControlNet embed with cond_blocks of length 2 only (ckpts/controlnet/energy_l.yml)')
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = _lib.load()
        c = _lib.EzditConfig(cfg['embed_dim'], cfg['num_heads'], cfg['depth'], cfg['in_chans'], cfg['out_chans'],
                             cfg['context_dim'], cfg['ada_sola_rank'], float(cfg['ada_sola_alpha']), float(cfg['mlp_ratio']),
                             max_len, 1, int(cond_in), int(cond_blocks[0]), int(cond_blocks[1]), 1 if cond_mask else 0)
        self._h = C.c_void_p()
        _lib.check(self.lib.ezdit_create(C.byref(c), C.byref(self._h)))
        self.C, self.D, self.n_half = cfg['out_chans'], cfg['embed_dim'], cfg['depth'] // 2
        self.cond_in = int(cond_in)
        self._blob = self._ws = self._ws_key = None
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self.lib.ezdit_destroy(self._h)
        except Exception:
            pass

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def load_state_dict(self, state_dict, strict=True):
        self._blob = pack_state_dict(self._h, state_dict, strict=strict).to(self.device)
        _lib.check(self.lib.ezdit_bind_weights(self._h, _ptr(self._blob), self._blob.numel()))
        self._ws_key = None
        return self

    # same plumbing as MaskDiT
    def bind(self, B, L, Lc, n_slots):
        key = (B, L, Lc, n_slots)
        if self._ws_key == key:
            return
        if self._blob is None:
            raise _lib.EzditError('load_state_dict first')
        need = self.lib.ezdit_workspace_bytes(self._h, B, L, Lc, n_slots)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.ezdit_bind_workspace(self._h, _ptr(self._ws), self._ws.numel(), B, L, Lc, n_slots, _stream()))
        self._ws_key = key

    def prepare_context(self, context, context_mask):
        context = context.to(self.device, torch.float32).contiguous()
        mask = None if context_mask is None else context_mask.to(self.device).to(torch.uint8).contiguous()
        _lib.check(self.lib.ezdit_prepare_context(self._h, _ptr(context), _ptr(mask), _stream()))

    def prepare_timesteps(self, ts, per_row):
        arr = (C.c_int32 * len(ts))(*[int(t) for t in ts])
        _lib.check(self.lib.ezdit_prepare_timesteps(self._h, arr, len(ts), 1 if per_row else 0, _stream()))

    def prepare_condition(self, condition):
        cond = condition.to(self.device, torch.float32).contiguous()
        if cond.dim() != 3 or cond.shape[1] != self.cond_in:
            raise AssertionError(f'condition must be [B, {self.cond_in}, 2L], got {tuple(cond.shape)}')
        _lib.check(self.lib.ezdit_prepare_condition(self._h, _ptr(cond), cond.shape[2], _stream()))
        self._keep = [cond]

    def residual_views(self, B, L):
        """Zero-copy views of the residuals of the last forward: list of [B, L, D] fp32 (unscaled)."""
        arr = (C.c_void_p * self.n_half)()
        _lib.check(self.lib.ezdit_controlnet_residuals(self._h, arr, self.n_half))
        base = self._ws.data_ptr()
        out = []
        for i in range(self.n_half):
            off = arr[i] - base
            out.append(self._ws[off:off + B * L * self.D * 4].view(torch.float32).reshape(B, L, self.D))
        return out

    def forward(self, x, timesteps, context, x_mask=None, context_mask=None, cls_token=None, condition=None,
                cond_mask_infer=None, conditioning_scale=1.0):
        if x_mask is not None or cls_token is not None or cond_mask_infer is not None:
            raise NotImplementedError('x_mask / cls_token / cond_mask_infer are not used at inference')
        B, cin, L = x.shape
        if cin != 2 * self.C + 1:
            raise AssertionError(f'ControlNet takes the assembled {2 * self.C + 1}-channel input (forward_model=False), got {cin}')
        ts = torch.as_tensor(timesteps)
        per_row = ts.dim() > 0
        t_list = [int(v) for v in ts.reshape(-1).tolist()] if per_row else [int(ts)]
        self.bind(B, L, context.shape[1], max(len(t_list), 1))
        self.prepare_context(context, context_mask)
        self.prepare_timesteps(t_list, per_row)
        self.prepare_condition(condition)
        x = x.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.ezdit_controlnet_forward(self._h, _ptr(x), cin, B, None, None, None, _stream()))
        return [r * conditioning_scale for r in self.residual_views(B, L)]

    __call__ = forward
