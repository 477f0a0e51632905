"""Drop-in denoiser operator: same constructor kwargs and call surface as the reference's
``MaskDiT`` (/root/reference/src/models/conditioners.py:123-183) and, through ``.model``, ``UDiT.forward``
(src/models/udit.py:281-362) -- so the reference's own ``inference()`` can drive it unmodified
(SURVEY.md section 8b surface B2) -- but every FLOP runs in libezaudio_hip.so on gfx950.

PyTorch is used for device memory, streams and RNG only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import validate_model_config
from .weights import pack_state_dict


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _UDiTView:
    """``unet.model(x=x257, timesteps=..., context=..., context_mask=..., cls_token=None, controlnet_skips=...)``
    as called by src/inference_controlnet.py:97-99."""

    def __init__(self, owner):
        self._o = owner

    def __call__(self, x, timesteps, context, x_mask=None, context_mask=None, cls_token=None, controlnet_skips=None):
        return self._o._run(x, timesteps, context, context_mask, None, None, controlnet_skips, in_ch=x.shape[1])


class MaskDiT:
    def __init__(self, mae=False, mae_prob=0.5, mask_ratio=(0.25, 1.0), mask_span=10, device='cuda', max_len=2048,
                 **kwargs):
        cfg = dict(kwargs)
        cfg['mae'] = mae
        validate_model_config(cfg)
        if not mae:
            raise NotImplementedError('mae=False (no [x|gt|mask] concat) is not implemented')
        self.cfg = cfg
        self.mae, self.mae_prob, self.mask_ratio, self.mask_span = mae, mae_prob, mask_ratio, mask_span
        self.device = torch.device(device)
        self.lib = _lib.load()
        c = _lib.EzditConfig(cfg['embed_dim'], cfg['num_heads'], cfg['depth'], cfg['in_chans'], cfg['out_chans'],
                             cfg['context_dim'], cfg['ada_sola_rank'], float(cfg['ada_sola_alpha']),
                             float(cfg['mlp_ratio']), max_len)
        self._h = C.c_void_p()
        _lib.check(self.lib.ezdit_create(C.byref(c), C.byref(self._h)))
        self.C = cfg['out_chans']
        self.D = cfg['embed_dim']
        self.n_half = cfg['depth'] // 2
        self._blob = None
        self._ws = None
        self._ws_key = None
        self._mask_embed = None
        self._keep = []  # tensors the library holds raw pointers to
        self.model = _UDiTView(self)

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self.lib.ezdit_destroy(self._h)
        except Exception:
            pass

    # -- nn.Module-like surface ---------------------------------------------------------------------
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if self._blob is not None and device != self._blob.device:
            raise NotImplementedError('move the model before load_state_dict')
        self.device = device
        return self

    def load_state_dict(self, state_dict, strict=True):
        blob = pack_state_dict(self._h, state_dict, strict=strict)
        self._blob = blob.to(self.device)
        self._mask_embed = torch.as_tensor(np.asarray(state_dict['mask_embed']), dtype=torch.float32).to(self.device)
        _lib.check(self.lib.ezdit_bind_weights(self._h, _ptr(self._blob), self._blob.numel()))
        self._ws_key = None
        return self

    # -- workspace ----------------------------------------------------------------------------------
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

    def debug_buffer(self, name, dtype, shape=None):
        """Copy of an internal workspace buffer (tests / debugging)."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.lib.ezdit_debug_buffer(self._h, name.encode(), C.byref(p), C.byref(n)))
        off = p.value - self._ws.data_ptr()
        t = self._ws[off:off + n.value].view(dtype).clone()
        return t if shape is None else t[:int(np.prod(shape))].reshape(shape)

    # -- forward ------------------------------------------------------------------------------------
    def _run(self, x, timesteps, context, context_mask, gt, gt_mask, controlnet_skips, in_ch):
        B, _, L = x.shape
        Lc = context.shape[1]
        if context.shape[0] != B:
            raise AssertionError(f'context batch {context.shape[0]} != x batch {B}')
        ts = torch.as_tensor(timesteps)
        per_row = ts.dim() > 0
        t_list = [int(v) for v in ts.reshape(-1).tolist()] if per_row else [int(ts)]
        if per_row and len(t_list) != B:
            raise AssertionError(f'timesteps shape {tuple(ts.shape)} does not match batch {B}')
        self.bind(B, L, Lc, max(len(t_list), 1))
        self.prepare_context(context, context_mask)
        self.prepare_timesteps(t_list, per_row)
        x = x.to(self.device, torch.float32).contiguous()
        gt_d = None if gt is None else gt.to(self.device, torch.float32).contiguous()
        gm_d = None if gt_mask is None else gt_mask.to(self.device).expand(B, self.C, L).to(torch.uint8).contiguous()
        out = torch.empty(B, self.C, L, dtype=torch.float32, device=self.device)
        cn_arr, n_cn, keep = None, 0, []
        if controlnet_skips:
            keep = [s.to(self.device, torch.float32).contiguous() for s in controlnet_skips]
            n_cn = len(keep)
            cn_arr = (C.c_void_p * n_cn)(*[s.data_ptr() for s in keep])
        _lib.check(self.lib.ezdit_forward(self._h, _ptr(x), in_ch, B, _ptr(gt_d), _ptr(gm_d), cn_arr, n_cn,
                                          _ptr(out), _stream()))
        self._keep = [x, gt_d, gm_d, keep]
        return out

    def forward(self, x, timesteps, context, x_mask=None, context_mask=None, cls_token=None, gt=None,
                mae_mask_infer=None, forward_model=True):
        if x_mask is not None or cls_token is not None:
            raise NotImplementedError('x_mask / cls_token are not used by the shipped configs')
        if gt is not None and mae_mask_infer is None:
            raise NotImplementedError('training-time random span masking is out of scope (pass mae_mask_infer)')
        mae_mask = torch.ones_like(x) if gt is None else mae_mask_infer.expand_as(gt).type_as(gt)
        if not forward_model:  # conditioners.py:178-183: return the assembled 257-channel input only
            me = self._mask_embed.view(1, -1, 1).to(x.device).expand_as(x)
            g = me if gt is None else torch.where(mae_mask_infer.expand_as(gt), me, gt)
            return torch.cat([x, g, mae_mask[:, 0:1, :]], dim=1), mae_mask
        pred = self._run(x, timesteps, context, context_mask, gt, mae_mask_infer, None, in_ch=self.C)
        return pred, mae_mask

    __call__ = forward

    @property
    def last_launch_count(self):
        return self.lib.ezdit_last_launch_count(self._h)
