"""ctypes binding of libezaudio_hip.so (include/ezdit.h).

There is NO fallback: if the HIP library is missing or does not export the ABI this module raises,
so a GPU box can never silently run a non-native path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libezaudio_hip.so')
ABI_VERSION = 4


class EzditConfig(C.Structure):
    _fields_ = [('embed_dim', C.c_int32), ('num_heads', C.c_int32), ('depth', C.c_int32),
                ('in_chans', C.c_int32), ('out_chans', C.c_int32), ('context_dim', C.c_int32),
                ('ada_sola_rank', C.c_int32), ('ada_sola_alpha', C.c_float), ('mlp_ratio', C.c_float),
                ('max_len', C.c_int32), ('controlnet', C.c_int32), ('cond_in', C.c_int32), ('cond_c0', C.c_int32),
                ('cond_c1', C.c_int32), ('cond_mask', C.c_int32)]


class EzditParamInfo(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('src', (C.c_char * 96) * 3), ('nsrc', C.c_int32),
                ('dtype', C.c_int32), ('transform', C.c_int32),
                ('rows', C.c_int64), ('cols', C.c_int64), ('rows_pad', C.c_int64), ('ld', C.c_int64),
                ('offset', C.c_int64)]


class EzditDdimCoef(C.Structure):
    _fields_ = [('sa', C.c_float), ('sb', C.c_float), ('c_x0', C.c_float), ('c_dir', C.c_float),
                ('sigma', C.c_float)]


P_F32, P_BF16 = 0, 1
T_NONE, T_GEGLU8, T_QKROPE = 0, 1, 2

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
PROTOTYPES = {
    'ezdit_abi_version': (C.c_int, []),
    'ezdit_last_error': (C.c_char_p, []),
    'ezdit_create': (C.c_int, [C.POINTER(EzditConfig), C.POINTER(C.c_void_p)]),
    'ezdit_destroy': (C.c_int, [C.c_void_p]),
    'ezdit_param_count': (C.c_int, [C.c_void_p]),
    'ezdit_param_info': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EzditParamInfo)]),
    'ezdit_param_bytes': (C.c_size_t, [C.c_void_p]),
    'ezdit_bind_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'ezdit_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    'ezdit_bind_workspace': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_prepare_context': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ezdit_prepare_timesteps': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p]),
    'ezdit_set_step': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'ezdit_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p]),
    'ezdit_prepare_condition': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'ezdit_controlnet_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ezdit_controlnet_residuals': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    'ezdit_sampler_attach_controlnet': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float]),
    'ezdit_set_cn_scale': (C.c_int, [C.c_void_p, C.c_float]),
    'ezdit_sampler_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(EzditDdimCoef), C.c_int,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ezdit_sampler_run': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_cfg_ddim_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    'ezvae_gemm': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    'ezvae_snake_bf16': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    'ezvae_conv_out1': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
    'ezvae_conv_in1': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
    'ezvae_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_test_gemm': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_debug_gemm_timestamps': (C.c_int, [C.c_void_p, C.c_long]),
    'ezdit_test_resid': (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_test_resid_skip': (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    'ezdit_test_attention': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'ezdit_debug_buffer': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    'ezdit_last_launch_count': (C.c_int, [C.c_void_p]),
    'ezdit_debug_stop_after': (C.c_int, [C.c_void_p, C.c_int]),
    'ezdit_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
}

_lib = None


class EzditError(RuntimeError):
    pass


def load():
    """Load the HIP library (raises if it has not been built: run `python -m ezaudio_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: import it FIRST so that this library binds to the runtime instance torch
    # initialises (loading us first was observed to leave the library with "no ROCm-capable device")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EzditError(f'{LIB_PATH} is missing: the HIP extension has not been built '
                         f'(python -m ezaudio_amd.build). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ezdit_abi_version() != ABI_VERSION:
        raise EzditError(f'ABI version mismatch: library {lib.ezdit_abi_version()} != binding {ABI_VERSION}')
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ezdit_last_error()
        code = {-1: 'EZDIT_E_INVALID', -2: 'EZDIT_E_UNSUPPORTED', -3: 'EZDIT_E_STATE', -4: 'EZDIT_E_HIP'}.get(rc, str(rc))
        text = f'{code}: {msg.decode() if msg else ""}'
        if rc == -2:
            raise NotImplementedError(text)  # mirrors src/models/udit.py:83,113,127
        if rc == -1:
            raise AssertionError(text)       # shape errors are AssertionError-class in the reference
        raise EzditError(text)
