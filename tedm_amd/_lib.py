"""ctypes binding of libtedm_hip.so (C ABI: include/tedm_hip.h, include/tedm_hip_ops.h).

The product path has no CPU fallback: if the HIP library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / 'libtedm_hip.so'

TEDM_MAX_LEVELS = 8
PREC_BF16X3, PREC_BF16, PREC_FP16X2, PREC_F16F4 = 0, 1, 2, 3
PREC_FP16 = 4          # one fp16 product: what f16f4 runs on the k x k convolutions with >= 256 output channels (not a network mode)
FWD_TAPS_ONLY, FWD_TRACE, FWD_PREC_SHIFT = 1 << 0, 1 << 1, 8
PREC_BY_NAME = {'bf16x3': PREC_BF16X3, 'bf16': PREC_BF16, 'fp16x2': PREC_FP16X2, 'f16f4': PREC_F16F4}


class UnetCfg(C.Structure):
    _fields_ = [('dim', C.c_int), ('init_dim', C.c_int), ('out_dim', C.c_int), ('n_mults', C.c_int),
                ('dim_mults', C.c_int * TEDM_MAX_LEVELS), ('channels', C.c_int), ('groups', C.c_int),
                ('heads', C.c_int), ('dim_head', C.c_int), ('learned_sinusoidal_dim', C.c_int)]


BACKWARD_MARK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)      # tedm_hip.h: tedm_backward_mark_fn


class TedmError(RuntimeError):
    pass


_P = C.c_void_p
_SIGNATURES = {
    'tedm_last_error': (C.c_char_p, []),
    'tedm_version': (C.c_char_p, []),
    'tedm_source_digest': (C.c_char_p, []),
    'tedm_unet_create': (C.c_int, [C.POINTER(UnetCfg), C.POINTER(_P)]),
    'tedm_unet_destroy': (None, [_P]),
    'tedm_unet_num_params': (C.c_int, [_P]),
    'tedm_unet_param_key': (C.c_char_p, [_P, C.c_int]),
    'tedm_unet_param_numel': (C.c_longlong, [_P, C.c_int]),
    'tedm_unet_set_param': (C.c_int, [_P, C.c_char_p, _P, C.c_longlong]),
    'tedm_unet_pack': (C.c_int, [_P, _P]),
    'tedm_unet_set_precision': (C.c_int, [_P, C.c_int]),
    'tedm_unet_workspace_bytes': (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    'tedm_unet_forward': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P), _P, C.c_size_t, _P, C.c_int]),
    'tedm_unet_set_grad': (C.c_int, [_P, C.c_char_p, _P, C.c_longlong]),
    'tedm_unet_pack_backward': (C.c_int, [_P, _P]),
    'tedm_unet_train_workspace_bytes': (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    'tedm_unet_forward_train': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, C.c_int]),
    'tedm_unet_backward': (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    'tedm_unet_backward_gen': (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_longlong]),
    'tedm_sampler_step': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_longlong,
                                    C.c_float, _P, _P, _P, _P, _P]),
    'tedm_unet_tape_generation': (C.c_longlong, [_P]),
    'tedm_unet_tape_tap': (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    'tedm_unet_set_backward_callback': (C.c_int, [_P, _P, _P]),
    'tedm_l1_loss': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_longlong, _P, _P, _P, _P]),
    'tedm_unet_module_channels': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'tedm_unet_module_workspace_bytes': (C.c_size_t, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    'tedm_unet_module_forward': (C.c_int, [_P, C.c_char_p, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, C.c_int]),
    'tedm_conv1x1_packed_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'tedm_conv1x1_pack': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P]),
    'tedm_conv1x1_nhwc': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P]),
    'tedm_conv1x1_wgrad_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'tedm_conv1x1_wgrad_nhwc': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_size_t, _P]),
    'tedm_upsample_sum_nhwc': (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    'tedm_pool_sum_nhwc': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'tedm_head_tail_nhwc': (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    'tedm_head_train_stats1': (C.c_int, [_P, C.c_longlong, C.c_int, _P, _P, C.c_int, _P]),
    'tedm_head_train_mid': (C.c_int, [_P, C.c_longlong, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    'tedm_head_train_out': (C.c_int, [_P, C.c_longlong, _P, _P, _P, _P, _P, _P]),
    'tedm_head_train_bwd1': (C.c_int, [_P, _P, C.c_longlong, _P, C.c_int, _P]),
    'tedm_head_train_bwd2': (C.c_int, [_P, _P, _P, C.c_longlong, C.c_int, _P, _P, C.c_int, _P]),
    'tedm_head_train_bwd3': (C.c_int, [_P, _P, _P, C.c_longlong, C.c_int, _P, _P, _P, _P, _P]),
    'tedm_unet_trace_count': (C.c_int, [_P]),
    'tedm_unet_trace_name': (C.c_char_p, [_P, C.c_int]),
    'tedm_unet_trace_info': (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    'tedm_q_sample': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_longlong, C.c_int, _P]),
    'tedm_tap_to_features': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    'tedm_set_tuning': (C.c_int, [C.c_char_p, C.c_int]),
    'tedm_op_conv2d': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                                  C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int,
                                  C.POINTER(C.c_int), C.c_int, _P]),
    'tedm_op_conv2d_stats_entries': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'tedm_q16_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'tedm_q16_encode': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tedm_q16_decode': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tedm_op_conv2d_q16': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P,
                                      C.POINTER(C.c_int), C.c_int, C.c_int, _P]),
    'tedm_op_resblock_out_q16': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'tedm_op_conv2d_bench': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                                        C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    'tedm_op_conv2d_backward': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           _P, _P, C.c_int, _P]),
    'tedm_op_gn_silu_backward': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_longlong, C.c_int, C.c_int,
                                            _P, _P, _P, _P, _P]),
    'tedm_op_ln_backward': (C.c_int, [_P, _P, _P, _P, C.c_longlong, C.c_int, _P, _P, _P]),
    'tedm_op_linattn_fwd_bwd': (C.c_int, [_P, _P, C.c_int, C.c_longlong, _P, _P, _P]),
    'tedm_op_full_attention_fwd_bwd': (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    'tedm_op_linattn_fused': (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_longlong, _P, _P, _P, C.c_int, _P]),
    'tedm_profile_enable': (C.c_int, [C.c_int]),
    'tedm_profile_reset': (C.c_int, []),
    'tedm_profile_filter': (C.c_int, [C.c_char_p]),
    'tedm_profile_collect': (C.c_int, []),
    'tedm_profile_count': (C.c_int, []),
    'tedm_profile_entry': (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get('TEDM_HIP_LIB', LIB_PATH))
        if not path.exists():
            raise ImportError(
                f'{path} not found: build it with `python -m tedm_amd.build` (hipcc, gfx950). '
                'tedm_amd has no CPU/PyTorch fallback for the UNet hot path.')
        handle = C.CDLL(str(path))
        if 'TEDM_HIP_LIB' not in os.environ and os.environ.get('TEDM_ALLOW_STALE_LIB') != '1':
            # the binary must be the one built from the sources next to it (tedm_amd/build.py: digest of csrc/ + include/)
            from . import build as _build
            try:
                handle.tedm_source_digest.restype = C.c_char_p
                have = handle.tedm_source_digest().decode()
            except AttributeError:
                have = None
            want = _build.source_digest()
            if have != want:
                raise ImportError(f'{path} was built from other sources (digest {have and have[:12]} != {want[:12]}): rebuild with '
                                  '`python -m tedm_amd.build` (or set TEDM_ALLOW_STALE_LIB=1 to load it anyway)')
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = lib().tedm_last_error().decode(errors='replace')
        raise TedmError(f'{what or "libtedm_hip"} failed (code {rc}): {msg}')


def profile_entries():
    """[(label, launches, ms, flops, bytes)] accumulated since the last tedm_profile_reset()."""
    L = lib()
    check(L.tedm_profile_collect(), 'tedm_profile_collect')
    out = []
    for i in range(L.tedm_profile_count()):
        name = C.create_string_buffer(192)
        n, ms, fl, by = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
        check(L.tedm_profile_entry(i, name, 192, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        out.append((name.value.decode(), n.value, ms.value, fl.value, by.value))
    return out
