"""ctypes binding of ``libtorchfx_hip.so`` (C ABI declared in ``include/torchfx_hip.h``).

There is NO CPU fallback: if the shared library is missing or a tensor is not on a
ROCm device the calls raise ``RuntimeError`` -- loudly, as the task contract requires.
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtorchfx_hip.so")

TFX_F32, TFX_F64 = 0, 1
PREC_AUTO, PREC_F32, PREC_F64 = 0, 1, 2
_PREC_NAMES = {"auto": PREC_AUTO, "f32": PREC_F32, "float32": PREC_F32, "f64": PREC_F64, "float64": PREC_F64}

_lock = threading.Lock()
_lib: ctypes.CDLL | None = None

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> (restype, argtypes); mirrors include/torchfx_hip.h one to one
SIGNATURES = {
    "tfx_version": (_int, []),
    "tfx_last_error": (ctypes.c_char_p, []),
    "tfx_device_info": (_int, [ctypes.c_char_p, _int, ctypes.POINTER(_int)]),
    "tfx_sos_forward": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "tfx_sos_bank_forward": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp]),
    "tfx_sos_bank_sum_forward": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp]),
    "tfx_sos_plan_info": (_int, [_vp, _i64, ctypes.POINTER(_int), ctypes.POINTER(_i64), ctypes.POINTER(_dbl)]),
    "tfx_biquad_forward": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _dbl, _dbl, _vp, _vp, _vp, _vp, _int, _vp]),
    "tfx_fir_direct_forward": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _i64, _vp]),
    "tfx_fft_conv_forward": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _i64, _i64, _i64, _vp]),
    "tfx_sos_forward_ep": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp]),
    "tfx_fft_conv_forward_ep": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tfx_sos_fft_conv_supported": (_int, [_i64, ctypes.POINTER(_dbl), _i64, _i64, _i64, _i64, _int]),
    "tfx_sos_fft_conv_warmup": (_i64, [ctypes.POINTER(_dbl), _i64]),
    "tfx_sos_fft_conv_plan_info": (_int, [_i64, ctypes.POINTER(_dbl), _i64, _i64, _i64, _i64, _int, ctypes.POINTER(_i64),
                                          ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "tfx_sos_fft_conv_forward": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _int, _vp, _vp]),
    "tfx_normalize_apply": (_int, [_vp, _vp, _int, _i64, _i64, _int, _int, _dbl, _vp, _vp]),
    "tfx_fir_stream_forward": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _i64, _int, _vp, _vp, _vp]),
    "tfx_quantile_abs": (_int, [_vp, _i64, _dbl, _vp, _vp]),
    "tfx_chunk_supported": (_int, [_i64, _i64, _i64, _i64]),
    "tfx_chunk_forward": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _dbl, _int, _int, _int, _vp]),
    "tfx_ols_plan_info": (_int, [_i64, _i64, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                 ctypes.POINTER(_i64), ctypes.POINTER(_int)]),
    "tfx_ols_plan_info2": (_int, [_i64, _i64, _i64, _i64, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                  ctypes.POINTER(_i64), ctypes.POINTER(_int)]),
    "tfx_prewarm": (_int, []),
    "tfx_env_reload": (_int, []),
    "tfx_delay_line_forward": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _dbl, _dbl, _vp]),
    "tfx_sum_forward": (_int, [_vp, _int, _vp, _int, _i64, _vp]),
    "tfx_gain_forward": (_int, [_vp, _vp, _int, _i64, _dbl, _int, _vp]),
    "tfx_stat_forward": (_int, [_vp, _int, _i64, _i64, _int, _int, _vp, _vp]),
    "tfx_normalize_forward": (_int, [_vp, _vp, _int, _i64, _i64, _int, _int, _dbl, _vp]),
    "tfx_deinterleave_forward": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _i64, _dbl, _vp]),
    "tfx_interleave_forward": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "tfx_prof_enable": (_int, [_int]),
    "tfx_prof_collect": (ctypes.c_char_p, []),
    "tfx_clear_caches": (_int, []),
    "tfx_set_workspace_allocator": (_int, [_vp, _vp, _vp]),
    "tfx_workspace_bytes": (_i64, []),
}


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"torchfx_amd: {LIB_PATH} not found -- build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C torchfx_amd/csrc`).  There is no CPU fallback."
                )
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError here = header/library mismatch
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def is_loaded() -> bool:
    return _lib is not None


def check(rc: int) -> None:
    if rc != 0:
        msg = load().tfx_last_error()
        raise RuntimeError((msg or b"unknown error").decode("utf-8", "replace"))


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TFX_F32
    if t.dtype == torch.float64:
        return TFX_F64
    raise RuntimeError(f"torchfx_amd: unsupported dtype {t.dtype} (float32 / float64 only)")


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"torchfx_amd: {what} must live on a ROCm device (got {t.device}); "
            "this backend has no CPU path -- move the tensor with .to('cuda')."
        )


def stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def default_precision() -> int:
    """IIR arithmetic: env TORCHFX_AMD_IIR_PRECISION = f64 (default) | f32 | auto."""
    name = os.environ.get("TORCHFX_AMD_IIR_PRECISION", "f64").lower()
    if name not in _PREC_NAMES:
        raise RuntimeError(f"TORCHFX_AMD_IIR_PRECISION={name!r}: expected one of {sorted(_PREC_NAMES)}")
    return _PREC_NAMES[name]


def precision_code(p) -> int:
    if p is None:
        return default_precision()
    if isinstance(p, int):
        return p
    return _PREC_NAMES[str(p).lower()]
