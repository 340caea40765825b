"""Drop-in for the reference's pybind module ``torchfx.torchfx_ext``.

Same three entry points, names, argument order and return tuples as
``src/torchfx/_csrc/binding.cpp:83-96`` (``biquad_forward``, ``sos_forward``,
``delay_line_forward``), backed by ``libtorchfx_hip.so`` through its C ABI
(``include/torchfx_hip.h``), plus the two ops the reference implements with torch
library calls (``F.conv1d`` / ``torch.fft``) and we implement in HIP:
``fir_direct_forward`` and ``fft_conv_forward``.

Differences a caller can observe, all deliberate:
  * tensors must be on a ROCm device -- there is no CPU path here;
  * ``x`` may be float32 *or* float64 (the reference's CUDA kernels need float64 and
    its Python layer upcasts, ``_ops.py:95,149``); by default the result has the dtype of
    ``x``.  The recurrences run in float64 unless ``precision`` says otherwise, so a
    float32-in/float32-out call equals "upcast, filter, downcast" with 8 B/sample of
    traffic instead of 32;
  * kernels are launched on PyTorch's *current* stream (the reference uses the default
    stream, ``parallel_scan.cu:299``).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
from torch import Tensor

from torchfx_amd import _lib as L

__all__ = [
    "biquad_forward", "sos_forward", "sos_bank_forward", "delay_line_forward",
    "fir_direct_forward", "fft_conv_forward", "sum_forward", "sos_plan_info", "ols_plan_info",
]

_TORCH_DT = {L.TFX_F32: torch.float32, L.TFX_F64: torch.float64}


def _host_f64(t, shape_last: int | None = None) -> np.ndarray:
    """Small coefficient tensor -> contiguous host float64 array (O(K) bytes)."""
    if isinstance(t, Tensor):
        a = t.detach().to(device="cpu", dtype=torch.float64).contiguous().numpy()
    else:
        a = np.ascontiguousarray(t, dtype=np.float64)
    if shape_last is not None and (a.ndim == 0 or a.shape[-1] != shape_last):
        raise RuntimeError(f"expected last dimension {shape_last}, got shape {tuple(a.shape)}")
    return a


def _state(t: Tensor | None, shape: tuple[int, ...], device, what: str) -> Tensor | None:
    if t is None:
        return None
    if tuple(t.shape) != shape:
        raise RuntimeError(f"{what} must have shape {shape}, got {tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float64).contiguous()


def _ptr(t: Tensor | None):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def sos_forward(x: Tensor, sos: Tensor, sos_cpu: Tensor | None, state_x: Tensor | None,
                state_y: Tensor | None, *, out_dtype: torch.dtype | None = None,
                precision=None, return_sections: bool = False):
    """SOS cascade forward -- ``binding.cpp:52-66``.

    ``x [C,T]``, ``sos [K,6]`` (device copy, unused here), ``sos_cpu [K,6]`` host float64
    (the reference's sync-avoidance argument; falls back to ``sos.cpu()``), states
    ``[K,C,2]`` float64 or ``None``.  Returns ``(y [C,T], new_state_x, new_state_y)``;
    inputs are never modified.
    """
    if x.dim() != 2:
        raise RuntimeError(f"sos_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    sos_h = _host_f64(sos_cpu if sos_cpu is not None else sos, 6)
    if sos_h.ndim != 2:
        raise RuntimeError("sos_forward: sos must be [K, 6]")
    K = sos_h.shape[0]
    sx = _state(state_x, (K, C, 2), x.device, "state_x")
    sy = _state(state_y, (K, C, 2), x.device, "state_y")
    odt = x.dtype if out_dtype is None else out_dtype
    y = torch.empty((C, T), dtype=odt, device=x.device)
    nsx = torch.empty((K, C, 2), dtype=torch.float64, device=x.device)
    nsy = torch.empty((K, C, 2), dtype=torch.float64, device=x.device)
    sec = torch.empty((K, C, T), dtype=odt, device=x.device) if return_sections else None
    with torch.cuda.device(x.device):
        L.check(lib.tfx_sos_forward(
            _ptr(x), L.dtype_code(x), _ptr(y), L.dtype_code(y), C, T,
            sos_h.ctypes.data_as(ctypes.c_void_p), K,
            _ptr(sx), _ptr(sy), _ptr(nsx), _ptr(nsy), _ptr(sec),
            L.precision_code(precision), ctypes.c_void_p(L.stream_ptr(x))))
    if return_sections:
        return y, nsx, nsy, sec
    return y, nsx, nsy


def sos_bank_forward(x: Tensor, sos_banks, state_x: Tensor | None, state_y: Tensor | None, *,
                     out_dtype: torch.dtype | None = None, precision=None):
    """Filter bank: ``sos_banks [NB,K,6]`` (host), ``x [C,T]`` -> ``y [NB,C,T]`` in one launch;
    states ``[K, NB*C, 2]`` (band-major rows) or ``None``.  Replaces the Python loop of
    ``LogFilterBank.forward`` (``filterbank.py:157-185``)."""
    if x.dim() != 2:
        raise RuntimeError(f"sos_bank_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    sos_h = _host_f64(sos_banks, 6)
    if sos_h.ndim != 3:
        raise RuntimeError("sos_bank_forward: sos_banks must be [NB, K, 6]")
    NB, K = sos_h.shape[0], sos_h.shape[1]
    sx = _state(state_x, (K, NB * C, 2), x.device, "state_x")
    sy = _state(state_y, (K, NB * C, 2), x.device, "state_y")
    odt = x.dtype if out_dtype is None else out_dtype
    y = torch.empty((NB, C, T), dtype=odt, device=x.device)
    nsx = torch.empty((K, NB * C, 2), dtype=torch.float64, device=x.device)
    nsy = torch.empty_like(nsx)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_sos_bank_forward(
            _ptr(x), L.dtype_code(x), _ptr(y), L.dtype_code(y), C, T,
            sos_h.ctypes.data_as(ctypes.c_void_p), NB, K,
            _ptr(sx), _ptr(sy), _ptr(nsx), _ptr(nsy),
            L.precision_code(precision), ctypes.c_void_p(L.stream_ptr(x))))
    return y, nsx, nsy


def sos_bank_sum_forward(x: Tensor, sos_banks, state_x: Tensor | None, state_y: Tensor | None, *, precision=None):
    """``f1 + f2 + ...`` of IIR branches in one launch: ``sos_banks [NB,K,6]`` (host), ``x [C,T]`` ->
    ``y [C,T] = sum_b cascade_b(x)`` with the reference's accumulation order and rounding
    (``__base.py:1019-1026``); states ``[K, NB*C, 2]`` (band-major rows) or ``None``."""
    if x.dim() != 2:
        raise RuntimeError(f"sos_bank_sum_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    sos_h = _host_f64(sos_banks, 6)
    if sos_h.ndim != 3:
        raise RuntimeError("sos_bank_sum_forward: sos_banks must be [NB, K, 6]")
    NB, K = sos_h.shape[0], sos_h.shape[1]
    sx = _state(state_x, (K, NB * C, 2), x.device, "state_x")
    sy = _state(state_y, (K, NB * C, 2), x.device, "state_y")
    y = torch.empty_like(x)
    nsx = torch.empty((K, NB * C, 2), dtype=torch.float64, device=x.device)
    nsy = torch.empty_like(nsx)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_sos_bank_sum_forward(
            _ptr(x), L.dtype_code(x), _ptr(y), L.dtype_code(y), C, T,
            sos_h.ctypes.data_as(ctypes.c_void_p), NB, K,
            _ptr(sx), _ptr(sy), _ptr(nsx), _ptr(nsy),
            L.precision_code(precision), ctypes.c_void_p(L.stream_ptr(x))))
    return y, nsx, nsy


def biquad_forward(x: Tensor, b: Tensor, a1: float, a2: float, state_x: Tensor | None,
                   state_y: Tensor | None, *, out_dtype: torch.dtype | None = None, precision=None):
    """Single biquad forward -- ``binding.cpp:30-50``: ``b [3]`` tensor, ``a1``/``a2``
    Python floats, states ``[C,2]``.  Returns ``(y, new_state_x, new_state_y)``."""
    if x.dim() != 2:
        raise RuntimeError(f"biquad_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    b_h = _host_f64(b, 3).reshape(3)
    sx = _state(state_x, (C, 2), x.device, "state_x")
    sy = _state(state_y, (C, 2), x.device, "state_y")
    odt = x.dtype if out_dtype is None else out_dtype
    y = torch.empty((C, T), dtype=odt, device=x.device)
    nsx = torch.empty((C, 2), dtype=torch.float64, device=x.device)
    nsy = torch.empty((C, 2), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_biquad_forward(
            _ptr(x), L.dtype_code(x), _ptr(y), L.dtype_code(y), C, T,
            b_h.ctypes.data_as(ctypes.c_void_p), float(a1), float(a2),
            _ptr(sx), _ptr(sy), _ptr(nsx), _ptr(nsy),
            L.precision_code(precision), ctypes.c_void_p(L.stream_ptr(x))))
    return y, nsx, nsy


def delay_line_forward(x: Tensor, delay_samples: int, decay: float, mix: float) -> Tensor:
    """``binding.cpp:68-81`` / ``delay_cpu.cpp:43-85``.  Like the reference, returns the
    input tensor itself when the signal is not longer than the delay."""
    L.require_device(x, "x")
    lib = L.load()
    xc = x.contiguous()
    T = xc.shape[-1] if xc.dim() else 1
    if T <= delay_samples:
        return x
    C = xc.numel() // T                       # any (..., T) layout is rows x T
    y = torch.empty_like(xc)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_delay_line_forward(_ptr(xc), _ptr(y), L.dtype_code(xc), C, T, int(delay_samples),
                                           float(decay), float(mix), ctypes.c_void_p(L.stream_ptr(x))))
    return y


_TAPS_HOST: dict = {}        # id(tensor) -> (weakref, version, dtype, host array)


def _kernel_host(kernel, dtype: torch.dtype) -> np.ndarray:
    """Flat host copy of the taps in the signal's dtype.  The C ABI takes the taps as a host array
    (they key its device-side caches), so a filter that was moved to the GPU would otherwise pay a
    blocking device-to-host copy on every forward: the copy is cached per tensor object and
    invalidated by its version counter (in-place edits) or its death."""
    npdt = np.float32 if dtype == torch.float32 else np.float64
    if not isinstance(kernel, Tensor):
        return np.ascontiguousarray(np.asarray(kernel).reshape(-1), dtype=npdt)
    ent = _TAPS_HOST.get(id(kernel))
    if ent is not None and ent[0]() is kernel and ent[1] == kernel._version and ent[2] == dtype:
        return ent[3]
    k = kernel.detach().to(device="cpu").reshape(-1).to(dtype).contiguous().numpy()
    if kernel.is_cuda:
        import weakref

        if len(_TAPS_HOST) > 64:
            for key in [key for key, e in _TAPS_HOST.items() if e[0]() is None] or list(_TAPS_HOST)[:32]:
                _TAPS_HOST.pop(key, None)
        _TAPS_HOST[id(kernel)] = (weakref.ref(kernel), kernel._version, dtype, k)
    return k


def fir_direct_forward(x: Tensor, kernel) -> Tensor:
    """Causal depthwise FIR, direct form: the ``conv_mode="direct"`` branch of
    ``FIR.forward`` (``fir.py:556-568``).  ``x [C,T]``; ``kernel`` = the FLIPPED taps
    (the module's ``[1,1,K]`` buffer, any shape with K elements)."""
    if x.dim() != 2:
        raise RuntimeError(f"fir_direct_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    k = _kernel_host(kernel, x.dtype)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_fir_direct_forward(_ptr(x), _ptr(y), L.dtype_code(x), C, T,
                                           k.ctypes.data_as(ctypes.c_void_p), k.shape[0],
                                           ctypes.c_void_p(L.stream_ptr(x))))
    return y


def fft_conv_forward(x: Tensor, kernel, padding: tuple[int, int] = (0, 0)) -> Tensor:
    """Overlap-save FFT convolution with ``fft_conv1d`` semantics (``_fftconv.py:70-141``)
    on ``x [C,T]``: returns ``[C, T + l + r - K + 1]``."""
    if x.dim() != 2:
        raise RuntimeError(f"fft_conv_forward: x must be [C, T], got {tuple(x.shape)}")
    L.require_device(x, "x")
    lib = L.load()
    x = x.contiguous()
    C, T = x.shape
    k = _kernel_host(kernel, x.dtype)
    pl, pr = int(padding[0]), int(padding[1])
    K = k.shape[0]
    tout = T + pl + pr - K + 1
    y = torch.empty((C, max(tout, 0)), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_fft_conv_forward(_ptr(x), _ptr(y), L.dtype_code(x), C, T,
                                         k.ctypes.data_as(ctypes.c_void_p), K, pl, pr,
                                         ctypes.c_void_p(L.stream_ptr(x))))
    return y


def sum_forward(tensors: list[Tensor]) -> Tensor:
    """Sum of equally-shaped tensors in list order (``__base.py:1022-1026``)."""
    if not tensors:
        raise RuntimeError("sum_forward: need at least one tensor")
    lib = L.load()
    ts = [t.contiguous() for t in tensors]
    for t in ts:
        L.require_device(t, "branch output")
        if t.shape != ts[0].shape or t.dtype != ts[0].dtype:
            raise RuntimeError("sum_forward: branch outputs differ in shape or dtype")
    out = torch.empty_like(ts[0])
    for i in range(0, len(ts), 15):   # the kernel takes up to 16 inputs per launch
        grp = ts[i:i + 15] if i == 0 else [out] + ts[i:i + 15]
        arr = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        with torch.cuda.device(out.device):
            L.check(lib.tfx_sum_forward(arr, len(grp), _ptr(out), L.dtype_code(out), out.numel(),
                                        ctypes.c_void_p(L.stream_ptr(out))))
    return out


STAT_ABSMAX, STAT_RMS = 0, 1


def _rows_view(x: Tensor) -> tuple[Tensor, int, int]:
    """``(..., T)`` -> contiguous tensor plus its ``[rows, T]`` geometry."""
    xc = x.contiguous()
    T = xc.shape[-1] if xc.dim() else 1
    return xc, (xc.numel() // T if T else 0), T


def gain_forward(x: Tensor, gain: float, clamp: bool = False) -> Tensor:
    """``y = x * gain`` (+ clip to [-1, 1]) -- ``Gain.forward``, ``effect.py:361-383``; ``gain`` is
    the linear factor."""
    L.require_device(x, "x")
    lib = L.load()
    xc = x.contiguous()
    y = torch.empty_like(xc)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_gain_forward(_ptr(xc), _ptr(y), L.dtype_code(xc), xc.numel(), float(gain), int(bool(clamp)),
                                     ctypes.c_void_p(L.stream_ptr(x))))
    return y


def stat_forward(x: Tensor, mode: int = STAT_ABSMAX, per_row: bool = False) -> Tensor:
    """``max|x|`` (``STAT_ABSMAX``) or ``sqrt(mean(x^2))`` (``STAT_RMS``) over everything, or per row
    of the ``[rows, T]`` view -- float64 on the device, no host sync."""
    L.require_device(x, "x")
    lib = L.load()
    xc, rows, T = _rows_view(x)
    out = torch.empty(rows if per_row else 1, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_stat_forward(_ptr(xc), L.dtype_code(xc), rows, T, int(mode), int(bool(per_row)), _ptr(out),
                                     ctypes.c_void_p(L.stream_ptr(x))))
    return out


def normalize_forward(x: Tensor, peak: float, mode: int = STAT_ABSMAX, per_row: bool = False) -> Tensor:
    """``s > 0 ? x / s * peak : x`` with ``s`` = abs-max or RMS, global or per row of the ``[rows, T]``
    view (``effect.py:696-698,719-721,775-786``); two streaming passes, statistic stays on device."""
    L.require_device(x, "x")
    lib = L.load()
    xc, rows, T = _rows_view(x)
    y = torch.empty_like(xc)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_normalize_forward(_ptr(xc), _ptr(y), L.dtype_code(xc), rows, T, int(mode),
                                          int(bool(per_row)), float(peak), ctypes.c_void_p(L.stream_ptr(x))))
    return y


def deinterleave_forward(frames: Tensor, out: Tensor | None = None, frame_base: int = 0,
                         scale: float = 1.0 / 32768.0) -> Tensor:
    """Interleaved ``[F, C]`` (float32, or int16 PCM scaled by ``scale``) -> planar float32 ``[C, F]``
    (the device-side ``data_np.T.copy()`` of ``wave.py:448-452``).  With ``out`` ``[C, F_total]`` the
    chunk lands at frames ``[frame_base, frame_base + F)`` of every row."""
    L.require_device(frames, "frames")
    if frames.dim() != 2 or frames.dtype not in (torch.float32, torch.int16):
        raise RuntimeError(f"deinterleave_forward: expected [F, C] float32 or int16, got {tuple(frames.shape)} {frames.dtype}")
    lib = L.load()
    fr = frames.contiguous()
    F, C = fr.shape
    if out is None:
        out = torch.empty((C, F), dtype=torch.float32, device=fr.device)
    if out.dim() != 2 or out.shape[0] != C or out.dtype != torch.float32 or not out.is_contiguous() or out.device != fr.device:
        raise RuntimeError("deinterleave_forward: out must be a contiguous float32 [C, F_total] tensor on the same device")
    with torch.cuda.device(fr.device):
        L.check(lib.tfx_deinterleave_forward(_ptr(fr), 0 if fr.dtype == torch.float32 else 1, _ptr(out), F, C,
                                             out.shape[1], int(frame_base), float(scale),
                                             ctypes.c_void_p(L.stream_ptr(fr))))
    return out


def interleave_forward(x: Tensor, frame_base: int = 0, frames: int | None = None) -> Tensor:
    """Planar float32 ``[C, F_total]`` -> interleaved ``[F, C]`` of frames ``[frame_base, frame_base+F)``
    (the device-side ``.numpy().T`` of ``wave.py:566-573``)."""
    L.require_device(x, "x")
    if x.dim() != 2 or x.dtype != torch.float32:
        raise RuntimeError(f"interleave_forward: expected float32 [C, F], got {tuple(x.shape)} {x.dtype}")
    lib = L.load()
    xc = x.contiguous()
    C, Ft = xc.shape
    F = Ft - int(frame_base) if frames is None else int(frames)
    out = torch.empty((max(F, 0), C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.tfx_interleave_forward(_ptr(xc), _ptr(out), F, C, Ft, int(frame_base),
                                           ctypes.c_void_p(L.stream_ptr(x))))
    return out


def sos_plan_info(sos) -> dict:
    """Host-side plan facts for an SOS matrix: warm-up halo length, the f32 worst-case
    error bound and what ``precision='auto'`` would choose.  Needs the library but no GPU."""
    lib = L.load()
    s = _host_f64(sos, 6)
    prec, warm, eb = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_double(0.0)
    L.check(lib.tfx_sos_plan_info(s.ctypes.data_as(ctypes.c_void_p), s.shape[0],
                                  ctypes.byref(prec), ctypes.byref(warm), ctypes.byref(eb)))
    return {"auto_precision": "f32" if prec.value == L.PREC_F32 else "f64",
            "warmup": warm.value, "f32_error_bound": eb.value}


def ols_plan_info(K: int, T: int, padding: tuple[int, int] = (0, 0)) -> dict:
    """Block geometry of the overlap-save op (FFT length N, hop S, blocks per row F, native path?)."""
    lib = L.load()
    n, s_, f, nat = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    L.check(lib.tfx_ols_plan_info(int(K), int(T), int(padding[0]), int(padding[1]), ctypes.byref(n),
                                  ctypes.byref(s_), ctypes.byref(f), ctypes.byref(nat)))
    return {"N": n.value, "S": s_.value, "F": f.value, "native": bool(nat.value)}
