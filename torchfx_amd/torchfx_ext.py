"""Python face of the native module: the reference's ``torchfx.torchfx_ext`` names plus our extra ops.

The three entry points of the reference's pybind module (``src/torchfx/_csrc/binding.cpp:83-96``:
``biquad_forward``, ``sos_forward``, ``delay_line_forward``) and the ops the reference implements with
torch library calls (``F.conv1d`` / ``torch.fft``) and we implement in HIP (``fir_direct_forward``,
``fft_conv_forward``, the filter-bank / sum forms, Gain / Normalize, the layout kernels).  Every tensor
call goes through the PyTorch dispatcher to the compiled extension (``torch.ops.torchfx_hip.*``,
``torchfx_amd/csrc/ext/torchfx_ext.cpp``), which calls the C ABI of ``libtorchfx_hip.so``
(``include/torchfx_hip.h``); the functions here only add keyword conveniences (``out_dtype``,
``precision``) and keep a host copy of FIR taps.  ``ctypes`` (``torchfx_amd._lib``) is used for the
host-only planning queries at the bottom, which take no tensors.

Differences a caller can observe, all deliberate:
  * tensors must be on a ROCm device -- there is no CPU path here;
  * ``x`` may be float32 *or* float64 (the reference's CUDA kernels need float64 and its Python layer
    upcasts, ``_ops.py:95,149``); by default the result has the dtype of ``x``.  The recurrences run in
    float64 unless ``precision`` says otherwise, so a float32-in/float32-out call equals "upcast,
    filter, downcast" with 8 B/sample of traffic instead of 32;
  * kernels are launched on PyTorch's *current* stream (the reference uses the default stream,
    ``parallel_scan.cu:299``).
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch
from torch import Tensor

from torchfx_amd import _lib as L
from torchfx_amd import native

__all__ = [
    "biquad_forward", "sos_forward", "sos_bank_forward", "sos_bank_sum_forward", "delay_line_forward",
    "fir_direct_forward", "fft_conv_forward", "sos_fft_conv_forward", "sos_fft_conv_supported", "sos_fft_conv_warmup", "sos_fft_conv_plan_info", "workspace_bytes", "clear_caches", "env_reload", "fir_stream_forward", "chunk_forward", "chunk_supported", "normalize_apply", "Epilogue", "sum_forward", "gain_forward", "quantile_abs", "stat_forward", "normalize_forward",
    "deinterleave_forward", "interleave_forward", "sos_plan_info", "ols_plan_info", "prewarm",
]


class Epilogue:
    """What the producing kernel does to the samples it stores (``include/torchfx_hip.h``, ``tfx_epilogue``):
    ``gain`` (linear factor) and ``clamp`` = a following ``Gain``; ``stat`` ("absmax" | "sumsq" | None, optionally
    ``per_row``) = the reduction half of a following ``Normalize``.  After the call ``stat_value`` holds the raw
    statistic on the device (float64 ``[rows]`` or ``[1]``) for :func:`normalize_apply`."""

    __slots__ = ("gain", "clamp", "stat", "per_row", "stat_value")

    def __init__(self, gain: float = 1.0, clamp: bool = False, stat: str | None = None, per_row: bool = False) -> None:
        if stat not in (None, "absmax", "sumsq"):
            raise ValueError(f"stat must be None, 'absmax' or 'sumsq', got {stat!r}")
        self.gain, self.clamp, self.stat, self.per_row = float(gain), bool(clamp), stat, bool(per_row)
        self.stat_value: Tensor | None = None

    @property
    def stat_mode(self) -> int:
        return {None: -1, "absmax": 0, "sumsq": 1}[self.stat]


def _prec(precision) -> int:
    return -1 if precision is None else L.precision_code(precision)


def _coeff(t) -> Tensor:
    """Coefficient arrays (numpy / lists / tensors) as a host float64 tensor."""
    if isinstance(t, Tensor):
        return t
    return torch.from_numpy(np.ascontiguousarray(t, dtype=np.float64))


def sos_forward(x: Tensor, sos: Tensor | None, sos_cpu: Tensor | None, state_x: Tensor | None,
                state_y: Tensor | None, *, out_dtype: torch.dtype | None = None,
                precision=None, return_sections: bool = False, epilogue: Epilogue | None = None):
    """SOS cascade forward -- ``binding.cpp:52-66``.

    ``x [C,T]``, ``sos [K,6]`` (device copy, unused here), ``sos_cpu [K,6]`` host float64 (the reference's
    sync-avoidance argument; falls back to ``sos``), states ``[K,C,2]`` float64 or ``None`` (= zeros).
    Returns ``(y [C,T], new_state_x, new_state_y)`` (+ every section's output ``[K,C,T]`` with
    ``return_sections``); inputs are never modified."""
    ops = native.ops()
    coeff = _coeff(sos_cpu if sos_cpu is not None else sos)
    if epilogue is not None:
        if return_sections:
            raise RuntimeError("sos_forward: no epilogue together with section taps")
        y, nsx, nsy, epilogue.stat_value = ops.sos_forward_ep(
            x, coeff, state_x, state_y, epilogue.gain, epilogue.clamp, epilogue.stat_mode, epilogue.per_row,
            out_dtype=out_dtype, precision=_prec(precision))
        return y, nsx, nsy
    if return_sections:
        return ops.sos_forward_sections(x, coeff, state_x, state_y, out_dtype=out_dtype, precision=_prec(precision))
    return ops.sos_forward(x, coeff, state_x, state_y, out_dtype=out_dtype, precision=_prec(precision))


def sos_bank_forward(x: Tensor, sos_banks, state_x: Tensor | None, state_y: Tensor | None, *,
                     out_dtype: torch.dtype | None = None, precision=None):
    """Filter bank: ``sos_banks [NB,K,6]`` (host), ``x [C,T]`` -> ``y [NB,C,T]`` in one launch; states
    ``[K, NB*C, 2]`` (band-major rows) or ``None``.  Replaces the Python loop of ``LogFilterBank.forward``
    (``filterbank.py:157-185``)."""
    return native.ops().sos_bank_forward(x, _coeff(sos_banks), state_x, state_y, out_dtype=out_dtype,
                                         precision=_prec(precision))


def sos_bank_sum_forward(x: Tensor, sos_banks, state_x: Tensor | None, state_y: Tensor | None, *, precision=None):
    """``f1 + f2 + ...`` of IIR branches in one launch: ``sos_banks [NB,K,6]`` (host), ``x [C,T]`` ->
    ``y [C,T] = sum_b cascade_b(x)`` with the reference's accumulation order and rounding
    (``__base.py:1019-1026``); states ``[K, NB*C, 2]`` (band-major rows) or ``None``."""
    return native.ops().sos_bank_sum_forward(x, _coeff(sos_banks), state_x, state_y, precision=_prec(precision))


def biquad_forward(x: Tensor, b: Tensor, a1: float, a2: float, state_x: Tensor | None,
                   state_y: Tensor | None, *, out_dtype: torch.dtype | None = None, precision=None):
    """Single biquad forward -- ``binding.cpp:30-50``: ``b [3]`` tensor, ``a1``/``a2`` Python floats, states
    ``[C,2]``.  Returns ``(y, new_state_x, new_state_y)``."""
    return native.ops().biquad_forward(x, _coeff(b), float(a1), float(a2), state_x, state_y, out_dtype=out_dtype,
                                       precision=_prec(precision))


def delay_line_forward(x: Tensor, delay_samples: int, decay: float, mix: float) -> Tensor:
    """``binding.cpp:68-81`` / ``delay_cpu.cpp:43-85``.  Like the reference, returns the input tensor itself
    when the signal is not longer than the delay."""
    return native.ops().delay_line_forward(x, int(delay_samples), float(decay), float(mix))


_TAPS_HOST: dict = {}        # (id(base tensor), offset, numel, dtype wanted) -> (weakref to base, version, host tensor)


def _kernel_host(kernel, dtype: torch.dtype) -> Tensor:
    """Flat host copy of the taps in the signal's dtype.  The C ABI takes the taps as a host array (they
    key its device-side caches), so a filter whose taps live on the GPU, or in another dtype (the planner's merged
    kernels are float64), would otherwise pay a copy on every forward -- a blocking device-to-host copy in the first
    case, a fresh quarter-megabyte host allocation in the second (and a process that keeps mapping and unmapping
    host memory while kernels are in flight gets its GPU queues stalled by the driver for tens of milliseconds:
    measured, `profiles/r02_experiments.txt`).  The copy is cached per BASE tensor object -- modules hand in
    `self.kernel.reshape(-1)`, a new view object per call -- and invalidated by the version counter (in-place
    edits) or the tensor's death."""
    if not isinstance(kernel, Tensor):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(kernel).reshape(-1))).to(dtype)
    base = kernel._base if kernel._base is not None else kernel
    key = (id(base), kernel.storage_offset(), kernel.numel(), tuple(kernel.stride()), dtype)
    ent = _TAPS_HOST.get(key)
    if ent is not None and ent[0]() is base and ent[1] == kernel._version:
        return ent[2]
    k = kernel.detach().to(device="cpu").reshape(-1).to(dtype).contiguous()
    if k.data_ptr() == kernel.data_ptr():          # nothing was copied: nothing to cache (and nothing allocated per call)
        return k
    if len(_TAPS_HOST) > 64:
        for dead in [kk for kk, e in _TAPS_HOST.items() if e[0]() is None] or list(_TAPS_HOST)[:32]:
            _TAPS_HOST.pop(dead, None)
    _TAPS_HOST[key] = (weakref.ref(base), kernel._version, k)
    return k


def fir_direct_forward(x: Tensor, kernel) -> Tensor:
    """Causal depthwise FIR, direct form: the ``conv_mode="direct"`` branch of ``FIR.forward``
    (``fir.py:556-568``).  ``x [C,T]``; ``kernel`` = the FLIPPED taps (the module's ``[1,1,K]`` buffer, any
    shape with K elements)."""
    return native.ops().fir_direct_forward(x, _kernel_host(kernel, x.dtype))


def fft_conv_forward(x: Tensor, kernel, padding: tuple[int, int] = (0, 0), epilogue: Epilogue | None = None) -> Tensor:
    """Overlap-save FFT convolution with ``fft_conv1d`` semantics (``_fftconv.py:70-141``) on ``x [C,T]``:
    returns ``[C, T + l + r - K + 1]``."""
    if epilogue is not None:
        y, epilogue.stat_value = native.ops().fft_conv_forward_ep(
            x, _kernel_host(kernel, x.dtype), int(padding[0]), int(padding[1]), epilogue.gain, epilogue.clamp,
            epilogue.stat_mode, epilogue.per_row)
        return y
    return native.ops().fft_conv_forward(x, _kernel_host(kernel, x.dtype), int(padding[0]), int(padding[1]))


def sos_fft_conv_supported(T: int, sos, taps: int, padding: tuple[int, int] = (0, 0), force_block: bool = False) -> bool:
    """Whether :func:`sos_fft_conv_forward` serves float32 rows of ``T`` samples with this cascade and tap count
    (``tfx_sos_fft_conv_supported``; host-only, no device needed)."""
    s = np.ascontiguousarray(_coeff(sos).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 6)
    return bool(L.load().tfx_sos_fft_conv_supported(
        ctypes.c_int64(int(T)), s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_int64(s.shape[0]),
        ctypes.c_int64(int(taps)), ctypes.c_int64(int(padding[0])), ctypes.c_int64(int(padding[1])), ctypes.c_int(int(force_block))))


def sos_fft_conv_plan_info(T: int, sos, taps: int, padding: tuple[int, int] = (0, 0), force_block: int = 0) -> dict | None:
    """Block length ``N``, hop ``S``, frames per row ``F`` and warm-up samples of :func:`sos_fft_conv_forward` for rows of
    ``T`` samples, or None where it does not serve the geometry (``tfx_sos_fft_conv_plan_info``; host-only)."""
    s = np.ascontiguousarray(_coeff(sos).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 6)
    n, h, f, w = (ctypes.c_int64(0) for _ in range(4))
    ok = L.load().tfx_sos_fft_conv_plan_info(ctypes.c_int64(int(T)), s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                             ctypes.c_int64(s.shape[0]), ctypes.c_int64(int(taps)), ctypes.c_int64(int(padding[0])),
                                             ctypes.c_int64(int(padding[1])), ctypes.c_int(int(force_block)), ctypes.byref(n),
                                             ctypes.byref(h), ctypes.byref(f), ctypes.byref(w))
    return {"N": n.value, "S": h.value, "F": f.value, "warmup": w.value, "workspace_bytes_held": workspace_bytes()} if ok else None


def clear_caches() -> None:
    """Drop every cached plan and hand all device workspaces back to their allocator (``tfx_clear_caches``): PyTorch's caching
    allocator under this module, so ``torch.cuda.empty_cache()`` afterwards returns them to the driver."""
    L.check(L.load().tfx_clear_caches())


def workspace_bytes() -> int:
    """Bytes of device workspace the library holds right now (overlap-save slabs, statistic partials ...; all streams and
    devices).  Under the torch module they come from PyTorch's caching allocator (``tfx_set_workspace_allocator``): they are
    part of ``torch.cuda.memory_allocated()`` and :func:`clear_caches` hands them back to it."""
    return int(L.load().tfx_workspace_bytes())


def sos_fft_conv_warmup(sos) -> int:
    """Samples a row's recursion starts early (from zero state) inside the column pass of :func:`sos_fft_conv_forward`."""
    s = np.ascontiguousarray(_coeff(sos).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 6)
    return int(L.load().tfx_sos_fft_conv_warmup(s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_int64(s.shape[0])))


def sos_fft_conv_forward(x: Tensor, sos, kernel, padding: tuple[int, int] = (0, 0), *, return_sections: bool = False,
                         force_block: int = 0, epilogue: Epilogue | None = None):
    """A zero-state SOS cascade followed by ``fft_conv1d`` as ONE overlap-save pipeline in the reference's arithmetic:
    float64 DF1 recursion (``_ops.py:119-176`` with ``state=None`` -> ``iir_cpu.cpp:64-159``), the downcast to float32
    (``iir.py:84-184``), float32 overlap-save (``_fftconv.py:70-141``).  The recursion runs inside the forward column
    pass of the transform.  ``x [C,T]`` float32; returns ``y [C, T+l+r-K+1]`` (+ the float64 output of every section
    ``[K,C,T]`` with ``return_sections``).  ``force_block``: 1 / 2 = take the 2^20 / 2^21-point block whatever the row length
    (tests at fixture size).  Raises when :func:`sos_fft_conv_supported` says no."""
    ep = epilogue if epilogue is not None else Epilogue()
    y, stat, sec = native.ops().sos_fft_conv_forward(
        x, _coeff(sos), _kernel_host(kernel, x.dtype), int(padding[0]), int(padding[1]), bool(return_sections),
        int(force_block), ep.gain, ep.clamp, ep.stat_mode, ep.per_row)
    if epilogue is not None:
        epilogue.stat_value = stat
    return (y, sec) if return_sections else y


def normalize_apply(x: Tensor, stat: Tensor, peak: float, mode: int = 0, per_row: bool = False) -> Tensor:
    """The apply half of ``Normalize`` on a raw statistic an epilogue left on the device (``stat``: float64
    ``[rows]`` or ``[1]``, max|x| for ``STAT_ABSMAX``, sum of squares for ``STAT_RMS``): one streaming pass."""
    return native.ops().normalize_apply(x, stat, float(peak), int(mode), bool(per_row))


def fir_stream_forward(x: Tensor, kernel, hist: Tensor | None, direct: bool = False) -> tuple[Tensor, Tensor]:
    """One chunk of a stateful FIR: ``x [C,T]`` continues the signal whose last ``K-1`` samples are ``hist
    [C,K-1]`` (``None`` = silence).  Returns ``(y [C,T], new_hist [C,K-1])``.  The kernels read history and
    chunk from their two buffers -- no concatenated copy of the chunk."""
    return native.ops().fir_stream_forward(x, _kernel_host(kernel, x.dtype), hist, bool(direct))


def chunk_supported(C: int, T: int, K: int, taps: int) -> bool:
    """Whether :func:`chunk_forward` takes a ``[C, T]`` chunk with ``K`` sections and ``taps`` FIR taps (host-only query)."""
    return bool(L.load().tfx_chunk_supported(int(C), int(T), int(K), int(taps)))


def chunk_forward(x: Tensor, sos, state_x: Tensor | None, state_y: Tensor | None, kernel, hist: Tensor | None,
                  gain: float | None = None, clamp: bool = False, *, precision=None):
    """ONE launch for one small streaming chunk ``x [C, T]`` (float32): SOS cascade with carried state -> stateful
    direct FIR (``kernel`` = flipped taps, ``hist [C, K-1]`` or None) -> ``* gain`` (None = no gain stage) and clip.
    Returns ``(y, new_state_x, new_state_y, new_hist)``; same arithmetic as ``sos_forward`` -> ``fir_stream_forward(direct)`` ->
    ``gain_forward`` (equal to float64 round-off of the recursion).  ``sos [K, 6]`` host float64 (``K`` may be 0), limits: :func:`chunk_supported`."""
    return native.ops().chunk_forward(x, _coeff(sos), state_x, state_y, _kernel_host(kernel, torch.float32), hist,
                                      1.0 if gain is None else float(gain), gain is not None, bool(clamp), _prec(precision))


def sum_forward(tensors: list[Tensor]) -> Tensor:
    """Sum of equally-shaped tensors in list order (``__base.py:1022-1026``)."""
    if not tensors:
        raise RuntimeError("sum_forward: need at least one tensor")
    return native.ops().sum_forward(list(tensors))


STAT_ABSMAX, STAT_RMS = 0, 1


def gain_forward(x: Tensor, gain: float, clamp: bool = False) -> Tensor:
    """``y = x * gain`` (+ clip to [-1, 1]) -- ``Gain.forward``, ``effect.py:361-383``; ``gain`` is the linear
    factor."""
    return native.ops().gain_forward(x, float(gain), bool(clamp))


def quantile_abs(x: Tensor, q: float) -> Tensor:
    """``torch.quantile(torch.abs(x), q, interpolation="linear")`` over all elements of a float32 signal as a three-pass radix
    select on the device (no sort, no 16 M element limit, no host sync): float64 ``[1]`` on the device, the same value
    torch.quantile returns wherever it runs; NaN anywhere in ``x`` -> NaN.  Feed it to :func:`normalize_apply`."""
    return native.ops().quantile_abs(x, float(q))


def stat_forward(x: Tensor, mode: int = STAT_ABSMAX, per_row: bool = False) -> Tensor:
    """``max|x|`` (``STAT_ABSMAX``) or ``sqrt(mean(x^2))`` (``STAT_RMS``) over everything, or per row of the
    ``[rows, T]`` view -- float64 on the device, no host sync."""
    return native.ops().stat_forward(x, int(mode), bool(per_row))


def normalize_forward(x: Tensor, peak: float, mode: int = STAT_ABSMAX, per_row: bool = False) -> Tensor:
    """``s > 0 ? x / s * peak : x`` with ``s`` = abs-max or RMS, global or per row of the ``[rows, T]`` view
    (``effect.py:696-698,719-721,775-786``); two streaming passes, statistic stays on device."""
    return native.ops().normalize_forward(x, float(peak), int(mode), bool(per_row))


def deinterleave_forward(frames: Tensor, out: Tensor | None = None, frame_base: int = 0,
                         scale: float = 1.0 / 32768.0) -> Tensor:
    """Interleaved ``[F, C]`` (float32, or int16 PCM scaled by ``scale``) -> planar float32 ``[C, F]`` (the
    device-side ``data_np.T.copy()`` of ``wave.py:448-452``).  With ``out`` ``[C, F_total]`` the chunk lands
    at frames ``[frame_base, frame_base + F)`` of every row."""
    if out is None:
        return native.ops().deinterleave_forward(frames, float(scale))
    native.ops().deinterleave_into(frames, out, int(frame_base), float(scale))
    return out


def interleave_forward(x: Tensor, frame_base: int = 0, frames: int | None = None) -> Tensor:
    """Planar float32 ``[C, F_total]`` -> interleaved ``[F, C]`` of frames ``[frame_base, frame_base+F)`` (the
    device-side ``.numpy().T`` of ``wave.py:566-573``)."""
    return native.ops().interleave_forward(x, int(frame_base), -1 if frames is None else int(frames))


# ---- host-only planning queries (no tensors, no GPU needed): ctypes over the C ABI --------------------
def sos_plan_info(sos) -> dict:
    """Host-side plan facts for an SOS matrix: warm-up halo length, the float32 error estimate and what
    ``precision='auto'`` would choose."""
    lib = L.load()
    s = np.ascontiguousarray(sos.detach().cpu().numpy() if isinstance(sos, Tensor) else sos, dtype=np.float64)
    if s.ndim != 2 or s.shape[-1] != 6:
        raise RuntimeError(f"expected [K, 6], got shape {tuple(s.shape)}")
    prec, warm, eb = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_double(0.0)
    L.check(lib.tfx_sos_plan_info(s.ctypes.data_as(ctypes.c_void_p), s.shape[0],
                                  ctypes.byref(prec), ctypes.byref(warm), ctypes.byref(eb)))
    return {"auto_precision": "f32" if prec.value == L.PREC_F32 else "f64",
            "warmup": warm.value, "f32_error_bound": eb.value}


def env_reload() -> None:
    """The library reads every ``TFX_*`` knob ONCE per process; after changing one in ``os.environ`` call this (``tfx_env_reload``)
    -- or start the process with ``TFX_ENV_DYNAMIC=1`` to make every lookup a fresh ``getenv``."""
    L.check(L.load().tfx_env_reload())


def prewarm(device=None) -> None:
    """Start the one-time per-device set-up of the overlap-save path on a helper thread (``tfx_prewarm``); returns at once."""
    lib = L.load()
    if device is not None and torch.device(device).type == "cuda":
        with torch.cuda.device(device):
            L.check(lib.tfx_prewarm())
    else:
        L.check(lib.tfx_prewarm())


def ols_plan_info(K: int, T: int, padding: tuple[int, int] = (0, 0), dtype: torch.dtype = torch.float32) -> dict:
    """Block geometry of the overlap-save op for a signal of `dtype` (FFT length N, hop S, blocks per row F) and the
    path that runs: "lds" (one launch, 4096-point transform in LDS), "passes" (three-pass four-step pipeline) or
    "rocfft"; `native` = a hand-written path; `bytes_per_sample` = modelled HBM traffic per output sample."""
    lib = L.load()
    n, s_, f, path = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    code = L.TFX_F64 if dtype == torch.float64 else L.TFX_F32
    L.check(lib.tfx_ols_plan_info2(int(K), int(T), int(padding[0]), int(padding[1]), code, ctypes.byref(n),
                                   ctypes.byref(s_), ctypes.byref(f), ctypes.byref(path)))
    esz = 8 if dtype == torch.float64 else 4
    bps = {2: esz * n.value / s_.value + esz, 1: (20.0 * n.value / s_.value + 4.0) * esz / 4, 0: 95.0 * esz / 4}[path.value]
    return {"N": n.value, "S": s_.value, "F": f.value, "native": path.value != 0,
            "path": ("rocfft", "passes", "lds")[path.value], "bytes_per_sample": bps}
