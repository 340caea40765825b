"""oracle/oracle.py -- CPU restatement of the torchfx.filter hot path (checker).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, from
``__graft_entry__.smoke()`` and from the ``cpu_baseline`` leg of ``bench.py``;
the product package ``torchfx_amd`` never imports it (tests/test_no_oracle_in_product.py
enforces that).  Parity status: PINNED -- ``oracle/make_golden.py`` validated
every function here against the real reference in the build container and the
committed vectors under ``tests/golden/`` re-check it everywhere
(``tests/test_oracle_golden.py``).

numpy + a small plain-C library (``oracle.c``, built by ``oracle/Makefile``).
Citations are relative to ``/root/reference/``.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def set_threads(n: int) -> int:
    """Pin the OpenMP thread count of the C oracle (returns the count now in force)."""
    lib = _lib()
    lib.oracle_set_threads.restype = ctypes.c_int
    return int(lib.oracle_set_threads(ctypes.c_int(int(n))))


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64


# --------------------------------------------------------------------------- IIR
def sos_forward(x, sos, state_x=None, state_y=None, sections=False):
    """Reference IIR call path (``_ops.py:119-176`` -> ``iir_cpu.cpp:64-159``).

    ``x`` is ``[C,T]`` float32 or float64; arithmetic is float64 either way
    (``_ops.py:149``).  Returns ``(y_f64, new_state_x, new_state_y[, y_sections])``
    exactly like ``torchfx_ext.sos_forward``: *y is float64*; the caller applies
    the ``out.to(x.dtype)`` of ``iir.py:176``.
    """
    x = np.ascontiguousarray(x)
    assert x.ndim == 2
    C, T = x.shape
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    K = sos.shape[0]
    sx = np.zeros((K, C, 2)) if state_x is None else np.array(state_x, dtype=np.float64, copy=True)
    sy = np.zeros((K, C, 2)) if state_y is None else np.array(state_y, dtype=np.float64, copy=True)
    xd = x.astype(np.float64)
    y = np.empty_like(xd)
    ysec = np.empty((K, C, T)) if sections else None
    _lib().oracle_sos_df1_f64(_p(xd), _p(y), _i64(C), _i64(T), _p(sos), _i64(K),
                              _p(sx), _p(sy), _p(ysec))
    return (y, sx, sy, ysec) if sections else (y, sx, sy)


def biquad_forward(x, b, a1, a2, state_x=None, state_y=None):
    """``biquad_forward_cpu`` (``iir_cpu.cpp:10-62``); states ``[C,2]``."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    C, T = x.shape
    b = np.ascontiguousarray(b, dtype=np.float64)
    sx = np.zeros((C, 2)) if state_x is None else np.array(state_x, dtype=np.float64, copy=True)
    sy = np.zeros((C, 2)) if state_y is None else np.array(state_y, dtype=np.float64, copy=True)
    y = np.empty_like(x)
    _lib().oracle_biquad_df1_f64(_p(x), _p(y), _i64(C), _i64(T), _p(b),
                                 ctypes.c_double(a1), ctypes.c_double(a2), _p(sx), _p(sy))
    return y, sx, sy


def iir_module_forward(x, sos, state_x=None, state_y=None):
    """``_sos_cascade_forward`` (``iir.py:84-184``) for a ``[C,T]`` signal:
    output keeps the input dtype (``iir.py:176``)."""
    y, sx, sy = sos_forward(x, sos, state_x, state_y)
    return y.astype(np.asarray(x).dtype), sx, sy


# --------------------------------------------------------------------------- FIR
def flipped_kernel(b) -> np.ndarray:
    """``FIR.__init__`` (``fir.py:516-518``): taps rounded to float32, flipped."""
    return np.asarray(b, dtype=np.float32)[::-1].copy()


def fir_direct(x, kernel):
    """conv_mode="direct" (``fir.py:556-568``): causal depthwise correlation with
    the flipped kernel, arithmetic in the input dtype."""
    x = np.ascontiguousarray(x)
    C, T = x.shape
    K = kernel.shape[-1]
    y = np.empty_like(x)
    if x.dtype == np.float32:
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        _lib().oracle_fir_direct_f32(_p(x), _p(y), _i64(C), _i64(T), _p(k), _i64(K))
    else:
        k = np.ascontiguousarray(kernel, dtype=np.float64)
        _lib().oracle_fir_direct_f64(_p(x), _p(y), _i64(C), _i64(T), _p(k), _i64(K))
    return y


def fft_conv1d(x, kernel, padding=(0, 0), block_ratio=5.0, threads=1):
    """Overlap-save, restating ``_fftconv.py:70-141`` line by line in numpy
    (numpy >= 2 transforms float32 in float32, like ``torch.fft``).
    ``threads`` > 1 only spreads the independent channels over a thread pool (bench.py's
    all-cores CPU baseline); the arithmetic per channel is unchanged.

    ``x`` ``[C,T]`` (the reference's ``[B,C,T]`` with B folded into C),
    ``kernel`` ``[K]`` *flipped* taps; returns ``[C, T+l+r-K+1]``.
    """
    x = np.asarray(x)
    kernel = np.asarray(kernel, dtype=x.dtype).reshape(-1)
    x = np.pad(x, ((0, 0), (int(padding[0]), int(padding[1]))))        # :107
    C, length = x.shape
    K = kernel.shape[-1]
    if length < K:                                                       # :111-115
        raise RuntimeError(
            f"Input should be at least as large as the kernel size {K}, "
            f"but it is only {length} samples long.")
    if block_ratio < 1:                                                  # :116-117
        raise RuntimeError("Block ratio must be greater than 1.")
    block = min(int(K * block_ratio), length)                            # :119
    hop = block - K + 1                                                  # :120
    kz = np.fft.rfft(np.pad(kernel, (0, block - K)))                     # :123-124
    n_frames = math.ceil((max(length, block) - block) / hop) + 1         # unfold :52
    tgt = (n_frames - 1) * hop + block
    xp = np.pad(x, ((0, 0), (0, tgt - length)))
    out = np.empty((C, n_frames * hop), dtype=x.dtype)
    def one(c):                  # channel by channel to bound memory (same arithmetic)
        frames = np.lib.stride_tricks.as_strided(
            xp[c], shape=(n_frames, block), strides=(xp.strides[1] * hop, xp.strides[1]))
        fz = np.fft.rfft(frames, axis=-1)                                # :130
        oz = fz * np.conj(kz)                                            # :131
        o = np.fft.irfft(oz, n=block, axis=-1)                           # :132
        out[c] = o[:, :hop].reshape(-1)                                  # :135-136

    if threads > 1 and C > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(threads)) as ex:
            list(ex.map(one, range(C)))
    else:
        for c in range(C):
            one(c)
    return out[:, : length - K + 1]                                      # :139-140


def fir_forward(x, kernel, conv_mode="fft", threads=1):
    """``FIR.forward`` (``fir.py:526-579``) on ``[C,T]``."""
    K = kernel.shape[-1]
    if conv_mode in ("fft", "auto"):
        return fft_conv1d(x, kernel, padding=(K - 1, 0), threads=threads)
    return fir_direct(x, kernel)


def delay_line(x, delay, decay, mix):
    """``delay_cpu.cpp:43-85`` (float32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    C, T = x.shape
    if T <= delay:
        return x
    y = np.empty_like(x)
    _lib().oracle_delay_line_f32(_p(x), _p(y), _i64(C), _i64(T), _i64(delay),
                                 ctypes.c_float(mix * decay))
    return y


# --------------------------------------------------------------------------- effects
def gain_linear(gain, gain_type="amplitude"):
    """The factor ``Gain.forward`` multiplies by (``effect.py:132-136,372-378``); None = identity."""
    if gain_type == "amplitude":
        return float(gain)
    if gain_type == "db":
        db = gain
    elif gain_type == "power":
        db = 10 * math.log10(gain)
    else:
        return None
    return None if db == 0 else 10 ** (db / 20)


def gain(x, gain, gain_type="amplitude", clamp=False):
    """``Gain.forward`` (``effect.py:361-383``) in the signal dtype."""
    x = np.asarray(x)
    g = gain_linear(gain, gain_type)
    y = x if g is None else x * x.dtype.type(g)
    return np.clip(y, -1.0, 1.0).astype(x.dtype) if clamp else y


def normalize(x, peak, strategy="peak", percentile=99.0):
    """The built-in normalization strategies (``effect.py:696-698,719-721,750-753,775-786``),
    signal-dtype arithmetic in the reference's order: ``x / s * peak`` when ``s > 0``."""
    x = np.asarray(x)
    dt = x.dtype.type
    if strategy == "per_channel":
        m = np.max(np.abs(x), axis=-1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(m > 0, x / m * dt(peak), x).astype(x.dtype)
    if strategy == "peak":
        s = np.max(np.abs(x))
    elif strategy == "rms":
        s = np.sqrt(np.mean(x * x, dtype=x.dtype))
    elif strategy == "percentile":
        s = np.quantile(np.abs(x), percentile / 100).astype(x.dtype)
    else:
        raise ValueError(strategy)
    return (x / dt(s) * dt(peak)).astype(x.dtype) if s > 0 else x


# --------------------------------------------------------------------------- chain
def chain_forward(x, sos, fir_kernels, threads=1):
    """The BASELINE cfg-5 pipe ``wave | iir... | FIR | FIR`` as the reference
    runs it: one fused SOS cascade (``wave.py:207-239``), then each FIR in its
    default fft mode (``fir.py:552-555``), all on a float32 ``[C,T]`` signal."""
    y, _, _ = iir_module_forward(x, sos)
    for k in fir_kernels:
        y = fir_forward(y, k, "fft", threads=threads)
    return y
