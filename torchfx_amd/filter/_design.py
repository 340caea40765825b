"""Host-side coefficient design (float64 scalars -> ``[K,6]`` SOS rows).

SciPy does the classical prototypes (it is a third-party dependency of the reference too:
``scipy.signal.butter/cheby1/cheby2/ellip`` at ``filter/iir.py:73``); the second-order
"cookbook" sections follow R. Bristow-Johnson's Audio-EQ-Cookbook, evaluated in the same
floating-point order as the reference so the rows are bit-identical
(``filter/biquad.py:269-509``, ``filter/iir.py:1098-1119,1249-1270,1441-1462,1634-1663,
1759-1790``; pinned by ``tests/golden/designs.npz``).
"""
from __future__ import annotations

import math

import numpy as np
from scipy import signal as _sig

_CLASSIC = {
    "butter": lambda order, wn, btype, **kw: _sig.butter(order, wn, btype=btype, output="sos"),
    "cheby1": lambda order, wn, btype, ripple, **kw: _sig.cheby1(order, ripple, wn, btype=btype, output="sos"),
    "cheby2": lambda order, wn, btype, ripple, **kw: _sig.cheby2(order, ripple, wn, btype=btype, output="sos"),
    "ellip": lambda order, wn, btype, rp, rs, **kw: _sig.ellip(order, rp, rs, wn, btype=btype, output="sos"),
}


def classic_sos(family: str, order: int, cutoff: float, fs: int, btype: str, **kw) -> np.ndarray:
    """SOS of a classical IIR prototype; ``cutoff`` in Hz, normalised as ``cutoff/(fs/2)``."""
    wn = cutoff / (0.5 * fs)
    return np.asarray(_CLASSIC[family](order, wn, btype, **kw), dtype=np.float64)


def _trig(f0: float, q: float, fs: int) -> tuple[float, float]:
    w0 = 2.0 * math.pi * f0 / fs
    return math.cos(w0), math.sin(w0) / (2.0 * q)


def _row(b0, b1, b2, a1, a2) -> np.ndarray:
    return np.array([[b0, b1, b2, 1.0, a1, a2]], dtype=np.float64)


def rbj_recip(kind: str, f0: float, q: float, fs: int) -> np.ndarray:
    """Second-order sections normalised by multiplying with 1/a0."""
    c, alpha = _trig(f0, q, fs)
    g = 1.0 / (1.0 + alpha)              # 1 / a0
    a1 = -2.0 * c * g
    a2 = (1.0 - alpha) * g
    if kind == "lpf":
        m = (1.0 - c) * g
        return _row(m / 2.0, m, m / 2.0, a1, a2)
    if kind == "hpf":
        m = (1.0 + c) * g
        return _row(m / 2.0, -m, m / 2.0, a1, a2)
    if kind == "notch":
        return _row(g, a1, g, a1, a2)
    if kind == "bpf":                    # constant 0 dB peak gain
        return _row(alpha * g, 0.0, -alpha * g, a1, a2)
    if kind == "bpf_peak":               # constant skirt gain, peak gain = Q
        return _row(q * alpha * g, 0.0, -q * alpha * g, a1, a2)
    if kind == "allpass":
        return _row(a2, a1, 1.0, a1, a2)
    raise ValueError(f"unknown biquad kind {kind!r}")


def rbj_div(kind: str, f0: float, q: float, fs: int, A: float) -> np.ndarray:  # noqa: N803
    """Gain-dependent sections (peaking / shelves), normalised by dividing by a0."""
    c, alpha = _trig(f0, q, fs)
    if kind == "peaking":
        num = (1 + alpha * A, -2 * c, 1 - alpha * A)
        den = (1 + alpha / A, -2 * c, 1 - alpha / A)
    elif kind in ("highshelf", "lowshelf"):
        r = math.sqrt(A)
        sgn = 1.0 if kind == "highshelf" else -1.0      # sign of the (A-1)cos term in b
        p, m = (A + 1), (A - 1)
        num = (A * (p + sgn * m * c + 2 * r * alpha),
               -sgn * 2 * A * (m + sgn * p * c),
               A * (p + sgn * m * c - 2 * r * alpha))
        den = (p - sgn * m * c + 2 * r * alpha,
               sgn * 2 * (m - sgn * p * c),
               p - sgn * m * c - 2 * r * alpha)
    else:
        raise ValueError(f"unknown biquad kind {kind!r}")
    a0 = den[0]
    return _row(num[0] / a0, num[1] / a0, num[2] / a0, den[1] / a0, den[2] / a0)
