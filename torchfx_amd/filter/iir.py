"""IIR filters as second-order-section cascades on the HIP backend.

Reference: ``src/torchfx/filter/iir.py`` -- ``_sos_cascade_forward`` (:84-184), the ``IIR``
base (:187-265) and the design classes (Butterworth :366-385, Chebyshev1 :493-519,
Chebyshev2 :624-650, shelves :1087-1270, ParametricEQ/Peaking :1430-1511, Notch :1626-1663,
AllPass :1751-1790, LinkwitzRiley :1919-1968, Elliptic :2213-2242 and their Hi*/Lo*
shorthands).  Every class reduces to "produce a ``[K,6]`` float64 SOS on the host"; one
GPU kernel serves them all.
"""
from __future__ import annotations

import abc
import math

import numpy as np
import torch
from torch import Tensor

from torchfx_amd.filter import _design
from torchfx_amd.filter._base import AbstractFilter
from torchfx_amd.filter.biquad import Biquad

NONE_FS_ERR = "Sample rate of the filter could not be None."


def _sos_cascade_forward(
    x: Tensor,
    sos_canonical: Tensor,
    sos_device_cache: Tensor | None,
    state_x: Tensor | None,
    state_y: Tensor | None,
    epilogue=None,
) -> tuple[Tensor, Tensor | None, Tensor | None, Tensor | None]:
    """Shared stateful SOS forward: shape, state and dtype rules of ``iir.py:84-184``.

    * ``[T]`` / ``[C,T]`` / ``[B,C,T]`` are processed as ``[B*C, T]`` rows (:119-126);
    * state is (re)zeroed when missing or when the row count changed (:135-138) and follows
      the signal's device (:139-142);
    * the result has the input's dtype and shape (:176-184).

    Unlike the reference there is no float64 copy of the signal and no separate K == 1 fast
    path (:149-172): the HIP cascade kernel handles K = 1 and writes the input dtype itself.
    ``sos_device_cache`` is passed through untouched (the coefficients travel as kernel
    tables built from the canonical host copy, so there is nothing to cache on the device).
    ``epilogue`` (``torchfx_ext.Epilogue``): a following Gain / the reduction half of a following
    Normalize, applied by the cascade kernel to the samples it stores (planner-attached).
    """
    from torchfx_amd._ops import parallel_iir_forward

    shape = x.shape
    if x.ndim == 1:
        rows = x.unsqueeze(0)
    elif x.ndim == 3:
        rows = x.reshape(shape[0] * shape[1], shape[2])
    else:
        rows = x
    n_rows = rows.shape[0]
    n_sec = sos_canonical.shape[0]

    if state_x is None or state_y is None or state_x.shape[1] != n_rows:
        state_x = state_y = None                     # kernel treats None as zeros
    elif state_x.device != rows.device:
        state_x, state_y = state_x.to(rows.device), state_y.to(rows.device)

    out, state_x, state_y = parallel_iir_forward(
        rows, sos_canonical, state_x, state_y, sos_cpu=sos_canonical, out_dtype=x.dtype, epilogue=epilogue)
    assert state_x.shape == (n_sec, n_rows, 2)
    return out.reshape(shape), sos_device_cache, state_x, state_y


class IIR(AbstractFilter):
    """Base of all SOS-cascade IIR filters: Direct Form I per section, state carried across
    calls until :meth:`reset_state` (reference ``iir.py:187-265``)."""

    fs: int | None
    cutoff: float

    @abc.abstractmethod
    def __init__(self, fs: int | None = None) -> None:
        super().__init__()
        self.fs = fs
        self._sos: Tensor | None = None               # [K,6] float64, host
        self._sos_device_cache: Tensor | None = None
        self._state_x: Tensor | None = None           # [K,C,2] float64
        self._state_y: Tensor | None = None

    def _set_sos(self, sos: np.ndarray) -> None:
        self._sos = torch.from_numpy(np.ascontiguousarray(sos, dtype=np.float64))

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None) -> Tensor:
        if self.fs is None:
            raise ValueError(NONE_FS_ERR)
        if self._sos is None:
            self.compute_coefficients()
            self._sos_device_cache = None
        result, self._sos_device_cache, self._state_x, self._state_y = _sos_cascade_forward(
            x, self._sos, self._sos_device_cache, self._state_x, self._state_y, epilogue)
        return result

    def reset_state(self) -> None:
        """Drop the carried state.  Like the reference (``iir.py:255-265``) this also drops the
        designed SOS, forcing a redesign on the next forward."""
        self._state_x = self._state_y = None
        self._sos = None
        self._sos_device_cache = None


# --------------------------------------------------------------------------- classical designs
class _Classic(IIR):
    """SciPy-designed prototype; subclasses name the family and its extra parameters."""

    _family = ""

    def _design_kwargs(self) -> dict:
        return {}

    def compute_coefficients(self) -> None:
        assert self.fs is not None
        self._set_sos(_design.classic_sos(self._family, self.order, self.cutoff, self.fs,
                                          self.btype, **self._design_kwargs()))


def _order(order: int, scale: str) -> int:
    # "db" means dB/octave: 6 dB per pole (iir.py:378)
    return order if scale == "linear" else order // 6


class Butterworth(_Classic):
    _family = "butter"

    def __init__(self, btype: str, cutoff: float, order: int = 4, order_scale: str = "linear",
                 fs: int | None = None) -> None:
        super().__init__(fs)
        self.btype, self.cutoff, self.order = btype, cutoff, _order(order, order_scale)


class Chebyshev1(_Classic):
    _family = "cheby1"

    def __init__(self, btype: str, cutoff: float, order: int = 4, ripple: float = 0.1,
                 fs: int | None = None) -> None:
        super().__init__(fs)
        self.btype, self.cutoff, self.order, self.ripple = btype, cutoff, order, ripple

    def _design_kwargs(self) -> dict:
        return {"ripple": self.ripple}


class Chebyshev2(Chebyshev1):
    _family = "cheby2"


class Elliptic(_Classic):
    _family = "ellip"

    def __init__(self, btype: str, cutoff: float, order: int = 4, passband_ripple: float = 0.1,
                 stopband_attenuation: float = 40, fs: int | None = None) -> None:
        super().__init__(fs)
        self.btype, self.cutoff, self.order = btype, cutoff, order
        self.passband_ripple, self.stopband_attenuation = passband_ripple, stopband_attenuation

    def _design_kwargs(self) -> dict:
        return {"rp": self.passband_ripple, "rs": self.stopband_attenuation}


class LinkwitzRiley(IIR):
    """Two identical Butterworth filters of half the order in series (``iir.py:1949-1968``)."""

    def __init__(self, btype: str, cutoff: float, order: int = 4, order_scale: str = "linear",
                 fs: int | None = None) -> None:
        super().__init__(fs)
        self.order = _order(order, order_scale)
        if order <= 0 or order % 2 != 0:      # the reference validates the unscaled argument (:1943-1945)
            raise ValueError("Linkwitz-Riley filter order must be a positive even integer.")
        self.btype, self.cutoff = btype, cutoff

    def compute_coefficients(self) -> None:
        assert self.fs is not None
        half = _design.classic_sos("butter", self.order // 2, self.cutoff, self.fs, self.btype)
        self._set_sos(np.vstack([half, half]))


def _band(base: type, btype: str, name: str, default_order: int | None = None) -> type:
    """Hi*/Lo* shorthand: the base class with ``btype`` fixed (e.g. ``iir.py:825-922``)."""

    def __init__(self, cutoff: float, *args, **kwargs) -> None:
        if default_order is not None and not args and "order" not in kwargs:
            kwargs["order"] = default_order
        base.__init__(self, btype, cutoff, *args, **kwargs)

    return type(name, (base,), {"__init__": __init__, "__doc__": f"{btype} {base.__name__}.",
                                "__module__": __name__})


# LoButterworth / HiButterworth default to order 5 (iir.py:868,918); all others to 4
HiButterworth = _band(Butterworth, "highpass", "HiButterworth", 5)
LoButterworth = _band(Butterworth, "lowpass", "LoButterworth", 5)
HiChebyshev1 = _band(Chebyshev1, "highpass", "HiChebyshev1")
LoChebyshev1 = _band(Chebyshev1, "lowpass", "LoChebyshev1")
HiChebyshev2 = _band(Chebyshev2, "highpass", "HiChebyshev2")
LoChebyshev2 = _band(Chebyshev2, "lowpass", "LoChebyshev2")
HiElliptic = _band(Elliptic, "highpass", "HiElliptic")
LoElliptic = _band(Elliptic, "lowpass", "LoElliptic")
HiLinkwitzRiley = _band(LinkwitzRiley, "highpass", "HiLinkwitzRiley")
LoLinkwitzRiley = _band(LinkwitzRiley, "lowpass", "LoLinkwitzRiley")


# --------------------------------------------------------------------------- cookbook sections
def _linear_gain(gain: float, scale: str) -> float:
    return gain if scale == "linear" else 10 ** (gain / 20)


class Shelving(Biquad):
    """Common base of the shelving filters (``iir.py:925-990``)."""

    _kind = ""

    def __init__(self, cutoff: float, q: float, fs: int | None = None) -> None:
        super().__init__(cutoff=cutoff, q=q, fs=fs)

    @property
    def _omega(self) -> float:
        if self.fs is None:
            raise ValueError(NONE_FS_ERR)
        return 2.0 * math.pi * self.cutoff / self.fs

    @property
    def _alpha(self) -> float:
        return math.sin(self._omega) / (2.0 * self.q)

    def compute_coefficients(self) -> None:
        if self.fs is None:
            raise ValueError(NONE_FS_ERR)
        self._sos = torch.from_numpy(_design.rbj_div(self._kind, self.cutoff, self.q, self.fs, self.gain))
        self._sos_device_cache = None


class HiShelving(Shelving):
    _kind = "highshelf"

    def __init__(self, cutoff: float, q: float, gain: float, gain_scale: str = "linear",
                 fs: int | None = None) -> None:
        super().__init__(cutoff=cutoff, q=q, fs=fs)
        self.gain = _linear_gain(gain, gain_scale)


class LoShelving(HiShelving):
    _kind = "lowshelf"


class ParametricEQ(Biquad):
    """Peaking EQ; ``gain`` in dB (``iir.py:1430-1462``)."""

    def __init__(self, frequency: float, q: float, gain: float, fs: int | None = None) -> None:
        super().__init__(cutoff=frequency, q=q, fs=fs)
        self.gain_db = gain
        self.gain = 10 ** (gain / 20)

    def compute_coefficients(self) -> None:
        assert self.fs is not None
        self._sos = torch.from_numpy(_design.rbj_div("peaking", self.cutoff, self.q, self.fs, self.gain))
        self._sos_device_cache = None


class Peaking(ParametricEQ):
    """ParametricEQ with a linear-or-dB gain argument; non-positive linear gain maps to 0 dB
    (``iir.py:1511``)."""

    def __init__(self, cutoff: float, q: float, gain: float, gain_scale: str, fs: int | None = None) -> None:
        if gain_scale == "db":
            gain_db = gain
        else:
            gain_db = 20 * math.log10(gain) if gain > 0 else 0
        super().__init__(frequency=cutoff, q=q, gain=gain_db, fs=fs)


class Notch(Biquad):
    _rbj = "notch"


class AllPass(Biquad):
    _rbj = "allpass"
