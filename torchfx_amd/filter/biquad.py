"""Single second-order sections (RBJ cookbook types) on the HIP backend.

Reference: ``src/torchfx/filter/biquad.py`` -- ``Biquad`` base (:42-236: a 1x6 SOS plus
DF1 state, ``b``/``a`` views, ``reset_state`` that keeps the coefficients) and the six
concrete types (:269-509).  The K = 1 cascade kernel does the work.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from torchfx_amd.filter import _design
from torchfx_amd.filter._base import AbstractFilter


class Biquad(AbstractFilter):
    """One second-order section with persistent Direct-Form-I state."""

    _rbj = ""          # cookbook kind for gain-free sections

    def __init__(self, cutoff: float, q: float, fs: int | None = None) -> None:
        super().__init__()
        self.cutoff, self.q, self.fs = cutoff, q, fs
        self._sos: Tensor | None = None
        self._sos_device_cache: Tensor | None = None
        self._state_x: Tensor | None = None
        self._state_y: Tensor | None = None

    # (b, a) views of the single SOS row (biquad.py:94-121)
    @property
    def b(self) -> Tensor | None:
        return None if self._sos is None else self._sos[0, :3]

    @property
    def a(self) -> Tensor | None:
        if self._sos is None:
            return None
        return torch.tensor([1.0, float(self._sos[0, 4]), float(self._sos[0, 5])], dtype=torch.float64)

    def _set_coefficients(self, b0: float, b1: float, b2: float, a1: float, a2: float) -> None:
        self._sos = torch.tensor([[b0, b1, b2, 1.0, a1, a2]], dtype=torch.float64)
        self._sos_device_cache = None

    def compute_coefficients(self) -> None:
        if not self._rbj:
            raise NotImplementedError("Biquad subclasses design their own coefficients")
        assert self.fs is not None
        self._sos = torch.from_numpy(_design.rbj_recip(self._rbj, self.cutoff, self.q, self.fs))
        self._sos_device_cache = None

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None) -> Tensor:
        if self.fs is None:
            raise ValueError("Sample rate (fs) must be set before filtering.")
        if self._sos is None:
            self.compute_coefficients()
            self._sos_device_cache = None
        from torchfx_amd.filter.iir import _sos_cascade_forward

        result, self._sos_device_cache, self._state_x, self._state_y = _sos_cascade_forward(
            x, self._sos, self._sos_device_cache, self._state_x, self._state_y, epilogue)
        return result

    def reset_state(self) -> None:
        """Clear the carried state; coefficients stay (``biquad.py:198-206``)."""
        self._state_x = self._state_y = None
        self._sos_device_cache = None

    @staticmethod
    def _compute_omega_alpha(cutoff: float, q: float, fs: int) -> tuple[float, float, float]:
        w0 = 2.0 * math.pi * cutoff / fs
        return math.sin(w0), math.cos(w0), math.sin(w0) / (2.0 * q)


class BiquadLPF(Biquad):
    _rbj = "lpf"


class BiquadHPF(Biquad):
    _rbj = "hpf"


class BiquadBPF(Biquad):
    _rbj = "bpf"


class BiquadBPFPeak(Biquad):
    _rbj = "bpf_peak"


class BiquadNotch(Biquad):
    _rbj = "notch"


class BiquadAllPass(Biquad):
    _rbj = "allpass"
