"""FIR filters on the HIP backend.

Reference: ``src/torchfx/filter/fir.py`` -- ``FIR`` (:510-579: taps rounded to float32,
stored flipped as buffer ``kernel [1,1,K]``; ``conv_mode`` "fft" (default) / "auto" (alias of
"fft") / "direct"; stateless; output keeps input shape and dtype) and ``DesignableFIR``
(:984-1021: ``scipy.signal.firwin`` window design, coefficients built once ``fs`` is known).
"""
from __future__ import annotations

from collections.abc import Sequence

import torch
from numpy.typing import ArrayLike
from scipy.signal import firwin
from torch import Tensor, nn

from torchfx_amd.filter._base import AbstractFilter


class FIR(AbstractFilter):
    """Causal FIR filter ``y = lfilter(b, [1], x)`` per channel."""

    def __init__(self, b: ArrayLike, conv_mode: str = "fft") -> None:
        super().__init__()
        self._init_taps(b, conv_mode)

    def _init_taps(self, b: ArrayLike, conv_mode: str) -> None:
        if conv_mode not in ("fft", "direct", "auto"):
            raise ValueError(f"conv_mode must be 'fft', 'direct', or 'auto', got {conv_mode!r}")
        self._conv_mode = conv_mode
        self.a = [1.0]
        taps = torch.tensor(b, dtype=torch.float32).flip(0)      # fir.py:516: float32, reversed
        self.register_buffer("kernel", taps.reshape(1, 1, -1))

    def compute_coefficients(self) -> None:
        """Nothing to design for explicit taps (kept for interface parity, ``fir.py:520-524``)."""

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None) -> Tensor:
        """``epilogue`` (``torchfx_ext.Epilogue``, planner-attached, FFT mode only): a following Gain / the
        reduction half of a following Normalize, applied where the overlap-save pass stores its output."""
        from torchfx_amd import torchfx_ext

        if x.ndim not in (1, 2, 3):
            raise ValueError("Input must be of shape [T], [C, T], or [B, C, T]")
        shape = x.shape
        rows = x.reshape(-1, shape[-1])                 # [T] -> [1,T]; [B,C,T] -> [B*C,T]
        taps = self.kernel.reshape(-1)
        if self._conv_mode == "direct":
            if epilogue is not None:
                raise RuntimeError("FIR: epilogues are attached to FFT-mode filters only")
            y = torchfx_ext.fir_direct_forward(rows, taps)
        elif epilogue is not None:
            y = torchfx_ext.fft_conv_forward(rows, taps, (taps.numel() - 1, 0), epilogue=epilogue)
        else:                                           # "fft" and its alias "auto" (fir.py:552)
            y = torchfx_ext.fft_conv_forward(rows, taps, (taps.numel() - 1, 0))
        return y.reshape(shape)


class DesignableFIR(FIR):
    """Window-method FIR (``firwin(num_taps, cutoff, fs=, pass_zero=, window=, scale=True)``).

    With ``fs=None`` the taps (and the ``nn.Module`` state holding them) are created by
    :meth:`compute_coefficients` once a ``Wave`` supplies the sample rate, like the
    reference (``fir.py:984-1005``).
    """

    def __init__(self, cutoff: float | Sequence[float], num_taps: int, fs: int | None = None,
                 pass_zero: bool = True, window: str = "hamming", conv_mode: str = "fft") -> None:
        # attributes first, without nn.Module machinery (it may not be initialised yet)
        object.__setattr__(self, "num_taps", num_taps)
        object.__setattr__(self, "cutoff", cutoff)
        object.__setattr__(self, "fs", fs)
        object.__setattr__(self, "pass_zero", pass_zero)
        object.__setattr__(self, "window", window)
        object.__setattr__(self, "_pending_conv_mode", conv_mode)
        object.__setattr__(self, "b", None)
        if fs is not None:
            self.compute_coefficients()
        else:
            nn.Module.__init__(self)     # usable in nn.Sequential / Wave.__or__ before fs is known
            if conv_mode not in ("fft", "direct", "auto"):
                raise ValueError(f"conv_mode must be 'fft', 'direct', or 'auto', got {conv_mode!r}")
            self._conv_mode = conv_mode

    @property
    def _has_computed_coeff(self) -> bool:
        return self.b is not None

    def compute_coefficients(self) -> None:
        assert self.fs is not None
        taps = firwin(self.num_taps, self.cutoff, fs=self.fs, pass_zero=self.pass_zero,
                      window=self.window, scale=True)
        if "_buffers" not in self.__dict__:
            nn.Module.__init__(self)
        if "kernel" in self._buffers:
            del self._buffers["kernel"]
        object.__setattr__(self, "b", taps)
        self._init_taps(taps, self._pending_conv_mode)

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None) -> Tensor:
        if self.b is None:
            if self.fs is None:
                raise ValueError("Sample rate (fs) must be set before filtering.")
            self.compute_coefficients()
        return super().forward(x, epilogue)
