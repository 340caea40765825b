"""``FX`` base class and the elementwise effects that sit between filters in a pipeline.

Reference: ``src/torchfx/effect.py`` -- ``FX.__or__`` (:253-258, builds a ``FilterChain``),
``Gain`` (:261-383) and ``Normalize`` with its strategy objects (:386-531, :534-790).  These are
SURVEY.md 8f rank 3: they are not filters, but a ``wave | iir | gain | fir`` pipeline passes through
them, so they run as streaming HIP passes (``csrc/effects.hip``) and a clamp-free ``Gain`` can be
folded into a neighbouring filter's coefficients by the ``Wave`` planner (opt-in, ``fuse_gain``).
``Reverb`` is the reference's one-tap feed-forward comb over ``delay_line_forward``; the BPM-synced
multi-tap ``Delay`` remains out of scope.
"""
from __future__ import annotations

import abc
import math
import typing as tp

import torch
from torch import Tensor, nn


class FX(nn.Module, abc.ABC):
    """Abstract base of every effect / filter: an ``nn.Module`` with ``forward(x)``."""

    def __init__(self) -> None:
        super().__init__()

    @abc.abstractmethod
    def forward(self, x: Tensor) -> Tensor: ...

    def __or__(self, other: nn.Module):
        # effect.py:253-258: NotImplemented for non-modules, else a flat chain
        if not isinstance(other, nn.Module):
            return NotImplemented
        from torchfx_amd.chain import FilterChain

        return FilterChain(self, other)


def _ext():
    from torchfx_amd import torchfx_ext

    return torchfx_ext


class Gain(FX):
    """Volume change: ``gain_type`` "amplitude" (factor), "db" (``10^(g/20)``) or "power"
    (``10*log10(g)`` dB); ``clamp=True`` clips the result to [-1, 1]  (``effect.py:261-383``)."""

    def __init__(self, gain: float, gain_type: str = "amplitude", clamp: bool = False) -> None:
        super().__init__()
        self.gain, self.gain_type, self.clamp = gain, gain_type, clamp
        if gain_type in ("amplitude", "power") and gain < 0:
            raise ValueError("If gain_type = amplitude or power, gain must be positive.")

    def linear_gain(self) -> float | None:
        """The factor the samples are multiplied by (None: unknown ``gain_type`` -> identity,
        0 dB -> identity as in ``_gain_db``, ``effect.py:132-136``)."""
        if self.gain_type == "amplitude":
            return float(self.gain)
        if self.gain_type == "db":
            db = self.gain
        elif self.gain_type == "power":
            db = 10 * math.log10(self.gain)
        else:
            return None
        return None if db == 0 else 10 ** (db / 20)

    @torch.no_grad()
    def forward(self, waveform: Tensor) -> Tensor:
        g = self.linear_gain()
        if g is None and not self.clamp:
            return waveform
        return _ext().gain_forward(waveform, 1.0 if g is None else g, self.clamp)


# ------------------------------------------------------------------------------- Normalize
class NormalizationStrategy(abc.ABC):
    """``(waveform, peak) -> waveform`` (``effect.py:534-611``)."""

    @abc.abstractmethod
    def __call__(self, waveform: Tensor, peak: float) -> Tensor: ...


class CustomNormalizationStrategy(NormalizationStrategy):
    """Wraps a user callable (``effect.py:614-675``); runs whatever the callable does."""

    def __init__(self, func: tp.Callable[[Tensor, float], Tensor]) -> None:
        assert callable(func), "func must be callable"
        self.func = func

    def __call__(self, waveform: Tensor, peak: float) -> Tensor:
        return self.func(waveform, peak)


class PeakNormalizationStrategy(NormalizationStrategy):
    """``x / max|x| * peak`` (unchanged if the signal is all zero) -- ``effect.py:678-698``."""

    def __call__(self, waveform: Tensor, peak: float) -> Tensor:
        return _ext().normalize_forward(waveform, peak, _ext().STAT_ABSMAX, per_row=False)


class RMSNormalizationStrategy(NormalizationStrategy):
    """``x / sqrt(mean(x^2)) * peak`` -- ``effect.py:700-721``."""

    def __call__(self, waveform: Tensor, peak: float) -> Tensor:
        return _ext().normalize_forward(waveform, peak, _ext().STAT_RMS, per_row=False)


class PercentileNormalizationStrategy(NormalizationStrategy):
    """``x / P_p(|x|) * peak`` (``effect.py:723-755``).  The percentile is a selection problem, not a sort: on float32 device
    signals it is a three-pass radix select (``torchfx_ext.quantile_abs`` -- the value ``torch.quantile(|x|, p / 100,
    interpolation="linear")`` returns, without its 16 M element limit), the threshold stays on the device and the scaling is the
    ``normalize_apply`` pass (unchanged signal when the threshold is not positive, as in the reference).  Other dtypes take
    ``torch.quantile`` like the reference."""

    def __init__(self, percentile: float = 99.0) -> None:
        assert 0 < percentile <= 100, "Percentile must be between 0 and 100."
        self.percentile = percentile

    def __call__(self, waveform: Tensor, peak: float) -> Tensor:
        if waveform.dtype == torch.float32 and waveform.numel() > 0:
            E = _ext()
            threshold = E.quantile_abs(waveform, self.percentile / 100)
            return E.normalize_apply(waveform, threshold, peak, E.STAT_ABSMAX, False)
        threshold = torch.quantile(torch.abs(waveform), self.percentile / 100, interpolation="linear")
        return waveform / threshold * peak if threshold > 0 else waveform


class PerChannelNormalizationStrategy(NormalizationStrategy):
    """Every channel to its own peak (``effect.py:757-786``): ``[C,T]`` or ``[B,C,T]``."""

    def __call__(self, waveform: Tensor, peak: float) -> Tensor:
        assert waveform.ndim >= 2, "Waveform must have at least 2 dimensions (channels, time)."
        if waveform.ndim not in (2, 3):
            raise ValueError("Waveform must have shape (C, T) or (B, C, T)")
        return _ext().normalize_forward(waveform, peak, _ext().STAT_ABSMAX, per_row=True)


class Normalize(FX):
    """Normalise to ``peak`` with a pluggable strategy, default peak (``effect.py:386-531``)."""

    def __init__(self, peak: float = 1.0,
                 strategy: NormalizationStrategy | tp.Callable[[Tensor, float], Tensor] | None = None) -> None:
        super().__init__()
        assert peak > 0, "Peak value must be positive."
        self.peak = peak
        if callable(strategy) and not isinstance(strategy, NormalizationStrategy):
            strategy = CustomNormalizationStrategy(strategy)
        self.strategy = strategy or PeakNormalizationStrategy()
        if not isinstance(self.strategy, NormalizationStrategy):
            raise TypeError("Strategy must be an instance of NormalizationStrategy.")

    @torch.no_grad()
    def forward(self, waveform: Tensor) -> Tensor:
        return self.strategy(waveform, self.peak)


class Epilogued(FX):
    """Planner product (``Wave.plan()``, ``fuse_epilogue``): a filter followed by ``Gain`` and / or ``Normalize``
    whose elementwise work rides on the filter's own kernel.  ``producer`` is an SOS filter / cascade or an
    FFT-mode FIR; its kernel multiplies and clips the samples it stores exactly like the ``Gain`` pass would
    (bit-identical) and gathers the statistic ``Normalize`` needs, so the run costs the filter's 8 B/sample
    plus, with a ``Normalize``, one apply pass -- instead of 8 + 8 + 4 + 8.  Nothing on the member modules is
    modified; a lone stateful IIR keeps carrying its state on the user's object."""

    def __init__(self, producer: nn.Module, gain: "Gain | None", norm: "Normalize | None") -> None:
        super().__init__()
        self.producer, self.gain, self.norm = producer, gain, norm

    @staticmethod
    def norm_kind(norm: "Normalize") -> tuple[str, bool] | None:
        """(statistic, per_row) of the strategies with a streaming reduction; None for the others."""
        st = norm.strategy
        if type(st) is PeakNormalizationStrategy:
            return "absmax", False
        if type(st) is RMSNormalizationStrategy:
            return "sumsq", False
        if type(st) is PerChannelNormalizationStrategy:
            return "absmax", True
        return None

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        E = _ext()
        g = 1.0 if self.gain is None else self.gain.linear_gain()
        kind = self.norm_kind(self.norm) if self.norm is not None else None
        if kind is not None and kind[1]:
            assert x.ndim >= 2, "Waveform must have at least 2 dimensions (channels, time)."
            if x.ndim not in (2, 3):
                raise ValueError("Waveform must have shape (C, T) or (B, C, T)")
        ep = E.Epilogue(gain=1.0 if g is None else g, clamp=bool(self.gain is not None and self.gain.clamp),
                        stat=None if kind is None else kind[0], per_row=bool(kind and kind[1]))
        y = self.producer(x, epilogue=ep)
        if kind is not None:
            y = E.normalize_apply(y, ep.stat_value, self.norm.peak, E.STAT_ABSMAX if kind[0] == "absmax" else E.STAT_RMS, kind[1])
        return y


class Reverb(FX):
    """``y[n] = x[n] + mix * decay * x[n - delay]`` -- the reference's "reverb" is one feed-forward
    tap (``effect.py:789-931`` over ``delay_line_forward``, ``_csrc/cpu/delay_cpu.cpp:17-41``); a
    signal not longer than the delay is returned unchanged (the same tensor)."""

    def __init__(self, delay: int = 4410, decay: float = 0.5, mix: float = 0.5) -> None:
        super().__init__()
        assert delay > 0, "Delay must be positive."
        assert 0 < decay < 1, "Decay must be between 0 and 1."
        assert 0 <= mix <= 1, "Mix must be between 0 and 1."
        self.delay, self.decay, self.mix = delay, decay, mix

    @torch.no_grad()
    def forward(self, waveform: Tensor) -> Tensor:
        if waveform.size(-1) <= self.delay:
            return waveform
        from torchfx_amd._ops import delay_line_forward

        return delay_line_forward(waveform, self.delay, self.decay, self.mix)
