"""Filter base classes: lazy coefficients, ``+`` (parallel sum).

Reference: ``src/torchfx/filter/__base.py`` -- ``AbstractFilter`` (:22-739, the parts with
behaviour: ``_has_computed_coeff``, ``__add__``/``__radd__``) and
``ParallelFilterCombination`` (:742-1026).  The sum of the branches is done by one HIP
kernel (``tfx_sum_forward``) instead of ``zeros_like`` + N in-place adds.
"""
from __future__ import annotations

import abc
from collections.abc import Sequence

import torch
from torch import Tensor

from torchfx_amd.effect import FX


class AbstractFilter(FX, abc.ABC):
    """A filter whose coefficients are designed lazily, once ``fs`` is known."""

    @abc.abstractmethod
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)

    @property
    def _has_computed_coeff(self) -> bool:
        # SOS-based filters first, then (b, a) style ones (__base.py:... same order)
        if getattr(self, "_sos", None) is not None:
            return True
        if hasattr(self, "b") and hasattr(self, "a"):
            return self.b is not None and self.a is not None
        return False

    @abc.abstractmethod
    def compute_coefficients(self) -> None: ...

    def __add__(self, other: "AbstractFilter") -> "ParallelFilterCombination":
        assert isinstance(other, AbstractFilter), "Can only add AbstractFilter instances"
        return ParallelFilterCombination(self, other)

    def __radd__(self, other: "AbstractFilter") -> "ParallelFilterCombination":
        assert isinstance(other, AbstractFilter), "Can only add AbstractFilter instances"
        return ParallelFilterCombination(other, self)


class ParallelFilterCombination(AbstractFilter):
    """``f1 + f2``: every branch filters the same input, outputs are summed
    (``__base.py:1019-1026``).  Setting ``fs`` propagates to branches that have none
    (``:1006-1012``)."""

    filters: Sequence[AbstractFilter]

    def __init__(self, *filters: AbstractFilter, fs: int | None = None) -> None:
        super().__init__()
        self.filters = filters
        self.fs = fs

    @property
    def _has_computed_coeff(self) -> bool:
        return all(f._has_computed_coeff for f in self.filters)

    @property
    def fs(self) -> int | None:
        return self._fs

    @fs.setter
    def fs(self, value: int | None) -> None:
        self._fs = value
        if value is None:
            return
        for f in self.filters:
            if getattr(f, "fs", 0) is None:
                f.fs = value

    def compute_coefficients(self) -> None:
        for f in self.filters:
            f.compute_coefficients()

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        from torchfx_amd import torchfx_ext

        branches = [f.forward(x) for f in self.filters]
        return torchfx_ext.sum_forward(branches)
