"""``FX`` base class -- the only part of the reference's ``effect.py`` on the filter hot
path (``src/torchfx/effect.py:253-258``: the ``|`` operator that builds a ``FilterChain``).
Gain / Normalize / Reverb / Delay are out of scope."""
from __future__ import annotations

import abc

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
