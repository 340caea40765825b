"""``FilterChain`` -- flat ``nn.Sequential`` built by ``|`` (reference: ``src/torchfx/chain.py:8-44``)."""
from __future__ import annotations

from torch import nn


class FilterChain(nn.Sequential):
    """``(f1 | f2) | f3`` and ``f1 | (f2 | f3)`` both give ``FilterChain(f1, f2, f3)``."""

    def __init__(self, *modules: nn.Module) -> None:
        steps: list[nn.Module] = []
        for m in modules:
            steps.extend(m.children() if isinstance(m, FilterChain) else [m])
        super().__init__(*steps)

    def __or__(self, other: nn.Module) -> "FilterChain":
        if not isinstance(other, nn.Module):
            return NotImplemented
        return FilterChain(*self.children(), other)

    def __ror__(self, other: object):
        return NotImplemented
