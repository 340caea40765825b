"""Filter base classes: lazy coefficients, ``+`` (parallel sum).

Reference: ``src/torchfx/filter/__base.py`` -- ``AbstractFilter`` (:22-739, the parts with
behaviour: ``_has_computed_coeff``, ``__add__``/``__radd__``) and
``ParallelFilterCombination`` (:742-1026).  When every branch is an SOS filter the whole sum is ONE
launch of the cascade kernel in sum mode (``tfx_sos_bank_sum_forward``: the input tile is read once,
every branch runs on it and the outputs are accumulated in registers -- 8 B/sample instead of
N x 8 + (N + 1) x 4); otherwise the branches run one by one and one HIP kernel (``tfx_sum_forward``)
adds them instead of ``zeros_like`` + N in-place adds.  Both keep the reference's accumulation order.
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

    def _leaves(self) -> list:
        """Branches with a left-nested combination flattened: ``f1 + f2 + f3`` builds
        ``(f1 + f2) + f3``, whose float accumulation 0 + y1 + y2 + y3 is the same sequence of additions
        as the flat sum (a combination in any other position is kept as one branch)."""
        first = self.filters[0]
        head = first._leaves() if isinstance(first, ParallelFilterCombination) else [first]
        return head + list(self.filters[1:])

    def _sos_branches(self):
        """``(filters, SOS matrices)`` if ALL leaves are plain stateful SOS filters, else None."""
        leaves = self._leaves()
        if len(leaves) < 2 or len(leaves) > 32:
            return None
        soses = []
        for f in leaves:
            if not hasattr(f, "_sos") or not hasattr(f, "_state_x") or getattr(f, "fs", None) is None:
                return None
            if f._sos is None:
                f.compute_coefficients()
            soses.append(f._sos.detach().to("cpu", torch.float64))
        return leaves, soses

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        from torchfx_amd import torchfx_ext

        found = self._sos_branches() if x.dim() >= 1 and x.dtype in (torch.float32, torch.float64) else None
        if found is None:
            branches = [f.forward(x) for f in self.filters]
            return torchfx_ext.sum_forward(branches)
        filters, soses = found
        # one launch: pad the shorter cascades with identity sections, gather / scatter the branch states
        rows = x.reshape(-1, x.shape[-1])
        c, kmax = rows.shape[0], max(s.shape[0] for s in soses)
        ident = torch.tensor([[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]], dtype=torch.float64)
        banks = torch.stack([torch.cat([s, ident.expand(kmax - s.shape[0], 6)]) for s in soses])
        sx = sy = None
        if any(f._state_x is not None and f._state_x.shape[1] == c for f in filters):
            sx = torch.zeros((kmax, len(soses) * c, 2), dtype=torch.float64, device=rows.device)
            sy = torch.zeros_like(sx)
            for i, (f, s) in enumerate(zip(filters, soses)):
                if f._state_x is not None and f._state_x.shape[1] == c:     # else: fresh zeros (iir.py:136-138)
                    sx[: s.shape[0], i * c:(i + 1) * c] = f._state_x.to(rows.device)
                    sy[: s.shape[0], i * c:(i + 1) * c] = f._state_y.to(rows.device)
        y, nsx, nsy = torchfx_ext.sos_bank_sum_forward(rows, banks, sx, sy)
        for i, (f, s) in enumerate(zip(filters, soses)):
            f._state_x = nsx[: s.shape[0], i * c:(i + 1) * c]
            f._state_y = nsy[: s.shape[0], i * c:(i + 1) * c]
            if hasattr(f, "_stateful"):
                f._stateful = True
        return y.reshape(x.shape)
