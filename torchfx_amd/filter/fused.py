"""``FusedSOSCascade``: several IIR / Biquad modules behind ONE launch of the HIP cascade kernel.

Public surface of the reference class (``src/torchfx/filter/fused.py:19-132``): constructor from
filters, ``from_chain``, ``fs``, ``_sos`` (``[sum K, 6]`` float64), ``_state_x`` / ``_state_y``,
``reset_state``, ``move_coeff``, stateful ``forward``.  Internally it is a thin module around the
planner's two objects (``filter/_sos.py``): a ``CascadeTable`` gathered from the members and the
``CascadeStream`` that carries the DF1 state -- the same objects ``Wave.plan()`` builds, so a planned
pipeline and a hand-made ``FusedSOSCascade`` run the identical code.  On this backend fusion is
where the chain pays off: all K_total sections run in one kernel that reads the signal once and
writes it once.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from torchfx_amd.filter._sos import CascadeStream, CascadeTable


class FusedSOSCascade(nn.Module):
    def __init__(self, *filters) -> None:
        super().__init__()
        self._stream = CascadeStream(CascadeTable.gather(filters))

    @classmethod
    def from_table(cls, table: CascadeTable) -> "FusedSOSCascade":
        """Planner entry: wrap an already gathered (possibly gain-folded) table."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self._stream = CascadeStream(table)
        return self

    @classmethod
    def from_chain(cls, chain: nn.Module) -> "FusedSOSCascade":
        """Every SOS filter found directly in an ``nn.Sequential`` (or a lone SOS filter)."""
        from torchfx_amd.filter.biquad import Biquad
        from torchfx_amd.filter.iir import IIR

        if isinstance(chain, (IIR, Biquad)):
            return cls(chain)
        if not isinstance(chain, nn.Sequential):
            raise TypeError(f"Expected nn.Sequential or IIR/Biquad, got {type(chain).__name__}")
        picked = tuple(m for m in chain.children() if isinstance(m, (IIR, Biquad)))
        if not picked:
            raise ValueError("No IIR/Biquad filters found in chain to fuse")
        return cls(*picked)

    # ---- the attributes callers and tests of the reference class read -------------------------
    @property
    def _sos(self) -> Tensor:
        return self._stream.table.sos

    @property
    def fs(self) -> int | None:
        return self._stream.table.fs

    @property
    def _state_x(self) -> Tensor | None:
        return self._stream.sx

    @_state_x.setter
    def _state_x(self, value: Tensor | None) -> None:
        self._stream.sx = value

    @property
    def _state_y(self) -> Tensor | None:
        return self._stream.sy

    @_state_y.setter
    def _state_y(self, value: Tensor | None) -> None:
        self._stream.sy = value

    @property
    def _num_sections(self) -> int:                # read by the reference's tests (tests/test_fused.py:60-102)
        return self._stream.table.sections

    @property
    def _stateful(self) -> bool:                   # "has run since the last reset" (tests/test_fused.py:176-184)
        return not self._stream.fresh

    def move_coeff(self, device) -> None:
        """API parity only: the canonical table stays on the host, where the HIP op reads it."""

    def reset_state(self) -> None:
        self._stream.reset()

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None) -> Tensor:
        return self._stream(x, epilogue)
