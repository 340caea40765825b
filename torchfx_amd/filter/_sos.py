"""Cascade tables and their stateful streams -- the two objects the fusion planner works with.

``CascadeTable`` is what the HIP cascade kernel consumes: a host-side float64 ``[K, 6]`` coefficient
matrix plus the sample rate it was designed for.  ``CascadeTable.gather`` is the single place where
several IIR / Biquad members become one table -- used by the ``Wave`` planner (``wave.py``), by
``FusedSOSCascade`` and by the parallel-sum path; optional per-member linear gains are folded into
the numerator of the member's first section (``H(g x) = g H(x)``).

``CascadeStream`` pairs a table with the DF1 state carried between calls (``[K, rows, 2]`` float64,
layout of ``src/torchfx/_csrc/cpu/iir_cpu.cpp:125-130``) and applies the shape / dtype / state rules of
the reference's ``_sos_cascade_forward`` (``src/torchfx/filter/iir.py:84-184``) through
``iir._sos_cascade_forward``.
"""
from __future__ import annotations

import dataclasses
import typing as tp

import torch
from torch import Tensor


@dataclasses.dataclass(frozen=True)
class CascadeTable:
    sos: Tensor                      # [K, 6] float64, host
    fs: int | None

    @property
    def sections(self) -> int:
        return int(self.sos.shape[0])

    @staticmethod
    def of(member) -> Tensor:
        """The member's designed ``[K, 6]`` rows (designing them now if that is still pending)."""
        if not hasattr(member, "_sos"):
            raise TypeError(f"Expected filter with SOS coefficients, got {type(member).__name__}")
        if member._sos is None:
            if getattr(member, "fs", None) is None:
                raise ValueError(f"Filter {type(member).__name__} has no sampling frequency set; "
                                 "give it an fs before fusing.")
            member.compute_coefficients()
        return member._sos.detach().to("cpu", torch.float64)

    @classmethod
    def gather(cls, members: tp.Sequence, gains: tp.Sequence[float] | None = None) -> "CascadeTable":
        if len(members) == 0:
            raise ValueError("a fused cascade needs at least one IIR filter")
        blocks = [cls.of(m) for m in members]
        rates = {m.fs for m in members if getattr(m, "fs", None) is not None}
        if len(rates) > 1:
            lo, hi = sorted(rates)[0], sorted(rates)[-1]
            raise ValueError(f"Cannot fuse filters designed for different sample rates: {lo} vs {hi}")
        if gains is not None:
            for i, g in enumerate(gains):
                if g != 1.0:
                    blocks[i] = blocks[i].clone()
                    blocks[i][0, :3] *= g
        return cls(torch.cat(blocks, dim=0).contiguous(), next(iter(rates), None))


class CascadeStream:
    """A table plus the state it carries from one chunk to the next."""

    __slots__ = ("table", "sx", "sy")

    def __init__(self, table: CascadeTable) -> None:
        self.table = table
        self.sx: Tensor | None = None
        self.sy: Tensor | None = None

    @property
    def fresh(self) -> bool:
        return self.sx is None

    def reset(self) -> None:
        self.sx = self.sy = None

    def __call__(self, x: Tensor, epilogue=None) -> Tensor:
        from torchfx_amd.filter.iir import _sos_cascade_forward

        y, _, self.sx, self.sy = _sos_cascade_forward(x, self.table.sos, None, self.sx, self.sy, epilogue)
        return y
