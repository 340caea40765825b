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


class CascadeFIR(nn.Module):
    """Planner product (``Wave.plan()``, ``fuse_recursive``): a FRESH SOS cascade followed by an FFT-mode FIR, run as ONE
    overlap-save pipeline in the reference's own arithmetic -- float64 DF1 recursion from zero state
    (``_ops.py:119-176`` -> ``iir_cpu.cpp:64-159``), the downcast to the signal's float32 (``iir.py:84-184``), then
    ``fft_conv1d`` (``fir.py:552-555`` -> ``_fftconv.py:70-141``) -- where the recursion runs inside the transform's forward
    column pass (``tfx_sos_fft_conv_forward``) instead of as a pass over the signal of its own.  Stateless like the
    ``FusedSOSCascade`` the reference builds per materialisation (``wave.py:221-233``: its state is dropped with it).

    Rows the kernel does not serve (another dtype, a length that is not a multiple of 32, a misaligned view, a host tensor)
    run the two steps staged -- same arithmetic, two launches."""

    MIN_PAIRS = 256          # frame pairs below which the staged pair of launches is the faster plan (profiles/r05_experiments.txt section 11)

    def __init__(self, table: CascadeTable, fir: nn.Module) -> None:
        super().__init__()
        self._table, self.fir = table, fir
        self._planner_built = True

    @property
    def _sos(self) -> Tensor:
        return self._table.sos

    @property
    def fs(self) -> int | None:
        return self._table.fs

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None, return_sections: bool = False):
        from torchfx_amd import torchfx_ext

        if x.ndim not in (1, 2, 3):
            raise ValueError("Input must be of shape [T], [C, T], or [B, C, T]")
        taps = self.fir.kernel.reshape(-1)
        k = int(taps.numel())
        rows = x.reshape(-1, x.shape[-1])
        info = None
        if x.is_cuda and x.dtype == torch.float32 and rows.is_contiguous() and rows.data_ptr() % 16 == 0:
            info = torchfx_ext.sos_fft_conv_plan_info(int(rows.shape[-1]), self._table.sos, k, (k - 1, 0))
        # the recursion pass runs one workgroup per frame PAIR for a whole frame: it needs a few hundred pairs in flight (64 ch x 600 s:
        # 480); a small batch is faster as two launches (cascade kernel, then the plain pipeline) -- unless sections are asked for
        if info is not None and (return_sections or int(rows.shape[0]) * info["F"] >= 2 * self.MIN_PAIRS):
            out = torchfx_ext.sos_fft_conv_forward(rows, self._table.sos, taps, (k - 1, 0), return_sections=return_sections,
                                                   epilogue=epilogue)
            if return_sections:
                return out[0].reshape(x.shape), out[1]
            return out.reshape(x.shape)
        if return_sections:
            raise RuntimeError("CascadeFIR: section taps come from the fused pass only (float32 rows of a multiple of 32 samples)")
        y = CascadeStream(self._table)(x)                  # fresh state, like the FusedSOSCascade of one materialisation
        return self.fir(y, epilogue) if epilogue is not None else self.fir(y)
