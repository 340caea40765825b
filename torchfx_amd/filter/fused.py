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

    Rows of any length and any float alignment are served (a row that is not a whole number of 128-byte lines moves its
    frame grid, ``row_shift`` in ``csrc/olsnative.hip``); non-finite samples poison the rest of their row exactly as in the
    staged pair.  What still takes the staged pair of launches -- same arithmetic, two launches -- is listed by
    :meth:`route`, which ``Wave.explain()`` prints: another dtype, a host tensor, or a batch below
    ``MIN_PAIRS`` frame pairs (there the two launches are the *faster* plan)."""

    MIN_PAIRS = 256          # frame pairs below which the staged pair of launches is the faster plan (profiles/r05_experiments.txt section 11)

    def __init__(self, table: CascadeTable, fir: nn.Module) -> None:
        super().__init__()
        self._table, self.fir = table, fir
        self._planner_built = True
        self.last_route: tuple[str, str] | None = None
        self._routes: dict = {}          # (shape, dtype, device type, sections asked) -> route(): the geometry query is host work per call otherwise

    @property
    def _sos(self) -> Tensor:
        return self._table.sos

    @property
    def fs(self) -> int | None:
        return self._table.fs

    def route(self, x: Tensor, return_sections: bool = False) -> tuple[str, str, dict | None]:
        """``("fused" | "staged", why, plan_info)`` for this tensor -- the decision :meth:`forward` takes, without running it."""
        key = (tuple(x.shape), x.dtype, x.device.type, bool(return_sections), self.MIN_PAIRS, self.fir.kernel._version)
        hit = self._routes.get(key)
        if hit is None:
            if len(self._routes) > 64:
                self._routes.clear()
            hit = self._routes[key] = self._route(x, return_sections)
        return hit

    def _route(self, x: Tensor, return_sections: bool) -> tuple[str, str, dict | None]:
        from torchfx_amd import torchfx_ext

        k = int(self.fir.kernel.numel())
        if x.ndim not in (1, 2, 3):
            return "staged", f"{x.ndim}-d input", None
        if not x.is_cuda:
            return "staged", "host tensor (the fused step is a device kernel)", None
        if x.dtype != torch.float32:
            return "staged", f"{x.dtype} signal (the fused step reads float32 rows)", None
        rows = x.reshape(-1, x.shape[-1])                     # a view where possible; the op makes strided rows contiguous itself
        info = torchfx_ext.sos_fft_conv_plan_info(int(rows.shape[-1]), self._table.sos, k, (k - 1, 0))
        if info is None:
            return "staged", "geometry not served (tfx_sos_fft_conv_supported)", None
        pairs = (int(rows.shape[0]) * info["F"] + 1) // 2
        if not return_sections and pairs < self.MIN_PAIRS:
            return "staged", (f"{pairs} frame pairs < {self.MIN_PAIRS}: the recursion pass needs hundreds of pairs in flight, "
                              "two launches are faster here"), info
        return "fused", f"recursion inside pass A, {1 << (info['N'].bit_length() - 1)}-point blocks, {pairs} frame pairs", info

    def extra_repr(self) -> str:
        r = self.last_route
        return f"sections={self._table.sections}, taps={int(self.fir.kernel.numel())}" + (f", route={r[0]} ({r[1]})" if r else "")

    @torch.no_grad()
    def forward(self, x: Tensor, epilogue=None, return_sections: bool = False):
        from torchfx_amd import torchfx_ext

        if x.ndim not in (1, 2, 3):
            raise ValueError("Input must be of shape [T], [C, T], or [B, C, T]")
        path, why, _ = self.route(x, return_sections)
        self.last_route = (path, why)
        if path == "fused":
            taps = self.fir.kernel.reshape(-1)
            k = int(taps.numel())
            rows = x.reshape(-1, x.shape[-1])
            out = torchfx_ext.sos_fft_conv_forward(rows, self._table.sos, taps, (k - 1, 0), return_sections=return_sections,
                                                   epilogue=epilogue)
            if return_sections:
                return out[0].reshape(x.shape), out[1]
            return out.reshape(x.shape)
        if return_sections:
            raise RuntimeError(f"CascadeFIR: section taps come from the fused pass only ({why})")
        y = CascadeStream(self._table)(x)                  # fresh state, like the FusedSOSCascade of one materialisation
        return self.fir(y, epilogue) if epilogue is not None else self.fir(y)
