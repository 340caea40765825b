"""``FusedSOSCascade``: several IIR / Biquad modules as ONE ``sos_forward`` call.

Reference: ``src/torchfx/filter/fused.py:19-132`` -- concatenate the members' SOS matrices
into ``[sum K, 6]``, own state, same stateful forward as a single IIR.  On the HIP backend
this is where chain fusion pays off: the K_total sections run inside one kernel launch that
reads the signal once and writes it once.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from torchfx_amd.filter.biquad import Biquad
from torchfx_amd.filter.iir import IIR, _sos_cascade_forward


class FusedSOSCascade(nn.Module):
    def __init__(self, *filters: IIR | Biquad) -> None:
        super().__init__()
        if not filters:
            raise ValueError("FusedSOSCascade requires at least one IIR filter")
        rows: list[Tensor] = []
        fs_seen: int | None = None
        for f in filters:
            if not hasattr(f, "_sos"):
                raise TypeError(f"Expected filter with SOS coefficients, got {type(f).__name__}")
            if f._sos is None:
                if f.fs is None:
                    raise ValueError(
                        f"Filter {type(f).__name__} has no sampling frequency set. Set fs before fusing.")
                f.compute_coefficients()
            rows.append(f._sos)
            if f.fs is not None:
                if fs_seen is None:
                    fs_seen = f.fs
                elif f.fs != fs_seen:
                    raise ValueError(f"Cannot fuse filters with different sample rates: {fs_seen} vs {f.fs}")
        self._sos: Tensor = torch.cat(rows, dim=0).to(dtype=torch.float64)
        self._num_sections: int = self._sos.shape[0]
        self.fs: int | None = fs_seen
        self._sos_device_cache: Tensor | None = None
        self._state_x: Tensor | None = None
        self._state_y: Tensor | None = None
        self._stateful: bool = False

    @classmethod
    def from_chain(cls, chain: nn.Module) -> "FusedSOSCascade":
        """Fuse every IIR / Biquad child of an ``nn.Sequential`` (``fused.py:87-107``)."""
        if isinstance(chain, nn.Sequential):
            members = [m for m in chain if isinstance(m, (IIR, Biquad))]
        elif isinstance(chain, (IIR, Biquad)):
            members = [chain]
        else:
            raise TypeError(f"Expected nn.Sequential or IIR/Biquad, got {type(chain).__name__}")
        if not members:
            raise ValueError("No IIR/Biquad filters found in chain to fuse")
        return cls(*members)

    def move_coeff(self, device) -> None:
        """Kept for API parity (``fused.py:109-111``).  The canonical SOS stays on the host
        (the HIP op reads it there); only the dtype is normalised."""
        self._sos = self._sos.to(dtype=torch.float64)

    def reset_state(self) -> None:
        self._state_x = self._state_y = None
        self._stateful = False
        self._sos_device_cache = None

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        result, self._sos_device_cache, self._state_x, self._state_y = _sos_cascade_forward(
            x, self._sos.cpu(), self._sos_device_cache, self._state_x, self._state_y)
        self._stateful = True
        return result
