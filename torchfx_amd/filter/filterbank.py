"""``LogFilterBank``: N logarithmically spaced constant-Q band-pass biquads -> ``[N, C, T]``.

Reference behaviour (``src/torchfx/filter/filterbank.py:19-185``): centre frequencies
``f_min * (f_max / f_min) ** (k / (N - 1))``, one ``BiquadBPF`` per band kept in ``filters``, the band
outputs stacked on a new leading axis, every band carrying its own DF1 state.  There the forward is a
Python loop of N filter calls plus ``torch.stack`` (8N B/sample); here the bank is ONE launch of the
cascade kernel in filter-bank mode (``tfx_sos_bank_forward``): the input row is read once and every
band writes its own output row -- (4 + 4N) B/sample.  SURVEY.md 8(f) rank 2.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from torchfx_amd.filter._base import AbstractFilter
from torchfx_amd.filter._sos import CascadeTable
from torchfx_amd.filter.biquad import BiquadBPF


def log_spaced(f_min: float, f_max: float, n: int) -> list[float]:
    """``n`` frequencies from ``f_min`` to ``f_max``, equal steps in octaves."""
    step = math.log2(f_max / f_min) / (n - 1)
    return [f_min * 2.0 ** (step * k) for k in range(n)]


class LogFilterBank(AbstractFilter):
    def __init__(self, n_bands: int, f_min: float = 20.0, f_max: float = 20000.0, q: float = 1.414,
                 fs: int | None = None) -> None:
        super().__init__()
        assert n_bands >= 2, "n_bands must be >= 2"
        assert 0 < f_min < f_max, "f_min must be positive and less than f_max"
        self.n_bands, self.f_min, self.f_max, self.q = n_bands, f_min, f_max, q
        self.filters = [BiquadBPF(cutoff=fc, q=q) for fc in log_spaced(f_min, f_max, n_bands)]
        self.a = self.b = None            # (b, a) are only the "designed" markers of AbstractFilter
        self._fs = None
        self.fs = fs

    @property
    def fs(self) -> int | None:
        return self._fs

    @fs.setter
    def fs(self, value: int | None) -> None:
        self._fs = value
        for band in self.filters if value is not None else ():
            band.fs = value

    @property
    def center_frequencies(self) -> list[float]:
        return [band.cutoff for band in self.filters]

    def compute_coefficients(self) -> None:
        self._band_tables()
        self.a = self.b = torch.ones(1)

    def _band_tables(self) -> Tensor:
        """``[N, 1, 6]`` host tables of all bands (designing whatever is still pending)."""
        for band in self.filters:
            if band.fs is None:
                band.fs = self._fs
        return torch.stack([CascadeTable.of(band) for band in self.filters])

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """``[T]`` / ``[C,T]`` / ``[B,C,T]`` -> ``[N, *x.shape]``; chunked calls are continuous because
        each band's state lives on its member filter, as in the reference."""
        from torchfx_amd import torchfx_ext

        if self._fs is None:
            raise ValueError("Sample rate (fs) must be set before filtering.")
        rows = x.reshape(-1, x.shape[-1])
        c = rows.shape[0]
        carried = all(b._state_x is not None and b._state_x.shape[1] == c for b in self.filters)
        sx = torch.cat([b._state_x.to(rows.device) for b in self.filters], dim=1) if carried else None
        sy = torch.cat([b._state_y.to(rows.device) for b in self.filters], dim=1) if carried else None
        y, nsx, nsy = torchfx_ext.sos_bank_forward(rows, self._band_tables(), sx, sy, out_dtype=x.dtype)
        for i, band in enumerate(self.filters):
            band._state_x, band._state_y = nsx[:, i * c:(i + 1) * c], nsy[:, i * c:(i + 1) * c]
        return y.reshape(self.n_bands, *x.shape)
