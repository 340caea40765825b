"""``LogFilterBank``: N logarithmically spaced constant-Q band-pass biquads -> ``[N, C, T]``.

Reference: ``src/torchfx/filter/filterbank.py:19-185`` (a Python loop of N ``BiquadBPF``
forwards + ``torch.stack``).  Here the N bands run in ONE launch of the cascade kernel in
filter-bank mode (``tfx_sos_bank_forward``): the input is read from HBM once, every band writes
its own output rows -- (4 + 4N) B/sample instead of 8N.  SURVEY.md 8(f) rank 2.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from torchfx_amd.filter._base import AbstractFilter
from torchfx_amd.filter.biquad import BiquadBPF


class LogFilterBank(AbstractFilter):
    def __init__(self, n_bands: int, f_min: float = 20.0, f_max: float = 20000.0, q: float = 1.414,
                 fs: int | None = None) -> None:
        super().__init__()
        assert n_bands >= 2, "n_bands must be >= 2"
        assert 0 < f_min < f_max, "f_min must be positive and less than f_max"
        self.n_bands, self.f_min, self.f_max, self.q = n_bands, f_min, f_max, q
        self._fs = fs
        octaves = math.log2(f_max / f_min)
        self._center_freqs = [f_min * (2.0 ** (k * octaves / (n_bands - 1))) for k in range(n_bands)]
        self.filters = [BiquadBPF(cutoff=f, q=q, fs=fs) for f in self._center_freqs]
        self.a: Tensor | None = None
        self.b: Tensor | None = None

    @property
    def fs(self) -> int | None:
        return self._fs

    @fs.setter
    def fs(self, value: int | None) -> None:
        self._fs = value
        if value is not None:
            for f in self.filters:
                f.fs = value

    @property
    def center_frequencies(self) -> list[float]:
        return list(self._center_freqs)

    def compute_coefficients(self) -> None:
        for f in self.filters:
            f.compute_coefficients()
        self.a = torch.tensor([1.0])
        self.b = torch.tensor([1.0])

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """``[T]`` / ``[C,T]`` / ``[B,C,T]`` -> ``[N, *x.shape]``.  Each band keeps its own DF1 state
        (on the member ``BiquadBPF`` objects, like the reference), so chunked calls are continuous."""
        from torchfx_amd import torchfx_ext

        if self._fs is None:
            raise ValueError("Sample rate (fs) must be set before filtering.")
        for f in self.filters:
            if f.fs is None:
                f.fs = self._fs
            if f._sos is None:
                f.compute_coefficients()
        rows = x.reshape(-1, x.shape[-1])
        c = rows.shape[0]
        banks = torch.stack([f._sos for f in self.filters])              # [N, 1, 6] host
        sx = sy = None
        if all(f._state_x is not None and f._state_x.shape[1] == c for f in self.filters):
            sx = torch.cat([f._state_x.to(rows.device) for f in self.filters], dim=1)
            sy = torch.cat([f._state_y.to(rows.device) for f in self.filters], dim=1)
        y, nsx, nsy = torchfx_ext.sos_bank_forward(rows, banks, sx, sy, out_dtype=x.dtype)
        for i, f in enumerate(self.filters):
            f._state_x = nsx[:, i * c:(i + 1) * c]
            f._state_y = nsy[:, i * c:(i + 1) * c]
        return y.reshape(self.n_bands, *x.shape)
