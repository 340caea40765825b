"""PyTorch custom-op registration of the HIP backend (``torch.ops.torchfx_hip.*``).

north_star asks for the kernels to be "exposed to Python through PyTorch-ROCm custom ops": these
are thin ``torch.library`` registrations over ``torchfx_amd.torchfx_ext`` (ctypes -> C ABI), so
the ops show up in ``torch.ops``, compose with ``torch.compile`` graphs as opaque calls and carry
shape/dtype meta functions.  Forward-only, like the reference (every forward there is
``@torch.no_grad()``).  Import this module to register; nothing else depends on it.
"""
from __future__ import annotations

import torch
from torch import Tensor

from torchfx_amd import torchfx_ext as _E

_lib = torch.library.Library("torchfx_hip", "DEF")
_lib.define("sos_forward(Tensor x, Tensor sos_cpu, Tensor? state_x, Tensor? state_y) -> (Tensor, Tensor, Tensor)")
_lib.define("fir_direct_forward(Tensor x, Tensor kernel) -> Tensor")
_lib.define("fft_conv_forward(Tensor x, Tensor kernel, int pad_left, int pad_right) -> Tensor")
_lib.define("sos_bank_forward(Tensor x, Tensor sos_banks_cpu, Tensor? state_x, Tensor? state_y) -> (Tensor, Tensor, Tensor)")
_lib.define("sos_bank_sum_forward(Tensor x, Tensor sos_banks_cpu, Tensor? state_x, Tensor? state_y) -> (Tensor, Tensor, Tensor)")
_lib.define("gain_forward(Tensor x, float gain, bool clamp) -> Tensor")
_lib.define("normalize_forward(Tensor x, float peak, int mode, bool per_row) -> Tensor")


def _sos(x: Tensor, sos_cpu: Tensor, state_x, state_y):
    return _E.sos_forward(x, sos_cpu, sos_cpu, state_x, state_y)


def _sos_meta(x: Tensor, sos_cpu: Tensor, state_x, state_y):
    k, c = sos_cpu.shape[0], x.shape[0]
    st = x.new_empty((k, c, 2), dtype=torch.float64)
    return torch.empty_like(x), st, torch.empty_like(st)


def _fir(x: Tensor, kernel: Tensor) -> Tensor:
    return _E.fir_direct_forward(x, kernel)


def _fft(x: Tensor, kernel: Tensor, pad_left: int, pad_right: int) -> Tensor:
    return _E.fft_conv_forward(x, kernel, (pad_left, pad_right))


def _fft_meta(x: Tensor, kernel: Tensor, pad_left: int, pad_right: int) -> Tensor:
    return x.new_empty((x.shape[0], x.shape[1] + pad_left + pad_right - kernel.numel() + 1))


def _bank_meta(x: Tensor, banks: Tensor, state_x, state_y):
    nb, k, c = banks.shape[0], banks.shape[1], x.shape[0]
    st = x.new_empty((k, nb * c, 2), dtype=torch.float64)
    return x.new_empty((nb, c, x.shape[1])), st, torch.empty_like(st)


def _bank_sum_meta(x: Tensor, banks: Tensor, state_x, state_y):
    nb, k, c = banks.shape[0], banks.shape[1], x.shape[0]
    st = x.new_empty((k, nb * c, 2), dtype=torch.float64)
    return torch.empty_like(x), st, torch.empty_like(st)


_lib.impl("sos_bank_forward", lambda x, b, sx, sy: _E.sos_bank_forward(x, b, sx, sy), "CUDA")
_lib.impl("sos_bank_sum_forward", lambda x, b, sx, sy: _E.sos_bank_sum_forward(x, b, sx, sy), "CUDA")
_lib.impl("gain_forward", lambda x, gain, clamp: _E.gain_forward(x, gain, clamp), "CUDA")
_lib.impl("normalize_forward", lambda x, peak, mode, per_row: _E.normalize_forward(x, peak, mode, per_row), "CUDA")
_lib.impl("sos_bank_forward", _bank_meta, "Meta")
_lib.impl("sos_bank_sum_forward", _bank_sum_meta, "Meta")
_lib.impl("gain_forward", lambda x, gain, clamp: torch.empty_like(x), "Meta")
_lib.impl("normalize_forward", lambda x, peak, mode, per_row: torch.empty_like(x), "Meta")
_lib.impl("sos_forward", _sos, "CUDA")            # "CUDA" dispatch key == ROCm device tensors
_lib.impl("fir_direct_forward", _fir, "CUDA")
_lib.impl("fft_conv_forward", _fft, "CUDA")
_lib.impl("sos_forward", _sos_meta, "Meta")
_lib.impl("fir_direct_forward", lambda x, kernel: torch.empty_like(x), "Meta")
_lib.impl("fft_conv_forward", _fft_meta, "Meta")
