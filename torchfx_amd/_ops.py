"""Op dispatch layer -- mirror of the reference's ``src/torchfx/_ops.py``.

Same public names and argument meaning (``biquad_forward``, ``parallel_iir_forward``,
``delay_line_forward``, ``is_native_available``, ``PARALLEL_SCAN_THRESHOLD``); the work
goes to the HIP backend (``torchfx_amd.torchfx_ext``).

What is NOT mirrored is the reference's full-signal ``x.to(float64)`` copy
(``_ops.py:95,149``): the HIP kernel reads float32 and does the float64 arithmetic in
registers.  The returned ``y`` is float64 by default, exactly like the reference's, unless
the caller passes ``out_dtype`` (``filter/iir.py`` passes the input dtype, which folds the
reference's ``out.to(x.dtype)`` of ``iir.py:176`` into the kernel's store).
"""
from __future__ import annotations

import logging
from typing import TYPE_CHECKING

import torch

from torchfx_amd import _lib
from torchfx_amd import torchfx_ext as _ext

if TYPE_CHECKING:
    from torch import Tensor

logger = logging.getLogger(__name__)

# Kept for API compatibility (tests/test_ops_dispatch.py:21-23 of the reference asserts 2048).
# Informational only there as well: the reference's real switch is hard-coded in
# cuda/parallel_scan.cu:328.  The HIP kernel has no such switch.
PARALLEL_SCAN_THRESHOLD = 2048


def is_native_available() -> bool:
    """True when the HIP extension can be loaded (reference: ``_ops.py:37-54``)."""
    try:
        _lib.load()
        return True
    except (RuntimeError, OSError):
        return False


def _wide(x: Tensor, out_dtype: torch.dtype | None):
    """Signals that are neither float32 nor float64 (float16 / bfloat16 / integers).  The reference upcasts every
    signal to float64 and casts the result back (``_ops.py:95,149``, ``iir.py:176``); the kernels take float32 and
    float64 only, so 16-bit floats travel as float32 (exact), integers as float64, the result is produced in float64
    and rounded ONCE to the requested dtype -- the same single rounding as the reference's ``out.to(x.dtype)``.
    Returns (signal for the kernel, dtype for the kernel's store, dtype to cast the result to or None)."""
    want = x.dtype if out_dtype is None else out_dtype
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32 if x.dtype in (torch.float16, torch.bfloat16) else torch.float64)
    if want in (torch.float32, torch.float64):
        return x, want, None
    return x, torch.float64, want


def biquad_forward(
    x: Tensor,
    b: Tensor,
    a: Tensor,
    state_x: Tensor | None,
    state_y: Tensor | None,
    *,
    a1_f64: float | None = None,
    a2_f64: float | None = None,
    out_dtype: torch.dtype | None = torch.float64,
    precision=None,
) -> tuple[Tensor, Tensor, Tensor]:
    """Single biquad (reference: ``_ops.py:57-116``).  Returns ``(y, new_sx, new_sy)``."""
    if a1_f64 is None or a2_f64 is None:
        a_host = a.detach().to(device="cpu", dtype=torch.float64)
        a1_f64 = float(a_host[1])
        a2_f64 = float(a_host[2])
    x, store, cast = _wide(x, out_dtype)
    y, sx, sy = _ext.biquad_forward(x, b, a1_f64, a2_f64, state_x, state_y, out_dtype=store, precision=precision)
    return (y if cast is None else y.to(cast)), sx, sy


def parallel_iir_forward(
    x: Tensor,
    sos: Tensor,
    state_x: Tensor | None,
    state_y: Tensor | None,
    *,
    sos_cpu: Tensor | None = None,
    out_dtype: torch.dtype | None = torch.float64,
    precision=None,
    epilogue=None,
) -> tuple[Tensor, Tensor, Tensor]:
    """K-section SOS cascade (reference: ``_ops.py:119-176``).  ``None`` states mean zeros
    (``:144-147``).  Returns ``(y, new_state_x [K,C,2], new_state_y [K,C,2])``."""
    if sos_cpu is None:
        sos_cpu = sos.detach().to(dtype=torch.float64, device="cpu") if sos.is_cuda else sos
    x, store, cast = _wide(x, out_dtype)
    if epilogue is not None:
        y, sx, sy = _ext.sos_forward(x, sos, sos_cpu, state_x, state_y, out_dtype=store, precision=precision, epilogue=epilogue)
    else:
        y, sx, sy = _ext.sos_forward(x, sos, sos_cpu, state_x, state_y, out_dtype=store, precision=precision)
    return (y if cast is None else y.to(cast)), sx, sy


def delay_line_forward(x: Tensor, delay_samples: int, decay: float, mix: float) -> Tensor:
    """Reference: ``_ops.py:179-191``."""
    return _ext.delay_line_forward(x, delay_samples, decay, mix)
