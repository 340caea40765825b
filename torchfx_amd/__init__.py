"""torchfx_amd -- MI355X-native backend for the ``torchfx.filter`` hot path.

Top-level names follow the reference (``src/torchfx/__init__.py:12-23``): ``Wave``, ``FX``,
``FilterChain``, ``filter``, ``is_native_available``.
"""
from torchfx_amd import filter  # noqa: A004
from torchfx_amd._ops import is_native_available
from torchfx_amd.chain import FilterChain
from torchfx_amd.effect import FX, Gain, Normalize, Reverb
from torchfx_amd.wave import Wave

__all__ = ["FX", "FilterChain", "Gain", "Normalize", "Reverb", "Wave", "filter", "is_native_available"]
__version__ = "0.1.0"
