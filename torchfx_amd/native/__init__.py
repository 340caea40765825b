"""Loader of the compiled boundary module ``torchfx_ext`` (``torchfx_amd/csrc/ext/torchfx_ext.cpp``).

``torchfx_ext`` is a pybind11 torch extension with the reference's module name and its three entry
points (``src/torchfx/_csrc/binding.cpp:83-96``); importing it also registers every op of the backend
with the PyTorch dispatcher (``torch.ops.torchfx_hip.*``).  It is built in-tree by
``torchfx_amd/csrc/Makefile`` (``__graft_entry__.build()``) next to ``libtorchfx_hip.so``, whose C ABI it
calls.  A maintainer of the reference drops the same ``.so`` into the ``torchfx`` package directory,
where ``from torchfx import torchfx_ext`` finds it.

Missing build products raise ``RuntimeError`` -- there is no Python or CPU fallback.
"""
from __future__ import annotations

import importlib
import os
import threading

_lock = threading.Lock()
_mod = None


def load():
    """Import the compiled module (once) and return it; ``torch.ops.torchfx_hip`` is populated afterwards."""
    global _mod
    if _mod is not None:
        return _mod
    with _lock:
        if _mod is None:
            import torch  # noqa: F401  (libtorch must be loaded before the extension)

            try:
                _mod = importlib.import_module("torchfx_amd.native.torchfx_ext")
            except ImportError as e:
                raise RuntimeError(
                    "torchfx_amd: the compiled extension torchfx_amd/native/torchfx_ext*.so is missing or does not load "
                    f"({e}); build it with `python __graft_entry__.py` (make -C torchfx_amd/csrc).") from e
            if os.environ.get("TORCHFX_AMD_PREWARM", "0") == "1":
                _prewarm()
    return _mod


def _prewarm() -> None:
    """Opt-in (`TORCHFX_AMD_PREWARM=1`): start the library's one-time device set-up (load of its code object, internal
    streams) on a helper thread as soon as the module is loaded -- on the device that is CURRENT then.  Importing a library
    does not touch a device by default (round 4 did, and a rank that addresses its GPU as `cuda:3` without
    `torch.cuda.set_device` got a context on GPU 0): the planner starts the same helper with the wave's own device when it
    first plans a pipeline with an FFT-mode FIR (`Wave.plan`), and `torchfx_ext.prewarm(device)` does it on request."""
    import torch

    if not torch.cuda.is_available():
        return
    try:
        from torchfx_amd import _lib

        _lib.load().tfx_prewarm()
    except Exception:           # never a reason to fail an import
        pass


def ops():
    """``torch.ops.torchfx_hip`` with the library loaded."""
    import torch

    load()
    return torch.ops.torchfx_hip
