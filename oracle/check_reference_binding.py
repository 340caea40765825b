#!/usr/bin/env python3
"""oracle/check_reference_binding.py -- the reference's Python over OUR compiled module (build container only).

INTEGRATION.md section 0 says a maintainer drops ``torchfx_amd/native/torchfx_ext*.so`` into the ``torchfx`` package.  This
script does that without copying anything: it installs our compiled module as ``torchfx.torchfx_ext`` in ``sys.modules``,
imports the REAL reference package from /root/reference/src and checks that

  1. ``torchfx._ops`` binds to our module (``from torchfx import torchfx_ext``, _ops.py:25) and reports the native
     path as available (``is_native_available``, tests/test_ops_dispatch.py:20-35 of the reference);
  2. the three entry points the reference calls exist with the reference's argument names, in its order
     (binding.cpp:83-96) -- read from the docstrings pybind11 generates;
  3. the reference's own call sequences reach our C++: ``_ops.parallel_iir_forward`` / ``_ops.biquad_forward`` /
     ``_ops.delay_line_forward`` and ``IIR.forward`` through ``filter/iir.py::_sos_cascade_forward`` on HOST tensors end
     in our explicit "no CPU path" RuntimeError (there is no GPU in the build container; on a device tensor the same
     calls are what tests/test_gpu_boundary.py makes on the GPU box, against fixtures generated from the reference).

Nothing of the reference is copied; nothing here runs on the GPU box (/root/reference does not exist there).
Exit code 0 = all checks passed.
"""
from __future__ import annotations

import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SRC = "/root/reference/src"


def main() -> int:
    if not os.path.isdir(REF_SRC):
        print("check_reference_binding: /root/reference is not here (GPU box?) -- nothing to do")
        return 0
    sys.path.insert(0, ROOT)
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))      # imported unconditionally at realtime/stream.py:28
    import torch

    from torchfx_amd import native

    ours = native.load()
    sys.modules["torchfx.torchfx_ext"] = ours                                # "dropped into the package"
    sys.path.insert(0, REF_SRC)
    import torchfx
    from torchfx import _ops
    from torchfx import filter as F

    assert torchfx.__file__.startswith(REF_SRC), torchfx.__file__
    assert _ops._ext is ours, "torchfx._ops did not bind to the HIP module"
    assert _ops.is_native_available() is True
    for name, args in (("biquad_forward", ["x", "b", "a1", "a2", "state_x", "state_y"]),
                       ("sos_forward", ["x", "sos", "sos_cpu", "state_x", "state_y"]),
                       ("delay_line_forward", ["x", "delay_samples", "decay", "mix"])):
        doc = getattr(_ops._ext, name).__doc__
        sig = doc.splitlines()[0]
        got = [a.split(":")[0].strip() for a in sig[sig.index("(") + 1: sig.index(")")].split(",")]
        assert got == args, (name, got)
    x = torch.randn(2, 256)
    sos = torch.tensor([[0.2, 0.4, 0.2, 1.0, -0.3, 0.1], [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]], dtype=torch.float64)
    calls = {
        "_ops.parallel_iir_forward": lambda: _ops.parallel_iir_forward(x, sos, None, None),
        "_ops.biquad_forward": lambda: _ops.biquad_forward(x, sos[0, :3], sos[0, 3:], None, None),
        "_ops.delay_line_forward": lambda: _ops.delay_line_forward(x, 10, 0.5, 0.3),
        "IIR.forward": lambda: F.LoButterworth(1000, order=4, fs=48000)(x),
        "Wave | iir | iir": lambda: (torchfx.Wave(x, 48000) | F.LoButterworth(1000, order=4) | F.HiButterworth(100, order=2)).ys,
    }
    for what, fn in calls.items():
        try:
            fn()
        except RuntimeError as e:
            assert "no CPU path" in str(e), (what, str(e))
            print(f"  {what}: reached the HIP module ({str(e).splitlines()[0][:90]} ...)")
        else:
            raise AssertionError(f"{what}: did not reach the HIP module")
    print("check_reference_binding: ok -- the reference package binds to torchfx_amd/native/torchfx_ext and its call paths end in it")
    return 0


if __name__ == "__main__":
    sys.exit(main())
