#!/usr/bin/env python3
"""oracle/check_reference_binding.py -- the reference's Python over OUR compiled module (build container only).

INTEGRATION.md section 0 says a maintainer drops ``torchfx_amd/native/torchfx_ext*.so`` into the ``torchfx`` package.  This
script does that without copying anything: it installs our compiled module as ``torchfx.torchfx_ext`` in ``sys.modules``,
imports the REAL reference package from /root/reference/src and checks that

  1. ``torchfx._ops`` binds to our module (``from torchfx import torchfx_ext``, _ops.py:25) and reports the native
     path as available (``is_native_available``, tests/test_ops_dispatch.py:20-35 of the reference);
  2. the three entry points the reference calls exist with the reference's argument names, in its order
     (binding.cpp:83-96) -- read from the docstrings pybind11 generates;
  3. the reference's own call sequences run THROUGH our C++ on host tensors -- the module dispatches on ``x.is_cuda()``
     like ``binding.cpp:30-81``, its host branch is ``torchfx_amd/csrc/ext/host_branch.h`` -- and give the reference's
     numbers: BASELINE.json's cfg 1 literally (``LoButterworth(1000, order=4, fs=48000)`` on ``iir_cfg1.npz``: the
     float32 output bit for bit), the cfg-2 cascade section by section (``iir_cfg2_sections.npz``), the reference's
     ``Wave | iir | iir`` pipe, ``_ops.biquad_forward`` and ``_ops.delay_line_forward`` against closed forms.  (On a
     device tensor the same calls are what tests/test_gpu_boundary.py makes on the GPU box.)

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
    import numpy as np
    import scipy.signal as sg

    gold = os.path.join(ROOT, "tests", "golden")
    g1 = np.load(os.path.join(gold, "iir_cfg1.npz"))
    g2 = np.load(os.path.join(gold, "iir_cfg2_sections.npz"))

    def err(a, b):
        return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())
    # cfg 1 of BASELINE.json: the reference's own module class, its forward, our C++ underneath
    f = F.LoButterworth(1000, order=4, fs=48000)
    y = f(torch.from_numpy(g1["x"]))
    assert y.dtype == torch.float32 and np.array_equal(y.numpy(), g1["y"]), err(y.numpy(), g1["y"])
    print("  IIR.forward (cfg 1, LoButterworth order 4, 1 x 48000): bit-identical to the reference fixture")
    # the cfg-2 cascade, every section, through the reference's public op (_ops.py:119-176)
    cur = torch.from_numpy(g2["x"]).double()
    sos2 = torch.from_numpy(g2["sos"])
    for k in range(sos2.shape[0]):
        cur, _, _ = _ops.parallel_iir_forward(cur, sos2[k:k + 1], None, None)
        e = err(cur.numpy(), g2["y_sections"][k])
        assert e <= 2e-11 * max(1.0, float(np.abs(g2["y_sections"][k]).max())), (k, e)
    yf, sx, sy = _ops.parallel_iir_forward(torch.from_numpy(g2["x"]), sos2, None, None)
    assert err(yf.numpy(), g2["y_sections"][-1]) <= 2e-11 and err(sx.numpy(), g2["state_x"]) <= 2e-10 and err(sy.numpy(), g2["state_y"]) <= 2e-10
    print("  _ops.parallel_iir_forward (cfg-2 cascade): every section within 2e-11 of the fixture, states within 2e-10")
    # the reference's pipe: Wave | iir | iir  (wave.py:207-239 -> FusedSOSCascade -> our module)
    x = torch.from_numpy(g2["x"])
    f1, f2 = F.LoButterworth(2000, order=6), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
    yw = (torchfx.Wave(x, 48000) | f1 | f2).ys
    assert yw.dtype == torch.float32 and np.array_equal(yw.numpy(), g2["y"]), err(yw.numpy(), g2["y"])
    print("  Wave | LoButterworth-6 | ParametricEQ (cfg 2's chain): bit-identical to the reference fixture")
    # the other two entry points against closed forms
    xr = torch.randn(3, 4000, dtype=torch.float64)
    b, a = sg.butter(2, 0.2)
    yb, _, _ = _ops.biquad_forward(xr, torch.from_numpy(b), torch.from_numpy(a), None, None)
    assert err(yb.numpy(), sg.lfilter(b, a, xr.numpy(), axis=-1)) <= 1e-12
    yd = _ops.delay_line_forward(xr.float(), 100, 0.5, 0.3)
    ex = xr.float().clone(); ex[:, 100:] += 0.15 * xr.float()[:, :-100]
    assert err(yd.numpy(), ex.numpy()) <= 1e-6 and _ops.delay_line_forward(xr[:, :50], 100, 0.5, 0.3).shape == (3, 50)
    print("  _ops.biquad_forward == scipy.lfilter (1e-12), _ops.delay_line_forward == x + mix*decay*x[n-D]")
    print("check_reference_binding: ok -- the reference package binds to torchfx_amd/native/torchfx_ext and computes through it")
    return 0


if __name__ == "__main__":
    sys.exit(main())
