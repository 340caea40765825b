"""Check of the float32-error estimate behind precision="auto" (sos.hip, f32_error_bound: a host replay of the
kernel's float32 arithmetic on 2^16 samples, x 2): for a set of cascades, the measured largest
|float32 recursion - float64 recursion| / max(1, max|y|) on 16 x 2.88 M uniform samples in [-1, 1] on the device
against the estimate (tfx_sos_plan_info).  Run on the GPU box; the table goes to profiles/."""
import os
import sys

import numpy as np
import scipy.signal as sg
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import filter as F  # noqa: E402
from torchfx_amd import torchfx_ext as E  # noqa: E402

FS = 48000


def sos_of(*fs):
    rows = []
    for f in fs:
        f.fs = FS
        f.compute_coefficients()
        rows.append(f._sos.numpy())
    return np.vstack(rows)


CASES = {
    "cfg2: LoButterworth-6 @2k | ParametricEQ 1k Q2 +3dB": sos_of(F.LoButterworth(2000, order=6), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)),
    "LoButterworth-2 @8k": sos_of(F.LoButterworth(8000, order=2)),
    "LoButterworth-4 @2k": sos_of(F.LoButterworth(2000, order=4)),
    "LoButterworth-8 @500": sos_of(F.LoButterworth(500, order=8)),
    "LoButterworth-4 @200": sos_of(F.LoButterworth(200, order=4)),
    "HiButterworth-2 @20": sos_of(F.HiButterworth(20, order=2)),
    "HiButterworth-4 @300": sos_of(F.HiButterworth(300, order=4)),
    "HiChebyshev1-4 @1k": sos_of(F.HiChebyshev1(1000, order=4)),
    "LoElliptic-6 @4k": sos_of(F.LoElliptic(4000, order=6)),
    "Notch 1k Q30": sos_of(F.Notch(1000, 30.0)),
    "ParametricEQ 100 Hz Q4 +12dB": sos_of(F.ParametricEQ(frequency=100, q=4.0, gain=12.0)),
    "HiShelving 3k +12dB | LoShelving 200 +6dB": sos_of(F.HiShelving(3000, q=0.7, gain=4.0), F.LoShelving(200, q=0.7, gain=2.0)),
    "BiquadBPF 1k Q5": sos_of(F.BiquadBPF(1000, 5.0)),
    "LinkwitzRiley-4 lo @2k": sos_of(F.LoLinkwitzRiley(2000, order=4)),
    "AllPass 2k Q2 x3": sos_of(F.AllPass(2000, 2.0), F.AllPass(500, 1.0), F.AllPass(8000, 0.7)),
}

g = torch.Generator(device="cuda").manual_seed(5)
x = (torch.rand(16, 2_880_000, device="cuda", generator=g) * 2 - 1).contiguous()
print(f"{'cascade':52s} {'estimate':>10s} {'measured':>10s} {'meas/est':>9s} {'max|y|':>8s}  auto")
worst = 0.0
for name, sos in CASES.items():
    info = E.sos_plan_info(sos)
    y64 = E.sos_forward(x, None, torch.from_numpy(sos), None, None, out_dtype=torch.float64, precision="f64")[0]
    y32 = E.sos_forward(x, None, torch.from_numpy(sos), None, None, precision="f32")[0]
    err = float((y32.double() - y64).abs().max()) / max(1.0, float(y64.abs().max()))     # relative to max(1, max|y|)
    ratio = err / info["f32_error_bound"]
    worst = max(worst, ratio)
    print(f"{name:52s} {info['f32_error_bound']:10.3e} {err:10.3e} {ratio:9.3f} {float(y64.abs().max()):8.3f}  {info['auto_precision']}")
print(f"largest measured / estimate = {worst:.3f}  (the estimate includes its calibration factor; it must stay <= 1)")
