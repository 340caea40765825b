"""Chain step (4 biquad sections | 66 559 merged FIR taps) on 64 x 28.8 M float32: the cascade inside pass A
(tfx_sos_fft_conv_forward) against the staged cascade kernel + overlap-save and the float32 spectral fold.
Knobs through the environment (TFX_OLS_SOS_PAIRS / _STREAMS / _SLAB_MB ...).  usage: python tools/sos_ols_bench.py [reps] [what]"""
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402
from torchfx_amd import filter as F  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
what = sys.argv[2] if len(sys.argv) > 2 else "fused,staged,fold"
C, T = int(os.environ.get("BENCH_C", 64)), int(os.environ.get("BENCH_T", 28_800_000))
f1 = F.LoButterworth(2000, order=6, fs=48000)
f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
K = 66559
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((C, T), device="cuda", generator=g) * 2 - 1


def timed(fn, name):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"{name:28s} min {ts[0]:7.3f}  med {ts[len(ts) // 2]:7.3f} ms   ({8 * C * T / ts[len(ts) // 2] / 1e9 / 8:.3f} of 8 TB/s at 8 B/sample)", flush=True)


if "fused" in what:
    timed(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), "cascade in pass A")
if "staged" in what:
    def staged():
        y, _, _ = E.sos_forward(x, None, sos, None, None)
        return E.fft_conv_forward(y, k, (K - 1, 0))
    timed(staged, "cascade kernel + OLS")
if "ols" in what:
    timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), "OLS alone (66559 taps)")
if "check" in what:
    y = E.sos_fft_conv_forward(x, sos, k, (K - 1, 0))
    ys, _, _ = E.sos_forward(x, None, sos, None, None)
    ys = E.fft_conv_forward(ys, k, (K - 1, 0))
    print("max |fused - staged| =", float((y - ys).abs().max()), " max|y| =", float(ys.abs().max()))
if "sweep" in what:
    for streams in (2, 3, 4):
        for pairs in (32, 64, 128, 256, 480):
            os.environ["TFX_OLS_SOS_STREAMS"] = str(streams)
            os.environ["TFX_OLS_SOS_PAIRS"] = str(pairs)
            timed(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), f"fused streams={streams} pairs={pairs}")
if "sustained" in what:
    def burst(fn, n, name):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"{name:28s} {n:3d} steps back to back: {(time.perf_counter() - t0) * 1e3 / n:7.3f} ms / step", flush=True)
    def staged2():
        y, _, _ = E.sos_forward(x, None, sos, None, None)
        return E.fft_conv_forward(y, k, (K - 1, 0))
    for n in (5, 20, 60):
        burst(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), n, "cascade in pass A")
        burst(staged2, n, "cascade kernel + OLS")
        burst(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), n, "OLS alone")
if "wavepath" in what:
    from scipy.signal import firwin
    from torchfx_amd import Wave
    fir = F.FIR(firwin(1024, 5000, fs=48000))
    rev = F.FIR((lambda ir: (ir / np.abs(ir).sum()).astype(np.float32))(np.random.default_rng(0).standard_normal(65536) * np.exp(-np.arange(65536) / 8000.0)))
    def burst2(fn, n, name):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        print(f"{name:44s} {n:3d} steps back to back: {(time.perf_counter() - t0) * 1e3 / n:7.3f} ms / step", flush=True)
    plan = (Wave(x, 48000, device=x.device) | f1 | f2 | fir | rev).plan()
    print([type(m).__name__ for m in plan])
    kk = plan[0].fir.kernel.reshape(-1)
    for rep in range(2):
        burst2(lambda: (Wave(x, 48000, device=x.device) | f1 | f2 | fir | rev).ys, 20, "Wave(...).ys")
        burst2(lambda: plan[0](x), 20, "plan[0](x)")
        burst2(lambda: E.sos_fft_conv_forward(x, sos, kk, (kk.numel() - 1, 0)), 20, "op, the plan's merged taps")
        burst2(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), 20, "op, the tool's 66559 taps")
    xn = torch.randn((C, T), device="cuda"); xn.mul_(1.0 / float(xn.abs().max()))
    burst2(lambda: E.sos_fft_conv_forward(xn, sos, k, (K - 1, 0)), 20, "op, normal-distributed input")
    burst2(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), 20, "op, uniform input")
