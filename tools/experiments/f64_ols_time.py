"""float64 overlap-save beyond 4096 taps: the float64 three-pass pipeline (olsnative64.hip) against the rocFFT path it replaces.
8 x 28.8 M float64, 65 536 taps.  usage: python tools/experiments/f64_ols_time.py"""
import os
os.environ["TFX_ENV_DYNAMIC"] = "1"
import sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E

C, T, K = 8, 28_800_000, 65536
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum())[::-1].copy())
x = torch.rand((C, T), device="cuda", dtype=torch.float64) * 2 - 1


def timed(name, reps=7):
    fn = lambda: E.fft_conv_forward(x, k, (K - 1, 0))
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); y = fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"{name:28s} {E.ols_plan_info(K, T, (K - 1, 0), torch.float64)['path']:7s} min {ts[0]:8.3f} med {ts[len(ts) // 2]:8.3f} ms "
          f"({16 * C * T / ts[len(ts) // 2] / 1e9 / 8:.3f} of 8 TB/s at 16 B/sample)", flush=True)
    return y


y1 = timed("float64 three passes")
for pairs in (4, 16, 32):
    os.environ["TFX_OLS64_PAIRS_PER_SLAB"] = str(pairs)
    timed(f"  slabs of {pairs} pairs")
os.environ.pop("TFX_OLS64_PAIRS_PER_SLAB")
os.environ["TFX_OLS_NATIVE64"] = "0"
y2 = timed("rocFFT path")
print("max |a - b| =", float((y1 - y2).abs().max()), " max|y| =", float(y2.abs().max()))
