"""Which block size serves 8193 ... 65 535 taps best on 64 x 2.88 M float32 (three-pass pipeline)?  TFX_FFT_LOG2N forces 16 / 18 / 20.
usage: python tools/experiments/midtaps_blocks.py"""
import os
os.environ["TFX_ENV_DYNAMIC"] = "1"
import sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E

C, T = int(os.environ.get("BENCH_C", 64)), int(os.environ.get("BENCH_T", 2_880_000))
x = torch.rand((C, T), device="cuda") * 2 - 1


def timed(K, lg):
    if lg: os.environ["TFX_FFT_LOG2N"] = str(lg)
    else: os.environ.pop("TFX_FFT_LOG2N", None)
    k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / (K / 6.0))
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    try:
        info = E.ols_plan_info(K, T, (K - 1, 0))
        fn = lambda: E.fft_conv_forward(x, kf, (K - 1, 0))
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
        ts.sort()
        return f"{info['path']}/2^{info['N'].bit_length() - 1} {ts[3]:.3f}"
    except Exception as e:
        return "n/a"


for K in [int(v) for v in os.environ.get("BENCH_KS", "8193,10000,12288,16384,20000,24576,32768,49152,65535").split(",")]:
    print(f"K={K:6d}: default {timed(K, 0):22s} 2^16 {timed(K, 16):22s} 2^18 {timed(K, 18):22s} 2^20 {timed(K, 20):22s}", flush=True)
