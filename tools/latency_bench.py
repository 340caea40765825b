"""Developer tool: per-call host+device latency of the ops at streaming chunk sizes (small T)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402

dev = "cuda:0"
from scipy.signal import butter, firwin  # noqa: E402

sos = torch.from_numpy(butter(6, 2000 / 24000, output="sos"))
k1024 = firwin(1024, 5000, fs=48000).astype(np.float32)[::-1].copy()
k101 = firwin(101, 5000, fs=48000).astype(np.float32)[::-1].copy()


def lat(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for C, T in ((2, 512), (2, 4096), (2, 65536), (8, 4096), (64, 4096)):
    x = torch.randn(C, T, device=dev)
    sx = torch.zeros(3, C, 2, dtype=torch.float64, device=dev)
    sy = torch.zeros(3, C, 2, dtype=torch.float64, device=dev)
    r = {
        "sos(stateful)": lat(lambda: E.sos_forward(x, None, sos, sx, sy)),
        "fir101 direct": lat(lambda: E.fir_direct_forward(x, k101)),
        "fir1024 direct": lat(lambda: E.fir_direct_forward(x, k1024)),
        "fir1024 fft": lat(lambda: E.fft_conv_forward(x, k1024, (1023, 0))),
        "gain": lat(lambda: E.gain_forward(x, 0.5)),
        "torch x*0.5": lat(lambda: x * 0.5),
    }
    print(f"[{C} x {T}] " + "  ".join(f"{k} {v:6.1f} us" for k, v in r.items()), flush=True)
