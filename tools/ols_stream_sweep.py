"""Developer sweep: cfg-4 overlap-save wall time over (slab MB, internal streams).  usage: python tools/ols_stream_sweep.py"""
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402

C, T, K = 64, 28_800_000, 65536
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()


def wall(n=8):
    y = E.fft_conv_forward(x, k, (K - 1, 0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = None
        y = E.fft_conv_forward(x, k, (K - 1, 0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for slab in [int(v) for v in os.environ.get('SWEEP_SLABS', '8,16,32,64,128,256,1024').split(',')]:
    row = []
    for streams in [int(v) for v in os.environ.get('SWEEP_STREAMS', '2,3,4,6,8').split(',')]:
        os.environ["TFX_OLS_SLAB_MB"] = str(slab)
        os.environ["TFX_OLS_SLAB_MIN_MB"] = str(min(slab, 64))
        os.environ["TFX_OLS_PAIRS_PER_SLAB"] = str(max(1, slab // 8))
        os.environ["TFX_OLS_STREAMS"] = str(streams)
        row.append(f"{streams} streams {wall():6.3f}")
    print(f"slab {slab:5d} MB: " + " | ".join(row), flush=True)
