"""Developer sweep: cfg-4 / chain overlap-save WALL time (no event profiling) under sets of env knobs.
usage: python tools/ols_wall.py "A=1,B=2" "A=0" ...   ('' = defaults); K via OLS_WALL_TAPS (default 65536)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402

C, T, K = int(os.environ.get("OLS_WALL_C", 64)), int(os.environ.get("OLS_WALL_T", 28_800_000)), int(os.environ.get("OLS_WALL_TAPS", 65536))
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
base_env = dict(os.environ)
for spec in (sys.argv[1:] or [""]):
    os.environ.clear()
    os.environ.update(base_env)
    for kv in filter(None, spec.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    ts = []
    for rep in range(3):
        y = E.fft_conv_forward(x, k, (K - 1, 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            y = None
            y = E.fft_conv_forward(x, k, (K - 1, 0))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 8 * 1e3)
    print(f"[{spec or 'defaults'}] wall ms/step {ts[0]:.3f} {ts[1]:.3f} {ts[2]:.3f}", flush=True)
