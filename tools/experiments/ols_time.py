"""Time fft_conv_forward (K taps, 64 x 2.88 M float32) -- used with probe builds of the library.  usage: ols_time.py [K]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C, T = 64, 2_880_000
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 200.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1
fn = lambda: E.fft_conv_forward(x, k, (K - 1, 0))
for _ in range(3): fn()
torch.cuda.synchronize()
ts = []
for _ in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): fn()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
ts.sort()
print(f"K={K}: min {ts[0]:.4f} med {ts[4]:.4f} max {ts[-1]:.4f} ms", flush=True)
